"""Whole-stack LightningDiT training route (-m gpu): csrc/dit_stack.hip's boundary / finalize / batched kernels against fp64 definitions and against the per-block
kernels they replace, `functional.DitStackFn` against the chain of `DitBlockFn`s, the batched per-sample Linears and weight transposes against their single forms
(diffusion/lightningdit/lightningdit.py:236-250 x depth; train_dmd.py:565-575, train_diffusion.py:290-297)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from test_oracle_dit import CFGS, build

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _rl2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("b,n,c", [(3, 64, 128), (16, 256, 1152), (5, 40, 144), (64, 32, 256)])
def test_boundary_kernel_equals_the_two_kernels_and_fp64(b, n, c):
    """dmvae_dit_boundary_bwd = rmsnorm_modulate_bwd's dx update followed by gated_residual_bwd on the updated gradient: dx to an f32 rounding of the per-block
    kernel's, dy the exact bf16 gate product of it, the deferred sums within f32 summation-order distance of the two kernels' and of fp64."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(b * 1000 + n + c)
    nl = 2
    x = (torch.randn(b, n, c, generator=g) * 2).to(DEV)
    w = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV)
    mod = (0.5 * torch.randn(nl, b, 6 * c, generator=g)).to(DEV).to(BF)
    da = torch.randn(b, n, c, generator=g).to(DEV).to(BF)
    y = torch.randn(b, n, c, generator=g).to(DEV).to(BF)
    dres = torch.randn(b, n, c, generator=g).to(DEV)
    # reference: the per-block kernels
    dmod_ref = torch.zeros(b, 6 * c, device=DEV)
    dx_ref = dres.clone()
    dw_ref = ops.rmsnorm_modulate_bwd_(dx_ref, da, x, w, mod[1], dmod_ref, 3 * c, 4 * c)
    dy_ref = ops.gated_residual_bwd(dx_ref, y, mod[1], dmod_ref, 2 * c)
    # the fused boundary in slot 3 (= norm2 + attention gate of block 1) of a two-block stack; the other slots get zero work through gate-only / norm-only calls
    S = ops.DitStackBwd(nl, b, n, c, 2 if c % 64 else c // 64, torch.device(DEV))
    S.part.zero_()
    S.qk_part.zero_()
    dx = dres.clone()
    dy = S.boundary(3, dx, da=da, x=x, w=w, mod=mod[1], scale_off=4 * c, y=y, gate_mod=mod[1], gate_off=2 * c)
    # the same expressions in another kernel: the compiler's fused-multiply-add choices may differ by an f32 rounding; dy is the bf16 gate product of the kernel's OWN dx
    assert rel_err(dx, dx_ref) < 2e-6 and torch.equal(dy, (mod[1][:, 2 * c:3 * c].float().unsqueeze(1) * dx).to(BF))
    assert (dy != dy_ref).float().mean().item() < 1e-3
    # gate-only (the top of the stack) and norm-only (the bottom) forms on the same data
    dx2 = dres.clone()
    dy2 = S.boundary(4, dx2, y=y, gate_mod=mod[1], gate_off=5 * c)
    assert torch.equal(dx2, dres) and torch.equal(dy2, (mod[1][:, 5 * c:].float().unsqueeze(1) * dres).to(BF))
    dx3 = dres.clone()
    assert S.boundary(0, dx3, da=da, x=x, w=w, mod=mod[0], scale_off=c) is None
    dx3_ref = dres.clone()
    dmod0 = torch.zeros(b, 6 * c, device=DEV)
    dw0_ref = ops.rmsnorm_modulate_bwd_(dx3_ref, da, x, w, mod[0], dmod0, 0, c)
    assert rel_err(dx3, dx3_ref) < 2e-6
    dmod = torch.zeros(nl, b, 6 * c, device=DEV, dtype=BF)
    d = c // S.heads
    norm_dws = [torch.zeros(c, device=DEV) for _ in range(2 * nl)]
    qd = [torch.zeros(d, device=DEV) for _ in range(nl)]
    kd = [torch.zeros(d, device=DEV) for _ in range(nl)]
    S.finalize(dmod, norm_dws, qd, kd)
    f = lambda t: t.float()
    # slot 3 -> block 1: shift_mlp, scale_mlp, gate_msa, norm2 weight
    assert _rl2(f(dmod[1][:, 3 * c:4 * c]), dmod_ref[:, 3 * c:4 * c]) < 4e-3 and _rl2(f(dmod[1][:, 4 * c:5 * c]), dmod_ref[:, 4 * c:5 * c]) < 4e-3
    assert _rl2(f(dmod[1][:, 2 * c:3 * c]), dmod_ref[:, 2 * c:3 * c]) < 4e-3
    assert rel_err(norm_dws[3], dw_ref) < 1e-5
    # slot 4 -> block 1's MLP gate; slot 0 -> block 0: shift_msa, scale_msa, norm1 weight
    assert _rl2(f(dmod[1][:, 5 * c:]), (dres.double() * y.double()).sum(1)) < 4e-3
    assert _rl2(f(dmod[0][:, :c]), dmod0[:, :c]) < 4e-3 and _rl2(f(dmod[0][:, c:2 * c]), dmod0[:, c:2 * c]) < 4e-3
    assert rel_err(norm_dws[0], dw0_ref) < 1e-5
    assert float(norm_dws[1].abs().max()) == 0.0 and float(norm_dws[2].abs().max()) == 0.0      # slots that did no work were zero-filled above
    assert float(torch.stack(qd + kd).abs().max()) == 0.0
    # fp64 of the gate sums on the UPDATED gradient
    assert _rl2(f(dmod[1][:, 2 * c:3 * c]), (dx_ref.double() * y.double()).sum(1)) < 4e-3


@pytest.mark.parametrize("nl,m,n,k", [(3, 16, 6912, 1152), (2, 64, 96, 128), (1, 5, 48, 40), (4, 33, 64, 1152)])
def test_batched_rows_linears(nl, m, n, k):
    """linear_rows_batched = L calls of linear_rows (bit for bit); linear_rows_wgrad_batched (MFMA) = dY^T X and the column sums in fp64."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(nl + m + n + k)
    x = torch.randn(m, k, generator=g).to(DEV).to(BF)
    ws = [(0.05 * torch.randn(n, k, generator=g)).to(DEV).to(BF) for _ in range(nl)]
    bs = [torch.randn(n, generator=g).to(DEV).to(BF) for _ in range(nl)]
    if k % 32 == 0 and n % 4 == 0:
        y = ops.linear_rows_batched(x, ws, bs)
        for i in range(nl):
            assert torch.equal(y[i], ops.linear_rows(x, ws[i], bs[i]))
        xl = torch.randn(nl, m, k, generator=g).to(DEV).to(BF)
        yl = ops.linear_rows_batched(xl, ws, None, out_f32=True)
        for i in range(nl):
            assert torch.equal(yl[i], ops.linear_rows(xl[i], ws[i], None, out_f32=True))
        if n % 32 == 0 and k % 8 == 0 and n % 32 == 0:
            # the input-gradient form: per-layer dy against the K-tile-major transposed copies
            dyl = torch.randn(nl, m, n, generator=g).to(DEV).to(BF)
            wts = [ops.linear_weight_t_kmajor(w) for w in ws]
            if ops.linear_rows_supported(m, k, n):
                dx = ops.linear_rows_batched(dyl, wts, None, out_f32=True)
                for i in range(nl):
                    assert torch.equal(dx[i], ops.linear_rows(dyl[i], wts[i], None, out_f32=True))
                    assert _rl2(dx[i], dyl[i].double() @ ws[i].double()) < 1e-5
    dy = torch.randn(nl, m, n, generator=g).to(DEV).to(BF)
    dws = [torch.full((n, k), 7.0, device=DEV) for _ in range(nl)]
    dbs = [torch.full((n,), 7.0, device=DEV) for _ in range(nl)]
    ops.linear_rows_wgrad_batched(dy, ops.rows_transposed(x), dws, dbs)
    for i in range(nl):
        assert rel_err(dws[i], dy[i].double().t() @ x.double()) < 2e-6
        assert rel_err(dbs[i], dy[i].double().sum(0)) < 2e-6
    # accumulate, and rerun bit identity
    again = [t.clone() for t in dws]
    ops.linear_rows_wgrad_batched(dy, ops.rows_transposed(x), again, None, accumulate=True)
    for i in range(nl):
        assert torch.equal(again[i], dws[i] + dws[i])
    dws2 = [torch.empty(n, k, device=DEV) for _ in range(nl)]
    ops.linear_rows_wgrad_batched(dy, ops.rows_transposed(x), dws2, None)
    assert all(torch.equal(a, b_) for a, b_ in zip(dws, dws2))


def test_batched_weight_transposes_equal_the_single_launches():
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(3)
    shapes = [(6912, 1152), (1152, 3072), (32, 8), (96, 200), (3456, 1152), (64, 64)]
    srcs = [torch.randn(n, k, generator=g).to(DEV).to(BF) for n, k in shapes]
    dsts = [torch.empty(n // 32, k, 32, device=DEV, dtype=BF) for n, k in shapes]
    ops.linear_weight_t_kmajor_batched(list(zip(srcs, dsts)))
    for s_, d_ in zip(srcs, dsts):
        assert torch.equal(d_, ops.linear_weight_t_kmajor(s_))
    ops.linear_weight_t_kmajor_batched(list(zip(srcs, dsts)))      # the cached table
    assert torch.equal(dsts[0], ops.linear_weight_t_kmajor(srcs[0]))


def test_transposed_shadow_follows_the_optimiser():
    """FlatParams.enable_transposed_shadow: `_bf_t` serves the copy the optimiser step refreshed in one launch -- equal to a fresh transpose of the updated weight after
    every step, and not served when something else changed the parameter."""
    from dmvae_amd import functional as Fn, ops
    from dmvae_amd.optim import FlatAdamWEMA, FlatParams
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(64, 96, device=DEV)), torch.nn.Parameter(torch.randn(17, device=DEV)), torch.nn.Parameter(torch.randn(96, 40, device=DEV)),
          torch.nn.Parameter(torch.randn(20, 24, device=DEV))]
    fp = FlatParams(ps, with_ema=False)
    fp.enable_bf16_shadow()
    fp.enable_transposed_shadow()
    opt = FlatAdamWEMA(fp, lr=0.05, warmup_steps=0, max_norm=0.0, weight_decay=0.0)
    assert hasattr(ps[0], "_dmvae_shadow_t") and hasattr(ps[2], "_dmvae_shadow_t") and not hasattr(ps[3], "_dmvae_shadow_t")
    for step in range(3):
        for p in (ps[0], ps[2]):
            t = Fn._bf_t(p)
            assert t.data_ptr() == p._dmvae_shadow_t.data_ptr()
            assert torch.equal(t, ops.linear_weight_t_kmajor(p.detach().to(BF)))
        fp.begin_step()
        fp.grad.normal_()
        opt.step()
    with torch.no_grad():
        ps[0].mul_(2.0)                                   # changed behind the optimiser's back: the stale copy must not be served
    t = Fn._bf_t(ps[0])
    assert torch.equal(t, ops.linear_weight_t_kmajor(ps[0].detach().to(BF)))


@pytest.mark.parametrize("tag", ["dit_small_hd64w", "dit_small_hd72"])
def test_stack_route_equals_block_route(tag):
    """`DitStackFn` (one node for all blocks) against the chain of `DitBlockFn`s fed by `LinearFn` modulations on the same weights: the forward is the same launches in
    another order -- identical bits --, the backward differs in the order of its f32 reductions only."""
    import copy
    from dmvae_amd.models import lightningdit_fast as lf
    g = load_golden(tag)
    m = build(tag, g).to(DEV)
    ref = copy.deepcopy(m)
    x, t, y = g.t("x").to(DEV), g.t("t").to(DEV), torch.from_numpy(np.asarray(g["y"])).to(DEV)
    dy = g.t("dy").to(DEV)
    outs = {}
    for name, mod, flag in (("stack", m, True), ("block", ref, False)):
        lf.STACK_FN = flag
        try:
            xa = x.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=BF):
                out = mod(xa, t, y)
            (out.float() * dy).sum().backward()
        finally:
            lf.STACK_FN = True
        outs[name] = (out.detach(), xa.grad)
    assert torch.equal(outs["stack"][0], outs["block"][0])
    assert _rl2(outs["stack"][1], outs["block"][1]) < 2e-3
    pa, pb = dict(m.named_parameters()), dict(ref.named_parameters())
    for n_, p in pa.items():
        if n_ == "pos_embed":
            continue
        assert p.grad is not None and pb[n_].grad is not None, n_
        e = _rl2(p.grad, pb[n_].grad)
        assert e < 5e-3, (n_, e)
    # rerun: bit-identical gradients
    first = {n_: p.grad.clone() for n_, p in pa.items() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    xb = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=BF):
        out2 = m(xb, t, y)
    (out2.float() * dy).sum().backward()
    assert torch.equal(xb.grad, outs["stack"][1])
    for n_, p in pa.items():
        if n_ in first:
            assert torch.equal(p.grad, first[n_]), n_


@pytest.mark.parametrize("shapes", [[(4096, 1152, 1152), (4096, 3456, 1152), (4096, 1152, 3072)], [(4112, 1024, 1024), (4112, 3072, 1024), (4112, 1024, 4096)],
                                    [(64, 128, 128), (33, 256, 128), (200, 128, 384)]])
def test_grouped_linear_weight_gradients(shapes):
    """ops.linear_wgrad_grouped: every problem's dW = dY^T X and db = column sums against fp64 and against the split-K kernel it replaces; ragged M (not a multiple
    of 32) included; a second call (cached table) and a rerun give the same bits."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(len(shapes) + shapes[0][0])
    probs, refs = [], []
    for i, (m, cout, cin) in enumerate(shapes):
        dy = torch.randn(m, cout, generator=g).to(DEV).to(BF)
        x = torch.randn(m, cin, generator=g).to(DEV).to(BF)
        dw = torch.full((cout, cin), 3.0, device=DEV)
        db = torch.full((cout,), 3.0, device=DEV) if i != 1 else None
        probs.append((dy, x, dw, db))
        refs.append((dy.double().t() @ x.double(), dy.double().sum(0)))
    ops.linear_wgrad_grouped(probs)
    first = [(p[2].clone(), None if p[3] is None else p[3].clone()) for p in probs]
    for (dy, x, dw, db), (rw, rb) in zip(probs, refs):
        assert rel_err(dw, rw) < 2e-6
        if db is not None:
            assert rel_err(db, rb) < 2e-6
        if dy.shape[0] % 32 == 0 and dy.shape[0] >= 4096:
            dw2, db2 = ops.conv2d_nhwc_wgrad(dy.view(1, 1, *dy.shape), x.view(1, 1, *x.shape), 1)
            assert rel_err(dw, dw2.view_as(dw)) < 2e-6 and rel_err(db if db is not None else db2, db2) < 2e-6
    for p in probs:
        p[2].zero_()
    ops.linear_wgrad_grouped(probs)
    for p, f in zip(probs, first):
        assert torch.equal(p[2], f[0]) and (p[3] is None or torch.equal(p[3], f[1]))
    # both placements of the same tiles (every problem spread over all XCDs / every chunk of ~32 tiles on one XCD: dmvae_linear_wgrad_grouped_plan): the same bits
    keep = (ops.WGRAD_GROUPED_XCD, ops.WGRAD_GROUPED_XCD_MIN)
    try:
        for placed in (True, False):
            ops.WGRAD_GROUPED_XCD, ops.WGRAD_GROUPED_XCD_MIN = placed, 1
            ops._PTR_TABLES.clear()
            for p in probs:
                p[2].fill_(5.0)
                if p[3] is not None:
                    p[3].fill_(5.0)
            ops.linear_wgrad_grouped(probs)
            for p, f in zip(probs, first):
                assert torch.equal(p[2], f[0]) and (p[3] is None or torch.equal(p[3], f[1])), placed
    finally:
        ops.WGRAD_GROUPED_XCD, ops.WGRAD_GROUPED_XCD_MIN = keep
        ops._PTR_TABLES.clear()


@pytest.mark.parametrize("tag", ["dit_small_hd64w", "dit_small_hd72"])
def test_inference_forward_with_batched_adaln_is_bit_identical(tag):
    """forward_inference with every block's adaLN Linear in one launch = one launch per block, bit for bit (the same kernel, blockIdx.y = block)."""
    from dmvae_amd.models import lightningdit_fast as lf
    g = load_golden(tag)
    m = build(tag, g).to(DEV).requires_grad_(False)
    x, t, y = g.t("x").to(DEV), g.t("t").to(DEV), torch.from_numpy(np.asarray(g["y"])).to(DEV)
    with torch.no_grad(), torch.autocast("cuda", dtype=BF):
        a = m(x, t, y)
        lf.BATCHED_ADALN = False
        try:
            b_ = m(x, t, y)
        finally:
            lf.BATCHED_ADALN = True
    assert torch.equal(a, b_)


@pytest.mark.parametrize("m,n,k,splits,kmajor", [(4096, 1152, 3072, 3, False), (4096, 1152, 6144, 3, True), (4096, 1152, 3456, 3, True), (300, 72, 768, 2, False),
                                                 (257, 1152, 1536, 4, True)])
def test_splitk_linear(m, n, k, splits, kmajor):
    """ops.linear_splitk = x w^T + bias with the reduction in `splits` parts: against fp64 at the bf16 result's rounding, against the unsplit kernel to a flip of the
    last bf16 bit on a few elements (another f32 summation order), rerun-identical."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).to(DEV).to(BF)
    w = (0.03 * torch.randn(n, k, generator=g)).to(DEV).to(BF)
    b = torch.randn(n, generator=g).to(DEV)
    assert ops.linear_splitk_supported(m, n, k, splits)
    wk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32) if kmajor else w
    y = ops.linear_splitk(x, wk, b, splits)
    ref = x.double() @ w.double().t() + b.double()
    assert _rl2(y, ref) < 3e-3
    y0 = ops.linear_bf16(x, wk, b) if ops.linear_supported(m, n, k) else None
    if y0 is not None:
        assert (y != y0).float().mean().item() < 2e-2 and _rl2(y, y0.double()) < 2e-3
    assert torch.equal(y, ops.linear_splitk(x, wk, b, splits))
    y2 = ops.linear_splitk(x, wk, None, splits)
    assert _rl2(y2, ref - b.double()) < 3e-3
