"""LightningDiT inference path on the HIP kernels (-m gpu): csrc/dit.hip kernels against their definitions, the fast forward against the
fixtures captured from the reference, the CPU oracle with bf16 rounding at the autocast sites, and the stock modules under autocast(bf16)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from test_oracle_dit import CFGS, build

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
Q = R.bf16_round


def _rl2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("c,n", [(1152, 256), (128, 64), (144, 64), (2048, 32)])
def test_dit_elementwise_kernels(c, n):
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(c)
    b = 3
    x = (torch.randn(b, n, c, generator=g) * 2).to(DEV)
    w = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV)
    mod = (0.5 * torch.randn(b, 6 * c, generator=g)).to(DEV).to(BF)
    y = ops.rmsnorm_modulate(x, w, mod, 0, c)
    xd = x.double()
    nrm = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-6) * w.double()
    ref = nrm * (1 + mod[:, c:2 * c].float()).to(BF).double().unsqueeze(1) + mod[:, :c].double().unsqueeze(1)
    assert (y.double() - ref).abs().max() <= 2 ** -7 * ref.abs().max()          # one bf16 rounding of the f32 result
    assert _rl2(y, ref) < 3e-3
    y2 = ops.rmsnorm_modulate(x, w, mod, -1, 4 * c)                              # no shift
    ref2 = nrm * (1 + mod[:, 4 * c:5 * c].float()).to(BF).double().unsqueeze(1)
    assert _rl2(y2, ref2) < 3e-3
    # gated residual
    yb = torch.randn(b, n, c, generator=g).to(DEV).to(BF)
    want = x.double() + (mod[:, 2 * c:3 * c].unsqueeze(1) * yb).double()        # bf16 * bf16 -> bf16 product, like the reference's autocast graph
    ops.gated_residual_(x, yb, mod, 2 * c)
    assert (x.double() - want).abs().max() < 1e-5
    # swiglu
    hid = 3072 if c == 1152 else 2 * c
    x12 = torch.randn(b * n, 2 * hid, generator=g).to(DEV).to(BF)
    want = F.silu(x12[:, :hid]) * x12[:, hid:]                                   # bf16 silu, then bf16 product
    assert torch.equal(ops.swiglu(x12), want) or (ops.swiglu(x12).float() - want.float()).abs().max() <= 2 ** -7 * want.float().abs().max()


@pytest.mark.parametrize("c,n", [(1152, 256), (144, 64), (2048, 32)])
def test_gated_residual_rmsnorm_modulate_is_the_two_kernels_back_to_back(c, n):
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(c + n)
    b = 3
    x = (torch.randn(b, n, c, generator=g) * 2).to(DEV)
    r = torch.randn(b, n, c, generator=g).to(DEV).to(BF)
    w = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV)
    gmod = (0.5 * torch.randn(b, 6 * c, generator=g)).to(DEV).to(BF)
    mod = (0.5 * torch.randn(b, 6 * c, generator=g)).to(DEV).to(BF)
    x1 = x.clone()
    ops.gated_residual_(x1, r, gmod, 5 * c)
    y1 = ops.rmsnorm_modulate(x1, w, mod, 3 * c, 4 * c)
    x2 = x.clone()
    y2 = ops.gated_residual_rmsnorm_modulate_(x2, r, gmod, 5 * c, w, mod, 3 * c, 4 * c)
    assert torch.equal(x1, x2) and torch.equal(y1, y2)
    xin = x.clone()
    x5, y5 = ops.gated_residual_out(xin, r, gmod, 5 * c, w, mod, 3 * c, 4 * c)                 # out of place: the input survives
    assert torch.equal(xin, x) and torch.equal(x5, x1) and torch.equal(y5, y1)
    x6, none = ops.gated_residual_out(xin, r, gmod, 5 * c)
    assert none is None and torch.equal(x6, x1)
    y3 = ops.gated_residual_rmsnorm_modulate_(x.clone(), r, gmod, 2 * c, w, mod, -1, c)       # no shift
    x4 = x.clone()
    ops.gated_residual_(x4, r, gmod, 2 * c)
    assert torch.equal(y3, ops.rmsnorm_modulate(x4, w, mod, -1, c))


@pytest.mark.parametrize("heads,d", [(16, 72), (2, 64), (3, 32)])
def test_qknorm_rope_kernel(heads, d):
    from dmvae_amd import ops
    from dmvae_amd.models.lightningdit import RMSNorm, VisionRotaryEmbeddingFast
    g = torch.Generator().manual_seed(d)
    b, side = 2, 8
    n = side * side
    qkv = torch.randn(b, n, 3 * heads * d, generator=g).to(DEV).to(BF)
    rope = VisionRotaryEmbeddingFast(dim=d // 2, pt_seq_len=side).to(DEV)
    qn, kn = RMSNorm(d).to(DEV), RMSNorm(d).to(DEV)
    with torch.no_grad():
        qn.weight.uniform_(0.5, 1.5); kn.weight.uniform_(0.5, 1.5)
    q, k, v = ops.qknorm_rope(qkv, qn.weight.detach(), kn.weight.detach(), rope.freqs_cos, rope.freqs_sin, heads)
    dp = (d + 31) // 32 * 32
    assert q.shape == (b * heads, n, dp) and v.shape == (b * heads, n, d)
    q5 = qkv.view(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)
    with torch.no_grad():
        qr, kr = rope(qn(q5[0])).to(BF), rope(kn(q5[1])).to(BF)                  # bf16 in -> RMSNorm rounds to bf16, weight / RoPE in f32, bf16 at SDPA
    assert (q[..., :d].float() - qr.reshape(b * heads, n, d).float()).abs().max() <= 2 ** -7 * qr.float().abs().max()
    assert _rl2(q[..., :d], qr.reshape(b * heads, n, d)) < 2e-3 and _rl2(k[..., :d], kr.reshape(b * heads, n, d)) < 2e-3
    assert torch.equal(v, q5[2].reshape(b * heads, n, d))
    if dp > d:
        assert float(q[..., d:].abs().max()) == 0.0 and float(k[..., d:].abs().max()) == 0.0


@pytest.mark.parametrize("b,heads,d,n", [(2, 16, 72, 256), (3, 2, 64, 64), (1, 4, 72, 288), (2, 3, 64, 37), (1, 2, 40, 96)])
def test_fused_attention_heads_kernel(b, heads, d, n):
    """dmvae_attention_heads_bf16 against softmax(scale q k^T) v in f64 on the same bf16 operands (P rounded to bf16 before the second product, as
    the kernel and the autocast graph's flash kernel do), including ragged key counts and the zero-padded head dim."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(b * 1000 + d + n)
    dp = (d + 31) // 32 * 32
    q = torch.zeros(b * heads, n, dp)
    k = torch.zeros(b * heads, n, dp)
    q[..., :d] = torch.randn(b * heads, n, d, generator=g)
    k[..., :d] = torch.randn(b * heads, n, d, generator=g)
    v = torch.randn(b * heads, n, d, generator=g)
    q, k, v = q.to(DEV).to(BF), k.to(DEV).to(BF), v.to(DEV).to(BF)
    assert ops.attention_heads_supported(n, d)
    out = ops.attention_heads(q, k, v, b, d ** -0.5)
    assert out.shape == (b, n, heads * d) and out.dtype == BF
    p = torch.softmax(q.double() @ k.double().transpose(1, 2) * d ** -0.5, dim=-1)
    ref = (p @ v.double()).view(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, heads * d)
    assert _rl2(out, ref) < 6e-3
    assert (out.double() - ref).abs().max() < 2 ** -6 * ref.abs().max() + 1e-3
    assert torch.equal(out, ops.attention_heads(q, k, v, b, d ** -0.5))
    if n % 32 or d % 8 or d < 64:
        return
    # and the composed route it replaces
    pc = ops.softmax_rows(ops.gemm_nt(q, k, out_f32=True), d ** -0.5)
    oc = ops.gemm_nt(pc, ops.transpose_last2(v)).view(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, heads * d)
    assert _rl2(out, oc) < 6e-3


@pytest.mark.parametrize("b,heads,d,n", [(2, 16, 72, 256), (3, 2, 64, 64), (1, 4, 72, 288), (2, 3, 64, 37), (1, 2, 40, 96)])
def test_attention_with_qknorm_rope_inside_equals_the_two_kernels(b, heads, d, n):
    """dmvae_attention_qknorm_rope_bf16 == dmvae_qknorm_rope_bf16 followed by dmvae_attention_heads_bf16 (same arithmetic; only the order of the
    f32 additions in the row's sum of squares differs)."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(b * 100 + d + n)
    qkv = (torch.randn(b, n, 3 * heads * d, generator=g) * 1.5).to(DEV).to(BF)
    qw = (1 + 0.3 * torch.randn(d, generator=g)).to(DEV)
    kw = (1 + 0.3 * torch.randn(d, generator=g)).to(DEV)
    ang = torch.rand(n, d, generator=g) * 6.28
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    q, k, v = ops.qknorm_rope(qkv, qw, kw, cos, sin, heads, 1e-6)
    want = ops.attention_heads(q, k, v, b, d ** -0.5)
    got = ops.attention_qknorm_rope(qkv, qw, kw, cos, sin, heads, 1e-6, d ** -0.5)
    assert got.shape == want.shape == (b, n, heads * d)
    assert _rl2(got, want) < 2e-3
    assert (got.float() - want.float()).abs().max() <= 2 ** -6 * want.float().abs().max() + 1e-3
    assert torch.equal(got, ops.attention_qknorm_rope(qkv, qw, kw, cos, sin, heads, 1e-6, d ** -0.5))


@pytest.mark.parametrize("tag", ["dit_small_hd64w", "dit_small_hd72"])
def test_lightningdit_fast_forward_vs_fixture_oracle_and_stock(tag):
    g = load_golden(tag)
    m = build(tag, g).to(DEV)
    x, t, y = g.t("x").to(DEV), g.t("t").to(DEV), torch.from_numpy(np.asarray(g["y"])).to(DEV)
    for p in m.parameters():
        p.requires_grad_(False)                                                  # the teacher's state (train_dmd.py:372-373)
    with torch.autocast("cuda", dtype=BF):
        out = m(x, t, y)                                                          # dispatches to lightningdit_fast.forward_inference
        ref_stock = m.forward_stock(x, t, y)
    assert out.dtype == BF and out.shape == x.shape
    with torch.no_grad():
        yo = R.lightningdit_forward(g.t("x"), g.t("t"), torch.from_numpy(np.asarray(g["y"])), {k: v.cpu() for k, v in m.state_dict().items()},
                                    CFGS[tag]["num_heads"], CFGS[tag]["patch_size"], q=Q)
    e_fast, e_stock, e_orc = _rl2(out.float().cpu(), g.t("out")), _rl2(ref_stock.float().cpu(), g.t("out")), _rl2(yo, g.t("out"))
    print(f"{tag}: rel-L2 to the f32 reference -- HIP path {e_fast:.2e}, stock autocast {e_stock:.2e}, bf16-site oracle {e_orc:.2e}")
    assert _rl2(out.float().cpu(), yo) < 2e-2                                    # same rounding sites, different accumulation order
    assert e_fast < 1.5 * max(e_stock, e_orc) + 2e-3                              # as close to the f32 reference as the reference's own autocast run
    # with gradients enabled and trainable parameters the module takes the stock route (the student's training turn)
    for p in m.parameters():
        p.requires_grad_(True)
    with torch.autocast("cuda", dtype=BF):
        o2 = m(x, t, y)
    assert o2.requires_grad
    with torch.no_grad(), torch.autocast("cuda", dtype=BF):
        assert torch.equal(m(x, t, y), out)                                       # no_grad: fast path again, deterministic


def test_lightningdit_xl1_fast_forward_shapes_and_determinism():
    """DiT-XL/1 at the DMD stage's shape (B = 16, 32 x 16 x 16 latents, head dim 72): finite, deterministic, close to the stock modules."""
    from dmvae_amd.models.lightningdit import LightningDiT_models
    from oracle.detweights import det_fill_
    torch.manual_seed(0)
    m = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).to(DEV).eval().requires_grad_(False)
    with torch.no_grad():
        for blk in m.blocks:                                                      # the reference zero-initialises these: give the blocks something to do
            blk.adaLN_modulation[1].weight.normal_(0, 0.02); blk.adaLN_modulation[1].bias.normal_(0, 0.3)
        m.final_layer.linear.weight.normal_(0, 0.02); m.final_layer.adaLN_modulation[1].bias.normal_(0, 0.3)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(16, 32, 16, 16, generator=g).to(DEV)
    t, y = torch.rand(16, generator=g).to(DEV), torch.randint(0, 1001, (16,), generator=g).to(DEV)
    with torch.autocast("cuda", dtype=BF):
        a, b = m(x, t, y), m(x, t, y)
        s = m.forward_stock(x, t, y)
    assert a.shape == (16, 32, 16, 16) and torch.isfinite(a).all() and torch.equal(a, b)
    assert _rl2(a.float(), s.float()) < 3e-2


@pytest.mark.parametrize("c,n", [(1152, 256), (128, 64), (144, 40)])
def test_dit_backward_kernels(c, n):
    """gated_residual_bwd, swiglu_bwd and rmsnorm_modulate_bwd against fp64 autograd of their forward definitions (straight-through at the
    forward's bf16 rounding sites)."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(c + n)
    b = 3
    x = (torch.randn(b, n, c, generator=g) * 2).to(DEV)
    w = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV)
    mod = (0.5 * torch.randn(b, 6 * c, generator=g)).to(DEV).to(BF)
    da = torch.randn(b, n, c, generator=g).to(DEV).to(BF)
    dres = torch.randn(b, n, c, generator=g).to(DEV)
    # rmsnorm_modulate
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    shr, scr = mod[:, :c].double().requires_grad_(True), mod[:, c:2 * c].double().requires_grad_(True)
    nrm = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6) * wr
    m_ = (1 + scr)
    m_ = m_ + ((1 + mod[:, c:2 * c].float()).to(BF).double() - m_).detach()                 # forward value = bf16(1 + scale), gradient straight through
    (nrm * m_.unsqueeze(1) + shr.unsqueeze(1)).backward(da.double())
    dmod = torch.zeros(b, 6 * c, device=DEV)
    dx = dres.clone()
    dw = ops.rmsnorm_modulate_bwd_(dx, da, x, w, mod, dmod, 0, c)
    assert rel_err(dx, dres.double() + xr.grad) < 2e-6
    assert rel_err(dw, wr.grad) < 1e-5
    assert rel_err(dmod[:, :c], shr.grad) < 1e-5 and rel_err(dmod[:, c:2 * c], scr.grad) < 1e-5
    assert float(dmod[:, 2 * c:].abs().max()) == 0.0
    # gated residual
    y = torch.randn(b, n, c, generator=g).to(DEV).to(BF)
    dmod2 = torch.zeros(b, 6 * c, device=DEV)
    dy = ops.gated_residual_bwd(dres, y, mod, dmod2, 2 * c)
    assert torch.equal(dy, (mod[:, 2 * c:3 * c].float().unsqueeze(1) * dres).to(BF))
    assert rel_err(dmod2[:, 2 * c:3 * c], (dres.double() * y.double()).sum(1)) < 1e-5
    # swiglu
    hid = 2 * c
    x12 = torch.randn(b * n, 2 * hid, generator=g).to(DEV).to(BF)
    dh = torch.randn(b * n, hid, generator=g).to(DEV).to(BF)
    xr = x12.double().requires_grad_(True)
    (F.silu(xr[:, :hid]) * xr[:, hid:]).backward(dh.double())
    assert _rl2(ops.swiglu_bwd(dh, x12), xr.grad) < 4e-3


@pytest.mark.parametrize("heads,d", [(16, 72), (2, 64)])
def test_qknorm_rope_bwd_kernel(heads, d):
    from dmvae_amd import ops
    from dmvae_amd.models.lightningdit import RMSNorm, VisionRotaryEmbeddingFast
    g = torch.Generator().manual_seed(d + 1)
    b, side = 2, 8
    n = side * side
    qkv = torch.randn(b, n, 3 * heads * d, generator=g).to(DEV).to(BF)
    rope = VisionRotaryEmbeddingFast(dim=d // 2, pt_seq_len=side).to(DEV).double()
    qn, kn = RMSNorm(d).to(DEV).double(), RMSNorm(d).to(DEV).double()
    with torch.no_grad():
        qn.weight.uniform_(0.5, 1.5); kn.weight.uniform_(0.5, 1.5)
    dp = (d + 31) // 32 * 32
    dq = torch.zeros(b * heads, n, dp, device=DEV, dtype=BF); dk = torch.zeros_like(dq)
    dq[..., :d] = torch.randn(b * heads, n, d, generator=g).to(DEV).to(BF); dk[..., :d] = torch.randn(b * heads, n, d, generator=g).to(DEV).to(BF)
    dv = torch.randn(b * heads, n, d, generator=g).to(DEV).to(BF)
    x = qkv.double().requires_grad_(True)
    q5 = x.view(b, n, 3, heads, d).permute(2, 0, 3, 1, 4)

    def norm_ste(t, mod_):      # RMSNorm with the forward's bf16 rounding of the normalised value (rms_norm.py:75), gradient straight through
        nh = t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-6)
        return (nh + (nh.float().to(BF).double() - nh).detach()) * mod_.weight
    qq, kk = rope(norm_ste(q5[0], qn)), rope(norm_ste(q5[1], kn))
    loss = (qq.reshape(b * heads, n, d) * dq[..., :d].double()).sum() + (kk.reshape(b * heads, n, d) * dk[..., :d].double()).sum() + \
        (q5[2].reshape(b * heads, n, d) * dv.double()).sum()
    loss.backward()
    dqkv, dqw, dkw = ops.qknorm_rope_bwd(dq, dk, dv, qkv, qn.weight.detach().float(), kn.weight.detach().float(), rope.freqs_cos.float(), rope.freqs_sin.float(), heads)
    assert _rl2(dqkv, x.grad) < 4e-3                                             # bf16 result
    assert _rl2(dqw, qn.weight.grad) < 2e-3 and _rl2(dkw, kn.weight.grad) < 2e-3   # the forward's bf16-rounded normalised value enters the weight gradient


@pytest.mark.parametrize("tag", ["dit_small_hd64w", "dit_small_hd72"])
def test_lightningdit_train_route_matches_stock_autocast(tag):
    """forward with gradients on the HIP kernels (functional.DitBlockFn) vs the stock modules under autocast(bf16) on the same weights: output, input
    gradient and every parameter gradient; and both as close to the reference's f32 gradients (fixture) as each other."""
    import copy
    g = load_golden(tag)
    m = build(tag, g).to(DEV)
    ref = copy.deepcopy(m)
    x, t, y = g.t("x").to(DEV), g.t("t").to(DEV), torch.from_numpy(np.asarray(g["y"])).to(DEV)
    dy = g.t("dy").to(DEV)
    xa = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=BF):
        out = m(xa, t, y)                                                        # grad enabled + trainable parameters: lightningdit_fast.forward_train
    (out.float() * dy).sum().backward()
    xb = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=BF):
        outs = ref.forward_stock(xb, t, y)
    (outs.float() * dy).sum().backward()
    assert _rl2(out.float(), outs.float()) < 1e-2
    assert _rl2(xa.grad, xb.grad) < 3e-2
    pa, pb = dict(m.named_parameters()), dict(ref.named_parameters())
    worst_hip = worst_stock = 0.0
    for n_, p in pa.items():
        if n_ == "pos_embed":
            continue
        assert p.grad is not None, n_
        assert _rl2(p.grad, pb[n_].grad) < 4e-2, (n_, _rl2(p.grad, pb[n_].grad))
        if "g." + n_ in g:
            worst_hip = max(worst_hip, _rl2(p.grad.cpu(), g.t("g." + n_)))
            worst_stock = max(worst_stock, _rl2(pb[n_].grad.cpu(), g.t("g." + n_)))
    print(f"{tag}: worst rel-L2 of the captured parameter gradients to the f32 reference -- HIP {worst_hip:.2e}, stock autocast {worst_stock:.2e}")
    assert worst_hip < 1.5 * worst_stock + 5e-3
    assert _rl2(xa.grad.cpu(), g.t("dx")) < 1.5 * _rl2(xb.grad.cpu(), g.t("dx")) + 5e-3
    # deterministic
    m.zero_grad(set_to_none=True)
    xc = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=BF):
        out2 = m(xc, t, y)
    (out2.float() * dy).sum().backward()
    assert torch.equal(out, out2) and torch.equal(xa.grad, xc.grad)


@pytest.mark.parametrize("tag", ["dit_small_hd64w", "dit_small_hd72"])
def test_lightningdit_train_route_vs_reference_capture_by_the_bf16_site_oracle(tag):
    """The production bf16 training route (lightningdit_fast.forward_train: DitStackFn, grouped weight gradients, fused boundary passes) against the REFERENCE's
    own f32 capture (oracle/capture_golden_dit.py: out, dx, six full parameter gradients, norm + sum of every parameter gradient), by the criterion of
    test_gpu_modules.py: as close to the f32 reference as the CPU oracle with bf16 rounding at exactly the autocast sites is (x 1.15 + 1e-3).  The oracle's
    backward is torch autograd over oracle.ref_cpu.lightningdit_forward (differentiable: straight-through rounding hooks) -- nothing of this comparison involves
    stock PyTorch on the GPU, so an error the HIP route shared with stock autocast would show here."""
    g = load_golden(tag)
    m = build(tag, g).to(DEV)
    x, t, y = g.t("x"), g.t("t"), torch.from_numpy(np.asarray(g["y"]))
    dy = g.t("dy")
    xa = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=BF):
        out = m(xa, t.to(DEV), y.to(DEV))
    (out.float() * dy.to(DEV)).sum().backward()
    names = [n for n, _ in m.named_parameters()]
    po = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for n in names:
        po[n].requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    yo = R.lightningdit_forward(xo, t, y, po, CFGS[tag]["num_heads"], CFGS[tag]["patch_size"], q=Q)
    yo.backward(dy)

    def floor(hip, orc, ref, what, slack=1.15, abs_floor=1e-3):
        e_hip, e_orc = _rl2(hip, ref), _rl2(orc, ref)
        print(f"{tag} {what}: rel-L2 to the f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
        assert e_hip < slack * e_orc + abs_floor, (what, e_hip, e_orc)
    floor(out.float().cpu(), yo.detach(), g.t("out"), "out")
    floor(xa.grad.cpu(), xo.grad, g.t("dx"), "dx")
    pa = dict(m.named_parameters())
    full = [k[2:] for k in g.keys() if k.startswith("g.")]
    assert len(full) >= 6
    for n in full:
        floor(pa[n].grad.cpu(), po[n].grad, g.t("g." + n), "grad " + n)
    for n in names:                      # EVERY parameter: gradient norm and sum against the reference's, to the oracle's own distance
        if n == "pos_embed" or "gn." + n not in g:
            continue
        want_norm = float(g["gn." + n][0])
        if want_norm < 1e-3:
            continue
        e_hip = abs(pa[n].grad.double().norm().item() - want_norm) / want_norm
        e_orc = abs(po[n].grad.double().norm().item() - want_norm) / want_norm
        assert e_hip < 1.15 * e_orc + 1e-2, (n, e_hip, e_orc)


@pytest.mark.parametrize("tag", ["dit_small_hd64w", "dit_small_hd72"])      # the two fixtures inside the HIP kernels' range (dit_small_hd64's SwiGLU width 341 is refused loudly)
def test_cond_and_uncond_as_one_2b_call_equals_two_b_calls(tag):
    """train_dmd.py:211-217 evaluates a velocity model twice per DMD loss (labels, then the null class); DMDTrainer batches the two evaluations into ONE call on
    2B samples.  Every op of the HIP LightningDiT forward is per sample / per token row and the GEMM's result does not depend on the tile the
    planner picks for the larger M, so the halves of the 2B call must equal the B-sized calls BIT FOR BIT -- and the DMD loss and gradient built from them with it."""
    from dmvae_amd import losses
    g = load_golden(tag)
    m = build(tag, g).to(DEV).requires_grad_(False)
    x, t = g.t("x").to(DEV), g.t("t").to(DEV)
    y = torch.from_numpy(np.asarray(g["y"])).to(DEV)
    null = torch.full_like(y, CFGS[tag]["num_classes"])
    with torch.no_grad(), torch.autocast("cuda", dtype=BF):
        vc, vu = m(x, t, y), m(x, t, null)
        v2 = m(torch.cat([x, x]), torch.cat([t, t]), torch.cat([y, null]))
    b = x.shape[0]
    assert torch.equal(v2[:b], vc) and torch.equal(v2[b:], vu)
    # what DMDTrainer's automatic switch keys on: this build's LightningDiT on the per-sample HIP route, not drawing label drop-outs
    from dmvae_amd.train import _batchable
    assert _batchable(m.eval()) and not _batchable(m.train()) and not _batchable(torch.nn.Linear(2, 2))
    m.eval()
    assert not torch.equal(vc, vu)                                   # the label embedding does change the output: the comparison above is not vacuous
    # the loss assembled either way (the "student" here is the same model scaled: only the plumbing is under test)
    lat = torch.randn_like(x.float())
    xt = losses.dmd_make_xt(lat, torch.randn_like(lat), t.float())
    a = losses.dmd_loss(lat, xt, t.float(), vc, (vc * 0.5).to(vc.dtype), vu, (vu * 0.5).to(vu.dtype), cfg=2.0)
    c = losses.dmd_loss(lat, xt, t.float(), v2[:b], (v2[:b] * 0.5).to(vc.dtype), v2[b:], (v2[b:] * 0.5).to(vc.dtype), cfg=2.0)
    assert torch.equal(a[0], c[0])
