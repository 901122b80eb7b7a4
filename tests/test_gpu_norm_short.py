"""csrc/norm_short.hip (-m gpu): a ResnetBlock's 1x1 nin_shortcut (models/flux_ae.py:67,77-82) evaluated inside the block's first GroupNorm passes --
forward (norm1's apply pass also writes xs = conv1x1(x)) and backward (norm1's backward apply pass forms the shortcut's input gradient from dy in place) --
against the stored-operand route (the 1x1 conv launches + dmvae_groupnorm_apply / dmvae_groupnorm_bwd[_colsum]: the same arithmetic up to the summation order
inside one bf16 rounding), against f64 on the same operands, ragged chunking, guard zones around every output, reruns bit-identical, and through
`ResnetBlockFn` with the switch on and off."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
PAIRS = [(256, 128), (512, 256)]      # (channels of the block input, of its output): up[0]'s first block (weight in LDS, a wave per run), up[1]'s (weight in registers, a block per run)

SHAPES = [(2, 16, 16), (1, 24, 32), (3, 8, 18), (2, 4, 4), (1, 48, 48)]      # (n, h, w): hw % 16 == 0; 144 / 768 / 2304 pixels are ragged against the 128-pixel rounds
CASES = [(c, cs, n, h, w) for (c, cs) in PAIRS for (n, h, w) in SHAPES]


def _case(n, h, w, seed=0, C=256, CS=128):
    g = torch.Generator().manual_seed(1000 * n + 10 * h + w + seed + C)
    x = (torch.randn(n, h, w, C, generator=g) * 1.5 + 0.3).to(DEV).to(BF)
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(C, generator=g)).to(DEV)
    sw = (torch.randn(CS, C, 1, 1, generator=g) * 0.06).to(DEV)
    sb = (0.1 * torch.randn(CS, generator=g)).to(DEV)
    da = torch.randn(n, h, w, C, generator=g).to(DEV).to(BF)
    dy = torch.randn(n, h, w, CS, generator=g).to(DEV).to(BF)
    return x, gamma, beta, sw, sb, da, dy


def _close_up_to_rounding(got, ref, frac=0.05):
    """the two routes add one element's products in a different order in f32: a few elements land on the other side of a bf16 rounding boundary"""
    d = (got.float() - ref.float()).abs()
    scale = ref.float().abs().max().item()
    assert d.max().item() <= 2.0 ** -6 * scale, (d.max().item(), scale)
    assert (d > 0).float().mean().item() < frac, (d > 0).float().mean().item()


@pytest.mark.parametrize("C,CS,n,h,w", CASES)
def test_forward_equals_apply_and_the_1x1_conv(C, CS, n, h, w):
    from dmvae_amd import ops
    from dmvae_amd.functional import packed
    assert ops.groupnorm_short_supported(n, h * w, C, CS)
    x, gamma, beta, sw, sb, _, _ = _case(n, h, w, C=C, CS=CS)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    a0 = ops.groupnorm_apply(x, st, gamma, beta, True)
    xs0 = ops.conv2d_nhwc(x, packed(sw), sb, ks=1)
    a, xs = ops.groupnorm_apply_short(x, st, gamma, beta, packed(sw), sb)
    assert a.shape == x.shape and xs.shape == (n, h, w, CS) and a.dtype == BF and xs.dtype == BF
    assert torch.equal(a, a0)                       # the elementwise half is the same arithmetic
    _close_up_to_rounding(xs, xs0)
    ref = torch.einsum("nhwc,kc->nhwk", x.double(), sw.view(CS, C).to(BF).double()) + sb.double()
    assert rel_err(xs.double(), ref) < 6e-3 and rel_err(xs.double(), ref) <= 1.1 * rel_err(xs0.double(), ref) + 1e-6
    a_nb, xs_nb = ops.groupnorm_apply_short(x, st, gamma, beta, packed(sw), None)
    _close_up_to_rounding(xs_nb, ops.conv2d_nhwc(x, packed(sw), None, ks=1))
    for _ in range(2):
        a2, xs2 = ops.groupnorm_apply_short(x, st, gamma, beta, packed(sw), sb)
        assert torch.equal(a2, a) and torch.equal(xs2, xs)


@pytest.mark.parametrize("colsum", [False, True])
@pytest.mark.parametrize("C,CS,n,h,w", CASES)
def test_backward_equals_the_stored_gradient_route(C, CS, n, h, w, colsum):
    from dmvae_amd import ops
    from dmvae_amd.functional import packed
    x, gamma, beta, sw, sb, da, dy = _case(n, h, w, seed=3, C=C, CS=CS)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    dxs = ops.conv2d_nhwc(dy, packed(sw, True), ks=1)
    dx0, dg0, db0 = ops.groupnorm_bwd(da, x, st, gamma, beta, True, dres=dxs, want_colsum=colsum)
    dx, dg, db = ops.groupnorm_bwd_short(da, x, dy, packed(sw, True), st, gamma, beta, True, want_colsum=colsum)
    assert dx.shape == x.shape and dx.dtype == BF
    _close_up_to_rounding(dx, dx0)
    assert torch.equal(dg, dg0) and torch.equal(db, db0)          # the reduction half is the same launches
    if colsum:
        cs, cs0 = dx._dmvae_colsum[0], dx0._dmvae_colsum[0]
        assert rel_err(cs, cs0) < 2e-3
        assert rel_err(cs.double(), dx.double().sum((0, 1, 2))) < 1e-5      # the sums of what was stored
    else:
        assert not hasattr(dx, "_dmvae_colsum")
    for _ in range(2):
        dx2, dg2, db2 = ops.groupnorm_bwd_short(da, x, dy, packed(sw, True), st, gamma, beta, True, want_colsum=colsum)
        assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)
        if colsum:
            assert torch.equal(dx2._dmvae_colsum[0], dx._dmvae_colsum[0])


@pytest.mark.parametrize("C,CS", PAIRS)
@pytest.mark.parametrize("n,h,w", SHAPES[:3])
def test_backward_vs_f64_autograd_of_the_oracle(n, h, w, C, CS):
    """f64 autograd through oracle.ref_cpu's group_norm -> swish on the HIP path's operands plus the shortcut's dy W in f64: what is left is the bf16 rounding of
    the shortcut gradient and of dx itself."""
    from dmvae_amd import ops
    from dmvae_amd.functional import packed
    from oracle import ref_cpu
    x, gamma, beta, sw, sb, da, dy = _case(n, h, w, seed=7, C=C, CS=CS)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    dx, dg, db = ops.groupnorm_bwd_short(da, x, dy, packed(sw, True), st, gamma, beta, True)
    xd = x.double().permute(0, 3, 1, 2).cpu().requires_grad_(True)
    gd, bd = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    a = ref_cpu.swish(ref_cpu.group_norm(xd, gd, bd, 32, 1e-6))
    (a * da.double().permute(0, 3, 1, 2).cpu()).sum().backward()
    ref = xd.grad.permute(0, 2, 3, 1) + torch.einsum("nhwk,kc->nhwc", dy.double().cpu(), sw.view(CS, C).to(BF).double().cpu())
    dxs = ops.conv2d_nhwc(dy, packed(sw, True), ks=1)
    dx0, _, _ = ops.groupnorm_bwd(da, x, st, gamma, beta, True, dres=dxs)
    assert rel_err(dx.double().cpu(), ref) < 8e-3
    assert rel_err(dx.double().cpu(), ref) <= 1.1 * rel_err(dx0.double().cpu(), ref) + 1e-6
    assert rel_err(dg.double().cpu(), gd.grad) < 5e-3 and rel_err(db.double().cpu(), bd.grad) < 5e-3


@pytest.mark.parametrize("C,CS", PAIRS)
def test_outputs_stay_inside_their_buffers(C, CS):
    """raw C-ABI calls on outputs carved out of sentinel-filled buffers: nothing before or behind them is written"""
    from dmvae_amd import _lib, ops
    from dmvae_amd.functional import packed
    n, h, w = 2, 8, 18
    x, gamma, beta, sw, sb, da, dy = _case(n, h, w, seed=11, C=C, CS=CS)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    L = _lib.lib()
    guard = 4096
    def carve(numel):
        buf = torch.full((numel + 2 * guard,), -7.0, dtype=BF, device=DEV)
        return buf, buf[guard:guard + numel]
    abuf, a = carve(x.numel())
    sbuf, xs = carve(n * h * w * CS)
    ops.check(L.dmvae_groupnorm_apply_short(x.data_ptr(), st.data_ptr(), gamma.data_ptr(), beta.data_ptr(), packed(sw).data_ptr(), sb.data_ptr(), a.data_ptr(),
                                            xs.data_ptr(), n, h * w, C, CS, 32, 1, ops._stream()), "apply_short")
    dbuf, dx = carve(x.numel())
    ws = torch.empty(L.dmvae_groupnorm_bwd_short_workspace(n, h * w, C, CS, 32), dtype=torch.uint8, device=DEV)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    ops.check(L.dmvae_groupnorm_bwd_short(da.data_ptr(), x.data_ptr(), dy.data_ptr(), packed(sw, True).data_ptr(), st.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                          dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, ws.data_ptr(), ws.numel(), n, h * w, C, CS, 32, 1, 0, 0, ops._stream()),
              "bwd_short")
    torch.cuda.synchronize()
    for buf in (abuf, sbuf, dbuf):
        assert (buf[:guard] == -7.0).all() and (buf[-guard:] == -7.0).all()
    a1, xs1 = ops.groupnorm_apply_short(x, st, gamma, beta, packed(sw), sb)
    assert torch.equal(a.view_as(a1), a1) and torch.equal(xs.view_as(xs1), xs1)
    dx1, _, _ = ops.groupnorm_bwd_short(da, x, dy, packed(sw, True), st, gamma, beta, True)
    assert torch.equal(dx.view_as(dx1), dx1)


def test_unsupported_shapes_are_refused():
    from dmvae_amd import ops
    assert not ops.groupnorm_short_supported(2, 64, 512, 128)
    assert not ops.groupnorm_short_supported(2, 72, 256, 128)        # hw % 16
    assert not ops.groupnorm_short_supported(2, 64, 256, 64)
    assert not ops.groupnorm_short_supported(2, 64, 128, 64)
    x = torch.zeros(1, 4, 4, 128, dtype=BF, device=DEV)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    with pytest.raises(ValueError):
        ops.groupnorm_apply_short(x, st, torch.ones(128, device=DEV), torch.zeros(128, device=DEV), torch.zeros(64, 1, 128, dtype=BF, device=DEV), None)


@pytest.mark.parametrize("C,CS", PAIRS)
def test_resnet_block_with_the_switch_on_and_off(monkeypatch, C, CS):
    """ResnetBlock(256 -> 128 / 512 -> 256) forward + backward through `ResnetBlockFn`: the fused route is the one taken by default, and it agrees with the stored-operand route."""
    from dmvae_amd import functional as Fn, ops
    from dmvae_amd.models.flux_ae import ResnetBlock
    torch.manual_seed(0)
    blk = ResnetBlock(C, CS).to(DEV)
    x0 = (torch.randn(2, 16, 16, C, device=DEV) * 1.2).to(BF)
    gout = torch.randn(2, 16, 16, CS, device=DEV).to(BF)
    calls = {"fwd": 0, "bwd": 0}
    f0, b0 = ops.groupnorm_apply_short, ops.groupnorm_bwd_short
    monkeypatch.setattr(ops, "groupnorm_apply_short", lambda *a, **k: (calls.__setitem__("fwd", calls["fwd"] + 1), f0(*a, **k))[1])
    monkeypatch.setattr(ops, "groupnorm_bwd_short", lambda *a, **k: (calls.__setitem__("bwd", calls["bwd"] + 1), b0(*a, **k))[1])

    def run(on):
        monkeypatch.setattr(Fn, "SHORTCUT_IN_NORM", on)
        blk.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        y = blk.forward_nhwc(x)
        y.backward(gout)
        return y.detach(), x.grad, {k: p.grad.clone() for k, p in blk.named_parameters()}

    y1, dx1, g1 = run(True)
    assert calls == {"fwd": 1, "bwd": 1}
    y0, dx0, g0 = run(False)
    assert calls == {"fwd": 1, "bwd": 1}
    _close_up_to_rounding(y1, y0)
    _close_up_to_rounding(dx1, dx0)
    for k in g0:
        assert rel_err(g1[k], g0[k]) < 2e-3, k


@pytest.mark.parametrize("C,CS,n,h,w", [(256, 128, 32, 256, 256), (512, 256, 32, 128, 128)])
def test_training_shapes_equal_the_stored_routes(C, CS, n, h, w):
    """the two shapes of configuration C2 at B = 32 (up[0].block[0], up[1].block[0]): 24 chunks per image, byte offsets past 2^31"""
    from dmvae_amd import ops
    from dmvae_amd.functional import packed
    g = torch.Generator(device=DEV).manual_seed(C)
    x = (torch.randn(n, h, w, C, generator=g, device=DEV) * 1.5 + 0.3).to(BF)
    da = torch.randn(n, h, w, C, generator=g, device=DEV).to(BF)
    dy = torch.randn(n, h, w, CS, generator=g, device=DEV).to(BF)
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g, device=DEV)
    beta = 0.1 * torch.randn(C, generator=g, device=DEV)
    sw = torch.randn(CS, C, 1, 1, generator=g, device=DEV) * 0.06
    sb = 0.1 * torch.randn(CS, generator=g, device=DEV)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    a, xs = ops.groupnorm_apply_short(x, st, gamma, beta, packed(sw), sb)
    assert torch.equal(a, ops.groupnorm_apply(x, st, gamma, beta, True))
    xs0 = ops.conv2d_nhwc(x, packed(sw), sb, ks=1)
    _close_up_to_rounding(xs, xs0)
    del a, xs0
    dx, dg, db = ops.groupnorm_bwd_short(da, x, dy, packed(sw, True), st, gamma, beta, True, want_colsum=True)
    dxs = ops.conv2d_nhwc(dy, packed(sw, True), ks=1)
    dx0, dg0, db0 = ops.groupnorm_bwd(da, x, st, gamma, beta, True, dres=dxs, want_colsum=True)
    _close_up_to_rounding(dx, dx0)
    assert torch.equal(dg, dg0) and torch.equal(db, db0)
    assert rel_err(dx._dmvae_colsum[0], dx0._dmvae_colsum[0]) < 2e-3
