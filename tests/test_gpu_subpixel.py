"""Upsample's conv (models/flux_ae.py:103-107: conv3x3 over the nearest-x2 image) in its sub-pixel form -- the route the decoder takes on the HIP path
(dmvae_amd/functional.py::ConvFn, include/dmvae_hip.h: dmvae_subpixel_weight).  The reference computes the layer in two steps (interpolate, conv2d); here
the taps that read one source pixel are pre-added into a 4x4 weight WD and the layer runs as a transposed 4x4 stride-2 conv (forward), a 4x4 stride-2 conv
(input gradient) and that conv's weight gradient folded back to 3x3.  Held to:
  * the reference's two-step form in fp64 on the same operands (the only difference is the bf16 rounding of WD, so WD is built from weights chosen such
    that the pre-added taps are exactly representable -- then the bar is the kernels' 1e-5);
  * the CPU identity itself (tests/test_oracle_golden.py::test_subpixel_identity, no GPU);
  * the first implementation (nine taps gathered from the half-resolution image) at the bf16 floor, with arbitrary weights."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _coarse(shape, g, scale):
    """Values on a coarse binary grid: sums of up to four of them are exact in bf16 (|v| <= 8 * 2^-4 steps: 5 significant bits + 2 for the sum)."""
    return torch.randint(-8, 9, shape, generator=g).float() * scale


def _ref_two_step(x, w, b, dy):
    """fp64: y = conv2d(nearest_x2(x), W, b, padding 1) and its gradients; x NHWC, dy NHWC."""
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.double().clone().requires_grad_(True)
    br = b.double().clone().requires_grad_(True)
    y = F.conv2d(xr.repeat_interleave(2, 2).repeat_interleave(2, 3), wr, br, padding=1)
    y.backward(dy.double().permute(0, 3, 1, 2))
    return y.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad, br.grad


def test_subpixel_weight_and_fold_vs_einsum():
    from dmvae_amd import ops
    from oracle import ref_cpu as R
    g = torch.Generator().manual_seed(1)
    w = torch.randn(40, 24, 3, 3, generator=g)
    wd = ops.subpixel_weight(w.to(DEV))
    assert wd.shape == (24, 40, 4, 4)
    assert rel_err(wd.cpu(), R.subpixel_weight(w.double())) < 1e-6
    dwd = torch.randn(24, 40, 4, 4, generator=g)
    wr = w.double().requires_grad_(True)
    (R.subpixel_weight(wr) * dwd.double()).sum().backward()                  # the fold is the transpose of the linear map W -> WD
    dw = ops.subpixel_weight_fold(dwd.to(DEV))
    assert rel_err(dw.cpu(), wr.grad) < 1e-6
    acc = torch.ones(40, 24, 3, 3, device=DEV)
    ops.subpixel_weight_fold(dwd.to(DEV), dw_out=acc, accumulate=True)
    assert rel_err(acc.cpu(), wr.grad + 1) < 1e-6


@pytest.mark.parametrize("shape", [(3, 50, 64), (1000, 512), (7, 33, 8)])
def test_colsum_bf16(shape):
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(*shape, generator=g).to(BF)
    ref = x.double().reshape(-1, shape[-1]).sum(0)
    out = ops.colsum(x.to(DEV))
    assert rel_err(out.cpu(), ref) < 1e-5
    out2 = ops.colsum(x.to(DEV), out=out.clone(), accumulate=True)
    assert rel_err(out2.cpu(), 2 * ref) < 1e-5


CASES = [  # N, H, W, Cin, Cout
    (2, 8, 8, 32, 32),        # the decoder's conv_in.0 (post_init: Upsample(z_channels)); small-shape kernels
    (1, 20, 28, 64, 96),      # ragged, W not a power of two
    (1, 64, 64, 64, 128),     # large-shape route: 128-row cout tile
    (2, 64, 64, 128, 256),    # large-shape route: 256-row cout tile, wgrad rows of 64 / 128 pixels
    (1, 64, 72, 96, 320),     # ragged cout, three channel chunks
]


@pytest.mark.parametrize("case", CASES)
def test_subpixel_upsample_conv_vs_two_step_fp64(case):
    """All three products against the reference's two-step form in fp64.  Operands are bf16-exact and the weights sit on a grid where the pre-added taps are
    bf16-exact too, so both forms are the same exact sum of products and only the f32 accumulation order differs."""
    from dmvae_amd import functional as Fn
    n, h, w_, cin, cout = case
    g = torch.Generator().manual_seed(100 + cin + cout)
    x = torch.randn(n, h, w_, cin, generator=g).to(BF)
    w = _coarse((cout, cin, 3, 3), g, 2.0 ** -7)
    b = torch.randn(cout, generator=g)
    dy = torch.randn(n, 2 * h, 2 * w_, cout, generator=g).to(BF)
    yr, dxr, dwr, dbr = _ref_two_step(x.float(), w, b, dy.float())
    xg = x.to(DEV).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True)
    y = Fn.ConvFn.apply(xg, wg, bg, 3, True)
    assert y.shape == (n, 2 * h, 2 * w_, cout) and y.dtype == BF
    y.backward(dy.to(DEV))
    assert rel_err(y.float().cpu(), yr) < 2.0 ** -8            # the bf16 store of the result
    assert rel_err(xg.grad.float().cpu(), dxr) < 2.0 ** -8
    assert rel_err(wg.grad.cpu(), dwr) < 1e-5
    assert rel_err(bg.grad.cpu(), dbr) < 1e-5
    # f32 results of the two conv launches themselves (no bf16 store in the way)
    from dmvae_amd import ops
    y32 = ops.conv2d_nhwc(x.to(DEV), Fn.packed(wg, True, subpixel=True), bg.detach(), ks=4, stride=2, transposed=True, out_f32=True)
    assert rel_err(y32.cpu(), yr) < 1e-5
    dx32 = ops.conv2d_nhwc(dy.to(DEV), Fn.packed(wg, False, subpixel=True), ks=4, stride=2, out_f32=True)
    assert rel_err(dx32.cpu(), dxr) < 1e-5
    # element-wise on a random 1 % sample (rel-to-max alone would hide a wrong border pixel)
    idx = torch.randint(0, yr.numel(), (max(64, yr.numel() // 100),), generator=g)
    a, r = y32.cpu().double().flatten()[idx], yr.flatten()[idx]
    assert ((a - r).abs() <= 1e-5 * yr.abs().max() + 1e-4 * r.abs()).all()
    idx = torch.randint(0, dxr.numel(), (max(64, dxr.numel() // 100),), generator=g)
    a, r = dx32.cpu().double().flatten()[idx], dxr.flatten()[idx]
    assert ((a - r).abs() <= 1e-5 * dxr.abs().max() + 1e-4 * r.abs()).all()


@pytest.mark.parametrize("case", [(1, 20, 28, 64, 96), (2, 64, 64, 128, 256)])
def test_subpixel_route_vs_gather_route(case, monkeypatch):
    """Arbitrary weights: the sub-pixel route and the first implementation (functional.UPS_SUBPIXEL = False) differ only by where bf16 rounding falls on the weights."""
    from dmvae_amd import functional as Fn
    n, h, w_, cin, cout = case
    g = torch.Generator().manual_seed(7)
    x = torch.randn(n, h, w_, cin, generator=g).to(BF).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    dy = torch.randn(n, 2 * h, 2 * w_, cout, generator=g).to(BF).to(DEV)
    res = []
    for flag in (True, False):
        monkeypatch.setattr(Fn, "UPS_SUBPIXEL", flag)
        xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = Fn.ConvFn.apply(xg, wg, bg, 3, True)
        y.backward(dy)
        res.append((y.float(), xg.grad.float(), wg.grad, bg.grad))
    rl2 = lambda a, r: ((a.double() - r.double()).norm() / r.double().norm()).item()
    assert rl2(res[0][0], res[1][0]) < 6e-3 and rl2(res[0][1], res[1][1]) < 8e-3          # bf16 outputs, weights rounded at different sites
    assert rl2(res[0][2], res[1][2]) < 1e-5 and rel_err(res[0][3], res[1][3]) < 1e-5       # weight / bias gradients: no weight rounding involved


@pytest.mark.parametrize("case", [  # N, H, W, Cin, Cout, ks, kind (0 plain, 1 per-parity transposed 4x4 stride 2), residual
    (2, 64, 64, 128, 256, 3, 0, True),      # 256-row tile
    (2, 64, 64, 64, 128, 3, 0, False),      # 128-row tile: 4 channels per group
    (1, 128, 128, 256, 512, 1, 0, True),    # 1x1 + residual (AttnBlock.proj_out), two cout tiles
    (3, 32, 32, 256, 256, 3, 1, False),     # Upsample's sub-pixel forward: parity-interleaved tiles
    (2, 48, 40, 128, 256, 3, 0, False),     # H*W not a multiple of the pixel tile: the two-pass route behind the same entry point
    (2, 16, 16, 32, 64, 3, 0, False),       # small shape (general conv kernel) + two-pass statistics
])
def test_conv_gnstats_matches_separate_statistics_pass(case):
    """dmvae_conv2d_nhwc_fwd_gnstats: same y as dmvae_conv2d_nhwc_fwd bit for bit, and the (mean, rstd) its epilogue partials give equal those of the
    statistics pass over y (and the f64 statistics of y) to f32 summation order."""
    from dmvae_amd import ops
    n, h, w_, cin, cout, ks, kind, res = case
    g = torch.Generator().manual_seed(31 + cin + cout)
    x = torch.randn(n, h, w_, cin, generator=g).to(BF).to(DEV)
    kw = dict(ks=4, stride=2, transposed=True) if kind else dict(ks=ks)
    taps = 16 if kind else ks * ks
    wt = (torch.randn(cout, taps, cin, generator=g) * 0.05).to(BF).to(DEV)
    b = (torch.randn(cout, generator=g) * 2).to(DEV)
    ho, wo = (2 * h, 2 * w_) if kind else (h, w_)
    r = torch.randn(n, ho, wo, cout, generator=g).to(BF).to(DEV) if res else None
    y0 = ops.conv2d_nhwc(x, wt, b, r, **kw)
    y, st = ops.conv2d_nhwc_gnstats(x, wt, b, r, **kw)
    assert torch.equal(y, y0) and st.shape == (n, 32, 2)
    st0 = ops.groupnorm_stats(y0)
    yd = y0.double().view(n, ho * wo, 32, cout // 32)
    mean = yd.mean(dim=(1, 3))
    rstd = 1.0 / torch.sqrt(yd.var(dim=(1, 3), unbiased=False) + 1e-6)
    assert rel_err(st[..., 0], mean) < 1e-5 and rel_err(st[..., 1], rstd) < 1e-5
    assert rel_err(st, st0) < 1e-5
    y2, st2 = ops.conv2d_nhwc_gnstats(x, wt, b, r, **kw)
    assert torch.equal(st, st2)                                          # fixed summation order


def test_bias_gradient_rides_on_the_groupnorm_backward():
    """Upsample -> ResnetBlock (flux_ae.py:257-262): the ResnetBlock's norm1 backward writes the Upsample conv's output gradient and sums it per channel on the
    way (dmvae_groupnorm_bwd_colsum); ConvFn.backward takes that as the bias gradient instead of a separate pass.  Same bias gradient as the separate pass
    (DMVAE_* off is not needed: the fallback is exercised by calling the layer alone), and the kernel pair agrees with the plain backward bit for bit on dx."""
    from dmvae_amd import ops
    from dmvae_amd.models.flux_ae import ResnetBlock, Upsample
    g = torch.Generator().manual_seed(17)
    up, blk = Upsample(128).to(DEV), ResnetBlock(128, 128).to(DEV)
    with torch.no_grad():
        for p in list(up.parameters()) + list(blk.parameters()):
            p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.3))
    x = torch.randn(2, 32, 32, 128, generator=g).to(BF).to(DEV)
    dy = torch.randn(2, 64, 64, 128, generator=g).to(BF).to(DEV)

    def run(chain):
        for p in list(up.parameters()) + list(blk.parameters()):
            p.grad = None
        h = up.forward_nhwc(x)
        (blk.forward_nhwc(h) if chain else h).backward(dy)
        return up.conv.bias.grad.clone(), up.conv.weight.grad.clone()
    h = up.forward_nhwc(x)
    assert getattr(h, "_dmvae_want_colsum", False)
    db_chain, dw_chain = run(True)
    # reference for the chained case: the block's input gradient summed over pixels, from a second backward through the block alone
    hh = up.forward_nhwc(x).detach().requires_grad_(True)
    blk.forward_nhwc(hh).backward(dy)
    assert getattr(hh.grad, "_dmvae_colsum", None) is None                         # not requested: hh does not come from the sub-pixel conv
    want = hh.grad.float().sum(dim=(0, 1, 2))
    assert rel_err(db_chain, want) < 1e-5
    db_alone, _ = run(False)                                                       # the layer alone: the separate column-sum pass over dy
    assert rel_err(db_alone, dy.float().sum(dim=(0, 1, 2))) < 1e-5
    # kernel level: dx identical with and without the by-product; column sums = sums of the stored dx
    xs = torch.randn(2, 64 * 64, 128, generator=g).to(BF).to(DEV)
    da = torch.randn(2, 64 * 64, 128, generator=g).to(BF).to(DEV)
    st = ops.groupnorm_stats(xs)
    gw, gb = torch.rand(128, generator=g).to(DEV) + 0.5, torch.randn(128, generator=g).to(DEV)
    dx0, dg0, db0 = ops.groupnorm_bwd(da, xs, st, gw, gb, True, dres=da)
    dx1, dg1, db1 = ops.groupnorm_bwd(da, xs, st, gw, gb, True, dres=da, want_colsum=True)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert rel_err(dx1._dmvae_colsum[0], dx1.float().sum(dim=(0, 1))) < 1e-5
