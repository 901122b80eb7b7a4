"""Per-kernel parity of the ping-pong conv kernels (csrc/conv_pp.hip forward / input gradient, csrc/conv_wgrad_pp.hip weight gradient) at the
channel counts and resolutions of the measured configuration C2 (512->512 @64^2 and @128^2, 512->256 @128^2, 256->256 @128^2,
256->128 @256^2, the folded nearest-x2 512->512 @64->128), batch 4 and one batch-32 case -- the shapes `bench.py` spends its time in: 16 channel
chunks per tap, the chunk-outer / tap-inner K order, persistent blocks walking many tiles, split-K over up to 2 M pixels.

Reference: the same contraction in fp64 ON THE GPU over the same bf16-rounded operands, written as nine shifted [pixels, Cin] x [Cin, Cout] matrix
products (rocBLAS dgemm; independent of any conv library and of this build's kernels).  Bars, per assert:
  * max |y - ref| / max |ref| < 1e-5 over EVERY element (f32 accumulation order is all that differs: products of bf16 operands are exact in f32);
  * on a random 1 % sample, element by element: |y_i - ref_i| <= 1e-4 |ref_i| + 2e-5 rms(ref)  (the second term is the f32 accumulation noise of a
    K = 4608 ... 2 M term sum, which no element can be expected to beat however small its own value is);
  * the bf16 result is bit-for-bit the round-to-nearest-even of the f32 result;
  * two runs are bit-identical (no atomics, fixed-order split-K).
Reference sites: models/flux_ae.py:63,65,67,101 (ResnetBlock / Upsample convs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

#        N,  H,   W,   Cin, Cout, ups
SHAPES = [
    (4, 64, 64, 512, 512, 0),
    (4, 128, 128, 512, 512, 0),
    (4, 128, 128, 512, 256, 0),
    (4, 128, 128, 256, 256, 0),
    (4, 256, 256, 256, 128, 0),
    (4, 64, 64, 512, 512, 1),        # Upsample: nearest x2 folded into the gather, 64^2 -> 128^2
    (32, 64, 64, 512, 512, 0),       # the bench's batch
    (32, 128, 128, 256, 256, 0),     # ... 2^19 pixels: from here on the weight gradient runs the 128 x 384 halo tile on these channel counts too
]
IDS = ["%dx%d^2_%d-%d%s" % (n, h, ci, co, "_ups" if u else "") for n, h, w, ci, co, u in SHAPES]


def _shift_views(x64, ks=3):
    """x64 [N,H,W,C] double -> zero-padded once; yields (ky, kx, contiguous [N*H*W, C] view of the input pixel each output pixel reads at that tap)."""
    n, h, w, c = x64.shape
    xp = torch.zeros(n, h + 2, w + 2, c, dtype=torch.double, device=x64.device)
    xp[:, 1:-1, 1:-1] = x64
    for ky in range(ks):
        for kx in range(ks):
            yield ky, kx, xp[:, ky:ky + h, kx:kx + w].reshape(n * h * w, c)


def conv3x3_ref64(x, w, ups=False):
    """y[n,h,w,co] = sum_{ky,kx,ci} xpad[n,h+ky,w+kx,ci] * w[co,ci,ky,kx] in fp64; x [N,H,W,Cin] (bf16 values), w [Cout,Cin,3,3]."""
    x64 = x.double()
    if ups:
        x64 = x64.repeat_interleave(2, 1).repeat_interleave(2, 2)
    n, h, wd, _ = x64.shape
    w64 = w.double()
    y = torch.zeros(n * h * wd, w.shape[0], dtype=torch.double, device=x.device)
    for ky, kx, xs in _shift_views(x64):
        y.addmm_(xs, w64[:, :, ky, kx].t())
    return y.view(n, h, wd, -1)


def wgrad3x3_ref64(dy, a, ups=False):
    """dW[co,ci,ky,kx] = sum_px dy[px,co] * apad[px + tap, ci], db[co] = sum_px dy[px,co], fp64."""
    a64 = a.double()
    if ups:
        a64 = a64.repeat_interleave(2, 1).repeat_interleave(2, 2)
    d2 = dy.double().reshape(-1, dy.shape[-1])
    dw = torch.empty(dy.shape[-1], a.shape[-1], 3, 3, dtype=torch.double, device=a.device)
    for ky, kx, xs in _shift_views(a64):
        dw[:, :, ky, kx] = d2.t() @ xs
    return dw, d2.sum(0)


def _check(y, ref, what, sample_frac=0.01, seed=0):
    ref = ref.to(y.device)
    scale = ref.abs().max().item()
    err = (y.double() - ref).abs().max().item() / scale
    assert err < 1e-5, f"{what}: max|d|/max|ref| = {err:.2e}"
    flat_y, flat_r = y.reshape(-1), ref.reshape(-1)
    g = torch.Generator(device=y.device).manual_seed(seed)
    idx = torch.randint(0, flat_r.numel(), (max(1000, int(flat_r.numel() * sample_frac)),), device=y.device, generator=g)
    rms = ref.pow(2).mean().sqrt().item()
    d = (flat_y[idx].double() - flat_r[idx]).abs()
    bound = 1e-4 * flat_r[idx].abs() + 2e-5 * rms
    worst = (d / bound).max().item()
    assert worst <= 1.0, f"{what}: element-wise bound exceeded by {worst:.2f}x on the 1% sample"
    return err


@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_conv_pp_forward_at_c2_shapes(shape):
    from dmvae_amd import ops
    n, h, w_, cin, cout, ups = shape
    g = torch.Generator().manual_seed(100 + cin + cout + h)
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin) ** 0.5)).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    wb = w.to(BF)
    ref = conv3x3_ref64(x, wb, bool(ups)) + b.double()
    wp = ops.pack_conv_weight(w)
    y32 = ops.conv2d_nhwc(x, wp, b, None, ks=3, upsample=bool(ups), act=0, out_f32=True)
    _check(y32, ref, "forward f32")
    y16 = ops.conv2d_nhwc(x, wp, b, None, ks=3, upsample=bool(ups), act=0)
    assert torch.equal(y16, y32.to(BF))                      # bf16 store = RNE of the f32 accumulator
    assert torch.equal(y16, ops.conv2d_nhwc(x, wp, b, None, ks=3, upsample=bool(ups), act=0))
    if n <= 4:                                               # residual + swish epilogue at the real channel counts
        r = torch.randn(y16.shape, generator=g).to(DEV).to(BF)
        pre = ref + r.double()
        ya = ops.conv2d_nhwc(x, wp, b, r, ks=3, upsample=bool(ups), act=1, out_f32=True)
        _check(ya, pre * torch.sigmoid(pre), "forward + residual + swish")


@pytest.mark.parametrize("shape", [s for s in SHAPES if not s[5]], ids=[i for i, s in zip(IDS, SHAPES) if not s[5]])
def test_conv_pp_input_gradient_at_c2_shapes(shape):
    """dx = conv(dy, W flipped and transposed): the same kernel on dmvae_pack_conv_weight(for_dgrad=1) operands."""
    from dmvae_amd import ops
    n, h, w_, cin, cout, _ = shape
    g = torch.Generator().manual_seed(200 + cin + cout + h)
    dy = torch.randn(n, h, w_, cout, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cout) ** 0.5)).to(DEV)
    wflip = w.to(BF).flip(2, 3).transpose(0, 1).contiguous()          # [cin, cout, ky, kx]
    ref = conv3x3_ref64(dy, wflip)
    dx = ops.conv2d_nhwc(dy, ops.pack_conv_weight(w, for_dgrad=True), ks=3, out_f32=True)
    _check(dx, ref, "input gradient f32")
    dx16 = ops.conv2d_nhwc(dy, ops.pack_conv_weight(w, for_dgrad=True), ks=3)
    assert torch.equal(dx16, dx.to(BF))


@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_wgrad_pp_at_c2_shapes(shape):
    from dmvae_amd import ops
    n, h, w_, cin, cout, ups = shape
    g = torch.Generator().manual_seed(300 + cin + cout + h)
    ho, wo = (2 * h, 2 * w_) if ups else (h, w_)
    a = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    dy = torch.randn(n, ho, wo, cout, generator=g).to(DEV).to(BF)
    dw_ref, db_ref = wgrad3x3_ref64(dy, a, bool(ups))
    dw, db = ops.conv2d_nhwc_wgrad(dy, a, 3, upsample=bool(ups))
    _check(dw, dw_ref, "weight gradient")
    _check(db, db_ref, "bias gradient", sample_frac=1.0)
    dw2, db2 = ops.conv2d_nhwc_wgrad(dy, a, 3, upsample=bool(ups))
    assert torch.equal(dw, dw2) and torch.equal(db, db2)     # fixed-order split-K: run-to-run bit-identical


def test_folded_upsample_input_gradient_512_64_to_128():
    """Upsample.forward (flux_ae.py:103-107) backward at C2 size: dx = sum over each 2x2 block of the stride-1 input gradient at 128^2."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(77)
    n, h, c = 4, 64, 512
    dy = torch.randn(n, 2 * h, 2 * h, c, generator=g).to(DEV).to(BF)
    w = (torch.randn(c, c, 3, 3, generator=g) * (1.0 / (9 * c) ** 0.5)).to(DEV)
    wflip = w.to(BF).flip(2, 3).transpose(0, 1).contiguous()
    hi = conv3x3_ref64(dy, wflip)
    ref = hi.view(n, h, 2, h, 2, c).sum(dim=(2, 4))
    dxh = ops.conv2d_nhwc(dy, ops.pack_conv_weight(w, for_dgrad=True), ks=3)           # bf16 at 128^2, as functional.UpsampleConvFn stores it
    dx = ops.sumpool2x2(dxh)
    # the product path rounds the 128^2 gradient to bf16 before the 2x2 sum: compare against the same rounding of the fp64 reference
    ref_q = hi.float().to(BF).double().view(n, h, 2, h, 2, c).sum(dim=(2, 4))
    scale = ref.abs().max().item()
    assert (dx.double() - ref_q).abs().max().item() / scale < 4e-3      # bf16 storage of dx (2^-8 relative) and of the hi-res gradient
    assert (dx.double() - ref).abs().max().item() / scale < 8e-3
