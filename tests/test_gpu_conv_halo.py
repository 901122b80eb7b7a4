"""The kx-halo form of csrc/conv_pp.hip (template flag HALO: plain 3x3, bf16 result -- every decoder ResnetBlock conv with 256 / 512 output
channels, models/flux_ae.py:63,65, and their input gradients): the three kx taps of a (channel chunk, ky) read one staged halo of the pixel tile.
What is new there is geometry, so the cases are geometric: image rows shorter / longer than a tile and not dividing it, tiles that span several images,
ragged pixel counts, odd chunk counts (halo-slot parity, the A-ring slot of the trailing all-zero K tiles), one chunk.  Reference: fp64 conv on the same
bf16-rounded operands; and bit equality with the f32-output route, which stages every tap on its own (the K order and hence every sum is the same)."""
import pytest
import torch

from conftest import rel_err
from test_gpu_kernels import _conv_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

CASES = [  # N, H, W, Cin, Cout
    (2, 96, 100, 32, 192),     # W does not divide the tile, ragged last tile, one chunk per tap, ragged cout
    (128, 12, 12, 64, 256),    # 144-pixel images: every tile spans image boundaries; two chunks (18 K tiles: trailing tiles start at A slot 2)
    (5, 60, 68, 96, 136),      # odd chunk count, ragged everything
    (1, 16, 1040, 160, 256),   # rows of four tiles + 16 pixels
    (72, 16, 16, 128, 512),    # one image per tile, two cout tiles
    (3, 8, 700, 32, 144),      # short, wide images
    (2, 96, 100, 64, 128),     # the 128 x 512 tile (Cout <= 128): 514-pixel halo, 33 pieces
    (160, 12, 12, 96, 64),     # ... tiles across images, odd chunk count, half of the tile's couts padding
    (1, 24, 1050, 32, 96),     # ... rows of two tiles + 26 pixels
    (8192, 1, 2, 32, 192),     # two-pixel images: every pixel is a left or a right edge, no row above or below
    (1, 1, 16400, 64, 64),     # one row: the 64 x 1024 tile, only the centre kernel row contributes
    (40, 20, 21, 32, 64),      # ... odd row length, ragged last tile
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("act,use_res", [(0, False), (1, True)])
def test_halo_conv_geometry(case, act, use_res):
    from dmvae_amd import ops
    n, h, w_, cin, cout = case
    g = torch.Generator(device="cpu").manual_seed(3 + cin + cout + w_)
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    r = torch.randn(n, h, w_, cout, generator=g).to(DEV).to(BF) if use_res else None
    wp = ops.pack_conv_weight(w)
    ref = _conv_ref(x, w.to(BF), b, r, 3, 0, act)
    y16 = ops.conv2d_nhwc(x, wp, b, r, ks=3, act=act)              # HALO instantiation
    y32 = ops.conv2d_nhwc(x, wp, b, r, ks=3, act=act, out_f32=True)  # every tap staged on its own
    assert rel_err(y32.cpu(), ref) < 1e-5
    assert torch.equal(y16, y32.to(BF))
    assert torch.equal(y16, ops.conv2d_nhwc(x, wp, b, r, ks=3, act=act))   # rerun: bit-identical


def test_halo_conv_edge_columns_are_zero_padded():
    """An all-ones image and all-ones weights: the result counts the taps inside the image, so a wrapped neighbour row shows up as a wrong integer."""
    from dmvae_amd import ops
    n, h, w_, cin, cout = 2, 100, 84, 32, 256
    x = torch.ones(n, h, w_, cin, device=DEV, dtype=BF)
    wp = ops.pack_conv_weight(torch.ones(cout, cin, 3, 3, device=DEV))
    y = ops.conv2d_nhwc(x, wp, ks=3, out_f32=False).float()
    ry = torch.full((h,), 3.0); ry[0] = ry[-1] = 2.0
    rx = torch.full((w_,), 3.0); rx[0] = rx[-1] = 2.0
    want = (ry[:, None] * rx[None, :] * cin).to(DEV)
    assert torch.equal(y, want[None, :, :, None].expand(n, h, w_, cout))


@pytest.mark.parametrize("case", [(2, 96, 100, 64, 192), (2, 96, 100, 64, 128), (1, 128, 130, 96, 64)])
@pytest.mark.parametrize("for_dgrad", [False, True])
def test_halo_conv_kmajor_weights_bit_identical(case, for_dgrad):
    """The K-tile-major weight copy (dmvae_pack_conv_weight_v2 / dmvae_conv_desc.w_layout = 1) holds the same values: same result, bit for bit."""
    from dmvae_amd import ops, _lib
    import ctypes
    n, h, w_, cin, cout = case
    g = torch.Generator(device="cpu").manual_seed(17 + cin + cout)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    if for_dgrad:
        cin, cout = cout, cin
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    b = torch.randn(cout, generator=g).to(DEV)
    w0 = ops.pack_conv_weight(w, for_dgrad)
    w1 = ops.pack_conv_weight(w, for_dgrad, kmajor=True)
    assert torch.equal(w0, w1) and hasattr(w1, "_dmvae_kmajor") and not hasattr(w0, "_dmvae_kmajor")
    T = 9
    assert torch.equal(w1._dmvae_kmajor, w0.view(cout, T, cin // 32, 32).permute(2, 1, 0, 3).contiguous())
    d = _lib.ConvDesc(n, h, w_, cin, cout, 3, 0, 0, 0, 1, 0)
    assert _lib.lib().dmvae_conv_halo_applies(ctypes.byref(d)) == 1
    assert torch.equal(ops.conv2d_nhwc(x, w1, b, ks=3), ops.conv2d_nhwc(x, w0, b, ks=3))
    y1, s1 = ops.conv2d_nhwc_gnstats(x, w1, b, ks=3, groups=32 if cout % 32 == 0 else 8)
    y0, s0 = ops.conv2d_nhwc_gnstats(x, w0, b, ks=3, groups=32 if cout % 32 == 0 else 8)
    assert torch.equal(y1, y0) and torch.equal(s1, s0)


def test_kmajor_layout_is_refused_where_the_large_tile_kernel_does_not_run():
    """w_layout = 1 is an error -- never a silent fallback -- for a call the first-generation kernel serves (dmvae_conv_kmajor_applies = 0); where the large-tile
    kernel runs without the halo form (1x1 here) the K-tile-major operand is accepted since round 3 (dmvae_conv_kmajor_applies = 1, dmvae_conv_halo_applies = 0)."""
    from dmvae_amd import _lib
    import ctypes
    L = _lib.lib()
    small = _lib.ConvDesc(1, 16, 16, 64, 64, 3, 0, 0, 0, 1, 0, 1)            # too few pixels for the large-tile kernel
    assert L.dmvae_conv_halo_applies(ctypes.byref(small)) == 0 and L.dmvae_conv_kmajor_applies(ctypes.byref(small)) == 0
    xx = torch.zeros(1, 16, 16, 64, device=DEV, dtype=BF); yy = torch.zeros(1, 16, 16, 64, device=DEV, dtype=BF)
    ww = torch.zeros(64, 9, 64, device=DEV, dtype=BF)
    assert L.dmvae_conv2d_nhwc_fwd(xx.data_ptr(), ww.data_ptr(), None, None, yy.data_ptr(), ctypes.byref(small), None) != 0
    assert b"w_layout" in L.dmvae_last_error()
    bad = _lib.ConvDesc(64, 16, 16, 64, 64, 1, 0, 0, 0, 1, 0, 2)             # no such layout
    assert L.dmvae_conv2d_nhwc_fwd(xx.data_ptr(), ww.data_ptr(), None, None, yy.data_ptr(), ctypes.byref(bad), None) != 0
    one = _lib.ConvDesc(64, 16, 16, 64, 64, 1, 0, 0, 0, 1, 0, 1)             # 1x1 on the large-tile kernel: not the halo form, K-tile-major accepted
    assert L.dmvae_conv_halo_applies(ctypes.byref(one)) == 0 and L.dmvae_conv_kmajor_applies(ctypes.byref(one)) == 1


def _kmajor_copy(wp):
    """[rows][T][cols] tap-major operand -> its K-tile-major copy [cols / 32][T][rows][32] by plain indexing."""
    rows, T, cols = wp.shape
    return wp.view(rows, T, cols // 32, 32).permute(2, 1, 0, 3).contiguous()


@pytest.mark.parametrize("kind", ["1x1", "4x4_stride2", "4x4_transposed", "4x4_transposed_gnstats"])
def test_kmajor_weights_bit_identical_on_the_non_halo_instantiations(kind):
    """Round 3: the per-parity (sub-pixel transpose), general-gather (4x4 stride 2) and 1x1 instantiations read the weight operand through the same three strides as
    the halo form, so they take the K-tile-major copy too: same values, same K order -> the same bits as with the tap-major operand (raw C-ABI calls, both layouts)."""
    from dmvae_amd import _lib, ops
    import ctypes
    L = _lib.lib()
    g = torch.Generator().manual_seed(23)
    n, h, w_, cin, cout = 4, 64, 64, 128, 256
    if kind == "1x1":
        ks, stride, transposed = 1, 0, 0
        ho, wo = h, w_
    elif kind == "4x4_stride2":
        ks, stride, transposed = 4, 2, 0
        n = 16                                   # 16 x 32 x 32 output pixels: the large-tile kernel's minimum
        ho, wo = h // 2, w_ // 2
    else:
        ks, stride, transposed = 4, 2, 1
        ho, wo = 2 * h, 2 * w_
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    wp = (torch.randn(cout, ks * ks, cin, generator=g) * 0.05).to(DEV).to(BF)     # any values: the two layouts only have to agree
    wk = _kmajor_copy(wp)
    b = torch.randn(cout, generator=g).to(DEV)
    outs = []
    for layout, wt in ((0, wp), (1, wk)):
        d = _lib.ConvDesc(n, h, w_, cin, cout, ks, 0, 0, 0, stride, transposed, layout)
        assert L.dmvae_conv_kmajor_applies(ctypes.byref(d)) == 1 and L.dmvae_conv_halo_applies(ctypes.byref(d)) == 0
        y = torch.full((n, ho, wo, cout), 7.0, device=DEV, dtype=BF)
        if kind.endswith("gnstats"):
            wsb = L.dmvae_conv2d_nhwc_fwd_gnstats_workspace(ctypes.byref(d), 32)
            ws = torch.zeros(max(wsb, 4) // 4, device=DEV)
            st = torch.zeros(n, 32, 2, device=DEV)
            ops.check(L.dmvae_conv2d_nhwc_fwd_gnstats(x.data_ptr(), wt.data_ptr(), b.data_ptr(), None, y.data_ptr(), st.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                                      32, 1e-6, ctypes.byref(d), ops._stream()), "conv2d_nhwc_fwd_gnstats")
            outs.append((y, st))
        else:
            ops.check(L.dmvae_conv2d_nhwc_fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), None, y.data_ptr(), ctypes.byref(d), ops._stream()), "conv2d_nhwc_fwd")
            outs.append((y,))
    torch.cuda.synchronize()
    for t0, t1 in zip(outs[0], outs[1]):
        assert torch.equal(t0, t1), kind
    assert float(outs[0][0].float().abs().max()) > 0.1 and not bool((outs[0][0] == 7.0).all())      # something was computed
