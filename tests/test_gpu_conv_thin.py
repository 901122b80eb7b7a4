"""The halo-tile kernel for 3x3 convs to four f32 output channels (csrc/conv_thin.hip): the decoder's conv_out (models/flux_ae.py:237,274, 128 -> 3 at
the image resolution) and the input gradient of the LPIPS trunk's first layer (models/lpips.py:116-153, 64 -> 3).  Reference: the contraction in fp64 on
the GPU over the same bf16-rounded operands (tests/test_gpu_conv_c2_shapes.py::conv3x3_ref64); bar 1e-5 of the largest value on every element plus the
element-wise bound on a 1 % sample; edge tiles (zero padding), several images, bias / no bias; run-to-run bit-identical."""
import pytest
import torch

from test_gpu_conv_c2_shapes import _check, conv3x3_ref64

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

#        N, H,   W,  Cin, bias
CASES = [
    (2, 4, 32, 128, True),          # one tile per image: every halo pixel is padding
    (3, 8, 64, 128, False),
    (1, 16, 96, 64, True),          # CIN 64: two pixels per LDS bank row
    (2, 12, 32, 64, False),
    (2, 256, 256, 128, True),       # conv_out at the image resolution
    (2, 256, 256, 64, False),       # VGG conv1_1 input gradient
]


@pytest.mark.parametrize("case", CASES, ids=["%dx%dx%d_%d%s" % (c[0], c[1], c[2], c[3], "_b" if c[4] else "") for c in CASES])
def test_conv_thin_matches_fp64(case):
    from dmvae_amd import ops
    n, h, w_, cin, with_bias = case
    g = torch.Generator().manual_seed(7 + h + cin)
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    w = (torch.randn(3, cin, 3, 3, generator=g) * (1.0 / (9 * cin) ** 0.5)).to(DEV)
    b = torch.randn(3, generator=g).to(DEV) if with_bias else None
    wp = ops.pack_conv_weight(w, rows_pad=4)                       # three real output channels in a four-row operand, as functional.NormConvOutFn packs it
    bp = torch.cat([b, b.new_zeros(1)]) if with_bias else None
    y = ops.conv2d_nhwc(x, wp, bp, None, ks=3, act=0, out_f32=True)
    assert y.shape == (n, h, w_, 4) and y.dtype == torch.float32
    ref = conv3x3_ref64(x, w.to(BF))
    if with_bias:
        ref = ref + b.double()
    _check(y[..., :3], ref, "conv_thin")
    assert torch.count_nonzero(y[..., 3]) == 0                     # the padding row: zero weights, zero bias
    assert torch.equal(y, ops.conv2d_nhwc(x, wp, bp, None, ks=3, act=0, out_f32=True))


#          N, H,  W,  cout
WCASES = [(2, 4, 32, 3), (1, 9, 64, 3), (3, 16, 96, 4), (2, 256, 256, 3), (1, 5, 32, 1)]


@pytest.mark.parametrize("case", WCASES, ids=["%dx%dx%d_co%d" % c for c in WCASES])
def test_conv_out_wgrad_matches_fp64(case):
    """csrc/wgrad_thin.hip: dW of the decoder's conv_out from the NCHW f32 image gradient and the NHWC bf16 conv input (flux_ae.py:237,274)."""
    from dmvae_amd import ops
    from test_gpu_conv_c2_shapes import wgrad3x3_ref64
    n, h, w_, cout = case
    g = torch.Generator().manual_seed(11 + h + w_)
    a = torch.randn(n, h, w_, 128, generator=g).to(DEV).to(BF)
    dy = torch.randn(n, cout, h, w_, generator=g).to(DEV)
    assert ops.conv_out_wgrad_supported(n, h, w_, 128, cout)
    dw = ops.conv_out_wgrad(dy, a)
    ref, _ = wgrad3x3_ref64(dy.to(BF).permute(0, 2, 3, 1).contiguous(), a)          # the kernel rounds the gradient to bf16, like the conv path it replaces
    _check(dw, ref, "conv_out_wgrad")
    assert torch.equal(dw, ops.conv_out_wgrad(dy, a))                               # fixed-order reduction
    acc = torch.full_like(dw, 0.5)
    ops.conv_out_wgrad(dy, a, dw_out=acc, accumulate=True)
    assert torch.allclose(acc, dw + 0.5, rtol=0, atol=1e-6 * dw.abs().max().item() + 1e-7)
    assert not ops.conv_out_wgrad_supported(n, h, w_ + 1, 128, cout) and not ops.conv_out_wgrad_supported(n, h, w_, 64, cout)
