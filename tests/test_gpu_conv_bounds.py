"""Rows past the last output pixel are never written: the conv output sits inside a sentinel-filled buffer and the pixel count is not a multiple of any pixel
tile (the buffer descriptor's range check does not see the scalar offset, so validity must ride on the per-lane offset; csrc/conv_pp.hip epilogues).
Covers the kx-halo instantiations (3x3, bf16 result: stores straight from the accumulators), the staged epilogue (1x1 and f32 results) and the residual
operand read through the same addressing.  Reference sites: nn.Conv2d at models/flux_ae.py:63,65,67."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.mark.parametrize("ks,cin,cout,out_f32,res", [(3, 64, 256, 0, 0), (3, 64, 256, 0, 1), (3, 32, 128, 0, 0), (3, 32, 64, 0, 0), (1, 64, 256, 0, 0), (3, 64, 256, 1, 0),
                                                     (1, 64, 128, 0, 1)])
def test_conv_writes_nothing_past_the_last_pixel(ks, cin, cout, out_f32, res):
    from dmvae_amd import _lib
    from dmvae_amd._lib import ConvDesc
    L = _lib.lib()
    n, h, w = 5, 60, 60                       # 18000 pixels: 70.3 tiles of 256, 35.2 of 512, 17.6 of 1024
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(BF)
    wt = (torch.randn(cout, ks * ks, cin, device="cuda", generator=g) * 0.05).to(BF)
    b = torch.randn(cout, device="cuda", generator=g)
    m = n * h * w
    es = 4 if out_f32 else 2
    pad = 1 << 20
    buf = torch.full((pad + m * cout * es + pad,), 0x5A, dtype=torch.uint8, device="cuda")
    r = torch.randn(m, cout, device="cuda", generator=g).to(BF) if res else None
    d = ConvDesc(n, h, w, cin, cout, ks, 0, 0, out_f32, 1, 0)
    rc = L.dmvae_conv2d_nhwc_fwd(x.data_ptr(), wt.data_ptr(), b.data_ptr(), None if r is None else r.data_ptr(), buf.data_ptr() + pad, ctypes.byref(d),
                                 torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((buf[:pad] == 0x5A).all()), "wrote in front of the output"
    assert bool((buf[pad + m * cout * es:] == 0x5A).all()), "wrote past the last pixel"
    y = buf[pad:pad + m * cout * es].view(torch.float32 if out_f32 else BF).view(m, cout)
    xp = torch.nn.functional.pad(x.double().permute(0, 3, 1, 2), (ks // 2,) * 4)
    ref = torch.nn.functional.conv2d(xp, wt.double().view(cout, ks, ks, cin).permute(0, 3, 1, 2), b.double()).permute(0, 2, 3, 1).reshape(m, cout)
    if res:
        ref = ref + r.double()
    err = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < (1e-5 if out_f32 else 6e-3), err
