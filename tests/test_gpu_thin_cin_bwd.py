"""ConvFn.backward of a 3x3 conv FROM few channels (the decoder's conv_in, models/flux_ae.py:196: z_channels = 32 -> 512 at 32 x 32) as GEMMs on the im2col
form (-m gpu): weight gradient = the 1x1 weight-gradient kernel on [M, 12 taps x cin] (three taps of zeros), input gradient = Linear GEMM + the gather adjoint
of im2col in f32 -- against the direct kernels they replace and against f64 autograd of the oracle's conv on the same bf16 operands."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _run(n, h, w, cin, cout, gemm, monkeypatch, seed=0):
    from dmvae_amd import functional as Fn
    monkeypatch.setattr(Fn, "THIN_CIN_BWD_AS_GEMM", gemm)
    g = torch.Generator().manual_seed(seed + n + cin)
    x = torch.randn(n, h, w, cin, generator=g).to(DEV).to(BF).requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV).requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).to(DEV).requires_grad_(True)
    dy = torch.randn(n, h, w, cout, generator=g).to(DEV).to(BF)
    y = Fn.ConvFn.apply(x, wt, b, 3, False)
    y.backward(dy)
    return x, wt, b, dy, y.detach(), x.grad, wt.grad, b.grad


@pytest.mark.parametrize("n,h,w,cin,cout", [(16, 32, 32, 32, 512), (4, 64, 64, 64, 384)])
def test_thin_cin_backward_as_gemms(n, h, w, cin, cout, monkeypatch):
    x, wt, b, dy, y1, dx1, dw1, db1 = _run(n, h, w, cin, cout, True, monkeypatch)
    _, _, _, _, y0, dx0, dw0, db0 = _run(n, h, w, cin, cout, False, monkeypatch)
    assert torch.equal(y1, y0)
    # f64 on the same bf16 operands
    xd = x.detach().double().permute(0, 3, 1, 2).requires_grad_(True)
    wd = wt.detach().to(BF).double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True)
    F.conv2d(xd, wd, bd, padding=1).backward(dy.double().permute(0, 3, 1, 2))
    ref_dx = xd.grad.permute(0, 2, 3, 1)
    assert rel_err(dw1, wd.grad) < 1e-4 and rel_err(db1, bd.grad) < 1e-4               # f32 accumulation of exact products
    assert rel_err(dx1.double(), ref_dx) < 2 ** -8                                     # one bf16 rounding of an f32 sum
    # and no further from f64 than the direct kernels
    assert rel_err(dw1, wd.grad) <= 1.5 * rel_err(dw0, wd.grad) + 1e-6
    assert rel_err(dx1.double(), ref_dx) <= 1.1 * rel_err(dx0.double(), ref_dx) + 1e-6
    assert rel_err(dw1, dw0) < 1e-4 and rel_err(dx1.float(), dx0.float()) < 2 ** -7 and rel_err(db1, db0) < 1e-5


def test_im2col_with_padded_taps():
    from dmvae_amd import ops
    x = torch.randn(2, 8, 16, 32, generator=torch.Generator().manual_seed(1)).to(DEV).to(BF)
    c9, c12 = ops.im2col(x, 3, 1, 1), ops.im2col(x, 3, 1, 1, taps_pad=12)
    assert tuple(c12.shape) == (2, 8, 16, 384)
    assert torch.equal(c12[..., :288], c9) and not c12[..., 288:].any()
