"""LPIPS on the HIP kernels (VGG16 trunk on the implicit-GEMM conv kernel + max pools + fused feature diff) vs the CPU oracle
with bf16 rounding at the same tensor sites (oracle/ref_cpu.py::lpips_forward, q=bf16_round)."""
import pytest
import torch

from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu


def _lpips(seed=0):
    from dmvae_amd.utils.lpips import LPIPS
    torch.manual_seed(seed)
    lp = LPIPS().eval()
    with torch.no_grad():
        for m in lp.net.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.normal_(0, (2.0 / (9 * m.weight.shape[1])) ** 0.5)
                m.bias.normal_(0, 0.05)
        for lin in (lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4):
            lin.model[-1].weight.uniform_(0.0, 2.0 / lin.model[-1].weight.shape[1])
    return lp


@pytest.mark.parametrize("shape", [(2, 64, 64), (3, 96, 128)])
def test_lpips_value_and_gradient_vs_oracle(shape):
    b, hh, ww = shape
    lp = _lpips()
    p = {k: v.detach().clone() for k, v in lp.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    img = torch.rand(b, 3, hh, ww, generator=g) * 2 - 1
    rec = (img + 0.3 * torch.randn(b, 3, hh, ww, generator=g)).clamp(-1, 1)
    lpd = lp.cuda()
    rd = rec.cuda().requires_grad_(True)
    val = lpd(img.cuda(), rd)
    val.backward()
    ro = rec.clone().requires_grad_(True)
    vo = R.lpips_forward(img, ro, p, q=R.bf16_round)
    vo.backward()
    assert abs(val.item() - vo.item()) < 5e-3 * abs(vo.item())
    rel = ((rd.grad.cpu().double() - ro.grad.double()).norm() / ro.grad.double().norm()).item()
    assert rel < 3e-2          # bf16 activations and activation gradients: noise floor of the rounding sites
    assert val.item() > 0 and lpd(img.cuda(), img.cuda()).item() == 0.0      # LPIPS(x, x) = 0 exactly
    # deterministic (no atomics anywhere on the path)
    rd2 = rec.cuda().requires_grad_(True)
    v2 = lpd(img.cuda(), rd2)
    v2.backward()
    assert torch.equal(v2, val) and torch.equal(rd2.grad, rd.grad)


def test_maxpool_and_relu_gate_bit_exact():
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 8, 12, 16, generator=g).relu().cuda().to(torch.bfloat16)
    x[0, :2, :2, 0] = 1.5                                                    # a tie inside one window
    y = ops.maxpool2x2(x)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(xr, 2, 2)
    assert torch.equal(y.float(), yr.permute(0, 2, 3, 1))
    dpool = torch.randn(2, 4, 6, 16, generator=g).cuda().to(torch.bfloat16)
    extra = torch.randn(2, 8, 12, 16, generator=g).cuda().to(torch.bfloat16)
    dx = ops.maxpool2x2_relu_bwd(dpool, x, extra)
    yr.backward(dpool.float().permute(0, 3, 1, 2))
    ref = ((xr.grad.permute(0, 2, 3, 1) + extra.float()).to(torch.bfloat16).float()) * (x.float() > 0)
    assert torch.equal(dx.float(), ref)
    dy = torch.randn(2, 8, 12, 16, generator=g).cuda().to(torch.bfloat16)
    assert torch.equal(ops.relu_bwd(dy, x).float(), dy.float() * (x.float() > 0))
