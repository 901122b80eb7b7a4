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


@pytest.mark.parametrize("b,hh,ww,c", [(2, 16, 16, 64), (1, 8, 24, 128), (3, 4, 6, 256), (2, 2, 2, 512), (1, 6, 10, 72)])
def test_diff_and_pool_in_one_pass_equals_the_two_launches(b, hh, ww, c):
    """ops.lpips_diff_pool (losses.hip::lpips_diff_pool_kernel): the feature gradient and the pooled features bit for bit, the level value up to the order of its
    f32 partial sums; with and without the gradient; accumulating; reruns identical."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(7 * b + hh + c)
    h = torch.relu(torch.randn(2 * b, hh, ww, c, generator=g)).cuda().to(torch.bfloat16)
    lin = torch.rand(c, generator=g).cuda()
    for need in (True, False):
        out0 = torch.full((1,), 0.25, device="cuda")
        out1 = out0.clone()
        df0 = ops.lpips_diff(h[:b], h[b:], lin, out0, 1.0 / (hh * ww * b), need, accumulate=True)
        pool0 = ops.maxpool2x2(h)
        df1, pool1 = ops.lpips_diff_pool(h, b, lin, out1, 1.0 / (hh * ww * b), need, accumulate=True)
        assert torch.equal(pool1, pool0)
        assert (df1 is None) == (not need) and (df0 is None or torch.equal(df1, df0))
        assert abs(out1.item() - out0.item()) <= 2e-6 * abs(out0.item())
        out2 = torch.full((1,), 0.25, device="cuda")
        df2, pool2 = ops.lpips_diff_pool(h, b, lin, out2, 1.0 / (hh * ww * b), need, accumulate=True)
        assert torch.equal(out2, out1) and torch.equal(pool2, pool1) and (df1 is None or torch.equal(df2, df1))
    same = torch.cat([h[:b], h[:b]])
    out = torch.zeros(1, device="cuda")
    ops.lpips_diff_pool(same, b, lin, out, 1.0 / (hh * ww * b), False, accumulate=False)
    assert out.item() == 0.0                                   # LPIPS(x, x) is exactly zero


def test_lpips_module_with_the_fused_pool_on_and_off(monkeypatch):
    from dmvae_amd.utils import lpips as L
    lp = _lpips().cuda()
    g = torch.Generator().manual_seed(3)
    img = (torch.rand(2, 3, 64, 96, generator=g) * 2 - 1).cuda()
    rec0 = (img + 0.3 * torch.randn(2, 3, 64, 96, generator=g).cuda()).clamp(-1, 1)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(L, "DIFF_POOL_FUSED", on)
        rec = rec0.clone().requires_grad_(True)
        v = lp(img, rec)
        v.backward()
        res[on] = (v.detach(), rec.grad.clone())
    assert abs(res[True][0].item() - res[False][0].item()) <= 1e-5 * abs(res[False][0].item())
    assert torch.equal(res[True][1], res[False][1])            # the pooled features and the feature gradients are the same bits: so is everything downstream
