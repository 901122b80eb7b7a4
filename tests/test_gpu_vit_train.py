"""Trainable ViT encoder path on the HIP kernels (-m gpu): the backward-side kernels of csrc/vit_bwd.hip against fp64 autograd on the same
operands, and the block / encoder autograd Functions against the stock PyTorch module under autocast(bf16) and the CPU oracle."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.mark.parametrize("rows,c", [(37, 1024), (8224, 1024), (5, 256), (300, 1536)])
def test_layernorm_bwd(rows, c):
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(rows + c)
    x = (torch.randn(rows, c, generator=g) * 2 + 0.3).to(DEV)
    gam, bet = (1 + 0.3 * torch.randn(c, generator=g)).to(DEV), torch.randn(c, generator=g).to(DEV)
    dy = torch.randn(rows, c, generator=g).to(DEV).to(BF)
    dres = torch.randn(rows, c, generator=g).to(DEV)
    xr = x.double().requires_grad_(True)
    gr, br = gam.double().requires_grad_(True), bet.double().requires_grad_(True)
    F.layer_norm(xr, (c,), gr, br, 1e-6).backward(dy.double())
    dx = dres.clone()
    dg, db = ops.layernorm_bwd_(dx, dy, x, gam, 1e-6)
    assert rel_err(dx, dres.double() + xr.grad) < 2e-6
    assert rel_err(dg, gr.grad) < 1e-5 and rel_err(db, br.grad) < 1e-5
    dx2 = dres.clone()
    dg2, db2 = ops.layernorm_bwd_(dx2, dy, x, gam, 1e-6)
    assert torch.equal(dx, dx2) and torch.equal(dg, dg2) and torch.equal(db, db2)        # fixed-order reductions
    acc_g, acc_b = dg.clone(), db.clone()
    ops.layernorm_bwd_(dres.clone(), dy, x, gam, 1e-6, dg_out=acc_g, db_out=acc_b, accumulate=True)
    assert rel_err(acc_g, 2 * gr.grad) < 1e-5 and rel_err(acc_b, 2 * br.grad) < 1e-5


@pytest.mark.parametrize("rows,c", [(37, 1024), (8224, 1024), (9, 256), (100, 512)])
def test_layerscale_bwd_and_gelu(rows, c):
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(rows)
    dt = torch.randn(rows, c, generator=g).to(DEV)
    y = torch.randn(rows, c, generator=g).to(DEV).to(BF)
    gam = (0.5 + 0.2 * torch.randn(c, generator=g)).to(DEV)
    dy, dg = ops.layerscale_bwd(dt, y, gam)
    assert torch.equal(dy, (gam * dt).to(BF))
    assert rel_err(dg, (dt.double() * y.double()).sum(0)) < 1e-5
    x = (torch.randn(rows, c, generator=g) * 2).to(DEV).to(BF)
    d = torch.randn(rows, c, generator=g).to(DEV).to(BF)
    xr = x.double().requires_grad_(True)
    F.gelu(xr).backward(d.double())
    assert rel_err(ops.gelu(x).float(), F.gelu(xr.detach())) < 4e-3                       # bf16 result
    assert (ops.gelu(x).float() - F.gelu(x.float()).to(BF).float()).abs().max() <= 2 ** -7 * F.gelu(x.float()).abs().max()
    assert rel_err(ops.gelu_bwd(d, x).float(), xr.grad) < 4e-3
