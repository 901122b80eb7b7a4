"""CPU: the oracle's ViT restatement (oracle/ref_cpu.py::vit_forward_features) against outputs and gradients captured from the reference's own
models/dinov2.py at the widths the HIP encoder kernels accept (oracle/capture_golden_vit.py -> tests/golden/vit_w256.npz, vit_w768.npz)."""
import warnings

import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from oracle.detweights import det_tensor


def vit_fixture(name):
    """(golden, our DinoV2ViT filled with the capture's name-seeded weights, its parameter dict, the capture's input image)."""
    from dmvae_amd.models.vit import DinoV2ViT
    g = load_golden(name)
    cfg = dict(vit_w256=dict(embed_dim=256, depth=2, num_heads=4), vit_w768=dict(embed_dim=768, depth=1, num_heads=12))[name]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vit = DinoV2ViT(patch_size=16, img_size=256, **cfg)
    p = {k: det_tensor(k, v.shape, int(g["seed"])) for k, v in vit.state_dict().items()}
    vit.load_state_dict(p, strict=True)
    b = g["out"].shape[0]
    x = torch.rand(b, 3, 256, 256, generator=torch.Generator().manual_seed(int(g["x_seed"]))) * 2 - 1
    return g, vit, p, x


def test_vit_w256_forward_and_gradients_vs_reference_capture():
    g, vit, p, x = vit_fixture("vit_w256")
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xo = x.clone().requires_grad_(True)
    out = R.vit_forward_features(xo, po, pre="", num_heads=int(g["heads"]))
    assert rel_err(out.detach(), g.t("out")) < 2e-5
    dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(int(g["dy_seed"])))
    (out * dy).sum().backward()
    assert rel_err(xo.grad[:, :, ::16, ::16], g.t("dx_slice")) < 1e-4
    for n, gn in zip(g["names"], g["gnorm"]):
        assert abs(po[str(n)].grad.double().norm().item() - float(gn)) < 1e-4 * float(gn), n
    for k in [k for k in g if k.startswith("g.")]:
        assert rel_err(po[k[2:]].grad, g.t(k)) < 1e-4, k
    # the stock module (parameter-name definition of the mirror) computes the same function
    with torch.no_grad():
        assert rel_err(vit.forward_features_stock(x), g.t("out")) < 2e-5


def test_vit_w768_forward_vs_reference_capture():
    g, vit, p, x = vit_fixture("vit_w768")
    with torch.no_grad():
        out = R.vit_forward_features(x, p, pre="", num_heads=int(g["heads"]))
    assert rel_err(out, g.t("out")) < 2e-5


def test_implicit_stock_route_needs_opt_in(monkeypatch):
    """A CPU tensor (or any call the HIP route does not cover) must not drop silently onto the stock modules (dmvae_amd/_stock.py)."""
    import pytest
    from dmvae_amd._lib import DmvaeHipError
    from dmvae_amd.models.lightningdit import LightningDiT
    from dmvae_amd.models.vit import DinoV2ViT
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vit = DinoV2ViT(embed_dim=64, depth=1, num_heads=2, patch_size=16, img_size=32)
    dit = LightningDiT(input_size=4, patch_size=1, in_channels=4, hidden_size=64, depth=1, num_heads=2, num_classes=10).eval()
    x = torch.zeros(1, 3, 32, 32)
    z, t, y = torch.zeros(1, 4, 4, 4), torch.zeros(1), torch.zeros(1, dtype=torch.long)
    monkeypatch.delenv("DMVAE_ALLOW_STOCK", raising=False)
    with pytest.raises(DmvaeHipError, match="DMVAE_ALLOW_STOCK"):
        vit.forward_features(x)
    with pytest.raises(DmvaeHipError, match="DMVAE_ALLOW_STOCK"):
        dit(z, t, y)
    monkeypatch.setenv("DMVAE_ALLOW_STOCK", "1")
    with pytest.warns(UserWarning, match="STOCK"):
        out = vit.forward_features(x)
    assert torch.equal(out, vit.forward_features_stock(x))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert torch.equal(dit(z, t, y), dit.forward_stock(z, t, y))
