"""LightningDiT (teacher / student velocity model of train_dmd.py): the CPU oracle (oracle/ref_cpu.py::lightningdit_forward) and the drop-in
module's stock PyTorch route (dmvae_amd/models/lightningdit.py) against the fixtures captured from the reference's own
diffusion/lightningdit package (oracle/capture_golden_dit.py), plus the XL/1 state_dict manifest."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from oracle.detweights import det_fill_

# CPU tests of the host mirrors (parameter layout, transport / sampler logic): the model's stock-PyTorch route is what they run on, explicitly
pytestmark = pytest.mark.usefixtures("allow_stock")

CFGS = {"dit_small_hd64": dict(input_size=8, patch_size=1, in_channels=8, hidden_size=128, depth=2, num_heads=2, num_classes=10),
        "dit_small_hd72": dict(input_size=8, patch_size=1, in_channels=8, hidden_size=144, depth=2, num_heads=2, num_classes=10),
        "dit_small_p2": dict(input_size=8, patch_size=2, in_channels=4, hidden_size=128, depth=1, num_heads=2, num_classes=10),
        # head dim 64 at a width whose SwiGLU inner size (512) the HIP kernels take; hidden 128 gives 341 (CPU oracle / stock module only)
        "dit_small_hd64w": dict(input_size=8, patch_size=1, in_channels=8, hidden_size=192, depth=2, num_heads=3, num_classes=10),
        # config C1's velocity model (toy_example_2d/dmd.py:436-454): LightningDiT-Mini/1 at ONE token per sample, SwiGLU width 682
        "dit_toy_mini1": dict(input_size=1, patch_size=1, in_channels=2, hidden_size=256, depth=6, num_heads=4, num_classes=1)}


def build(tag, g):
    from dmvae_amd.models.lightningdit import LightningDiT
    m = LightningDiT(**CFGS[tag]).eval()
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    for k, v in g.sub("fix.").items():                       # the fixed sin-cos position table and the RoPE tables are reproduced exactly
        assert torch.allclose(m.state_dict()[k], v, rtol=0, atol=2e-6), k
    det_fill_(m, int(g["seed"]), skip=("pos_embed",))
    return m


@pytest.mark.parametrize("tag", list(CFGS))
def test_lightningdit_oracle_and_stock_module(tag):
    g = load_golden(tag)
    m = build(tag, g)
    x, t, y = g.t("x"), g.t("t"), torch.from_numpy(np.asarray(g["y"]))
    with torch.no_grad():
        yo = R.lightningdit_forward(x, t, y, dict(m.state_dict()), CFGS[tag]["num_heads"], CFGS[tag]["patch_size"])
    assert rel_err(yo, g.t("out")) < 2e-5
    xr = x.clone().requires_grad_(True)
    out = m.forward_stock(xr, t, y)
    assert rel_err(out.detach(), g.t("out")) < 2e-5
    out.backward(g.t("dy"))
    assert rel_err(xr.grad, g.t("dx")) < 1e-4
    grads = dict(m.named_parameters())
    for k, v in g.sub("g.").items():
        assert rel_err(grads[k].grad, v) < 2e-4, k
    for n, prm in m.named_parameters():
        if prm.grad is None:
            assert n == "pos_embed"
            continue
        gn = g["gn." + n][0]
        if gn > 1e-4:
            assert abs(prm.grad.double().norm().item() - gn) < 1e-3 * gn, n
    # forward() dispatches to the stock route on the CPU
    with torch.no_grad():
        assert torch.equal(m(x, t, y), m.forward_stock(x, t, y))


def test_lightningdit_xl1_manifest():
    from dmvae_amd.models.lightningdit import LightningDiT_models
    g = load_golden("dit_xl1_manifest")
    m = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000)      # train_dmd.py:353-357
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]] and len(sd) == 407
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"])
    for k in ("pos_embed", "feat_rope.freqs_cos", "feat_rope.freqs_sin"):
        ck = g["ck." + k]
        assert abs(sd[k].double().sum().item() - ck[0]) < 1e-4 * max(1.0, ck[1]) and abs(sd[k].double().abs().sum().item() - ck[1]) < 1e-5 * ck[1], k
    assert torch.allclose(sd["pos_embed"][0, ::37, ::97], g.t("pos_embed_slice"), atol=2e-6)
    assert torch.allclose(sd["feat_rope.freqs_cos"][::29, ::7], g.t("rope_cos_slice"), atol=2e-6)
    assert not m.pos_embed.requires_grad
    # reference initialisation (lightningdit.py:342-376): zeroed adaLN / output layers
    assert float(m.blocks[3].adaLN_modulation[1].weight.detach().abs().max()) == 0.0 and float(m.final_layer.linear.weight.detach().abs().max()) == 0.0
