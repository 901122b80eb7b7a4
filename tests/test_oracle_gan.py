"""The CPU oracle's discriminator branch (oracle/ref_cpu.py: patchgan_forward, diffaug, forward_discriminator,
forward_generator_gan) against the fixtures captured from the reference's own models/patchgan.py, utils/diffaug.py and
train_tokenizer.VAELossFunction (oracle/capture_golden_gan.py)."""
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from oracle.detweights import det_fill_patchgan_, det_tensor

TOL = 2e-5


def patchgan_params(g, seed, buf_prefix="buf0."):
    """state_dict of the reference discriminator rebuilt from names + seed (shapes from this build's own drop-in class)."""
    from dmvae_amd.models.patchgan import NLayerDiscriminator
    sd = {k: v.clone() for k, v in NLayerDiscriminator(use_syncbn=True).state_dict().items()}
    assert list(sd.keys()) == [str(k) for k in g["keys"]] if "keys" in g else True
    det_fill_patchgan_(sd, seed)
    for k, v in g.sub(buf_prefix).items():
        assert torch.equal(sd[k], v), k              # the captured running statistics are the deterministic fill
    return {k: v for k, v in sd.items() if not k.endswith("num_batches_tracked")}


def test_patchgan_forward_backward():
    g = load_golden("patchgan_small")
    p = patchgan_params(g, int(g["seed"]))
    assert sum(v.numel() for k, v in p.items() if "running" not in k) == int(g["n_params"]) == 2766529
    with torch.no_grad():
        y_eval, _ = R.patchgan_forward(g.t("x"), p, training=False)
    assert rel_err(y_eval, g.t("y_eval")) < TOL
    pr = {k: (v.clone().requires_grad_(True) if "running" not in k else v) for k, v in p.items()}
    x = g.t("x").requires_grad_(True)
    y, buf = R.patchgan_forward(x, pr, training=True)
    assert y.shape == (2, 1, 6, 6)
    assert rel_err(y, g.t("y")) < TOL
    y.backward(g.t("dy"))
    assert rel_err(x.grad, g.t("dx")) < 1e-4
    for k, v in buf.items():
        assert rel_err(v, g.t("buf1." + k)) < TOL, k
    for k, v in g.sub("g.").items():
        if v.abs().max() < 1e-5:       # a conv bias in front of a BatchNorm: removed by the mean subtraction, exactly zero gradient
            assert k == "main.2.bias" and pr[k].grad.abs().max() < 1e-5
            continue
        assert rel_err(pr[k].grad, v) < 1e-4, k
    for k in pr:
        if "running" in k:
            continue
        gn = g["gn." + k]
        if gn[0] > 1e-3:
            assert abs(pr[k].grad.double().norm().item() - gn[0]) < 1e-3 * gn[0], k


@pytest.mark.parametrize("tag", ["a", "b"])
def test_diffaug(tag):
    g = load_golden("diffaug")
    x = g.t(f"{tag}.x")
    for name, flags in (("trans", (True, False, False)), ("color", (False, True, False)), ("cut", (False, False, True)), ("all", (True, True, True))):
        y = R.diffaug(x, g.t(f"{tag}.{name}.rand01"), *flags)
        ref = g.t(f"{tag}.{name}.y")
        if name in ("trans", "cut"):
            assert torch.equal(y, ref), name              # index ops: bit-exact
        else:
            assert rel_err(y, ref) < 1e-6, name
    xr = x.clone().requires_grad_(True)
    R.diffaug(xr, g.t(f"{tag}.all.rand01")).backward(g.t(f"{tag}.all.dy"))
    assert rel_err(xr.grad, g.t(f"{tag}.all.dx")) < 1e-5


def test_diffaug_extreme_draws():
    g = load_golden("diffaug")
    y = R.diffaug(g.t("a.x"), g.t("a.edge.rand01"))
    assert rel_err(y, g.t("a.edge.y")) < 1e-6
    assert (y == 0).float().mean() > 0.1                 # shifted-out rows / columns and the cut-out are exact zeros
    assert torch.equal(y == 0, g.t("a.edge.y") == 0)


def test_gan_losses():
    g = load_golden("gan_losses")
    from test_oracle_golden import lpips_params
    assert int(g["vgg_seed"]) == 41                      # lpips_params regenerates the VGG trunk from this seed
    lpp = lpips_params(g)
    disc_p = patchgan_params(g, int(g["disc_seed"]))
    img, feat = g.t("images"), g.t("feat")
    last = g.t("last").requires_grad_(True)
    recon = torch.nn.functional.conv2d(feat, last, padding=1) + 0.9 * img
    assert rel_err(recon.detach(), g.t("recon")) < 1e-6
    total, log = R.forward_generator_gan(img, recon, lpp, disc_p, last, g.t("gen_rand01"), disc_weight=0.5)
    assert abs(log["rec_loss"].item() - float(g["gen_rec_loss"])) < 1e-4 * abs(float(g["gen_rec_loss"]))
    assert abs(log["d_weight"].item() - float(g["d_weight"])) < 1e-3 * float(g["d_weight"])
    assert abs(total.item() - float(g["gen_loss"])) < 1e-4 * abs(float(g["gen_loss"]))
    total.backward()
    assert rel_err(last.grad, g.t("g_last")) < 1e-3
    # discriminator step (bcr weight 4, strong cut-out 0.5 as captured)
    dp = {k: (v.clone().requires_grad_(True) if "running" not in k else v) for k, v in disc_p.items()}
    d_total, dlog, buf = R.forward_discriminator(img, g.t("recon"), dp, g.t("d_rand01_a"), g.t("d_rand01_b"), bcr_weight=4.0, cutout_a=0.2, cutout_b=0.5)
    assert abs(d_total.item() - float(g["d_total"])) < 1e-4 * abs(float(g["d_total"]))
    for k in ("d_loss", "bcr_loss", "acc_real", "acc_fake", "acc_mean"):
        assert abs(float(dlog[k]) - float(g["dlog." + k])) < 1e-4 * max(1.0, abs(float(g["dlog." + k]))), k
    for k, v in buf.items():
        assert rel_err(v, g.t("buf2." + k)) < TOL, k
    d_total.backward()
    for k, v in g.sub("dg.").items():
        assert rel_err(dp[k].grad, v) < 1e-3, k
    for k in dp:
        if "running" in k:
            continue
        gn = g["dgn." + k]
        if gn[0] > 1e-3:
            assert abs(dp[k].grad.double().norm().item() - gn[0]) < 2e-3 * gn[0], k
