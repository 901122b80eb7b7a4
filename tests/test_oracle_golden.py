"""Pins oracle/ref_cpu.py (the CPU restatement) to golden vectors captured from the reference's own modules
(oracle/capture_golden.py).  CPU only.  Tolerance 1e-5 relative-to-max (fp32 vs fp32, different op order)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from oracle.detweights import det_tensor

TOL = 1e-5


def _grads(out, dy, inputs):
    return torch.autograd.grad(out, inputs, dy)


@pytest.mark.parametrize("name", ["resblock_same", "resblock_short"])
def test_resblock(name):
    g = load_golden(name)
    p = {k: v.requires_grad_(True) for k, v in g.sub("p.").items()}
    x = g.t("x").requires_grad_(True)
    y = R.resnet_block(x, p, "")
    assert rel_err(y, g.t("y")) < TOL
    keys = sorted(p)
    gr = _grads(y, g.t("dy"), [x] + [p[k] for k in keys])
    assert rel_err(gr[0], g.t("dx")) < TOL
    for k, v in zip(keys, gr[1:]):
        assert rel_err(v, g.t("g." + k)) < 5e-5, k


def test_attnblock():
    g = load_golden("attnblock")
    p = {k: v.requires_grad_(True) for k, v in g.sub("p.").items()}
    x = g.t("x").requires_grad_(True)
    y = R.attn_block(x, p, "")
    assert rel_err(y, g.t("y")) < TOL
    keys = sorted(p)
    gr = _grads(y, g.t("dy"), [x] + [p[k] for k in keys])
    assert rel_err(gr[0], g.t("dx")) < TOL
    for k, v in zip(keys, gr[1:]):
        if k == "k.bias":   # softmax is invariant to a per-query constant: d/d k.bias is exactly 0 up to fp noise
            assert v.abs().max() < 1e-5 and g.t("g." + k).abs().max() < 1e-5
            continue
        assert rel_err(v, g.t("g." + k)) < 5e-5, k


@pytest.mark.parametrize("name,fn", [("upsample", R.upsample), ("downsample", R.downsample)])
def test_updown(name, fn):
    g = load_golden(name)
    p = {k: v.requires_grad_(True) for k, v in g.sub("p.").items()}
    x = g.t("x").requires_grad_(True)
    y = fn(x, p, "")
    assert rel_err(y, g.t("y")) < TOL
    gr = _grads(y, g.t("dy"), [x, p["conv.weight"], p["conv.bias"]])
    assert rel_err(gr[0], g.t("dx")) < TOL
    assert rel_err(gr[1], g.t("g.conv.weight")) < TOL and rel_err(gr[2], g.t("g.conv.bias")) < TOL


def test_upsample_is_index_replication_bit_exact():
    x = torch.randn(2, 3, 5, 7)
    up = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    assert torch.equal(up, torch.nn.functional.interpolate(x, scale_factor=2.0, mode="nearest"))


def decoder_params(shapes, seed):
    return {k: det_tensor(k, s, seed) for k, s in shapes.items()}


def _decoder_shapes(ch, z):
    from dmvae_amd.models import flux_ae
    dec = flux_ae.Decoder(ch=ch, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, resolution=256, z_channels=16)
    dec.post_init(z_channels=z)
    return {k: tuple(v.shape) for k, v in dec.state_dict().items()}


def test_decoder_small():
    g = load_golden("decoder_small")
    p = {k: v.requires_grad_(True) for k, v in decoder_params(_decoder_shapes(32, 32), 12).items()}
    z = g.t("z").requires_grad_(True)
    y = R.decoder_forward(z, p)
    assert rel_err(y, g.t("y")) < TOL
    keys = sorted(p)
    gr = _grads(y, g.t("dy"), [z] + [p[k] for k in keys])
    assert rel_err(gr[0], g.t("dz")) < 5e-5
    for k, v in zip(keys, gr[1:]):
        gn = g["gn." + k]
        # some gradients are exactly zero in exact arithmetic (attn k.bias: softmax shift invariance; a conv bias
        # feeding a GroupNorm with one channel per group): both sides are fp noise there, hence the absolute floor
        assert abs(v.double().norm().item() - gn[0]) <= 5e-5 * gn[0] + 2e-5, k
        if "g." + k in g:
            assert rel_err(v, g.t("g." + k)) < 5e-5, k


def test_decoder_full_b1():
    g = load_golden("decoder_full_b1")
    p = decoder_params(_decoder_shapes(128, 32), 22)
    with torch.no_grad():
        y = R.decoder_forward(g.t("z"), p)
    assert y.shape == (1, 3, 256, 256)
    assert rel_err(y[0, :, ::8, ::8], g.t("y_slice")) < TOL
    assert abs(y.double().abs().sum().item() - g["y_sum"][1]) < 1e-5 * g["y_sum"][1]


def test_decoder_full_b1_backward():
    """The oracle at FULL width, forward and backward, against the reference's capture (oracle/capture_golden_bwd.py): d z, every parameter-gradient norm, six
    gradient slices.  dy is regenerated from its seed."""
    from oracle.capture_golden_bwd import SLICES
    g = load_golden("decoder_full_b1_bwd")
    p = decoder_params(_decoder_shapes(128, 32), 22)
    keys = list(p.keys())
    z = load_golden("decoder_full_b1").t("z").requires_grad_(True)
    ps = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    y = R.decoder_forward(z, ps)
    dy = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(23))
    gr = torch.autograd.grad(y, [z] + [ps[k] for k in keys], dy)
    assert abs(y.double().abs().sum().item() - g["y_sum"][1]) < 1e-5 * g["y_sum"][1]
    assert rel_err(gr[0], g.t("dz")) < 5e-5
    for k, v in zip(keys, gr[1:]):
        gn = g["gn." + k]
        assert abs(v.double().norm().item() - gn[0]) <= 5e-5 * gn[0] + 2e-5, k
        if "g." + k in g:
            assert rel_err(v[SLICES[k]], g.t("g." + k)) < 5e-5, k


def test_flux_encoder_small():
    g = load_golden("flux_encoder_small")
    with torch.no_grad():
        y = R.encoder_forward(g.t("x"), g.sub("p."), num_resolutions=2, num_res_blocks=1)
    assert rel_err(y, g.t("y")) < TOL


def test_mlp():
    g = load_golden("mlp")
    p = {("bottle_neck." + k): v.requires_grad_(True) for k, v in g.sub("p.").items()}
    x = g.t("x").requires_grad_(True)
    y = R.mlp_forward(x, p)
    assert rel_err(y, g.t("y")) < TOL
    keys = sorted(p)
    gr = _grads(y, g.t("dy"), [x] + [p[k] for k in keys])
    assert rel_err(gr[0], g.t("dx")) < TOL
    for k, v in zip(keys, gr[1:]):
        assert rel_err(v, g.t("g." + k[len("bottle_neck."):])) < TOL


def vae_tiny_params(seed=33, width=64):
    """Parameters of the tiny-ViT VAE fixture, regenerated from the capture's names (reference adapter inserts 'vit.')."""
    from dmvae_amd.models.vae import VAE
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=width, depth=2, num_heads=4))
    p = {}
    params = dict(vae.named_parameters())
    for k, v in vae.state_dict().items():
        if k in params:
            ref_name = k.replace("encoder.model.", "encoder.model.vit.", 1) if k.startswith("encoder.model.") else k
            p[k] = det_tensor(ref_name, v.shape, seed)
        else:
            p[k] = v.clone()
    return p, vae


def test_vae_forward_tiny():
    g = load_golden("vae_forward_tiny")
    p, _ = vae_tiny_params()
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(32)) * 2 - 1
    with torch.no_grad():
        rec, lat = R.vae_forward(x, p, num_heads=4, return_latent=True)
    assert rel_err(lat, g.t("latent")) < TOL
    assert rel_err(rec[0, :, ::8, ::8], g.t("rec_slice")) < 2e-5
    assert float(g["encode_equal"]) == 0.0 and float(g["decode_equal"]) == 0.0


def test_vae_large_key_manifest():
    """state_dict keys of the build's VAE(large, z=32) == the reference's (ViT stand-in adapter prefix removed)."""
    g = load_golden("vae_large_manifest")
    ref_keys = [k.replace("encoder.model.vit.", "encoder.model.") for k in g["keys"]]
    ref_shapes = dict(zip(ref_keys, g["shapes"]))
    ref_keys = [k for k in ref_keys if not k.endswith("mask_token")]  # models/dinov2.py-only extra
    from dmvae_amd.models.vae import VAE
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.device("meta"):
            vae = VAE(z_channels=32, model_size="large")
    sd = vae.state_dict()
    assert sorted(sd.keys()) == sorted(ref_keys)
    for k, v in sd.items():
        assert str(tuple(v.shape)) == ref_shapes[k], k


def lpips_params(g, prefix="p."):
    p = dict(g.sub(prefix))
    chans = [3] + [c for c in R.VGG_CFG if c != "M"]
    idx, ci, bounds = 0, 0, (4, 9, 16, 23, 30)
    for v in R.VGG_CFG:
        if v == "M":
            idx += 1
            continue
        sl = 1 + sum(idx >= b for b in bounds)
        name = f"net.slice{sl}.{idx}"
        p[name + ".weight"] = det_tensor(name + ".weight", (v, chans[ci], 3, 3), 41) * (2.0 ** 0.5)
        p[name + ".bias"] = det_tensor(name + ".bias", (v,), 41) * 0.5
        ci += 1
        idx += 2
    return p


def test_gen_loss():
    g = load_golden("gen_loss")
    p = lpips_params(g)
    for k in [k for k in g if k.startswith("ck.")]:
        v = p[k[3:]].double()
        assert abs(v.sum().item() - g[k][0]) <= 1e-6 * max(1.0, abs(g[k][1])), k
    rec = g.t("recon").requires_grad_(True)
    loss, log = R.forward_generator(g.t("images"), rec, p)
    assert abs(loss.item() - float(g["rec_loss"])) < 2e-5 * abs(float(g["rec_loss"]))
    assert abs(log["L1"].item() - float(g["L1"])) < 1e-6 and abs(log["L2"].item() - float(g["L2"])) < 1e-6
    assert abs(log["LPIPS"].item() - float(g["LPIPS"])) < 2e-5 * float(g["LPIPS"])
    (d,) = torch.autograd.grad(loss, rec)
    assert rel_err(d, g.t("d_recon")) < 1e-4


def test_lpips_diff():
    g = load_golden("lpips_diff")
    f0 = [g.t(f"f0_{k}") for k in range(5)]
    f1 = [g.t(f"f1_{k}").requires_grad_(True) for k in range(5)]
    val = R.lpips_from_feats(f0, f1, [g.t(f"w_{k}") for k in range(5)])
    assert abs(val.item() - float(g["value"])) < 1e-5 * float(g["value"])
    gr = torch.autograd.grad(val, f1)
    for k in range(5):
        assert rel_err(gr[k], g.t(f"df1_{k}")) < TOL


@pytest.mark.parametrize("name", ["dmd_loss_cfg5", "dmd_loss_cfg1"])
def test_dmd_loss(name):
    g = load_golden(name)
    lat = g.t("latents").requires_grad_(True)
    t = g.t("t_raw") * (float(g["t1"]) - float(g["t0"])) + float(g["t0"])
    loss, gnorm, grad = R.dmd_loss(lat, t, g.t("x0"), g.t("v_teacher"), g.t("v_student"), g.t("v_teacher_u"), g.t("v_student_u"),
                                   cfg=float(g["cfg"]))
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert abs(gnorm.item() - float(g["dmd_gradient_norm"])) < 1e-5 * float(g["dmd_gradient_norm"])
    (d,) = torch.autograd.grad(loss, lat)
    assert rel_err(d, g.t("dlatents")) < TOL
    assert rel_err(grad / grad.numel(), g.t("dlatents")) < TOL  # dL/dlatents == grad/numel


def test_dmd_loss_toy():
    g = load_golden("dmd_loss_toy")
    pts = g.t("points").requires_grad_(True)
    loss, _, grad = R.dmd_loss(pts[:, :, None, None], g.t("t_raw"), g.t("x0"), g.t("v_teacher"), g.t("v_student"), weight_factor=False)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    (d,) = torch.autograd.grad(loss, pts)
    assert rel_err(d, g.t("dpoints")) < TOL


def test_transport():
    g = load_golden("transport")
    xt, ut = R.transport_plan(g.t("t"), g.t("x0"), g.t("x1"))
    assert torch.equal(xt, g.t("xt")) and torch.equal(ut, g.t("ut"))
    assert rel_err(R.transport_loss(g.t("model_out"), g.t("t"), g.t("x0"), g.t("x1")), g.t("loss")) < 1e-6


def test_latents_to_spatial_bit_exact():
    g = load_golden("latents_to_spatial")
    assert torch.equal(R.latents_to_spatial(g.t("tokens")), g.t("spatial"))


def test_opt_tail():
    g = load_golden("opt_tail")
    names = ["0.weight", "0.bias", "2.weight", "2.bias"]
    p = [g.t("p0." + n) for n in names]
    ema = [x.clone() for x in p]
    m = [torch.zeros_like(x) for x in p]
    v = [torch.zeros_like(x) for x in p]
    warm = int(g["warmup_steps"])
    for it in range(6):
        grads = [g.t(f"g{it}.{i}") for i in range(4)]
        norm, grads = R.clip_grad_norm(grads, 1.0)
        assert abs(norm.item() - float(g["norms"][it])) < 1e-5 * float(g["norms"][it])
        lr = R.warmup_lr(it, 1e-4, warm)
        assert lr == pytest.approx(float(g["lrs"][it]), rel=1e-12, abs=0)      # the reference's own LambdaLR sequence (lr 0 at step 0)
        for i in range(4):
            p[i], m[i], v[i] = R.adamw_step(p[i], grads[i], m[i], v[i], it + 1, lr)
            ema[i] = R.ema_update(ema[i], p[i])
    for i, n in enumerate(names):
        assert rel_err(p[i], g.t("p6." + n)) < 1e-6 and rel_err(ema[i], g.t("ema6." + n)) < 1e-6


def test_sshape():
    g = load_golden("sshape")
    assert np.array_equal(R.sshape_sample(1536, 42), g["samples"])


# ---- build-defined KL / MMD: no reference => closed-form properties only ("parity unpinned") -------------
def test_kl_mmd_closed_forms():
    gen = torch.Generator().manual_seed(0)
    z = torch.randn(64, 256, 32, generator=gen)
    kl, klm = R.kl_moment(z)
    assert kl.shape == (32,) and klm.item() < 2e-3            # N(0,1) samples -> KL ~ 0
    kl2, _ = R.kl_moment(z * 2 + 1)
    expect = 0.5 * (1 + 4 - 1 - np.log(4.0))
    assert abs(kl2.mean().item() - expect) < 0.05
    x = torch.randn(3, 64, 32, generator=gen)
    y = torch.randn(3, 48, 32, generator=gen)
    assert torch.allclose(R.mmd_rbf(x, x), torch.zeros(3), atol=1e-6)
    assert torch.allclose(R.mmd_rbf(x, y), R.mmd_rbf(y, x), atol=1e-6)
    a, b = R.mmd_rbf(x, y + 0.5), R.mmd_rbf(x, y + 1.5)
    assert (b > a).all() and (a > 0).all()


def test_subpixel_identity():
    """conv3x3(nearest_x2(x), W) == conv_transpose2d(x, WD, stride 2, padding 1) with WD = R.subpixel_weight(W) (the form the HIP path evaluates, flux_ae.py:103-107):
    output, input gradient (= the 4x4 stride-2 conv of dy with WD) and weight gradient (through the linear map W -> WD) in fp64, and the golden of the layer."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 5, 6, 7, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(4, 5, 3, 3, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, generator=g, dtype=torch.float64)
    dy = torch.randn(2, 4, 12, 14, generator=g, dtype=torch.float64)
    y0 = F.conv2d(x.repeat_interleave(2, 2).repeat_interleave(2, 3), w, b, padding=1)
    gx0, gw0 = torch.autograd.grad(y0, (x, w), dy)
    wd = R.subpixel_weight(w)
    y1 = F.conv_transpose2d(x, wd, b, stride=2, padding=1)
    gx1, gw1 = torch.autograd.grad(y1, (x, w), dy)
    assert rel_err(y1, y0) < 1e-13 and rel_err(gx1, gx0) < 1e-13 and rel_err(gw1, gw0) < 1e-13
    assert rel_err(F.conv2d(dy, wd.detach(), stride=2, padding=1), gx0) < 1e-13
    gold = load_golden("upsample")
    p = gold.sub("p.")
    yq = R.upsample(gold.t("x"), p, "", q=R.bf16_round)                    # sub-pixel form with bf16 rounding sites
    assert rel_err(R.upsample(gold.t("x"), p, ""), gold.t("y")) < 1e-5 and rel_err(yq, gold.t("y")) < 2e-2
