"""Generate tests/golden/*.npz by importing the reference's own PyTorch modules (read-only at
/root/reference) in THIS container and recording inputs -> outputs.  The reference cannot travel to
the GPU box, so only these data files do; no reference source is copied.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden.py            (CPU, a few minutes)

Missing third-party packages (timm, torchvision, tap, wandb, torchdiffeq, fairscale, torchmetrics)
are satisfied by import stubs that carry no hot-path arithmetic, with two documented stand-ins whose
real counterparts are third-party and unavailable offline:
  * timm's DINOv2 ViT  -> the reference's own models/dinov2.py (same block algebra / key names)
  * torchvision VGG16  -> the canonical 'D' layer list (Conv3x3+ReLU / MaxPool), random weights
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.detweights import det_fill_  # noqa: E402

REF = os.environ.get("DMVAE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_stubs():
    os.environ.setdefault("TORCHDYNAMO_DISABLE", "1")
    pkg = types.ModuleType("models")
    pkg.__path__ = [REF + "/models"]
    sys.modules["models"] = pkg
    _load("models.flux_ae", REF + "/models/flux_ae.py")
    _load("models.init_param", REF + "/models/init_param.py")

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
            super().__init__()
            self.patch_size = (patch_size, patch_size)
            self.num_patches = (img_size // patch_size) ** 2
            self.proj = nn.Conv2d(in_chans, embed_dim, patch_size, patch_size, bias=bias)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, act_layer=nn.GELU, drop=0.0):
            super().__init__()
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, in_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    class TimmLike(nn.Module):
        """Adapter exposing timm's interface over the reference's models/dinov2.py ViT."""
        num_prefix_tokens = 1

        def __init__(self, vit):
            super().__init__()
            self.vit = vit

        def forward_features(self, x):
            d = self.vit.forward_features(x)
            return torch.cat([d["x_norm_clstoken"][:, None], d["x_norm_patchtokens"]], 1)

    def create_model(name, pretrained=True, patch_size=16, img_size=256, **kw):
        dinov2 = importlib.import_module("models.dinov2")
        arch = os.environ.get("DMVAE_GOLDEN_VIT", "vit_large" if "large" in name else "vit_base")
        kwargs = dict(patch_size=patch_size, img_size=img_size, init_values=1e-5, block_chunks=0)
        if arch == "vit_tiny":  # reduced stand-in for small fixtures
            return TimmLike(dinov2.DinoVisionTransformer(embed_dim=64, depth=2, num_heads=4, mlp_ratio=4, **kwargs))
        if arch == "vit_w256":  # reduced stand-in inside the bf16 encoder kernels' range (heads of 64 channels): capture_golden_step.py --width 256
            return TimmLike(dinov2.DinoVisionTransformer(embed_dim=256, depth=2, num_heads=4, mlp_ratio=4, **kwargs))
        return TimmLike(getattr(dinov2, arch)(**kwargs))

    timm, tm, vt = (types.ModuleType(n) for n in ("timm", "timm.models", "timm.models.vision_transformer"))
    vt.PatchEmbed, vt.Mlp, tm.create_model, tm.vision_transformer, timm.models = PatchEmbed, Mlp, create_model, vt, tm
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.vision_transformer": vt})
    for n in ("fairscale", "fairscale.nn", "fairscale.nn.model_parallel", "fairscale.nn.model_parallel.initialize",
              "fairscale.nn.model_parallel.layers"):
        sys.modules[n] = types.ModuleType(n)
    lay = sys.modules["fairscale.nn.model_parallel.layers"]
    lay.ColumnParallelLinear = lay.ParallelEmbedding = lay.RowParallelLinear = object
    sys.modules["fairscale.nn.model_parallel"].initialize = sys.modules["fairscale.nn.model_parallel.initialize"]
    td = types.ModuleType("torchdiffeq")
    td.odeint = None
    sys.modules["torchdiffeq"] = td

    def vgg16(pretrained=True, **k):
        cfg, layers, c = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"], [], 3
        for v in cfg:
            if v == "M":
                layers.append(nn.MaxPool2d(2, 2))
            else:
                layers += [nn.Conv2d(c, v, 3, padding=1), nn.ReLU(inplace=True)]
                c = v
        m = nn.Module()
        m.features = nn.Sequential(*layers)
        return m

    for n in ("wandb", "torchvision", "torchvision.transforms", "torchvision.datasets", "torchvision.datasets.folder",
              "torchmetrics", "torchmetrics.image", "torchmetrics.image.fid", "torchmetrics.image.lpip"):
        try:
            importlib.import_module(n)
        except ImportError:
            sys.modules[n] = MagicMock(name=n)
    tvm = types.ModuleType("torchvision.models")
    tvm.vgg16 = vgg16
    sys.modules["torchvision.models"] = tvm
    sys.modules["torchvision"].models = tvm
    sys.modules["torchmetrics.image.fid"].FrechetInceptionDistance = type("FrechetInceptionDistance", (nn.Module,), {})
    tap = types.ModuleType("tap")
    tap.Tap = type("Tap", (), {"__init__": lambda s, *a, **k: None})
    sys.modules["tap"] = tap
    vae_mod = _load("models.vae", REF + "/models/vae.py")
    _load("models.patchgan", REF + "/models/patchgan.py")
    pkg.VAE, pkg.NLayerDiscriminator, pkg.DinoDisc = vae_mod.VAE, sys.modules["models.patchgan"].NLayerDiscriminator, None
    if REF not in sys.path:
        sys.path.insert(0, REF)


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


def checksum(module):
    """name -> (sum, sum|.|) in float64: pins fixed-seed initialisation without storing the weights."""
    return {k: np.array([v.double().sum().item(), v.double().abs().sum().item()]) for k, v in module.state_dict().items()}


def save(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def randomize(module, seed, std=0.05):
    """Give every parameter a non-degenerate value (reference init zeroes biases / ones norms)."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p_ in module.named_parameters():
            if p_.dim() == 1 and ("norm" in n and n.endswith("weight")):
                p_.copy_(1.0 + 0.2 * torch.randn(p_.shape, generator=g))
            elif p_.dim() == 1:
                p_.copy_(0.1 * torch.randn(p_.shape, generator=g))
            else:
                p_.copy_(std * torch.randn(p_.shape, generator=g))


def main():
    install_stubs()
    torch.set_grad_enabled(True)
    fa = sys.modules["models.flux_ae"]
    g = torch.Generator().manual_seed(1234)
    rn = lambda *s: torch.randn(*s, generator=g)

    # ---- G1 resblock (with / without shortcut): out, dx, dW* -------------------------------
    for tag, cin, cout, hw in (("resblock_same", 64, 64, 8), ("resblock_short", 128, 64, 8)):
        m = fa.ResnetBlock(cin, cout)
        randomize(m, 7)
        x = rn(2, cin, hw, hw).requires_grad_(True)
        y = m(x)
        dy = rn(*y.shape)
        y.backward(dy)
        save(tag, x=x, y=y, dy=dy, dx=x.grad, **{"p." + k: v for k, v in sd_np(m).items()},
             **{"g." + n: p_.grad for n, p_ in m.named_parameters()})

    # ---- G2 attnblock ----------------------------------------------------------------------
    m = fa.AttnBlock(64)
    randomize(m, 8, std=0.1)
    x = rn(2, 64, 8, 8).requires_grad_(True)
    y = m(x)
    dy = rn(*y.shape)
    y.backward(dy)
    save("attnblock", x=x, y=y, dy=dy, dx=x.grad, **{"p." + k: v for k, v in sd_np(m).items()},
         **{"g." + n: p_.grad for n, p_ in m.named_parameters()})

    # ---- G3 up / down sample ---------------------------------------------------------------
    for tag, cls in (("upsample", fa.Upsample), ("downsample", fa.Downsample)):
        m = cls(32)
        randomize(m, 9)
        x = rn(2, 32, 6, 6).requires_grad_(True)
        y = m(x)
        dy = rn(*y.shape)
        y.backward(dy)
        save(tag, x=x, y=y, dy=dy, dx=x.grad, **{"p." + k: v for k, v in sd_np(m).items()},
             **{"g." + n: p_.grad for n, p_ in m.named_parameters()})

    # ---- G4 decoder_small: reduced config, 4-D latent, fwd + grads -------------------------
    torch.manual_seed(11)
    dec = fa.Decoder(ch=32, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, resolution=64, z_channels=16)
    dec.post_init(z_channels=32)
    sys.modules["models.init_param"].init_weights(dec, 0.02)
    init_ck = checksum(dec)  # fixed-seed init pin (seed 11, ctor + post_init + init_weights order)
    det_fill_(dec, 12)
    z = rn(2, 32, 4, 4).requires_grad_(True)
    y = dec(z)
    dy = rn(*y.shape)
    y.backward(dy)
    grads = {n: p_.grad for n, p_ in dec.named_parameters()}
    save("decoder_small", z=z, y=y, dy=dy, dz=z.grad,
         **{"ck." + k: v for k, v in init_ck.items()},
         **{"gn." + n: np.array([v.double().norm().item(), v.double().sum().item()]) for n, v in grads.items()},
         **{"g." + n: grads[n] for n in ("conv_out.weight", "conv_in.0.conv.weight", "mid.attn_1.q.weight", "up.0.block.0.nin_shortcut.weight",
                                          "up.3.block.1.norm1.weight", "up.1.upsample.conv.bias", "norm_out.bias")})

    # ---- G4b decoder_full_b1: full-size decoder, fixed-seed init, token input; output slice ----
    torch.manual_seed(21)
    dec_full = fa.Decoder(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, resolution=256, z_channels=16)
    dec_full.post_init(z_channels=32)
    sys.modules["models.init_param"].init_weights(dec_full.conv_in, 0.02)
    sys.modules["models.init_param"].init_weights(dec_full, 0.02)
    full_ck = checksum(dec_full)  # fixed-seed init pin (seed 21)
    det_fill_(dec_full, 22)
    zt = torch.randn(1, 256, 32, generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        yf = dec_full(zt)
    save("decoder_full_b1", z=zt, y_slice=yf[0, :, ::8, ::8], y_sum=np.array([yf.double().sum().item(), yf.double().abs().sum().item()]),
         n_params=np.array(sum(p_.numel() for p_ in dec_full.parameters())), keys=np.array(list(dec_full.state_dict().keys())),
         **{"ck." + k: v for k, v in full_ck.items()})

    # ---- flux Encoder (dead code in the scripts; conv/downsample oracle) ---------------------
    enc = fa.Encoder(resolution=32, in_channels=32, ch=32, ch_mult=(1, 2), num_res_blocks=1, z_channels=16)
    randomize(enc, 13, std=0.04)
    x = rn(1, 32, 16, 16)
    with torch.no_grad():
        y = enc(x)
    save("flux_encoder_small", x=x, y=y, **{"p." + k: v for k, v in sd_np(enc).items()})

    # ---- G5 vae_forward: reduced ViT stand-in (tiny) + MLP + full decoder graph at small z ----
    os.environ["DMVAE_GOLDEN_VIT"] = "vit_tiny"
    vae_mod = sys.modules["models.vae"]
    torch.manual_seed(31)
    orig_dim = None

    class TinyDINO(vae_mod.DINOEncoder):
        def __init__(self, model_size="base", patch_size=16, image_size=256):
            super().__init__(model_size, patch_size, image_size)
            self.dim = 64

    saved = vae_mod.DINOEncoder
    vae_mod.DINOEncoder = TinyDINO
    vae = vae_mod.VAE(z_channels=32, model_size="base")
    vae_mod.DINOEncoder = saved
    vae_keys = list(vae.state_dict().keys())
    init_ck = {k: v for k, v in checksum(vae).items() if not k.startswith("encoder.model")}  # seed-31 init pin (decoder + MLP)
    det_fill_(vae, 33)
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(32)) * 2 - 1
    with torch.no_grad():
        rec, lat = vae(x, return_latent=True)
        enc_lat = vae.encode(x)
        dec_out = vae.decode(lat)
    save("vae_forward_tiny", x_seed=np.array(32), rec_slice=rec[0, :, ::8, ::8], latent=lat,
         rec_sum=np.array([rec.double().sum().item(), rec.double().abs().sum().item()]),
         encode_equal=np.array(float((enc_lat - lat).abs().max())), decode_equal=np.array(float((dec_out - rec).abs().max())),
         keys=np.array(vae_keys), **{"ck." + k: v for k, v in init_ck.items()})
    # full-size key manifest (ViT-L stand-in, no forward): names + shapes only
    os.environ["DMVAE_GOLDEN_VIT"] = "vit_large"
    vae_l = vae_mod.VAE(z_channels=32, model_size="large")
    save("vae_large_manifest", keys=np.array(list(vae_l.state_dict().keys())),
         shapes=np.array([str(tuple(v.shape)) for v in vae_l.state_dict().values()]),
         n_params=np.array(sum(p_.numel() for p_ in vae_l.parameters())))
    del vae_l

    # ---- G6/G7 LPIPS + forward_generator ------------------------------------------------------
    from utils.lpips import LPIPS
    import train_tokenizer
    import train_dmd
    torch.manual_seed(41)
    lp = LPIPS(ckpt_path=REF + "/ckpt_vae/vgg.pth").eval()
    # He-style init for the random VGG trunk keeps activations O(1) through 13 layers
    with torch.no_grad():
        for n_, p_ in lp.net.named_parameters():
            from oracle.detweights import det_tensor
            t_ = det_tensor("net." + n_, p_.shape, 41)
            p_.copy_(t_ * (2.0 ** 0.5) if p_.dim() > 1 else t_ * 0.5)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    rec = (img + 0.2 * rn(2, 3, 64, 64)).requires_grad_(True)
    self_ns = SimpleNamespace(lpips_loss=lp, l1=1.0, l2=0.0, lpips=1.0, disc_weight=0.5, args=SimpleNamespace(disc_start_step=5000))
    rec_loss, log = train_tokenizer.VAELossFunction.forward_generator(self_ns, img, rec, 0)
    rec_loss.backward()
    lp_sd = {k: v for k, v in sd_np(lp).items()}
    save("gen_loss", images=img, recon=rec, rec_loss=rec_loss, d_recon=rec.grad, L1=np.array(log["L1"]), L2=np.array(log["L2"]),
         LPIPS=np.array(log["LPIPS"]), **{"p." + k: v for k, v in lp_sd.items() if k.startswith("lin")},
         vgg_seed=np.array(41), **{"ck." + k: np.array([float(v.astype(np.float64).sum()), float(np.abs(v.astype(np.float64)).sum())])
                                   for k, v in lp_sd.items() if k.startswith("net.")})
    save("lpips_vgg_small", **{"p." + k: v for k, v in lp_sd.items() if k.startswith("net.slice1") or k.startswith("net.slice2")})
    # feature-diff only (G7): random positive feats at reduced sizes
    chns = [64, 128, 256, 512, 512]
    sizes = [16, 8, 4, 2, 1]
    f0 = [torch.relu(rn(2, c, s, s)) for c, s in zip(chns, sizes)]
    f1 = [torch.relu(rn(2, c, s, s)).requires_grad_(True) for c, s in zip(chns, sizes)]
    lins = [lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4]
    lpmod = sys.modules["utils.lpips"]
    val = 0
    for k in range(5):
        d = (lpmod.normalize_tensor(f0[k]) - lpmod.normalize_tensor(f1[k])) ** 2
        val = val + lpmod.spatial_average(lins[k].model(d), keepdim=True)
    val = val.mean()
    val.backward()
    save("lpips_diff", value=val, **{f"f0_{k}": f0[k] for k in range(5)}, **{f"f1_{k}": f1[k] for k in range(5)},
         **{f"df1_{k}": f1[k].grad for k in range(5)}, **{f"w_{k}": lins[k].model[1].weight.reshape(-1) for k in range(5)})

    # ---- G8 dmd_loss (train_dmd + toy) with injected (t, x0, v_*) ------------------------------
    from diffusion.transport import create_transport

    class Inject(nn.Module):
        def __init__(self, vc, vu, ncls):
            super().__init__()
            self.vc, self.vu, self.ncls = vc, vu, ncls

        def forward(self, xt, t, y):
            return self.vu if bool((y == self.ncls).all()) else self.vc

    for tag, cfg in (("dmd_loss_cfg5", 5.0), ("dmd_loss_cfg1", 1.0)):
        B = 4
        lat = (0.8 * rn(B, 32, 16, 16)).requires_grad_(True)
        labels = torch.randint(0, 1000, (B,), generator=g)
        t_raw = torch.rand(B, generator=g)
        x0 = rn(B, 32, 16, 16)
        vtc, vtu, vsc, vsu = (rn(B, 32, 16, 16) for _ in range(4))
        tr = create_transport("Linear", "velocity")
        tr.sample = lambda x1, _t=t_raw, _x0=x0: (_t.clone(), _x0, x1)
        args = SimpleNamespace(t0=0.02, t1=0.98, dmd_cfg_scale=cfg, num_classes=1000)
        ns = SimpleNamespace(args=args, transport=tr, base_model=Inject(vtc, vtu, 1000), sit_wo_ddp=Inject(vsc, vsu, 1000))
        loss, log = train_dmd.VAELossFunction.compute_distribution_matching_loss(ns, lat, labels)
        loss.backward()
        save(tag, latents=lat, labels=labels, t_raw=t_raw, t0=np.array(0.02), t1=np.array(0.98), x0=x0, v_teacher=vtc, v_teacher_u=vtu,
             v_student=vsc, v_student_u=vsu, cfg=np.array(cfg), loss=loss, dmd_loss=np.array(log["dmd_loss"]),
             dmd_gradient_norm=np.array(log["dmd_gradient_norm"]), dlatents=lat.grad)
    # toy "dmd" branch
    sys.path.insert(0, REF + "/toy_example_2d")
    for n_ in ("matplotlib", "matplotlib.pyplot", "matplotlib.colors"):
        try:
            importlib.import_module(n_)
        except ImportError:
            sys.modules[n_] = MagicMock(name=n_)
    toy = _load("toy_dmd", REF + "/toy_example_2d/dmd.py")
    B = 64
    pts = (rn(B, 2)).requires_grad_(True)
    t_raw = torch.rand(B, generator=g)
    x0 = rn(B, 2, 1, 1)
    vt, vs = rn(B, 2, 1, 1), rn(B, 2, 1, 1)
    tr = create_transport("Linear", "velocity")
    tr.sample = lambda x1, _t=t_raw, _x0=x0: (_t.clone(), _x0, x1)
    ns = SimpleNamespace(args=SimpleNamespace(t0=0.0, t1=1.0, dmd_loss_type="dmd"), transport=tr,
                         base_model=Inject(vt, vt, -1), sit_wo_ddp=Inject(vs, vs, -1))
    out = toy.DMDLossFunction.compute_distribution_matching_loss(ns, pts, torch.zeros(B, dtype=torch.long))
    loss = out[0]
    loss.backward()
    save("dmd_loss_toy", points=pts, t_raw=t_raw, x0=x0, v_teacher=vt, v_student=vs, loss=loss, dpoints=pts.grad)

    # ---- G9 transport --------------------------------------------------------------------------
    tr = create_transport("Linear", "velocity")
    x1 = rn(3, 8, 4, 4)
    x0 = rn(3, 8, 4, 4)
    t = torch.rand(3, generator=g)
    _, xt, ut = tr.path_sampler.plan(t, x0, x1)
    mo = rn(3, 8, 4, 4)
    tr.sample = lambda x, _t=t, _x0=x0: (_t, _x0, x)
    _, terms = tr.training_losses(lambda xt_, t_, **k: mo, x1)
    save("transport", x1=x1, x0=x0, t=t, xt=xt, ut=ut, model_out=mo, loss=terms["loss"])

    # ---- G10 latents_to_spatial (bit-exact) -----------------------------------------------------
    tok = rn(2, 256, 32)
    save("latents_to_spatial", tokens=tok, spatial=train_dmd.latents_to_spatial(tok))

    # ---- G11 opt_tail: clip + AdamW + EMA, 6 steps with LambdaLR warm-up ------------------------
    torch.manual_seed(51)
    net = nn.Sequential(nn.Linear(16, 32), nn.SiLU(), nn.Linear(32, 8))
    import copy
    ema = copy.deepcopy(net)
    opt = torch.optim.AdamW(net.parameters(), lr=1e-4, betas=(0.9, 0.95), weight_decay=0.005)
    warm = 4        # the reference's lr_lambda (train_tokenizer.py:385-389) with a short warm-up: lr = 0, 1/4, 2/4, 3/4, 1, 1 x base

    def lr_lambda(step):
        if step < warm:
            return step / warm
        return 1.0
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)
    p0 = {k: v.clone() for k, v in net.state_dict().items()}
    grads_hist, norms, lrs = [], [], []
    for it in range(6):
        lrs.append(opt.param_groups[0]["lr"])
        for p_ in net.parameters():
            p_.grad = 3.0 * torch.randn(p_.shape, generator=g)
        grads_hist.append([p_.grad.clone() for p_ in net.parameters()])
        norms.append(float(torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)))
        opt.step()
        opt.zero_grad()
        sched.step()
        train_tokenizer.update_ema(ema, net)
    save("opt_tail", norms=np.array(norms), lrs=np.array(lrs), warmup_steps=np.array(warm), **{"p0." + k: v for k, v in p0.items()},
         **{f"g{it}.{i}": gr for it, gl in enumerate(grads_hist) for i, gr in enumerate(gl)},
         **{"p6." + k: v for k, v in sd_np(net).items()}, **{"ema6." + k: v for k, v in sd_np(ema).items()})

    # ---- G13 sshape ----------------------------------------------------------------------------
    ss = _load("sshpae", REF + "/toy_example_2d/sshpae.py")
    save("sshape", samples=ss.SShapeDistribution2D(random_state=42).sample(1536)[0])

    # ---- bottleneck MLP ------------------------------------------------------------------------
    mlp = vae_mod.MLP(64, 32, hidden_dim=128)
    randomize(mlp, 61, std=0.1)
    x = rn(2, 16, 64).requires_grad_(True)
    y = mlp(x)
    dy = rn(*y.shape)
    y.backward(dy)
    save("mlp", x=x, y=y, dy=dy, dx=x.grad, **{"p." + k: v for k, v in sd_np(mlp).items()},
         **{"g." + n: p_.grad for n, p_ in mlp.named_parameters()})


if __name__ == "__main__":
    main()
