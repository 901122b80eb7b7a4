"""Generate the discriminator-branch fixtures (tests/golden/{patchgan_small,diffaug,gan_losses}.npz) by importing the reference's
own models/patchgan.py, utils/diffaug.py and train_tokenizer.VAELossFunction (read-only at /root/reference) in THIS container.
Separate from capture_golden.py so that re-running it does not disturb the generator stream of the older fixtures.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_gan.py            (CPU, under a minute)

Stand-ins (documented, no hot-path arithmetic of the reference is replaced):
  * nn.SyncBatchNorm refuses CPU tensors, so after constructing the reference NLayerDiscriminator() (SyncBatchNorm => conv biases
    on) its three SyncBatchNorm modules are swapped for nn.BatchNorm2d holding the same parameters / buffers: on one rank
    SyncBatchNorm *is* F.batch_norm (torch/nn/modules/batchnorm.py: world_size == 1 path).
  * DiffAug draws torch.rand(3) and torch.rand(7, B, 1, 1) internally (diffaug.py:66,69); torch.rand is wrapped for the duration of
    each call so that the draws are recorded in the fixture (and, for the single-transform cases, chosen).
Weights are oracle.detweights.det_tensor values (regenerated from names + seed in the tests), inputs are stored.
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import capture_golden as cg  # noqa: E402
from oracle.detweights import det_fill_patchgan_, det_tensor  # noqa: E402


class RandTap:
    """Context manager: torch.rand returns queued tensors first (then real draws) and records everything it returned."""

    def __init__(self, queued=()):
        self.queue, self.seen = list(queued), []

    def __enter__(self):
        self._orig = torch.rand

        def rand(*size, **kw):
            if self.queue:
                t = self.queue.pop(0)
            else:
                t = self._orig(*size, **kw)
            self.seen.append(t.clone())
            return t
        torch.rand = rand
        return self

    def __exit__(self, *a):
        torch.rand = self._orig


def make_disc(seed):
    pg = sys.modules["models.patchgan"]
    d = pg.NLayerDiscriminator()                      # reference defaults: 3 -> 64 -> 128 -> 256 -> 512 -> 1, SyncBatchNorm
    mods = list(d.main)
    for i, m in enumerate(mods):
        if isinstance(m, nn.SyncBatchNorm):
            bn = nn.BatchNorm2d(m.num_features, eps=m.eps, momentum=m.momentum)
            bn.load_state_dict(m.state_dict())
            d.main[i] = bn
    det_fill_patchgan_(d.state_dict(), seed)
    return d


def main():
    cg.install_stubs()
    torch.set_grad_enabled(True)
    torch.manual_seed(4321)          # DiffAug draws from the GLOBAL generator (utils/diffaug.py:74-113): seed it so the capture is reproducible
    g = torch.Generator().manual_seed(4321)

    # ---- PatchGAN: eval-mode logits, then one training forward + backward ---------------------------------------------------
    disc = make_disc(51)
    buf0 = {k: v.clone() for k, v in disc.state_dict().items() if "running" in k}
    x = (torch.rand(2, 3, 64, 64, generator=g) * 2 - 1).requires_grad_(True)
    disc.eval()
    with torch.no_grad():
        y_eval = disc(x)
    disc.train()
    y = disc(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    grads = {n: p.grad for n, p in disc.named_parameters()}
    cg.save("patchgan_small", seed=np.array(51), x=x.detach(), y_eval=y_eval, y=y.detach(), dy=dy, dx=x.grad,
            keys=np.array(list(disc.state_dict().keys())), n_params=np.array(sum(p.numel() for p in disc.parameters())),
            **{"buf0." + k: v for k, v in buf0.items()},
            **{"buf1." + k: v.clone() for k, v in disc.state_dict().items() if "running" in k or "num_batches" in k},
            **{"gn." + n: np.array([v.double().norm().item(), v.double().sum().item()]) for n, v in grads.items()},
            **{"g." + n: grads[n] for n in ("main.0.bias", "main.3.weight", "main.3.bias", "main.6.weight", "main.9.bias", "main.11.weight",
                                             "main.11.bias", "main.2.bias")})

    # ---- DiffAug: every transform alone and all together, two geometries (even / odd cut-out sizes), gradient of the full chain -----
    da_mod = cg._load("utils.diffaug", cg.REF + "/utils/diffaug.py")
    out = {}
    for tag, (b, h, w) in (("a", (4, 32, 32)), ("b", (3, 36, 28))):
        xi = torch.rand(b, 3, h, w, generator=g) * 2 - 1
        out[f"{tag}.x"] = xi
        for name, flags in (("trans", (0.0, 2.0, 2.0)), ("color", (2.0, 0.0, 2.0)), ("cut", (2.0, 2.0, 0.0)), ("all", (0.0, 0.0, 0.0))):
            aug = da_mod.DiffAug(prob=1.0, cutout=0.2)
            r7 = torch.rand(7, b, 1, 1, generator=g)
            xin = xi.clone().requires_grad_(True)
            with RandTap([torch.tensor(flags), r7]):
                yo = aug.aug(xin, 0)
            out[f"{tag}.{name}.rand01"] = r7.view(7, b)
            out[f"{tag}.{name}.y"] = yo.detach()
            if name == "all":
                dyo = torch.randn(yo.shape, generator=g)
                yo.backward(dyo)
                out[f"{tag}.all.dy"], out[f"{tag}.all.dx"] = dyo, xin.grad
    # extreme draws: translations of -delta / +delta, cut-out centred on the first / last pixel
    xi = out["a.x"]
    r7 = torch.tensor([0.0, 0.999999, 0.5, 0.5, 0.5, 0.0, 0.999999]).view(7, 1, 1, 1).repeat(1, 4, 1, 1)
    with RandTap([torch.zeros(3), r7]):
        out["a.edge.y"] = da_mod.DiffAug(prob=1.0, cutout=0.2).aug(xi.clone(), 0)
    out["a.edge.rand01"] = r7.view(7, 4)
    cg.save("diffaug", **out)

    # ---- discriminator step and generator step with the discriminator branch (train_tokenizer.py:179-227) --------------------------
    import train_tokenizer
    from utils.lpips import LPIPS
    lp = LPIPS(ckpt_path=cg.REF + "/ckpt_vae/vgg.pth").eval()
    with torch.no_grad():
        for n_, p_ in lp.net.named_parameters():
            t_ = det_tensor("net." + n_, p_.shape, 41)
            p_.copy_(t_ * (2.0 ** 0.5) if p_.dim() > 1 else t_ * 0.5)
    disc = make_disc(52)
    img = torch.rand(2, 3, 64, 64, generator=g) * 2 - 1
    # a stand-in "decoder last layer": recon = conv3x3(feat; last) so that the adaptive weight has a parameter to differentiate to
    feat = torch.randn(2, 8, 64, 64, generator=g) * 0.5
    last = nn.Parameter(det_tensor("last.weight", (3, 8, 3, 3), 53))
    recon = F.conv2d(feat, last, padding=1) + 0.9 * img
    ns = SimpleNamespace(lpips_loss=lp, l1=1.0, l2=0.0, lpips=1.0, disc_weight=0.5, args=SimpleNamespace(disc_start_step=0),
                         disc_wo_ddp=disc, disc_ddp=disc, daug=da_mod.DiffAug(prob=1.0, cutout=0.2), bcr_weight=4.0,
                         bcr_strong_aug=da_mod.DiffAug(prob=1, cutout=0.5),
                         vae_wo_ddp=SimpleNamespace(decoder=SimpleNamespace(get_last_layer=lambda: last)))
    with RandTap() as tap_g:
        gen_loss, gen_log = train_tokenizer.VAELossFunction.forward_generator(ns, img, recon, 10)
    gen_loss.backward()
    g_last = last.grad.clone()
    buf_g = {k: v.clone() for k, v in disc.state_dict().items() if "running" in k}
    with RandTap() as tap_d:
        d_loss, d_log = train_tokenizer.VAELossFunction.forward_discriminator(ns, img, recon.detach())
    d_loss.backward()
    dgr = {n: p.grad for n, p in disc.named_parameters()}
    cg.save("gan_losses", disc_seed=np.array(52), vgg_seed=np.array(41), images=img, feat=feat, last=last.detach(), recon=recon.detach(),
            **{"p." + k: v for k, v in cg.sd_np(lp).items() if k.startswith("lin")},
            **{"buf0." + k: v for k, v in buf_g.items()},
            gen_rand01=tap_g.seen[1].view(7, -1), gen_loss=gen_loss.detach(), d_weight=np.array(gen_log["d_weight"]),
            gen_rec_loss=np.array(gen_log["rec_loss"]), g_last=g_last,
            d_rand01_a=tap_d.seen[1].view(7, -1), d_rand01_b=tap_d.seen[3].view(7, -1), d_total=d_loss.detach(),
            **{"dlog." + k: np.array(v) for k, v in d_log.items()},
            **{"buf2." + k: v.clone() for k, v in disc.state_dict().items() if "running" in k or "num_batches" in k},
            **{"dgn." + n: np.array([v.double().norm().item(), v.double().sum().item()]) for n, v in dgr.items()},
            **{"dg." + n: dgr[n] for n in ("main.0.bias", "main.6.weight", "main.11.weight")})


if __name__ == "__main__":
    main()
