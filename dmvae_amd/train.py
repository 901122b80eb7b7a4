"""Step harness: the build's counterpart of the reference's tokenizer / DMD train-step bodies
(train_tokenizer.py:403-437 and train_dmd.py:506-575, loss assembly :179-204 / :204-262).

`TokenizerTrainer.step(images)` = VAE forward (frozen ViT encoder under bf16 autocast, bottleneck MLP and decoder on the
HIP kernels) -> L1 (+L2) + LPIPS (+ build-defined KL/MMD, weight 0 by default) -> backward -> [bucketed RCCL
all-reduce overlapped with backward] -> clip + AdamW + EMA (two launches on flat buffers).  No per-step host sync:
the log scalars stay in one device tensor; call `read_log()` when you want them.

With a discriminator attached (`disc=`; reference default `--disc_type patchgan`, train_tokenizer.py:306-319) and
`global_step >= disc_start_step` the step also runs the reference's adversarial branch: generator term -mean D(aug(recon)) with the
adaptive weight |d rec/d last| / |d g/d last| (train_tokenizer.py:190-203) in eval mode, then the discriminator update -- hinge loss on
D(aug([images; recon])) plus the BCR consistency term, clip, AdamW (train_tokenizer.py:207-227,420-427).
"""
from __future__ import annotations

import warnings
from typing import Dict, Optional

import torch

from . import dist, losses
from .models.vae import VAE
from .optim import FlatAdamWEMA, FlatParams
from .utils.lpips import LPIPS


def frozen_bf16_shadow(module: torch.nn.Module) -> torch.nn.Module:
    """Copy of a FROZEN module whose Linear/Conv2d weights+biases are stored in bf16 -- exactly the values autocast(bf16)
    would cast them to on every call (the reference re-casts the frozen ViT / VGG weights each step under autocast,
    train_tokenizer.py:295-297,410-411).  Norm / LayerScale / embedding parameters stay f32 like under autocast.  The owner
    module (and its state_dict / checkpoint) is untouched."""
    import copy
    sh = copy.deepcopy(module).eval().requires_grad_(False)
    for m in sh.modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)):
            m.weight.data = m.weight.data.to(torch.bfloat16)
            if m.bias is not None:
                m.bias.data = m.bias.data.to(torch.bfloat16)
    return sh


def backward_order_params(vae: VAE):
    """Trainable parameters (decoder + bottleneck) in the order their gradients complete in backward."""
    dec = vae.decoder
    order = [dec.conv_out, dec.norm_out]
    for lvl in range(dec.num_resolutions):                 # up[0] is the highest resolution: last in forward, first in backward
        if hasattr(dec.up[lvl], "upsample"):
            order.append(dec.up[lvl].upsample)
        order += list(reversed(list(dec.up[lvl].block)))
    order += [dec.mid.block_2, dec.mid.attn_1, dec.mid.block_1, dec.conv_in, vae.bottle_neck]
    params, seen = [], set()
    for m in order:
        for p in m.parameters():
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                params.append(p)
    return params


class TokenizerTrainer:
    def __init__(self, vae: VAE, lpips: Optional[LPIPS], lr: float = 1e-4, l1: float = 1.0, l2: float = 0.0, lpips_w: float = 1.0,
                 kl_w: float = 0.0, mmd_w: float = 0.0, warmup_steps: int = 1000, ema_decay: float = 0.9999, max_norm: float = 1.0,
                 bucket_bytes: int = 64 << 20, disc: Optional[torch.nn.Module] = None, disc_weight: float = 0.5,
                 disc_start_step: int = 5000, disc_lr: float = 1e-4, disc_wd: float = 0.0005, disc_warmup_steps: Optional[int] = None,
                 bcr: float = 1.0, bcr_cut: float = 0.2):
        self.vae, self.lpips = vae, lpips
        self.disc = disc if (disc is not None and disc_weight > 0) else None
        self.disc_weight, self.disc_start_step, self.bcr_weight = disc_weight, disc_start_step, bcr
        if self.disc is not None:
            from .utils.diffaug import DiffAug
            self.daug = DiffAug(prob=1.0, cutout=0.2)                  # train_tokenizer.py:174
            self.bcr_strong_aug = DiffAug(prob=1, cutout=bcr_cut)      # :176
            # every discriminator parameter receives two gradients per backward (logits and the BCR pass): accumulate into the
            # flat buffer through autograd, no direct writes; AdamW(betas=(0.9, 0.95), wd=disc_wd), no EMA (:383)
            self.dfp = FlatParams(list(self.disc.parameters()), with_ema=False)
            self.dopt = FlatAdamWEMA(self.dfp, lr=disc_lr, weight_decay=disc_wd, max_norm=max_norm,
                                     warmup_steps=warmup_steps if disc_warmup_steps is None else disc_warmup_steps)
            self.dlog = torch.zeros(8, dtype=torch.float32, device=self.dfp.flat.device)
        self.w = dict(l1=l1, l2=l2, lpips=lpips_w, kl=kl_w, mmd=mmd_w)
        vae.encoder.eval()
        for p in vae.encoder.parameters():                 # train_tokenizer.py:295-297
            p.requires_grad_(False)
        params = backward_order_params(vae)
        n_train = sum(p.numel() for p in vae.parameters() if p.requires_grad)
        assert n_train == sum(p.numel() for p in params), "parameter ordering lost a trainable parameter"
        self.fp = FlatParams(params, with_ema=True)
        self.fp.enable_direct_grads()
        self.opt = FlatAdamWEMA(self.fp, lr=lr, warmup_steps=warmup_steps, ema_decay=ema_decay, max_norm=max_norm)
        self.sync = dist.FlatGradSync(params, self.fp.grad, self.fp.offsets, bucket_bytes=bucket_bytes)
        self.log = torch.zeros(8, dtype=torch.float32, device=self.fp.flat.device)
        self.global_step = 0
        self.refresh_frozen_shadows()

    def refresh_frozen_shadows(self) -> None:
        """(Re)build the bf16 shadows of the frozen encoder and LPIPS trunk; call after loading new weights into them."""
        self._enc = frozen_bf16_shadow(self.vae.encoder)
        self._lpips = self.lpips          # LPIPS runs on the HIP conv kernels with cached bf16 operands: no shadow needed

    def _encode(self, images: torch.Tensor) -> torch.Tensor:
        """Frozen encoder forward (DINOEncoder.forward, models/vae.py:52-53) on the bf16 shadow; widths the HIP LayerNorm covers
        take the fused elementwise path, anything else the stock module."""
        enc = self._enc
        if enc.model.embed_dim % 256 == 0 and enc.model.embed_dim <= 1536:
            from .models.vit_fast import frozen_forward_features
            return frozen_forward_features(enc.model, enc.scale(enc.de_scale(images)))[:, enc.model.num_prefix_tokens:]
        return enc(images)

    def step(self, images: torch.Tensor) -> torch.Tensor:
        vae, w = self.vae, self.w
        self.fp.begin_step()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            with torch.no_grad():
                tokens = self._encode(images)
            latent = vae.bottle_neck(tokens)
            recon = vae.decoder(latent).float()
            l1, l2 = losses.l1_mse(recon, images, w["l1"], w["l2"])
            loss = l1 * w["l1"] + l2 * w["l2"]
            lp = None
            if self.lpips is not None and w["lpips"] != 0:
                lp = self._lpips(images, recon)
                loss = loss + lp * w["lpips"]
            kl = None
            if w["kl"] != 0 or w["mmd"] != 0:
                dm, kl, mmd = losses.kl_mmd_loss(latent, w_kl=w["kl"], w_mmd=w["mmd"])
                loss = loss + dm
            rec_loss = loss
            gan = self.disc is not None and self.global_step >= self.disc_start_step
            if gan:
                loss, d_weight = self._generator_gan_term(rec_loss, recon)
        loss.backward()
        self.sync.wait()
        norm = self.opt.step()
        with torch.no_grad():
            self.log[0], self.log[1], self.log[3] = l1.detach(), l2.detach(), rec_loss.detach()
            if lp is not None:
                self.log[2] = lp.detach()
            self.log[4] = norm[0]
            if kl is not None:
                self.log[5], self.log[6] = kl[-1], mmd.mean()
            if gan:
                self.log[7] = d_weight
        if gan:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                self._discriminator_step(images, recon.detach())
        self.global_step += 1
        return loss.detach()

    def _generator_gan_term(self, rec_loss: torch.Tensor, recon: torch.Tensor):
        return losses.generator_gan_term(rec_loss, recon, self.disc, self.daug, self.vae.decoder.get_last_layer(), self.disc_weight)

    def _discriminator_step(self, images: torch.Tensor, recon: torch.Tensor) -> None:
        """train_tokenizer.py:420-427: discriminator loss, backward, clip_grad_norm_(1.0), AdamW, warm-up schedule."""
        self.dfp.begin_step()
        d_total, log = losses.discriminator_loss(images, recon, self.disc, self.daug, self.bcr_strong_aug, self.bcr_weight)
        d_total.backward()
        if dist.initialized() and dist.get_world_size() > 1:
            self.dfp.grad.div_(dist.get_world_size())
            dist.allreduce(self.dfp.grad)
        dnorm = self.dopt.step()
        with torch.no_grad():
            self.dlog[0], self.dlog[1], self.dlog[2], self.dlog[3] = log["d_loss"], log["bcr_loss"], log["acc_real"], log["acc_fake"]
            self.dlog[4] = dnorm[0]

    def read_disc_log(self) -> Dict[str, float]:
        v = self.dlog.tolist()
        return {"d_loss": v[0], "bcr_loss": v[1], "acc_real": v[2], "acc_fake": v[3], "acc_mean": 0.5 * (v[2] + v[3]), "disc_norm": v[4]}

    def read_log(self) -> Dict[str, float]:
        v = self.log.tolist()       # the single D2H sync
        return {"L1": v[0], "L2": v[1], "LPIPS": v[2], "rec_loss": v[3], "vae_norm": v[4], "KL": v[5], "MMD": v[6], "d_weight": v[7]}

    def checkpoint(self) -> dict:
        """vae.pt layout of the reference (train_tokenizer.py:440-450): vae_wo_ddp + vae_ema state_dicts."""
        sd = {k: v.detach().clone() for k, v in self.vae.state_dict().items()}
        ema = dict(sd)
        names = {id(p): n for n, p in self.vae.named_parameters()}
        for p, e in zip(self.fp.params, self.fp.ema_state()):
            ema[names[id(p)]] = e.detach().clone()
        out = {"vae_wo_ddp": sd, "vae_ema": ema, "steps": self.global_step}
        if self.disc is not None:
            out["disc_wo_ddp"] = {k: v.detach().clone() for k, v in self.disc.state_dict().items()}
        return out


def build_tokenizer_trainer(device="cuda", z_channels=32, model_size="large", seed=42, lpips_ckpt=None, **kw) -> TokenizerTrainer:
    """Random-init model in the reference's constructor order under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=z_channels, model_size=model_size).to(device)
    lp = LPIPS(ckpt_path=lpips_ckpt).eval().requires_grad_(False).to(device)
    if lpips_ckpt is None:          # no trunk / lin weights offline: deterministic positive lin weights
        with torch.no_grad():
            for lin in (lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4):
                lin.model[-1].weight.fill_(1.0 / lin.model[-1].weight.shape[1])
    if kw.pop("with_disc", False):      # reference default discriminator (train_tokenizer.py:316-317: NLayerDiscriminator() + init_weights(disc, 0.02))
        from .models.init_param import init_weights
        from .models.patchgan import NLayerDiscriminator
        disc = NLayerDiscriminator()
        init_weights(disc, 0.02)
        kw["disc"] = disc.to(device)
    return TokenizerTrainer(vae, lp, **kw)
