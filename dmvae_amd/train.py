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

import os
import warnings
from typing import Dict, Optional

import torch

from . import dist, losses, parity
from .transport import cpu_rand_like_batch
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


class _AdversarialBranch:
    """The discriminator side shared by the tokenizer and the DMD stage (train_tokenizer.py:190-227,420-427; train_dmd.py:244-256,264-283,546-556):
    generator term -mean D(aug(recon)) with the adaptive weight at the decoder's last layer, and the discriminator update (hinge + BCR, clip, AdamW with
    its own warm-up counter, bucketed asynchronous gradient all-reduce).  `self.disc` is None when the branch is off."""

    def _init_disc(self, disc, disc_weight, disc_start_step, disc_lr, disc_wd, warmup_steps, max_norm, bcr, bcr_cut, bucket_bytes):
        self.disc = disc if (disc is not None and disc_weight > 0) else None
        self.disc_weight, self.disc_start_step, self.bcr_weight = disc_weight, disc_start_step, bcr
        if self.disc is None:
            return
        from .utils.diffaug import DiffAug
        self.daug = DiffAug(prob=1.0, cutout=0.2)                  # train_tokenizer.py:174
        self.bcr_strong_aug = DiffAug(prob=1, cutout=bcr_cut)      # :176
        # every discriminator parameter receives two gradients per backward (logits and the BCR pass): accumulate into the
        # flat buffer through autograd, no direct writes; AdamW(betas=(0.9, 0.95), wd=disc_wd), no EMA (:383)
        self.dfp = FlatParams(list(self.disc.parameters()), with_ema=False)
        self.dopt = FlatAdamWEMA(self.dfp, lr=disc_lr, weight_decay=disc_wd, max_norm=max_norm, warmup_steps=warmup_steps)
        self.dlog = torch.zeros(8, dtype=torch.float32, device=self.dfp.flat.device)
        # discriminator gradients (train_tokenizer.py:319: its own DDP wrapper): bucketed asynchronous all-reduce from the gradient hooks like the
        # VAE's; FlatParams holds them in forward order, backward completes them last-to-first, so the buckets fill back to front
        self.dsync = dist.FlatGradSync(self.dfp.params, self.dfp.grad, self.dfp.offsets, bucket_bytes=min(bucket_bytes, 4 << 20))

    def _gan_active(self) -> bool:
        return self.disc is not None and self.global_step >= self.disc_start_step

    def _generator_gan_term(self, rec_loss: torch.Tensor, recon: torch.Tensor):
        return losses.generator_gan_term(rec_loss, recon, self.disc, self.daug, self.vae.decoder.get_last_layer(), self.disc_weight)

    def _discriminator_step(self, images: torch.Tensor, recon: torch.Tensor) -> None:
        """train_tokenizer.py:420-427 / train_dmd.py:546-556: discriminator loss, backward, clip_grad_norm_(1.0), AdamW, warm-up schedule."""
        self.dfp.begin_step()
        d_total, log = losses.discriminator_loss(images, recon, self.disc, self.daug, self.bcr_strong_aug, self.bcr_weight)
        d_total.backward()
        self.dsync.wait()
        dnorm = self.dopt.step()
        with torch.no_grad():
            self.dlog[0], self.dlog[1], self.dlog[2], self.dlog[3] = log["d_loss"], log["bcr_loss"], log["acc_real"], log["acc_fake"]
            self.dlog[4] = dnorm[0]

    def read_disc_log(self) -> Dict[str, float]:
        v = self.dlog.tolist()
        return {"d_loss": v[0], "bcr_loss": v[1], "acc_real": v[2], "acc_fake": v[3], "acc_mean": 0.5 * (v[2] + v[3]), "disc_norm": v[4]}


def _rng_state(all_ranks: bool = False) -> dict:
    """CPU and current-device generator states of THIS rank, labelled with (rank, world): what DiffAug (utils/diffaug.py: `torch.rand(3)` on the CPU,
    `torch.rand(7, B, 1, 1)` on the device) and the transport's sampling (`randn_like` on the device, `rand` on the CPU) consume.  The reference seeds rank r
    with seed + 10000 r (train_dmd.py:140-146) and writes checkpoints from the master only (train_tokenizer.py:438), so one rank's generators are NOT the
    job's: `all_ranks=True` (a collective -- every rank must call it) gathers every rank's states into `ranks`, so that each rank of a resumed run of the
    same world size draws what it would have drawn uninterrupted; without it the entry restores the rank that wrote it and leaves the others alone."""
    st = {"rank": dist.get_rank(), "world": dist.get_world_size(), "cpu": torch.get_rng_state()}
    if torch.cuda.is_available():
        st["cuda"] = torch.cuda.get_rng_state()
    if all_ranks and dist.initialized() and st["world"] > 1:
        import torch.distributed as tdist
        mine = {k: v for k, v in st.items() if k in ("cpu", "cuda")}
        got = [None] * st["world"]
        tdist.all_gather_object(got, mine)
        st["ranks"] = {r: g for r, g in enumerate(got)}
    return st


def _set_rng_state(st) -> bool:
    """Install the generator states a checkpoint holds FOR THIS RANK: its entry of `ranks` when the checkpoint was gathered over the same world size, the
    single entry when this rank (of the same world size) wrote it; a checkpoint without labels is a single-process one (rank 0 of 1).  Any other rank keeps
    the generators it was seeded with (seed + 10000 rank): installing rank 0's state everywhere would make every rank draw the same DiffAug parameters, DMD
    timesteps / noise and label drop-outs on different data shards.  Returns whether a state was installed."""
    rank, world = dist.get_rank(), dist.get_world_size()
    if not st:
        return False
    if st.get("ranks") is not None and int(st.get("world", 1)) == world and rank in st["ranks"]:
        mine = st["ranks"][rank]
    elif int(st.get("rank", 0)) == rank and int(st.get("world", 1)) == world:
        mine = st
    else:
        if world > 1:       # a resumed N-rank job whose checkpoint holds only the master's generators: this rank replays its seed + 10000 rank stream from the start
            warnings.warn(f"checkpoint holds no generator state for rank {rank} of {world} (written with all_ranks_rng=False?): this rank restarts its seeded "
                          "RNG streams (DiffAug draws, DMD noise / timesteps, label drop-outs) instead of resuming them", stacklevel=3)
        return False
    torch.set_rng_state(mine["cpu"].cpu())
    if "cuda" in mine and torch.cuda.is_available():
        torch.cuda.set_rng_state(mine["cuda"].cpu())
    return True


class TokenizerTrainer(_AdversarialBranch):
    def __init__(self, vae: VAE, lpips: Optional[LPIPS], lr: float = 1e-4, l1: float = 1.0, l2: float = 0.0, lpips_w: float = 1.0,
                 kl_w: float = 0.0, mmd_w: float = 0.0, warmup_steps: int = 1000, ema_decay: float = 0.9999, max_norm: float = 1.0,
                 bucket_bytes: int = 64 << 20, disc: Optional[torch.nn.Module] = None, disc_weight: float = 0.5,
                 disc_start_step: int = 5000, disc_lr: float = 1e-4, disc_wd: float = 0.0005, disc_warmup_steps: Optional[int] = None,
                 bcr: float = 1.0, bcr_cut: float = 0.2, posterior_kl_w: float = 0.0):
        """`kl_w`, `mmd_w`, `posterior_kl_w` weigh BUILD-DEFINED terms the reference does not have (all 0 by default = the reference's
        loss): batch-moment KL, RBF-mixture MMD, and -- for a `VAE(reparameterize=True)` -- the posterior-form KL of the sampled latent."""
        self.vae, self.lpips = vae, lpips
        self.posterior_kl_w = float(posterior_kl_w)
        if self.posterior_kl_w != 0 and not getattr(vae, "reparameterize", False):
            raise ValueError("posterior_kl_w needs a VAE built with reparameterize=True (the reference's VAE has no (mu, logvar) head: models/vae.py:90-98)")
        self._init_disc(disc, disc_weight, disc_start_step, disc_lr, disc_wd, warmup_steps if disc_warmup_steps is None else disc_warmup_steps,
                        max_norm, bcr, bcr_cut, bucket_bytes)
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
        self.sync_initial_state()
        self.refresh_frozen_shadows()

    def sync_initial_state(self) -> int:
        """DistributedDataParallel's constructor-time broadcast (train_tokenizer.py:302,319): every rank starts from rank 0's parameters and buffers
        (frozen encoder, LPIPS, BatchNorm running statistics included) and from rank 0's EMA / optimiser state, whatever seed each rank built its
        modules with.  No-op on a single rank.  Returns the number of collectives."""
        extra = [self.fp.flat, self.fp.ema, self.opt.exp_avg, self.opt.exp_avg_sq]
        if self.disc is not None:
            extra += [self.dfp.flat, self.dopt.exp_avg, self.dopt.exp_avg_sq]
        n = dist.broadcast_module_state(self.vae, self.lpips, self.disc, extra=extra)
        if n:
            self.fp.after_external_update()
            if self.disc is not None:
                self.dfp.after_external_update()
        return n

    def refresh_frozen_shadows(self) -> None:
        """(Re)build the bf16 shadows of the frozen encoder and LPIPS trunk; call after loading new weights into them."""
        self._enc = self.vae.encoder if parity.on() else frozen_bf16_shadow(self.vae.encoder)      # parity mode: the f32 weights themselves
        self._lpips = self.lpips          # LPIPS runs on the HIP conv kernels with cached bf16 operands: no shadow needed

    def _encode(self, images: torch.Tensor) -> torch.Tensor:
        """Frozen encoder forward (DINOEncoder.forward, models/vae.py:52-53) on the bf16 shadow; widths the HIP LayerNorm covers
        take the fused elementwise path, anything else the stock module."""
        enc = self._enc
        if parity.on():
            return enc(images)                # DinoV2ViT.forward_features -> vit_fast.parity_forward_features (f32, split-operand GEMMs)
        from .models.vit_fast import frozen_forward_features, hip_path_supported
        if hip_path_supported(enc.model, enc.model.pos_embed.shape[1]):
            return frozen_forward_features(enc.model, enc.scale(enc.de_scale(images)))[:, enc.model.num_prefix_tokens:]
        return enc(images)                    # raises unless DMVAE_ALLOW_STOCK=1 (dmvae_amd/_stock.py): no silent ATen route

    def step(self, images: torch.Tensor) -> torch.Tensor:
        vae, w = self.vae, self.w
        self.fp.begin_step()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not parity.on()):      # the reference's autocast (train_tokenizer.py:410); off in the fp32 parity mode
            with torch.no_grad():
                tokens = self._encode(images)
            latent = vae.latent(tokens)          # = vae.bottle_neck(tokens) unless the build-defined reparameterise hook is on
            recon = vae.decoder(latent).float()
            l1, l2 = losses.l1_mse(recon, images, w["l1"], w["l2"])
            loss = l1 * w["l1"] + l2 * w["l2"]
            lp = None
            if self.lpips is not None and w["lpips"] != 0:
                lp = self._lpips(images, recon)
                loss = loss + lp * w["lpips"]
            kl = dm = None
            if w["kl"] != 0 or w["mmd"] != 0:
                dm, kl, mmd = losses.kl_mmd_loss(latent, w_kl=w["kl"], w_mmd=w["mmd"])
                loss = loss + dm
            if self.posterior_kl_w != 0:
                pk = vae.posterior_kl * self.posterior_kl_w
                loss = loss + pk
                dm = pk if dm is None else dm + pk          # `extra` of the single-pass adversarial backward: terms that do not run through recon
            rec_loss = loss
            gan = self._gan_active()
            single = gan and losses.GAN_SINGLE_PASS and not parity.on()
            if single:      # forward of the adversarial term AND the whole backward, one sweep through LPIPS and through the discriminator (losses.generator_gan_backward)
                loss, d_weight = losses.generator_gan_backward(rec_loss, recon, self.disc, self.daug, self.vae.decoder.get_last_layer(), self.disc_weight, extra=dm)
            elif gan:
                loss, d_weight = self._generator_gan_term(rec_loss, recon)
        if not single:
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

    def read_log(self) -> Dict[str, float]:
        v = self.log.tolist()       # the single D2H sync
        return {"L1": v[0], "L2": v[1], "LPIPS": v[2], "rec_loss": v[3], "vae_norm": v[4], "KL": v[5], "MMD": v[6], "d_weight": v[7]}

    def _opt_param_order(self):
        """The parameter list the reference builds `optimizer_vae` over in THIS stage: `[p for p in vae_wo_ddp.parameters() if p.requires_grad]`
        (train_tokenizer.py:381-382, after `requires_grad(vae.encoder, False)` at :297) -- indices 0 .. n_trainable - 1, the frozen encoder's ~300 tensors
        are not part of it.  (train_dmd.py:474 and train_diffusion.py:209 build theirs over every parameter.)"""
        return [p for p in self.vae.parameters() if p.requires_grad]

    def checkpoint(self, all_ranks_rng: bool = False) -> dict:
        """NOT a collective by default: callable from the master alone, the way the reference writes checkpoints (`if dist.is_master() and ...`,
        train_tokenizer.py:439).  `all_ranks_rng=True` is the opt-in that gathers EVERY rank's generator states (`tdist.all_gather_object`): then every rank
        must call checkpoint(), whoever writes the file.

        The reference's checkpoint dict (train_tokenizer.py:440-450): vae_wo_ddp / vae_ema / disc_wo_ddp state_dicts, opt_vae in torch.optim.AdamW's
        layout over the TRAINABLE parameters (`_opt_param_order`), opt_disc over `disc.parameters()`, scheduler_vae / scheduler_disc, steps, and -- beyond the
        reference -- the CPU / device generator states (`rng`: DiffAug's and the transport's draws continue where they stopped).  `torch.save` it as
        `{step:07d}.pt`; `VAE.load_pretrained` reads the first two entries."""
        sd = {k: v.detach().clone() for k, v in self.vae.state_dict().items()}
        ema = dict(sd)
        names = {id(p): n for n, p in self.vae.named_parameters()}
        for p, e in zip(self.fp.params, self.fp.ema_state()):
            ema[names[id(p)]] = e.detach().clone()
        out = {"vae_wo_ddp": sd, "vae_ema": ema, "steps": self.global_step, "opt_vae": self.opt.state_dict(self._opt_param_order()),
               "scheduler_vae": self.opt.scheduler_state_dict(), "disc_wo_ddp": None, "opt_disc": None, "scheduler_disc": None}
        if self.disc is not None:
            out["disc_wo_ddp"] = {k: v.detach().clone() for k, v in self.disc.state_dict().items()}
            out["opt_disc"] = self.dopt.state_dict(list(self.disc.parameters()))
            out["scheduler_disc"] = self.dopt.scheduler_state_dict()
        out["rng"] = _rng_state(bool(all_ranks_rng))      # all_ranks_rng: a collective (every rank calls checkpoint()); default: this rank's generators only
        return out

    def load(self, ckpt: dict) -> None:
        """Resume from `checkpoint()` (or a reference checkpoint of the same layout): weights into the flat buffer (the parameters are views of it),
        EMA, optimiser moments and step counters, then the derived state -- cached bf16 operands, the frozen encoder's bf16 shadow."""
        self.vae.load_state_dict(ckpt["vae_wo_ddp"], strict=True)
        names = {id(p): n for n, p in self.vae.named_parameters()}
        ema_sd = ckpt.get("vae_ema") or ckpt["vae_wo_ddp"]
        with torch.no_grad():
            for p, e in zip(self.fp.params, self.fp.ema_state()):
                e.copy_(ema_sd[names[id(p)]])
        if ckpt.get("opt_vae") is not None:
            self.opt.load_state_dict(ckpt["opt_vae"], self._opt_param_order())
        if self.disc is not None and ckpt.get("disc_wo_ddp") is not None:
            self.disc.load_state_dict(ckpt["disc_wo_ddp"], strict=True)
            if ckpt.get("opt_disc") is not None:
                self.dopt.load_state_dict(ckpt["opt_disc"], list(self.disc.parameters()))
            self.dfp.after_external_update()
        self.global_step = int(ckpt.get("steps", 0))
        self.fp.after_external_update()
        self.refresh_frozen_shadows()
        _set_rng_state(ckpt.get("rng"))


def backward_order_params_full(vae: VAE):
    """Every trainable parameter of the VAE, decoder / bottleneck first (backward_order_params), then the encoder's blocks last-to-first and
    its embeddings -- the order their gradients complete when the encoder trains too (train_dmd.py:519)."""
    params = backward_order_params(vae)
    seen = {id(p) for p in params}
    enc = vae.encoder.model
    order = [enc.norm] + list(reversed(list(enc.blocks))) + [enc.patch_embed]
    for m in order:
        for p in m.parameters():
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                params.append(p)
    for p in vae.parameters():
        if id(p) not in seen and p.requires_grad:
            seen.add(id(p))
            params.append(p)
    return params


def _batchable(model) -> bool:
    """A velocity model whose 2B-sample call equals two B-sample calls: this build's LightningDiT on its HIP route (every op per sample), not drawing label
    drop-outs (eval mode)."""
    from .models.lightningdit import LightningDiT
    from .models import lightningdit_fast
    return isinstance(model, LightningDiT) and not model.training and lightningdit_fast.structurally_supported(model)


class DMDTrainer(_AdversarialBranch):
    """Step harness of the distribution-matching stage (train_dmd.py:506-575): every `vae_train_every`-th step the whole VAE (encoder
    included, :519) trains on  rec_loss + [from `disc_start_step` on, with a discriminator attached] the adaptive-weight adversarial term
    + dmd_weight * DMD loss (:233-262, :204-230) and the discriminator takes its hinge + BCR step (:546-556); then -- every step -- the student
    velocity model trains on the flow-matching loss of the current latents (transport.training_losses, transport.py:119-164).  Module modes
    follow the reference: the student is in eval mode (no label dropout) for the DMD evaluations of a VAE turn (:534) and in train mode for its
    own turn (:561-562).

    `teacher` / `student` are callables `f(xt [B,C,h,w], t [B], labels [B]) -> velocity` (the reference's LightningDiT = models/lightningdit.py here;
    any nn.Module works).  The VAE side runs on the HIP kernels: trainable ViT encoder (functional.VitBlockFn), bottleneck,
    decoder, LPIPS, `losses.dmd_loss` (csrc/losses.hip::dmd_*).  The student is updated with torch.optim.AdamW like the reference's
    optimizer_sit (:473).  The reference keeps no EMA in this stage."""

    def __init__(self, vae: VAE, lpips: Optional[LPIPS], teacher, student, lr: float = 1e-4, diff_lr: float = 1e-4, wd: float = 0.005,
                 l1: float = 1.0, l2: float = 0.0, lpips_w: float = 1.0, dmd_weight: float = 5.0, dmd_cfg_scale: float = 5.0, num_classes: int = 1000,
                 t0: float = 0.0, t1: float = 1.0, latent_mean: float = 0.0, latent_scale: float = 1.0, vae_train_every: int = 5,
                 time_dist_shift: float = 1.0, warmup_steps: int = 1000, max_norm: float = 1.0, bucket_bytes: int = 64 << 20,
                 disc: Optional[torch.nn.Module] = None, disc_weight: float = 0.5, disc_start_step: int = 0, disc_lr: float = 1e-4,
                 disc_wd: float = 0.0005, bcr: float = 1.0, bcr_cut: float = 0.2, batch_cfg: Optional[bool] = None, direct_grads: bool = True,
                 overlap_optimizer: bool = False):
        self.vae, self.lpips, self.teacher, self.student = vae, lpips, teacher, student
        self._init_disc(disc, disc_weight, disc_start_step, disc_lr, disc_wd, warmup_steps, max_norm, bcr, bcr_cut, bucket_bytes)     # train_dmd.py:92-93,475
        self.w = dict(l1=l1, l2=l2, lpips=lpips_w)
        self.dmd_weight, self.cfg, self.num_classes = dmd_weight, dmd_cfg_scale, num_classes
        # Conditional + unconditional evaluation of a velocity model as ONE 2B-sample call.  None (default): only when BOTH models are this build's
        # LightningDiT on its per-sample HIP route and in eval mode at the call (`_batchable`) -- any other callable / nn.Module (a graph captured at
        # batch B, batch-coupled ops, a per-call RNG draw such as label drop-out in train mode) gets the reference's four B-sized calls
        # (train_dmd.py:211-217).  True: the caller vouches that its models are per-sample; False: never.
        self.batch_cfg = batch_cfg
        self.t0, self.t1, self.latent_mean, self.latent_scale = t0, t1, latent_mean, latent_scale
        self.vae_train_every, self.time_dist_shift, self.max_norm = vae_train_every, time_dist_shift, max_norm
        for p in vae.parameters():
            p.requires_grad_(True)                                         # train_dmd.py:519
        params = backward_order_params_full(vae)
        assert sum(p.numel() for p in params) == sum(p.numel() for p in vae.parameters())
        # decoder, bottleneck and encoder-block gradients are written by the HIP Functions straight into the flat buffer (each of those parameters
        # receives exactly one gradient per backward); the embeddings' gradients arrive through stock autograd and accumulate into their zeroed views
        self.fp = FlatParams(params, with_ema=False)
        if direct_grads:
            from .models.vit_fast import hip_path_supported
            vit = vae.encoder.model
            pre = ("decoder.", "bottle_neck.")
            if getattr(vit, "blocks", None) is not None and hasattr(vit, "pos_embed") and hip_path_supported(vit, vit.pos_embed.shape[1]):
                pre += ("encoder.model.blocks.", "encoder.model.norm.")      # else the encoder runs on the stock modules and autograd owns its gradients
            self.fp.enable_direct_grads(only=[p for n_, p in vae.named_parameters() if p.requires_grad and n_.startswith(pre)])
        self.fp.enable_bf16_shadow()          # Linear weights of the trainable ViT / bottleneck: bf16 GEMM operands written by the optimiser step
        self.fp.enable_transposed_shadow()    # ... and their transposed copies (the input-gradient operands) by one batched launch after it
        self.opt = FlatAdamWEMA(self.fp, lr=lr, weight_decay=wd, warmup_steps=warmup_steps, max_norm=max_norm)
        self.sync = dist.FlatGradSync(params, self.fp.grad, self.fp.offsets, bucket_bytes=bucket_bytes)
        # the student's AdamW (train_dmd.py:473, :565-575) on flat buffers like the VAE's: one norm pass + one fused update instead of torch's
        # multi-pass foreach kernels over 675 M parameters; block parameters receive their gradients directly from functional.DitBlockFn
        sp = [p for p in student.parameters() if p.requires_grad] if hasattr(student, "parameters") else []
        self.sfp = self.sopt = None
        if sp:
            self.sfp = FlatParams(sp, with_ema=False)
            self.sfp.enable_bf16_shadow()
            self.sfp.enable_transposed_shadow()      # 142 Linear weights of LightningDiT-XL/1: one transpose launch per step instead of one per weight
            # every block parameter, the adaLN modulation Linears included (functional.LinearFn on csrc/linear_rows.hip writes their gradients in place too: a
            # third of the model's parameters, 28 x 32 MB of accumulate launches per step before); one gradient per parameter and backward: the student's own turn
            direct = [p for n_, p in student.named_parameters() if p.requires_grad and n_.startswith("blocks.")]
            from .models.lightningdit import LightningDiT
            if isinstance(student, LightningDiT) and direct:
                self.sfp.enable_direct_grads(only=direct)
            self.sopt = FlatAdamWEMA(self.sfp, lr=diff_lr, weight_decay=wd, warmup_steps=warmup_steps, max_norm=max_norm)
            # train_dmd.py:355: the student's DDP wrapper -- 2.7 GB of gradients, bucketed and overlapped with its backward
            self.ssync = dist.FlatGradSync(self.sfp.params, self.sfp.grad, self.sfp.offsets, bucket_bytes=bucket_bytes)
        self.log = torch.zeros(10, dtype=torch.float32, device=self.fp.flat.device)
        self.global_step = 0
        self.sync_initial_state()
        if overlap_optimizer:
            # the two AdamW updates leave the critical path (optim.FlatAdamWEMA.enable_overlap): the VAE's runs beside the student's turn of the same step, the
            # student's (20 GB of HBM traffic, ~4 ms) beside the NEXT step's encoder forward -- neither of which touches the weights being updated
            self.opt.enable_overlap()
            if self.sopt is not None:
                self.sopt.enable_overlap()

    def wait_optimizers(self) -> None:
        """Current stream waits for any optimiser step still running on its side stream: call before reading weights / flat buffers outside `step`."""
        self.opt.wait()
        if self.sopt is not None:
            self.sopt.wait()

    def sync_initial_state(self) -> int:
        """DDP's constructor-time broadcast for vae_ddp / sit_ddp (train_dmd.py:348,355) plus the frozen teacher and LPIPS."""
        extra = [self.fp.flat, self.opt.exp_avg, self.opt.exp_avg_sq]
        if self.sfp is not None:
            extra += [self.sfp.flat, self.sopt.exp_avg, self.sopt.exp_avg_sq]
        if self.disc is not None:
            extra += [self.dfp.flat, self.dopt.exp_avg, self.dopt.exp_avg_sq]
        mods = [m for m in (self.vae, self.lpips, self.teacher, self.student, self.disc) if isinstance(m, torch.nn.Module)]
        n = dist.broadcast_module_state(*mods, extra=extra)
        if n:
            self.fp.after_external_update()
            if self.sfp is not None:
                self.sfp.after_external_update()
            if self.disc is not None:
                self.dfp.after_external_update()
        return n

    def _sample(self, x1: torch.Tensor):
        """Transport.sample (transport.py:105-116): x0 on the device generator, t on the CPU generator, optional time shift."""
        x0 = torch.randn_like(x1)
        t = cpu_rand_like_batch(x1).to(x1)        # the CPU generator's draw, copied without stalling the launch queue
        s = self.time_dist_shift
        t = 1 - s * (1 - t) / (1 + (s - 1) * (1 - t))
        return t, x0

    def _dmd(self, latents: torch.Tensor, labels: torch.Tensor):
        """compute_distribution_matching_loss (train_dmd.py:204-230)."""
        if self.sopt is not None:
            self.sopt.wait()                      # the student's weights: its last update may still be on the side stream
        t, x0 = self._sample(latents)
        t = (t * (self.t1 - self.t0) + self.t0)
        xt = losses.dmd_make_xt(latents, x0, t)
        with torch.no_grad():
            if self.cfg > 1 and (self.batch_cfg if self.batch_cfg is not None else (_batchable(self.teacher) and _batchable(self.student))):
                # the conditional and the unconditional evaluation of a model as ONE call on 2B samples: every op of the velocity
                # model is per sample (per token row, per (sample, head)), so each half equals the B-sized call; twice the rows per GEMM fill the chip
                # (B = 16: 4096 -> 8192 token rows) and half the launches
                b = xt.shape[0]
                x2, t2 = torch.cat([xt, xt]), torch.cat([t, t])
                y2 = torch.cat([labels, torch.ones_like(labels) * self.num_classes])
                v2t, v2s = self.teacher(x2, t2, y2), self.student(x2, t2, y2)
                vt, vtu, vs, vsu = v2t[:b], v2t[b:], v2s[:b], v2s[b:]
            else:
                vt, vs = self.teacher(xt, t, labels), self.student(xt, t, labels)
                vtu = vsu = None
                if self.cfg > 1:
                    un = torch.ones_like(labels) * self.num_classes
                    vtu, vsu = self.teacher(xt, t, un), self.student(xt, t, un)
        return losses.dmd_loss(latents, xt, t, vt, vs, vtu, vsu, cfg=self.cfg)

    def step(self, images: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        vae, w = self.vae, self.w
        vae_turn = self.global_step % self.vae_train_every == 0
        student_is_module = isinstance(self.student, torch.nn.Module)
        self.opt.wait()                           # the encoder's weights (every step reads them)
        for p in getattr(self.student, "parameters", lambda: [])():
            p.requires_grad_(False)
        if vae_turn and student_is_module:
            self.student.eval()           # train_dmd.py:534-535: no label dropout in the four no-grad velocity evaluations of the DMD loss
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not parity.on()):      # the reference's autocast (train_dmd.py:516); off in the fp32 parity mode
            if vae_turn:
                self.fp.begin_step()
                vae.train()
                recon, z = vae(images, return_latent=True)
            else:
                with torch.no_grad():                               # student-only step: the latents are all that is needed (train_dmd.py:520-523)
                    enc = vae.encoder
                    from .models import vit_fast
                    if not parity.on() and vit_fast.hip_path_supported(enc.model, enc.model.pos_embed.shape[1]):
                        tok = vit_fast.trainable_forward_features(enc.model, enc.scale(enc.de_scale(images)))[:, enc.model.num_prefix_tokens:]
                    else:
                        tok = enc(images)
                    z = vae.bottle_neck(tok)
                    recon = None
            latents = losses.latents_to_spatial((z.float() - self.latent_mean) * self.latent_scale)
            if vae_turn:
                l1, l2 = losses.l1_mse(recon, images, w["l1"], w["l2"])
                loss = l1 * w["l1"] + l2 * w["l2"]
                lp = None
                if self.lpips is not None and w["lpips"] != 0:
                    lp = self.lpips(images, recon)
                    loss = loss + lp * w["lpips"]
                rec_loss = loss
                gan = self._gan_active()
                if gan:                                               # train_dmd.py:244-256 (before the DMD term, like there)
                    loss, d_weight = self._generator_gan_term(rec_loss, recon)
                dlog = None
                if self.dmd_weight > 0:
                    dmd, dlog = self._dmd(latents, labels)
                    loss = loss + dmd * self.dmd_weight
        if vae_turn:
            loss.backward()
            self.sync.wait()

            def vae_log(norm):
                with torch.no_grad():
                    self.log[0], self.log[1], self.log[3], self.log[4] = l1.detach(), l2.detach(), rec_loss.detach(), norm[0]
                    if lp is not None:
                        self.log[2] = lp.detach()
                    if dlog is not None:
                        self.log[5], self.log[6] = dlog[0], dlog[1]
                    if gan:
                        self.log[9] = d_weight
            self.opt.step(then=vae_log)
            if gan:                                                   # :546-556
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    self._discriminator_step(images, recon.detach())
        # student turn (every step): flow-matching loss on the current latents (transport.training_losses)
        sloss = None
        if self.sopt is not None:
            for p in self.sfp.params:
                p.requires_grad_(True)
            if student_is_module:
                self.student.train()      # :561-562: label dropout (classifier-free guidance training) in the student's own turn
            self.sfp.begin_step()
            x1 = latents.detach()
            t, x0 = self._sample(x1)
            te = t.view(-1, *([1] * (x1.dim() - 1)))
            xt, ut = te * x1 + (1 - te) * x0, x1 - x0                    # ICPlan.plan (path.py:114-136)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not parity.on()):
                out = self.student(xt, t, labels)
                sloss = ((out.float() - ut) ** 2).flatten(1).mean(1).mean()
            sloss.backward()
            self.ssync.wait()

            def student_log(snorm):
                with torch.no_grad():
                    self.log[7], self.log[8] = sloss.detach(), snorm[0]
            self.sopt.step(then=student_log)
        self.global_step += 1
        return (loss if vae_turn else sloss).detach()

    def checkpoint(self, all_ranks_rng: bool = False) -> dict:
        """train_dmd.py:577-590: model (the student) / vae_wo_ddp / disc_wo_ddp state_dicts, opt_sit / opt_vae / opt_disc, steps; `rng` = this rank's
        generator states.  NOT a collective by default -- the reference writes from the master only (train_dmd.py:593) --; all_ranks_rng=True gathers every
        rank's generator states instead and is then a collective: every rank calls checkpoint()."""
        self.wait_optimizers()
        clone = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
        out = {"model": clone(self.student) if isinstance(self.student, torch.nn.Module) else None, "vae_wo_ddp": clone(self.vae),
               "disc_wo_ddp": clone(self.disc) if self.disc is not None else None,
               "opt_sit": self.sopt.state_dict(list(self.student.parameters())) if self.sopt is not None else None,
               "opt_vae": self.opt.state_dict(list(self.vae.parameters())),
               "opt_disc": self.dopt.state_dict(list(self.disc.parameters())) if self.disc is not None else None, "steps": self.global_step,
               "rng": _rng_state(bool(all_ranks_rng))}
        return out

    def load(self, ckpt: dict) -> None:
        """Resume from `checkpoint()` (or a reference checkpoint, train_dmd.py:577-590); entries the checkpoint holds as None (absent branches) are skipped.
        The warm-up position comes back with the optimiser steps; the generator states (`rng`) when the checkpoint has them."""
        self.wait_optimizers()
        self.vae.load_state_dict(ckpt["vae_wo_ddp"], strict=True)
        if ckpt.get("opt_vae") is not None:
            self.opt.load_state_dict(ckpt["opt_vae"], list(self.vae.parameters()))
        self.fp.after_external_update()
        if self.sopt is not None and ckpt.get("model") is not None:
            self.student.load_state_dict(ckpt["model"], strict=True)
            if ckpt.get("opt_sit") is not None:
                self.sopt.load_state_dict(ckpt["opt_sit"], list(self.student.parameters()))
            self.sfp.after_external_update()
        if self.disc is not None and ckpt.get("disc_wo_ddp") is not None:
            self.disc.load_state_dict(ckpt["disc_wo_ddp"], strict=True)
            if ckpt.get("opt_disc") is not None:
                self.dopt.load_state_dict(ckpt["opt_disc"], list(self.disc.parameters()))
            self.dfp.after_external_update()
        self.global_step = int(ckpt.get("steps", 0))
        _set_rng_state(ckpt.get("rng"))

    def read_log(self) -> Dict[str, float]:
        self.wait_optimizers()
        v = self.log.tolist()
        return {"L1": v[0], "L2": v[1], "LPIPS": v[2], "rec_loss": v[3], "vae_norm": v[4], "dmd_loss": v[5], "dmd_gradient_norm": v[6],
                "diffusion_loss": v[7], "sit_norm": v[8], "d_weight": v[9]}


class DiffusionTrainer:
    """Step of the downstream latent-diffusion trainer (train_diffusion.py:268-297): frozen `vae.encode` -> (tokens - latent_mean) * latent_scale ->
    [B, C, h, w] -> `transport.training_losses` (flow matching, velocity target) on LightningDiT -> clip_grad_norm_(1.0) -> AdamW(lr, betas (0.9, 0.95),
    weight_decay 0, constant lr; :204) -> EMA(0.9999; `update_ema`, :129-139).

    Device work: the frozen ViT-L encoder and the bottleneck on the HIP path (models/vit_fast.py), LightningDiT forward + backward through
    `functional.DitBlockFn` (models/lightningdit_fast.forward_train; the reference's `use_checkpoint` recomputation is a memory knob and not needed
    at 288 GB), gradient norm + clip + AdamW + EMA as two passes over flat buffers (`optim.FlatAdamWEMA`), DP gradient all-reduce over the flat
    gradient.  The EMA covers the trainable parameters; the reference also "updates" the fixed sin-cos `pos_embed` with itself (:137-139), which
    only perturbs it by rounding."""

    def __init__(self, model, vae: VAE, lr: float = 1e-4, latent_mean: float = 0.0, latent_scale: float = 1.0, max_norm: float = 1.0,
                 ema_decay: float = 0.9999, path_type: str = "Linear", prediction: str = "velocity", loss_weight=None, train_eps=0.0, sample_eps=0.0,
                 bucket_bytes: int = 64 << 20, overlap_optimizer: bool = False):
        from .transport import create_transport
        from .models.lightningdit import LightningDiT
        self.model, self.vae = model, vae
        self.latent_mean, self.latent_scale = latent_mean, latent_scale
        self.transport = create_transport(path_type, prediction, loss_weight, train_eps, sample_eps)          # train_diffusion.py:190-196: no time shift
        for p in vae.parameters():
            p.requires_grad_(False)
        vae.eval()
        params = [p for p in model.parameters() if p.requires_grad]
        self.fp = FlatParams(params, with_ema=True)
        self.fp.enable_bf16_shadow()
        self.fp.enable_transposed_shadow()
        direct = [p for n_, p in model.named_parameters() if p.requires_grad and n_.startswith("blocks.")]      # adaLN modulations included: see DMDTrainer
        if isinstance(model, LightningDiT) and direct:
            self.fp.enable_direct_grads(only=direct)
        self.opt = FlatAdamWEMA(self.fp, lr=lr, betas=(0.9, 0.95), weight_decay=0.0, warmup_steps=0, max_norm=max_norm, ema_decay=ema_decay)
        self.sync = dist.FlatGradSync(self.fp.params, self.fp.grad, self.fp.offsets, bucket_bytes=bucket_bytes)      # train_diffusion.py:215 (DDP)
        self.log = torch.zeros(2, dtype=torch.float32, device=self.fp.flat.device)
        self.train_steps = 0
        if dist.broadcast_module_state(self.model, self.vae, extra=[self.fp.flat, self.fp.ema, self.opt.exp_avg, self.opt.exp_avg_sq]):
            self.fp.after_external_update()
        if overlap_optimizer:
            self.opt.enable_overlap()             # the update (norm + AdamW + EMA + operand refresh: 23 GB) runs beside the NEXT step's frozen encoder forward

    def wait_optimizers(self) -> None:
        self.opt.wait()

    def latents(self, images: torch.Tensor) -> torch.Tensor:
        """train_diffusion.py:276-287."""
        from .sample import tokens_to_dit_input
        with torch.no_grad():
            tok = self.vae.encode(images)
        return tokens_to_dit_input(tok.float(), self.latent_mean, self.latent_scale)

    def step(self, images: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        self.model.train()                                                  # label dropout for classifier-free guidance (:232)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not parity.on()):
            x = self.latents(images)                                        # frozen encoder: does not touch the model's weights (an overlapped update may still run)
            self.fp.begin_step()                                            # waits for it
            _, terms = self.transport.training_losses(self.model, x, dict(y=labels))
        loss = terms["loss"].mean().float()
        loss.backward()
        self.sync.wait()

        def log(norm):
            with torch.no_grad():
                self.log[0], self.log[1] = loss.detach(), norm[0]
        self.opt.step(then=log)
        self.train_steps += 1
        return loss.detach()

    def ema_state_dict(self):
        """The `ema` entry of the reference's checkpoint (:316-323): the model's state_dict with the trainable parameters replaced by their averages."""
        self.opt.wait()
        sd = {k: v.clone() for k, v in self.model.state_dict().items()}
        names = {id(p): n for n, p in self.model.named_parameters()}
        for p, e in zip(self.fp.params, self.fp.ema_state()):
            sd[names[id(p)]] = e.clone()
        return sd

    def checkpoint(self) -> dict:
        """train_diffusion.py:318-325: model / ema state_dicts, opt (torch.optim.AdamW layout over model.parameters()), steps."""
        self.opt.wait()
        return {"model": {k: v.detach().clone() for k, v in self.model.state_dict().items()}, "ema": self.ema_state_dict(),
                "opt": self.opt.state_dict(list(self.model.parameters())), "steps": self.train_steps}

    def load(self, ckpt: dict) -> None:
        self.opt.wait()
        self.model.load_state_dict(ckpt["model"], strict=True)
        names = {id(p): n for n, p in self.model.named_parameters()}
        with torch.no_grad():
            for p, e in zip(self.fp.params, self.fp.ema_state()):
                e.copy_(ckpt["ema"][names[id(p)]])
        self.opt.load_state_dict(ckpt["opt"], list(self.model.parameters()))
        self.train_steps = int(ckpt.get("steps", 0))
        self.fp.after_external_update()

    def read_log(self) -> Dict[str, float]:
        self.opt.wait()
        v = self.log.tolist()
        return {"loss": v[0], "grad_norm": v[1]}


def build_tokenizer_trainer(device="cuda", z_channels=32, model_size="large", seed=42, lpips_ckpt=None, **kw) -> TokenizerTrainer:
    """Random-init model in the reference's constructor order under torch.manual_seed(seed)."""
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=z_channels, model_size=model_size).to(device)
    with warnings.catch_warnings():
        if lpips_ckpt is None:
            warnings.simplefilter("ignore")        # synthetic benchmark / tests: a random trunk is the stated configuration (no weights offline)
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


def _randomised_dit(device, seed_offset: int = 0):
    """LightningDiT-XL/1 at the DMD stage's shape (16 x 16 latent tokens of 32 channels, 1000 classes) with the reference's zero-initialised adaLN / output
    layers (lightningdit.py:367-376) drawn from N(0, 0.02) instead: with them at zero v == 0 and every DMD / flow-matching quantity is vacuous."""
    from .models.lightningdit import LightningDiT_models
    m = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).to(device)
    with torch.no_grad():
        for blk in m.blocks:
            blk.adaLN_modulation[1].weight.normal_(0, 0.02)
        m.final_layer.linear.weight.normal_(0, 0.02)
    return m


def build_dmd_trainer(device="cuda", seed=42, **kw) -> DMDTrainer:
    """Config C3 at its real size (train_dmd.py:506-575, scripts/train_dmd.sh:16-35): VAE(large, z = 32) with the ViT-L/16 encoder trainable, LPIPS, LightningDiT-XL/1
    as frozen teacher and trainable student, CFG 5, dmd_weight 10, the VAE on every 5th step -- random-init weights in the reference's constructor order under
    torch.manual_seed(seed) (what tests/test_gpu_fullsize.py::test_dmd_stage_full_size_cycle_c3 builds)."""
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="large").to(device)
        lp = LPIPS().eval().requires_grad_(False).to(device)
    with torch.no_grad():
        for lin in (lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4):
            lin.model[-1].weight.fill_(1.0 / lin.model[-1].weight.shape[1])
    teacher, student = _randomised_dit(device).eval().requires_grad_(False), _randomised_dit(device).eval()
    cfg = dict(dmd_weight=10.0, dmd_cfg_scale=5.0, num_classes=1000, vae_train_every=5, warmup_steps=10, lr=2e-5, diff_lr=2e-5)
    cfg.update(kw)
    return DMDTrainer(vae, lp, teacher, student, **cfg)


def build_diffusion_trainer(device="cuda", seed=0, **kw) -> DiffusionTrainer:
    """Config C4 (train_diffusion.py:268-297, scripts/train_diffusion.sh:19-20): LightningDiT-XL/1 on the frozen VAE(large, z = 32)'s latents, lr 2e-4."""
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="large").to(device).eval().requires_grad_(False)
    cfg = dict(lr=2e-4)
    cfg.update(kw)
    return DiffusionTrainer(_randomised_dit(device), vae, **cfg)
