"""C3 `dmd_step_small`: four steps of the REFERENCE's DMD-stage loop (train_dmd.py:506-575) run on the CPU with the reference's own modules and its own
`VAELossFunction` (called unbound on a namespace, as oracle/capture_golden.py does for the tokenizer's): per step

  VAE turn (global_step % vae_train_every == 0; :516-545):  `vae.train(); requires_grad(vae, True)` -- the ViT encoder TRAINS in this stage --,
      `recon, z = vae(images, return_latent=True)`, latents = latents_to_spatial((z - latent_mean) * latent_scale), student frozen and in eval mode,
      `forward_generator(images, recon, latents, labels, compute_dmd=True)` = L1 + LPIPS + dmd_weight * compute_distribution_matching_loss (:204-230: xt from
      transport.sample / ICPlan.plan, four no-grad velocity evaluations -- teacher / student, conditional / null class --, CFG, the score-gradient surrogate),
      backward, clip_grad_norm_(1.0), AdamW(lr, wd, betas (0.9, 0.95)) over EVERY VAE parameter, LambdaLR;
  otherwise (:521-523): latents from no-grad encode;
  student turn (every step; :558-575): `requires_grad(sit, True); sit.train()`, transport.training_losses(sit, latents.detach(), dict(y=labels)) with label
      drop-out, backward, clip, AdamW(diff_lr, wd), LambdaLR.

Recorded: the log entries of every turn (L1, L2, LPIPS, rec_loss, dmd_loss, dmd_gradient_norm, vae_norm, diffusion_loss, sit_norm), the latents of step 0, the
gradient norm + sum of EVERY VAE parameter at the first VAE turn and of every student parameter at step 0, a handful of gradients in full, per-tensor parameter
checksums after the last step -- and what the steps drew (`Transport.sample`'s (t, x0) for the DMD loss and for the student, the student's dropped labels),
recorded by wrapping those two calls (see oracle/capture_golden_diffusion.py for why the values cannot be re-derived from a seed).

It pins the build's counterpart of that loop -- dmvae_amd.train.DMDTrainer on the HIP path and oracle.ref_cpu.dmd_train_steps -- the way `step_small` pins C2 and
`diffusion_step_small` pins C4.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_dmd_step.py            (CPU, ~2 minutes)            -> tests/golden/dmd_step_small.npz

Models: the reduced ViT stand-in at embed 256 (4 heads x 64; inside the bf16 encoder kernels' range), the bottleneck MLP and the FULL-width flux decoder (as in
step_small_w256); teacher and student LightningDiT of hidden 192 = 3 heads x 64, depth 2, 16 x 16 latent tokens of 32 channels, 10 classes (inside the HIP DiT
kernels' range).  fp32, no autocast (CUDA-only in the reference); name-seeded deterministic weights (oracle/detweights.py).  The same two images every step."""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.capture_golden import REF, install_stubs, save  # noqa: E402
from oracle.detweights import det_fill_, det_tensor  # noqa: E402

SEED_VAE, SEED_VGG, SEED_IMG, SEED_TEACHER, SEED_STUDENT, DRAW_SEED = 71, 41, 72, 81, 82, 9100
STEPS, BATCH, VAE_EVERY, WARMUP = 4, 2, 2, 1
LR, DIFF_LR, WD = 2e-6, 1e-4, 0.005            # the VAE's rate as in capture_golden_step.py (name-seeded weights: 1e-4 throws the first Adam step into a chaotic regime)
LATENT_MEAN, LATENT_SCALE, CFG, DMD_WEIGHT = 0.05, 0.8, 2.0, 5.0
DIT_KW = dict(input_size=16, patch_size=1, in_channels=32, hidden_size=192, depth=2, num_heads=3, num_classes=10, class_dropout_prob=0.5)
SMALL_VAE = ("decoder.conv_out.weight", "decoder.norm_out.weight", "bottle_neck.mlp.2.bias", "bottle_neck.mlp.0.bias", "decoder.conv_in.1.bias",
             "encoder.model.vit.norm.weight", "encoder.model.vit.blocks.1.ls2.gamma", "encoder.model.vit.blocks.0.attn.proj.bias", "encoder.model.vit.cls_token")
SMALL_SIT = ("final_layer.linear.weight", "blocks.0.norm1.weight", "blocks.1.attn.q_norm.weight", "t_embedder.mlp.2.bias", "y_embedder.embedding_table.weight",
             "blocks.1.adaLN_modulation.1.bias")


def stats(t, t0):
    t, t0 = t.detach().double(), t0.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t - t0).sum().item(), (t - t0).abs().sum().item()])


def main():
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    install_stubs()
    os.environ["DMVAE_GOLDEN_VIT"] = "vit_w256"
    vae_mod = sys.modules["models.vae"]

    class TinyDINO(vae_mod.DINOEncoder):
        def __init__(self, model_size="base", patch_size=16, image_size=256):
            super().__init__(model_size, patch_size, image_size)
            self.dim = 256

    saved = vae_mod.DINOEncoder
    vae_mod.DINOEncoder = TinyDINO
    torch.manual_seed(SEED_VAE)
    vae = vae_mod.VAE(z_channels=32, model_size="base")
    vae_mod.DINOEncoder = saved
    det_fill_(vae, SEED_VAE)

    from diffusion.lightningdit.lightningdit import LightningDiT
    from diffusion.transport import create_transport
    from utils.lpips import LPIPS
    import train_dmd
    base_model, sit = LightningDiT(**DIT_KW), LightningDiT(**DIT_KW)
    det_fill_(base_model, SEED_TEACHER, skip=("pos_embed",))
    det_fill_(sit, SEED_STUDENT, skip=("pos_embed",))
    base_model.eval()
    train_dmd.requires_grad(base_model, False)                                   # build_models (:366-368)
    lp = LPIPS(ckpt_path=REF + "/ckpt_vae/vgg.pth").eval().requires_grad_(False)
    with torch.no_grad():
        for n_, p_ in lp.net.named_parameters():
            t_ = det_tensor("net." + n_, p_.shape, SEED_VGG)
            p_.copy_(t_ * (2.0 ** 0.5) if p_.dim() > 1 else t_ * 0.5)
    args = SimpleNamespace(t0=0.0, t1=1.0, dmd_cfg_scale=CFG, num_classes=DIT_KW["num_classes"], dmd_weight=DMD_WEIGHT, disc_start_step=10 ** 9,
                           latent_mean=LATENT_MEAN, latent_scale=LATENT_SCALE, vae_train_every=VAE_EVERY)
    transport = create_transport("Linear", "velocity", None, None, None)         # :489-495
    loss_fn = SimpleNamespace(args=args, lpips_loss=lp, l1=1.0, l2=0.0, lpips=1.0, disc_weight=0.0, dmd_weight=DMD_WEIGHT, transport=transport,
                              base_model=base_model, sit_wo_ddp=sit, sit_ddp=sit, vae_wo_ddp=vae)
    loss_fn.compute_distribution_matching_loss = types.MethodType(train_dmd.VAELossFunction.compute_distribution_matching_loss, loss_fn)

    optimizer_sit = torch.optim.AdamW(sit.parameters(), lr=DIFF_LR, weight_decay=WD, betas=(0.9, 0.95), eps=1e-8)      # :473-474
    optimizer_vae = torch.optim.AdamW(vae.parameters(), lr=LR, weight_decay=WD, betas=(0.9, 0.95), eps=1e-8)

    def lr_lambda(step):                                                          # :477-481
        if step < WARMUP:
            return step / WARMUP
        return 1.0
    scheduler_sit = torch.optim.lr_scheduler.LambdaLR(optimizer_sit, lr_lambda)
    scheduler_vae = torch.optim.lr_scheduler.LambdaLR(optimizer_vae, lr_lambda)
    vae.eval()                                                                    # :500-501
    sit.eval()

    images = torch.rand(BATCH, 3, 256, 256, generator=torch.Generator().manual_seed(SEED_IMG)) * 2 - 1
    labels = torch.tensor([3, 7])
    vnames = [n_ for n_, _ in vae.named_parameters()]
    snames = [n_ for n_, p_ in sit.named_parameters() if p_.requires_grad]
    v0 = {n_: p_.detach().clone() for n_, p_ in vae.named_parameters()}
    s0 = {n_: p_.detach().clone() for n_, p_ in sit.named_parameters()}
    out = {"vae_seed": np.array(SEED_VAE), "vgg_seed": np.array(SEED_VGG), "images_seed": np.array(SEED_IMG), "teacher_seed": np.array(SEED_TEACHER),
           "student_seed": np.array(SEED_STUDENT), "batch": np.array(BATCH), "labels": labels.numpy(), "lr": np.array(LR), "diff_lr": np.array(DIFF_LR), "wd": np.array(WD),
           "warmup_steps": np.array(WARMUP), "vae_train_every": np.array(VAE_EVERY), "cfg": np.array(CFG), "dmd_weight": np.array(DMD_WEIGHT),
           "latent_mean": np.array(LATENT_MEAN), "latent_scale": np.array(LATENT_SCALE), "class_dropout_prob": np.array(DIT_KW["class_dropout_prob"]),
           "vae_names": np.array(vnames), "student_names": np.array(snames)}
    for k, v in lp.state_dict().items():
        if k.startswith("lin"):
            out["lp." + k] = v.numpy()
    drawn = []
    orig_sample, orig_drop = transport.sample, sit.y_embedder.token_drop

    def sample_rec(x1):
        t_, x0_, x1_ = orig_sample(x1)
        drawn.append(("sample", t_.clone(), x0_.contiguous().clone()))
        return t_, x0_, x1_

    def drop_rec(lab, force_drop_ids=None):
        res = orig_drop(lab, force_drop_ids)
        drawn.append(("drop", res != lab))
        return res
    transport.sample, sit.y_embedder.token_drop = sample_rec, drop_rec

    global_step = 0
    for step in range(STEPS):
        torch.manual_seed(DRAW_SEED + step)
        drawn.clear()
        vae_training_turn = (global_step % VAE_EVERY == 0)                         # :509
        # ---- train_dmd.py:516-526 ----
        if vae_training_turn:
            vae.train()
            train_dmd.requires_grad(vae, True)
            recon_image, z = vae(images, return_latent=True)
        else:
            with torch.no_grad():
                z = vae.encode(images)
                recon_image = vae.decode(z)
        latents = (z - LATENT_MEAN) * LATENT_SCALE
        latents = train_dmd.latents_to_spatial(latents)
        if vae_training_turn:                                                      # :529-545
            sit.eval()
            train_dmd.requires_grad(sit, False)
            vae_loss, vae_loss_dict = train_dmd.VAELossFunction.forward_generator(loss_fn, images, recon_image, latents, labels, compute_dmd=True, step=global_step)
            vae_loss = vae_loss.mean()
            vae_loss.backward()
            if step == 0:
                out["latents0"] = latents.detach().numpy()
                out["recon0_slice"] = recon_image.detach()[:, :, ::16, ::16].numpy()
                for n_, p_ in vae.named_parameters():
                    if p_.grad is None:           # the stand-in's mask_token: not on the forward path
                        continue
                    g_ = p_.grad.double()
                    out["vgn0." + n_] = np.array([g_.norm().item(), g_.sum().item()])
                for n_ in SMALL_VAE:
                    out["vg0." + n_] = dict(vae.named_parameters())[n_].grad.numpy().copy()
            g_norm = torch.nn.utils.clip_grad_norm_(vae.parameters(), max_norm=1.0)
            vae_loss_dict["vae_norm"] = g_norm.item()
            optimizer_vae.step()
            optimizer_vae.zero_grad(set_to_none=True)
            scheduler_vae.step()
            for k, v in vae_loss_dict.items():
                out[f"log{step}.{k}"] = np.array(v, dtype=np.float64)
        train_dmd.requires_grad(sit, True)                                         # :558-559
        sit.train()
        t, loss_dict = transport.training_losses(sit, latents.detach(), dict(y=labels))      # :563-571
        loss = loss_dict["loss"].mean()
        loss.backward()
        if step == 0:
            for n_ in snames:
                g_ = dict(sit.named_parameters())[n_].grad.double()
                out["sgn0." + n_] = np.array([g_.norm().item(), g_.sum().item()])
            for n_ in SMALL_SIT:
                out["sg0." + n_] = dict(sit.named_parameters())[n_].grad.numpy().copy()
        sit_norm = torch.nn.utils.clip_grad_norm_(sit.parameters(), max_norm=1.0)
        optimizer_sit.step()
        optimizer_sit.zero_grad(set_to_none=True)
        scheduler_sit.step()
        out[f"log{step}.diffusion_loss"], out[f"log{step}.sit_norm"] = np.array(loss.item(), dtype=np.float64), np.array(sit_norm.item(), dtype=np.float64)
        # what this step drew, in order: [DMD loss's sample (VAE turns)], the student's sample, the student's dropped labels
        samples = [d for d in drawn if d[0] == "sample"]
        drops = [d for d in drawn if d[0] == "drop"]
        assert len(samples) == (2 if vae_training_turn else 1) and len(drops) == 1
        if vae_training_turn:
            out[f"dmd_t_{step}"], out[f"dmd_x0_{step}"] = samples[0][1].numpy(), samples[0][2].numpy()
        out[f"sit_t_{step}"], out[f"sit_x0_{step}"], out[f"sit_drop_{step}"] = samples[-1][1].numpy(), samples[-1][2].numpy(), drops[0][1].numpy()
        print(f"step {step}: vae_turn {vae_training_turn} " + (" ".join(f"{k} {v:.5f}" for k, v in vae_loss_dict.items()) if vae_training_turn else "") +
              f" diffusion_loss {loss.item():.5f} sit_norm {sit_norm.item():.4f} dropped {drops[0][1].tolist()}", flush=True)
        global_step += 1
    cur_v, cur_s = dict(vae.named_parameters()), dict(sit.named_parameters())
    out["vck"] = np.stack([stats(cur_v[n_], v0[n_]) for n_ in vnames])
    out["sck"] = np.stack([stats(cur_s[n_], s0[n_]) for n_ in snames])
    for n_ in SMALL_VAE:
        out["vd." + n_] = (cur_v[n_].detach() - v0[n_]).numpy()
    for n_ in SMALL_SIT:
        out["sd." + n_] = (cur_s[n_].detach() - s0[n_]).numpy()
    save("dmd_step_small", **out)


if __name__ == "__main__":
    main()
