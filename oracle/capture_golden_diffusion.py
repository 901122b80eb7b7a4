"""C4 `diffusion_step_small`: two steps of the REFERENCE's latent-diffusion training loop (train_diffusion.py:268-297) run on the CPU
with the reference's own modules -- frozen `vae.encode` (models/vae.py:100-103) -> latent normalisation and the [B, T, C] -> [B, C, h, w] permutation (:279-286) ->
`transport.training_losses` (diffusion/transport/transport.py:119-164: x0 ~ N(0, I), t ~ U(0, 1), xt = t x1 + (1 - t) x0, loss = mean((v - (x1 - x0))^2)) with
LightningDiT in TRAIN mode (label drop-out for classifier-free guidance, lightningdit.py:156-163; activation checkpointing on, as train_diffusion.py:62,188 set
it) -> `clip_grad_norm_(1.0)` -> `torch.optim.AdamW(lr, betas (0.9, 0.95), weight_decay 0, fused=True)` (:209) -> `update_ema` (:137-147) --, recorded as losses,
gradient norms, per-tensor parameter / EMA checksums, the normalised latents, the complete first-step gradient and update of a few small tensors.

It pins the build's counterpart of that loop -- dmvae_amd.train.DiffusionTrainer on the HIP path (tests/test_gpu_sampler.py) and the oracle's restatement
(tests/test_oracle_sampler.py) -- the way `step_small` pins the tokenizer step.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_diffusion.py            (CPU, under a minute)            -> tests/golden/diffusion_step_small.npz

The random draws of a step are the reference's own calls on the CPU generator, seeded per step (DRAW_SEED + step): `th.randn_like(x1)`, `th.rand((B,))` and the
time shift (transport.py:111-115), then `torch.rand(B) < class_dropout_prob` inside LabelEmbedder.token_drop.  `Transport.sample` and `token_drop` are wrapped
for the duration of the step ONLY to record what they returned (t after the shift, x0, which labels were dropped): x1 is a strided view here, `randn_like`
fills it in memory order, so the values cannot be re-derived by drawing `randn(shape)` from the same seed.  A replay on another generator (the GPU's) injects
the recorded values.
Models: the reduced ViT stand-in at embed 256 (4 heads x 64: inside the bf16 encoder kernels' range) with the bottleneck MLP, and a LightningDiT of hidden 192 =
3 heads x 64 (SwiGLU width 512), depth 2, 16 x 16 latent tokens of 32 channels, 10 classes -- inside the HIP DiT kernels' range.  fp32, no autocast (the
reference's autocast is CUDA-only); weights are the name-seeded deterministic fill of oracle/detweights.py (the reference zero-initialises the adaLN and output
layers, which would make the step vacuous)."""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.capture_golden import install_stubs, save  # noqa: E402
from oracle.detweights import det_fill_  # noqa: E402

SEED_VAE, SEED_DIT, SEED_IMG, DRAW_SEED = 71, 75, 76, 9000
STEPS, BATCH, LR = 2, 4, 1e-4
LATENT_MEAN, LATENT_SCALE = 0.05, 0.8
DIT_KW = dict(input_size=16, patch_size=1, in_channels=32, hidden_size=192, depth=2, num_heads=3, num_classes=10, class_dropout_prob=0.25, use_checkpoint=True)
SMALL = ("final_layer.linear.weight", "final_layer.linear.bias", "blocks.1.attn.q_norm.weight", "blocks.0.norm1.weight", "blocks.0.mlp.w3.bias",
         "t_embedder.mlp.2.bias", "x_embedder.proj.weight", "y_embedder.embedding_table.weight", "blocks.1.adaLN_modulation.1.bias")


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
    vae = vae.eval()                                                             # train_diffusion.py:225-226
    for p_ in vae.parameters():
        p_.requires_grad = False

    from diffusion.lightningdit.lightningdit import LightningDiT
    from diffusion.transport import create_transport
    import train_diffusion
    model = LightningDiT(**DIT_KW)
    det_fill_(model, SEED_DIT, skip=("pos_embed",))
    ema = copy.deepcopy(model)                                                   # :193
    for p_ in ema.parameters():
        p_.requires_grad_(False)
    transport = create_transport("Linear", "velocity", None, None, None)        # :196-202 with the script's defaults
    try:
        opt = torch.optim.AdamW(model.parameters(), lr=LR, betas=(0.9, 0.95), weight_decay=0, fused=True)      # :209
        fused = True
    except RuntimeError:                                                         # a torch build without the fused CPU kernel: the same update, unfused
        opt = torch.optim.AdamW(model.parameters(), lr=LR, betas=(0.9, 0.95), weight_decay=0)
        fused = False
    train_diffusion.update_ema(ema, model, decay=0)                              # :230
    model.train()                                                                # :231
    ema.eval()

    images = torch.rand(BATCH, 3, 256, 256, generator=torch.Generator().manual_seed(SEED_IMG)) * 2 - 1
    labels = torch.tensor([3, 7, 1, 9])
    names = [n_ for n_, _ in model.named_parameters()]
    p0 = {n_: p_.detach().clone() for n_, p_ in model.named_parameters()}
    out = {"vae_seed": np.array(SEED_VAE), "dit_seed": np.array(SEED_DIT), "images_seed": np.array(SEED_IMG), "draw_seed": np.array(DRAW_SEED), "batch": np.array(BATCH),
           "lr": np.array(LR), "latent_mean": np.array(LATENT_MEAN), "latent_scale": np.array(LATENT_SCALE), "labels": labels.numpy(), "names": np.array(names),
           "fused_adamw": np.array(int(fused)), "class_dropout_prob": np.array(DIT_KW["class_dropout_prob"])}
    losses, norms = [], []
    for step in range(STEPS):
        torch.manual_seed(DRAW_SEED + step)
        drawn = {}
        orig_sample, orig_drop = transport.sample, model.y_embedder.token_drop

        def sample_rec(x1):                      # transport.py:105-116 itself, its results recorded
            t_, x0_, x1_ = orig_sample(x1)
            drawn["t"], drawn["x0"] = t_.clone(), x0_.clone()
            return t_, x0_, x1_

        def drop_rec(lab, force_drop_ids=None):  # lightningdit.py:156-163 itself, its result recorded
            res = orig_drop(lab, force_drop_ids)
            drawn["dropped"] = res != lab
            return res
        transport.sample, model.y_embedder.token_drop = sample_rec, drop_rec
        # ---- train_diffusion.py:276-297, verbatim order (autocast(cuda) is a no-op on the CPU) ----
        with torch.no_grad():
            x = vae.encode(images)
        x = (x - LATENT_MEAN) * LATENT_SCALE
        c = x.shape[-1]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        p = 1
        x = x.reshape(shape=(x.shape[0], h, w, p, p, c))
        x = torch.einsum("nhwpqc->nchpwq", x)
        x = x.reshape(shape=(x.shape[0], c, h * p, h * p))
        model_kwargs = dict(y=labels)
        t_used, loss_dict = transport.training_losses(model, x, model_kwargs)
        transport.sample, model.y_embedder.token_drop = orig_sample, orig_drop
        assert torch.equal(t_used, drawn["t"])
        out[f"x0_{step}"], out[f"t_{step}"], out[f"drop_{step}"] = drawn["x0"].contiguous().numpy(), drawn["t"].numpy(), drawn["dropped"].numpy()
        loss = loss_dict["loss"].mean().float()
        opt.zero_grad()
        loss.backward()
        if step == 0:
            out["latents"] = x.detach().numpy()
            out["pred0_slice"] = loss_dict["pred"].detach()[:, ::4, ::4, ::4].numpy()
            out["loss_per_sample0"] = loss_dict["loss"].detach().numpy()
            for n_, p_ in model.named_parameters():
                if p_.grad is None:              # pos_embed: requires_grad False (lightningdit.py:318)
                    continue
                g_ = p_.grad.double()
                out["gn0." + n_] = np.array([g_.norm().item(), g_.sum().item()])
            for n_ in SMALL:
                out["g0." + n_] = dict(model.named_parameters())[n_].grad.numpy().copy()
        grad_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        train_diffusion.update_ema(ema, model)
        losses.append(loss.item())
        norms.append(grad_norm.item())
        cur, cur_ema = dict(model.named_parameters()), dict(ema.named_parameters())
        out[f"ck{step}"] = np.stack([stats(cur[n_], p0[n_]) for n_ in names])
        out[f"ema_ck{step}"] = np.stack([stats(cur_ema[n_], p0[n_]) for n_ in names])
        for n_ in SMALL:
            out[f"d{step}." + n_] = (cur[n_].detach() - p0[n_]).numpy()
            out[f"dema{step}." + n_] = (cur_ema[n_].detach() - p0[n_]).numpy()
        print(f"step {step}: loss {loss.item():.6f} |g| {grad_norm.item():.5f} dropped {drawn['dropped'].tolist()}", flush=True)
    out["loss"], out["grad_norm"] = np.array(losses, dtype=np.float64), np.array(norms, dtype=np.float64)
    save("diffusion_step_small", **out)


if __name__ == "__main__":
    main()
