"""G12 `step_small`: four full tokenizer train steps of the REFERENCE's own modules on the CPU, recorded as a
loss trajectory + parameter / EMA checksums + the complete update of a few small tensors.  It pins the build's step harness
(dmvae_amd.train.TokenizerTrainer and the oracle's restatement of the loop) end to end: model forward, loss assembly,
backward, clip_grad_norm_, AdamW, LambdaLR warm-up, update_ema -- in the order of train_tokenizer.py:403-437.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_step.py            (CPU, ~2 minutes)            -> tests/golden/step_small.npz
      TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_step.py --width 256                             -> tests/golden/step_small_w256.npz
      (the reduced ViT stand-in at embed 256 = 4 heads x 64: the width from which the build's bf16 encoder KERNELS run, so that the bf16 step test
      takes the whole step -- encoder included -- through the HIP path; the width-64 capture needs stock modules for its encoder there)

What runs is the reference: models/vae.py::VAE (frozen encoder = the reduced ViT stand-in of capture_golden.py, bottleneck MLP, the
full-width flux decoder), utils/lpips.py::LPIPS, train_tokenizer.py::VAELossFunction.forward_generator (called unbound, as in
capture_golden.py), torch.nn.utils.clip_grad_norm_, torch.optim.AdamW(lr BASE_LR, wd 0.005, betas (0.9, 0.95), eps 1e-8),
LambdaLR(lr_lambda of train_tokenizer.py:385-389 with warmup_steps = 2), train_tokenizer.update_ema.  fp32, no autocast (the
reference's autocast is CUDA-only); weights are the name-seeded deterministic fill of oracle/detweights.py so that the tests
can rebuild them without shipping 50 M parameters.  The same two images are used for every step (as bench.py does).
"""
from __future__ import annotations

import copy
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.capture_golden import REF, install_stubs, save  # noqa: E402
from oracle.detweights import det_fill_, det_tensor  # noqa: E402

SEED_VAE, SEED_VGG, SEED_IMG = 71, 41, 72
STEPS, WARMUP, BATCH = 4, 2, 2
# The name-seeded fill has fan-in-scaled (not std 0.02) weights: at the script default 1e-4 the very first Adam update (|delta| = lr per element, 52 M
# coherent elements) throws the loss from 0.78 to 1.44, i.e. a chaotic trajectory that pins nothing.  lr is an argument of the reference (args.lr); 2e-6
# keeps the four steps in the smooth regime while every term of the optimiser tail stays exercised.
BASE_LR = 2e-6
SMALL = ("decoder.conv_out.weight", "decoder.conv_out.bias", "decoder.norm_out.weight", "decoder.norm_out.bias",
         "bottle_neck.mlp.2.weight", "bottle_neck.mlp.2.bias", "decoder.conv_in.1.weight", "decoder.mid.attn_1.norm.weight")


def stats(t, t0):
    t, t0 = t.detach().double(), t0.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t - t0).sum().item(), (t - t0).abs().sum().item()])


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=64, choices=(64, 256))
    width = ap.parse_args().width
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    install_stubs()
    os.environ["DMVAE_GOLDEN_VIT"] = "vit_tiny" if width == 64 else "vit_w256"
    vae_mod = sys.modules["models.vae"]

    class TinyDINO(vae_mod.DINOEncoder):
        def __init__(self, model_size="base", patch_size=16, image_size=256):
            super().__init__(model_size, patch_size, image_size)
            self.dim = width

    saved = vae_mod.DINOEncoder
    vae_mod.DINOEncoder = TinyDINO
    torch.manual_seed(SEED_VAE)
    vae = vae_mod.VAE(z_channels=32, model_size="base")
    vae_mod.DINOEncoder = saved
    det_fill_(vae, SEED_VAE)
    # build_models (train_tokenizer.py:295-297): the encoder is frozen and in eval mode
    vae.encoder.eval()
    for p_ in vae.encoder.parameters():
        p_.requires_grad_(False)

    from utils.lpips import LPIPS
    import train_tokenizer
    lp = LPIPS(ckpt_path=REF + "/ckpt_vae/vgg.pth").eval().requires_grad_(False)
    with torch.no_grad():
        for n_, p_ in lp.net.named_parameters():
            t_ = det_tensor("net." + n_, p_.shape, SEED_VGG)
            p_.copy_(t_ * (2.0 ** 0.5) if p_.dim() > 1 else t_ * 0.5)
    loss_ns = SimpleNamespace(lpips_loss=lp, l1=1.0, l2=0.0, lpips=1.0, disc_weight=0.5, args=SimpleNamespace(disc_start_step=5000))

    params_train = [p_ for p_ in vae.parameters() if p_.requires_grad]                       # train_tokenizer.py:381
    opt = torch.optim.AdamW(params_train, lr=BASE_LR, weight_decay=0.005, betas=(0.9, 0.95), eps=1e-8)   # :382

    def lr_lambda(step):                                                                      # :385-389
        if step < WARMUP:
            return step / WARMUP
        return 1.0
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda)
    ema = copy.deepcopy(vae)                                                                  # :397-399
    for p_ in ema.parameters():
        p_.requires_grad_(False)
    ema.eval()

    images = torch.rand(BATCH, 3, 256, 256, generator=torch.Generator().manual_seed(SEED_IMG)) * 2 - 1
    names = [n_ for n_, p_ in vae.named_parameters() if p_.requires_grad]
    p0 = {n_: p_.detach().clone() for n_, p_ in vae.named_parameters() if p_.requires_grad}
    out = {"images_seed": np.array(SEED_IMG), "vae_seed": np.array(SEED_VAE), "vgg_seed": np.array(SEED_VGG), "warmup_steps": np.array(WARMUP),
           "batch": np.array(BATCH), "base_lr": np.array(BASE_LR), "names": np.array(names)}
    if width != 64:
        out["width"] = np.array(width)          # the default capture keeps exactly the committed step_small.npz's keys (it regenerates byte for byte)
    for k, v in lp.state_dict().items():
        if k.startswith("lin"):
            out["lp." + k] = v.numpy()
    logs = {k: [] for k in ("L1", "L2", "LPIPS", "rec_loss", "vae_norm", "lr")}
    for step in range(STEPS):
        logs["lr"].append(opt.param_groups[0]["lr"])
        recon = vae(images, return_latent=False)                                              # :411 (DDP wrapper is the identity at world 1)
        loss, log = train_tokenizer.VAELossFunction.forward_generator(loss_ns, images, recon, step=step)
        loss = loss.mean()
        loss.backward()
        if step == 0:
            out["recon0_slice"] = recon.detach()[:, :, ::16, ::16].numpy()
            for n_ in SMALL:
                out["g0." + n_] = dict(vae.named_parameters())[n_].grad.numpy().copy()
        g_norm = torch.nn.utils.clip_grad_norm_(vae.parameters(), max_norm=1.0)               # :415
        opt.step()
        opt.zero_grad(set_to_none=True)
        sched.step()
        train_tokenizer.update_ema(ema, vae)                                                  # :437
        for k in ("L1", "L2", "LPIPS", "rec_loss"):
            logs[k].append(log[k])
        logs["vae_norm"].append(g_norm.item())
        cur, cur_ema = dict(vae.named_parameters()), dict(ema.named_parameters())
        out[f"ck{step}"] = np.stack([stats(cur[n_], p0[n_]) for n_ in names])
        out[f"ema_ck{step}"] = np.stack([stats(cur_ema[n_], p0[n_]) for n_ in names])
        for n_ in SMALL:          # the complete update of a few small tensors after every step
            out[f"d{step}." + n_] = (cur[n_].detach() - p0[n_]).numpy()
            out[f"dema{step}." + n_] = (cur_ema[n_].detach() - p0[n_]).numpy()
        print(f"step {step}: lr {logs['lr'][-1]:.2e} rec {log['rec_loss']:.6f} L1 {log['L1']:.6f} LPIPS {log['LPIPS']:.6f} |g| {g_norm.item():.4f}", flush=True)
    # update_ema walks EVERY parameter (train_tokenizer.py:147-150), the frozen encoder included: its EMA copy is p*0.9999 + p*0.0001 in f32, i.e. p up to
    # one rounding per step (the reference's own TODO at :149) -- recorded so the tests can state that it is rounding and nothing else
    out["encoder_moved"] = np.array(max(float((a - b).abs().max()) for a, b in zip(vae.encoder.parameters(), ema.encoder.parameters())))
    for k, v in logs.items():
        out[k] = np.array(v, dtype=np.float64)
    save("step_small" if width == 64 else "step_small_w%d" % width, **out)


if __name__ == "__main__":
    main()
