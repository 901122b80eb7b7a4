"""Soak of the latent-diffusion stage (config C4 shapes, B = 64): N steps on rotating batches; losses finite and falling, memory flat, no hang.  STEPS (default 150)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
from dmvae_amd.models.vae import VAE
from dmvae_amd.train import DiffusionTrainer
N = int(os.environ.get("STEPS", "150")); B = 64
torch.manual_seed(42)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vae = VAE(z_channels=32, model_size="large").cuda().eval()
dit = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda()
with torch.no_grad():
    for blk in dit.blocks:
        blk.adaLN_modulation[1].weight.normal_(0, 0.02)
    dit.final_layer.linear.weight.normal_(0, 0.02)
tr = DiffusionTrainer(dit, vae, lr=1e-4, latent_mean=0.0685, latent_scale=0.1763)
g = torch.Generator(device="cuda").manual_seed(1)
batches = [(torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1, torch.randint(0, 1000, (B,), device="cuda", generator=g)) for _ in range(4)]
t0 = time.time()
peak0 = None
for it in range(N):
    x, y = batches[it % 4]
    tr.step(x, y)
    if it % 25 == 24 or it == N - 1:
        log = tr.read_log()
        ok = all(v == v and abs(v) < 1e6 for v in log.values())
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        peak0 = peak if peak0 is None else peak0
        print(f"step {it+1}: " + " ".join(f"{k} {v:.4f}" for k, v in log.items()) + f" finite={ok} peak {peak:.1f} GiB {(time.time()-t0)/(it+1)*1e3:.0f} ms/step", flush=True)
        assert ok and peak <= peak0 + 0.1
