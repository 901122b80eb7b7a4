"""Soak of the DMD stage (config C3 shapes, B = 16): N steps on rotating batches; losses finite, memory flat, no hang.  STEPS (default 120)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
from dmvae_amd.models.vae import VAE
from dmvae_amd.train import DMDTrainer
from dmvae_amd.utils.lpips import LPIPS
N = int(os.environ.get("STEPS", "120")); B = 16
torch.manual_seed(42)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vae = VAE(z_channels=32, model_size="large").cuda()
    lp = LPIPS().eval().requires_grad_(False).cuda()
mk = lambda: LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda()
teacher, student = mk().eval().requires_grad_(False), mk().eval()
with torch.no_grad():
    for m in (teacher, student):
        for blk in m.blocks:
            blk.adaLN_modulation[1].weight.normal_(0, 0.02)
        m.final_layer.linear.weight.normal_(0, 0.02)
tr = DMDTrainer(vae, lp, teacher, student, dmd_weight=5.0, dmd_cfg_scale=5.0, num_classes=1000, vae_train_every=5, warmup_steps=10)
g = torch.Generator(device="cuda").manual_seed(1)
batches = [(torch.rand(B, 3, 256, 256, device="cuda", generator=g) * 2 - 1, torch.randint(0, 1000, (B,), device="cuda", generator=g)) for _ in range(4)]
t0 = time.time()
for it in range(N):
    x, y = batches[it % 4]
    tr.step(x, y)
    if it % 20 == 19 or it == N - 1:
        log = tr.read_log()
        ok = all(v == v and abs(v) < 1e6 for v in log.values())
        print(f"step {it+1}: " + " ".join(f"{k} {v:.4f}" for k, v in log.items() if k in ("rec_loss", "dmd_loss", "diffusion_loss", "vae_norm", "sit_norm")) +
              f" finite={ok} tables {__import__('dmvae_amd.ops', fromlist=['x']).TABLE_BUILDS[0]} peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB {(time.time()-t0)/(it+1)*1e3:.0f} ms/step", flush=True)
        assert ok
