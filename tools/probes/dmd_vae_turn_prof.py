import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
from dmvae_amd.models.vae import VAE
from dmvae_amd.train import DMDTrainer
from dmvae_amd.utils.lpips import LPIPS
B = 16
torch.manual_seed(42)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vae = VAE(z_channels=32, model_size="large").cuda()
lp = LPIPS().eval().requires_grad_(False).cuda()
mk = lambda: LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda()
teacher, student = mk().eval().requires_grad_(False), mk().eval()
tr = DMDTrainer(vae, lp, teacher, student, vae_train_every=1, warmup_steps=10)      # every step is a VAE turn (+ the student turn)
images = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
labels = torch.randint(0, 1000, (B,), device="cuda")
for _ in range(3): tr.step(images, labels)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(4): tr.step(images, labels)
torch.cuda.synchronize(); print(f"vae turn + student turn: {(time.perf_counter()-t0)/4*1e3:.1f} ms", flush=True)
