"""Kernel mix of train.DiffusionTrainer's step (DiT-XL/1, B from env, default 64): run under rocprofv3 --kernel-trace --stats."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
from dmvae_amd.models.vae import VAE
from dmvae_amd.train import DiffusionTrainer
B = int(os.environ.get("B", "64")); N = int(os.environ.get("N", "5"))
torch.manual_seed(0)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vae = VAE(z_channels=32, model_size="large").cuda().eval().requires_grad_(False)
dit = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda()
with torch.no_grad():
    for blk in dit.blocks:
        blk.adaLN_modulation[1].weight.normal_(0, 0.02)
    dit.final_layer.linear.weight.normal_(0, 0.02)
tr = DiffusionTrainer(dit, vae, lr=2e-4)
images = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
labels = torch.randint(0, 1000, (B,), device="cuda")
for _ in range(N): tr.step(images, labels)
torch.cuda.synchronize()
