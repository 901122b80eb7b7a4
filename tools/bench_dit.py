#!/usr/bin/env python
"""LightningDiT-XL/1 inference forward at the DMD stage's shape (B=16 per GPU, 32x16x16 latents; train_dmd.py runs four of these per VAE turn):
HIP-kernel path (models/lightningdit_fast.py) vs the stock PyTorch modules under autocast(bf16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
B = int(os.environ.get("B", "16"))
torch.manual_seed(0)
m = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda().eval().requires_grad_(False)
with torch.no_grad():
    for blk in m.blocks:
        blk.adaLN_modulation[1].weight.normal_(0, 0.02)
    m.final_layer.linear.weight.normal_(0, 0.02)
x = torch.randn(B, 32, 16, 16, device="cuda"); t = torch.rand(B, device="cuda"); y = torch.randint(0, 1001, (B,), device="cuda")
flop = B * 237e9
for name, fn in (("hip", m), ("stock autocast", m.forward_stock)):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(3): fn(x, t, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): fn(x, t, y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"{name}: {dt*1e3:.2f} ms / forward ({flop/dt/1e12:.0f} TFLOP/s)", flush=True)
