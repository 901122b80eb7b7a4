#!/usr/bin/env python
"""Trainable ViT-L/16 encoder forward + backward at B=32, 256x256 (the encoder's share of a train_dmd.py vae-turn): the HIP-kernel path
(vit_fast.trainable_forward_features) vs the stock PyTorch modules under autocast(bf16)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd.models.vit import DinoV2ViT
B = int(os.environ.get("B", "32"))
torch.manual_seed(0)
vit = DinoV2ViT(embed_dim=1024, depth=24, num_heads=16, patch_size=16, img_size=256).cuda()
with torch.no_grad():
    for blk in vit.blocks:
        blk.ls1.gamma.fill_(1.0); blk.ls2.gamma.fill_(1.0)
x = torch.randn(B, 3, 256, 256, device="cuda")
flop = 3 * B * 162e9
for name, fn in (("hip", vit.forward_features), ("stock autocast", vit.forward_features_stock)):
    def step():
        vit.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn(x)
        y.float().square().mean().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {dt*1e3:.1f} ms fwd+bwd ({flop/dt/1e12:.0f} TFLOP/s), peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    torch.cuda.reset_peak_memory_stats()
