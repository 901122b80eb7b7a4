"""Kernel mix of the LightningDiT-XL/1 no-grad forward at the sampler's batch (B=25): run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
B = int(os.environ.get("B", "25")); N = int(os.environ.get("N", "10"))
torch.manual_seed(0)
m = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda().eval().requires_grad_(False)
with torch.no_grad():
    for blk in m.blocks:
        blk.adaLN_modulation[1].weight.normal_(0, 0.02)
x = torch.randn(B, 32, 16, 16, device="cuda"); t = torch.rand(B, device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(N):
        m(x, t, y)
torch.cuda.synchronize()
