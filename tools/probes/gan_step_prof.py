"""Kernel mix of the steady-state tokenizer step (adversarial branch on): run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.train import build_tokenizer_trainer
N = int(os.environ.get("N", "6"))
tr = build_tokenizer_trainer(device="cuda", seed=42, with_disc=os.environ.get("DISC", "1") != "0", disc_start_step=0)
images = torch.rand(32, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(42)) * 2 - 1
for _ in range(N): tr.step(images)
torch.cuda.synchronize()
