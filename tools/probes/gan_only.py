import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmvae_amd.train import build_tokenizer_trainer
tr = build_tokenizer_trainer(device="cuda", seed=42, with_disc=True, disc_start_step=0)
images = torch.rand(32, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(42)) * 2 - 1
for _ in range(3): tr.step(images)
torch.cuda.synchronize()
for _ in range(4): tr.step(images)
torch.cuda.synchronize()
