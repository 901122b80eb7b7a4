#!/usr/bin/env python
"""Driver for rocprofv3 over the DMD stage (C3) or the latent-diffusion step (C4): runs warm steps, then `CYCLES` cycles with a MARKER launch
(`sde_euler_kernel` on 64 elements: a kernel neither stage uses) before every step, so tools/stage_trace_summary.py can cut the kernel trace into steps.
    STAGE=dmd|diffusion|gan  CYCLES=2  python tools/prof_stage.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops
from dmvae_amd.train import build_dmd_trainer, build_diffusion_trainer

STAGE = os.environ.get("STAGE", "dmd")
CYCLES = int(os.environ.get("CYCLES", "2"))
mx = torch.zeros(64, device="cuda")


def marker():
    ops.sde_euler_step(mx, mx, None, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)


if STAGE == "dmd":
    B = int(os.environ.get("B", "16"))
    tr = build_dmd_trainer()
    images = torch.rand(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 2 - 1
    labels = torch.randint(0, 1000, (B,), device="cuda")
    for _ in range(6):
        tr.step(images, labels)
    torch.cuda.synchronize()
    for _ in range(5 * CYCLES):
        marker()
        tr.step(images, labels)
    marker()
elif STAGE == "gan":
    from dmvae_amd.train import build_tokenizer_trainer
    B = int(os.environ.get("B", "32"))
    tr = build_tokenizer_trainer(device="cuda", seed=42, with_disc=True, disc_start_step=0)
    images = torch.rand(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(42)) * 2 - 1
    for _ in range(3):
        tr.step(images)
    torch.cuda.synchronize()
    for _ in range(3 * CYCLES):
        marker()
        tr.step(images)
    marker()
else:
    B = int(os.environ.get("B", "64"))
    tr = build_diffusion_trainer()
    images = torch.rand(B, 3, 256, 256, device="cuda") * 2 - 1
    labels = torch.randint(0, 1000, (B,), device="cuda")
    for _ in range(3):
        tr.step(images, labels)
    torch.cuda.synchronize()
    for _ in range(3 * CYCLES):
        marker()
        tr.step(images, labels)
    marker()
torch.cuda.synchronize()
print(tr.read_log())
