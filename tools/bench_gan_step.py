#!/usr/bin/env python
"""Steady-state tokenizer step (global_step >= disc_start_step: generator GAN term + discriminator update) vs the warm-up-phase step that
bench.py reports; B=32, same synthetic batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd.train import build_tokenizer_trainer
B = int(os.environ.get("B", "32"))
for with_disc in (False, True):
    tr = build_tokenizer_trainer(device="cuda", seed=42, with_disc=with_disc, disc_start_step=0) if with_disc else build_tokenizer_trainer(device="cuda", seed=42)
    images = torch.rand(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(42)) * 2 - 1
    for _ in range(3): tr.step(images)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): tr.step(images)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"with_disc={with_disc}: {dt*1e3:.1f} ms/step, {B/dt:.1f} img/s", tr.read_log(), tr.read_disc_log() if with_disc else "", flush=True)
    del tr
    torch.cuda.empty_cache()
