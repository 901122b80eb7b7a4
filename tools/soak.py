#!/usr/bin/env python
"""Soak run: N steps of the tokenizer step (optionally with the discriminator branch from step D) on a rotating set of synthetic batches;
reports loss trajectory, finiteness and peak memory at intervals (no growth expected after the first steps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd.train import build_tokenizer_trainer
N = int(os.environ.get("STEPS", "150")); D = int(os.environ.get("DISC_START", "100")); B = int(os.environ.get("B", "16"))
tr = build_tokenizer_trainer(device="cuda", seed=42, with_disc=True, disc_start_step=D, warmup_steps=20)
gen = torch.Generator(device="cuda").manual_seed(0)
batches = [torch.rand(B, 3, 256, 256, device="cuda", generator=gen) * 2 - 1 for _ in range(4)]
t0 = time.time()
for it in range(N):
    loss = tr.step(batches[it % 4])
    if it % 25 == 24 or it == N - 1:
        log = tr.read_log()
        dlog = tr.read_disc_log() if it >= D else {}
        ok = all(v == v and abs(v) < 1e6 for v in list(log.values()) + list(dlog.values()))
        print(f"step {it+1}: rec {log['rec_loss']:.4f} L1 {log['L1']:.4f} LPIPS {log['LPIPS']:.4f} |g| {log['vae_norm']:.3f} d_w {log['d_weight']:.3f} "
              f"{'d_loss %.3f acc %.1f' % (dlog['d_loss'], dlog['acc_mean']) if dlog else ''} finite={ok} peak {torch.cuda.max_memory_allocated()/2**30:.1f} GiB "
              f"{(time.time()-t0)/(it+1)*1e3:.0f} ms/step", flush=True)
        assert ok
