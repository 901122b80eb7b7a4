#!/usr/bin/env python
"""The shapes bench.py's `kl_mmd` key quotes, as ONE command for rocprofv3 (kernel stats, then one --pmc pass per counter; tools/job_kl_evidence.sh):
  kl_pass_268MB   KL moment pass + its gradient alone on z [8192, 256, 32] f32 (268 MB): kl_moments_kernel, kl_final_kernel, kl_mmd_grad_kernel
  fused_G32       the fused KL + MMD call at the training shape (B = 32 images x 256 tokens x 32 latents against 256 prior samples each)
  fused_G1024     the same call at 1024 groups (32 x the work: the quadratic MMD dominates)
Prints wall microseconds per call (HIP events) and the algorithmic bytes of each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops
REPS = int(os.environ.get("REPS", "20"))
ONLY = [t for t in os.environ.get("SHAPES", "kl268,g32,g1024").split(",") if t]      # one shape per rocprofv3 run gives per-shape kernel statistics


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


z = torch.randn(8192 if "kl268" in ONLY else 1, 256, 32, device="cuda") * 0.7 + 0.2
nb = z.numel() * 4
for grad in ((False, True) if "kl268" in ONLY else ()):
    us = timed(lambda: ops.kl_mmd(z, None, need_grad=grad))
    alg = nb * (3 if grad else 1)            # moments: one read; gradient: one more read + one write
    print(f"kl_pass_268MB grad={int(grad)}: {us:8.1f} us/call, algorithmic {alg/1e6:.0f} MB -> {alg/us/1e6:.2f} TB/s = {alg/us/1e6/8:.3f} of 8 TB/s", flush=True)
del z
for g in [g_ for g_ in (32, 1024) if "g%d" % g_ in ONLY]:
    z = torch.randn(g, 256, 32, device="cuda") * 0.7 + 0.2
    y = torch.randn(g, 256, 32, device="cuda")
    us = timed(lambda: ops.kl_mmd(z, y, need_grad=True))
    alg = 3 * z.numel() * 4
    pairs = g * 3 * 256 * 256
    print(f"fused_G{g}: {us:8.1f} us/call, algorithmic {alg/1e6:.2f} MB -> {alg/us/1e6:.3f} TB/s = {alg/us/1e6/8:.4f} of 8 TB/s; {pairs/us/1e6:.3f} T kernel evaluations/s", flush=True)
