#!/usr/bin/env python
"""Micro-benchmark of dmvae_kl_mmd: wall time per call (HIP events) at the real shape and at large-batch shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops
for (g, n, m, grad) in [(32, 256, 256, True), (32, 256, 256, False), (1024, 256, 256, True), (4096, 32, 32, True), (4096, 256, 16, False)]:
    z = torch.randn(g, n, 32, device="cuda") * 0.7 + 0.2
    y = torch.randn(g, m, 32, device="cuda")
    for _ in range(3): ops.kl_mmd(z, y, need_grad=grad)
    reps = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.kl_mmd(z, y, need_grad=grad)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    byt = (g * n + g * m + (g * n if grad else 0)) * 32 * 4
    pairs = g * (n * n + n * m + m * m)
    print(f"G={g} n={n} m={m} grad={grad}: {us:9.1f} us/call  compulsory {byt/1e6:8.2f} MB -> {byt/us/1e6:7.3f} TB/s   pairs {pairs/1e6:8.1f} M -> {pairs/us/1e6:7.3f} Tpair/s")
