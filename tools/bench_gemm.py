#!/usr/bin/env python
"""ViT-L GEMM shapes (tokens = 32 x 257): conv_pp as a 1x1 conv vs torch F.linear (hipBLASLt), TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dmvae_amd import ops
M = 32 * 257
def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, k, n in [("qkv", 1024, 3072), ("proj", 1024, 1024), ("fc1", 1024, 4096), ("fc2", 4096, 1024)]:
    x = torch.randn(M, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(n, device="cuda")
    bb = b.to(torch.bfloat16)
    fl = 2.0 * M * k * n
    t_mine = timed(lambda: ops.conv2d_nhwc(x.view(1, 1, M, k), w.view(n, 1, k), b, ks=1))
    t_blas = timed(lambda: F.linear(x, w, bb))
    print(f"{name:5s} M={M} K={k} N={n}: conv_pp {t_mine:7.1f} us {fl/t_mine/1e6:6.1f} TF/s | hipBLASLt {t_blas:7.1f} us {fl/t_blas/1e6:6.1f} TF/s")
