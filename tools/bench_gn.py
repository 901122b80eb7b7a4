#!/usr/bin/env python
"""GroupNorm kernels: achieved HBM GB/s per launch at the decoder's shapes (B=32)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (n, h, w, c) in [(32, 256, 256, 128), (32, 128, 128, 256), (32, 64, 64, 512), (32, 32, 32, 512)]:
    x = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)
    da = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)
    dres = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)
    g, b = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    S = x.numel() * 2
    st = ops.groupnorm_stats(x)
    t1 = timed(lambda: ops.groupnorm_stats(x))
    t2 = timed(lambda: ops.groupnorm_apply(x, st, g, b, True))
    t3 = timed(lambda: ops.groupnorm_bwd(da, x, st, g, b, True, dres=dres))
    print(f"[{n},{h},{w},{c}] S={S/1e6:.0f} MB  stats {t1:7.1f} us {S/t1/1e6:5.2f} TB/s | apply {t2:7.1f} us {2*S/t2/1e6:5.2f} TB/s | bwd(partial+apply, dres) {t3:7.1f} us {(2*S+4*S)/t3/1e6:5.2f} TB/s")
