#!/usr/bin/env python
"""Does running GroupNorm's two passes per batch CHUNK (so that the second pass finds the chunk in the 256 MB Infinity Cache) beat running each pass over
the whole batch?  stats -> apply and bwd(partial -> apply) at the decoder's shapes, whole batch vs chunks of 16 / 8 / 4 images."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops


def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (n, h, w, c) in [(32, 256, 256, 128), (32, 128, 128, 256), (32, 64, 64, 512)]:
    x = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)
    da = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)
    dres = torch.randn(n, h, w, c, device="cuda").to(torch.bfloat16)
    g, b = torch.randn(c, device="cuda"), torch.randn(c, device="cuda")
    other = torch.randn(256 << 20, device="cuda", dtype=torch.bfloat16)      # 512 MB of unrelated traffic between repetitions: cold cache each time
    st = ops.groupnorm_stats(x)
    out = f"[{n},{h},{w},{c}] {x.numel() * 2 / 1e6:.0f} MB:"
    for ch in (n, 16, 8, 4):
        def fwd():
            other.add_(1)
            for i in range(0, n, ch):
                s = ops.groupnorm_stats(x[i:i + ch])
                ops.groupnorm_apply(x[i:i + ch], s, g, b, True)
        def bwd():
            other.add_(1)
            for i in range(0, n, ch):
                ops.groupnorm_bwd(da[i:i + ch], x[i:i + ch], st[i:i + ch], g, b, True, dres=dres[i:i + ch])
        base = timed(lambda: other.add_(1))
        out += f"  chunk {ch:2d}: fwd {timed(fwd) - base:7.1f} us  bwd {timed(bwd) - base:7.1f} us |"
    print(out, flush=True)
