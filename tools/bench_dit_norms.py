#!/usr/bin/env python
"""HBM-bound passes of the LightningDiT block (csrc/dit.hip, dit_stack.hip) at the DMD stage's and the diffusion trainer's shapes: microseconds per call and achieved
TB/s on the algorithmic bytes (every tensor once).  Cold inputs: the calls rotate over enough buffers to defeat the 256-MB Infinity Cache.
    python tools/bench_dit_norms.py            DMVAE_RM8=0 python tools/bench_dit_norms.py   (the four-channel norm kernels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops
BF = torch.bfloat16
C, N, H = 1152, 256, 16


def t(fns, reps=6):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3


for B in (16, 64):
    nb = max(2, int(700e6 // (B * N * C * 4)) + 1)
    hs = [torch.randn(B, N, C, device="cuda") for _ in range(nb)]
    rs = [torch.randn(B, N, C, device="cuda").to(BF) for _ in range(nb)]
    w = torch.rand(C, device="cuda") + 0.5
    mod = (0.3 * torch.randn(B, 6 * C, device="cuda")).to(BF)
    MB = B * N * C / 1e6
    us = t([lambda h=h: ops.rmsnorm_modulate(h, w, mod, 0, C) for h in hs])
    print(f"B={B:3d} rmsnorm_modulate            {us:7.1f} us  {MB * 6 / us / 1e6 * 1e6 / 1e6:5.2f} TB/s")
    us = t([lambda h=h, r=r: ops.gated_residual_out(h, r, mod, 2 * C, w, mod, 3 * C, 4 * C) for h, r in zip(hs, rs)])
    print(f"B={B:3d} gated_residual_out + norm   {us:7.1f} us  {MB * 12 / us:5.2f} TB/s")
    S = ops.DitStackBwd(2, B, N, C, H, torch.device("cuda"))
    das = rs
    us = t([lambda h=h, da=da, x=x, y=y: S.boundary(1, h, da=da, x=x, w=w, mod=mod, scale_off=4 * C, y=y, gate_mod=mod, gate_off=2 * C)
            for h, da, x, y in zip(hs, das, hs[::-1], rs[::-1])])
    print(f"B={B:3d} boundary (rowstat + apply+gate) {us:7.1f} us  {MB * (2 + 4 + 2 + 4 + 4 + 2 + 2 + 4) / us:5.2f} TB/s (incl. the row-statistics pass)")
    x12s = [torch.randn(B * N, 6144, device="cuda").to(BF) for _ in range(max(2, nb // 2))]
    dhs = [torch.randn(B * N, 3072, device="cuda").to(BF) for _ in range(len(x12s))]
    us = t([lambda a=a: ops.swiglu(a) for a in x12s])
    print(f"B={B:3d} swiglu                      {us:7.1f} us  {B * N * 3072 * 6 / 1e6 / us:5.2f} TB/s")
    us = t([lambda a=a, d=d: ops.swiglu_bwd(d, a) for a, d in zip(x12s, dhs)])
    print(f"B={B:3d} swiglu_bwd                  {us:7.1f} us  {B * N * 3072 * 10 / 1e6 / us:5.2f} TB/s")
