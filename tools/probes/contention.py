#!/usr/bin/env python
"""conv_pp under CU contention on ONE GPU: a side stream keeps `OCC` workgroups resident (each holding LDS, so a persistent conv block cannot share
its CU) while the main stream runs the decoder's large convolutions -- the situation an overlapped RCCL all-reduce creates on a multi-GPU run.
Run once with DMVAE_PP_DYNAMIC=0 and once with =1 (the flag is read at the first launch)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops, _lib
L = _lib.lib()
L.dmvae_debug_occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
shapes = [(32, 128, 128, 256, 256), (32, 64, 64, 512, 512), (32, 256, 256, 128, 128)]
side = torch.cuda.Stream()
for occ in (0, 16, 32, 64):
    res = []
    for n, h, w, cin, cout in shapes:
        x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
        wt = (torch.randn(cout, 9, cin, device="cuda") * 0.02).to(torch.bfloat16)
        b = torch.randn(cout, device="cuda")
        for _ in range(3): ops.conv2d_nhwc(x, wt, b, ks=3)
        torch.cuda.synchronize()
        if occ:
            L.dmvae_debug_occupy(occ, 64 * 1024, 40000, side.cuda_stream)     # resident for 40 ms
            time.sleep(0.005)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv2d_nhwc(x, wt, b, ks=3)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    wres = []
    for n, h, w, cin, cout in shapes:                       # the same layers' weight gradients (csrc/conv_wgrad_pp.hip: one round of ~252 blocks by default)
        a_ = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
        dy = torch.randn(n, h, w, cout, device="cuda").to(torch.bfloat16)
        for _ in range(3): ops.conv2d_nhwc_wgrad(dy, a_, 3)
        torch.cuda.synchronize()
        if occ:
            L.dmvae_debug_occupy(occ, 64 * 1024, 40000, side.cuda_stream)
            time.sleep(0.005)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv2d_nhwc_wgrad(dy, a_, 3)
        e1.record(); torch.cuda.synchronize()
        wres.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"DYNAMIC={os.environ.get('DMVAE_PP_DYNAMIC', '0')} MIN_ROUNDS={os.environ.get('DMVAE_WGRAD_PP_MIN_ROUNDS', '1')} occupied CUs {occ:3d}: fwd " +
          "  ".join(f"{s[3]}>{s[4]}@{s[1]}: {t:7.1f}" for s, t in zip(shapes, res)) + " | wgrad " + "  ".join(f"{t:7.1f}" for t in wres) + " us")
