#!/usr/bin/env python
"""Where a K tile's cycles go in conv_pp's main loop (library built with -DDMVAE_PP_TRACE, DMVAE_LIB=...): per tile, for wave 0 (first group) and wave 4 (second group):
cycles spent in the LOAD interval's own work, waiting at the barrier behind it, in the COMPUTE interval's MFMA issue, waiting at the barrier behind that."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops, _lib
L = _lib.lib()
L.dmvae_debug_timing.argtypes = [ctypes.c_void_p]
for name, n, h, w, cin, cout, ks in [("256>256@128", 32, 128, 128, 256, 256, 3), ("128>128@256", 32, 256, 256, 128, 128, 3), ("512>512@64", 32, 64, 64, 512, 512, 3)]:
    x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
    wt = (torch.randn(cout, ks * ks, cin, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(cout, device="cuda")
    for _ in range(2): ops.conv2d_nhwc(x, wt, b, ks=ks)
    buf = torch.zeros(1 << 21, dtype=torch.int64, device="cuda")
    L.dmvae_debug_timing(buf.data_ptr())
    ops.conv2d_nhwc(x, wt, b, ks=ks)
    torch.cuda.synchronize()
    L.dmvae_debug_timing(None)
    t = buf.view(-1, 16).cpu().double()
    t = t[(t[:, 8] != 0)]
    nk = ks * ks * cin // 32
    for wv, o in (("wave 0", 8), ("wave 4", 12)):
        m = t[:, o:o + 4].mean(0) / nk
        print(f"{name} {wv}: per K tile: LOAD work {m[0]:.0f}, wait behind LOAD {m[1]:.0f}, COMPUTE issue {m[2]:.0f}, wait behind COMPUTE {m[3]:.0f}; sum {m.sum():.0f}  ({len(t)} tiles, {nk} K tiles each)")
