#!/usr/bin/env python
"""Effective shader clock under the conv kernel: s_memtime span of XCD 0 (conv_pp's debug stamps) / event-timed duration,
for random and all-zero operands (diagnostics; run with DMVAE_CONV_HP=0 so that conv_pp takes the shapes)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops, _lib
L = _lib.lib()
L.dmvae_debug_timing.argtypes = [ctypes.c_void_p]
SHAPES = [("512>512@128", 32, 128, 128, 512, 512, 3), ("256>256@256", 32, 256, 256, 256, 256, 3), ("128>128@256", 32, 256, 256, 128, 128, 3)]
for name, n, h, w, cin, cout, ks in SHAPES:
    for data in ("randn", "zeros"):
        x = (torch.randn(n, h, w, cin, device="cuda") if data == "randn" else torch.zeros(n, h, w, cin, device="cuda")).to(torch.bfloat16)
        wt = (torch.randn(cout, ks * ks, cin, device="cuda") * 0.02).to(torch.bfloat16)
        if data == "zeros": wt.zero_()
        b = torch.randn(cout, device="cuda")
        for _ in range(10): ops.conv2d_nhwc(x, wt, b, ks=ks)
        buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        L.dmvae_debug_timing(buf.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv2d_nhwc(x, wt, b, ks=ks); e1.record(); torch.cuda.synchronize()
        L.dmvae_debug_timing(None)
        ms = e0.elapsed_time(e1)
        t = buf.view(-1, 8).cpu().double()
        t = t[(t[:, 0] != 0)]
        tx = t[0::8]
        span = tx[:, 5].max() - tx[:, 0].min()
        d = t[:, 1:6] - t[:, 0:5]
        nK = 9 * cin // 32
        flops = 2.0 * n * h * w * cout * cin * 9
        print(f"{name} {data}: {ms*1e3:.0f} us ({flops/ms/1e9:.0f} TF/s), XCD0 span {span:.0f} ticks -> {span/ms/1e6:.2f} GHz; loop {d[:,2].mean()/nK:.0f} ticks/Ktile, epilogue {d[:,3].mean():.0f}, first-tile {d[:,1].mean():.0f}, setup {d[:,0].mean():.0f}, drain {d[:,4].mean():.0f}")
