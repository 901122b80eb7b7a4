#!/usr/bin/env python
"""Per-block phase timing of conv_pp_kernel from s_memtime stamps (diagnostics).  Stamps are compared within an XCD only
(dispatch index % 8), since the shader clocks of different XCDs are not aligned."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops, _lib
L = _lib.lib()
DB = False
L.dmvae_debug_timing.argtypes = [ctypes.c_void_p]

set_dbg = L.dmvae_debug_timing
SHAPES = [("512>512@32", 32, 32, 32, 512, 512, 3), ("128>128@256", 32, 256, 256, 128, 128, 3), ("256>256@128", 32, 128, 128, 256, 256, 3)]
for name, n, h, w, cin, cout, ks in SHAPES:
    x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
    wt = (torch.randn(cout, ks * ks, cin, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(cout, device="cuda")
    for _ in range(2): ops.conv2d_nhwc(x, wt, b, ks=ks)
    buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
    set_dbg(buf.data_ptr())
    ops.conv2d_nhwc(x, wt, b, ks=ks)
    torch.cuda.synchronize()
    set_dbg(None)
    t = buf.view(-1, 8).cpu().double()
    nb = int((t[:, 0] != 0).sum())
    t = t[:nb]
    d = t[:, 1:6] - t[:, 0:5]
    print(f"{name}: blocks {nb}; mean cycles: setup {d[:,0].mean():.0f} first-tile {d[:,1].mean():.0f} loop {d[:,2].mean():.0f} "
          f"epilogue {d[:,3].mean():.0f} (min {d[:,3].min():.0f} max {d[:,3].max():.0f}) drain {d[:,4].mean():.0f} | block {(t[:,5]-t[:,0]).mean():.0f}")
    print(f"   epilogue split: loop end -> ring free {(t[:,6]-t[:,3]).mean():.0f}, bias + next setup + 2 DMA issues {(t[:,7]-t[:,6]).mean():.0f}, staging + stores {(t[:,4]-t[:,7]).mean():.0f}")
    xcd = 0
    tx = t[xcd::8]
    t0 = tx[:, 0].min()
    order = tx[:, 0].argsort()
    tx = tx[order]
    span = tx[:, 5].max() - t0
    print(f"   XCD0: {tx.shape[0]} blocks, span {span:.0f} cycles; sum of block time / (32 CUs x span) = {((tx[:,5]-tx[:,0]).sum() / (32 * span)):.3f}")
    k = min(tx.shape[0], 40)
    print("   first starts:", [int(v) for v in (tx[:k, 0] - t0).tolist()])
    print("   their epilogue durations:", [int(v) for v in (tx[:k, 4] - tx[:k, 3]).tolist()])
    if tx.shape[0] > 64:
        print("   starts 32..71:", [int(v) for v in (tx[32:72, 0] - t0).tolist()])
    # persistent mode: tiles of block b are b, b+256, ...: gap between consecutive tiles of the same block
    if nb > 512:
        tb = t[0:nb:256][:8]
        print("   block 0 tile starts:", [int(v) for v in (tb[:, 0] - tb[0, 0]).tolist()], " per-tile phases:", [[int(v) for v in (r[1:6] - r[0:5]).tolist()] for r in tb[:3]])
