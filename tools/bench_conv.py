#!/usr/bin/env python
"""Micro-benchmark of the conv kernels on the decoder's shapes (B=32): TFLOP/s per shape, fwd and wgrad."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd import ops

SHAPES = [  # (name, n, h, w, cin, cout, ks, ups)
    ("512>512@32", 32, 32, 32, 512, 512, 3, 0),
    ("512>512@64", 32, 64, 64, 512, 512, 3, 0),
    ("512>512@128", 32, 128, 128, 512, 512, 3, 0),
    ("256>256@128", 32, 128, 128, 256, 256, 3, 0),
    ("256>256@256", 32, 256, 256, 256, 256, 3, 0),
    ("128>128@256", 32, 256, 256, 128, 128, 3, 0),
    ("512>512@64ups", 32, 64, 64, 512, 512, 3, 1),
    ("512>256@128_1x1", 32, 128, 128, 512, 256, 1, 0),
]
which = sys.argv[1] if len(sys.argv) > 1 else "fwd,wgrad"
reps = int(os.environ.get("REPS", "20"))
DATA = os.environ.get("DATA", "randn")
for name, n, h, w, cin, cout, ks, ups in SHAPES:
    ho, wo = (2 * h, 2 * w) if ups else (h, w)
    x = (torch.zeros(n, h, w, cin, device="cuda") if DATA == "zeros" else torch.randn(n, h, w, cin, device="cuda")).to(torch.bfloat16)
    if DATA == "act": x = (x.float() * torch.sigmoid(x.float())).to(torch.bfloat16)
    wt = (torch.randn(cout, ks * ks, cin, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(cout, device="cuda")
    dy = torch.randn(n, ho, wo, cout, device="cuda").to(torch.bfloat16)
    flops = 2.0 * n * ho * wo * cout * cin * ks * ks
    out = f"{name:18s}"
    if "fwd" in which:
        for _ in range(2): ops.conv2d_nhwc(x, wt, b, ks=ks, upsample=bool(ups))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ops.conv2d_nhwc(x, wt, b, ks=ks, upsample=bool(ups))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out += f"  fwd {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s"
    if "res" in which:
        r = torch.randn(n, ho, wo, cout, device="cuda").to(torch.bfloat16)
        for _ in range(2): ops.conv2d_nhwc(x, wt, b, r, ks=ks, upsample=bool(ups))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ops.conv2d_nhwc(x, wt, b, r, ks=ks, upsample=bool(ups))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out += f"  fwd+residual {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s"
    if "wgrad" in which:
        for _ in range(2): ops.conv2d_nhwc_wgrad(dy, x, ks, upsample=bool(ups))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): ops.conv2d_nhwc_wgrad(dy, x, ks, upsample=bool(ups))
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out += f"  wgrad(+bias) {ms*1e3:8.1f} us {flops/ms/1e9:7.1f} TF/s"
    print(out, flush=True)
    del x, wt, dy
