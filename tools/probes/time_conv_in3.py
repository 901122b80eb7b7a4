"""LPIPS first layer at the C2 shape (64 x 256 x 256 -> 64 channels): ops.conv_in3 against the zero-padded 32-channel route.  usage: python tools/probes/time_conv_in3.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dmvae_amd import ops                     # noqa: E402
from dmvae_amd.functional import packed      # noqa: E402

dev = "cuda"
g = torch.Generator().manual_seed(0)
a = (torch.rand(32, 3, 256, 256, generator=g) * 2 - 1).to(dev)
b = (torch.rand(32, 3, 256, 256, generator=g) * 2 - 1).to(dev)
w = (0.2 * torch.randn(64, 3, 3, 3, generator=g)).to(dev)
bias = (0.1 * torch.randn(64, generator=g)).to(dev)
sh, sc = torch.tensor([-.030, -.088, -.188], device=dev), torch.tensor([.458, .448, .450], device=dev)
scratch = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def padded():
    x = torch.cat([(a - sh.view(1, 3, 1, 1)) / sc.view(1, 3, 1, 1), (b - sh.view(1, 3, 1, 1)) / sc.view(1, 3, 1, 1)], 0).contiguous()
    return ops.conv2d_nhwc(ops.nchw_to_nhwc_bf16(x, c_pad=32), packed(w, False, 0, 32), bias, ks=3, act=ops.ACT_RELU)


def fused():
    return ops.conv_in3(a, b, w, bias, sh, sc, act=ops.ACT_RELU)


def timeit(f, reps=10):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        scratch.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


if "--fused-only" not in sys.argv:
    print("padded 32-channel route: median %.1f us (min %.1f)" % timeit(padded))
print("conv_in3               : median %.1f us (min %.1f)" % timeit(fused))
