"""Decoder tail backward at the C2 shape (B = 32, 256 x 256 x 128): the stored-operand route (gradient pack -> input-gradient conv -> GroupNorm backward) against
ops.norm_conv_out_bwd (csrc/groupnorm.hip::convout_bwd_kernel).  usage: python tools/probes/time_norm_conv_out.py [B]"""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dmvae_amd import ops                     # noqa: E402
from dmvae_amd.functional import packed      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda"
g = torch.Generator().manual_seed(0)
x = torch.randn(B, 256, 256, 128, generator=g).to(dev).to(torch.bfloat16)
gamma, beta = (1 + 0.1 * torch.randn(128, generator=g)).to(dev), (0.1 * torch.randn(128, generator=g)).to(dev)
cw = (0.05 * torch.randn(3, 128, 3, 3, generator=g)).to(dev)
dy = torch.randn(B, 3, 256, 256, generator=g).to(dev)
st = ops.groupnorm_stats(x)
scratch = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def stored():
    dyp = ops.nchw_to_nhwc_bf16(dy, c_pad=32)
    da = ops.conv2d_nhwc(dyp, packed(cw, True, cols_pad=32), ks=3)
    return ops.groupnorm_bwd(da, x, st, gamma, beta, True)


def fused():
    return ops.norm_conv_out_bwd(dy, cw, x, st, gamma, beta)


def timeit(f, reps=10):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        scratch.zero_()                      # cold caches, like the step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


a, b = stored(), fused()
print("dx rel diff %.2e  dgamma %.2e  dbeta %.2e" % tuple(((p.float() - q.float()).abs().max() / q.float().abs().max()).item() for p, q in zip(b, a)))
print("stored-operand route: median %.1f us (min %.1f)" % timeit(stored))
print("fused route         : median %.1f us (min %.1f)" % timeit(fused))
