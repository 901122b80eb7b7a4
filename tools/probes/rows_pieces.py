"""Per-piece time of one adaLN Linear (16 x 1152 -> 6912) forward + backward: this build's kernels vs the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import ops
from dmvae_amd.functional import packed, _bf
BF = torch.bfloat16
def bench(fn, n=200):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (16, 32):
    N, K = 6912, 1152
    ws = [torch.randn(N, K, device="cuda") * 0.02 for _ in range(24)]          # rotate over 24 weights: cold operands, like 28 blocks in a row
    wb = [w.to(BF) for w in ws]
    wt = [w.t().contiguous().to(BF) for w in ws]
    x = torch.randn(M, K, device="cuda").to(BF); dy = torch.randn(M, N, device="cuda").to(BF); b = torch.randn(N, device="cuda")
    i = [0]
    def nxt():
        i[0] = (i[0] + 1) % 24; return i[0]
    print(f"M={M}")
    print("  fwd  linear_rows      %6.1f us" % bench(lambda: ops.linear_rows(x, wb[nxt()], b)))
    print("  fwd  F.linear         %6.1f us" % bench(lambda: torch.nn.functional.linear(x, wb[nxt()], b.to(BF))))
    print("  dx   linear_rows(W^T) %6.1f us" % bench(lambda: ops.linear_rows(dy, wt[nxt()])))
    print("  dx   dy @ W           %6.1f us" % bench(lambda: dy @ wb[nxt()]))
    print("  dW   conv wgrad       %6.1f us" % bench(lambda: ops.conv2d_nhwc_wgrad(dy.view(1, 1, M, N), x.view(1, 1, M, K), 1)))
    print("  dW   dy^T @ x (bf16)  %6.1f us" % bench(lambda: dy.t() @ x))
    print("  pack W^T from f32     %6.1f us" % bench(lambda: ops.pack_conv_weight(ws[nxt()], True)))
    print("  pack kmajor-t from bf %6.1f us" % bench(lambda: ops.linear_weight_t_kmajor(wb[nxt()])))
