#!/usr/bin/env python
"""How fast does the vendor GEMM (hipBLASLt via torch.matmul) run large bf16 problems on this part with random vs all-zero
operands, sustained over ~0.3 s?  Calibrates the power-limited MFMA ceiling the conv kernels are compared against."""
import torch
def timed(fn, reps):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
for m, n, k in [(8192, 8192, 8192), (16384, 8192, 4608), (524288, 512, 4608), (2097152, 128, 1152)]:
    for data in ("randn", "zeros"):
        a = torch.randn(m, k, device="cuda").to(torch.bfloat16) if data == "randn" else torch.zeros(m, k, device="cuda", dtype=torch.bfloat16)
        b = torch.randn(n, k, device="cuda").to(torch.bfloat16) if data == "randn" else torch.zeros(n, k, device="cuda", dtype=torch.bfloat16)
        fl = 2.0 * m * n * k
        t1 = timed(lambda: a @ b.t(), 3)
        reps = max(10, int(0.3 / t1))
        t = timed(lambda: a @ b.t(), reps)
        print(f"hipBLASLt bf16 {m}x{n}x{k} {data}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s  ({reps} launches back to back)", flush=True)
        del a, b
