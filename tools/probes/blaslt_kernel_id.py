#!/usr/bin/env python
"""Which hipBLASLt kernel serves the decoder's conv shapes as plain bf16 GEMMs?  Run under `rocprofv3 --kernel-trace`: the Tensile kernel name carries the
macro tile, MFMA shape, K depth, prefetch and LDS settings; the trace row carries grid, workgroup, LDS and register sizes."""
import torch
for m, n, k in [(524288, 512, 4608), (131072, 512, 4608), (524288, 256, 2304), (2097152, 128, 1152), (8224, 3072, 1024), (8224, 1024, 4096)]:
    a = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    b = torch.randn(n, k, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    del a, b, c
