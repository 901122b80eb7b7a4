"""Which hipBLASLt kernels F.linear picks for the M = 4096 Linear shapes (run under rocprofv3 --kernel-trace --stats): the macro-tile (MT), split-K (GSU) and
stream-K (SK) fields of the Tensile kernel names say how the vendor library decomposes the few-tile deep-K problems."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import torch, torch.nn.functional as F
import gemm_select
gemm_select.enable()
SHAPES = [("w3", 4096, 1152, 3072), ("d_qkv", 4096, 1152, 3456), ("d_w12", 4096, 1152, 6144), ("fc2_16", 4112, 1024, 4096), ("w12", 4096, 6144, 1152), ("qkv", 4096, 3456, 1152)]
for name, m, n, k in SHAPES:
    x = torch.randn(m, k, device="cuda").bfloat16(); w = (torch.randn(n, k, device="cuda") * 0.02).bfloat16(); b = torch.randn(n, device="cuda").bfloat16()
    torch.cuda.synchronize()
    marker = torch.zeros(1 + len(name), device="cuda")      # a tiny ATen fill of a distinct size separates the shapes in the trace order
    for _ in range(5):
        F.linear(x, w, b)
    torch.cuda.synchronize()
    print(name, m, n, k, flush=True)
