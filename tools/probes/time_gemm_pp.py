"""Per-tile phase times of gemm_pp_kernel from s_memtime stamps (dmvae_debug_gemm_timing): per block and tile [tile start, operands landed + accumulators
initialised, K loop done, stores issued].  usage: python tools/probes/time_gemm_pp.py M N K [cfg]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import _lib, ops
m, n, k = (int(v) for v in sys.argv[1:4])
cfg = int(sys.argv[4]) if len(sys.argv) > 4 else -1
L = ctypes.CDLL(_lib.LIB_PATH)
L.dmvae_debug_gemm_cfg.argtypes = [ctypes.c_int]; L.dmvae_debug_gemm_timing.argtypes = [ctypes.c_void_p]
x = torch.randn(m, k, device="cuda").to(torch.bfloat16); w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16); b = torch.randn(n, device="cuda").to(torch.bfloat16)
L.dmvae_debug_gemm_cfg(cfg)
for _ in range(5): ops.linear_bf16(x, w, b)
buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device="cuda")
L.dmvae_debug_gemm_timing(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.linear_bf16(x, w, b); e1.record(); torch.cuda.synchronize()
L.dmvae_debug_gemm_timing(None)
t = buf.view(256, 4, 8).cpu().double()
print(f"{m}x{n}x{k} cfg {cfg} plan {ops.linear_plan(m, n, k)}: call {e0.elapsed_time(e1) * 1e3:.1f} us (phases in k-cycles of s_memtime)")
live = t[:, 0, 3] > 0
t = t * 1.0
for ti in range(4):
    sel = live & (t[:, ti, 3] > 0)
    if not sel.any(): break
    tt = t[sel][:, ti]
    d = lambda a, b_: (tt[:, b_] - tt[:, a]).mean().item() * 1e-3
    print(f"  tile {ti}: blocks {int(sel.sum())} | wait+init {d(0, 1):6.2f} | K loop {d(1, 2):6.2f} | epilogue (all waves) {d(2, 3):6.2f} k-cycles")
