"""Phase times of the SK instantiation of gemm_pp_kernel from s_memtime stamps (dmvae_debug_gemm_timing; [block][unit 0..3][8]): unit start, accumulators initialised,
K loop done, unit done; inside the hand-over of a PART: partial stores issued, stores acknowledged, arrival counted.  usage: time_gemm_sk.py M N K splits tile"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import _lib, ops
m, n, k, splits, tile = (int(v) for v in sys.argv[1:6])
L = ctypes.CDLL(_lib.LIB_PATH)
L.dmvae_debug_gemm_timing.argtypes = [ctypes.c_void_p]
x = torch.randn(m, k, device="cuda").to(torch.bfloat16); w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16); b = torch.randn(n, device="cuda").to(torch.bfloat16)
wk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32)
for _ in range(5): ops.linear_sk(x, wk, b, splits=splits, tile=tile)
buf = torch.zeros(256 * 4 * 8, dtype=torch.int64, device="cuda")
L.dmvae_debug_gemm_timing(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.linear_sk(x, wk, b, splits=splits, tile=tile); e1.record(); torch.cuda.synchronize()
L.dmvae_debug_gemm_timing(None)
t = buf.view(256, 4, 8).cpu().double()
print(f"{m}x{n}x{k} splits {splits} tile {tile}: call {e0.elapsed_time(e1) * 1e3:.1f} us (with the stamps' barrier); phases in k-ticks of s_memtime")
t0 = t[:, 0, 0][t[:, 0, 0] > 0].min()
for ui in range(4):
    sel = t[:, ui, 3] > 0
    if not sel.any(): break
    tt = t[sel][:, ui]
    part = tt[:, 4] > 0
    d = lambda a, b_, s_: ((tt[s_][:, b_] - tt[s_][:, a]).mean().item() * 1e-3) if s_.any() else float("nan")
    al = torch.ones_like(part)
    print(f"  unit {ui}: blocks {int(sel.sum())} (parts {int(part.sum())}) | start after launch {((tt[:, 0] - t0).mean().item()) * 1e-3:6.2f} | wait+init {d(0, 1, al):6.2f} | K loop {d(1, 2, al):6.2f} |"
          f" part: stores issued {d(2, 4, part):6.2f}, acknowledged {d(4, 5, part):6.2f}, counted {d(5, 6, part):6.2f}, rest (last arriver: sum + epilogue) {d(6, 3, part):6.2f} |"
          f" whole unit {d(0, 3, al):6.2f} | last end after launch {((tt[:, 3].max() - t0).item()) * 1e-3:6.2f}")
