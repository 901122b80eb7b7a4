"""A/B of gemm_pp's two main loops (DMVAE_GEMM_LOOP: 0 ping-pong, 1 pipelined single stream) on forced tiles: correctness against the other loop's bits, then time."""
import ctypes, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import _lib, ops
L = ctypes.CDLL(_lib.LIB_PATH)
L.dmvae_debug_gemm_cfg.argtypes = [ctypes.c_int]; L.dmvae_debug_gemm_loop.argtypes = [ctypes.c_int]
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "4,7").split(",")]
LB = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # the loop compared with loop 0 (1: pipelined single stream, 2: the same with a barrier every second K step)
SHAPES = [("vit fc2", 8224, 1024, 4096), ("vit proj", 8224, 1024, 1024), ("vit qkv", 8224, 3072, 1024), ("vit fc1", 8224, 4096, 1024), ("dit16 w3", 4096, 1152, 3072),
          ("dit64 w12", 16384, 6144, 1152), ("sq 8192", 8192, 8192, 4096), ("ragged", 777, 520, 384)]
def once(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, m, n, k in SHAPES:
    nbuf = max(2, int(600e6 // (m * k * 2)) + 1)
    xs = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(nbuf)]
    w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
    wk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32)
    b = torch.randn(n, device="cuda").to(torch.bfloat16)
    line = f"{name:10s} {m}x{n}x{k}"
    for cfg in cfgs:
        L.dmvae_debug_gemm_cfg(cfg)
        L.dmvae_debug_gemm_loop(0); y0 = ops.linear_bf16(xs[0], wk, b, out_f32=True)
        L.dmvae_debug_gemm_loop(LB); y1 = ops.linear_bf16(xs[0], wk, b, out_f32=True); y1b = ops.linear_bf16(xs[0], wk, b, out_f32=True)
        torch.cuda.synchronize()
        ok = torch.equal(y0, y1) and torch.equal(y1, y1b)
        ts = {}
        for loop in (0, LB):
            ctr = [0]
            def f():
                ctr[0] = (ctr[0] + 1) % nbuf
                ops.linear_bf16(xs[ctr[0]], wk, b)
            L.dmvae_debug_gemm_loop(loop)
            for _ in range(3): f()
            ts[loop] = []
        for _ in range(5):
            for loop in (0, LB):
                L.dmvae_debug_gemm_loop(loop)
                ts[loop].append(once(f, 20))
        fl = 2.0 * m * n * k
        t0, t1 = statistics.median(ts[0]), statistics.median(ts[LB])
        line += f" | cfg{cfg} {'OK ' if ok else 'BAD'} loop0 {t0:6.1f}us {fl/t0/1e6:5.0f}TF  loopB {t1:6.1f}us {fl/t1/1e6:5.0f}TF"
    print(line, flush=True)
L.dmvae_debug_gemm_cfg(-1); L.dmvae_debug_gemm_loop(0)
