#!/usr/bin/env python
"""Linear-layer GEMM shapes of the transformer blocks (ViT-L at batch 32: tokens = 32 x 257; LightningDiT-XL/1 at batch 16 / 64): the hand-written GEMM
(csrc/gemm_pp.hip, `ops.linear_bf16`) vs the vendor library under this build's tuned solution table (F.linear -> hipBLASLt, tools/gemm_select.py + tools/tuned/) vs
conv_pp as a 1x1 conv.  Random-normal operands, interleaved rounds in one process, median microseconds per call and TFLOP/s.
  python tools/bench_gemm.py            the planned tile per shape
  python tools/bench_gemm.py --sweep    every tile of the menu per shape (calibrates csrc/gemm_pp.hip::g_cfg's cost column)
  python tools/bench_gemm.py --json F   also write the rows to F"""
import argparse, ctypes, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dmvae_amd import _lib, ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gemm_select

ap = argparse.ArgumentParser()
ap.add_argument("--sweep", action="store_true")
ap.add_argument("--json", default="")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--shapes", default="")
ap.add_argument("--kmajor", action="store_true", help="with --sweep: the weights K-tile-major (what frozen weights are served as)")
ap.add_argument("--sk", action="store_true", help="also the stream-K (sk0) / fused split-K (sk2, sk3, sk4) instantiation where it takes the shape")
ap.add_argument("--cold", action="store_true", help="rotate over enough distinct activation buffers (> 512 MB) that no call finds its input in the 256-MB Infinity Cache -- "
                "inside a training step a GEMM's input was written by the previous kernel, not read by the same GEMM 50 us earlier")
args = ap.parse_args()
gemm_select.enable()
L = _lib.lib()
dbg = ctypes.CDLL(_lib.LIB_PATH).dmvae_debug_gemm_cfg
dbg.argtypes, dbg.restype = [ctypes.c_int], None

SHAPES = [("vit qkv", 8224, 3072, 1024), ("vit proj", 8224, 1024, 1024), ("vit fc1", 8224, 4096, 1024), ("vit fc2", 8224, 1024, 4096),
          ("patch", 8192, 1024, 768),
          ("dit16 qkv", 4096, 3456, 1152), ("dit16 proj", 4096, 1152, 1152), ("dit16 w12", 4096, 6144, 1152), ("dit16 w3", 4096, 1152, 3072),
          ("dit64 qkv", 16384, 3456, 1152), ("dit64 proj", 16384, 1152, 1152), ("dit64 w12", 16384, 6144, 1152), ("dit64 w3", 16384, 1152, 3072),
          ("dit32 qkv", 8192, 3456, 1152), ("dit32 w12", 8192, 6144, 1152),
          # round 5: the DMD stage's batch-16 problems incl. the input-gradient shapes (N = in features, K = out features) and the ViT-L encoder at 16 x 257 tokens
          ("dit16 d_qkv", 4096, 1152, 3456), ("dit16 d_w12", 4096, 1152, 6144), ("dit16 d_w3", 4096, 3072, 1152),
          ("dit64 d_qkv", 16384, 1152, 3456), ("dit64 d_w12", 16384, 1152, 6144), ("dit64 d_w3", 16384, 3072, 1152),
          ("vit16 qkv", 4112, 3072, 1024), ("vit16 proj", 4112, 1024, 1024), ("vit16 fc1", 4112, 4096, 1024), ("vit16 fc2", 4112, 1024, 4096)]
if args.shapes:
    SHAPES = [s for s in SHAPES if any(t in s[0] for t in args.shapes.split(","))]
NCFG = 12


def once(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rows = []
for name, m, n, k in SHAPES:
    nbuf = max(2, int(600e6 // (m * k * 2)) + 1) if args.cold else 1
    xs = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(nbuf)]
    x = xs[0]
    ctr = [0]

    def nx():
        ctr[0] = (ctr[0] + 1) % nbuf
        return xs[ctr[0]]
    w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
    b = torch.randn(n, device="cuda")
    bb = b.to(torch.bfloat16)
    fl = 2.0 * m * k * n
    arms = {"hipblaslt": lambda: F.linear(nx(), w, bb)}
    if n >= 64 and k % 32 == 0 and m >= 16384:
        arms["conv_pp"] = lambda: ops.conv2d_nhwc(x.view(1, 1, m, k), w.view(n, 1, k), b, ks=1)
    plan_idx, tc, tr = ops.linear_plan(m, n, k)

    wsw = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32) if (args.kmajor and k % 32 == 0) else w

    def mk(cfg):
        def f():
            dbg(cfg)
            ops.linear_bf16(nx(), wsw, bb)
        return f
    if args.sweep:
        for c in range(NCFG):
            arms["gemm_pp[%d]" % c] = mk(c)
    else:
        arms["gemm_pp"] = mk(-1)
        if k % 32 == 0:
            wk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32)     # frozen weights are served K-tile-major

            def fk():
                dbg(-1)
                ops.linear_bf16(nx(), wk, bb)
            arms["gemm_pp_kmajor"] = fk
    if args.sk:
        wsk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32) if k % 32 == 0 else w      # K-tile-major, as the training route serves them
        for tl in (0, 1):
            for sp in (0, 2, 3):
                if ops.linear_sk_supported(m, n, k, sp, tl):
                    arms["sk%d%s" % (sp, "b" if tl else "a")] = (lambda sp_, tl_: (lambda: ops.linear_sk(nx(), wsk, bb, splits=sp_, tile=tl_)))(sp, tl)
    for f in arms.values():
        for _ in range(3):
            f()
    t = {a: [] for a in arms}
    for _ in range(args.rounds):
        for a, f in arms.items():
            t[a].append(once(f, args.reps))
    dbg(-1)
    med = {a: statistics.median(v) for a, v in t.items()}
    line = f"{name:10s} M={m:5d} N={n:4d} K={k:4d} plan={plan_idx}({tc}x{tr}) | " + " | ".join(f"{a} {med[a]:6.1f}us {fl / med[a] / 1e6:6.0f}TF" for a in arms)
    print(line, flush=True)
    rows.append({"name": name, "M": m, "N": n, "K": k, "plan": plan_idx, "tile": [tc, tr], "us": med, "tflops": {a: fl / med[a] / 1e6 for a in arms}})
if args.json:
    json.dump(rows, open(args.json, "w"), indent=1)
