#!/usr/bin/env python
"""EXPERIMENT: the whole C2 tokenizer step -- forward, backward, clip, AdamW, EMA, operand re-packs -- captured ONCE in a hipGraph
(thread-local capture mode, static shapes, single rank) and replayed, against the same steps issued eagerly.  What it measures: how much of the step's
wall-minus-kernel time (0.9 ms in profiles/r4_end_step_trace_summary.txt) is launch overhead a graph removes.  NOT a production path: the optimiser's step count
and learning rate are kernel arguments and are baked into the captured launches (the replayed steps repeat one bias correction), so the weights diverge from the
eager run's after the first replay -- timing only.
    python tools/experiments/graph_step.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.train import build_tokenizer_trainer

B, N = 32, 20
tr = build_tokenizer_trainer(device="cuda", seed=42, warmup_steps=0)
images = torch.rand(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(42)) * 2 - 1


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    tr.step(images)
eager = [timed(lambda: tr.step(images), N) for _ in range(3)]
print("eager ms/step:", [round(v, 3) for v in eager], flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        tr.step(images)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        tr.step(images)
except Exception as e:      # noqa: BLE001
    print("capture failed:", type(e).__name__, str(e)[:600])
    sys.exit(0)
for _ in range(3):
    g.replay()
graphed = [timed(g.replay, N) for _ in range(3)]
print("graph replay ms/step:", [round(v, 3) for v in graphed], flush=True)
eager2 = [timed(lambda: tr.step(images), N) for _ in range(2)]
print("eager again ms/step:", [round(v, 3) for v in eager2])
print(f"median eager {sorted(eager + eager2)[2]:.3f}  median graph {sorted(graphed)[1]:.3f}  loss {tr.read_log()['rec_loss']:.5f}")
