"""How long does the one-launch repack take against the per-weight packs it replaces?  (C2 tokenizer trainer, after a few steps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd import functional as Fn, ops
from dmvae_amd.train import build_tokenizer_trainer
dev = torch.device("cuda", 0)
tr = build_tokenizer_trainer(device=dev, seed=42)
x = torch.rand(32, 3, 256, 256, device=dev) * 2 - 1
for _ in range(3): tr.step(x)
reg = tr.fp.pack_reg
print("entries", reg["n"], "elements", reg["total"])
def timed(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("batched repack: %.1f us" % timed(lambda: Fn.repack_all(reg, tr.fp.epoch)))
ents = list(reg["entries"].values())
def lazy():
    for (w, key, p, for_dgrad, subpixel, ptr) in ents:
        src = w.detach()
        if subpixel: src = ops.subpixel_weight(src)
        ops.pack_conv_weight(src, for_dgrad, p.shape[0], p.shape[2], kmajor=hasattr(p, "_dmvae_kmajor"))
print("per-weight packs: %.1f us" % timed(lazy))
import collections
c = collections.Counter((tuple(w.shape), for_dgrad, subpixel, hasattr(p, "_dmvae_kmajor")) for (w, key, p, for_dgrad, subpixel, ptr) in ents)
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:30]: print(v, k)
