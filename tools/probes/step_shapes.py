"""Per-shape conv / wgrad times inside one tokenizer step (HIP events around every call, dmvae_amd.ops.KERNEL_TIMING), sorted by time."""
import os, sys, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmvae_amd import ops
from dmvae_amd.train import build_tokenizer_trainer
tr = build_tokenizer_trainer(device="cuda", seed=42)
images = torch.rand(32, 3, 256, 256, device="cuda") * 2 - 1
for _ in range(3): tr.step(images)
torch.cuda.synchronize()
ops.KERNEL_TIMING = []
for _ in range(3): tr.step(images)
torch.cuda.synchronize()
t, ops.KERNEL_TIMING = ops.KERNEL_TIMING, None
agg = collections.OrderedDict()
for label, e0, e1, fl in t:
    a = agg.setdefault((label[-60:], fl), [0.0, 0])
    a[0] += e0.elapsed_time(e1); a[1] += 1
tot = 0
for (label, fl), (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{ms/3:7.3f} ms/step {n//3:3d}x {ms/n*1e3:8.1f} us {fl/1e9:8.1f} GF {fl/(ms/n)/1e9:7.1f} TF/s  {label}")
    tot += ms / 3
print("total", round(tot, 2))
