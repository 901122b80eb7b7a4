"""Which host-side ops issue the ~190 device-to-device copies and ~160 f32 -> bf16 conversion kernels per DiffusionTrainer step (torch.profiler, grouped by call stack)."""
import os, sys, warnings
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from torch.profiler import profile, ProfilerActivity
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffusion_step_prof.py")).read().split("for _ in range(N)")[0])
for _ in range(3): tr.step(images, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.step(images, labels)
    torch.cuda.synchronize()
import collections
agg = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::_to_copy", "aten::clone", "aten::contiguous") and e.device_time_total > 0:
        st = [s for s in (e.stack or []) if "dmvae_amd" in s or "torch/autograd" in s][:2]
        agg[(e.name, tuple(e.input_shapes[0]) if e.input_shapes else (), tuple(st))] += 1
for (name, shape, st), n in agg.most_common(25):
    print(n, name, shape, " <- ".join(s.split("/")[-1][:70] for s in st))
