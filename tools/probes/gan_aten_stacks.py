"""Where the small ATen launches / device-to-device copies of the ADVERSARIAL tokenizer step come from: torch.profiler with Python stacks over one step; every aten op
with device time grouped by (op, shapes, the innermost dmvae_amd frame)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from dmvae_amd.train import build_tokenizer_trainer
tr = build_tokenizer_trainer(device="cuda", seed=42, with_disc=True, disc_start_step=0)
images = torch.rand(32, 3, 256, 256, device="cuda") * 2 - 1
for _ in range(4): tr.step(images)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(images)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0.0, 0])
for e in prof.events():
    if not e.name.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    site = "?"
    for fr in (e.stack or []):
        if "dmvae_amd" in fr and "site-packages" not in fr:
            site = fr.split("dmvae_amd/")[-1][:70]
            break
    k = (e.name, str(e.input_shapes)[:60], site)
    agg[k][0] += e.self_device_time_total; agg[k][1] += 1
rows = sorted(((v[0], v[1]) + k for k, v in agg.items()), reverse=True)
print("ATen ops with device time: %.1f us, %d launches per step" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
bysite = collections.defaultdict(lambda: [0.0, 0])
for t, c, name, shp, site in rows:
    bysite[site][0] += t; bysite[site][1] += c
for site, (t, c) in sorted(bysite.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.1f us %4d launches  %s" % (t, c, site))
print("---- top ops")
for r in rows[:40]:
    print("%8.1f us %4d x %-20s %-60s %s" % r)
