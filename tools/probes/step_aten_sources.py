"""Where the ATen launches of the tokenizer step (configuration C2) come from: torch.profiler over one step, every aten op with device time, grouped by op and input shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from dmvae_amd.train import build_tokenizer_trainer
tr = build_tokenizer_trainer(device="cuda", seed=42, **({"with_disc": True, "disc_start_step": 0} if os.environ.get("GAN") else {}))
images = torch.rand(32, 3, 256, 256, device="cuda") * 2 - 1
for _ in range(4): tr.step(images)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    tr.step(images)
    torch.cuda.synchronize()
rows, tot = [], 0.0
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") and e.device_time_total > 0:
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:90]))
        tot += e.self_device_time_total
rows.sort(reverse=True)
print("ATen ops with device time: %.1f us per step in total" % tot)
for r in rows[:45]:
    print("%8.1f us %4d x %-22s %s" % r)
