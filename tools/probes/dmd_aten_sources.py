"""Where the ATen fill / copy / add launches of a DMD student-only step come from: torch.profiler over one step, grouped by (kernel-launching op, input shapes, python source line)."""
import os, sys, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from torch.profiler import profile, ProfilerActivity
from dmvae_amd.models.lightningdit import LightningDiT_models
from dmvae_amd.models.vae import VAE
from dmvae_amd.train import DMDTrainer
from dmvae_amd.utils.lpips import LPIPS
torch.manual_seed(42)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vae = VAE(z_channels=32, model_size="large").cuda()
    lp = LPIPS().eval().requires_grad_(False).cuda()
mk = lambda: LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda()
teacher, student = mk().eval().requires_grad_(False), mk().eval()
tr = DMDTrainer(vae, lp, teacher, student, dmd_weight=5.0, dmd_cfg_scale=5.0, num_classes=1000, vae_train_every=5, warmup_steps=10)
images = torch.rand(16, 3, 256, 256, device="cuda") * 2 - 1
labels = torch.randint(0, 1000, (16,), device="cuda")
for _ in range(int(os.environ.get("WARM", "7"))): tr.step(images, labels)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    tr.step(images, labels)          # a student-only step
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    if e.key.startswith("aten::") and e.self_device_time_total > 0:
        st = [s for s in e.stack if "dmvae_amd" in s or "tools/" in s][:3]
        rows.append((e.self_device_time_total, e.count, e.key, str(e.input_shapes)[:70], " <- ".join(s.split("/root/repo/")[-1][:60] for s in st)))
rows.sort(reverse=True)
for r in rows[:40]:
    print("%8.1f us %4d x %-14s %-70s %s" % r)
