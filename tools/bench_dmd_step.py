#!/usr/bin/env python
"""The DMD stage's step (train_dmd.py, config C3: B=16 per GPU, ViT-L encoder trainable, LightningDiT-XL/1 teacher and student) on
train.DMDTrainer: time per VAE-turn step and per student-only step."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
from dmvae_amd.models.vae import VAE
from dmvae_amd.train import DMDTrainer
from dmvae_amd.utils.lpips import LPIPS
B = int(os.environ.get("B", "16"))
torch.manual_seed(42)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    vae = VAE(z_channels=32, model_size="large").cuda()
lp = LPIPS().eval().requires_grad_(False).cuda()
with torch.no_grad():
    for lin in (lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4):
        lin.model[-1].weight.fill_(1.0 / lin.model[-1].weight.shape[1])
mk = lambda: LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda()
teacher, student = mk().eval().requires_grad_(False), mk().eval()
with torch.no_grad():
    for m in (teacher, student):
        for blk in m.blocks:
            blk.adaLN_modulation[1].weight.normal_(0, 0.02)
        m.final_layer.linear.weight.normal_(0, 0.02)
tr = DMDTrainer(vae, lp, teacher, student, dmd_weight=5.0, dmd_cfg_scale=5.0, num_classes=1000, vae_train_every=5, warmup_steps=10)
images = torch.rand(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 2 - 1
labels = torch.randint(0, 1000, (B,), device="cuda")
for _ in range(6): tr.step(images, labels)           # one VAE turn + five student-only steps: warm
times = {"vae_turn": [], "student_only": []}
for it in range(10):
    turn = tr.global_step % tr.vae_train_every == 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tr.step(images, labels)
    torch.cuda.synchronize(); times["vae_turn" if turn else "student_only"].append(time.perf_counter() - t0)
log = tr.read_log()
print({k: round(v, 4) for k, v in log.items()})
for k, v in times.items():
    print(f"{k}: {sum(v)/len(v)*1e3:.1f} ms/step over {len(v)} steps")
print(f"average over the 5-step cycle: {(sum(times['vae_turn'])/len(times['vae_turn']) + 4*sum(times['student_only'])/len(times['student_only']))/5*1e3:.1f} ms/step, "
      f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
