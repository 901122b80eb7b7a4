import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
B = 16
torch.manual_seed(0)
m = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda()
with torch.no_grad():
    for blk in m.blocks: blk.adaLN_modulation[1].weight.normal_(0, 0.02)
    m.final_layer.linear.weight.normal_(0, 0.02)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=0.005, betas=(0.9, 0.95), eps=1e-8)
xt = torch.randn(B, 32, 16, 16, device="cuda"); t = torch.rand(B, device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda"); ut = torch.randn_like(xt)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m(xt, t, y)
        loss = ((out.float() - ut) ** 2).flatten(1).mean(1).mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print(f"student train step (HIP route): {(time.perf_counter()-t0)/5*1e3:.1f} ms", flush=True)
