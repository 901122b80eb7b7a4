"""Is the student DiT's stock train step launch-bound, and does hipGraph capture of (fwd + bwd + clip + AdamW) fix it?"""
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
opt = torch.optim.AdamW(m.parameters(), lr=torch.tensor(1e-4, device="cuda"), weight_decay=0.005, betas=(0.9, 0.95), eps=1e-8, capturable=True)
xt = torch.randn(B, 32, 16, 16, device="cuda"); t = torch.rand(B, device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda"); ut = torch.randn_like(xt)
def step():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m.forward_stock(xt, t, y)
        loss = ((out.float() - ut) ** 2).flatten(1).mean(1).mean()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
    opt.step()
    opt.zero_grad(set_to_none=False)
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); print(f"eager: {(time.perf_counter()-t0)/5*1e3:.1f} ms/step", flush=True)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = step()
torch.cuda.synchronize()
for _ in range(2): g.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize(); print(f"graph replay: {(time.perf_counter()-t0)/5*1e3:.1f} ms/step, loss {static_loss.item():.4f}", flush=True)
