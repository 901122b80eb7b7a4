"""Is the HIP-path LightningDiT forward at B = 16 bound by the host's launch rate?  Enqueue time (host loop, no sync) vs wall time, and a HIP-graph replay of the same forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dmvae_amd.models.lightningdit import LightningDiT_models
B = int(os.environ.get("B", "16"))
torch.manual_seed(0)
m = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda().eval().requires_grad_(False)
x = torch.randn(B, 32, 16, 16, device="cuda"); t = torch.rand(B, device="cuda"); y = torch.randint(0, 1001, (B,), device="cuda")
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    for _ in range(3): m(x, t, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(x, t, y)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"eager: host enqueue {(t1-t0)/10*1e3:.2f} ms / forward, wall {(t2-t0)/10*1e3:.2f} ms / forward", flush=True)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): m(x, t, y)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m(x, t, y)
    torch.cuda.synchronize()
    for _ in range(2): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize(); print(f"graph replay: {(time.perf_counter()-t0)/10*1e3:.2f} ms / forward", flush=True)
