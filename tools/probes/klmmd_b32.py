import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmvae_amd import ops
z = torch.randn(32, 256, 32, device="cuda") * 0.7 + 0.2
y = torch.randn(32, 256, 32, device="cuda")
for grad in (True, False):
    for _ in range(5): ops.kl_mmd(z, y, need_grad=grad)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): ops.kl_mmd(z, y, need_grad=grad)
    e1.record(); t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"grad={grad}: events {e0.elapsed_time(e1)/200*1e3:.1f} us/call, host issue {(t1-t0)/200*1e6:.1f} us/call")
