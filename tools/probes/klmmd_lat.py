import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dmvae_amd import ops
for g, n, m in [(1, 256, 256), (1, 32, 32), (1, 128, 32), (8, 256, 256), (32, 256, 256), (32, 256, 32)]:
    z = torch.randn(g, n, 32, device="cuda") * 0.7 + 0.2
    y = torch.randn(g, m, 32, device="cuda")
    for grad in (True, False):
        for _ in range(5): ops.kl_mmd(z, y, need_grad=grad)
        torch.cuda.synchronize()
        for _ in range(30): ops.kl_mmd(z, y, need_grad=grad)
        torch.cuda.synchronize()
