import sys, os, time
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, R_ + "/tests")
import torch
from test_gpu_vit_train import _vit
vit = _vit(1024, 2, 16, 256, seed=3)
x = torch.randn(4, 3, 256, 256, generator=torch.Generator().manual_seed(2)).cuda()
outs = []
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    vit.zero_grad(set_to_none=True)
    xa = x.clone().requires_grad_(True)
    y = vit.forward_features(xa)
    torch.cuda.synchronize(); t1 = time.time()
    y.float().square().mean().backward()
    torch.cuda.synchronize(); t2 = time.time()
    print(f"iter {it}: fwd {t1-t0:.2f}s bwd {t2-t1:.2f}s")
    outs.append([("y", y.detach().clone()), ("dx", xa.grad.clone())] + [(n, p.grad.clone()) for n, p in vit.named_parameters()])
for (n, a), (_, b) in zip(outs[1], outs[2]):
    fin = torch.isfinite(a).all().item()
    eq = torch.equal(a, b)
    if not fin or not eq:
        print(n, "finite", fin, "equal", eq, "maxdiff", (a.float() - b.float()).abs().max().item(), "max", a.float().abs().max().item())
print("done")
