import sys; sys.path.insert(0,"."); sys.path.insert(0,"tests")
import torch
from test_gpu_train_step import _trainer
g = torch.Generator(device="cuda").manual_seed(0)
images = torch.rand(2, 3, 256, 256, device="cuda", generator=g) * 2 - 1
for da, db in [(True, True), (False, False), (True, False)]:
    a, b = _trainer(da), _trainer(db)
    for it in range(2):
        la, lb = a.step(images), b.step(images)
        d = (a.fp.flat - b.fp.flat).abs().max().item()
        dg = (a.fp.grad - b.fp.grad).abs().max().item()
        print(da, db, it, "loss eq", torch.equal(la, lb), "max param diff", d, "max grad diff", dg, "grad max", a.fp.grad.abs().max().item())
