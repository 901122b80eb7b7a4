"""Per-call times of the conv / weight-gradient / GEMM calls of one tokenizer step WITH the adversarial branch (ops.KERNEL_TIMING: HIP events around every call), the calls
of the adversarial branch only (labels that do not occur in the step without it are marked)."""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from dmvae_amd import ops
from dmvae_amd.train import build_tokenizer_trainer
images = torch.rand(32, 3, 256, 256, device="cuda") * 2 - 1
res = {}
for with_disc in (False, True):
    tr = build_tokenizer_trainer(device="cuda", seed=42, **({"with_disc": True, "disc_start_step": 0} if with_disc else {}))
    for _ in range(4): tr.step(images)
    torch.cuda.synchronize()
    ops.KERNEL_TIMING = t = []
    tr.step(images)
    torch.cuda.synchronize()
    ops.KERNEL_TIMING = None
    agg = collections.OrderedDict()
    for label, e0, e1, fl in t:
        a = agg.setdefault(label, [0, 0.0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1) * 1e3; a[2] += fl
    res[with_disc] = agg
    del tr
base = res[False]
rows = []
for label, (n, us, fl) in res[True].items():
    n0, us0, _ = base.get(label, (0, 0.0, 0.0))
    if n > n0:
        rows.append((us - us0, n - n0, (us - us0) / (n - n0), (fl / n) * 1e-6 / ((us - us0) / (n - n0)) if us > us0 else 0.0, label))
rows.sort(reverse=True)
print("calls the adversarial branch adds: total us, calls, us per call, TFLOP/s, label")
for r in rows[:30]:
    print("%9.1f us %3d x %8.1f us %7.1f TF/s  %s" % r)
print("sum %.2f ms" % (sum(r[0] for r in rows) / 1e3))
