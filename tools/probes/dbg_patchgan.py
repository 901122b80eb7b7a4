import sys, os
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_); sys.path.insert(0, R_ + "/tests")
import torch
from conftest import load_golden
from oracle import ref_cpu as R
from test_oracle_gan import patchgan_params
from test_gpu_gan import _disc
Q = R.bf16_round
g = load_golden("patchgan_small")
p = patchgan_params(g, int(g["seed"]))
disc = _disc(p); disc.train()
xg = g.t("x").cuda().requires_grad_(True)
y = disc(xg); y.backward(g.t("dy").cuda())
po = {k: (v.clone().requires_grad_(True) if "running" not in k else v.clone()) for k, v in p.items()}
pe = {k: (v.clone().requires_grad_(True) if "running" not in k else v.clone()) for k, v in p.items()}
xo = g.t("x").requires_grad_(True)
yo, _ = R.patchgan_forward(xo, po, q=Q, training=True); yo.backward(Q(g.t("dy")))
ye, _ = R.patchgan_forward(g.t("x"), pe, training=True); ye.backward(g.t("dy"))
rl2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
for n, prm in disc.named_parameters():
    h, o, e = prm.grad.cpu(), po[n].grad, pe[n].grad
    print(f"{n:18s} |exact| {e.norm():.3e}  hip-vs-exact {rl2(h,e):.3e}  orc-vs-exact {rl2(o,e):.3e}  hip-vs-orc {rl2(h,o):.3e}  maxabs-rel hip-vs-orc {((h-o).abs().max()/o.abs().max()).item():.3e}")
