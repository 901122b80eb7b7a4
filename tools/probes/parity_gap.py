"""What happens today when the f2 / f3 modules run under the fp32 parity mode (probe for the round-6 parity work)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import load_golden, rel_err
from dmvae_amd import parity
DEV = "cuda"

def patchgan():
    from test_oracle_gan import patchgan_params
    from dmvae_amd.models.patchgan import NLayerDiscriminator
    g = load_golden("patchgan_small")
    p = patchgan_params(g, int(g["seed"]))
    disc = NLayerDiscriminator()
    sd = disc.state_dict()
    for k in sd:
        if k in p: sd[k] = p[k].clone()
    disc.load_state_dict(sd, strict=True); disc = disc.to(DEV).train()
    xg = g.t("x").to(DEV).requires_grad_(True)
    y = disc(xg); y.backward(g.t("dy").to(DEV))
    print("patchgan y", rel_err(y.detach().cpu(), g.t("y")), "dx", rel_err(xg.grad.cpu(), g.t("dx")))
    for n, prm in disc.named_parameters():
        if "g." + n in g: print("  grad", n, rel_err(prm.grad.cpu(), g.t("g." + n)))

def dit():
    from test_oracle_dit import CFGS, build
    for tag in ("dit_small_hd64w",):
        g = load_golden(tag); m = build(tag, g).to(DEV)
        xa = g.t("x").to(DEV).requires_grad_(True)
        out = m(xa, g.t("t").to(DEV), torch.from_numpy(np.asarray(g["y"])).to(DEV))
        (out.float() * g.t("dy").to(DEV)).sum().backward()
        print(tag, "out", rel_err(out.float().cpu(), g.t("out")), "dx", rel_err(xa.grad.cpu(), g.t("dx")))
        for n, prm in m.named_parameters():
            if "g." + n in g: print("  grad", n, rel_err(prm.grad.cpu(), g.t("g." + n)))

for name, fn in (("patchgan", patchgan), ("dit", dit)):
    try:
        with parity.enabled():
            fn()
    except Exception:
        print(name, "FAILED under parity mode:"); traceback.print_exc(limit=6)
