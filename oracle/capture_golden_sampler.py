"""Generate the downstream-consumer fixtures (tests/golden/sampler_*.npz, image_u8.npz, sample50k_plan.npz, transport_train.npz) by running the
reference's own `diffusion.transport` package (Sampler / sde / Transport, read-only at /root/reference) in THIS container, on the small
LightningDiT of capture_golden_dit.py with deterministic weights.  torchdiffeq (imported by integrators.py:4, absent here) is one of
capture_golden.install_stubs' import stubs; none of the captured paths calls it.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_sampler.py        (CPU, seconds)

The sampler draws its per-step noise from the global CPU generator (`th.randn(x.size())`, integrators.py:28,38); each fixture records the seed set
just before the call, so a test reproduces the identical stream with `torch.manual_seed(seed)`."""
from __future__ import annotations

import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import capture_golden as cg  # noqa: E402
from oracle.detweights import det_fill_  # noqa: E402

DIT_KW = dict(input_size=8, patch_size=1, in_channels=8, hidden_size=144, depth=2, num_heads=2, num_classes=10)
DIT_SEED = 72


def main():
    cg.install_stubs()
    torch.set_grad_enabled(False)
    from diffusion.lightningdit.lightningdit import LightningDiT
    from diffusion.transport import Sampler, create_transport
    m = LightningDiT(**DIT_KW).eval()
    det_fill_(m, DIT_SEED, skip=("pos_embed",))
    g = torch.Generator().manual_seed(4242)
    z = torch.randn(4, 8, 8, 8, generator=g)
    y = torch.tensor([3, 0, 9, 10])
    tr = create_transport("Linear", "velocity", None, None, None, time_dist_shift=2.5)
    sampler = Sampler(tr)
    cases = (("sampler_euler_sigma_mean", dict(sampling_method="Euler", diffusion_form="sigma", diffusion_norm=1.0, last_step="Mean", last_step_size=0.04, num_steps=10), 11),
             ("sampler_heun_linear_mean", dict(sampling_method="Heun", diffusion_form="linear", diffusion_norm=0.7, last_step="Mean", last_step_size=0.04, num_steps=6), 12),
             ("sampler_euler_decreasing_euler", dict(sampling_method="Euler", diffusion_form="decreasing", diffusion_norm=1.0, last_step="Euler", last_step_size=0.1, num_steps=7), 13),
             ("sampler_euler_incdec_tweedie", dict(sampling_method="Euler", diffusion_form="inccreasing-decreasing", diffusion_norm=1.3, last_step="Tweedie", last_step_size=0.05, num_steps=5), 14),
             ("sampler_euler_sigma_none", dict(sampling_method="Euler", diffusion_form="sigma", diffusion_norm=1.0, last_step=None, last_step_size=0.04, num_steps=5), 16))
    for tag, kw, seed in cases:
        fn = sampler.sample_sde(**kw)
        torch.manual_seed(seed)
        xs = fn(z, m.forward, y=y)
        cg.save(tag, seed=np.array(seed), z=z, y=y, xs=torch.stack(xs), dit_seed=np.array(DIT_SEED),
                **{k: np.array(v if v is not None else "None") for k, v in kw.items()})
    # with classifier-free guidance through the model's own forward_with_cfg (train_diffusion.py:248-256 / sample_50k's using_cfg branch)
    fn = sampler.sample_sde(sampling_method="Euler", diffusion_form="sigma", last_step="Mean", last_step_size=0.04, num_steps=6)
    zz = torch.cat([z[:2], z[:2]], 0)
    yy = torch.tensor([3, 7, 10, 10])
    torch.manual_seed(15)
    xs = fn(zz, m.forward_with_cfg, y=yy, cfg_scale=2.5, standard_cfg=True)
    cg.save("sampler_euler_cfg", seed=np.array(15), z=zz, y=yy, xs=torch.stack(xs), cfg_scale=np.array(2.5), dit_seed=np.array(DIT_SEED), num_steps=np.array(6))
    # check_interval table (transport.py:75-102) for the velocity / Linear transport
    rows = []
    for form in ("SBDM", "sigma"):
        for sde in (False, True):
            for ev in (False, True):
                for rev in (False, True):
                    for lss in (0.0, 0.04):
                        t0, t1 = tr.check_interval(tr.train_eps, tr.sample_eps, diffusion_form=form, sde=sde, eval=ev, reverse=rev, last_step_size=lss)
                        rows.append([form == "SBDM", sde, ev, rev, lss, t0, t1])
    # Transport.sample + training_losses (transport.py:105-143) with the time shift of scripts/sample50k.sh / train_dmd.py
    x1 = torch.randn(4, 8, 8, 8, generator=g)
    torch.manual_seed(21)
    t, terms = tr.training_losses(m, x1, dict(y=y))
    torch.manual_seed(21)
    t2, x0, _ = tr.sample(x1)
    assert torch.equal(t, t2)
    cg.save("transport_train", seed=np.array(21), x1=x1, y=y, t=t, x0=x0, pred=terms["pred"], loss=terms["loss"], time_dist_shift=np.array(2.5),
            intervals=np.array(rows, dtype=np.float64), dit_seed=np.array(DIT_SEED))
    # sample_50k.py:143-151: DiT output -> tokens -> (decode) -> uint8; the conversion on values that hit every clamp / truncation case
    s = torch.randn(2, 3, 16, 16, generator=g) * 0.8
    s.view(-1)[:12] = torch.tensor([-1.0, 1.0, -1.5, 1.5, 0.0, 0.99607843, 0.996, -1.0039216, 127.5 / 127.5 - 1e-7, 1e-3 - 1.0, 0.5, -0.5])
    u8 = torch.clamp(127.5 * s + 128.0, 0, 255).permute(0, 2, 3, 1).to("cpu", dtype=torch.uint8).numpy()
    sb = s.to(torch.bfloat16).float()
    u8b = torch.clamp(127.5 * sb + 128.0, 0, 255).permute(0, 2, 3, 1).to("cpu", dtype=torch.uint8).numpy()
    samples = torch.randn(2, 8, 4, 4, generator=g)
    n_batch, c, h, w = samples.shape
    p = 1
    tok = torch.einsum('nchpwq->nhwpqc', samples.reshape(shape=(n_batch, c, h, p, h, p))).reshape(shape=(n_batch, h * w, p * p * c)) / 0.1763 + 0.0685
    back = torch.einsum('nhwpqc->nchpwq', ((tok - 0.0685) * 0.1763).reshape(shape=(n_batch, h, w, p, p, c))).reshape(shape=(n_batch, c, h * p, h * p))
    cg.save("image_u8", s=s, u8=u8, u8_bf16=u8b, samples=samples, tokens=tok, back=back, latent_mean=np.array(0.0685), latent_scale=np.array(0.1763))
    # the label / file-index split of sample_50k.py:128-157, replayed from the script's own statements for two world sizes
    plan = {}
    for ws, n, nfid, ncls in ((8, 25, 50000, 1000), (2, 5, 40, 10)):
        for rank in (0, ws - 1):
            label_list = list(range(ncls)) * (nfid // ncls)
            per_rank = len(label_list) // ws
            mine = label_list[per_rank * rank: per_rank * rank + per_rank]
            total, ys, idx = 0, [], []
            for it in range(int(math.ceil(per_rank / n))):
                total += n * ws
                ys.append(mine[it * n:(it + 1) * n])
                idx.append([li * ws + rank + total for li in range(n)])
            plan[f"y_{ws}_{rank}"] = np.array(ys)
            plan[f"i_{ws}_{rank}"] = np.array(idx)
    cg.save("sample50k_plan", **plan)


if __name__ == "__main__":
    main()
