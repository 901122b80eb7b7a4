"""Config C1 (toy_example_2d/dmd.py) on the HIP path: dmvae_amd.toy.ToyDMDTrainer.

The points turn is held to the reference's own compute_distribution_matching_loss ("dmd" branch, golden dmd_loss_toy: injected t, x0 and velocities)
-- loss and gradient at 1e-4 -- and to torch.optim.AdamW for the update; the loop's turn-taking (vae_train_every / fake_warmup_steps), the student's
flow-matching turn and the qualitative behaviour (points move towards the teacher's distribution) are checked on an analytic velocity field."""
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _Inject(torch.nn.Module):
    def __init__(self, v):
        super().__init__()
        self.v = v

    def forward(self, xt, t, y):
        return self.v


def test_points_turn_vs_reference_fixture_and_adamw():
    from dmvae_amd.toy import ToyDMDTrainer
    g = load_golden("dmd_loss_toy")
    pts0 = g.t("points")
    tr = ToyDMDTrainer(_Inject(g.t("v_teacher").to(DEV)), _Inject(g.t("v_student").to(DEV)), points=pts0.to(DEV), lr=1e-2)
    labels = torch.zeros(64, dtype=torch.long, device=DEV)
    loss, log = tr.dmd_loss(labels, t=g.t("t_raw").to(DEV), x0=g.t("x0").to(DEV))
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert rel_err(tr.points.grad.cpu(), g.t("dpoints")) < 1e-4
    # the update: clip at 1e5 (inactive) + AdamW(lr, wd 0, betas (0.9, 0.95), eps 1e-8), first step (toy_example_2d/dmd.py:628, :677-679)
    ref = torch.nn.Parameter(pts0.clone())
    ref.grad = g.t("dpoints").clone()
    opt = torch.optim.AdamW([ref], lr=1e-2, weight_decay=0, betas=(0.9, 0.95), eps=1e-8)
    torch.nn.utils.clip_grad_norm_(ref, max_norm=100000.0)
    opt.step()
    norm = tr.popt.step()
    assert rel_err(tr.points.detach().cpu(), ref.detach()) < 1e-5
    assert abs(norm[0].item() - g.t("dpoints").norm().item()) < 1e-4 * g.t("dpoints").norm().item()


class _Field(torch.nn.Module):
    """Velocity of the linear path towards a point mass at `target`: v(xt, t) = (target - xt) / (1 - t) (clamped), i.e. pred_x1 = xt + (1-t) v = target."""
    def __init__(self, target, trainable=False):
        super().__init__()
        self.target = torch.nn.Parameter(torch.tensor(target, dtype=torch.float32).view(1, 2, 1, 1), requires_grad=trainable)

    def forward(self, xt, t, y):
        return (self.target - xt.float()) / (1 - t.float()).clamp_min(1e-3).view(-1, 1, 1, 1)


def test_toy_loop_turns_and_convergence():
    from dmvae_amd.toy import ToyDMDTrainer
    teacher = _Field([0.5, -0.25]).to(DEV)
    student = _Field([-1.0, 1.0], trainable=True).to(DEV)
    tr = ToyDMDTrainer(teacher, student, num_points=256, lr=5e-2, diff_lr=5e-2, vae_train_every=2, fake_warmup_steps=4, t0=0.02, t1=0.98, seed=3)
    p0 = tr.points.detach().clone()
    assert p0.shape == (256, 2) and p0.abs().max() <= 1.5
    turns = []
    for _ in range(10):
        out = tr.step()
        turns.append(out["dmd_loss"] is not None)
        assert out["sit_loss"] is not None                    # the student trains every step (:690-709)
    # steps 0..3: every = fake_warmup_steps = 4 -> only step 0; from step 4 on every 2nd step (:650-655)
    assert turns == [True, False, False, False, True, False, True, False, True, False]
    assert tr.global_step == 10 and student.training             # left in train mode by its own turn (:688)
    log = tr.read_log()
    assert all(v == v for v in log.values()) and log["points_grad_norm"] > 0
    # the student's turn is flow matching on the current points: with this field its loss is |target_s - x1|^2 / (1 - t)^2, minimised by the points' mean
    s0 = torch.tensor([-1.0, 1.0])
    for _ in range(300):
        tr.step()
    mean_now = tr.points.detach().mean(0).cpu()
    s_now = student.target.detach().view(2).cpu()
    assert (s_now - mean_now).norm() < 0.5 * (s0 - p0.mean(0).cpu()).norm(), (s_now, mean_now)
    # the points' turn with a FROZEN student: DMD gradient = pred_student - pred_teacher = target_s - target_t for every point, so AdamW moves all
    # points along target_t - target_s (:349-360: grad = p_real - p_student; the surrogate loss's gradient is grad / numel)
    frozen = _Field([-1.0, 1.0]).to(DEV)
    tr2 = ToyDMDTrainer(teacher, frozen, num_points=128, lr=1e-2, vae_train_every=1, seed=5)
    q0 = tr2.points.detach().clone()
    for _ in range(20):
        out = tr2.step()
        assert out["dmd_loss"] is not None and out["sit_loss"] is None          # nothing to train in the student
    disp = (tr2.points.detach() - q0).cpu()
    want = torch.tensor([0.5, -0.25]) - torch.tensor([-1.0, 1.0])
    cos = torch.nn.functional.cosine_similarity(disp, want.expand_as(disp), dim=1)
    assert cos.min() > 0.9 and disp.norm(dim=1).min() > 0.1, (cos.min(), disp.norm(dim=1).min())       # ~20 Adam steps of 1e-2 per coordinate
    ck = tr.checkpoint()
    assert set(ck) == {"model", "points", "opt_sit", "steps"} and ck["steps"] == 310 and ck["points"].shape == (256, 2)


def test_toy_rejects_cpu_points():
    from dmvae_amd._lib import DmvaeHipError
    from dmvae_amd.toy import ToyDMDTrainer
    with pytest.raises(DmvaeHipError):
        ToyDMDTrainer(lambda *a: None, lambda *a: None, num_points=8, device="cpu")


def _rl2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def test_one_token_velocity_model_vs_reference_capture():
    """Config C1's velocity model -- LightningDiT-Mini/1 at ONE token per sample (toy_example_2d/dmd.py:436-454) -- on the HIP route (lightningdit_fast.forward_tokens1:
    no stock module, no DMVAE_ALLOW_STOCK) against the capture taken from the reference's own module (oracle/capture_golden_dit.py, `dit_toy_mini1`): output, input
    gradient, the eight fully captured parameter gradients and every parameter's gradient norm, by the bf16-site criterion (as close to the reference's f32 numbers
    as the CPU oracle with bf16 rounding at the autocast sites, x 1.15 + a floor).  q / k and their RMSNorm weights get exactly zero gradient (one key: softmax = 1)."""
    import numpy as np
    from oracle import ref_cpu as R
    from test_oracle_dit import CFGS, build
    tag = "dit_toy_mini1"
    g = load_golden(tag)
    m = build(tag, g).to(DEV)
    x, t, y, dy = g.t("x"), g.t("t"), torch.from_numpy(np.asarray(g["y"])), g.t("dy")
    xa = x.to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = m(xa, t.to(DEV), y.to(DEV))                     # one token per sample -> forward_tokens1
    (out.float() * dy.to(DEV)).sum().backward()
    names = [n for n, _ in m.named_parameters()]
    po = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for n in names:
        po[n].requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    yo = R.lightningdit_forward(xo, t, y, po, CFGS[tag]["num_heads"], 1, q=R.bf16_round)
    yo.backward(dy)

    def floor(hip, orc, ref, what, slack=1.15, abs_floor=2e-3):
        e_hip, e_orc = _rl2(hip, ref), _rl2(orc, ref)
        print(f"{tag} {what}: rel-L2 to the f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
        assert e_hip < slack * e_orc + abs_floor, (what, e_hip, e_orc)
    floor(out.float().cpu(), yo.detach(), g.t("out"), "out")
    floor(xa.grad.cpu(), xo.grad, g.t("dx"), "dx")
    pa = dict(m.named_parameters())
    for k in [k[2:] for k in g.keys() if k.startswith("g.")]:
        ref = g.t("g." + k)
        if ref.abs().max() < 1e-5:                            # q_norm / k_norm weights: analytically zero (1e-7 of rounding noise in the reference's own backward)
            assert pa[k].grad is None or pa[k].grad.abs().max() == 0, k
            continue
        floor(pa[k].grad.cpu(), po[k].grad, ref, "grad " + k)
    for n in names:
        if n == "pos_embed" or "gn." + n not in g:
            continue
        want = float(g["gn." + n][0])
        got = 0.0 if pa[n].grad is None else pa[n].grad.double().norm().item()
        if want < 1e-6:
            assert got < 1e-6, n
            continue
        e_orc = abs(po[n].grad.double().norm().item() - want) / want
        assert abs(got - want) / want < 1.15 * e_orc + 2e-2, (n, got, want)


def test_toy_trainer_with_the_reference_models_needs_no_stock_opt_in(monkeypatch):
    """ToyDMDTrainer with the reference's own models -- LightningDiT-Mini/1 teacher and student at one token per point (toy_example_2d/dmd.py:436-454) -- and WITHOUT
    DMVAE_ALLOW_STOCK: the velocity evaluations of the DMD loss and the student's flow-matching turn run on the HIP one-token route.  40 steps: finite losses, the
    student's loss falls, the points move, and a second run from the same seeds is bit-identical."""
    monkeypatch.delenv("DMVAE_ALLOW_STOCK", raising=False)
    from dmvae_amd.models.lightningdit import LightningDiT_models
    from dmvae_amd.toy import ToyDMDTrainer
    from oracle.detweights import det_fill_

    def run():
        mk = lambda seed: det_fill_(LightningDiT_models["LightningDiT-Mini/1"](input_size=1, in_channels=2, num_classes=1), seed, skip=("pos_embed",)) or None
        teacher = LightningDiT_models["LightningDiT-Mini/1"](input_size=1, in_channels=2, num_classes=1)
        student = LightningDiT_models["LightningDiT-Mini/1"](input_size=1, in_channels=2, num_classes=1)
        det_fill_(teacher, 5, skip=("pos_embed",))
        det_fill_(student, 6, skip=("pos_embed",))
        teacher, student = teacher.to(DEV).eval().requires_grad_(False), student.to(DEV)
        torch.manual_seed(11)
        tr = ToyDMDTrainer(teacher, student, num_points=1536, lr=1e-3, diff_lr=1e-3, vae_train_every=2, seed=42)
        p0 = tr.points.detach().clone()
        logs = []
        for _ in range(40):
            tr.step()
            logs.append(tr.read_log())
        return p0, tr.points.detach().clone(), logs, tr.sfp.flat.clone()
    p0, p1, logs, w1 = run()
    assert all(v == v and abs(v) < 1e6 for lg in logs for v in lg.values()), logs[-1]
    first, last = sum(lg["sit_loss"] for lg in logs[:5]) / 5, sum(lg["sit_loss"] for lg in logs[-5:]) / 5
    assert last < first, (first, last)
    assert (p1 - p0).abs().max() > 1e-3 and logs[-1]["points_grad_norm"] > 0
    _, p1b, logs_b, w1b = run()
    assert torch.equal(p1, p1b) and torch.equal(w1, w1b) and logs == logs_b
