"""Config C1 of BASELINE.json: the 2-D toy of the reference (toy_example_2d/dmd.py) on the HIP path.

What the reference's script does (main, :621-714): `num_points` learnable 2-D points, initialised uniformly in [-1.5, 1.5]^2
(create_learnable_points, :139-145), are moved by the DMD score-gradient loss (DMDLossFunction.compute_distribution_matching_loss, branch "dmd",
:320-360 -- the tokenizer stage's loss without classifier-free guidance and WITHOUT the per-sample weight factor) between a frozen teacher velocity
model trained on the S-shaped data distribution (sshpae.py:6-71) and a student that is trained, every step, with the flow-matching loss on the
current points (transport.training_losses, :702-714).  Each point is a one-token "image" [2, 1, 1]; all points carry label 0.

Built here: the S-shape sampler (host-side numpy like the reference's: it feeds the teacher's own training, not the device path), the point
initialiser, and `ToyDMDTrainer`, whose points turn runs on csrc/losses.hip::dmd_pre / dmd_post (through losses.dmd_make_xt / losses.dmd_loss)
and on the fused clip + AdamW of csrc/optim.hip; the student's turn is the same code path as train.DMDTrainer's.  The velocity models are
callables f(xt [B,2,1,1], t [B], labels [B]) -> velocity: LightningDiT-Mini/1 in the reference (toy_example_2d/dmd.py:436-454), which this build runs on its
one-token HIP route (models/lightningdit_fast.forward_tokens1, round 6: with one key the attention output is v, so a block is five Linears on the GEMM kernels
plus csrc/dit.hip's norm / modulate / gate / SwiGLU steps; pinned by tests/golden/dit_toy_mini1.npz) -- no stock module, no DMVAE_ALLOW_STOCK.
The other fifteen `dmd_loss_type` variants, plotting and wandb logging are out of scope.

Pinned by tests/golden/dmd_loss_toy.npz (the reference's own compute_distribution_matching_loss on injected velocities) and tests/golden/sshape.npz
(SShapeDistribution2D(random_state=42).sample(1536)): tests/test_gpu_toy.py, tests/test_host_logic.py."""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import losses
from .optim import FlatAdamWEMA, FlatParams


class SShapeDistribution2D:
    """The toy's data distribution (sshpae.py:6-71): a thick, slightly skewed S, noisier towards its ends, clipped to [-1, 1]^2.

    Same constructor arguments, same numpy Generator consumption order (curve parameter, normal-direction offset, isotropic diffusion), so a given
    `random_state` yields the reference's samples bit for bit."""

    def __init__(self, thickness: float = 0.06, diffusion: float = 0.03, x_range=(-1.0, 1.0), y_range=(-1.0, 1.0), amplitude: float = 0.85,
                 vertical_scale: float = 0.85, skew: float = 0.15, flip_y: bool = True, random_state: Optional[int] = None):
        self.thickness, self.diffusion, self.x_range, self.y_range = thickness, diffusion, x_range, y_range
        self.amplitude, self.vertical_scale, self.skew, self.flip_y = amplitude, vertical_scale, skew, flip_y
        self.rng = np.random.default_rng(random_state)

    def _curve(self, u: np.ndarray) -> np.ndarray:
        """Centre line and its unit normal direction angle at parameter u in [-1, 1]."""
        centre = np.stack([self.amplitude * np.sin(np.pi * u), self.vertical_scale * u - self.skew * np.sin(2 * np.pi * u)], axis=1)
        d = np.stack([self.amplitude * np.pi * np.cos(np.pi * u), self.vertical_scale - 2 * np.pi * self.skew * np.cos(2 * np.pi * u)], axis=1)
        d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-8
        return centre, np.arctan2(d[:, 0], -d[:, 1])              # tangent rotated by 90 degrees

    def sample(self, n: int):
        u = self.rng.uniform(-1.0, 1.0, size=n)
        pts, ang = self._curve(u)
        off = self.rng.normal(0.0, self.thickness, size=n)          # thickness: offset along the normal
        pts[:, 0] += off * np.cos(ang)
        pts[:, 1] += off * np.sin(ang)
        pts += self.rng.normal(0.0, (self.diffusion * (0.4 + 0.6 * np.abs(u)))[:, None])      # the ends of the S are more diffuse
        if self.flip_y:
            pts[:, 1] *= -1
        pts[:, 0] = np.clip(pts[:, 0], *self.x_range)
        pts[:, 1] = np.clip(pts[:, 1], *self.y_range)
        return pts, np.zeros(n, dtype=int)


def create_learnable_points(num_points: int, z_channels: int, device, seed: int = 42) -> torch.nn.Parameter:
    """toy_example_2d/dmd.py:139-145: seeds the global generator, draws on `device`, uniform in [-1.5, 1.5]."""
    torch.manual_seed(seed)
    return torch.nn.Parameter(torch.rand(num_points, z_channels, device=device) * 3.0 - 1.5)


class ToyDMDTrainer:
    """One iteration of toy_example_2d/dmd.py's loop (:646-714).

    points turn (every `vae_train_every`-th step; during the first `fake_warmup_steps` steps only at step 0, :650-653): student frozen and in eval mode,
    loss = 0.5 * mse(points, (points - grad).detach()) with grad = (points - pred_teacher) - (points - pred_student) evaluated at
    xt = t * points + (1 - t) * x0, t ~ U(t0, t1) (:334-360); clip_grad_norm_(points, 1e5); AdamW(lr, weight_decay 0, betas (0.9, 0.95)) (:628, :677-679).
    student turn (every step): flow-matching loss on the detached points, clip_grad_norm_(1.0), AdamW(diff_lr, wd) (:629, :690-709)."""

    def __init__(self, teacher: Callable, student, num_points: int = 1536, z_channels: int = 2, lr: float = 1e-3, diff_lr: float = 1e-4,
                 wd: float = 0.0, vae_train_every: int = 1, fake_warmup_steps: int = 0, t0: float = 0.0, t1: float = 1.0, seed: int = 42,
                 device="cuda", points: Optional[torch.Tensor] = None):
        self.teacher, self.student = teacher, student
        self.points = create_learnable_points(num_points, z_channels, device, seed) if points is None else torch.nn.Parameter(points.detach().clone())
        if not self.points.is_cuda:
            from ._lib import DmvaeHipError
            raise DmvaeHipError("ToyDMDTrainer: the points live on the GPU; dmvae_amd has no CPU path")
        self.vae_train_every, self.fake_warmup_steps, self.t0, self.t1 = vae_train_every, fake_warmup_steps, t0, t1
        self.pfp = FlatParams([self.points], with_ema=False)
        self.popt = FlatAdamWEMA(self.pfp, lr=lr, weight_decay=0.0, betas=(0.9, 0.95), eps=1e-8, warmup_steps=0, max_norm=100000.0)
        sp = [p for p in student.parameters() if p.requires_grad] if hasattr(student, "parameters") else []
        self.sfp = self.sopt = None
        if sp:
            self.sfp = FlatParams(sp, with_ema=False)
            self.sopt = FlatAdamWEMA(self.sfp, lr=diff_lr, weight_decay=wd, betas=(0.9, 0.95), eps=1e-8, warmup_steps=0, max_norm=1.0)
        self.log = torch.zeros(5, dtype=torch.float32, device=self.points.device)
        self.global_step = 0

    @staticmethod
    def _sample(x1: torch.Tensor):
        """Transport.sample (transport.py:105-116): x0 from the device generator, t from the CPU generator."""
        x0 = torch.randn_like(x1)
        t = torch.rand((x1.shape[0],)).to(x1)
        return t, x0

    def dmd_loss(self, labels: torch.Tensor, t: Optional[torch.Tensor] = None, x0: Optional[torch.Tensor] = None, force_t: Optional[float] = None):
        """compute_distribution_matching_loss, branch "dmd" (:320-360).  t / x0 may be injected (tests); force_t as in the reference's eval (:337-340)."""
        pts = self.points.view(self.points.shape[0], -1, 1, 1)
        if t is None or x0 is None:
            t, x0 = self._sample(pts)
        t = t * (self.t1 - self.t0) + self.t0 if force_t is None else torch.ones_like(t) * force_t
        xt = losses.dmd_make_xt(pts, x0, t)
        with torch.no_grad():
            vt, vs = self.teacher(xt, t, labels), self.student(xt, t, labels)
        return losses.dmd_loss(pts, xt, t, vt.float(), vs.float(), cfg=1.0, weight_factor=False)

    def step(self) -> Dict[str, Optional[torch.Tensor]]:
        every = self.fake_warmup_steps if self.global_step < self.fake_warmup_steps else self.vae_train_every
        points_turn = self.global_step % every == 0
        labels = torch.zeros(self.points.shape[0], dtype=torch.long, device=self.points.device)       # a single class (:663)
        student_is_module = isinstance(self.student, torch.nn.Module)
        out: Dict[str, Optional[torch.Tensor]] = {"dmd_loss": None, "sit_loss": None}
        if points_turn:
            for p in getattr(self.student, "parameters", lambda: [])():
                p.requires_grad_(False)
            if student_is_module:
                self.student.eval()
            self.pfp.zero_grad()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss, dlog = self.dmd_loss(labels)
            loss.backward()
            norm = self.popt.step()
            with torch.no_grad():
                self.log[0], self.log[1], self.log[2] = loss.detach(), dlog[1], norm[0]
            out["dmd_loss"] = loss.detach()
        if self.sopt is not None:
            for p in self.sfp.params:
                p.requires_grad_(True)
            if student_is_module:
                self.student.train()
            self.sfp.zero_grad()
            x1 = self.points.detach().view(self.points.shape[0], -1, 1, 1)
            t, x0 = self._sample(x1)
            te = t.view(-1, 1, 1, 1)
            xt, ut = te * x1 + (1 - te) * x0, x1 - x0                  # ICPlan.plan (path.py:114-136), velocity target
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = self.student(xt, t, labels)
                sloss = ((pred.float() - ut) ** 2).flatten(1).mean(1).mean()
            sloss.backward()
            snorm = self.sopt.step()
            with torch.no_grad():
                self.log[3], self.log[4] = sloss.detach(), snorm[0]
            out["sit_loss"] = sloss.detach()
        self.global_step += 1
        return out

    def checkpoint(self) -> dict:
        """:716-725: model / points / opt_sit / steps."""
        return {"model": {k: v.detach().clone() for k, v in self.student.state_dict().items()} if isinstance(self.student, torch.nn.Module) else None,
                "points": self.points.data.cpu().clone(),
                "opt_sit": self.sopt.state_dict(list(self.student.parameters())) if self.sopt is not None else None, "steps": self.global_step}

    def read_log(self) -> Dict[str, float]:
        v = self.log.tolist()
        return {"dmd_loss": v[0], "dmd_gradient_norm": v[1], "points_grad_norm": v[2], "sit_loss": v[3], "sit_grad_norm": v[4]}
