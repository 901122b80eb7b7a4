"""Host mirror of the reference's `diffusion/transport` package for the configuration its scripts run (Linear path, velocity prediction):
`create_transport` (__init__.py:3-64), `Transport` (transport.py:39-221), `ICPlan` (path.py:18-136), `Sampler` (transport.py:223-458) and the
`sde` / `ode` integrators (integrators.py:8-118) -- same names, arguments, defaults, random-number consumption and error behaviour.

What runs where: a sampler step is one model forward (LightningDiT on the HIP kernels, models/lightningdit_fast.py) plus the state update;
the Euler-Maruyama update -- velocity -> score, drift, mean, noise injection -- is ONE kernel pass over the state (`ops.sde_euler_step`, csrc/sampler.hip)
with the reference's f32 arithmetic order, so for the same model output the trajectory is bit-identical to the PyTorch reference.  Heun and
the fixed-grid ODE methods are composed from device tensor ops.  The per-step noise is drawn on the CPU generator and moved to the state's
device exactly as the reference does (`th.randn(x.size()).to(x)`, integrators.py:28,38), so a seeded run consumes the same stream.

Not built: the GVP / VP plans (path.py:138-191; never selected by the reference's scripts: `path_type` is "Linear" everywhere), the likelihood
sampler (transport.py:409-458) and adaptive ODE solvers unless `torchdiffeq` (an unpinned third-party dependency, absent from this image) is importable."""
from __future__ import annotations

import enum
import os

import numpy as np
import torch as th

from . import ops

FUSED_STATE_UPDATE = True      # False composes the Euler-Maruyama update from tensor ops (tests/test_gpu_sampler.py compares the two)


class ModelType(enum.Enum):
    NOISE = enum.auto()
    SCORE = enum.auto()
    VELOCITY = enum.auto()


class PathType(enum.Enum):
    LINEAR = enum.auto()
    GVP = enum.auto()
    VP = enum.auto()


class WeightType(enum.Enum):
    NONE = enum.auto()
    VELOCITY = enum.auto()
    LIKELIHOOD = enum.auto()


def expand_t_like_x(t, x):
    """path.py:5-13."""
    return t.view(t.size(0), *([1] * (x.dim() - 1)))


def mean_flat(x):
    """utils.py:12-16."""
    return th.mean(x, dim=list(range(1, x.dim())))


class ICPlan:
    """Linear coupling plan x_t = t x1 + (1 - t) x0 (path.py:18-136).  Every method works on tensors of any device, like the reference's."""

    def __init__(self, sigma=0.0):
        self.sigma = sigma

    def compute_alpha_t(self, t):
        return t, 1

    def compute_sigma_t(self, t):
        return 1 - t, -1

    def compute_d_alpha_alpha_ratio_t(self, t):
        return 1 / t

    def compute_drift(self, x, t):
        t = expand_t_like_x(t, x)
        ratio = self.compute_d_alpha_alpha_ratio_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        return -(ratio * x), ratio * (sigma_t ** 2) - sigma_t * d_sigma_t

    def compute_diffusion(self, x, t, form="constant", norm=1.0):
        t = expand_t_like_x(t, x)
        if form == "constant":
            return norm
        if form == "SBDM":
            return norm * self.compute_drift(x, t)[1]
        if form == "sigma":
            return norm * self.compute_sigma_t(t)[0]
        if form == "linear":
            return norm * (1 - t)
        if form == "decreasing":
            return 0.25 * (norm * th.cos(np.pi * t) + 1) ** 2
        if form == "inccreasing-decreasing":            # the reference's spelling (path.py:59)
            return norm * th.sin(np.pi * t) ** 2
        raise NotImplementedError(f"Diffusion form {form} not implemented")

    def _score_coeffs(self, t):
        """(reverse_alpha_ratio, var) of get_score_from_velocity, in the reference's operation order (path.py:82-88)."""
        alpha_t, d_alpha_t = self.compute_alpha_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        rar = alpha_t / d_alpha_t
        return rar, sigma_t ** 2 - rar * d_sigma_t * sigma_t

    def get_score_from_velocity(self, velocity, x, t):
        rar, var = self._score_coeffs(expand_t_like_x(t, x))
        return (rar * velocity - x) / var

    def get_noise_from_velocity(self, velocity, x, t):
        t = expand_t_like_x(t, x)
        alpha_t, d_alpha_t = self.compute_alpha_t(t)
        sigma_t, d_sigma_t = self.compute_sigma_t(t)
        rar = alpha_t / d_alpha_t
        return (rar * velocity - x) / (rar * d_sigma_t - sigma_t)

    def get_velocity_from_score(self, score, x, t):
        drift, var = self.compute_drift(x, expand_t_like_x(t, x))
        return var * score - drift

    def compute_mu_t(self, t, x0, x1):
        t = expand_t_like_x(t, x1)
        return self.compute_alpha_t(t)[0] * x1 + self.compute_sigma_t(t)[0] * x0

    def compute_xt(self, t, x0, x1):
        return self.compute_mu_t(t, x0, x1)

    def compute_ut(self, t, x0, x1, xt):
        t = expand_t_like_x(t, x1)
        return self.compute_alpha_t(t)[1] * x1 + self.compute_sigma_t(t)[1] * x0

    def plan(self, t, x0, x1):
        xt = self.compute_xt(t, x0, x1)
        return t, xt, self.compute_ut(t, x0, x1, xt)


_RAND_RING = {}


def cpu_rand_like_batch(x1):
    """`th.rand((B,)).to(x1)` (transport.py:110-111): the batch's times drawn on the CPU generator, like the reference, and handed to x1's device WITHOUT stalling
    the launch queue: the draw lands in one of four pinned staging buffers and is copied with non_blocking (a pageable copy waits for the stream -- once per
    training step the host then stood still until the device had caught up, and every kernel after it was launched with the device on its heels:
    tools/probes/host_ahead.py; the sampler's `_noise` does the same for its per-step noise)."""
    n = x1.shape[0]
    if not x1.is_cuda:
        return th.rand((n,))
    key = (n, x1.device)
    ring = _RAND_RING.get(key)
    if ring is None:
        ring = _RAND_RING[key] = [[[th.empty((n,), dtype=th.float32).pin_memory(), None] for _ in range(4)], 0]
    slot = ring[0][ring[1]]
    ring[1] = (ring[1] + 1) % len(ring[0])
    if slot[1] is not None:
        slot[1].synchronize()                        # the copy that last used this buffer (four draws ago) has finished
    th.rand((n,), out=slot[0])
    t = slot[0].to(device=x1.device, non_blocking=True)
    slot[1] = th.cuda.Event()
    slot[1].record()
    return t                                          # f32 on x1's device: the caller casts where the reference's `.to(x1)` stands


class Transport:
    """transport.py:39-221 for the Linear path."""

    def __init__(self, *, model_type, path_type, loss_type, train_eps, sample_eps, time_dist_shift=1.0):
        if path_type != PathType.LINEAR:
            raise NotImplementedError("only the Linear path (path.ICPlan) is built; the reference's scripts never select GVP / VP")
        self.loss_type = loss_type
        self.model_type = model_type
        self.path_sampler = ICPlan()
        self.train_eps = train_eps
        self.sample_eps = sample_eps
        self.time_dist_shift = time_dist_shift

    def prior_logp(self, z):
        n = z[0].numel()
        return -n / 2.0 * np.log(2 * np.pi) - th.sum(z.flatten(1) ** 2, dim=1) / 2.0

    def check_interval(self, train_eps, sample_eps, *, diffusion_form="SBDM", sde=False, reverse=False, eval=False, last_step_size=0.0):
        """transport.py:75-102 (ICPlan branch)."""
        t0, t1 = 0, 1
        eps = train_eps if not eval else sample_eps
        if self.model_type != ModelType.VELOCITY or sde:
            t0 = eps if (diffusion_form == "SBDM" and sde) or self.model_type != ModelType.VELOCITY else 0
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        if reverse:
            t0, t1 = 1 - t0, 1 - t1
        return t0, t1

    def sample(self, x1):
        """transport.py:105-116: x0 on x1's device generator, t on the CPU generator, then the time shift."""
        x0 = th.randn_like(x1)
        t0, t1 = self.check_interval(self.train_eps, self.sample_eps)
        t = cpu_rand_like_batch(x1) * (t1 - t0) + t0
        t = t.to(x1)
        t = 1 - self.time_dist_shift * (1 - t) / (1 + (self.time_dist_shift - 1) * (1 - t))
        return t, x0, x1

    def training_losses(self, model, x1, model_kwargs=None):
        """transport.py:119-164: flow-matching loss per sample, `terms = {"pred", "loss"}`."""
        if model_kwargs is None:
            model_kwargs = {}
        t, x0, x1 = self.sample(x1)
        t, xt, ut = self.path_sampler.plan(t, x0, x1)
        model_output = model(xt, t, **model_kwargs)
        assert model_output.size() == xt.size()
        return t, {"pred": model_output, "loss": mean_flat((model_output - ut) ** 2)}

    def get_drift(self):
        """transport.py:167-199."""
        ps = self.path_sampler

        def score_ode(x, t, model, **kw):
            drift_mean, drift_var = ps.compute_drift(x, t)
            return -drift_mean + drift_var * model(x, t, **kw)

        def noise_ode(x, t, model, **kw):
            drift_mean, drift_var = ps.compute_drift(x, t)
            sigma_t, _ = ps.compute_sigma_t(expand_t_like_x(t, x))
            return -drift_mean + drift_var * (model(x, t, **kw) / -sigma_t)

        def velocity_ode(x, t, model, **kw):
            return model(x, t, **kw)

        fn = {ModelType.NOISE: noise_ode, ModelType.SCORE: score_ode}.get(self.model_type, velocity_ode)

        def body_fn(x, t, model, **kw):
            out = fn(x, t, model, **kw)
            assert out.shape == x.shape, "Output shape from ODE solver must match input shape"
            return out

        return body_fn

    def get_score(self):
        """transport.py:202-216."""
        ps = self.path_sampler
        if self.model_type == ModelType.NOISE:
            return lambda x, t, model, **kw: model(x, t, **kw) / -ps.compute_sigma_t(expand_t_like_x(t, x))[0]
        if self.model_type == ModelType.SCORE:
            return lambda x, t, model, **kw: model(x, t, **kw)
        if self.model_type == ModelType.VELOCITY:
            return lambda x, t, model, **kw: ps.get_score_from_velocity(model(x, t, **kw), x, t)
        raise NotImplementedError()

    def convert_score(self, score, x, t):
        return self.path_sampler.get_score_from_velocity(score, x, t)


def create_transport(path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None, time_dist_shift=1.0):
    """__init__.py:3-64 (including its quirk that `sample_eps` falls back on `train_eps is None`)."""
    model_type = {"noise": ModelType.NOISE, "score": ModelType.SCORE}.get(prediction, ModelType.VELOCITY)
    loss_type = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD}.get(loss_weight, WeightType.NONE)
    path = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}[path_type]
    if path == PathType.VP:
        train_eps, sample_eps = (1e-5 if train_eps is None else train_eps), (1e-3 if train_eps is None else sample_eps)
    elif model_type != ModelType.VELOCITY:
        train_eps, sample_eps = (1e-3 if train_eps is None else train_eps), (1e-3 if train_eps is None else sample_eps)
    else:
        train_eps = sample_eps = 0
    return Transport(model_type=model_type, path_type=path, loss_type=loss_type, train_eps=train_eps, sample_eps=sample_eps,
                     time_dist_shift=time_dist_shift)


# ---- integrators -----------------------------------------------------------------------------------------------------------------------
class sde:
    """integrators.py:8-77.  `sampler_type` "Euler" (Euler-Maruyama) or "Heun".

    With `fused` (set by `Sampler.sample_sde` for the Linear path with a velocity model) the Euler step calls the model for the velocity and does
    the whole update in `ops.sde_euler_step`; otherwise it evaluates the caller's `drift` / `diffusion` callables like the reference."""

    def __init__(self, drift, diffusion, *, t0, t1, num_steps, sampler_type, fused=None):
        assert t0 < t1, "SDE sampler has to be in forward time"
        self.num_timesteps = num_steps
        self.t = th.linspace(t0, t1, num_steps)
        self.dt = self.t[1] - self.t[0]
        self.drift = drift
        self.diffusion = diffusion
        self.sampler_type = sampler_type
        self.fused = fused                     # (path_sampler, diffusion_form, diffusion_norm) or None

    def _noise(self, x):
        """`th.randn(x.size()).to(x)` (integrators.py:28,38): drawn on the CPU generator like the reference.  For a device state the draw lands in one of
        four pinned staging buffers and is copied without blocking the host, so the next kernels are queued while the previous step still runs (a
        pageable copy stalls the launch queue once per step: 1.4 ms of 13 at sample_50k's batch)."""
        if not x.is_cuda:
            return th.randn(x.size()).to(x)
        ring = getattr(self, "_ring", None)
        if ring is None or ring[0][0].shape != x.shape:
            ring = self._ring = [[th.empty(x.size(), dtype=th.float32).pin_memory(), None] for _ in range(4)]
            self._ring_i = 0
        slot = ring[self._ring_i]
        self._ring_i = (self._ring_i + 1) % len(ring)
        if slot[1] is not None:
            slot[1].synchronize()                        # the copy that last used this buffer (four steps ago) has finished
        th.randn(x.size(), out=slot[0])
        w = slot[0].to(device=x.device, non_blocking=True)
        slot[1] = th.cuda.Event()
        slot[1].record()
        return w if w.dtype == x.dtype else w.to(x.dtype)

    def _coeffs(self, ti):
        """The scalars of one fused step, computed in f32 the way the reference's broadcast graph computes them."""
        ps, form, norm = self.fused
        te = ti.view(1, 1)
        rar, var = ps._score_coeffs(te)
        diff = ps.compute_diffusion(te, te.view(1), form=form, norm=norm)
        diff = diff if th.is_tensor(diff) else th.tensor(float(diff), dtype=th.float32)
        return float(rar), float(var), float(diff), float(th.sqrt(2 * diff))

    def _euler_maruyama_step(self, x, mean_x, t, model, **model_kwargs):
        w_cur = self._noise(x)
        if FUSED_STATE_UPDATE and self.fused is not None and x.is_cuda and x.dtype == th.float32 and x.numel() % 4 == 0:
            tv = th.full((x.size(0),), float(t), device=x.device, dtype=x.dtype)      # == th.ones(B).to(x) * t, without the blocking host copy
            v = model(x, tv, **model_kwargs)
            assert v.shape == x.shape, "Output shape from ODE solver must match input shape"
            rar, var, diff, sq2d = self._coeffs(t)
            return ops.sde_euler_step(x.contiguous(), v.contiguous(), w_cur, rar, var, diff, float(self.dt), sq2d, float(th.sqrt(self.dt)),
                                      need_mean=True)
        t = th.ones(x.size(0)).to(x) * t
        dw = w_cur * th.sqrt(self.dt)
        drift = self.drift(x, t, model, **model_kwargs)
        diffusion = self.diffusion(x, t)
        mean_x = x + drift * self.dt
        return mean_x + th.sqrt(2 * diffusion) * dw, mean_x

    def _heun_step(self, x, _, t, model, **model_kwargs):
        w_cur = self._noise(x)
        dw = w_cur * th.sqrt(self.dt)
        t_cur = th.ones(x.size(0)).to(x) * t
        diffusion = self.diffusion(x, t_cur)
        xhat = x + th.sqrt(2 * diffusion) * dw
        k1 = self.drift(xhat, t_cur, model, **model_kwargs)
        xp = xhat + self.dt * k1
        k2 = self.drift(xp, t_cur + self.dt, model, **model_kwargs)
        return xhat + 0.5 * self.dt * (k1 + k2), xhat

    def sample(self, init, model, **model_kwargs):
        try:
            step = {"Euler": self._euler_maruyama_step, "Heun": self._heun_step}[self.sampler_type]
        except KeyError:
            raise NotImplementedError("Smapler type not implemented.")
        x, mean_x, samples = init, init, []
        for ti in self.t[:-1]:
            with th.no_grad():
                x, mean_x = step(x, mean_x, ti, model, **model_kwargs)
                samples.append(x)
        return samples


_FIXED_GRID = ("euler", "midpoint", "heun3", "rk4")


class ode:
    """integrators.py:79-118.  The reference hands the drift to `torchdiffeq.odeint`; that package is not in this image, so the fixed-grid methods
    (torchdiffeq's fixed-grid set: explicit Euler, midpoint, Heun's third-order rule, the 3/8-rule RK4 -- one solver step per interval of the time grid; unpinned,
    there is no torchdiffeq here to compare against) are
    integrated here, and anything else ("dopri5", the reference's default) is delegated to torchdiffeq when it can be imported."""

    def __init__(self, drift, *, t0, t1, sampler_type, num_steps, atol, rtol, time_dist_shift=1.0):
        assert t0 < t1, "ODE sampler has to be in forward time"
        self.drift = drift
        t = th.linspace(t0, t1, num_steps)
        self.t = 1 - time_dist_shift * (1 - t) / (1 + (time_dist_shift - 1) * (1 - t))
        self.atol, self.rtol = atol, rtol
        self.sampler_type = sampler_type

    def sample(self, x, model, **model_kwargs):
        device = x[0].device if isinstance(x, tuple) else x.device

        def fn(t, x):
            n = x[0].size(0) if isinstance(x, tuple) else x.size(0)
            return self.drift(x, th.ones(n).to(device) * t, model, **model_kwargs)

        t = self.t.to(device)
        if self.sampler_type not in _FIXED_GRID:
            try:
                from torchdiffeq import odeint
            except ImportError as e:
                raise NotImplementedError(f"ODE method {self.sampler_type!r} needs torchdiffeq (not installed); fixed-grid methods available: "
                                          f"{_FIXED_GRID}") from e
            k = len(x) if isinstance(x, tuple) else 1
            return odeint(fn, x, t, method=self.sampler_type, atol=[self.atol] * k, rtol=[self.rtol] * k)
        if isinstance(x, tuple):
            raise NotImplementedError("tuple states (the likelihood sampler) are not built")
        out = [x]
        with th.no_grad():
            for i in range(t.numel() - 1):
                t0, dt = t[i], t[i + 1] - t[i]
                k1 = fn(t0, x)
                if self.sampler_type == "euler":
                    x = x + dt * k1
                elif self.sampler_type == "midpoint":
                    x = x + dt * fn(t0 + 0.5 * dt, x + 0.5 * dt * k1)
                elif self.sampler_type == "heun3":
                    k2 = fn(t0 + dt / 3, x + dt * k1 / 3)
                    k3 = fn(t0 + dt * 2 / 3, x + dt * k2 * (2 / 3))
                    x = x + dt * (0.25 * k1 + 0.75 * k3)
                else:                                                    # rk4, 3/8 rule
                    k2 = fn(t0 + dt / 3, x + dt * k1 / 3)
                    k3 = fn(t0 + dt * 2 / 3, x + dt * (k2 - k1 / 3))
                    k4 = fn(t0 + dt, x + dt * (k1 - k2 + k3))
                    x = x + dt * (k1 + 3 * (k2 + k3) + k4) * 0.125
                out.append(x)
        return th.stack(out)


class Sampler:
    """transport.py:223-407 (`sample_sde`, `sample_ode`)."""

    def __init__(self, transport: Transport):
        self.transport = transport
        self.drift = transport.get_drift()
        self.score = transport.get_score()

    def _sde_diffusion_and_drift(self, *, diffusion_form="SBDM", diffusion_norm=1.0):
        ps = self.transport.path_sampler

        def diffusion_fn(x, t):
            return ps.compute_diffusion(x, t, form=diffusion_form, norm=diffusion_norm)

        def sde_drift(x, t, model, **kw):
            temp = self.drift(x, t, model, **kw)
            return temp + diffusion_fn(x, t) * self.transport.convert_score(temp, x, t)

        return sde_drift, diffusion_fn

    def _last_step(self, sde_drift, *, last_step, last_step_size, fused=None):
        ps = self.transport.path_sampler
        if last_step is None:
            return lambda x, t, model, **kw: x
        if last_step == "Mean":
            def mean_step(x, t, model, **kw):
                if FUSED_STATE_UPDATE and fused is not None and x.is_cuda and x.dtype == th.float32 and x.numel() % 4 == 0:
                    v = self.drift(x, t, model, **kw)
                    te = t[:1].float().cpu().view(1, 1)
                    rar, var = ps._score_coeffs(te)
                    diff = float(ps.compute_diffusion(te, te.view(1), form=fused[1], norm=fused[2]))
                    return ops.sde_euler_step(x.contiguous(), v.contiguous(), None, float(rar), float(var), diff, float(last_step_size), 0.0, 0.0)[0]
                return x + sde_drift(x, t, model, **kw) * last_step_size
            return mean_step
        if last_step == "Tweedie":
            alpha, sigma = ps.compute_alpha_t, ps.compute_sigma_t
            return lambda x, t, model, **kw: x / alpha(t)[0][0] + (sigma(t)[0][0] ** 2) / alpha(t)[0][0] * self.score(x, t, model, **kw)
        if last_step == "Euler":
            return lambda x, t, model, **kw: x + self.drift(x, t, model, **kw) * last_step_size
        raise NotImplementedError()

    def sample_sde(self, *, sampling_method="Euler", diffusion_form="SBDM", diffusion_norm=1.0, last_step="Mean", last_step_size=0.04, num_steps=250):
        """-> `sample_fn(init, model, **model_kwargs)` returning the list of `num_steps` states (transport.py:298-354)."""
        if last_step is None:
            last_step_size = 0.0
        sde_drift, sde_diffusion = self._sde_diffusion_and_drift(diffusion_form=diffusion_form, diffusion_norm=diffusion_norm)
        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, diffusion_form=diffusion_form, sde=True, eval=True,
                                               reverse=False, last_step_size=last_step_size)
        fused = (self.transport.path_sampler, diffusion_form, diffusion_norm) if self.transport.model_type == ModelType.VELOCITY else None
        _sde = sde(sde_drift, sde_diffusion, t0=t0, t1=t1, num_steps=num_steps, sampler_type=sampling_method, fused=fused)
        last_step_fn = self._last_step(sde_drift, last_step=last_step, last_step_size=last_step_size, fused=fused)

        def _sample(init, model, **model_kwargs):
            xs = _sde.sample(init, model, **model_kwargs)
            ts = th.ones(init.size(0), device=init.device) * t1
            xs.append(last_step_fn(xs[-1], ts, model, **model_kwargs))
            assert len(xs) == num_steps, "Samples does not match the number of steps"
            return xs

        return _sample

    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False):
        """transport.py:356-407."""
        if reverse:
            drift = lambda x, t, model, **kw: self.drift(x, th.ones_like(t) * (1 - t), model, **kw)
        else:
            drift = self.drift
        t0, t1 = self.transport.check_interval(self.transport.train_eps, self.transport.sample_eps, sde=False, eval=True, reverse=reverse,
                                               last_step_size=0.0)
        return ode(drift=drift, t0=t0, t1=t1, sampler_type=sampling_method, num_steps=num_steps, atol=atol, rtol=rtol,
                   time_dist_shift=self.transport.time_dist_shift).sample
