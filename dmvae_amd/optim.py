"""Flat-buffer optimiser tail: clip_grad_norm_(1.0) + AdamW(betas=(.9,.95), eps=1e-8, wd=.005) + EMA + LambdaLR warm-up
(train_tokenizer.py:140-150,382-392,415-419,437) as two HIP launches over one contiguous f32 parameter buffer.

`FlatParams.flatten` re-homes the given parameters (and their .grad) as views into flat buffers, in the order given
(use backward-completion order so FlatGradSync's buckets fill front to back).  state_dict()/load_state_dict() of the
owning module keep working: the views are ordinary nn.Parameters."""
from __future__ import annotations

from typing import List, Sequence

import os

import torch

from . import functional as Fn
from . import ops


_CHECK_DIRECT = os.environ.get("DMVAE_CHECK_DIRECT_GRADS", "0") not in ("", "0")


class FlatParams:
    def __init__(self, params: Sequence[torch.nn.Parameter], with_ema: bool = True):
        self.params: List[torch.nn.Parameter] = list(params)
        assert all(p.dtype == torch.float32 for p in self.params)
        dev = self.params[0].device
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every tensor 16-B aligned
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            self.flat[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + p.numel()].view(p.shape)
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
        self.ema = self.flat.clone() if with_ema else None
        self.epoch = [0]                      # bumped whenever the flat buffer is updated through raw pointers (functional.packed / _bf caches)
        self.pack_reg = Fn.new_pack_registry()   # the packed conv operands of these weights: refreshed by ONE launch after every optimiser step (functional.repack_all)
        for p in self.params:
            p._dmvae_epoch = self.epoch
            p._dmvae_pack_reg = self.pack_reg

    def zero_grad(self):
        self.grad.zero_()

    def enable_bf16_shadow(self) -> None:
        """Keep a bf16 copy of the whole flat buffer that the fused optimiser step refreshes in the pass that writes the new weights
        (`dmvae_adamw_ema_step_shadow`); `functional._bf` then hands GEMMs views of it instead of converting every Linear weight after every
        step (the DMD stage: ~680 conversion launches per step over ViT-L + DiT-XL).  A parameter changed by anything else (load_state_dict,
        an in-place torch op: `_version` moves) is re-converted on its next use."""
        self.shadow = torch.empty(self.numel, dtype=torch.bfloat16, device=self.flat.device)
        self.shadow.copy_(self.flat)
        for p, off in zip(self.params, self.offsets):
            p._dmvae_shadow = self.shadow[off:off + p.numel()].view(p.shape)
            p._dmvae_shadow_ver = (p.data_ptr(), p._version)

    def enable_transposed_shadow(self, only=None) -> None:
        """Keep, next to the bf16 shadow, the K-tile-major bf16 copy of the TRANSPOSE of every 2-D Linear weight [out, in] (out % 32 == 0, in % 8 == 0) -- the operand
        of its input gradient dX = dY . W (`functional._bf_t`) -- refreshed by ONE batched transpose launch after every optimiser step (`refresh_transposed`), instead
        of one ~6.6-us launch per weight on its first use in the backward pass (142 per step for LightningDiT-XL/1).  Needs `enable_bf16_shadow` first."""
        assert getattr(self, "shadow", None) is not None, "enable_bf16_shadow() first"
        ids = None if only is None else {id(p) for p in only}
        self.shadow_t = torch.empty(self.numel, dtype=torch.bfloat16, device=self.flat.device)
        self._t_pairs, self._t_params = [], []
        for p, off in zip(self.params, self.offsets):
            if p.dim() == 2 and p.shape[0] % 32 == 0 and p.shape[1] % 8 == 0 and (ids is None or id(p) in ids):
                dst = self.shadow_t[off:off + p.numel()].view(p.shape[0] // 32, p.shape[1], 32)
                self._t_pairs.append((self.shadow[off:off + p.numel()].view(p.shape), dst))
                self._t_params.append(p)
                p._dmvae_shadow_t = dst
                p._dmvae_shadow_t_key = None
        self.refresh_transposed()

    def refresh_transposed(self) -> None:
        if not getattr(self, "_t_pairs", None):
            return
        ops.linear_weight_t_kmajor_batched(self._t_pairs)
        for p in self._t_params:
            p._dmvae_shadow_t_key = (p.data_ptr(), p._version, self.epoch[0])

    def enable_direct_grads(self, only=None) -> None:
        """The HIP backward Functions (dmvae_amd.functional) then WRITE each parameter gradient straight into its slice of the
        flat buffer and hand that view to autograd, instead of returning a fresh tensor that AccumulateGrad adds into `.grad`
        (193 small add kernels + one memset per step for the tokenizer).  Requires every parameter to receive exactly one
        gradient per backward (true for the decoder / bottleneck); call `begin_step()` before each backward."""
        ids = None if only is None else {id(p) for p in only}
        for p, off in zip(self.params, self.offsets):
            if ids is None or id(p) in ids:
                p._dmvae_grad_view = self.grad[off:off + p.numel()].view(p.shape)
        self.direct = True
        self.partial = ids is not None        # the other parameters keep accumulating into their (zeroed) flat views through autograd
        # what begin_step has to zero in that case: the slices of the parameters WITHOUT a direct view, as maximal contiguous runs (the student DiT: the embedders
        # in front of the blocks and the output layer behind them -- zeroing all 2.7 GB of its gradient buffer was 0.43 ms per step)
        runs = []
        for p, off in zip(self.params, self.offsets):
            if not hasattr(p, "_dmvae_grad_view"):
                end = off + (p.numel() + 3) // 4 * 4
                if runs and runs[-1][1] == off:
                    runs[-1][1] = end
                else:
                    runs.append([off, end])
        self.accum_runs = [(a, min(b, self.numel)) for a, b in runs]

    def begin_step(self) -> None:
        """Start of a step.  INVARIANT of the direct-gradient mode: every parameter with a `_dmvae_grad_view` is WRITTEN (not accumulated into) exactly once per
        backward by the Function that owns it, so its slice of the gradient buffer is not zeroed here -- a direct parameter that a backward skips (an unused or
        conditional block, a Function returning None) would keep the previous step's gradient and AdamW / grad_norm would consume it silently.  No such
        parameter exists in the VAE or the DiT routes; DMVAE_CHECK_DIRECT_GRADS=1 (or `check_direct_writes`) verifies it per step, and a trainer that cannot
        guarantee it sets `self.zero_all = True` to fall back to zeroing the whole buffer."""
        opt = getattr(self, "_opt", None)
        if opt is not None:
            opt.wait()            # an overlapped optimiser step still reads the gradient buffer
        if _CHECK_DIRECT and getattr(self, "direct", False):
            self.check_direct_writes()
        if getattr(self, "direct", False):
            if getattr(self, "zero_all", False):
                self.grad.zero_()
            elif getattr(self, "partial", False):
                if len(self.accum_runs) <= 16:
                    for a, b in self.accum_runs:
                        self.grad[a:b].zero_()
                else:
                    self.grad.zero_()
            for p, off in zip(self.params, self.offsets):
                if hasattr(p, "_dmvae_grad_view"):
                    p.grad = None      # autograd adopts the returned flat-buffer view (no accumulation kernel)
                elif p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + off * 4:
                    p.grad = self.grad[off:off + p.numel()].view(p.shape)
        else:
            self.zero_grad()

    def check_direct_writes(self) -> None:
        """Debug mode of the invariant above (DMVAE_CHECK_DIRECT_GRADS=1, or called by hand at the start of a step): raises if the backward(s) since the previous
        call handed out the destination view of some direct parameter of THIS buffer zero times while others of it were written (functional._dst counts;
        a step in which none of this buffer's parameters took part -- another trainer's turn -- is not judged)."""
        from . import functional
        if functional.DIRECT_GRAD_WRITES is None:
            functional.DIRECT_GRAD_WRITES = {}
            return
        seen = functional.DIRECT_GRAD_WRITES
        mine = [(i, seen.pop(id(p), 0)) for i, p in enumerate(self.params) if hasattr(p, "_dmvae_grad_view")]
        if not any(n for _, n in mine):
            return
        bad = [i for i, n in mine if n == 0]
        if bad:
            raise RuntimeError(f"FlatParams: {len(bad)} direct-gradient parameter(s) were not written by the last backward (indices {bad[:8]} ...): "
                               "their gradient slices hold the previous step's values; set `zero_all = True` on this FlatParams or exclude them from enable_direct_grads")

    def after_external_update(self, reset_ema: bool = False) -> None:
        """The flat buffer was written by something other than the fused optimiser step (initial broadcast from rank 0, checkpoint load):
        invalidate the cached bf16 / packed operands of these parameters, refresh the bf16 shadow, optionally restart the EMA from the weights."""
        self.epoch[0] += 1
        if getattr(self, "shadow", None) is not None:
            self.shadow.copy_(self.flat)
            for p in self.params:
                p._dmvae_shadow_ver = (p.data_ptr(), p._version)
            self.refresh_transposed()
        if reset_ema and self.ema is not None:
            self.ema.copy_(self.flat)

    def ema_state(self):
        """name-less list of EMA tensors aligned with self.params (views)."""
        return [self.ema[off:off + p.numel()].view(p.shape) for p, off in zip(self.params, self.offsets)]


class FlatAdamWEMA:
    def __init__(self, fp: FlatParams, lr=1e-4, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.005, max_norm=1.0, ema_decay=0.9999,
                 warmup_steps=1000):
        self.fp = fp
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_norm, self.ema_decay, self.warmup_steps = max_norm, ema_decay, warmup_steps
        self.exp_avg = torch.zeros_like(fp.flat)
        self.exp_avg_sq = torch.zeros_like(fp.flat)
        self.norm = torch.zeros(3, dtype=torch.float32, device=fp.flat.device)   # [norm, clip coef, sumsq] stays on device
        self.t = 0
        self.side, self._done, self._then = None, None, None
        fp._opt = self

    _SIDE = {}      # device index -> THE side stream of this process: every overlapped optimiser shares it

    def enable_overlap(self) -> None:
        """Run `step` on a side HIP stream: the HBM-bound update (norm pass + fused AdamW + the operand refreshes: 20 GB of traffic for LightningDiT-XL/1) then overlaps
        whatever the caller's stream does next that does not touch these weights -- the next step's encoder forward, another model's turn -- instead of serialising
        behind the backward pass.  Everything that reads the weights (or their bf16 / packed / transposed operands) or writes the gradient buffer must come after
        `wait()`: `FlatParams.begin_step` does it, the trainers do it before a forward of the model.

        Two rules keep this race-free.  ONE side stream per device for all optimisers: the updates of two optimisers are ordered against each
        other, so the `ops.workspace()` slots they share (norm partials, repack / transpose tables; keyed by stream) are never written by two streams at once.
        And `step(then=...)` callbacks are NOT run on the side stream: they read tensors the caller's stream allocated (losses, norms of other terms), which the
        caching allocator may hand out again once the closure is gone while the side stream is still reading; they run in `wait()`, on the caller's stream, behind
        the event of the update they belong to -- the closure keeps its tensors alive until then."""
        if self.fp.flat.is_cuda and self.side is None:
            dev = self.fp.flat.device
            key = dev.index if dev.index is not None else torch.cuda.current_device()
            if key not in FlatAdamWEMA._SIDE:
                FlatAdamWEMA._SIDE[key] = torch.cuda.Stream(device=dev)
            self.side = FlatAdamWEMA._SIDE[key]

    def wait(self) -> None:
        """The caller's current stream waits for the last overlapped `step`, then runs that step's deferred `then` callback (no-op without one)."""
        if self._done is not None:
            torch.cuda.current_stream(self.fp.flat.device).wait_event(self._done)
            self._done = None
        then, self._then = self._then, None
        if then is not None:
            then(self.norm)

    def current_lr(self) -> float:
        """The reference's LambdaLR (train_tokenizer.py:385-392): lr_lambda(s) = s / warmup if s < warmup else 1, evaluated at the
        number of completed steps -- the first optimiser step runs at lr 0."""
        if self.t < self.warmup_steps:
            return self.lr * (self.t / self.warmup_steps)
        return self.lr

    def step(self, then=None):
        """One optimiser step; `then(norm)` (optional) runs right behind it on the caller's stream.  With `enable_overlap` the step goes to the side stream, ordered
        after everything the current stream has queued so far (the backward pass, the gradient all-reduce's wait), and `then` is deferred to `wait()`."""
        if self.side is None:
            norm = self._step()
            if then is not None:
                then(norm)
            return norm
        self.wait()
        self.side.wait_stream(torch.cuda.current_stream(self.fp.flat.device))
        with torch.cuda.stream(self.side):
            norm = self._step()
            self._done = torch.cuda.Event()
            self._done.record(self.side)
        self._then = then
        return norm

    def _step(self):
        lr = self.current_lr()
        self.t += 1
        ops.grad_norm(self.fp.grad, self.max_norm, norm_out=self.norm)
        ops.adamw_ema_step(self.fp.flat, self.fp.grad, self.exp_avg, self.exp_avg_sq, self.fp.ema, self.norm if self.max_norm > 0 else None,
                           lr, self.betas[0], self.betas[1], self.eps, self.wd, self.t, self.ema_decay, shadow=getattr(self.fp, "shadow", None))
        self.fp.epoch[0] += 1       # weights changed through raw pointers: the cached bf16 operands of THESE parameters are stale ...
        Fn.repack_all(self.fp.pack_reg, self.fp.epoch)     # ... and the packed conv operands among them are rewritten here, in one launch
        self.fp.refresh_transposed()                       # ... like the transposed Linear operands (when enabled)
        return self.norm            # device tensor; no host sync

    # ---- checkpoint / resume (train_tokenizer.py:440-450 saves opt_vae / opt_disc / scheduler_*; train_dmd.py:577-590; train_diffusion.py:318-325) ----
    def state_dict(self, param_order=None) -> dict:
        """torch.optim.AdamW's layout, so that the entry is interchangeable with the reference's `opt_*` checkpoint entries:
        {"state": {index: {"step", "exp_avg", "exp_avg_sq"}}, "param_groups": [{..., "params": [indices]}]} with index = position of the parameter in
        `param_order` -- the list the reference built THAT optimiser over: the trainable parameters in train_tokenizer.py:381-382, every parameter in
        train_dmd.py:473-475 / train_diffusion.py:209 (frozen ones then simply have no state) -- default: this buffer's own order.  The LambdaLR position is the step counter (`scheduler_state_dict`)."""
        order = list(param_order) if param_order is not None else self.fp.params
        index = {id(p): i for i, p in enumerate(order)}
        state = {}
        for p, off in zip(self.fp.params, self.fp.offsets):
            if self.t > 0:
                n = p.numel()
                state[index[id(p)]] = {"step": torch.tensor(float(self.t)), "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                                       "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        group = {"lr": self.current_lr(), "initial_lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict, param_order=None) -> None:
        """Inverse of `state_dict` (also accepts a reference `torch.optim.AdamW.state_dict()` taken over `param_order`)."""
        order = list(param_order) if param_order is not None else self.fp.params
        index = {id(p): i for i, p in enumerate(order)}
        missing = [p for p in self.fp.params if id(p) not in index]
        if missing:
            raise ValueError(f"FlatAdamWEMA.load_state_dict: {len(missing)} of this optimiser's parameters are not in param_order")
        g_params = sd["param_groups"][0].get("params")
        if g_params is not None and len(g_params) != len(order):
            raise ValueError(f"FlatAdamWEMA.load_state_dict: the entry was taken over {len(g_params)} parameters, param_order has {len(order)} -- "
                             "the optimiser it comes from was built over another parameter list (e.g. trainable-only vs all parameters)")
        have = [sd["state"].get(index[id(p)]) is not None for p in self.fp.params]
        if any(have) and not all(have):
            raise ValueError(f"FlatAdamWEMA.load_state_dict: {have.count(False)} of {len(have)} trainable parameters have no state while the others do")
        steps = set()
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for p, off in zip(self.fp.params, self.fp.offsets):
            st = sd["state"].get(index[id(p)])
            if st is None:
                continue
            n = p.numel()
            if st["exp_avg"].numel() != n:
                raise ValueError(f"FlatAdamWEMA.load_state_dict: state {index[id(p)]} has {st['exp_avg'].numel()} elements, the parameter {n}")
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FlatAdamWEMA.load_state_dict: parameters carry different step counts {sorted(steps)}; one fused update has one counter")
        self.t = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        self.lr = g.get("initial_lr", self.lr)
        self.betas, self.eps, self.wd = tuple(g.get("betas", self.betas)), g.get("eps", self.eps), g.get("weight_decay", self.wd)

    def scheduler_state_dict(self) -> dict:
        """What torch's LambdaLR.state_dict() carries that matters on resume: the number of completed scheduler steps."""
        return {"last_epoch": self.t, "base_lrs": [self.lr], "_last_lr": [self.current_lr()]}
