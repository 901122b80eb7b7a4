"""torch.distributed plumbing for the data-parallel DMVAE step: one process per GPU, RCCL over xGMI
(backend string "nccl" IS RCCL on PyTorch-ROCm), gloo on CPU for tests.

Mirrors the parts of the reference's utils/dist.py the hot path touches (init_distributed_mode :228-239,
allreduce/barrier :106-170 degrade to no-ops when uninitialised) and replaces DistributedDataParallel
(train_tokenizer.py:302) by `FlatGradSync`: all trainable parameters' gradients live in ONE flat f32 buffer laid
out in backward-completion order; contiguous buckets are all-reduced asynchronously on RCCL's stream as soon as
their last gradient has been accumulated, overlapping the rest of backward.  A few large collectives instead of
DDP's 25 MB default: xGMI is point-to-point, per-link bound, so fewer/larger messages amortise launch + sync.
"""
from __future__ import annotations

import datetime
import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as tdist

__all__ = ["init_distributed_mode", "initialized", "get_rank", "get_world_size", "get_local_rank", "barrier", "allreduce",
           "is_master", "FlatGradSync", "broadcast_tensors", "broadcast_module_state", "require_initialized"]

_initialized = False
_rank, _world, _local_rank = 0, 1, 0


def init_distributed_mode(backend: Optional[str] = None, timeout_minutes: int = 30) -> None:
    """RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the env (torchrun).  No env -> single process, collectives are no-ops."""
    global _initialized, _rank, _world, _local_rank
    force = os.environ.get("DMVAE_FORCE_DIST", "0") != "0"   # diagnostics: run the collective path on a single rank
    if "RANK" not in os.environ or (int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not force):
        if torch.cuda.is_available():
            torch.cuda.set_device(0)
        return
    _rank, _world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    _local_rank = int(os.environ.get("LOCAL_RANK", _rank))
    if _world > 1:
        # From here on the collectives' kernels share the GPU with the persistent conv blocks, which need a whole CU each (128-160 KB of LDS): claim tiles
        # dynamically so that a launch slows down by the fraction of CUs taken instead of a whole block-time (csrc/conv_pp.hip, DYN; measured with a resident
        # side-stream kernel on one GPU, tools/probes/contention.py: 16 CUs taken cost +6..+38 % instead of +48..+75 %; 0.3-1.4 % slower with no contention, which
        # is why a single rank keeps the static stride).  One process-wide switch in the library (dmvae_set_dynamic), so it does not matter which convs have
        # already run; DMVAE_PP_DYNAMIC=0 in the environment keeps it off (A/B runs).  The variable is also set for ops._conv_label / bench.py, which name the
        # kernel a launch dispatches to.
        if os.environ.get("DMVAE_PP_DYNAMIC", "1") != "0":
            os.environ["DMVAE_PP_DYNAMIC"] = "1"
            if torch.cuda.is_available():
                from . import _lib
                _lib.lib().dmvae_set_dynamic(1)
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(_local_rank % torch.cuda.device_count())
    if backend is None:       # DMVAE_DIST_BACKEND=gloo: diagnostics (several ranks sharing one GPU, which RCCL refuses)
        backend = os.environ.get("DMVAE_DIST_BACKEND") or ("nccl" if use_gpu else "gloo")
    if not tdist.is_initialized():
        kw = {}
        if use_gpu and backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        tdist.init_process_group(backend=backend, timeout=datetime.timedelta(minutes=timeout_minutes), **kw)
    _initialized = True


def initialized() -> bool:
    return _initialized


def get_rank() -> int:
    return _rank


def get_world_size() -> int:
    return _world


def get_local_rank() -> int:
    return _local_rank


def is_master() -> bool:
    return _rank == 0


def barrier() -> None:
    if _initialized:
        tdist.barrier()


def allreduce(t: torch.Tensor, async_op: bool = False, op=None):
    if _initialized:
        return tdist.all_reduce(t, op=op or tdist.ReduceOp.SUM, async_op=async_op)
    return None


def require_initialized(what: str) -> None:
    """A launcher exported WORLD_SIZE > 1 but `init_distributed_mode()` has not run: anything that latches "is this job distributed?" at
    construction time (FlatGradSync, the trainers' initial broadcast) would silently train every rank on its own gradients.  Fail loudly."""
    if not _initialized and int(os.environ.get("WORLD_SIZE", "1")) > 1 and "RANK" in os.environ:
        raise RuntimeError(f"{what}: WORLD_SIZE={os.environ['WORLD_SIZE']} but dmvae_amd.dist.init_distributed_mode() has not been called; "
                           "call it before building models / trainers (gradients would not be synchronised)")


def broadcast_tensors(tensors: Sequence[torch.Tensor], src: int = 0, chunk_bytes: int = 256 << 20) -> int:
    """Rank `src`'s values of `tensors` -> every rank, in place.  Small tensors are coalesced per dtype into staging buffers of at most
    `chunk_bytes` (one collective per ~256 MB instead of one per tensor: xGMI links are per-message bound); a tensor that is already large and
    contiguous is broadcast as it is.  Returns the number of collectives issued (0 when the job is not distributed)."""
    if not _initialized or _world <= 1:
        return 0
    calls = 0
    by_dtype = {}
    for t in tensors:
        if t.numel() == 0:
            continue
        if t.is_contiguous() and t.numel() * t.element_size() >= chunk_bytes // 4:
            tdist.broadcast(t, src=src)
            calls += 1
        else:
            by_dtype.setdefault((t.dtype, t.device), []).append(t)
    for (dtype, device), group in by_dtype.items():
        i = 0
        while i < len(group):
            j, n = i, 0
            while j < len(group) and (j == i or (n + group[j].numel()) * group[j].element_size() <= chunk_bytes):
                n += group[j].numel()
                j += 1
            stage = torch.empty(n, dtype=dtype, device=device)
            o = 0
            for t in group[i:j]:
                stage[o:o + t.numel()].copy_(t.detach().reshape(-1))
                o += t.numel()
            tdist.broadcast(stage, src=src)
            calls += 1
            o = 0
            with torch.no_grad():
                for t in group[i:j]:
                    t.copy_(stage[o:o + t.numel()].view(t.shape))
                    o += t.numel()
            i = j
    return calls


def broadcast_module_state(*modules, extra: Sequence[torch.Tensor] = (), src: int = 0) -> int:
    """What DistributedDataParallel's constructor does for the reference (train_tokenizer.py:302,319; train_dmd.py:348,355: `_sync_module_states`
    over every parameter AND buffer of the wrapped module, trainable or not): after it every rank holds rank 0's weights, BatchNorm running
    statistics included, whatever each rank's RNG produced at construction.  `extra`: optimiser / EMA state kept outside the modules.
    Tensors that alias each other (parameters re-homed as views of a flat buffer) are sent once."""
    seen, todo = set(), []
    for t in extra:
        if t is not None and t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            todo.append(t)
    covered = [(t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in todo]
    for m in modules:
        if m is None:
            continue
        for t in list(m.parameters()) + list(m.buffers()):
            a = t.data_ptr()
            if a in seen or any(lo <= a < hi for lo, hi in covered):
                continue
            seen.add(a)
            todo.append(t.data if isinstance(t, torch.nn.Parameter) else t)
    return broadcast_tensors(todo, src=src)


class FlatGradSync:
    """Bucketed asynchronous gradient averaging over a flat gradient buffer.

    params   : trainable parameters in the order their gradients become ready in backward (first-ready first);
               their .grad are views into `flat_grad` in that same order.
    The hook on each parameter fires after its gradient has been accumulated; when every parameter of a bucket
    has fired, the bucket's slice is averaged over the ranks with async_op=True: ReduceOp.AVG under RCCL (the division happens inside the collective), divide +
    SUM under gloo.
    `wait()` must be called before the optimiser reads the gradients.
    """

    def __init__(self, params: Sequence[torch.nn.Parameter], flat_grad: torch.Tensor, offsets: Sequence[int],
                 bucket_bytes: int = 64 << 20):
        require_initialized("FlatGradSync")
        self.flat_grad = flat_grad
        self.world = get_world_size()
        self.enabled = initialized() and (self.world > 1 or os.environ.get("DMVAE_FORCE_DIST", "0") != "0")
        self.buckets: List[List[int]] = []      # [start, end, n_params]
        self.param_bucket: List[int] = []
        cur_start, cur_n = 0, 0
        for i, p in enumerate(params):
            self.param_bucket.append(len(self.buckets))
            cur_n += 1
            end = offsets[i] + p.numel()
            if (end - cur_start) * 4 >= bucket_bytes or i == len(params) - 1:
                self.buckets.append([cur_start, end, cur_n])
                cur_start, cur_n = end, 0
        self._pending = [b[2] for b in self.buckets]
        self._works = []
        self._handles = []
        # the mean over ranks: RCCL / NCCL average INSIDE the collective (ncclAvg) -- no `div_` launch per bucket on the compute stream (42 per student step of the
        # DMD stage); gloo has no AVG: divide, then SUM (the CPU tests' transport)
        self.avg_in_collective = bool(self.enabled and tdist.get_backend() == "nccl" and os.environ.get("DMVAE_DIST_AVG", "1") != "0")
        self.hook_launches = 0      # buckets whose all-reduce was started from a gradient hook, i.e. DURING backward, in the last step
        self._in_wait = False
        self.time_wait = False      # bench.py: bracket wait() with events on the compute stream -- what of the collectives backward did NOT cover
        self._wait_events: List[tuple] = []
        self.last_hook_launches = 0
        if self.enabled:
            for i, p in enumerate(params):
                self._handles.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def _make_hook(self, i):
        def hook(_p):
            b = self.param_bucket[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        if not self._in_wait:
            self.hook_launches += 1
        s, e, _ = self.buckets[b]
        sl = self.flat_grad[s:e]
        if self.avg_in_collective:
            self._works.append(tdist.all_reduce(sl, op=tdist.ReduceOp.AVG, async_op=True))
            return
        sl.div_(self.world)
        self._works.append(tdist.all_reduce(sl, op=tdist.ReduceOp.SUM, async_op=True))

    def wait(self) -> None:
        """Block the current stream until every bucket's all-reduce has completed; re-arm for the next backward."""
        if not self.enabled:
            return
        self._in_wait = True
        for b, left in enumerate(self._pending):      # parameters that received no gradient this step
            if left > 0:
                self._launch(b)
        self._in_wait = False
        timed = self.time_wait and self.flat_grad.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for w in self._works:
            w.wait()
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))
        self._works.clear()
        self._pending = [b[2] for b in self.buckets]
        self.last_hook_launches, self.hook_launches = self.hook_launches, 0

    def comm_stats(self) -> dict:
        """For the bench line of an N > 1 run: gradient bytes all-reduced per step, bucket count, how many buckets' collectives were launched from gradient
        hooks DURING the last backward (the rest start in wait()), and the mean time the compute stream spent in wait() -- the exposed, un-overlapped part
        (needs `time_wait = True` before the timed steps; synchronises)."""
        ms = None
        if self._wait_events:
            torch.cuda.synchronize()
            v = [a.elapsed_time(b) for a, b in self._wait_events]
            ms = sum(v) / len(v)
            self._wait_events.clear()
        return {"bytes": int(self.buckets[-1][1]) * 4 if self.buckets else 0, "buckets": len(self.buckets),
                "launched_in_backward": int(self.last_hook_launches), "wait_ms": None if ms is None else round(ms, 4),
                "mean": "ncclAvg inside the all-reduce" if self.avg_in_collective else "div_ per bucket, then SUM"}

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles.clear()
