"""fp32 parity mode of the hot path (DMVAE_PARITY=1, or `with parity.enabled():`): f32 NHWC activations end to end.

north_star's bar is "within 1e-4 relative fp32 of the reference PyTorch-CPU path".  The production path stores activations in bf16 (2^-9 per
rounding), so it can meet that bar only against an oracle fed the same rounded operands.  In this mode nothing is rounded to bf16 for storage and
every contraction STILL runs on the production MFMA kernels (csrc/conv_pp.hip, conv_fwd.hip, conv_wgrad_pp.hip, conv_wgrad.hip): each f32
operand is split exactly into three bf16 terms (csrc/parity.hip::split3_kernel) and the six partial products of order <= 2 are laid out along the
reduction dimension of ONE launch, whose f32 accumulator then holds x*w to ~2^-24.  The HBM-bound elementwise / normalisation steps use the
f32-in / f32-out kernels of csrc/parity.hip.  `dmvae_amd.ops` dispatches here when the mode is on; `dmvae_amd.functional` is unchanged -- the same
hand-scheduled forward / backward sequences run in both modes.  ~6x the MFMA work and 3x the activation bytes: a verification mode, not a
training mode.  Tests: tests/test_gpu_parity_fp32.py (every tolerance there is 1e-4 against the reference's own f32 captures).

Reference sites restated by the call sequences this serves: models/flux_ae.py:37-107,239-269, models/vae.py:56-65,90-98, utils/lpips.py:81-162,
train_tokenizer.py:179-204,403-437."""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch

from . import _lib
from ._lib import check

bf16 = torch.bfloat16
f32 = torch.float32

_ON = os.environ.get("DMVAE_PARITY", "0") not in ("", "0")


def on() -> bool:
    return _ON


def set_enabled(flag: bool) -> None:
    global _ON
    _ON = bool(flag)


@contextlib.contextmanager
def enabled(flag: bool = True):
    """Scope the mode (tests).  Build / load modules as usual; cached weight operands are keyed on the mode."""
    global _ON
    prev, _ON = _ON, bool(flag)
    try:
        yield
    finally:
        _ON = prev


def act_dtype():
    """Storage type of activations between kernels: f32 in parity mode, bf16 otherwise."""
    return f32 if _ON else bf16


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _f(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.DmvaeHipError(f"{name}: expected a GPU tensor; dmvae_amd has no CPU path")
    if t.dtype != f32:
        raise TypeError(f"{name}: parity mode expects float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


A_SIDE, W_SIDE = 0, 1      # split patterns: [hi, mid, lo, hi, mid, hi] x [hi, hi, hi, mid, mid, lo]


def _split(x, out, rows, cols, rpb, batch_stride, part_stride, row_stride, pattern):
    check(_lib.lib().dmvae_split3_bf16(x.data_ptr(), out.data_ptr(), rows, cols, rpb, batch_stride, part_stride, row_stride, pattern, _stream()),
          "split3_bf16")
    return out


def split_channels(x: torch.Tensor, pattern: int) -> torch.Tensor:
    """[..., C] f32 -> [..., 6C] bf16: the six parts side by side along the (reduction) channel axis."""
    x = _f(x, "x")
    c = x.shape[-1]
    rows = x.numel() // c
    out = torch.empty(*x.shape[:-1], 6 * c, dtype=bf16, device=x.device)
    return _split(x, out, rows, c, rows, 0, c, 6 * c, pattern)


def split_batch(x: torch.Tensor, pattern: int) -> torch.Tensor:
    """[N, ...] f32 -> [6N, ...] bf16: the six parts as extra images (the weight gradient reduces over images x pixels)."""
    x = _f(x, "x")
    c = x.shape[-1]
    rows = x.numel() // c
    out = torch.empty(6 * x.shape[0], *x.shape[1:], dtype=bf16, device=x.device)
    return _split(x, out, rows, c, rows, 0, x.numel(), c, pattern)


def split_rows(x: torch.Tensor, pattern: int) -> torch.Tensor:
    """[B, K, M] f32 -> [B, 6K, M] bf16 (TN GEMM: the reduction runs over the slow dimension of each batch entry)."""
    x = _f(x, "x")
    b, k, m = x.shape
    out = torch.empty(b, 6 * k, m, dtype=bf16, device=x.device)
    return _split(x, out, b * k, m, k, 6 * k * m, k * m, m, pattern)


# ---- contractions on the production kernels ---------------------------------------------------------------------------------------------
def pack_conv_weight(w: torch.Tensor, for_dgrad: bool = False, rows_pad: int = 0, cols_pad: int = 0) -> torch.Tensor:
    """f32 [cout, cin, k, k] (or [out, in]) -> bf16 [rows, k*k, 6*cols]: dmvae_pack_conv_weight's layout with every tap's channel run replaced by
    the six weight-side parts.  Input gradient: rows = cin, taps flipped, cols = cout -- the same convention as the bf16 packer."""
    w = _f(w.detach(), "weight")
    if w.dim() == 2:
        w = w.view(w.shape[0], w.shape[1], 1, 1)
    cout, cin, ks, _ = w.shape
    wt = w.permute(1, 2, 3, 0).flip(1, 2) if for_dgrad else w.permute(0, 2, 3, 1)        # [rows, k, k, cols]: layout plumbing, values untouched
    rows, cols = wt.shape[0], wt.shape[3]
    rp, cp = max(rows_pad, rows), max(cols_pad, cols)
    if (rp, cp) != (rows, cols):
        full = torch.zeros(rp, ks, ks, cp, dtype=f32, device=w.device)
        full[:rows, :, :, :cols] = wt
        wt = full
    return split_channels(wt.contiguous(), W_SIDE).view(rp, ks * ks, 6 * cp)


def eltwise(op: int, a: torch.Tensor, b: Optional[torch.Tensor] = None, g: Optional[torch.Tensor] = None, act: int = 0, param: float = 0.0,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    a = _f(a, "a")
    if b is not None:
        b = _f(b, "b")
        assert b.shape == a.shape, (a.shape, b.shape)
    out = torch.empty_like(a) if out is None else out
    cols = a.shape[-1] if g is not None else 1
    check(_lib.lib().dmvae_eltwise_f32(op, a.data_ptr(), _ptr(b), _ptr(g), out.data_ptr(), a.numel(), cols, act, float(param), _stream()), "eltwise_f32")
    return out


def epilogue(y: torch.Tensor, residual: Optional[torch.Tensor], act: int) -> torch.Tensor:
    """act(y + residual) in f32 (ops.ACT_*: 1 SiLU, 2 ReLU, 3 ReLU gate by `residual`, 4 LeakyReLU 0.2) -- what the bf16 kernels do in their epilogue."""
    if residual is None and act == 0:
        return y
    return eltwise(0, y, residual, act=act, param=0.2, out=y)


def colsum(x2: torch.Tensor) -> torch.Tensor:
    """[rows, cols] f32 -> [cols]: the bias gradient.  On the TN GEMM kernel: sum_r x[r, c] = (ones^T x)[c], with the split parts of x paired with a
    [1, 1, 1, 0, 0, 0] mask so that hi + mid + lo (= x exactly) is what gets summed."""
    from . import ops
    rows, cols = x2.shape
    xs = split_rows(x2.view(1, rows, cols), A_SIDE)                     # [1, 6 rows, cols]
    ones = torch.zeros(1, 6 * rows, 8, dtype=bf16, device=x2.device)
    ones[:, :3 * rows] = 1
    return ops.gemm_tn(ones, xs, out_f32=True)[0, 0].contiguous()        # [8, cols] -> row 0


def transpose_last2(x: torch.Tensor) -> torch.Tensor:
    return _f(x, "x").transpose(-1, -2).contiguous()                    # layout only


# ---- GroupNorm --------------------------------------------------------------------------------------------------------------------------
def groupnorm_stats(x: torch.Tensor, groups: int = 32, eps: float = 1e-6) -> torch.Tensor:
    x = _f(x, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    stats = torch.empty(n, groups, 2, dtype=f32, device=x.device)
    check(_lib.lib().dmvae_groupnorm_stats_f32(x.data_ptr(), stats.data_ptr(), n, hw, c, groups, float(eps), _stream()), "groupnorm_stats_f32")
    return stats


def groupnorm_apply(x, stats, gamma, beta, act: int, groups: int = 32) -> torch.Tensor:
    x = _f(x, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty_like(x)
    check(_lib.lib().dmvae_groupnorm_apply_f32(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), n, hw, c, groups, int(act),
                                               _stream()), "groupnorm_apply_f32")
    return y


def groupnorm_bwd(da, x, stats, gamma, beta, act: int, dres=None, groups: int = 32, need_param_grads: bool = True, dg_out=None, db_out=None,
                  inv_count: float = 0.0):
    from . import ops
    da, x = _f(da, "da"), _f(x, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    L = _lib.lib()
    ws = ops.workspace(L.dmvae_groupnorm_f32_workspace(n, c, groups), x.device, slot="gn32")
    dx = torch.empty_like(x)
    dg = (dg_out if dg_out is not None else torch.empty(c, dtype=f32, device=x.device)) if need_param_grads else None
    db = (db_out if db_out is not None else torch.empty(c, dtype=f32, device=x.device)) if need_param_grads else None
    if dres is not None:
        dres = _f(dres, "dres")
    check(L.dmvae_groupnorm_bwd_f32(da.data_ptr(), x.data_ptr(), _ptr(dres), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(), _ptr(dg),
                                    _ptr(db), ws.data_ptr(), ws.numel(), n, hw, c, groups, int(act), 0, float(inv_count), _stream()), "groupnorm_bwd_f32")
    return dx, dg, db


# ---- softmax / pools / layout / losses ---------------------------------------------------------------------------------------------------------
def softmax_rows(s: torch.Tensor, scale: float) -> torch.Tensor:
    s = _f(s, "S")
    p = torch.empty_like(s)
    check(_lib.lib().dmvae_softmax_rows_fwd_f32(s.data_ptr(), p.data_ptr(), s.numel() // s.shape[-1], s.shape[-1], float(scale), _stream()),
          "softmax_rows_fwd_f32")
    return p


def softmax_rows_bwd(dp: torch.Tensor, p: torch.Tensor, scale: float) -> torch.Tensor:
    dp, p = _f(dp, "dP"), _f(p, "P")
    ds = torch.empty_like(p)
    check(_lib.lib().dmvae_softmax_rows_bwd_f32(dp.data_ptr(), p.data_ptr(), ds.data_ptr(), p.numel() // p.shape[-1], p.shape[-1], float(scale), _stream()),
          "softmax_rows_bwd_f32")
    return ds


def pool2x2(op: int, a, x=None, extra=None) -> torch.Tensor:
    src = _f(x if op == 2 else a, "x")
    n, h2, w2, c = src.shape
    out = torch.empty_like(src) if op == 2 else torch.empty(n, h2 // 2, w2 // 2, c, dtype=f32, device=src.device)
    a = None if a is None else _f(a, "a")
    extra = None if extra is None else _f(extra, "extra")
    check(_lib.lib().dmvae_pool2x2_f32(op, _ptr(a), _ptr(x), _ptr(extra), out.data_ptr(), n, h2 // 2, w2 // 2, c, _stream()), "pool2x2_f32")
    return out


def nchw_to_nhwc(x: torch.Tensor, c_pad: int = 0) -> torch.Tensor:
    x = _f(x, "x")
    n, c, h, w = x.shape
    c_pad = max(c_pad, c)
    out = torch.empty(n, h, w, c_pad, dtype=f32, device=x.device)
    check(_lib.lib().dmvae_nchw_f32_to_nhwc_f32(x.data_ptr(), out.data_ptr(), n, c, h * w, c_pad, _stream()), "nchw_f32_to_nhwc_f32")
    return out


def lpips_diff(f0, f1, lin_w, out, gscale: float, need_grad: bool, accumulate: bool):
    from . import ops
    f0, f1 = _f(f0, "f0"), _f(f1, "f1")
    n, c = f0.shape[0], f0.shape[-1]
    hw = f0.numel() // (n * c)
    ws = ops.workspace(2048 * 8, f0.device, slot="lpips32")
    df1 = torch.empty_like(f1) if need_grad else None
    check(_lib.lib().dmvae_lpips_diff_f32(f0.data_ptr(), f1.data_ptr(), lin_w.data_ptr(), _ptr(df1), out.data_ptr(), ws.data_ptr(), ws.numel(), n, hw, c,
                                          float(gscale), int(accumulate), _stream()), "lpips_diff_f32")
    return df1


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    x = _f(x, "x")
    c = x.shape[-1]
    y = torch.empty_like(x)
    check(_lib.lib().dmvae_layernorm_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), x.numel() // c, c, float(eps), _stream()),
          "layernorm_f32")
    return y
