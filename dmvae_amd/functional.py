"""torch.autograd.Functions that orchestrate the HIP kernels for one reference module each.

Granularity follows models/flux_ae.py: one Function per ResnetBlock (:69-82), AttnBlock (:37-52),
Upsample (:103-107), plain conv (:237,274), norm_out+swish+conv_out (:266-268) and the bottleneck
MLP (models/vae.py:56-65).  Forward and backward are hand-scheduled sequences of C-ABI calls
(dmvae_amd.ops); nothing here computes on the CPU.  Internal activations are NHWC bf16.

What each backward saves: the block input, the GroupNorm statistics, and the bf16 post-norm
activations `a = swish(GN(x))` that the forward materialised anyway (they are the wgrad operand).
"""
from __future__ import annotations

from typing import Optional
import os

import weakref

import torch

from . import ops, parity
from .ops import bf16, f32

_EPOCH = [0]   # bumped by optimisers that update weights through raw pointers


def bump_weight_epoch() -> None:
    _EPOCH[0] += 1


def _epoch_of(w) -> int:
    """Update counter of the flat buffer that owns `w` (optim.FlatParams tags its parameters), else the global one: an optimiser step on one
    model must not invalidate the cached bf16 operands of another (the frozen DiT teacher next to the training student)."""
    return getattr(w, "_dmvae_epoch", _EPOCH)[0]


def packed(w: torch.Tensor, for_dgrad: bool = False, rows_pad: int = 0, cols_pad: int = 0, frozen: bool = False,
           transposed: bool = False, subpixel: bool = False, kmajor: bool = False) -> torch.Tensor:
    """bf16 kernel operand of an f32 conv/linear weight, cached ON the parameter object until the weight changes
    (in-place updates bump ``_version``; raw-pointer optimisers call bump_weight_epoch, which `frozen` weights -- not owned
    by any optimiser, e.g. the LPIPS trunk -- ignore).  subpixel: the operand of the 4x4 stride-2 conv D that Upsample's
    3x3 conv turns into (ops.subpixel_weight; for_dgrad=True then packs D's transpose, i.e. the FORWARD operand of the upsample conv)."""
    cache = getattr(w, "_dmvae_packed", None)
    if cache is None:
        cache = {}
        try:
            w._dmvae_packed = cache
        except AttributeError:      # non-leaf views etc.: no caching
            pass
    key = (for_dgrad, rows_pad, cols_pad, transposed, parity.on(), subpixel, kmajor)
    ver = (w.data_ptr(), _ver(w), -1 if frozen else _epoch_of(w))      # _ver: inference tensors (a frozen module built under inference_mode) keep no version counter
    hit = cache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    src = w.detach().contiguous()
    if src.dtype != f32:            # a bf16 shadow module's weight (train.frozen_bf16_shadow): the pack kernel reads f32
        src = src.float()
    if subpixel:
        src = ops.subpixel_weight(src)
    # 3x3 / 4x4 convs (Upsample's sub-pixel operand included): the pack also writes the K-tile-major copy the large-tile conv kernel reads (whole 128-B lines per weight tile; ops._weight_operand)
    # kmajor: the same for a Linear weight -- the operand `ops.linear_bf16` reads whole 128-B lines from (csrc/gemm_pp.hip, w_layout = 1)
    p = ops.pack_conv_weight(src, for_dgrad, rows_pad, cols_pad,
                             kmajor=(src.dim() == 4 and src.shape[2] in (3, 4) and not transposed) or (kmajor and src.dim() == 2 and not transposed))
    if transposed:       # [rows][taps*cols] -> [taps*cols][rows]: the B operand of the im2col convs' input-gradient GEMM
        p = p.view(p.shape[0], -1).t().contiguous()
    cache[key] = (ver, p)
    reg = getattr(w, "_dmvae_pack_reg", None)
    if reg is not None and _PACK_BATCHED and not frozen and not transposed and not parity.on() and w.dtype == f32 and w.is_contiguous():
        # a weight of an optimiser-owned flat buffer: from now on its operands are rewritten in place by the ONE launch that follows the optimiser step (repack_all)
        reg["entries"][(id(w), key)] = (w, key, p, bool(for_dgrad), bool(subpixel), w.data_ptr())
        reg["dirty"] = True
    return p


_PACK_BATCHED = True      # tests/test_gpu_pack_batched.py switches it off to compare with the per-weight lazy route


def new_pack_registry() -> dict:
    """Per flat parameter buffer (optim.FlatParams): the packed operands `packed` has produced for its weights, so that they can all be refreshed by one launch."""
    return {"entries": {}, "dirty": True, "table": None, "n": 0, "total": 0}


def repack_all(reg: dict, epoch: list) -> None:
    """Rewrite, IN PLACE and in one launch (dmvae_pack_weights_batched), every packed weight operand registered for a flat parameter buffer, and mark the
    cache entries current.  Called by the fused optimiser step right after it has changed the weights (the per-weight route: one pack launch per weight and
    direction on its next use -- 84 launches per tokenizer step).  Entries whose parameter moved or whose cached operand was replaced are dropped (the lazy
    route repacks and re-registers them)."""
    import ctypes
    from . import _lib
    ents = reg["entries"]
    if not ents or not _PACK_BATCHED or parity.on():
        return
    live = {}
    for k, (w, key, p, for_dgrad, subpixel, ptr) in ents.items():
        hit = getattr(w, "_dmvae_packed", {}).get(key)
        if hit is None or hit[1] is not p or w.data_ptr() != ptr:
            reg["dirty"] = True
            continue
        live[k] = ents[k]
    if len(live) != len(ents):
        reg["entries"] = ents = live
    if not ents:
        return
    if reg["dirty"]:
        arr = (_lib.PackEntry * len(ents))()
        start, taps = 0, 1
        for i, (w, key, p, for_dgrad, subpixel, ptr) in enumerate(ents.values()):
            if w.dim() == 2:
                cout, cin, T = w.shape[0], w.shape[1], 1
            else:
                cout, cin, T = w.shape[0], w.shape[1], w.shape[2] * w.shape[2]
            if subpixel:         # the tensor being packed is WD [cin_w][cout_w][4][4]
                cout, cin, T = cin, cout, 16
            rows_pad, cols_pad = p.shape[0], p.shape[2]
            assert p.shape[1] == T and p.is_contiguous()
            p2 = getattr(p, "_dmvae_kmajor", None)
            e = arr[i]
            e.src, e.dst, e.dst2 = ptr, p.data_ptr(), (p2.data_ptr() if p2 is not None else None)
            e.cout, e.cin, e.T, e.rows_pad, e.cols_pad, e.mode, e.subpixel = cout, cin, T, rows_pad, cols_pad, int(for_dgrad), int(subpixel)
            e.start, e.count = start, ((rows_pad + 31) // 32) * ((cols_pad + 31) // 32)      # 32 x 32 tiles of (row, column) pairs, all taps
            start += e.count
            taps = max(taps, 9 if subpixel else T)
        assert ctypes.sizeof(_lib.PackEntry) == _lib.lib().dmvae_pack_entry_bytes()
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        dev = next(iter(ents.values()))[0].device
        reg["table"], reg["n"], reg["total"], reg["taps"], reg["dirty"] = host.to(dev), len(ents), start, taps, False
    ops.check(_lib.lib().dmvae_pack_weights_batched(reg["table"].data_ptr(), reg["n"], reg["total"], reg["taps"], ops._stream()), "pack_weights_batched")
    ep = epoch[0]
    for (w, key, p, for_dgrad, subpixel, ptr) in ents.values():
        w._dmvae_packed[key] = ((ptr, _ver(w), ep), p)


DIRECT_GRAD_WRITES = None     # a dict while optim.FlatParams.check_direct_writes counts destination hand-outs (debug mode, DMVAE_CHECK_DIRECT_GRADS=1)


def _dst(param):
    """Destination view for a parameter's gradient when its owner enabled direct flat-buffer gradients (optim.FlatParams)."""
    if param is None:
        return None
    v = getattr(param, "_dmvae_grad_view", None)
    if v is not None and DIRECT_GRAD_WRITES is not None:      # debug: which direct-gradient parameters a backward wrote (optim.FlatParams.check_direct_writes)
        DIRECT_GRAD_WRITES[id(param)] = DIRECT_GRAD_WRITES.get(id(param), 0) + 1
    # a FRESH view object per use: AccumulateGrad adopts a gradient without copying only if nobody else holds its TensorImpl
    return None if v is None else v.view(v.shape)


_GN_CHECK = os.environ.get("DMVAE_GN_STATS_CHECK", "0") != "0"
_GN_GROUPS, _GN_EPS = 32, 1e-6     # every GroupNorm of flux_ae.py (Normalize(): num_groups=32, eps=1e-6, flux_ae.py:28) -- what the conv epilogues' statistics are for


def _tag_stats(y: torch.Tensor, st: torch.Tensor, groups: int = _GN_GROUPS, eps: float = _GN_EPS) -> torch.Tensor:
    """Leave the GroupNorm(32, eps 1e-6) statistics a conv epilogue computed on its result tensor, for the norm that consumes it next (the modules of
    flux_ae.Decoder hand their outputs straight to each other).  Keyed on the tensor's version and storage: a changed or re-created tensor falls back to the
    statistics pass."""
    try:
        y._dmvae_gn = (st, _ver(y), y.data_ptr(), groups, eps)
    except AttributeError:
        pass
    return y


def _ver(t: torch.Tensor) -> int:
    return -1 if t.is_inference() else t._version        # inference tensors keep no version counter


def _stats_of(x: torch.Tensor, groups: int = _GN_GROUPS, eps: float = _GN_EPS) -> torch.Tensor:
    """Statistics of x for GroupNorm(groups, eps): the producing conv's by-product when x still carries it for the SAME (groups, eps), version and storage
    (raw-pointer in-place kernels do not bump `_version`: one applied to a tagged tensor must `del x._dmvae_gn`), else the statistics pass.
    DMVAE_GN_STATS_CHECK=1 recomputes them and asserts the by-product equals the pass (debugging aid)."""
    tag = getattr(x, "_dmvae_gn", None)
    if tag is not None and tag[1] == _ver(x) and tag[2] == x.data_ptr() and tag[0].shape[0] == x.shape[0] and tag[3] == groups and tag[4] == eps:
        if _GN_CHECK:
            ref = ops.groupnorm_stats(x, groups, eps)
            assert torch.allclose(tag[0], ref, rtol=1e-4, atol=1e-6), "stale GroupNorm statistics tag"
        return tag[0]
    return ops.groupnorm_stats(x, groups, eps)


def _gn_swish(x, gw, gb, swish=True):
    st = _stats_of(x)
    return st, ops.groupnorm_apply(x, st, gw, gb, swish)


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _lib_mod():
    from . import _lib
    return _lib


SHORTCUT_IN_NORM = True      # tests/test_gpu_norm_short.py switches it off to compare with the stored-operand route
SHORTCUT_IN_NORM_512 = True  # ... the 512 -> 256 block's alone (its kernels differ: csrc/norm_short.hip, namespace coop)


def _shortcut_in_norm(x: torch.Tensor, sw) -> bool:
    if sw is None or not SHORTCUT_IN_NORM or parity.on() or x.dtype != bf16 or sw.dim() != 4 or sw.shape[2] != 1:
        return False
    n, c = x.shape[0], x.shape[-1]
    if c == 512 and not SHORTCUT_IN_NORM_512:
        return False
    return sw.shape[1] == c and ops.groupnorm_short_supported(n, x.numel() // (n * c), c, sw.shape[0])


class ResnetBlockFn(torch.autograd.Function):
    """x + conv2(swish(GN(conv1(swish(GN(x)))))) with optional 1x1 shortcut (flux_ae.py:69-82)."""

    @staticmethod
    def forward(ctx, x, n1w, n1b, c1w, c1b, n2w, n2b, c2w, c2b, sw, sb):
        ctx.short = _shortcut_in_norm(x, sw)
        if ctx.short:        # norm1 and the 1x1 shortcut from one read of x (csrc/norm_short.hip)
            st1 = _stats_of(x)
            a1, xs = ops.groupnorm_apply_short(x, st1, n1w, n1b, packed(sw), sb)
        else:
            st1, a1 = _gn_swish(x, n1w, n1b)
            xs = x if sw is None else ops.conv2d_nhwc(x, packed(sw), sb, ks=1)
        h1, st2 = ops.conv2d_nhwc_gnstats(a1, packed(c1w), c1b, ks=3)             # norm2's statistics from conv1's epilogue
        a2 = ops.groupnorm_apply(h1, st2, n2w, n2b, True)
        y, sty = ops.conv2d_nhwc_gnstats(a2, packed(c2w), c2b, residual=xs, ks=3)   # ... and the next module's norm from conv2's
        ctx.save_for_backward(x, st1, a1, h1, st2, a2, n1w, n1b, c1w, n2w, n2b, c2w, sw)
        ctx.bias_params = (c1b, c2b, sb)     # only for their gradient destinations
        ctx.want_colsum = bool(getattr(x, "_dmvae_want_colsum", False)) and not parity.on()
        return _tag_stats(y, sty)

    @staticmethod
    def backward(ctx, dy):
        x, st1, a1, h1, st2, a2, n1w, n1b, c1w, n2w, n2b, c2w, sw = ctx.saved_tensors
        dy = _c(dy)
        c1b, c2b, sb = ctx.bias_params
        dc2w, dc2b = ops.conv2d_nhwc_wgrad(dy, a2, 3, dw_out=_dst(c2w), db_out=_dst(c2b))
        da2 = ops.conv2d_nhwc(dy, packed(c2w, True), ks=3)
        dh1, dn2w, dn2b = ops.groupnorm_bwd(da2, h1, st2, n2w, n2b, True, dg_out=_dst(n2w), db_out=_dst(n2b))
        dc1w, dc1b = ops.conv2d_nhwc_wgrad(dh1, a1, 3, dw_out=_dst(c1w), db_out=_dst(c1b))
        da1 = ops.conv2d_nhwc(dh1, packed(c1w, True), ks=3)
        if sw is None:
            dxs, dsw, dsb = dy, None, None
        else:
            dsw, dsb = ops.conv2d_nhwc_wgrad(dy, x, 1, dw_out=_dst(sw), db_out=_dst(sb))
            if ctx.short:    # the shortcut's input gradient is formed inside norm1's backward apply pass: the stored gradient (twice x's channels) never exists
                dx, dn1w, dn1b = ops.groupnorm_bwd_short(da1, x, dy, packed(sw, True), st1, n1w, n1b, True, dg_out=_dst(n1w), db_out=_dst(n1b),
                                                         want_colsum=ctx.want_colsum)
                return dx, dn1w, dn1b, dc1w, dc1b, dn2w, dn2b, dc2w, dc2b, dsw, dsb
            dxs = ops.conv2d_nhwc(dy, packed(sw, True), ks=1)
        dx, dn1w, dn1b = ops.groupnorm_bwd(da1, x, st1, n1w, n1b, True, dres=dxs, dg_out=_dst(n1w), db_out=_dst(n1b), want_colsum=ctx.want_colsum)
        return dx, dn1w, dn1b, dc1w, dc1b, dn2w, dn2b, dc2w, dc2b, dsw, dsb


class AttnBlockFn(torch.autograd.Function):
    """x + proj_out(SDPA(q,k,v)) over (h w) tokens, single head, d = C (flux_ae.py:37-52)."""

    @staticmethod
    def forward(ctx, x, nw, nb, qw, qb, kw, kb, vw, vb, pw, pb):
        n, h, w, c = x.shape
        s = h * w
        st, hn = _gn_swish(x, nw, nb, swish=False)
        q = ops.conv2d_nhwc(hn, packed(qw), qb, ks=1).view(n, s, c)
        k = ops.conv2d_nhwc(hn, packed(kw), kb, ks=1).view(n, s, c)
        v = ops.conv2d_nhwc(hn, packed(vw), vb, ks=1).view(n, s, c)
        scale = float(c) ** -0.5
        p = ops.softmax_rows(ops.gemm_nt(q, k, out_f32=True), scale)          # [n, s, s] bf16
        o = ops.gemm_nt(p, ops.transpose_last2(v))                            # [n, s, c]
        y, sty = ops.conv2d_nhwc_gnstats(o.view(n, h, w, c), packed(pw), pb, residual=x, ks=1)
        ctx.save_for_backward(x, st, hn, q, k, v, p, o, nw, nb, qw, kw, vw, pw)
        ctx.bias_params = (qb, kb, vb, pb)
        ctx.scale = scale
        return _tag_stats(y, sty)

    @staticmethod
    def backward(ctx, dy):
        x, st, hn, q, k, v, p, o, nw, nb, qw, kw, vw, pw = ctx.saved_tensors
        n, h, w, c = x.shape
        s = h * w
        dy = _c(dy)
        qb, kb, vb, pb = ctx.bias_params
        dpw, dpb = ops.conv2d_nhwc_wgrad(dy, o.view(n, h, w, c), 1, dw_out=_dst(pw), db_out=_dst(pb))
        do = ops.conv2d_nhwc(dy, packed(pw, True), ks=1).view(n, s, c)
        dp = ops.gemm_nt(do, v, out_f32=True)                                  # dP[q][key] = do[q].v[key]
        ds = ops.softmax_rows_bwd(dp, p, ctx.scale)                            # bf16, includes the scale
        dv = ops.gemm_tn(p, do)                                                # [n, key, c]
        dq = ops.gemm_nt(ds, ops.transpose_last2(k))                           # [n, q, c]
        dk = ops.gemm_tn(ds, q)                                                # [n, key, c]
        dq4, dk4, dv4 = dq.view(n, h, w, c), dk.view(n, h, w, c), dv.view(n, h, w, c)
        dqw, dqb = ops.conv2d_nhwc_wgrad(dq4, hn, 1, dw_out=_dst(qw), db_out=_dst(qb))
        dkw, dkb = ops.conv2d_nhwc_wgrad(dk4, hn, 1, dw_out=_dst(kw), db_out=_dst(kb))
        dvw, dvb = ops.conv2d_nhwc_wgrad(dv4, hn, 1, dw_out=_dst(vw), db_out=_dst(vb))
        dh = ops.conv2d_nhwc(dq4, packed(qw, True), ks=1)
        dh = ops.conv2d_nhwc(dk4, packed(kw, True), residual=dh, ks=1)
        dh = ops.conv2d_nhwc(dv4, packed(vw, True), residual=dh, ks=1)
        dx, dnw, dnb = ops.groupnorm_bwd(dh, x, st, nw, nb, False, dres=dy, dg_out=_dst(nw), db_out=_dst(nb))
        return dx, dnw, dnb, dqw, dqb, dkw, dkb, dvw, dvb, dpw, dpb


UPS_SUBPIXEL = None      # None: the default below; True / False: tests force a route (tests/test_gpu_subpixel.py, test_gpu_parity_fp32.py)


def _subpixel_upsample() -> bool:
    """Upsample's conv in its sub-pixel form (DESIGN_HISTORY.md 8.1) -- on, except in the fp32 parity mode, which keeps the reference's own evaluation order: nine taps
    per output pixel gathered from the half-resolution image (the layer agrees to 1e-4 either way -- tests/test_gpu_parity_fp32.py -- but the four-step Adam
    trajectory test amplifies any reordering of f32 sums past that bar)."""
    return (not parity.on()) if UPS_SUBPIXEL is None else bool(UPS_SUBPIXEL)


class ConvFn(torch.autograd.Function):
    """Plain conv (3x3 pad 1 or 1x1), optionally on the nearest-x2 upsampled input (flux_ae.py:103-107).

    The upsampled case runs in its sub-pixel form (include/dmvae_hip.h, dmvae_subpixel_weight): with WD the 4x4 weight obtained by adding the taps of W
    that read the same source pixel, forward = the transposed 4x4 stride-2 conv of x with WD, input gradient = the 4x4 stride-2 conv of dy with WD,
    weight gradient = that conv's weight gradient (operands' roles exchanged) folded back to 3x3 -- 16 taps per source pixel instead of 9 per output
    pixel, 4/9 of the multiply-adds in all three; no upsampled tensor, no full-resolution input gradient to pool."""

    @staticmethod
    def forward(ctx, x, w, b, ks, upsample):
        sub = bool(upsample) and ks == 3 and _subpixel_upsample()
        sty = None
        if sub and w.shape[0] % 128 == 0:                                   # Upsample inside the decoder: a ResnetBlock's norm1 reads the result next
            y, sty = ops.conv2d_nhwc_gnstats(x, packed(w, True, subpixel=True), b, ks=4, stride=2, transposed=True)
        elif sub:                                                           # conv_in.0 (z_channels wide): feeds a conv, no statistics wanted
            y = ops.conv2d_nhwc(x, packed(w, True, subpixel=True), b, ks=4, stride=2, transposed=True)
        elif not upsample and w.shape[0] % 128 == 0:
            y, sty = ops.conv2d_nhwc_gnstats(x, packed(w), b, ks=ks)        # conv_in.1: a ResnetBlock's norm1 reads the result next
        else:
            y = ops.conv2d_nhwc(x, packed(w), b, ks=ks, upsample=upsample)
        ctx.save_for_backward(x, w)
        ctx.ks, ctx.upsample, ctx.bias_param, ctx.sub = ks, upsample, b, sub
        if sub and b is not None:
            try:
                y._dmvae_want_colsum = True      # the module that consumes y can hand back the column sums of dL/dy (this layer's bias gradient) for free
            except AttributeError:
                pass
        return y if sty is None else _tag_stats(y, sty)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        if ctx.sub:
            dwd, _ = ops.conv2d_nhwc_wgrad(x, dy, 4, need_bias=False, stride=2)          # D's weight gradient: D maps dy [N,2H,2W,Cout] to [N,H,W,Cin]
            dw = ops.subpixel_weight_fold(dwd, dw_out=_dst(w))
            db = None
            if ctx.bias_param is not None:
                tag, dst = getattr(dy, "_dmvae_colsum", None), _dst(ctx.bias_param)
                if tag is not None and tag[1] == dy._version and tag[2] == dy.data_ptr():
                    db = tag[0] if dst is None else dst.copy_(tag[0])       # summed by the GroupNorm backward that wrote dy (ResnetBlockFn.backward)
                else:
                    db = ops.colsum(dy, out=dst)
            dx = ops.conv2d_nhwc(dy, packed(w, False, subpixel=True), ks=4, stride=2) if ctx.needs_input_grad[0] else None
            return dx, dw, db, None, None
        cout, cin = w.shape[0], w.shape[1]
        m = dy.numel() // cout
        if THIN_CIN_BWD_AS_GEMM and ctx.ks == 3 and not ctx.upsample and not parity.on() and cin in (32, 64) and cout % 128 == 0 and m >= 16384 and m % 32 == 0 \
                and dy.dtype == bf16 and w.dtype == f32 and ops.linear_supported(m, 9 * cin, cout):
            # A 3x3 conv FROM few channels (the decoder's conv_in: z_channels = 32 -> 512 at 32 x 32, flux_ae.py:196): its weight gradient has 288 columns -- the
            # large weight-gradient kernel's 128-column tiles do not take it and the small-shape kernel runs it at 70 TFLOP/s (138 us); its input gradient has 32
            # output channels (101 us on the general kernel).  Both as GEMMs on the im2col form instead: col [M, 12 taps x cin] (three taps of zeros) against dy
            # through the 1x1 weight-gradient kernel, and dy [M, cout] x W [cout, 9 cin] on the Linear GEMM followed by the gather adjoint of im2col in f32.
            n, h, wd, _ = x.shape
            col = ops.im2col(x, 3, 1, 1, taps_pad=12)
            g2, db = ops.conv2d_nhwc_wgrad(dy.view(1, 1, m, cout), col.view(1, 1, m, 12 * cin), 1, db_out=_dst(ctx.bias_param), need_bias=ctx.bias_param is not None)
            dwv = g2.view(cout, 12, cin)[:, :9].permute(0, 2, 1).reshape(cout, cin, 3, 3)
            dw = _dst(w)
            dw = dw.copy_(dwv) if dw is not None else dwv.contiguous()
            dx = None
            if ctx.needs_input_grad[0]:
                wl = w.detach().permute(2, 3, 1, 0).reshape(9 * cin, cout).to(bf16)          # [(tap, ci)][co]
                dcol = ops.linear_bf16(dy.view(m, cout), wl, None, out_f32=True)              # [M, 9 cin] f32: no rounding before the taps are summed
                dx = ops.col2im(dcol.view(n, h, wd, 9 * cin), h, wd, 3, 1, 1)
            return dx, dw, db, None, None
        dw, db = ops.conv2d_nhwc_wgrad(dy, x, ctx.ks, upsample=ctx.upsample, dw_out=_dst(w), db_out=_dst(ctx.bias_param))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_nhwc(dy, packed(w, True), ks=ctx.ks)
            if ctx.upsample:
                dx = ops.sumpool2x2(dx)
        return dx, dw, db, None, None


class DownsampleFn(torch.autograd.Function):
    """Downsample conv (flux_ae.py:85-95): F.pad(x, (0,1,0,1)) then conv3x3 stride 2, as one strided gather in the conv
    kernel.  Its input gradient is the stride-1 dgrad conv over dy zero-inserted at odd positions (desc.upsample = 2)."""

    @staticmethod
    def forward(ctx, x, w, b):
        y = ops.conv2d_nhwc(x, packed(w), b, ks=3, stride=2)
        ctx.save_for_backward(x, w)
        ctx.bias_param = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw, db = ops.conv2d_nhwc_wgrad(dy, x, 3, stride=2, dw_out=_dst(w), db_out=_dst(ctx.bias_param))
        dx = ops.conv2d_nhwc(dy, packed(w, True), ks=3, upsample=2) if ctx.needs_input_grad[0] else None
        return dx, dw, db


class ConvInFn(torch.autograd.Function):
    """conv3x3 on an NCHW f32 image whose channel count is not a multiple of 32 (Encoder.conv_in, flux_ae.py:127): channels are
    zero-padded to 32 on the way into NHWC bf16."""

    @staticmethod
    def forward(ctx, x, w, b):
        cin = x.shape[1]
        xp = ops.nchw_to_nhwc_bf16(_c(x.float()), c_pad=32)
        y = ops.conv2d_nhwc(xp, packed(w, False, cols_pad=32), b, ks=3, flop_channels=(cin, w.shape[0]))
        ctx.save_for_backward(xp, w)
        ctx.bias_param, ctx.cin = b, cin
        return y

    @staticmethod
    def backward(ctx, dy):
        xp, w = ctx.saved_tensors
        dy = _c(dy)
        dwp, db = ops.conv2d_nhwc_wgrad(dy, xp, 3, db_out=_dst(ctx.bias_param))
        dw = _dst(w)
        if dw is not None:
            dw.copy_(dwp[:, :ctx.cin])
        else:
            dw = dwp[:, :ctx.cin].contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.nhwc_to_nchw_f32(ops.conv2d_nhwc(dy, packed(w, True, rows_pad=32), ks=3, flop_channels=(w.shape[0], ctx.cin)), ctx.cin)
        return dx, dw, db


class NormSwishConvFn(torch.autograd.Function):
    """conv3x3(swish(GroupNorm(x))) with an NHWC bf16 result (Encoder tail, flux_ae.py:178-180)."""

    @staticmethod
    def forward(ctx, x, nw, nb, cw, cb):
        st, a = _gn_swish(x, nw, nb)
        y = ops.conv2d_nhwc(a, packed(cw), cb, ks=3)
        ctx.save_for_backward(x, st, a, nw, nb, cw)
        ctx.bias_param = cb
        return y

    @staticmethod
    def backward(ctx, dy):
        x, st, a, nw, nb, cw = ctx.saved_tensors
        dy = _c(dy)
        dcw, dcb = ops.conv2d_nhwc_wgrad(dy, a, 3, dw_out=_dst(cw), db_out=_dst(ctx.bias_param))
        da = ops.conv2d_nhwc(dy, packed(cw, True), ks=3)
        dx, dnw, dnb = ops.groupnorm_bwd(da, x, st, nw, nb, True, dg_out=_dst(nw), db_out=_dst(nb))
        return dx, dnw, dnb, dcw, dcb


class ImageToNhwcFn(torch.autograd.Function):
    """NCHW f32 image -> NHWC bf16 with the channel count zero-padded to c_pad (boundary of the discriminator / VGG trunk)."""

    @staticmethod
    def forward(ctx, x, c_pad):
        ctx.c = x.shape[1]
        return ops.nchw_to_nhwc_bf16(_c(x.float()), c_pad=c_pad)

    @staticmethod
    def backward(ctx, dy):
        return ops.nhwc_to_nchw_f32(_c(dy), ctx.c), None


_TAIL_WEIGHT_ONLY = [False]


class tail_weight_only:
    """Context: the decoder tail's backward node (NormConvOutFn) returns only conv_out's weight gradient (see its backward).  DMVAE_TAIL_WEIGHT_ONLY=0 disables it."""
    ON = os.environ.get("DMVAE_TAIL_WEIGHT_ONLY", "1") != "0"

    def __enter__(self):
        self.prev = _TAIL_WEIGHT_ONLY[0]
        _TAIL_WEIGHT_ONLY[0] = self.ON
        return self

    def __exit__(self, *exc):
        _TAIL_WEIGHT_ONLY[0] = self.prev
        return False


K4_L1_LEAN = os.environ.get("DMVAE_K4_L1_LEAN", "1") != "0"      # first PatchGAN layer's weight gradient without the sliced copy of x / the padded copy of dY (0: with them, for A/B)
K4_COUT1 = os.environ.get("DMVAE_K4_COUT1", "1") != "0"      # the one-output-channel 4x4 conv (PatchGAN logits) on csrc/conv_c1.hip's vector-unit kernels
K4_WGRAD_THIN_CIN = True      # the <= 8-input-channel 4x4 conv's weight gradient on the im2col form of an 8-channel copy (ConvK4Fn.backward)
K4_WGRAD_AS_GEMM = (1,)      # strides of the 4x4 convs whose weight gradient runs on the im2col form (ConvK4Fn.backward); () = never (tests compare the routes)


class ConvK4Fn(torch.autograd.Function):
    """nn.Conv2d(kernel 4, stride 1|2, padding 1) on NHWC bf16 (models/patchgan.py:125-147) as a strided 16-tap gather in the implicit-GEMM
    conv kernel -- no im2col tensor.  x: [N,H,W,Cp] with Cp >= w.shape[1] zero-padded channels (Cp % 32 == 0); act: ops.ACT_NONE |
    ops.ACT_LEAKY fused into the epilogue.  out_f32: logits layer -- the single output channel is computed with 4 padded rows and
    returned as [N,Ho,Wo,1] f32.  Backward: weight gradient from the same gather (desc.stride), input gradient as the transposed gather
    (desc.transposed; dy's channels zero-padded to a multiple of 32 as the reduction dimension)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, act, out_f32):
        cout, cin = w.shape[0], w.shape[1]
        cp = x.shape[-1]
        if (K4_COUT1 and cout == 1 and stride == 1 and out_f32 and act == ops.ACT_NONE and cp == cin and not parity.on() and x.dtype == bf16
                and ops.conv_k4c1_supported(x.shape[0], x.shape[1], x.shape[2], cp)):
            # the logits layer (models/patchgan.py:146): one output channel -- vector-unit kernels bound by one pass over the map (csrc/conv_c1.hip), not a
            # matrix product with one useful row
            y = ops.conv_k4c1_fwd(x, w, b)
            ctx.save_for_backward(x, w, None)
            ctx.cfg = (stride, act, b)
            ctx.c1 = True
            return y
        ctx.c1 = False
        rows = cout if cout % 4 == 0 else (cout + 3) // 4 * 4
        bp = b
        if b is not None and rows != cout:
            bp = torch.zeros(rows, dtype=f32, device=x.device)
            bp[:cout] = b
        y = ops.conv2d_nhwc(x, packed(w, False, rows_pad=rows, cols_pad=cp), bp, ks=4, stride=stride, act=act, out_f32=out_f32, flop_channels=(cin, cout))
        if rows != cout:
            y = y[..., :cout].contiguous()
        ctx.save_for_backward(x, w, y if act == ops.ACT_LEAKY else None)
        ctx.cfg = (stride, act, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, act, b = ctx.cfg
        cout, cin = w.shape[0], w.shape[1]
        n, h, wd, cp = x.shape
        dy = _c(dy)
        if ctx.c1:
            dy = dy if dy.dtype == f32 else dy.float()
            dw = db = None
            if ctx.needs_input_grad[1] or (b is not None and ctx.needs_input_grad[2]):       # a frozen discriminator (the generator's term) wants neither
                dw, db = _dst(w), (_dst(b) if b is not None else None)
                dw, db = ops.conv_k4c1_wgrad(x, dy, dw, db, need_bias=b is not None)
            dx = ops.conv_k4c1_dgrad(dy, w, h) if ctx.needs_input_grad[0] else None
            return dx, dw, db, None, None, None
        if act == ops.ACT_LEAKY:
            dy = ops.leaky_relu_bwd(dy, y)
        cpad = cout if cout % 32 == 0 else (cout + 31) // 32 * 32
        adt = parity.act_dtype()                    # bf16; f32 in the fp32 parity mode (the gradient operand is split exactly there, not rounded)
        if cpad != cout or dy.dtype != adt:
            dyp = torch.zeros(dy.shape[0], dy.shape[1], dy.shape[2], cpad, dtype=adt, device=dy.device)
            dyp[..., :cout] = dy
            dy = dyp
        ho, wo = dy.shape[1], dy.shape[2]
        m = n * ho * wo
        # The generator's adversarial term runs through a FROZEN discriminator (losses.generator_gan_term: requires_grad False on its parameters), twice per step
        # (the adaptive weight's autograd.grad and the final backward): no weight / bias gradient is wanted there -- computing them anyway was 2 x 1.1 ms per step
        want_w = ctx.needs_input_grad[1] or (b is not None and ctx.needs_input_grad[2])
        if not want_w:
            dwv = dbp = None
        elif K4_WGRAD_AS_GEMM and not parity.on() and stride in K4_WGRAD_AS_GEMM and cpad == cout and cout % 128 == 0 and (16 * cp) % 128 == 0 and m % 32 == 0 and m >= 4096:
            # The large weight-gradient kernel takes 4x4 convs with stride 2 and 128-channel groups only; the 256 -> 512 stride-1 layer of the PatchGAN
            # (models/patchgan.py:125-147, 31 x 31 outputs: no 32-pixel K tiles either) ran on the small-shape kernel at ~130 TFLOP/s -- a millisecond per
            # discriminator pass, four passes per step.  On the im2col form it is the 1x1 weight gradient of a [rows, 16 cp] operand: one gather pass
            # (252 MB at B = 32) + the large kernel.
            col = ops.im2col(x, 4, stride, 1)
            g2, dbp = ops.conv2d_nhwc_wgrad(dy.view(1, 1, m, cout), col.view(1, 1, m, 16 * cp), 1, need_bias=b is not None)
            dwv = g2.view(cout, 16, cp)[:, :, :cin].permute(0, 2, 1).reshape(cout, cin, 4, 4)
        elif K4_WGRAD_THIN_CIN and not parity.on() and cin <= 8 and cp >= 8 and cpad <= 128 and dy.dtype == bf16 and m % 32 == 0 and m >= 16384:
            # The PatchGAN's first layer (3 -> 64, stride 2, models/patchgan.py:125): its input travels zero-padded to 32 channels and the small-shape kernel spent
            # 0.95 ms per discriminator pass (B = 64) on a weight gradient with 3 real input channels.  On the im2col form of an EIGHT-channel copy -- 16 taps x 8 =
            # one 128-channel group -- against the output gradient zero-padded to one group it is the large kernel's 1x1 case, bound by reading ~0.5 GB.
            # (the 8-channel im2col straight from the 32-channel tensor, and dY as it is when it has 64 channels -- the kernel masks the rows past Cout: the sliced
            # copy of x, the zero-padded copy of dY and half of dY's read were 0.33 ms per step)
            col = ops.im2col(x, 4, stride, 1, c_take=8) if K4_L1_LEAN else ops.im2col(x[..., :8].contiguous(), 4, stride, 1)
            cg = cpad if cpad in ((64, 128) if K4_L1_LEAN else (128,)) else 128
            dyg = dy if cpad == cg else torch.nn.functional.pad(dy, (0, cg - cpad))
            g2, dbp = ops.conv2d_nhwc_wgrad(dyg.view(1, 1, m, cg), col.view(1, 1, m, 128), 1, need_bias=b is not None)
            dwv = g2.view(cg, 16, 8)[:cout, :, :cin].permute(0, 2, 1).reshape(cout, cin, 4, 4)
        else:
            dwp, dbp = ops.conv2d_nhwc_wgrad(dy, x, 4, stride=stride, need_bias=b is not None)       # [cpad, cp, 4, 4]
            dwv = dwp[:cout, :cin]
        dw, db = (_dst(w) if want_w else None), None
        if not want_w:
            pass
        elif dw is not None:
            dw.copy_(dwv)
        else:
            dw = dwv.contiguous()
        if b is not None and want_w:
            db = _dst(b)
            if db is not None:
                db.copy_(dbp[:cout])
            else:
                db = dbp[:cout].contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv2d_nhwc(dy, packed(w, True, rows_pad=cp, cols_pad=cpad), ks=4, stride=stride, transposed=True)
            if dx.shape[1] != h or dx.shape[2] != wd:      # (h - 2) % stride != 0: the conv never read the last row / column
                full = torch.zeros_like(x)
                full[:, :dx.shape[1], :dx.shape[2]] = dx[:, :h, :wd]
                dx = full
        return dx, dw, db, None, None, None


class BatchNormActFn(torch.autograd.Function):
    """act(BatchNorm(x)) on NHWC bf16 with caller-provided per-channel statistics (models/patchgan.py:134-145): the GroupNorm
    kernels with one "image" of N*H*W pixels and one channel per group.  `stats` [1,C,2] = (mean, rstd): batch statistics
    (combined over ranks for SyncBatchNorm) when `batch_stats`, the running estimates otherwise.  In the first case the backward
    all-reduces the two per-channel sums over `group` like torch.nn.SyncBatchNorm; in the second the statistics are constants."""

    @staticmethod
    def forward(ctx, x, gamma, beta, stats, act, batch_stats, count, group):
        c = x.shape[-1]
        y = ops.groupnorm_apply(x.view(1, -1, c), stats, gamma, beta, act, groups=c).view(x.shape)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (act, batch_stats, count, group)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, stats = ctx.saved_tensors
        act, batch_stats, count, group = ctx.cfg
        c = x.shape[-1]
        dy3, x3 = _c(dy).view(1, -1, c), x.view(1, -1, c)
        need_p = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dg = db = None
        if parity.on():
            # fp32 parity mode: the f32 GroupNorm backward (one "image", one channel per group) -- batch statistics of ONE rank only (the cross-rank sums and the
            # constant-statistics form of an eval-mode discriminator are not needed by the 1e-4 tests and are refused rather than approximated)
            if group is not None:
                raise NotImplementedError("BatchNormActFn backward in the fp32 parity mode: a single rank only (the cross-rank sums of SyncBatchNorm are not built in f32)")
            if batch_stats:
                dx, dg, db = parity.groupnorm_bwd(dy3.float(), x3, stats, gamma, beta, act, groups=c, need_param_grads=need_p, dg_out=_dst(gamma) if need_p else None,
                                                  db_out=_dst(beta) if need_p else None, inv_count=1.0 / count)
                return (dx.view(x.shape) if ctx.needs_input_grad[0] else None), dg, db, None, None, None, None, None
            # running statistics (an eval-mode discriminator: the generator's adversarial term, train_tokenizer.py:190-193): the statistics are constants, so
            # dx = g * gamma * rstd with g = dy * act'(u); d gamma = sum g * xh, d beta = sum g -- composed from the f32 kernels (no mean-subtraction terms)
            from .models.lightningdit_parity import colsum_groups
            rows = x3.shape[1]
            g = _c(dy3.float())
            if act == 2:
                g = parity.eltwise(3, g, parity.groupnorm_apply(x3, stats, gamma, beta, act, groups=c), param=0.2)      # LeakyReLU backward from the activation's output
            elif act != 0:
                raise NotImplementedError("BatchNormActFn backward in the fp32 parity mode: LeakyReLU or no activation")
            L, st = _lib_mod().lib(), torch.cuda.current_stream().cuda_stream
            scale = (gamma.detach().float() * stats[0, :, 1]).contiguous().view(1, c)
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(g)
                _lib_mod().check(L.dmvae_bcast_rows_f32(1, g.data_ptr(), None, scale.data_ptr(), dx.data_ptr(), rows, c, rows, c, st), "bcast_rows_f32")
                dx = dx.view(x.shape)
            dg = db = None
            if need_p:
                xh = parity.groupnorm_apply(x3, stats, torch.ones_like(gamma), torch.zeros_like(beta), 0, groups=c)
                prod = torch.empty_like(g)
                _lib_mod().check(L.dmvae_bcast_rows_f32(2, g.data_ptr(), xh.data_ptr(), None, prod.data_ptr(), rows, c, rows, 0, st), "bcast_rows_f32")
                dg, db = colsum_groups(prod.view(rows, c), 1).view(c), colsum_groups(g.view(rows, c), 1).view(c)
            return dx, dg, db, None, None, None, None, None
        if batch_stats or need_p:
            sums, dg, db = ops.groupnorm_bwd_reduce(dy3, x3, stats, gamma, beta, act, groups=c, need_param_grads=need_p,
                                                    dg_out=_dst(gamma) if need_p else None, db_out=_dst(beta) if need_p else None)
        if batch_stats:
            if group is not None:
                import torch.distributed as tdist
                tdist.all_reduce(sums, group=group)
        else:
            sums = torch.zeros(1, c, 2, dtype=f32, device=x.device)
        dx = ops.groupnorm_bwd_apply(dy3, x3, stats, sums, gamma, beta, act, groups=c, inv_count=1.0 / count).view(x.shape) \
            if ctx.needs_input_grad[0] else None
        return dx, dg, db, None, None, None, None, None


class NormConvOutFn(torch.autograd.Function):
    """conv_out(swish(norm_out(h))) -> NCHW f32 image (flux_ae.py:266-268).  Cout (3) is padded to 4 for the
    forward store and to 32 for the bf16 gradient operand."""

    @staticmethod
    def forward(ctx, x, nw, nb, cw, cb):
        cout = cw.shape[0]
        if NORM_CONV_OUT_FUSED_FWD and not parity.on() and x.dtype == bf16 and ops.norm_conv_out_fwd_supported(x.shape[0], x.shape[1], x.shape[2], x.shape[3], cout):
            # GroupNorm + swish on the way into the conv's halo tile, the NCHW image straight out of its epilogue: one launch for five (csrc/conv_thin.hip, NORM)
            st = _stats_of(x)
            a, y = ops.norm_conv_out_fwd(x, st, nw, nb, packed(cw, False, rows_pad=4), cb, cout)
            ctx.save_for_backward(x, st, a, nw, nb, cw)
            ctx.bias_param = cb
            return y
        st, a = _gn_swish(x, nw, nb)
        cbp = torch.zeros(4, dtype=f32, device=x.device)
        cbp[:cout] = cb
        y4 = ops.conv2d_nhwc(a, packed(cw, False, rows_pad=4), cbp, ks=3, out_f32=True, flop_channels=(cw.shape[1], cout))
        ctx.save_for_backward(x, st, a, nw, nb, cw)
        ctx.bias_param = cb
        return ops.nhwc_to_nchw_f32(y4, cout)

    @staticmethod
    def backward(ctx, dy):
        x, st, a, nw, nb, cw = ctx.saved_tensors
        cout = cw.shape[0]
        dyf = _c(dy.float())
        if _TAIL_WEIGHT_ONLY[0] and not parity.on() and ops.conv_out_wgrad_supported(a.shape[0], a.shape[1], a.shape[2], a.shape[3], cout):
            # losses.generator_gan_backward asks this node for the LAST LAYER's weight gradient alone, twice per step (the adaptive weight's two norms,
            # train_tokenizer.py:190-203): autograd cannot tell a Function which of its gradients are wanted (needs_input_grad is fixed at forward), so the caller
            # says so -- the input gradient through conv_out + norm_out (two passes over the 537 MB map) and the bias sum were computed and dropped
            dwp = ops.conv_out_wgrad(dyf, a)
            dcw = _dst(cw)
            if dcw is not None:
                dcw.copy_(dwp[:cout])
                return None, None, None, dcw, None
            return None, None, None, dwp[:cout].contiguous(), None
        fused = NORM_CONV_OUT_FUSED_BWD and not parity.on() and ops.norm_conv_out_bwd_supported(x.shape[0], x.shape[1], x.shape[2], x.shape[3], cout)
        # the stored-operand route's gradient operand: the input-gradient conv's reduction dimension in 32-channel K steps
        dyp = None if fused else ops.nchw_to_nhwc_bf16(dyf, c_pad=32)
        if not parity.on() and ops.conv_out_wgrad_supported(a.shape[0], a.shape[1], a.shape[2], a.shape[3], cout):
            # three output channels: `a` is read once against three x-shifted planar copies of the gradient (csrc/wgrad_thin.hip: 570 -> ~150 us at B = 32);
            # the bias gradient is the plain sum of the gradient
            dwp = ops.conv_out_wgrad(dyf, a)
            dbp = dyf.sum(dim=(0, 2, 3))
        elif cout <= 8 and a.numel() // a.shape[-1] >= 16384 and not parity.on():
            # With 3 output channels a 128-row weight-gradient tile is 98 % padding (1.4 ms at B = 32).  Instead: im2col of the GRADIENT
            # (8 padded channels x 9 taps = 72 columns, col[q][t*8+co] = dy[q + off(t)][co]) and ONE 1x1 weight-gradient GEMM against `a`:
            #   G[(t, co)][ci] = sum_q dy[q + off(t)][co] * a[q][ci]  =  dW[co][ci][8 - t]   (the tap seen from the other side),
            # and the centre tap's column sums are the bias gradient.
            n_, h_, w_, c_ = a.shape
            m_ = n_ * h_ * w_
            col = ops.im2col(ops.nchw_to_nhwc_bf16(dyf, c_pad=8), 3, 1, 1)
            g_, gb = ops.conv2d_nhwc_wgrad(col.view(1, 1, m_, 72), a.view(1, 1, m_, c_), 1)
            dwp = g_.view(9, 8, c_).flip(0).permute(1, 2, 0).reshape(8, c_, 3, 3)
            dbp = gb.view(9, 8)[4]
        else:
            dwp, dbp = ops.conv2d_nhwc_wgrad(dyp if dyp is not None else ops.nchw_to_nhwc_bf16(dyf, c_pad=32), a, 3)
        if fused:
            # the 3 -> C input-gradient conv inside both GroupNorm backward passes (csrc/groupnorm.hip::convout_bwd_kernel): its result is 537 MB at B = 32,
            # written once and read twice on the stored-operand route
            dx, dnw, dnb = ops.norm_conv_out_bwd(dyf, cw, x, st, nw, nb, dg_out=_dst(nw), db_out=_dst(nb))
        else:
            da = ops.conv2d_nhwc(dyp, packed(cw, True, cols_pad=32), ks=3, flop_channels=(cw.shape[0], cw.shape[1]))
            dx, dnw, dnb = ops.groupnorm_bwd(da, x, st, nw, nb, True, dg_out=_dst(nw), db_out=_dst(nb))
        dcw, dcb = _dst(cw), _dst(ctx.bias_param)
        if dcw is not None:
            dcw.copy_(dwp[:cout])
            dcb.copy_(dbp[:cout])
            return dx, dnw, dnb, dcw, dcb
        return dx, dnw, dnb, dwp[:cout].contiguous(), dbp[:cout].contiguous()


def _nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [M, K] @ w [N, K]^T + bias on the large-tile Linear GEMM where it takes the shape (bf16 operands, outside the fp32 parity mode), else the batched NT kernel."""
    if x.dtype == bf16 and w.dtype == bf16 and not parity.on() and x.dim() == 2 and ops.linear_supported(x.shape[0], w.shape[0], x.shape[1]):
        return ops.linear_bf16(x, w, bias)
    return ops.gemm_nt(x, w, bias)


class MLPFn(torch.autograd.Function):
    """Linear -> SiLU -> Linear on tokens (models/vae.py:56-65); x [M, Cin] bf16."""

    @staticmethod
    def forward(ctx, x, w0, b0, w2, b2):
        h = _nt(x, packed(w0).view(w0.shape[0], -1), b0)          # [out, in] (parity mode: [out, 6*in], the split weight-side parts)
        a = ops.silu(h)
        y = _nt(a, packed(w2).view(w2.shape[0], -1), b2)
        ctx.save_for_backward(x, h, a, w0, w2)
        ctx.bias_params = (b0, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, h, a, w0, w2 = ctx.saved_tensors
        dy = _c(dy)
        m = x.shape[0]
        b0, b2 = ctx.bias_params
        v4 = lambda t: None if t is None else t.view(t.shape[0], t.shape[1], 1, 1)
        dw2, db2 = ops.conv2d_nhwc_wgrad(dy.view(1, 1, m, -1), a.view(1, 1, m, -1), 1, dw_out=v4(_dst(w2)), db_out=_dst(b2))
        da = _nt(dy, packed(w2, True).view(w2.shape[1], -1))
        dh = ops.silu_bwd(h, da)
        dw0, db0 = ops.conv2d_nhwc_wgrad(dh.view(1, 1, m, -1), x.view(1, 1, m, -1), 1, dw_out=v4(_dst(w0)), db_out=_dst(b0))
        dx = _nt(dh, packed(w0, True).view(w0.shape[1], -1)) if ctx.needs_input_grad[0] else None
        return dx, dw0.view(w0.shape), db0, dw2.view(w2.shape), db2


class ReparamKLFn(torch.autograd.Function):
    """Reparameterised sample and posterior-form KL of a (mu | logvar) bottleneck head, one HIP pass each way (csrc/reparam.hip).  BUILD-DEFINED: the reference's
    VAE.forward has no such step (models/vae.py:90-98); models/vae.py VAE(reparameterize=True) is the caller.  moments [rows, 2C], eps [rows, C] f32 or None
    (posterior mode) -> (z [rows, C], kl scalar = mean over latents of mean over rows, per-latent kl [C+1] detached)."""

    @staticmethod
    def forward(ctx, moments, eps):
        moments = _c(moments)
        z, kl = ops.reparam_kl_fwd(moments, eps)
        ctx.save_for_backward(moments, eps)
        ctx.mark_non_differentiable(kl)
        return z, kl[-1].clone(), kl

    @staticmethod
    def backward(ctx, dz, g_kl, _):
        moments, eps = ctx.saved_tensors
        dz = None if dz is None else _c(dz.to(moments.dtype))
        g = None if g_kl is None else g_kl.detach().float().reshape(1).contiguous()
        if dz is None and g is None:
            return torch.zeros_like(moments), None
        return ops.reparam_kl_bwd(moments, eps, dz, g, 1.0 if g is not None else 0.0), None


# ---- trainable ViT encoder block (timm VisionTransformer block reached through models/vae.py:47-53; stages with a trainable encoder:
# train_dmd.py:349,519) ------------------------------------------------------------------------------------------------------------
def _bf(w: torch.Tensor) -> torch.Tensor:
    """bf16 copy of an f32 Linear parameter -- what autocast(bf16) feeds the GEMM -- cached until the parameter changes."""
    sh = getattr(w, "_dmvae_shadow", None)
    if sh is not None:                                   # optim.FlatParams.enable_bf16_shadow: kept current by the fused optimiser step
        ver = (w.data_ptr(), w._version)
        if w._dmvae_shadow_ver != ver:                   # changed by something else (load_state_dict, in-place op) or re-pointed: convert again
            if w._dmvae_shadow_ver[0] != ver[0] or sh.shape != w.shape:      # the parameter no longer lives in the flat buffer
                sh = None
            else:
                sh.copy_(w.detach())
                w._dmvae_shadow_ver = ver
        if sh is not None:
            return sh
    cache = getattr(w, "_dmvae_bf16", None)
    ver = (w.data_ptr(), w._version, _epoch_of(w))
    if cache is not None and cache[0] == ver:
        return cache[1]
    v = w.detach().to(bf16)
    try:
        w._dmvae_bf16 = (ver, v)
    except AttributeError:
        pass
    return v


def _bf_t(w: torch.Tensor) -> torch.Tensor:
    """bf16 copy of the TRANSPOSE of a Linear weight [out, in] -- the operand that makes the input gradient dX = dY . W the same NT GEMM as the forward --,
    cached until the weight changes: K-tile-major [out / 32, in, 32], one tiled-transpose launch from the weight's bf16 copy (`_bf`: the optimiser's shadow
    where there is one) when `out` is a multiple of 32; else row-major [in, out] from the element-wise pack."""
    cout, cin = w.shape
    if cout % 32 == 0 and cin % 8 == 0 and not parity.on():
        st = getattr(w, "_dmvae_shadow_t", None)
        if st is not None and w._dmvae_shadow_t_key == (w.data_ptr(), w._version, _epoch_of(w)):
            return st                                    # optim.FlatParams.enable_transposed_shadow: refreshed by one batched launch after the optimiser step
        cache = getattr(w, "_dmvae_bft", None)
        ver = (w.data_ptr(), _ver(w), _epoch_of(w))
        if cache is not None and cache[0] == ver:
            return cache[1]
        v = ops.linear_weight_t_kmajor(_bf(w))
        try:
            w._dmvae_bft = (ver, v)
        except AttributeError:
            pass
        return v
    return packed(w, True).view(cin, cout)


def _bf_km(w: torch.Tensor) -> torch.Tensor:
    """K-tile-major bf16 copy [in / 32, out, 32] of a FROZEN Linear weight [out, in] (f32 parameter or a bf16 shadow module's weight): packed once, then every K
    tile the GEMM fetches is one contiguous run of whole 128-B lines."""
    cout, cin = w.shape
    return packed(w, kmajor=True, frozen=True)._dmvae_kmajor.view(cin // 32, cout, 32)


SPLITK = int(os.environ.get("DMVAE_SPLITK", "3"))                # DitStackFn: parts of the reduction for its few-tile deep-K GEMMs (0: off; tests compare)
SPLITK_MIN_K = int(os.environ.get("DMVAE_SPLITK_MINK", "3072"))
SPLITK_FUSED = os.environ.get("DMVAE_SPLITK_FUSED", "1") != "0"  # the parts summed inside the GEMM by the last one to arrive (ops.linear_sk, round 6) instead of slabs + a sum pass (ops.linear_splitk)
_SPLITK_ACTIVE = [0]      # > 0 only inside DitStackFn's forward / backward: the inference route keeps one accumulation order for every batch size


def _use_splitk(m: int, n: int, k: int) -> int:
    """Parts to cut this GEMM's reduction into: LightningDiT-XL/1 at batch 16 -- M = 4096 rows x N = 1152 columns are 80 tiles of 256 x 256 (160 of 256 x 128: 60 % of the
    chip for the whole reduction); with K >= 3072 three parts make 240 work units of a third of the length (w3: 47 -> ~41 us incl. the slab sum, the input gradient of
    w12: 88 -> ~70)."""
    s_ = _SPLITK_ACTIVE[0]
    if s_ < 2 or parity.on() or not (2048 <= m <= 6144) or k < SPLITK_MIN_K:
        return 0
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    if tiles > 100:
        return 0
    if SPLITK_FUSED:
        return s_ if ops.linear_sk_supported(m, n, k, s_) else 0
    return s_ if ops.linear_splitk_supported(m, n, k, s_) else 0


def _splitk_linear(x2: torch.Tensor, w: torch.Tensor, bias, parts: int) -> torch.Tensor:
    return ops.linear_sk(x2, w, bias, splits=parts) if SPLITK_FUSED else ops.linear_splitk(x2, w, bias, parts)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None, act: int = ops.ACT_NONE) -> torch.Tensor:
    """nn.Linear under autocast(bf16) on this build's GEMM kernels (no vendor library): x [..., K] bf16; w [N, K] and b [N] the f32 parameters (their bf16
    copies are cached / shadowed, `_bf`) or already-bf16 tensors (a frozen bf16 shadow module).  f32 accumulation, bias added in f32, bf16 result, optional
    fused GELU / SiLU (on the bf16-rounded pre-activation: the reference's Linear -> activation pair).  Shapes the large-tile kernel does not take
    (csrc/gemm_pp.hip: K < 384 or not a multiple of 32, N not a multiple of 8, fewer than 64 rows) run on the small batched NT kernel.
    Reference sites: timm blocks via models/vae.py:47-53; diffusion/lightningdit/lightningdit.py:34-93,173-252; swiglu_ffn.py:15-36."""
    bb = None if b is None else (b if b.dtype == bf16 else _bf(b))
    k = x.shape[-1]
    n = w.shape[0]
    m = x.numel() // k
    x = _c(x)
    if x.dtype != bf16:
        x = x.to(bf16)
    if m <= 64 and act in (ops.ACT_NONE, ops.ACT_SILU) and ops.linear_rows_supported(m, n, k) and not parity.on():
        # one row per SAMPLE (adaLN modulations, timestep embedder): the weight-streaming kernel, csrc/linear_rows.hip
        return ops.linear_rows(x.view(m, k), (w if w.dtype == bf16 else _bf(w)).view(n, k), bb, act).view(*x.shape[:-1], n)
    sk = _use_splitk(m, n, k) if act == ops.ACT_NONE else 0
    if sk:
        return _splitk_linear(x.view(m, k), (w if w.dtype == bf16 else _bf(w)).view(n, k), bb, sk).view(*x.shape[:-1], n)
    if ops.linear_supported(m, n, k) and (act != ops.ACT_SWIGLU or n % 16 == 0):
        # frozen weights -- an nn.Parameter that is not trainable AND not owned by one of this build's optimisers (a student DiT switched to requires_grad
        # False for the DMD loss's evaluations still changes every few steps: it keeps the row-major shadow its optimiser maintains) -- : the K-tile-major
        # copy, packed once per (storage, version).  Anything that is not the parameter object itself -- a view of an optimiser's bf16 shadow, which the fused
        # AdamW step rewrites through raw pointers without moving data_ptr or _version -- is read as it is, live.
        if isinstance(w, torch.nn.Parameter) and not w.requires_grad and not hasattr(w, "_dmvae_epoch") and w.dim() == 2:
            return ops.linear_bf16(x, _bf_km(w), bb, act)
        return ops.linear_bf16(x, (w if w.dtype == bf16 else _bf(w)).view(n, k), bb, act)
    wb = w if w.dtype == bf16 else _bf(w)
    x2, w2 = x.view(m, k), wb.view(n, k)
    if k % 32 != 0 or n % 4 != 0:      # reduced test models (a patch embedding with K = 8): zero-pad the reduction / the output columns up to the kernel's steps
        kp, np_ = (-k) % 32, (-n) % 4
        x2 = torch.nn.functional.pad(x2, (0, kp))
        w2 = torch.nn.functional.pad(w2, (0, kp, 0, np_))
        bb = None if bb is None else torch.nn.functional.pad(bb, (0, np_))
    y = ops.gemm_nt(x2, w2, None if bb is None else bb.float())[:, :n]
    if act == ops.ACT_GELU:
        y = ops.gelu(y)
    elif act == ops.ACT_SILU:
        y = ops.silu(y)
    elif act == ops.ACT_SWIGLU:
        y = ops.swiglu(_c(y))
    return y.reshape(*x.shape[:-1], y.shape[-1])


SWIGLU_IN_W12 = os.environ.get("DMVAE_SWIGLU_IN_W12", "1") != "0"      # the training route's w12 Linear writes silu(x1) * x2 AND the pre-activation in one launch (0: Linear, then the swiglu pass; tests compare)


def linear_swiglu(a2: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor]):
    """SwiGLUFFN's first half on the training route (swiglu_ffn.py:31-36): -> (x12 = bf16(a2 @ w12^T + b), g = silu(x1) * x2), both kept for the backward.  One launch of
    the Linear GEMM with the gated epilogue that also stores the pre-activation (ops.linear_swiglu_pre) where the large-tile kernel takes the shape -- the bits of
    the Linear followed by the swiglu pass, which is the fallback."""
    k, n = a2.shape[-1], w.shape[0]
    m = a2.numel() // k
    if SWIGLU_IN_W12 and not parity.on() and a2.dtype == bf16 and n % 16 == 0 and ops.linear_supported(m, n, k) and not _use_splitk(m, n, k):
        bb = None if b is None else (b if b.dtype == bf16 else _bf(b))
        g, x12 = ops.linear_swiglu_pre(_c(a2), (w if w.dtype == bf16 else _bf(w)).view(n, k), bb)
        return x12, g
    x12 = linear(a2, w, b)
    return x12, ops.swiglu(x12)


WGRAD_GROUPED = True      # DitStackFn / VitBlockFn: the Linear weight gradients of a stack / block as ONE grouped launch (ops.linear_wgrad_grouped) instead of one split-K call each


def _lin_grads(dy2: torch.Tensor, x2: torch.Tensor, w: torch.Tensor, b: torch.Tensor, need_dx: bool = True, defer: Optional[list] = None):
    """Gradients of y = x @ w^T + b for bf16 operands [rows, .]: dW and db from the split-K weight-gradient kernel (f32 results, bias
    gradient fused on the matrix pipe; the 1x1 case of the conv wgrad) when its shape constraints hold, else a library GEMM + column sum;
    dx = dy @ w on the Linear GEMM kernel against the transposed bf16 copy of w (bf16 like the reference's autocast backward)."""
    rows, cout = dy2.shape
    cin = x2.shape[1]
    if defer is not None and WGRAD_GROUPED and not parity.on() and ops.linear_wgrad_grouped_supported(rows, cout, cin):
        # the caller collects (dy, x, dW, db) and launches ALL of them together when its backward pass ends: the destinations are returned now, filled then
        dst_w, dst_b = _dst(w), _dst(b)
        dw = dst_w.view(cout, cin) if dst_w is not None else torch.empty(cout, cin, dtype=f32, device=dy2.device)
        db = None if b is None else (dst_b if dst_b is not None else torch.empty(cout, dtype=f32, device=dy2.device))
        defer.append((_c(dy2), _c(x2), dw, db))
    elif rows <= 64 and cin % 8 == 0 and not parity.on():
        # one row per SAMPLE (adaLN modulations, timestep embedder): the gradient is an outer-product sum of <= 64 terms, bound by writing it (csrc/linear_rows.hip)
        dst_w = _dst(w)
        dw, db = ops.linear_rows_wgrad(_c(dy2), _c(x2), need_bias=b is not None, dw_out=None if dst_w is None else dst_w.view(cout, cin), db_out=_dst(b))
    elif cin % 8 == 0 and cout % 8 == 0:
        dst_w = _dst(w)
        dw, db = ops.conv2d_nhwc_wgrad(dy2.view(1, 1, rows, cout), x2.view(1, 1, rows, cin), 1,
                                       dw_out=None if dst_w is None else dst_w.view(cout, cin, 1, 1), db_out=_dst(b))
        dw = dw.view(cout, cin)
    elif not parity.on():
        # widths that are not multiples of 8 (the 2-D toy's embedders: 2 input / output channels, toy_example_2d/dmd.py:436-440): both operands zero-padded to the
        # weight-gradient kernel's granule -- zero columns contribute zero rows / columns of dW, which are cut off
        pc, pi = (-cout) % 8, (-cin) % 8
        dwp, dbp = ops.conv2d_nhwc_wgrad(torch.nn.functional.pad(dy2, (0, pc)).view(1, 1, rows, cout + pc), torch.nn.functional.pad(x2, (0, pi)).view(1, 1, rows, cin + pi), 1)
        dw, db = dwp.view(cout + pc, cin + pi)[:cout, :cin].contiguous(), (None if dbp is None else dbp[:cout].contiguous())
        dst_w, dst_b = _dst(w), _dst(b)
        if dst_w is not None:
            dst_w.view(cout, cin).copy_(dw)
            dw = dst_w.view(cout, cin)
        if dst_b is not None and db is not None:
            dst_b.copy_(db)
            db = dst_b
    else:
        # widths that are not multiples of 8: no kernel of this build takes them -- a library GEMM only behind the explicit opt-in (no silent rocBLAS)
        from ._stock import require_opt_in
        require_opt_in("functional._lin_grads (weight gradient)", f"Linear {cout} x {cin}: in / out features must be multiples of 8 for the weight-gradient kernels")
        dw, db = (dy2.t() @ x2).float(), dy2.float().sum(0)
    if not need_dx:
        return None, dw, db
    if rows <= 64 and ops.linear_rows_supported(rows, cin, cout) and not parity.on():
        # per-sample rows: dX = dY . W against the transposed copy -- K-tile-major, one 8-us tiled transpose of the bf16 weight per optimiser step (`_bf_t`),
        # where the out-feature count allows; else the row-major transposed pack from the f32 master
        wt = _bf_t(w)
        return ops.linear_rows(_c(dy2), wt if wt.dim() == 3 else wt.view(cin, cout)), dw, db
    if ops.linear_supported(rows, cin, cout) and not parity.on():
        sk = _use_splitk(rows, cin, cout)
        if sk:
            return _splitk_linear(_c(dy2), _bf_t(w), None, sk), dw, db
        return ops.linear_bf16(_c(dy2), _bf_t(w)), dw, db                 # dX = dY . W as an NT GEMM against the transposed copy
    if cout % 32 == 0 and cin % 4 == 0:
        return ops.gemm_nt(_c(dy2), packed(w, True).view(cin, cout)), dw, db
    if cin % 4 == 0:
        # few out features (the reduced test models' 8-channel output head): the reduction zero-padded to the small batched kernel's 32-wide K step
        pad = (-cout) % 32
        wt = torch.nn.functional.pad(_bf(w).t(), (0, pad)).contiguous()                   # [in, out + pad]
        return ops.gemm_nt(torch.nn.functional.pad(_c(dy2), (0, pad)), wt), dw, db
    if not parity.on():
        # in features not a multiple of 4 (the 2-D toy's 2-channel patch embedding): W^T zero-padded to the small batched kernel's granules -- output columns to 4,
        # the reduction to 32 --, the padding columns cut off
        pad_k, pad_n = (-cout) % 32, (-cin) % 4
        wt = torch.nn.functional.pad(_bf(w).t(), (0, pad_k, 0, pad_n)).contiguous()          # [in + pad_n, out + pad_k]
        return ops.gemm_nt(torch.nn.functional.pad(_c(dy2), (0, pad_k)), wt)[:, :cin].contiguous(), dw, db
    from ._stock import require_opt_in
    require_opt_in("functional._lin_grads (input gradient)", f"Linear {cout} x {cin}: the input-gradient GEMM kernels need in features % 4 == 0")
    return dy2 @ _bf(w), dw, db


class LinearFn(torch.autograd.Function):
    """nn.Linear under autocast(bf16) with gradients, entirely on this build's kernels: forward `linear`, backward `_lin_grads` (weight / bias gradient from
    the split-K kernel in f32, input gradient as an NT GEMM against the transposed bf16 weight copy).  x [..., K] (any float dtype: cast to bf16 like autocast
    does), w [N, K] / b [N] f32 parameters.  For the Linear layers outside the transformer-block Functions: patch embeddings, adaLN modulations, output heads."""

    @staticmethod
    def forward(ctx, x, w, b):
        xb = _c(x).to(bf16)
        ctx.save_for_backward(xb, w)
        ctx.bias = b
        ctx.x_dtype = x.dtype
        return linear(xb, w, b)

    @staticmethod
    def backward(ctx, dy):
        xb, w = ctx.saved_tensors
        n, k = w.shape[0], xb.shape[-1]
        rows = xb.numel() // k
        # the PARAMETER itself where it is 2-D: a view would shed what is tagged on it -- the optimiser's bf16 shadow (a view's transposed operand is cast and
        # transposed again on every call: 28 casts of a 6912 x 1152 weight per DiT step), the flat-buffer gradient destination, the cached operands
        dx, dw, db = _lin_grads(_c(dy).to(bf16).view(rows, n), xb.view(rows, k), w if w.dim() == 2 else w.view(n, k), ctx.bias, need_dx=ctx.needs_input_grad[0])
        if ctx.bias is None:
            db = None
        return (None if dx is None else dx.view(xb.shape).to(ctx.x_dtype)), dw.view(w.shape), db


class SiluFn(torch.autograd.Function):
    """nn.SiLU on a bf16 tensor (TimestepEmbedder.mlp[1] under autocast, lightningdit.py:108-112): csrc/misc.hip silu_fwd / silu_bwd."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.silu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.silu_bwd(x, _c(dy).to(bf16))


ATTN_BWD_LSE = True        # the forward attention kernels also write the row statistics and the backward takes the eight-wave form built on them (False: recomputed; tests compare)
QK_UNPADDED = os.environ.get("DMVAE_QK_UNPADDED", "1") != "0"      # DitStackFn: q / k (and dq / dk) rows of D = 72 channels instead of 96 zero-padded ones between QK-norm + RoPE and the attention kernels (0: padded, for A/B)
ATTN_BWD_FUSED = True      # False: the GEMM-composed attention backward (probabilities through HBM; the first implementation) -- tests compare the two
THIN_CIN_BWD_AS_GEMM = True      # ConvFn.backward of a 3x3 conv from 32 / 64 channels: weight and input gradient as GEMMs on the im2col form
NORM_CONV_OUT_FUSED_FWD = True      # NormConvOutFn.forward: ops.norm_conv_out_fwd where the shape allows
NORM_CONV_OUT_FUSED_BWD = True      # NormConvOutFn.backward: ops.norm_conv_out_bwd where the shape allows (tests compare it with the stored-operand route)


def _fused_attn_bwd() -> bool:
    return ATTN_BWD_FUSED


def _attention_bwd(qkv: torch.Tensor, do: torch.Tensor, heads: int, scale: float) -> torch.Tensor:
    """d(qkv) of multi-head self-attention from the qkv Linear's output [B,S,3*C] and d(out) [B,S,C] (bf16): the probabilities are
    recomputed (QK^T GEMM + f32 softmax, keys padded to a multiple of 32 and masked), then the four GEMMs of the decoder AttnBlock's
    backward per (batch, head) as batched launches.  Head dim 64."""
    b, s, c3 = qkv.shape
    c = c3 // 3
    hd = c // heads
    sp = (s + 31) // 32 * 32
    bh = b * heads
    buf = torch.zeros(3, bh, sp, hd, dtype=bf16, device=qkv.device)
    buf[:, :, :s] = qkv.view(b, s, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, bh, s, hd)
    q, k, v = buf[0], buf[1], buf[2]
    dop = torch.zeros(bh, sp, hd, dtype=bf16, device=qkv.device)
    dop[:, :s] = do.view(b, s, heads, hd).permute(0, 2, 1, 3).reshape(bh, s, hd)
    sc = ops.gemm_nt(q, k, out_f32=True)                       # [bh, sp, sp]
    if sp != s:
        sc[:, :, s:] = float("-inf")                           # padded keys
    p = ops.softmax_rows(sc, scale)
    dp = ops.gemm_nt(dop, v, out_f32=True)
    ds = ops.softmax_rows_bwd(dp, p, scale)
    dv = ops.gemm_tn(p, dop)                                   # [bh, key, hd]
    dq = ops.gemm_nt(ds, ops.transpose_last2(k))
    dk = ops.gemm_tn(ds, q)
    out = torch.stack([dq[:, :s], dk[:, :s], dv[:, :s]], dim=0).view(3, b, heads, s, hd).permute(1, 3, 0, 2, 4)
    return out.reshape(b, s, c3).contiguous()


class VitBlockFn(torch.autograd.Function):
    """One pre-norm transformer block on the f32 residual stream t [B,S,C]:
        t += ls1 * proj(MHA(LN1(t)));  t += ls2 * fc2(GELU(fc1(LN2(t))))
    with bf16 Linear operands / results (autocast semantics).  Forward = the frozen path's kernels (LayerNorm -> bf16, fused attention,
    LayerScale + residual) plus the HIP GELU; backward = csrc/vit_bwd.hip kernels, the fused attention backward, and for the Linear layers the
    Linear GEMM kernel (csrc/gemm_pp.hip: forward and input gradient) and the split-K weight-gradient kernel."""

    @staticmethod
    def forward(ctx, t, n1w, n1b, qkvw, qkvb, pw, pb, ls1, n2w, n2b, f1w, f1b, f2w, f2b, ls2, heads, eps):
        import torch.nn.functional as F
        b, s, c = t.shape
        hd = c // heads
        hn1 = ops.layernorm_bf16(t, n1w, n1b, eps)
        qkv = linear(hn1, qkvw, qkvb)
        lse = None
        if ATTN_BWD_LSE and hd == 64 and s <= 288 and any(ctx.needs_input_grad):
            o, lse = ops.attention_qkv(qkv, heads, hd ** -0.5, need_lse=True)      # + the row statistics the backward kernel rebuilds P from
        else:
            o = ops.attention_qkv(qkv, heads, hd ** -0.5)
        o2 = linear(o, pw, pb)
        t_mid = ops.scale_residual_(t.clone(), o2, ls1)
        hn2 = ops.layernorm_bf16(t_mid, n2w, n2b, eps)
        h1 = linear(hn2, f1w, f1b)
        g = ops.gelu(h1)
        o3 = linear(g, f2w, f2b)
        # nothing requires a gradient: t_mid is nobody's saved tensor, update it in place.  (The outer grad mode cannot be read here -- Function.forward always runs with
        # grad mode off -- so a no_grad call with trainable weights still clones; such calls no longer come here: vit_fast.trainable_forward_features sends them to
        # the frozen route's fused launches.)
        t_out = ops.scale_residual_(t_mid if not any(ctx.needs_input_grad) else t_mid.clone(), o3, ls2)
        ctx.save_for_backward(t, hn1, qkv, o, o2, t_mid, hn2, h1, g, o3, n1w, qkvw, pw, ls1, n2w, f1w, f2w, ls2)
        ctx.others = (n1b, qkvb, pb, n2b, f1b, f2b, heads, eps)
        ctx.lse = lse
        return t_out

    @staticmethod
    def backward(ctx, dt_out):
        t, hn1, qkv, o, o2, t_mid, hn2, h1, g, o3, n1w, qkvw, pw, ls1, n2w, f1w, f2w, ls2 = ctx.saved_tensors
        n1b, qkvb, pb, n2b, f1b, f2b, heads, eps = ctx.others
        b, s, c = t.shape
        rows = b * s
        dt = dt_out.float().clone()                                   # becomes d(t_mid), then d(t)
        # MLP branch
        do3, dls2 = ops.layerscale_bwd(dt, o3, ls2, dg_out=_dst(ls2))
        pend = []                                                     # this block's four weight gradients: one grouped launch at the end (ops.linear_wgrad_grouped)
        dg, df2w, df2b = _lin_grads(do3.view(rows, c), g.view(rows, -1), f2w, f2b, defer=pend)
        dh1 = ops.gelu_bwd(dg.view_as(h1), h1)
        dhn2, df1w, df1b = _lin_grads(dh1.view(rows, -1), hn2.view(rows, c), f1w, f1b, defer=pend)
        dn2w, dn2b = ops.layernorm_bwd_(dt, dhn2.view(b, s, c), t_mid, n2w, eps, dg_out=_dst(n2w), db_out=_dst(n2b))
        # attention branch
        do2, dls1 = ops.layerscale_bwd(dt, o2, ls1, dg_out=_dst(ls1))
        do, dpw, dpb = _lin_grads(do2.view(rows, c), o.view(rows, c), pw, pb, defer=pend)
        if c // heads == 64 and s <= 288 and _fused_attn_bwd():
            dqkv = ops.attention_bwd_qkv(qkv, o, do.view(b, s, c), heads, (c // heads) ** -0.5, lse=ctx.lse)      # one kernel, nothing S x S in HBM
        else:
            dqkv = _attention_bwd(qkv, do.view(b, s, c), heads, (c // heads) ** -0.5)
        dhn1, dqkvw, dqkvb = _lin_grads(dqkv.view(rows, 3 * c), hn1.view(rows, c), qkvw, qkvb, defer=pend)
        dn1w, dn1b = ops.layernorm_bwd_(dt, dhn1.view(b, s, c), t, n1w, eps, dg_out=_dst(n1w), db_out=_dst(n1b))
        if pend:
            ops.linear_wgrad_grouped(pend)
        return dt, dn1w, dn1b, dqkvw, dqkvb, dpw, dpb, dls1, dn2w, dn2b, df1w, df1b, df2w, df2b, dls2, None, None


class LayerNormBf16Fn(torch.autograd.Function):
    """LayerNorm of the f32 residual stream with a bf16 result (the encoder's final norm feeding the bottleneck MLP)."""

    @staticmethod
    def forward(ctx, t, w, b, eps):
        ctx.save_for_backward(t, w)
        ctx.others = (b, eps)
        return ops.layernorm_bf16(t, w, b, eps)

    @staticmethod
    def backward(ctx, dy):
        t, w = ctx.saved_tensors
        b, eps = ctx.others
        dt = torch.zeros_like(t)
        dw, db = ops.layernorm_bwd_(dt, _c(dy).to(bf16), t, w, eps, dg_out=_dst(w), db_out=_dst(b))
        return dt, dw, db, None


# ---- LightningDiT block with gradients (the student's training turn, train_dmd.py:565-575) -------------------------------------------
class DitBlockFn(torch.autograd.Function):
    """LightningDiTBlock.forward (lightningdit.py:236-250) on the f32 residual stream h [B,N,C] with the adaLN chunks `mod` [B,6C] (bf16, computed by
    stock autograd outside): forward = the inference path's kernels (csrc/dit.hip), backward = their backward kernels, the GEMM-composed
    attention backward and `_lin_grads` for the four Linear layers.  Returns the new residual stream; gradients flow to h, mod and all block
    parameters."""

    @staticmethod
    def forward(ctx, h, mod, n1w, qkvw, qkvb, qnw, knw, pw, pb, n2w, w12w, w12b, w3w, w3b, cos, sin, heads, eps):
        import torch.nn.functional as F
        b, n, c = h.shape
        d = c // heads
        mod = _c(mod)
        a1 = ops.rmsnorm_modulate(h, n1w, mod, 0, c, eps)
        qkv = linear(a1, qkvw, qkvb)
        q, k, v = ops.qknorm_rope(qkv, qnw, knw, cos, sin, heads, eps)
        fused = ops.attention_heads_supported(n, d) and _fused_attn_bwd()
        if fused:       # the inference kernel; its backward recomputes the probabilities in registers (csrc/attention_bwd.hip)
            p = None
            o = ops.attention_heads(q, k, v, b, d ** -0.5)
        else:
            p = ops.softmax_rows(ops.gemm_nt(q, k, out_f32=True), d ** -0.5)
            o = ops.gemm_nt(p, ops.transpose_last2(v)).view(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)
        o2 = linear(o, pw, pb)
        h_mid, a2 = ops.gated_residual_out(h, o2, mod, 2 * c, n2w, mod, 3 * c, 4 * c, eps)       # h itself is saved for the backward pass
        x12, g = linear_swiglu(a2, w12w, w12b)
        o3 = linear(g, w3w, w3b)
        h_out, _ = ops.gated_residual_out(h_mid, o3, mod, 5 * c)
        ctx.save_for_backward(h, mod, a1, qkv, q, k, v, p, o, o2, h_mid, a2, x12, g, o3, n1w, qkvw, qnw, knw, pw, n2w, w12w, w3w, cos, sin)
        ctx.others = (qkvb, pb, w12b, w3b, heads, eps)
        ctx.fused_attn = fused
        return h_out

    @staticmethod
    def backward(ctx, dh_out):
        import torch.nn.functional as F
        h, mod, a1, qkv, q, k, v, p, o, o2, h_mid, a2, x12, g, o3, n1w, qkvw, qnw, knw, pw, n2w, w12w, w3w, cos, sin = ctx.saved_tensors
        qkvb, pb, w12b, w3b, heads, eps = ctx.others
        b, n, c = h.shape
        d, rows = c // heads, b * n
        dp_ = q.shape[-1]
        dt = dh_out.float().clone()                                 # becomes d(h_mid), then d(h)
        dmod = torch.zeros(b, mod.shape[1], dtype=f32, device=h.device)
        # MLP branch
        do3 = ops.gated_residual_bwd(dt, o3, mod, dmod, 5 * c)
        dg, dw3, db3 = _lin_grads(do3.view(rows, c), g.view(rows, -1), w3w, w3b)
        dx12 = ops.swiglu_bwd(dg.view_as(g), x12)
        da2, dw12, db12 = _lin_grads(dx12.view(rows, -1), a2.view(rows, c), w12w, w12b)
        dn2w = ops.rmsnorm_modulate_bwd_(dt, da2.view(b, n, c), h_mid, n2w, mod, dmod, 3 * c, 4 * c, eps, dw_out=_dst(n2w))
        # attention branch
        do2 = ops.gated_residual_bwd(dt, o2, mod, dmod, 2 * c)
        do, dpw, dpb = _lin_grads(do2.view(rows, c), o.view(rows, c), pw, pb)
        if ctx.fused_attn:
            dq, dk, dv = ops.attention_bwd_heads(q, k, v, o, do.view(b, n, c), b, d ** -0.5)
        else:
            do_h = do.view(b, n, heads, d).permute(0, 2, 1, 3).reshape(b * heads, n, d)
            pad = dp_ - d                                                   # head dim padded to the GEMM kernel's 32-wide K step (72 -> 96)
            do_p, v_p = (F.pad(do_h, (0, pad)), F.pad(v, (0, pad))) if pad else (do_h.contiguous(), v)
            ds = ops.softmax_rows_bwd(ops.gemm_nt(do_p, v_p, out_f32=True), p, d ** -0.5)
            dv = ops.gemm_tn(p, do_h.contiguous())                          # [B*H, key, D]
            dq = ops.gemm_nt(ds, ops.transpose_last2(k))                    # [B*H, N, Dp]
            dk = ops.gemm_tn(ds, q)
        dqkv, dqnw, dknw = ops.qknorm_rope_bwd(dq, dk, dv, qkv, qnw, knw, cos, sin, heads, eps, dqw_out=_dst(qnw), dkw_out=_dst(knw))
        da1, dqkvw, dqkvb = _lin_grads(dqkv.view(rows, 3 * c), a1.view(rows, c), qkvw, qkvb)
        dn1w = ops.rmsnorm_modulate_bwd_(dt, da1.view(b, n, c), h, n1w, mod, dmod, 0, c, eps, dw_out=_dst(n1w))
        return (dt, dmod.to(mod.dtype), dn1w, dqkvw, dqkvb, dqnw, dknw, dpw, dpb, dn2w, dw12, db12, dw3, db3, None, None, None, None)


# gradient tensors this build's backward Functions created and handed to autograd for the f32 residual stream: the next Function down the chain may update such a
# tensor in place instead of cloning it (nobody else holds it: autograd's input buffer passes a single incoming gradient through as it is).  A strong reference is
# kept until the tensor is taken, so its address cannot be reused by another tensor in between; cleared at the start of every training forward.
_OWNED_GRADS = {}


def _own(t: torch.Tensor) -> torch.Tensor:
    if not _OWNED_GRADS:      # first hand-over of this backward pass: drop whatever is left when the pass ends (the last segment's d h_in is taken by nobody)
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_OWNED_GRADS.clear)
        except RuntimeError:  # not inside a backward pass (a test calling backward() of a Function by hand)
            pass
    _OWNED_GRADS[t.data_ptr()] = t
    return t


def _observed(out_ref) -> bool:
    """Whether somebody besides the consuming node can see the gradient of the tensor `out_ref` (a weakref to a Function's output) points at: a tensor hook or
    retain_grad() on it hands the SAME gradient tensor to user code, which an in-place update would then overwrite with d(input)."""
    out = out_ref() if out_ref is not None else None
    if out is None:
        return False
    return bool(getattr(out, "retains_grad", False)) or bool(getattr(out, "_backward_hooks", None)) or bool(getattr(out, "_post_accumulate_grad_hooks", None))


def _take_owned(t: torch.Tensor, out_ref=None) -> bool:
    """True when `t` is a gradient buffer THIS build produced for the node that now receives it (`_own`) and nobody else can observe it: the caller may then update
    it in place instead of cloning (B x N x C x 4 bytes per segment).  Autograd's rule is that backward must not modify its incoming gradients; the cases where that
    is observable -- hooks / retain_grad on the producing Function's output -- are excluded through `out_ref`, and `torch.autograd.grad(..., inputs=[h])` on the
    stack's output is not part of any route of this package (lightningdit_fast.forward_train feeds it to the next segment or the final layer only)."""
    o = _OWNED_GRADS.pop(t.data_ptr(), None)
    if o is None or _observed(out_ref):
        return False
    return o.shape == t.shape and t.dtype == f32 and o.dtype == f32 and t.is_contiguous() and o.untyped_storage().data_ptr() == t.untyped_storage().data_ptr()


def _with_splitk(fn):
    """Run `fn` with DitStackFn's split-K choice active and ALWAYS switch it off again -- an exception inside (out of memory, a failed check()) must not leave the
    process-wide flag set: functional.linear on the inference route would then change accumulation order with M and break '2B call == two B calls' bit for bit."""
    import functools

    @functools.wraps(fn)
    def scoped(*a, **kw):
        _SPLITK_ACTIVE[0] = SPLITK
        try:
            return fn(*a, **kw)
        finally:
            _SPLITK_ACTIVE[0] = 0
    return scoped


DIT_STACK_PARAMS_PER_BLOCK = 14


def dit_stack_supported(b: int, n: int, c: int, heads: int) -> bool:
    """Shapes `DitStackFn` takes (else the per-block `DitBlockFn` route): the fused attention kernels' token / head-dim range, at most 64 samples per call (the
    per-sample Linears' row limit), outside the fp32 parity mode."""
    return bool(ops.attention_heads_supported(n, c // heads) and _fused_attn_bwd() and b <= 64 and c % 32 == 0 and not parity.on())


class DitStackFn(torch.autograd.Function):
    """ALL LightningDiT blocks (lightningdit.py:236-250 x depth, with each block's adaLN_modulation Linear, :236-240) as ONE autograd node on the f32 residual stream
    h [B,N,C]; `sc` = SiLU(conditioning) [B,C] f32 (adaLN_modulation[0] of every block: the same values).  Same arithmetic as a chain of `DitBlockFn` fed by `LinearFn`
    modulations; what the single node buys is launch structure:
      forward   the 28 adaLN Linears as one batched launch; a block's closing gated residual folded into the next block's RMSNorm + modulate (dit.hip, as on the inference route);
      backward  one pass over the residual-stream gradient per sub-layer boundary (RMSNorm backward + the next gated residual's backward: csrc/dit_stack.hip) with ALL
                per-sample / per-channel reductions deferred to one finalize launch; the adaLN weight gradients of all blocks in one MFMA launch, their input gradient
                in one batched launch; no per-block clone / cast / zero-fill.
    Parameters per block, in order: norm1.weight, qkv.weight, qkv.bias, q_norm.weight, k_norm.weight, proj.weight, proj.bias, norm2.weight, w12.weight, w12.bias,
    w3.weight, w3.bias, adaLN_modulation[1].weight, adaLN_modulation[1].bias."""

    @staticmethod
    @_with_splitk
    def forward(ctx, h, sc, cos, sin, heads, eps, *params):
        P = DIT_STACK_PARAMS_PER_BLOCK
        nl = len(params) // P
        b, n, c = h.shape
        d = c // heads
        h = _c(h)
        scb = _c(sc).to(bf16)
        blocks = [params[i * P:(i + 1) * P] for i in range(nl)]
        mod_all = ops.linear_rows_batched(scb, [_bf(bp[12]) for bp in blocks], [_bf(bp[13]) for bp in blocks])          # [L, B, 6C] bf16
        saved, lses = [], []
        h_in, h_mid, o3 = h, None, None
        for i, (n1w, qkvw, qkvb, qnw, knw, pw, pb, n2w, w12w, w12b, w3w, w3b, _aw, _ab) in enumerate(blocks):
            mod = mod_all[i]
            if i == 0:
                a1 = ops.rmsnorm_modulate(h_in, n1w, mod, 0, c, eps)
            else:
                h_in, a1 = ops.gated_residual_out(h_mid, o3, mod_all[i - 1], 5 * c, n1w, mod, 0, c, eps)
            qkv = linear(a1, qkvw, qkvb)
            q, k, v = ops.qknorm_rope(qkv, qnw, knw, cos, sin, heads, eps, padded=not QK_UNPADDED)
            if ATTN_BWD_LSE:
                o, lse = ops.attention_heads(q, k, v, b, d ** -0.5, need_lse=True)
                lses.append(lse)
            else:
                o = ops.attention_heads(q, k, v, b, d ** -0.5)
            o2 = linear(o, pw, pb)
            h_mid, a2 = ops.gated_residual_out(h_in, o2, mod, 2 * c, n2w, mod, 3 * c, 4 * c, eps)
            x12, g = linear_swiglu(a2, w12w, w12b)
            o3 = linear(g, w3w, w3b)
            saved += [h_in, a1, qkv, q, k, v, o, o2, h_mid, a2, x12, g, o3]
        h_out, _ = ops.gated_residual_out(h_mid, o3, mod_all[nl - 1], 5 * c)
        ctx.save_for_backward(scb, mod_all, cos, sin, *saved, *params)
        ctx.cfg = (nl, heads, eps, sc.dtype)
        ctx.lses = lses
        ctx.out_ref = weakref.ref(h_out)
        return h_out

    @staticmethod
    @_with_splitk
    def backward(ctx, dh_out):
        P = DIT_STACK_PARAMS_PER_BLOCK
        nl, heads, eps, sc_dtype = ctx.cfg
        scb, mod_all, cos, sin = ctx.saved_tensors[:4]
        acts = ctx.saved_tensors[4:4 + 13 * nl]
        params = ctx.saved_tensors[4 + 13 * nl:]
        b, n, c = acts[0].shape
        d, rows = c // heads, b * n
        dt = dh_out if _take_owned(dh_out, ctx.out_ref) else _c(dh_out).float().clone()      # becomes d(h_mid), d(h) of every block in turn
        S = ops.DitStackBwd(nl, b, n, c, heads, dt.device)
        grads = [None] * (P * nl)
        fresh = lambda p: _dst(p) if _dst(p) is not None else torch.empty(p.shape, dtype=f32, device=dt.device)
        norm_dws, qn_dws, kn_dws = [None] * (2 * nl), [None] * nl, [None] * nl
        pend = []      # (dy, x, dW, db) of every block Linear: ONE grouped weight-gradient launch when the chain is done -- 4 x depth problems, each unsplit, fill the chip
        do3 = S.boundary(2 * nl, dt, y=acts[13 * (nl - 1) + 12], gate_mod=mod_all[nl - 1], gate_off=5 * c)
        for i in range(nl - 1, -1, -1):
            h_in, a1, qkv, q, k, v, o, o2, h_mid, a2, x12, g, o3 = acts[13 * i:13 * i + 13]
            n1w, qkvw, qkvb, qnw, knw, pw, pb, n2w, w12w, w12b, w3w, w3b, _aw, _ab = params[P * i:P * i + P]
            mod = mod_all[i]
            G = grads
            dg, G[P * i + 10], G[P * i + 11] = _lin_grads(do3.view(rows, c), g.view(rows, -1), w3w, w3b, defer=pend)
            dx12 = ops.swiglu_bwd(dg.view_as(g), x12)
            da2, G[P * i + 8], G[P * i + 9] = _lin_grads(dx12.view(rows, -1), a2.view(rows, c), w12w, w12b, defer=pend)
            do2 = S.boundary(2 * i + 1, dt, da=da2.view(b, n, c), x=h_mid, w=n2w, mod=mod, scale_off=4 * c, eps=eps, y=o2, gate_mod=mod, gate_off=2 * c)
            do, G[P * i + 5], G[P * i + 6] = _lin_grads(do2.view(rows, c), o.view(rows, c), pw, pb, defer=pend)
            dq, dk, dv = ops.attention_bwd_heads(q, k, v, o, do.view(b, n, c), b, d ** -0.5, lse=ctx.lses[i] if ctx.lses else None)
            dqkv = S.qknorm_rope_bwd(i, dq, dk, dv, qkv, qnw, knw, cos, sin, eps)
            da1, G[P * i + 1], G[P * i + 2] = _lin_grads(dqkv.view(rows, 3 * c), a1.view(rows, c), qkvw, qkvb, defer=pend)
            if i > 0:
                do3 = S.boundary(2 * i, dt, da=da1.view(b, n, c), x=h_in, w=n1w, mod=mod, scale_off=c, eps=eps, y=acts[13 * (i - 1) + 12], gate_mod=mod_all[i - 1],
                                 gate_off=5 * c)
            else:
                S.boundary(0, dt, da=da1.view(b, n, c), x=h_in, w=n1w, mod=mod, scale_off=c, eps=eps)
            norm_dws[2 * i], norm_dws[2 * i + 1], qn_dws[i], kn_dws[i] = fresh(n1w), fresh(n2w), fresh(qnw), fresh(knw)
            G[P * i + 0], G[P * i + 7], G[P * i + 3], G[P * i + 4] = norm_dws[2 * i], norm_dws[2 * i + 1], qn_dws[i], kn_dws[i]
        if pend:
            ops.linear_wgrad_grouped(pend)
        dmod = torch.empty_like(mod_all)
        S.finalize(dmod, norm_dws, qn_dws, kn_dws)
        # the adaLN Linears of every block: weight / bias gradients in one launch, the input gradient d sc = sum_l d mod_l . W_l as one batched launch + a sum over layers
        aws = [params[P * i + 12] for i in range(nl)]
        abs_ = [params[P * i + 13] for i in range(nl)]
        dws, dbs = [fresh(w) for w in aws], [fresh(bb) for bb in abs_]
        ops.linear_rows_wgrad_batched(dmod, ops.rows_transposed(scb), [t.view(t.shape[0], -1) for t in dws], dbs)
        for i in range(nl):
            grads[P * i + 12], grads[P * i + 13] = dws[i], dbs[i]
        dsc = None
        if ctx.needs_input_grad[1]:
            wts = [_bf_t(w) for w in aws]
            if all(t.dim() == 3 for t in wts):
                dsc = ops.linear_rows_batched(dmod, wts, None, out_f32=True).sum(0).to(sc_dtype)
            else:      # out features not a multiple of 32: no K-tile-major copy -- per layer on the row-major transposed pack
                dsc = sum(ops.linear_rows(dmod[i], wts[i].view(c, -1), out_f32=True) for i in range(nl)).to(sc_dtype)
        return (_own(dt), dsc, None, None, None, None, *grads)




class RmsnormModulateFn(torch.autograd.Function):
    """bf16( RMSNorm(h) * bf16(1 + scale) + shift ) with gradients to h, the norm weight and the adaLN chunks (FinalLayer, lightningdit.py:266-273)."""

    @staticmethod
    def forward(ctx, h, w, mod, shift_off, scale_off, eps):
        mod = _c(mod)
        ctx.save_for_backward(h, w, mod)
        ctx.cfg = (shift_off, scale_off, eps)
        return ops.rmsnorm_modulate(h, w, mod, shift_off, scale_off, eps)

    @staticmethod
    def backward(ctx, da):
        h, w, mod = ctx.saved_tensors
        shift_off, scale_off, eps = ctx.cfg
        dt = torch.zeros_like(h)
        dmod = torch.zeros(mod.shape, dtype=f32, device=h.device)
        dw = ops.rmsnorm_modulate_bwd_(dt, _c(da).to(bf16), h, w, mod, dmod, shift_off, scale_off, eps, dw_out=_dst(w))
        return _own(dt), dw, dmod.to(mod.dtype), None, None, None


class GatedResidualFn(torch.autograd.Function):
    """h + gate * y with gate = mod[:, off : off + C] per sample (lightningdit.py:245,249) on the f32 residual stream; y, mod bf16.  The unfused step of DitBlockFn, for
    routes that compose a block from single Functions (the one-token route of the 2-D toy, lightningdit_fast.forward_tokens1)."""

    @staticmethod
    def forward(ctx, h, y, mod, off):
        out = h.float().clone()
        y, mod = _c(y), _c(mod)
        ops.gated_residual_(out, y, mod, off)
        ctx.save_for_backward(y, mod)
        ctx.off = off
        return out

    @staticmethod
    def backward(ctx, dout):
        y, mod = ctx.saved_tensors
        dout = _c(dout.float())
        dmod = torch.zeros(mod.shape, dtype=f32, device=mod.device)
        dy = ops.gated_residual_bwd(dout, y, mod, dmod, ctx.off)
        return dout, dy, dmod.to(mod.dtype), None


class SwigluFn(torch.autograd.Function):
    """silu(x1) * x2 over [x1 | x2] (swiglu_ffn.py:32-35), bf16, as its own node (csrc/dit.hip::swiglu_kernel / swiglu_bwd_kernel)."""

    @staticmethod
    def forward(ctx, x12):
        x12 = _c(x12)
        ctx.save_for_backward(x12)
        return ops.swiglu(x12)

    @staticmethod
    def backward(ctx, dg):
        (x12,) = ctx.saved_tensors
        return ops.swiglu_bwd(_c(dg).to(bf16), x12)


def to_nhwc_bf16(x: torch.Tensor) -> torch.Tensor:
    """NCHW (any float dtype) -> NHWC bf16 (f32 in parity mode) contiguous, differentiable (boundary plumbing)."""
    return x.permute(0, 2, 3, 1).contiguous().to(parity.act_dtype())


def to_nchw(x: torch.Tensor, dtype=None) -> torch.Tensor:
    y = x.permute(0, 3, 1, 2).contiguous()
    return y if dtype is None else y.to(dtype)
