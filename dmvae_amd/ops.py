"""Thin tensor-level wrappers over the C ABI (include/dmvae_hip.h).

Every function takes CUDA(HIP) tensors, validates dtype/contiguity, launches on the current stream
and returns freshly allocated outputs.  No function here has a CPU path: CPU tensors raise.
Activations are NHWC bf16 ``[N, H, W, C]`` -- or, in the fp32 parity mode (dmvae_amd/parity.py, DMVAE_PARITY=1), NHWC f32: every wrapper
below that sees f32 activations while the mode is on routes the contraction through the SAME kernels on exactly-split bf16 operands and the
elementwise work through the f32 kernels of csrc/parity.hip.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch

from . import _lib, parity
from ._lib import ConvDesc, check

bf16 = torch.bfloat16
f32 = torch.float32

ACT_NONE, ACT_SILU, ACT_RELU, ACT_RELU_GATE, ACT_LEAKY, ACT_GELU = 0, 1, 2, 3, 4, 5
ACT_SWIGLU = 6     # linear_bf16 only: N = 2 H columns [x1 | x2] -> H columns silu(x1) * x2 (swiglu_ffn.py:32-35)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.DmvaeHipError(f"{name}: expected a GPU tensor; dmvae_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_WS = {}

# bench.py sets this to a list to time the dominant kernel with HIP events on the launch stream:
# entries are (name, start_event, end_event, algorithmic_flops)
KERNEL_TIMING = None
_PP_HALO = 3          # csrc/conv_pp.hip::halo_mode: both tile shapes run the kx-halo form (bench.py names the dominant instantiation with it)


def workspace(nbytes: int, device, slot: str = "main") -> torch.Tensor:
    """Persistent per-(device, stream, slot) scratch buffer, grown on demand (never shrinks)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(), slot)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


# ---- conv / GEMM ------------------------------------------------------------------------------
def pack_conv_weight(w: torch.Tensor, for_dgrad: bool = False, rows_pad: int = 0, cols_pad: int = 0, kmajor: bool = False) -> torch.Tensor:
    """f32 [cout, cin, ks, ks] (or [out, in] for Linear) -> bf16 [rows, ks*ks, cols] kernel operand (parity mode: [rows, ks*ks, 6*cols]).
    kmajor: the same launch also writes the K-tile-major copy [cols/32, ks*ks, rows, 32] (include/dmvae_hip.h: dmvae_conv_desc.w_layout = 1) and hangs it on
    the result as ``_dmvae_kmajor``; conv2d_nhwc / conv2d_nhwc_gnstats hand that copy to the calls the library runs on its kx-halo kernel."""
    if parity.on():
        return parity.pack_conv_weight(w, for_dgrad, rows_pad, cols_pad)
    w = _req(w, f32, "weight")
    if w.dim() == 2:
        cout, cin, ks = w.shape[0], w.shape[1], 1
    else:
        cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
    rows, cols = (cin, cout) if for_dgrad else (cout, cin)
    rows_pad, cols_pad = max(rows_pad, rows), max(cols_pad, cols)
    out = torch.empty(rows_pad, ks * ks, cols_pad, dtype=bf16, device=w.device)
    out2 = torch.empty(cols_pad // 32, ks * ks, rows_pad, 32, dtype=bf16, device=w.device) if (kmajor and cols_pad % 32 == 0) else None
    check(_lib.lib().dmvae_pack_conv_weight_v2(w.data_ptr(), out.data_ptr(), _ptr(out2), cout, cin, ks, rows_pad, cols_pad, int(for_dgrad), _stream()),
          "pack_conv_weight")
    if out2 is not None:
        out._dmvae_kmajor = out2
    return out


def _weight_operand(w_packed: torch.Tensor, d) -> int:
    """Address of the weight operand for descriptor d: the K-tile-major copy (and d.w_layout = 1) where the packed tensor carries one and the library says
    the call runs on the large-shape kernel (dmvae_conv_kmajor_applies), else the tap-major tensor itself."""
    wk = getattr(w_packed, "_dmvae_kmajor", None)
    if wk is not None and _lib.lib().dmvae_conv_kmajor_applies(ctypes.byref(d)):
        d.w_layout = 1
        return wk.data_ptr()
    return w_packed.data_ptr()


def subpixel_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d weight f32 [cout, cin, 3, 3] of Upsample's conv (flux_ae.py:101-107) -> WD f32 [cin, cout, 4, 4], the weight of the 4x4 stride-2
    conv D (cout -> cin) with conv3x3(nearest-x2(x), W) == conv_transpose2d(x, WD, stride 2, padding 1)  (include/dmvae_hip.h: dmvae_subpixel_weight)."""
    w = _req(w, f32, "weight")
    cout, cin = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3), w.shape
    wd = torch.empty(cin, cout, 4, 4, dtype=f32, device=w.device)
    check(_lib.lib().dmvae_subpixel_weight(w.data_ptr(), wd.data_ptr(), cout, cin, _stream()), "subpixel_weight")
    return wd


def subpixel_weight_fold(dwd: torch.Tensor, dw_out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dL/dWD f32 [cin, cout, 4, 4] -> dL/dW f32 [cout, cin, 3, 3] (the transpose of subpixel_weight's linear map)."""
    dwd = _req(dwd, f32, "dwd")
    cin, cout = dwd.shape[0], dwd.shape[1]
    dw = dw_out if dw_out is not None else torch.empty(cout, cin, 3, 3, dtype=f32, device=dwd.device)
    assert dw.is_contiguous() and dw.numel() == cout * cin * 9
    check(_lib.lib().dmvae_subpixel_weight_fold(dwd.data_ptr(), dw.data_ptr(), cout, cin, int(accumulate), _stream()), "subpixel_weight_fold")
    return dw


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """sum over all leading dimensions of x [..., C] -> f32 [C]: a conv's bias gradient on its own."""
    c = x.shape[-1]
    if x.dtype == f32 and parity.on():
        s = parity.colsum(x.reshape(-1, c))
        if out is None:
            return s
        return out.add_(s) if accumulate else out.copy_(s)
    x = _req(x, bf16, "x")
    out = out if out is not None else torch.empty(c, dtype=f32, device=x.device)
    ws = workspace(512 * c * 4, x.device, slot="colsum")
    check(_lib.lib().dmvae_colsum_bf16(x.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), x.numel() // c, c, int(accumulate), _stream()), "colsum_bf16")
    return out


def conv_out_size(h: int, w: int, ks: int, upsample: int = 0, stride: int = 1, transposed: bool = False):
    """Output height / width of dmvae_conv2d_nhwc_fwd for a descriptor (csrc/conv_fwd.hip::dmvae_conv_geometry)."""
    if ks == 4:
        return ((h - 1) * stride + 2, (w - 1) * stride + 2) if transposed else ((h - 2) // stride + 1, (w - 2) // stride + 1)
    if upsample or transposed:
        return 2 * h, 2 * w
    return (h // 2, w // 2) if stride == 2 else (h, w)


def _conv_label(n, ho, wo, cin, cout, ks, upsample, out_f32, stride, transposed, stats):
    """(kernel the call dispatches to -- csrc/conv_pp.hip::dmvae_conv_pp_try -- as rocprofv3 prints it, multiply-add FLOPs of the call), so that bench.py's
    per-kernel average can be checked against rocprofv3's per-kernel-name average."""
    sub = bool(transposed) and ks == 4 and stride == 2    # per-parity 2x2 decomposition: 4 of the 16 taps per output pixel are multiply-adds
    if cin % 32 == 0 and cout >= 64 and cout % 8 == 0 and n * ho * wo >= 16384:
        ups1 = int(upsample) == 1                  # nearest x2 folded into the gather: its own template variant
        gen = (int(upsample) == 2 or stride == 2 or ks == 4 or bool(transposed)) and not sub   # the general-gather instantiation
        t = lambda f: "true" if f else "false"
        dyn = os.environ.get("DMVAE_PP_DYNAMIC", "0") not in ("", "0") and not ups1 and not out_f32 and (n * ho * wo // (512 if cout <= 128 else 256)) * ((cout + 255) // 256 if cout > 128 else 1) > 256
        # the kx-halo form (conv_pp.hip, HALO): plain 3x3 with a bf16 result in the chunk-outer K order
        halo = ks == 3 and not gen and not sub and not ups1 and not out_f32
        label = "conv_pp_kernel<%s, %s, %s, %s, %s, %s, %s, %s, %s>" % (("64, 1024, 1, 8, 4" if (halo and cout <= 64 and not stats) else ("128, 512" if cout <= 128 else "256, 256") + ", 2, 4, 4"), t(ups1), t(out_f32),
                                                                       "false" if ups1 else "true", t(gen), t(sub), t(dyn), t(stats), t(halo))
    else:
        label = "conv_fwd_kernel"
    return label, 2.0 * n * ho * wo * cout * cin * (4 if sub else ks * ks)


def conv2d_nhwc(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, ks: int = 3, upsample=False, act: int = ACT_NONE,
                out_f32: bool = False, stride: int = 1, transposed: bool = False, flop_channels: Optional[Tuple[int, int]] = None) -> torch.Tensor:
    """y = act(conv(x, w) + bias + residual); x [N,H,W,Cin] bf16, w_packed [Cout, ks*ks, Cin] bf16.
    upsample: False/0 none, True/1 nearest x2 folded into the gather, 2 zero-insertion x2 (dgrad of the stride-2 3x3 conv);
    stride 2 with ks 3: the Downsample conv (input padded bottom/right by one, flux_ae.py:85-95); ks 4 (stride 1 | 2, padding 1): the
    PatchGAN convs (patchgan.py:125-147); transposed: the input gradient of the ks-4 conv with that stride (w_packed packed for_dgrad).
    flop_channels: the layer's true (cin, cout) when the operands are zero-padded -- only used by the bench's FLOP accounting."""
    if x.dtype == f32 and parity.on():
        # f32 activations: the same kernel over the six exact bf16 partial products laid out along the channel (reduction) axis; bias in the
        # kernel's f32 epilogue, residual / activation afterwards in f32
        y = conv2d_nhwc(parity.split_channels(x, parity.A_SIDE), w_packed, bias, None, ks, upsample, ACT_NONE, True, stride, transposed)
        return parity.epilogue(y, residual, act)
    x = _req(x, bf16, "x")
    w_packed = _req(w_packed, bf16, "w_packed")
    n, h, w_, cin = x.shape
    cout = w_packed.shape[0]
    assert w_packed.shape[1] == ks * ks and w_packed.shape[2] == cin, (w_packed.shape, ks, cin)
    ho, wo = conv_out_size(h, w_, ks, int(upsample), stride, transposed)
    y = torch.empty(n, ho, wo, cout, dtype=f32 if out_f32 else bf16, device=x.device)
    if bias is not None:
        _req(bias, f32, "bias")
    if residual is not None:
        _req(residual, bf16, "residual")
        assert residual.shape == y.shape
    d = ConvDesc(n, h, w_, cin, cout, ks, int(upsample), act, int(out_f32), stride, int(transposed))
    timing = KERNEL_TIMING
    if timing is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().dmvae_conv2d_nhwc_fwd(x.data_ptr(), _weight_operand(w_packed, d), _ptr(bias), _ptr(residual), y.data_ptr(), ctypes.byref(d),
                                           _stream()), "conv2d_nhwc_fwd")
    if timing is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        # which kernel dmvae_conv2d_nhwc_fwd dispatched to (csrc/conv_pp.hip::dmvae_conv_pp_try), so that the bench's per-kernel
        # average can be checked against rocprofv3's per-kernel-name average
        label, fl = _conv_label(n, ho, wo, cin, cout, ks, upsample, out_f32, stride, transposed, False)
        if flop_channels is not None:       # zero-padded operands (a 3-channel image as 32, 3 output channels as 4): count the model's multiply-adds, not the padding's
            fl *= (flop_channels[0] * flop_channels[1]) / float(cin * cout)
        timing.append((label, e0, e1, fl))
    return y


def conv2d_nhwc_gnstats(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, ks: int = 3,
                        act: int = ACT_NONE, stride: int = 1, transposed: bool = False, groups: int = 32, eps: float = 1e-6):
    """(y, stats): conv2d_nhwc plus the GroupNorm statistics [N, groups, 2] = (mean, rstd) of its bf16 result, which the large-shape conv kernel sums in its
    epilogue (no separate pass over y); other shapes / the f32 parity mode: the conv followed by groupnorm_stats."""
    if x.dtype == f32 and parity.on():
        y = conv2d_nhwc(x, w_packed, bias, residual, ks=ks, act=act, stride=stride, transposed=transposed)
        return y, groupnorm_stats(y, groups, eps)
    x = _req(x, bf16, "x")
    w_packed = _req(w_packed, bf16, "w_packed")
    n, h, w_, cin = x.shape
    cout = w_packed.shape[0]
    assert w_packed.shape[1] == ks * ks and w_packed.shape[2] == cin, (w_packed.shape, ks, cin)
    ho, wo = conv_out_size(h, w_, ks, 0, stride, transposed)
    y = torch.empty(n, ho, wo, cout, dtype=bf16, device=x.device)
    if bias is not None:
        _req(bias, f32, "bias")
    if residual is not None:
        _req(residual, bf16, "residual")
        assert residual.shape == y.shape
    d = ConvDesc(n, h, w_, cin, cout, ks, 0, act, 0, stride, int(transposed))
    L = _lib.lib()
    wsb = L.dmvae_conv2d_nhwc_fwd_gnstats_workspace(ctypes.byref(d), groups)
    if wsb == 0:
        raise ValueError(f"conv2d_nhwc_gnstats: unsupported shape {tuple(y.shape)} with {groups} groups")
    ws = workspace(wsb, x.device, slot="gnstats")
    stats = torch.empty(n, groups, 2, dtype=f32, device=x.device)
    timing = KERNEL_TIMING
    if timing is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.dmvae_conv2d_nhwc_fwd_gnstats(x.data_ptr(), _weight_operand(w_packed, d), _ptr(bias), _ptr(residual), y.data_ptr(), stats.data_ptr(), ws.data_ptr(),
                                          ws.numel(), groups, float(eps), ctypes.byref(d), _stream()), "conv2d_nhwc_fwd_gnstats")
    if timing is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        sub = bool(transposed) and ks == 4 and stride == 2
        plain = ks in (1, 3) and stride == 1 and not transposed
        tp = 512 if cout <= 128 else 256
        fused = (plain or sub) and cout % groups == 0 and (cout // groups) % 4 == 0 and ((h * w_) if sub else (ho * wo)) % tp == 0
        label, fl = _conv_label(n, ho, wo, cin, cout, ks, 0, False, stride, transposed, fused)
        timing.append((label + (" + gn stats" if label == "conv_fwd_kernel" or not fused else ""), e0, e1, fl))
    return y, stats


def conv2d_nhwc_wgrad(dy: torch.Tensor, a: torch.Tensor, ks: int, upsample: bool = False, need_bias: bool = True,
                      dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None,
                      accumulate: bool = False, stride: int = 1) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """dW [Cout,Cin,ks,ks] f32 and db [Cout] f32 from dy [N,Ho,Wo,Cout] and the conv input a [N,H,W,Cin] (bf16)."""
    if dy.dtype == f32 and parity.on():
        # the reduction runs over images x pixels: the six partial products become six times the images
        dw, _ = conv2d_nhwc_wgrad(parity.split_batch(dy, parity.A_SIDE), parity.split_batch(a, parity.W_SIDE), ks, upsample, False, dw_out, None,
                                  accumulate, stride)
        db = None
        if need_bias:
            db = parity.colsum(dy.reshape(-1, dy.shape[-1]))
            if db_out is not None:
                db = db_out.add_(db) if accumulate else db_out.copy_(db)
        return dw, db
    dy = _req(dy, bf16, "dy")
    a = _req(a, bf16, "a")
    n, h, w_, cin = a.shape
    cout = dy.shape[-1]
    d = ConvDesc(n, h, w_, cin, cout, ks, int(upsample), 0, 0, stride, 0)
    L = _lib.lib()
    wsb = L.dmvae_conv2d_nhwc_wgrad_workspace(ctypes.byref(d))
    ws = workspace(wsb, a.device)
    dw = dw_out if dw_out is not None else torch.empty(cout, cin, ks, ks, dtype=f32, device=a.device)
    db = (db_out if db_out is not None else torch.empty(cout, dtype=f32, device=a.device)) if need_bias else None
    timing = KERNEL_TIMING
    if timing is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.dmvae_conv2d_nhwc_wgrad(dy.data_ptr(), a.data_ptr(), dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel(), ctypes.byref(d),
                                    int(accumulate), _stream()), "conv2d_nhwc_wgrad")
    if timing is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        # the whole weight-gradient call: split-K main kernel + fixed-order slab reduce (+ fused bias gradient); csrc/conv_wgrad_pp.hip takes
        # stride-1 3x3 / 1x1 and stride-2 4x4 shapes whose rows are multiples of 32 pixels with >= 4096 reduction rows, csrc/conv_wgrad.hip the rest
        ho, wo = dy.shape[1], dy.shape[2]
        big = ((stride == 1 and ks in (1, 3)) or (stride == 2 and ks == 4 and h % 2 == 0 and w_ % 2 == 0)) and wo % 32 == 0 and cin % 128 == 0 \
            and cout % 128 == 0 and n * ho * wo >= 4096
        timing.append(("wgrad_pp" if big else "wgrad_small", e0, e1, 2.0 * n * ho * wo * cout * cin * ks * ks))
    return dw, db


def conv_out_wgrad_supported(n: int, h: int, w: int, cin: int, cout: int) -> bool:
    return _lib.lib().dmvae_conv_out_wgrad_workspace(n, h, w, cin, cout) > 0


def conv_out_wgrad(dy_nchw: torch.Tensor, a: torch.Tensor, dw_out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """dW [cout,128,3,3] f32 of a 3x3 stride-1 conv with <= 4 output channels from its NCHW f32 output gradient [N,cout,H,W] and its NHWC bf16 input
    a [N,H,W,128] (the decoder's conv_out, flux_ae.py:237,274; csrc/wgrad_thin.hip: `a` is read once)."""
    dy_nchw = _req(dy_nchw, f32, "dy")
    a = _req(a, bf16, "a")
    n, h, w_, cin = a.shape
    cout = dy_nchw.shape[1]
    assert dy_nchw.shape == (n, cout, h, w_), (dy_nchw.shape, a.shape)
    L = _lib.lib()
    wsb = L.dmvae_conv_out_wgrad_workspace(n, h, w_, cin, cout)
    if wsb == 0:
        raise ValueError(f"conv_out_wgrad: unsupported shape a={tuple(a.shape)} cout={cout}")
    ws = workspace(wsb, a.device)
    dw = dw_out if dw_out is not None else torch.empty(cout, cin, 3, 3, dtype=f32, device=a.device)
    check(L.dmvae_conv_out_wgrad(dy_nchw.data_ptr(), a.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), n, h, w_, cin, cout, int(accumulate), _stream()),
          "conv_out_wgrad")
    return dw


GEMM_NT_LARGE_TILES = True      # gemm_nt: batched products with K >= 384 on csrc/gemm_pp.hip's BATCHED instantiations (tests compare the two kernels)


def gemm_nt(a: torch.Tensor, b: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
            act: int = ACT_NONE, out_f32: bool = False) -> torch.Tensor:
    """C[..., m, n] = act(A[..., m, k] @ B[..., n, k]^T + bias + residual).  A/B bf16; a 2-D operand is shared by the batch."""
    if a.dtype == f32 and parity.on():
        bs = b if b.dtype == bf16 else parity.split_channels(b, parity.W_SIDE)          # bf16: a weight operand packed in parity mode (already split)
        return parity.epilogue(gemm_nt(parity.split_channels(a, parity.A_SIDE), bs, bias, None, ACT_NONE, True), residual, act)
    a = _req(a, bf16, "A")
    b = _req(b, bf16, "B")
    batch = a.shape[0] if a.dim() == 3 else (b.shape[0] if b.dim() == 3 else 1)
    m, k = a.shape[-2:]
    n = b.shape[-2]
    assert b.shape[-1] == k
    shape = (batch, m, n) if (a.dim() == 3 or b.dim() == 3) else (m, n)
    c = torch.empty(shape, dtype=f32 if out_f32 else bf16, device=a.device)
    if residual is not None:
        _req(residual, bf16, "residual")
    L = _lib.lib()
    if GEMM_NT_LARGE_TILES and a.dim() == 3 and b.dim() == 3 and bias is None and residual is None and act == ACT_NONE and m >= 256 and n >= 128 \
            and L.dmvae_linear_bf16_batched_supported(batch, m, n, k) and batch * m * max(n, k) * (4 if out_f32 else 2) < (1 << 31) and batch * n * k * 2 < (1 << 31):
        # per-sample products with a deep reduction (the decoder attention's q k^T, p v and their input gradients: 1024 x 1024 x 512 per sample): the large-tile
        # Linear GEMM with a batch index in its tile decode (csrc/gemm_pp.hip, BATCHED) instead of the 128 x 128-tile kernel (~ 2.3 x its rate)
        check(L.dmvae_linear_bf16_batched(a.data_ptr(), b.data_ptr(), c.data_ptr(), batch, m, n, k, k, k, n, m * k, n * k, m * n, int(out_f32), _stream()),
              "linear_bf16_batched")
        return c
    check(_lib.lib().dmvae_gemm_nt_batched(a.data_ptr(), b.data_ptr(), _ptr(bias), _ptr(residual), c.data_ptr(), m, n, k, batch,
                                           m * k if a.dim() == 3 else 0, n * k if b.dim() == 3 else 0, m * n, act, int(out_f32), _stream()),
          "gemm_nt_batched")
    return c


def linear_plan(m: int, n: int, k: int) -> Tuple[int, int, int]:
    """(menu index, tile columns, tile rows) dmvae_linear_bf16 uses for an [m, k] x [n, k]^T problem (csrc/gemm_pp.hip::plan)."""
    tc, tr = ctypes.c_int(0), ctypes.c_int(0)
    idx = _lib.lib().dmvae_linear_bf16_plan(m, n, k, ctypes.byref(tc), ctypes.byref(tr))
    return idx, tc.value, tr.value


def linear_supported(m: int, n: int, k: int) -> bool:
    """Shapes dmvae_linear_bf16 takes: K a multiple of 32 and at least 384, N a multiple of 8, every operand below 2 GiB.  Problems with fewer than 64 rows
    (adaLN / embedder Linears on one row per sample) are left to the small batched NT kernel (gemm_nt): a 128-256-row tile would be nearly all padding."""
    return k >= 384 and k % 32 == 0 and n % 8 == 0 and m >= 64 and m * max(n, k) * 2 < (1 << 31) and n * k * 2 < (1 << 31)


def linear_bf16(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, out_f32: bool = False) -> torch.Tensor:
    """F.linear(x, w, bias) under autocast(bf16) on the hand-written GEMM (csrc/gemm_pp.hip): x [..., K] bf16; w bf16, either [N, K] (the parameter's bf16 copy;
    a [K_in, N_out]-transposed copy makes the same call the input gradient) or its K-tile-major copy [K / 32, N, 32] (`pack_conv_weight(..., kmajor=True)`
    leaves it as `._dmvae_kmajor`; whole 128-B lines per K tile -- what frozen weights are served as); bias [N] bf16 (autocast's operand) or f32; f32
    accumulation, bf16 result [..., N] (f32 with out_f32).  act = ACT_GELU / ACT_SILU fuses the activation on the bf16-rounded pre-activation (bit-identical
    to the two-kernel route); act = ACT_SWIGLU writes [..., N / 2]: silu(x1) * x2 of the [x1 | x2] halves of the bf16-rounded columns, the bits of `swiglu(linear(.))`."""
    x = _req(x, bf16, "x")
    w = _req(w, bf16, "w")
    k = x.shape[-1]
    kmajor = w.dim() == 3
    if kmajor:
        assert w.shape[2] == 32 and w.shape[0] * 32 == k, (x.shape, w.shape)
        n = w.shape[1]
    else:
        assert w.dim() == 2 and w.shape[1] == k, (x.shape, w.shape)
        n = w.shape[0]
    m = x.numel() // k
    bias_bf16 = 0
    if bias is not None:
        if bias.dtype == bf16:
            bias_bf16 = 1
        _req(bias, bf16 if bias_bf16 else f32, "bias")
        assert bias.numel() == n
    ny = n // 2 if act == ACT_SWIGLU else n
    y = torch.empty(*x.shape[:-1], ny, dtype=f32 if out_f32 else bf16, device=x.device)
    timing = KERNEL_TIMING
    if timing is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().dmvae_linear_bf16(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), m, n, k, k, k, ny, act, bias_bf16, int(out_f32), int(kmajor),
                                       _stream()), "linear_bf16")
    if timing is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        timing.append(("gemm_pp_kernel", e0, e1, 2.0 * m * n * k))
    return y


def linear_swiglu_pre(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """(g [..., H], x12 [..., 2H]) = (silu(x1) * x2, bf16(x @ w^T + bias)) in ONE launch of the Linear GEMM (include/dmvae_hip.h dmvae_linear_bf16_swiglu_pre): the bits
    of `linear_bf16(x, w, bias)` followed by `swiglu(.)`; x12 is kept for SwiGLU's backward.  w bf16 [2H, K] row-major or K-tile-major [K / 32, 2H, 32]."""
    x = _req(x, bf16, "x")
    w = _req(w, bf16, "w")
    k = x.shape[-1]
    kmajor = w.dim() == 3
    n = w.shape[1] if kmajor else w.shape[0]
    assert (w.shape[0] * 32 == k and w.shape[2] == 32) if kmajor else (w.shape[1] == k), (x.shape, w.shape)
    m = x.numel() // k
    bias_bf16 = 0
    if bias is not None:
        bias_bf16 = int(bias.dtype == bf16)
        _req(bias, bf16 if bias_bf16 else f32, "bias")
    g = torch.empty(*x.shape[:-1], n // 2, dtype=bf16, device=x.device)
    x12 = torch.empty(*x.shape[:-1], n, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_linear_bf16_swiglu_pre(x.data_ptr(), w.data_ptr(), _ptr(bias), g.data_ptr(), x12.data_ptr(), m, n, k, k, k, n // 2, n, bias_bf16, int(kmajor),
                                                  _stream()), "linear_bf16_swiglu_pre")
    return g, x12


_SK_WS = {}


def _sk_workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch of `linear_sk`, per (device, stream): ZERO-initialised when (re)allocated -- its first 64 KB are the tiles' arrival counters, which the kernel expects
    zero and leaves zero; the partial-tile slots behind them are written before they are read."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream())
    ws = _SK_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        _SK_WS[key] = ws
    return ws


def linear_sk_supported(m: int, n: int, k: int, splits: int = 0, tile: int = 0) -> bool:
    return bool(_lib.lib().dmvae_linear_bf16_sk_supported(int(m), int(n), int(k), int(splits), int(tile)))


def linear_sk(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0, splits: int = 0, tile: int = 0) -> torch.Tensor:
    """y bf16 [M, N] = act(x @ w^T + bias) on the stream-K (splits = 0) / fused split-K (splits >= 2 uniform parts per tile) instantiation of the Linear GEMM
    (include/dmvae_hip.h dmvae_linear_bf16_sk): ONE launch, the parts of a tile summed in K order by the last one to arrive.  w bf16 [N, K] row-major or
    K-tile-major [K / 32, N, 32].  tile 0: 256 x 256 output tiles, 1: 256 columns x 128 rows."""
    x = _req2d(x, "x")
    m, k = x.shape
    if w.dim() == 3:
        w = _req(w, bf16, "w")
        assert w.shape[0] * 32 == k and w.shape[2] == 32, f"K-tile-major weight {tuple(w.shape)} does not match K = {k}"
        n, layout, ldw = w.shape[1], 1, k
    else:
        w = _req2d(w, "w")
        n, layout, ldw = w.shape[0], 0, w.stride(0)
        assert w.shape[1] == k
    L = _lib.lib()
    ws = _sk_workspace(L.dmvae_linear_bf16_sk_workspace(m, n, k, int(splits), int(tile)), x.device)
    nout = n // 2 if act == ACT_SWIGLU else n
    y = torch.empty(m, nout, dtype=bf16, device=x.device)
    check(L.dmvae_linear_bf16_sk(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), ws.data_ptr(), ws.numel(), int(splits), int(tile), m, n, k, x.stride(0), ldw, nout,
                                 int(act), int(bias is not None and bias.dtype == bf16), layout, _stream()), "linear_bf16_sk")
    return y


def linear_splitk_supported(m: int, n: int, k: int, splits: int) -> bool:
    return bool(_lib.lib().dmvae_linear_bf16_splitk_supported(int(m), int(n), int(k), int(splits)))


def linear_splitk(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, splits: int = 3) -> torch.Tensor:
    """y [M, N] bf16 = x [M, K] @ w^T + bias with the reduction cut into `splits` parts computed as independent work units (f32 slabs) and summed in order
    (include/dmvae_hip.h dmvae_linear_bf16_splitk / dmvae_splitk_sum_bf16): for few-tile, deep-K problems.  w bf16 [N, K] row-major or K-tile-major [K / 32, N, 32]."""
    x = _req2d(x, "x")
    m, k = x.shape
    if w.dim() == 3:
        assert w.dtype == bf16 and w.is_contiguous() and w.shape[0] * 32 == k and w.shape[2] == 32
        n, layout, ldw = w.shape[1], 1, k
    else:
        w = _req2d(w, "w")
        n, layout, ldw = w.shape[0], 0, w.stride(0)
        assert w.shape[1] == k
    L = _lib.lib()
    slabs = workspace(splits * m * n * 4, x.device, "splitk_slabs")
    y = torch.empty(m, n, dtype=bf16, device=x.device)
    check(L.dmvae_linear_bf16_splitk(x.data_ptr(), w.data_ptr(), slabs.data_ptr(), int(splits), m, n, k, x.stride(0), ldw, layout, _stream()), "linear_bf16_splitk")
    if bias is not None:
        assert bias.is_contiguous() and bias.numel() == n and bias.dtype in (bf16, f32)
    check(L.dmvae_splitk_sum_bf16(slabs.data_ptr(), int(splits), _ptr(bias), int(bias is not None and bias.dtype == bf16), y.data_ptr(), m, n, _stream()), "splitk_sum_bf16")
    return y


def linear_rows_supported(m: int, n: int, k: int) -> bool:
    """Shapes csrc/linear_rows.hip takes: 1 <= m <= 64 rows (one per sample), k % 32 == 0, n % 4 == 0."""
    return bool(_lib.lib().dmvae_linear_rows_supported(int(m), int(n), int(k)))


def linear_rows(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, out_f32: bool = False) -> torch.Tensor:
    """y [M, N] = act(x [M, K] @ w [N, K]^T + bias) for M <= 64 rows -- the per-sample conditioning Linears of LightningDiT (adaLN modulations, timestep
    embedder; diffusion/lightningdit/lightningdit.py:96-139,236-240,266-268) and, with w the transposed copy, their input gradients (include/dmvae_hip.h:
    dmvae_linear_rows_bf16).  x bf16 row-major (a row stride larger than K is fine); w bf16 [N, K] row-major, or its K-tile-major copy [K / 32, N, 32]
    (linear_weight_t_kmajor's result); bias f32 or bf16; act ACT_NONE | ACT_SILU."""
    x = _req2d(x, "x")
    m, k = x.shape
    if w.dim() == 3:                                     # K-tile-major
        if not (w.is_cuda and w.dtype == bf16 and w.is_contiguous() and w.shape[0] * 32 == k and w.shape[2] == 32):
            raise TypeError(f"w: expected the K-tile-major copy [K / 32, N, 32] for K = {k}, got {tuple(w.shape)}")
        n, layout, ldw = w.shape[1], 1, 0
    else:
        w = _req2d(w, "w")
        n, layout, ldw = w.shape[0], 0, w.stride(0)
        assert w.shape[1] == k, (x.shape, w.shape)
    if bias is not None:
        assert bias.is_contiguous() and bias.numel() == n and bias.dtype in (bf16, f32)
    y = torch.empty(m, n, dtype=f32 if out_f32 else bf16, device=x.device)
    check(_lib.lib().dmvae_linear_rows_bf16(x.data_ptr(), w.data_ptr(), _ptr(bias), y.data_ptr(), m, n, k, x.stride(0), ldw, n, int(act),
                                            int(bias is not None and bias.dtype == bf16), int(out_f32), layout, _stream()), "linear_rows_bf16")
    return y


def linear_rows_wgrad(dy: torch.Tensor, x: torch.Tensor, need_bias: bool = True, dw_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None,
                      accumulate: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """(dW [N, K] f32, db [N] f32) of a per-sample Linear from dy [M, N] and x [M, K] (bf16, M <= 64): include/dmvae_hip.h dmvae_linear_rows_wgrad."""
    dy, x = _req2d(dy, "dy"), _req2d(x, "x")
    m, n = dy.shape
    k = x.shape[1]
    assert x.shape[0] == m
    dw = dw_out if dw_out is not None else torch.empty(n, k, dtype=f32, device=x.device)
    assert dw.is_contiguous() and dw.numel() == n * k and dw.dtype == f32
    db = (db_out if db_out is not None else torch.empty(n, dtype=f32, device=x.device)) if need_bias else None
    check(_lib.lib().dmvae_linear_rows_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _ptr(db), m, n, k, dy.stride(0), x.stride(0), int(accumulate), _stream()),
          "linear_rows_wgrad")
    return dw, db


def _req2d(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.DmvaeHipError(f"{name}: expected a GPU tensor; dmvae_amd has no CPU path")
    if t.dtype != bf16 or t.dim() != 2 or t.stride(1) != 1:
        raise TypeError(f"{name}: expected a 2-D bf16 tensor with unit column stride, got {t.dtype} {tuple(t.shape)} strides {t.stride()}")
    return t


def linear_weight_t_kmajor(w: torch.Tensor) -> torch.Tensor:
    """bf16 Linear weight [N, K] -> [N / 32, K, 32]: the K-tile-major operand of its transpose -- with it `linear_bf16(dy, .)` is the input gradient dY . W."""
    w = _req(w, bf16, "w")
    n, k = w.shape
    out = torch.empty(n // 32, k, 32, dtype=bf16, device=w.device)
    check(_lib.lib().dmvae_linear_weight_t_kmajor(w.data_ptr(), out.data_ptr(), n, k, _stream()), "linear_weight_t_kmajor")
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, alpha: float = 1.0, out_f32: bool = False) -> torch.Tensor:
    """C[b] = alpha * A[b]^T @ B[b] with A [batch, K, M], B [batch, K, N] bf16 (reduction over the slow dim)."""
    if a.dtype == f32 and parity.on():
        return gemm_tn(parity.split_rows(a, parity.A_SIDE), parity.split_rows(b, parity.W_SIDE), alpha, True)
    a = _req(a, bf16, "A")
    b = _req(b, bf16, "B")
    assert a.dim() == 3 and b.dim() == 3 and a.shape[:2] == b.shape[:2]
    batch, k, m = a.shape
    n = b.shape[2]
    L = _lib.lib()
    wsb = L.dmvae_gemm_tn_batched_workspace(m, n, k, batch)
    ws = workspace(wsb, a.device)
    c = torch.empty(batch, m, n, dtype=f32 if out_f32 else bf16, device=a.device)
    check(L.dmvae_gemm_tn_batched(a.data_ptr(), b.data_ptr(), c.data_ptr(), ws.data_ptr(), ws.numel(), m, n, k, batch, k * m, k * n, m * n,
                                  float(alpha), int(out_f32), _stream()), "gemm_tn_batched")
    return c


def softmax_rows(s: torch.Tensor, scale: float) -> torch.Tensor:
    if parity.on():
        return parity.softmax_rows(s, scale)
    s = _req(s, f32, "S")
    p = torch.empty(s.shape, dtype=bf16, device=s.device)
    check(_lib.lib().dmvae_softmax_rows_fwd(s.data_ptr(), p.data_ptr(), s.numel() // s.shape[-1], s.shape[-1], float(scale), _stream()), "softmax_rows_fwd")
    return p


def softmax_rows_bwd(dp: torch.Tensor, p: torch.Tensor, scale: float) -> torch.Tensor:
    if p.dtype == f32 and parity.on():
        return parity.softmax_rows_bwd(dp, p, scale)
    dp = _req(dp, f32, "dP")
    p = _req(p, bf16, "P")
    ds = torch.empty_like(p)
    check(_lib.lib().dmvae_softmax_rows_bwd(dp.data_ptr(), p.data_ptr(), ds.data_ptr(), p.numel() // p.shape[-1], p.shape[-1], float(scale), _stream()),
          "softmax_rows_bwd")
    return ds


def transpose_last2(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == f32 and parity.on():
        return parity.transpose_last2(x)
    x = _req(x, bf16, "x")
    assert x.dim() == 3
    out = torch.empty(x.shape[0], x.shape[2], x.shape[1], dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_transpose_bf16(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1], x.shape[2], _stream()), "transpose_bf16")
    return out


# ---- GroupNorm ----------------------------------------------------------------------------------
def groupnorm_stats(x: torch.Tensor, groups: int = 32, eps: float = 1e-6) -> torch.Tensor:
    if x.dtype == f32 and parity.on():
        return parity.groupnorm_stats(x, groups, eps)
    x = _req(x, bf16, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    L = _lib.lib()
    wsb = L.dmvae_groupnorm_workspace(n, hw, c, groups)
    if wsb == 0:
        raise ValueError(f"groupnorm: unsupported shape {tuple(x.shape)} with {groups} groups")
    ws = workspace(wsb, x.device)
    stats = torch.empty(n, groups, 2, dtype=f32, device=x.device)
    check(L.dmvae_groupnorm_stats(x.data_ptr(), stats.data_ptr(), ws.data_ptr(), ws.numel(), n, hw, c, groups, float(eps), _stream()), "groupnorm_stats")
    return stats


def groupnorm_apply(x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, swish: bool, groups: int = 32) -> torch.Tensor:
    if x.dtype == f32 and parity.on():
        return parity.groupnorm_apply(x, stats, gamma, beta, int(swish), groups)
    x = _req(x, bf16, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    y = torch.empty_like(x)
    check(_lib.lib().dmvae_groupnorm_apply(x.data_ptr(), _req(stats, f32, "stats").data_ptr(), _req(gamma, f32, "gamma").data_ptr(),
                                           _req(beta, f32, "beta").data_ptr(), y.data_ptr(), n, hw, c, groups, int(swish), _stream()), "groupnorm_apply")
    return y


def groupnorm_bwd(da: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, swish: bool,
                  dres: Optional[torch.Tensor] = None, groups: int = 32, need_param_grads: bool = True,
                  dg_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None, want_colsum: bool = False):
    """-> (dx, dgamma, dbeta).  want_colsum: dx additionally carries `_dmvae_colsum` = (sum of dx over (n, hw) per channel [C] f32, version, data_ptr) -- the
    bias gradient of the conv that produced x, a by-product of the pass that writes dx (dmvae_groupnorm_bwd_colsum)."""
    if x.dtype == f32 and parity.on():
        return parity.groupnorm_bwd(da, x, stats, gamma, beta, int(swish), dres, groups, need_param_grads, dg_out, db_out)
    da = _req(da, bf16, "da")
    x = _req(x, bf16, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    L = _lib.lib()
    wsb = L.dmvae_groupnorm_workspace(n, hw, c, groups)
    ws = workspace(wsb, x.device)
    dx = torch.empty_like(x)
    dg = (dg_out if dg_out is not None else torch.empty(c, dtype=f32, device=x.device)) if need_param_grads else None
    db = (db_out if db_out is not None else torch.empty(c, dtype=f32, device=x.device)) if need_param_grads else None
    if dres is not None:
        _req(dres, bf16, "dres")
    if want_colsum:
        cs = torch.empty(c, dtype=f32, device=x.device)
        check(L.dmvae_groupnorm_bwd_colsum(da.data_ptr(), x.data_ptr(), _ptr(dres), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(),
                                           _ptr(dg), _ptr(db), cs.data_ptr(), ws.data_ptr(), ws.numel(), n, hw, c, groups, int(swish), 0, 0, _stream()),
              "groupnorm_bwd_colsum")
        dx._dmvae_colsum = (cs, dx._version, dx.data_ptr())
        return dx, dg, db
    check(L.dmvae_groupnorm_bwd(da.data_ptr(), x.data_ptr(), _ptr(dres), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(),
                                _ptr(dg), _ptr(db), ws.data_ptr(), ws.numel(), n, hw, c, groups, int(swish), 0, _stream()), "groupnorm_bwd")
    return dx, dg, db


def conv_in3_supported(n: int, h: int, w: int, cout: int) -> bool:
    return bool(_lib.lib().dmvae_conv_in3_supported(n, h, w, cout))


def conv_in3(x0: torch.Tensor, x1: Optional[torch.Tensor], w: torch.Tensor, bias: Optional[torch.Tensor], shift: Optional[torch.Tensor] = None,
             scale: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """act(conv3x3((cat(x0, x1) - shift) / scale, w) + bias) -> NHWC bf16 [N0 + N1, H, W, cout] from NCHW f32 images with THREE channels (x1 optional), cout 64 /
    128: the LPIPS trunk's first layer with its ScalingLayer and the concatenation of the two branches folded in (include/dmvae_hip.h: dmvae_conv_in3)."""
    x0 = _req(x0, f32, "x0")
    w = _req(w, f32, "weight")
    n0, c, h, wd = x0.shape
    n1 = 0
    if x1 is not None:
        x1 = _req(x1, f32, "x1")
        assert tuple(x1.shape[1:]) == (c, h, wd), (x0.shape, x1.shape)
        n1 = x1.shape[0]
    cout = w.shape[0]
    assert c == 3 and tuple(w.shape) == (cout, 3, 3, 3), (x0.shape, w.shape)
    n = n0 + n1
    L = _lib.lib()
    if not L.dmvae_conv_in3_supported(n, h, wd, cout):
        raise ValueError(f"conv_in3: unsupported shape {n} x {h} x {wd} -> {cout}")
    if bias is not None:
        bias = _req(bias, f32, "bias")
    if shift is not None:
        shift, scale = _req(shift.reshape(-1), f32, "shift"), _req(scale.reshape(-1), f32, "scale")
        assert shift.numel() == 3 and scale.numel() == 3
    ws = workspace(L.dmvae_conv_in3_workspace(n, h, wd), x0.device, slot="conv_in3")
    y = torch.empty(n, h, wd, cout, dtype=bf16, device=x0.device)
    check(L.dmvae_conv_in3(x0.data_ptr(), _ptr(x1), n0, _ptr(shift), _ptr(scale), w.data_ptr(), _ptr(bias), y.data_ptr(), ws.data_ptr(), ws.numel(), n, h, wd,
                           cout, act, _stream()), "conv_in3")
    return y


def conv_k4c1_supported(n: int, h: int, w: int, c: int) -> bool:
    return bool(_lib.lib().dmvae_conv_k4c1_supported(n, h, w, c))


def conv_k4c1_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """nn.Conv2d(C, 1, 4, 1, 1) on NHWC bf16 x [N, H, 31, C] with the f32 parameter w [1, C, 4, 4] -> f32 logits [N, H - 1, 30, 1] (include/dmvae_hip.h dmvae_conv_k4c1_fwd)."""
    x = _req(x, bf16, "x"); w = _req(w, f32, "weight")
    n, h, wd, c = x.shape
    assert tuple(w.shape) == (1, c, 4, 4), (x.shape, w.shape)
    out = torch.empty(n, h - 1, wd - 1, 1, dtype=f32, device=x.device)
    check(_lib.lib().dmvae_conv_k4c1_fwd(x.data_ptr(), w.data_ptr(), _ptr(None if bias is None else _req(bias, f32, "bias")), out.data_ptr(), n, h, wd, c, _stream()), "conv_k4c1_fwd")
    return out


def conv_k4c1_dgrad(dy: torch.Tensor, w: torch.Tensor, h: int) -> torch.Tensor:
    """Input gradient of `conv_k4c1_fwd`: dy f32 [N, H - 1, 30, 1] -> dx bf16 [N, H, 31, C]."""
    dy = _req(dy, f32, "dy"); w = _req(w, f32, "weight")
    n, c = dy.shape[0], w.shape[1]
    assert dy.shape[1] == h - 1 and dy.shape[2] == 30
    dx = torch.empty(n, h, 31, c, dtype=bf16, device=dy.device)
    check(_lib.lib().dmvae_conv_k4c1_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), n, h, 31, c, _stream()), "conv_k4c1_dgrad")
    return dx


def conv_k4c1_wgrad(x: torch.Tensor, dy: torch.Tensor, dw: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None, need_bias: bool = True):
    """Weight / bias gradient of `conv_k4c1_fwd`: (dw f32 [1, C, 4, 4], db f32 [1] or None), written into dw / db when given (flat-buffer views)."""
    x = _req(x, bf16, "x"); dy = _req(dy, f32, "dy")
    n, h, wd, c = x.shape
    L = _lib.lib()
    if dw is None:
        dw = torch.empty(1, c, 4, 4, dtype=f32, device=x.device)
    if db is None and need_bias:
        db = torch.empty(1, dtype=f32, device=x.device)
    assert dw.dtype == f32 and dw.is_contiguous() and dw.numel() == 16 * c and (db is None or (db.dtype == f32 and db.numel() == 1))
    ws = workspace(L.dmvae_conv_k4c1_wgrad_workspace(c), x.device, slot="conv_k4c1")
    check(L.dmvae_conv_k4c1_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), ws.data_ptr(), ws.numel(), n, h, wd, c, _stream()), "conv_k4c1_wgrad")
    return dw, db


def conv_to_image_supported(n: int, h: int, w: int, cin: int, cout: int) -> bool:
    return bool(_lib.lib().dmvae_conv_to_image_supported(n, h, w, cin, cout))


def conv_to_image(x: torch.Tensor, w_packed: torch.Tensor, cout: int, bias: Optional[torch.Tensor] = None, mul: Optional[torch.Tensor] = None) -> torch.Tensor:
    """conv3x3(x [N, H, W, Cin] bf16, Cin = 64 / 128) -> NCHW f32 image [N, cout, H, W] (cout <= 4), times mul[cout] when given: include/dmvae_hip.h
    dmvae_conv_to_image.  w_packed: bf16 [4, 9, Cin] (pack_conv_weight with rows_pad = 4)."""
    x = _req(x, bf16, "x")
    w_packed = _req(w_packed, bf16, "w_packed")
    n, h, wd, cin = x.shape
    assert tuple(w_packed.shape) == (4, 9, cin), w_packed.shape
    if bias is not None:
        bias = _req(bias, f32, "bias")
    if mul is not None:
        mul = _req(mul, f32, "mul")
        assert mul.numel() == cout
    L = _lib.lib()
    if not L.dmvae_conv_to_image_supported(n, h, wd, cin, cout):
        raise ValueError(f"conv_to_image: unsupported shape {tuple(x.shape)} -> {cout}")
    y = torch.empty(n, cout, h, wd, dtype=f32, device=x.device)
    check(L.dmvae_conv_to_image(x.data_ptr(), w_packed.data_ptr(), _ptr(bias), _ptr(mul), y.data_ptr(), n, h, wd, cin, cout, _stream()), "conv_to_image")
    return y


def norm_conv_out_fwd_supported(n: int, h: int, w: int, c: int, cout: int, groups: int = 32) -> bool:
    return bool(_lib.lib().dmvae_norm_conv_out_fwd_supported(n, h, w, c, groups, cout))


def norm_conv_out_fwd(x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor],
                      cout: int, groups: int = 32):
    """-> (a, y): a = swish(GroupNorm(x)) bf16 [N, H, W, C] and y = conv3x3(a, conv_out) + bias as the NCHW f32 image [N, cout, H, W] in ONE launch
    (flux_ae.py:266-268; include/dmvae_hip.h: dmvae_norm_conv_out_fwd).  w_packed: pack_conv_weight(conv_out.weight, rows_pad=4) -- bf16 [4, 9, C]."""
    x = _req(x, bf16, "x")
    w_packed = _req(w_packed, bf16, "w_packed")
    n, h, wd, c = x.shape
    assert tuple(w_packed.shape) == (4, 9, c), w_packed.shape
    if bias is not None:
        bias = _req(bias, f32, "bias")
        assert bias.numel() == cout
    L = _lib.lib()
    if not L.dmvae_norm_conv_out_fwd_supported(n, h, wd, c, groups, cout):
        raise ValueError(f"norm_conv_out_fwd: unsupported shape x {tuple(x.shape)}, cout {cout}, {groups} groups")
    a = torch.empty_like(x)
    y = torch.empty(n, cout, h, wd, dtype=f32, device=x.device)
    check(L.dmvae_norm_conv_out_fwd(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w_packed.data_ptr(), _ptr(bias), a.data_ptr(), y.data_ptr(),
                                    n, h, wd, c, groups, cout, _stream()), "norm_conv_out_fwd")
    return a, y


def norm_conv_out_bwd_supported(n: int, h: int, w: int, c: int, cout: int, groups: int = 32) -> bool:
    return bool(_lib.lib().dmvae_norm_conv_out_bwd_supported(n, h, w, c, groups, cout))


def norm_conv_out_bwd(dy: torch.Tensor, w: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int = 32,
                      dg_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None):
    """-> (dx, dgamma, dbeta) of conv_out(swish(norm_out(x))) (flux_ae.py:266-268) given the image gradient dy f32 [N, 3, H, W] (NCHW), conv_out.weight w f32
    [3, C, 3, 3], norm_out's input x bf16 [N, H, W, C] and its statistics: the 3 -> C input-gradient conv runs inside both GroupNorm backward passes
    (include/dmvae_hip.h: dmvae_norm_conv_out_bwd) -- the gradient of the activation never exists in memory."""
    dy = _req(dy, f32, "dy")
    w = _req(w, f32, "weight")
    x = _req(x, bf16, "x")
    n, h, wd, c = x.shape
    cout = w.shape[0]
    assert tuple(dy.shape) == (n, cout, h, wd) and tuple(w.shape) == (cout, c, 3, 3), (dy.shape, w.shape, x.shape)
    L = _lib.lib()
    wsb = L.dmvae_norm_conv_out_bwd_workspace(n, h, wd, c, groups)
    if wsb == 0 or not L.dmvae_norm_conv_out_bwd_supported(n, h, wd, c, groups, cout):
        raise ValueError(f"norm_conv_out_bwd: unsupported shape x {tuple(x.shape)}, cout {cout}, {groups} groups")
    ws = workspace(wsb, x.device)
    dx = torch.empty_like(x)
    dg = dg_out if dg_out is not None else torch.empty(c, dtype=f32, device=x.device)
    db = db_out if db_out is not None else torch.empty(c, dtype=f32, device=x.device)
    check(L.dmvae_norm_conv_out_bwd(dy.data_ptr(), w.data_ptr(), x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(), dg.data_ptr(),
                                    db.data_ptr(), ws.data_ptr(), ws.numel(), n, h, wd, c, groups, cout, 0, _stream()), "norm_conv_out_bwd")
    return dx, dg, db


def groupnorm_short_supported(n: int, hw: int, c: int, cs: int, groups: int = 32) -> bool:
    """The shapes whose 1x1 shortcut rides on the GroupNorm passes (csrc/norm_short.hip): c = 256 channels of the block input, cs = 128 of its output, hw % 16 == 0."""
    return bool(_lib.lib().dmvae_groupnorm_short_supported(n, hw, c, cs, groups))


def groupnorm_apply_short(x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor],
                          swish: bool = True, groups: int = 32):
    """-> (a, xs): a = act(GroupNorm(x)) and xs = conv1x1(x, w) + bias from ONE read of x -- a ResnetBlock's norm1 and its nin_shortcut (flux_ae.py:67,71,77-82).
    w: the shortcut's packed forward operand bf16 [cs, 1, c] (functional.packed)."""
    x = _req(x, bf16, "x")
    w = _req(w, bf16, "packed weight")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    cs = w.shape[0]
    assert w.numel() == cs * c, (w.shape, x.shape)
    L = _lib.lib()
    if not L.dmvae_groupnorm_short_supported(n, hw, c, cs, groups):
        raise ValueError(f"groupnorm_apply_short: unsupported shape x {tuple(x.shape)} -> {cs} channels, {groups} groups")
    if bias is not None:
        bias = _req(bias, f32, "bias")
    a = torch.empty_like(x)
    xs = torch.empty(*x.shape[:-1], cs, dtype=bf16, device=x.device)
    check(L.dmvae_groupnorm_apply_short(x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), w.data_ptr(), _ptr(bias), a.data_ptr(), xs.data_ptr(),
                                        n, hw, c, cs, groups, int(swish), _stream()), "groupnorm_apply_short")
    return a, xs


def groupnorm_bwd_short(da: torch.Tensor, x: torch.Tensor, dys: torch.Tensor, wt: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                        swish: bool = True, groups: int = 32, dg_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None,
                        want_colsum: bool = False):
    """groupnorm_bwd(da, x, ..., dres = conv1x1 input gradient of dys) without the stored gradient: dys bf16 [N, H, W, cs] is the ResnetBlock's output gradient,
    wt the shortcut's packed input-gradient operand bf16 [c, 1, cs] (functional.packed(sw, True)).  -> (dx, dgamma, dbeta); want_colsum as in groupnorm_bwd."""
    da = _req(da, bf16, "da")
    x = _req(x, bf16, "x")
    dys = _req(dys, bf16, "dys")
    wt = _req(wt, bf16, "packed weight")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    cs = dys.shape[-1]
    assert wt.numel() == cs * c and wt.shape[0] == c and dys.numel() == n * hw * cs and da.shape == x.shape, (wt.shape, dys.shape, x.shape)
    L = _lib.lib()
    wsb = L.dmvae_groupnorm_bwd_short_workspace(n, hw, c, cs, groups)
    if wsb == 0:
        raise ValueError(f"groupnorm_bwd_short: unsupported shape x {tuple(x.shape)}, dys {tuple(dys.shape)}, {groups} groups")
    ws = workspace(wsb, x.device)
    dx = torch.empty_like(x)
    dg = dg_out if dg_out is not None else torch.empty(c, dtype=f32, device=x.device)
    db = db_out if db_out is not None else torch.empty(c, dtype=f32, device=x.device)
    cs_t = torch.empty(c, dtype=f32, device=x.device) if want_colsum else None
    check(L.dmvae_groupnorm_bwd_short(da.data_ptr(), x.data_ptr(), dys.data_ptr(), wt.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(),
                                      dg.data_ptr(), db.data_ptr(), _ptr(cs_t), ws.data_ptr(), ws.numel(), n, hw, c, cs, groups, int(swish), 0, 0, _stream()),
          "groupnorm_bwd_short")
    if want_colsum:
        dx._dmvae_colsum = (cs_t, dx._version, dx.data_ptr())
    return dx, dg, db


def groupnorm_bwd_reduce(da: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, act: int,
                         groups: int = 32, need_param_grads: bool = True, dg_out: Optional[torch.Tensor] = None,
                         db_out: Optional[torch.Tensor] = None):
    """Reduction half of groupnorm_bwd: sums [n, groups, 2] = (sum g, sum g*x_hat), plus dgamma / dbeta."""
    da = _req(da, bf16, "da")
    x = _req(x, bf16, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    L = _lib.lib()
    ws = workspace(L.dmvae_groupnorm_workspace(n, hw, c, groups), x.device)
    sums = torch.empty(n, groups, 2, dtype=f32, device=x.device)
    dg = (dg_out if dg_out is not None else torch.empty(c, dtype=f32, device=x.device)) if need_param_grads else None
    db = (db_out if db_out is not None else torch.empty(c, dtype=f32, device=x.device)) if need_param_grads else None
    check(L.dmvae_groupnorm_bwd_reduce(da.data_ptr(), x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), beta.data_ptr(), sums.data_ptr(),
                                       _ptr(dg), _ptr(db), ws.data_ptr(), ws.numel(), n, hw, c, groups, int(act), 0, _stream()), "groupnorm_bwd_reduce")
    return sums, dg, db


def groupnorm_bwd_apply(da: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, sums: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                        act: int, groups: int = 32, inv_count: float = 0.0, dres: Optional[torch.Tensor] = None) -> torch.Tensor:
    da = _req(da, bf16, "da")
    x = _req(x, bf16, "x")
    n, c = x.shape[0], x.shape[-1]
    hw = x.numel() // (n * c)
    dx = torch.empty_like(x)
    check(_lib.lib().dmvae_groupnorm_bwd_apply(da.data_ptr(), x.data_ptr(), _ptr(dres), stats.data_ptr(), _req(sums, f32, "sums").data_ptr(),
                                               gamma.data_ptr(), beta.data_ptr(), dx.data_ptr(), n, hw, c, groups, int(act), float(inv_count),
                                               _stream()), "groupnorm_bwd_apply")
    return dx


# ---- layout / elementwise ---------------------------------------------------------------------------
def _im2col_out(h, w, ks, stride, pad):
    return (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1


def im2col(x: torch.Tensor, ks: int, stride: int, pad: int, taps_pad: int = 0, c_take: int = 0) -> torch.Tensor:
    """[N,H,W,C] bf16 -> [N,Ho,Wo,ks*ks*C] (tap-major, zero padding): the PatchGAN convs as GEMMs (models/patchgan.py:125-147).  taps_pad > ks*ks: that many
    taps per pixel, the extra ones columns of zeros (a reduction dimension padded to the consumer's tile: dmvae_im2col_nhwc_taps).  c_take: only the first
    c_take channels of every pixel (dmvae_im2col_nhwc_sub)."""
    x = _req(x, bf16, "x")
    n, h, w, cs = x.shape
    c = int(c_take) or cs
    ho, wo = _im2col_out(h, w, ks, stride, pad)
    tp = max(int(taps_pad), ks * ks)
    col = torch.empty(n, ho, wo, tp * c, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_im2col_nhwc_sub(x.data_ptr(), col.data_ptr(), n, h, w, cs, c, ks, stride, pad, tp, _stream()), "im2col_nhwc")
    return col


def col2im(dcol: torch.Tensor, h: int, w: int, ks: int, stride: int, pad: int) -> torch.Tensor:
    """Adjoint of im2col: [N,Ho,Wo,ks*ks*C] -> [N,H,W,C]."""
    assert dcol.dtype in (bf16, f32) and dcol.is_cuda and dcol.is_contiguous()
    n, ho, wo, k = dcol.shape
    c = k // (ks * ks)
    assert (ho, wo) == _im2col_out(h, w, ks, stride, pad) and c * ks * ks == k
    dx = torch.empty(n, h, w, c, dtype=bf16, device=dcol.device)
    check(_lib.lib().dmvae_col2im_nhwc(dcol.data_ptr(), dx.data_ptr(), n, h, w, c, ks, stride, pad, int(dcol.dtype == f32), _stream()), "col2im_nhwc")
    return dx


def batchnorm_running_update(stats: torch.Tensor, running_mean: torch.Tensor, running_var: torch.Tensor, eps: float, momentum: float, unbias: float) -> None:
    """nn.BatchNorm2d's running-estimate update from (mean, rstd) pairs stats [..., C, 2] f32, in place, one launch (include/dmvae_hip.h)."""
    c = running_mean.numel()
    assert stats.dtype == f32 and stats.is_contiguous() and stats.numel() == 2 * c and running_mean.dtype == f32 and running_var.dtype == f32
    check(_lib.lib().dmvae_batchnorm_running_update(stats.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), c, float(eps), float(momentum), float(unbias),
                                                    _stream()), "batchnorm_running_update")


def _diffaug_args(x, rand01, flags, cutout):
    assert x.dtype == f32 and x.is_cuda and x.is_contiguous() and x.dim() == 4
    b, c, h, w = x.shape
    rand01 = _req(rand01, f32, "rand01")
    assert rand01.numel() == 7 * b
    ws = torch.empty(36 * b, dtype=f32, device=x.device)      # b means + 16 f64 partials per image (include/dmvae_hip.h)
    # Python round() on the same float expressions as utils/diffaug.py:73-75,92-94
    return rand01, ws, (b, c, h, w, int(flags), round(h * 0.125), round(w * 0.125), round(h * cutout), round(w * cutout))


def diffaug(x: torch.Tensor, rand01: torch.Tensor, flags: int = 7, cutout: float = 0.2) -> torch.Tensor:
    """DiffAug forward on an NCHW f32 batch; flags: 1 translate | 2 colour | 4 cut-out; rand01 [7, B] device f32."""
    rand01, ws, a = _diffaug_args(x, rand01, flags, cutout)
    y = torch.empty_like(x)
    check(_lib.lib().dmvae_diffaug_fwd(x.data_ptr(), rand01.data_ptr(), y.data_ptr(), ws.data_ptr(), *a, _stream()), "diffaug_fwd")
    return y


def diffaug_bwd(dy: torch.Tensor, rand01: torch.Tensor, flags: int = 7, cutout: float = 0.2) -> torch.Tensor:
    rand01, ws, a = _diffaug_args(dy, rand01, flags, cutout)
    dx = torch.empty_like(dy)
    check(_lib.lib().dmvae_diffaug_bwd(dy.data_ptr(), rand01.data_ptr(), dx.data_ptr(), ws.data_ptr(), *a, _stream()), "diffaug_bwd")
    return dx


def leaky_relu_bwd(dy: torch.Tensor, y: torch.Tensor, slope: float = 0.2) -> torch.Tensor:
    if dy.dtype == f32 and parity.on():
        return parity.eltwise(3, dy, y, param=slope)
    dy = _req(dy, bf16, "dy")
    y = _req(y, bf16, "y")
    dx = torch.empty_like(dy)
    check(_lib.lib().dmvae_leaky_relu_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), float(slope), _stream()), "leaky_relu_bwd")
    return dx



def sumpool2x2(dy: torch.Tensor) -> torch.Tensor:
    if dy.dtype == f32 and parity.on():
        return parity.pool2x2(0, dy)
    dy = _req(dy, bf16, "dy")
    n, h2, w2, c = dy.shape
    dx = torch.empty(n, h2 // 2, w2 // 2, c, dtype=bf16, device=dy.device)
    check(_lib.lib().dmvae_sumpool2x2_nhwc(dy.data_ptr(), dx.data_ptr(), n, h2 // 2, w2 // 2, c, _stream()), "sumpool2x2_nhwc")
    return dx


def maxpool2x2(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == f32 and parity.on():
        return parity.pool2x2(1, x)
    x = _req(x, bf16, "x")
    n, h2, w2, c = x.shape
    y = torch.empty(n, h2 // 2, w2 // 2, c, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_maxpool2x2_nhwc(x.data_ptr(), y.data_ptr(), n, h2 // 2, w2 // 2, c, _stream()), "maxpool2x2_nhwc")
    return y


def maxpool2x2_relu_bwd(dpool: Optional[torch.Tensor], x: torch.Tensor, extra: Optional[torch.Tensor]) -> torch.Tensor:
    """dx = x > 0 ? route(dpool to the first maximum of each 2x2 window of x) + extra : 0."""
    if x.dtype == f32 and parity.on():
        return parity.pool2x2(2, dpool, x, extra)
    x = _req(x, bf16, "x")
    n, h2, w2, c = x.shape
    for name, t in (("dpool", dpool), ("extra", extra)):
        if t is not None:
            _req(t, bf16, name)
    dx = torch.empty_like(x)
    check(_lib.lib().dmvae_maxpool2x2_relu_bwd_nhwc(_ptr(dpool), x.data_ptr(), _ptr(extra), dx.data_ptr(), n, h2 // 2, w2 // 2, c, _stream()),
          "maxpool2x2_relu_bwd_nhwc")
    return dx


def relu_bwd(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    if dy.dtype == f32 and parity.on():
        return parity.eltwise(3, dy, y, param=0.0)
    dy = _req(dy, bf16, "dy")
    y = _req(y, bf16, "y")
    dx = torch.empty_like(dy)
    check(_lib.lib().dmvae_relu_bwd(dy.data_ptr(), y.data_ptr(), dx.data_ptr(), dy.numel(), _stream()), "relu_bwd")
    return dx


def nchw_to_nhwc_bf16(x: torch.Tensor, c_pad: int = 0) -> torch.Tensor:
    """NCHW f32 image -> NHWC activation (bf16; f32 in parity mode), channels zero-padded to c_pad."""
    if parity.on():
        return parity.nchw_to_nhwc(x, c_pad)
    x = _req(x, f32, "x")
    n, c, h, w = x.shape
    c_pad = max(c_pad, c)
    out = torch.empty(n, h, w, c_pad, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_nchw_f32_to_nhwc_bf16(x.data_ptr(), out.data_ptr(), n, c, h * w, c_pad, _stream()), "nchw_f32_to_nhwc_bf16")
    return out


def nhwc_to_nchw_f32(x: torch.Tensor, c: int) -> torch.Tensor:
    assert x.dtype in (bf16, f32) and x.is_cuda and x.is_contiguous()
    n, h, w, c_pad = x.shape
    out = torch.empty(n, c, h, w, dtype=f32, device=x.device)
    check(_lib.lib().dmvae_nhwc_to_nchw_f32(x.data_ptr(), out.data_ptr(), n, c, h * w, c_pad, int(x.dtype == f32), _stream()), "nhwc_to_nchw_f32")
    return out


def silu(x: torch.Tensor) -> torch.Tensor:
    if x.dtype == f32 and parity.on():
        return parity.eltwise(1, x)
    x = _req(x, bf16, "x")
    y = torch.empty_like(x)
    check(_lib.lib().dmvae_silu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "silu_fwd")
    return y


def silu_bwd(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    if x.dtype == f32 and parity.on():
        return parity.eltwise(2, x, dy)
    x = _req(x, bf16, "x")
    dy = _req(dy, bf16, "dy")
    dx = torch.empty_like(x)
    check(_lib.lib().dmvae_silu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "silu_bwd")
    return dx


# ---- frozen ViT encoder, elementwise ---------------------------------------------------------------------
def layernorm_bf16(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """LayerNorm over the last dim of an f32 tensor, bf16 result (what autocast feeds the next Linear)."""
    x = _req(x, f32, "x")
    c = x.shape[-1]
    y = torch.empty(x.shape, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_layernorm_f32_bf16(x.data_ptr(), _req(gamma, f32, "gamma").data_ptr(), _req(beta, f32, "beta").data_ptr(), y.data_ptr(),
                                              x.numel() // c, c, float(eps), _stream()), "layernorm_f32_bf16")
    return y


def softmax_rows_bf16(s: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(scale * s) over the last dim, bf16 in / bf16 out, f32 inside."""
    s = _req(s, bf16, "S")
    p = torch.empty_like(s)
    check(_lib.lib().dmvae_softmax_rows_bf16(s.data_ptr(), p.data_ptr(), s.numel() // s.shape[-1], s.shape[-1], float(scale), _stream()), "softmax_rows_bf16")
    return p


def attention_qkv(qkv: torch.Tensor, heads: int, scale: float, need_lse: bool = False):
    """Fused multi-head self-attention on the qkv Linear's output [B, S, 3*heads*64] (bf16) -> [B, S, heads*64]; with need_lse also the row statistics
    lse [B*heads, S] f32 (scale * max + log(sum) per query) for `attention_bwd_qkv`."""
    qkv = _req(qkv, bf16, "qkv")
    b, s, c3 = qkv.shape
    c = c3 // 3
    out = torch.empty(b, s, c, dtype=bf16, device=qkv.device)
    lse = torch.empty(b * heads, s, dtype=f32, device=qkv.device) if need_lse else None
    check(_lib.lib().dmvae_attention_qkv_lse_bf16(qkv.data_ptr(), out.data_ptr(), _ptr(lse), b, s, heads, c // heads, float(scale), _stream()), "attention_qkv_bf16")
    return (out, lse) if need_lse else out


def attention_heads(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, scale: float, need_lse: bool = False):
    """q, k [B*H, N, Dp], v [B*H, N, D] (bf16, as `qknorm_rope` returns them) -> softmax(scale q k^T) v as [B, N, H*D] bf16, one fused kernel; with need_lse also
    lse [B*H, N] f32 for `attention_bwd_heads`."""
    q = _req(q, bf16, "q"); k = _req(k, bf16, "k"); v = _req(v, bf16, "v")
    bh, n, dp = q.shape
    d = v.shape[-1]
    heads = bh // batch
    out = torch.empty(batch, n, heads * d, dtype=bf16, device=q.device)
    lse = torch.empty(bh, n, dtype=f32, device=q.device) if need_lse else None
    check(_lib.lib().dmvae_attention_heads_lse_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _ptr(lse), batch, n, heads, d, dp, float(scale),
                                                    _stream()), "attention_heads_bf16")
    return (out, lse) if need_lse else out


def attention_qknorm_rope(qkv: torch.Tensor, qw: torch.Tensor, kw: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, eps: float,
                          scale: float) -> torch.Tensor:
    """qkv [B,N,3*H*D] bf16 -> softmax(scale rope(norm(q)) rope(norm(k))^T) v as [B,N,H*D] bf16: `qknorm_rope` + `attention_heads` in one kernel."""
    qkv = _req(qkv, bf16, "qkv")
    b, n, c3 = qkv.shape
    c = c3 // 3
    out = torch.empty(b, n, c, dtype=bf16, device=qkv.device)
    check(_lib.lib().dmvae_attention_qknorm_rope_bf16(qkv.data_ptr(), _req(qw, f32, "qw").data_ptr(), _req(kw, f32, "kw").data_ptr(),
                                                      _req(cos, f32, "cos").data_ptr(), _req(sin, f32, "sin").data_ptr(), out.data_ptr(), b, n, heads,
                                                      c // heads, float(eps), float(scale), _stream()), "attention_qknorm_rope_bf16")
    return out


def attention_bwd_qkv(qkv: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, heads: int, scale: float, lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d(qkv) [B,S,3*C] of `attention_qkv` from its input, its result `out` [B,S,C] and d(out) (bf16): one fused kernel (csrc/attention_bwd.hip); with the forward's
    `lse` the eight-wave form that rebuilds the probabilities from it."""
    qkv = _req(qkv, bf16, "qkv"); out = _req(out, bf16, "out"); dout = _req(dout, bf16, "dout")
    b, s, c3 = qkv.shape
    c = c3 // 3
    assert out.shape == (b, s, c) and dout.shape == (b, s, c)
    dqkv = torch.empty_like(qkv)
    assert lse is None or (lse.dtype == f32 and lse.is_contiguous() and lse.shape == (b * heads, s))
    check(_lib.lib().dmvae_attention_bwd_qkv_lse_bf16(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), _ptr(lse), dqkv.data_ptr(), b, s, heads, c // heads, float(scale),
                                                      _stream()), "attention_bwd_qkv_bf16")
    return dqkv


def attention_bwd_heads(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, dout: torch.Tensor, batch: int, scale: float,
                        lse: Optional[torch.Tensor] = None):
    """(dq, dk [B*H,N,Dp], dv [B*H,N,D]) of `attention_heads` from its operands, its result `out` [B,N,H*D] and d(out)."""
    q = _req(q, bf16, "q"); k = _req(k, bf16, "k"); v = _req(v, bf16, "v"); out = _req(out, bf16, "out"); dout = _req(dout, bf16, "dout")
    bh, n, dp = q.shape
    d = v.shape[-1]
    heads = bh // batch
    assert out.shape == (batch, n, heads * d) and dout.shape == out.shape
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    assert lse is None or (lse.dtype == f32 and lse.is_contiguous() and lse.shape == (bh, n))
    check(_lib.lib().dmvae_attention_bwd_heads_lse_bf16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr(), _ptr(lse), dq.data_ptr(),
                                                        dk.data_ptr(), dv.data_ptr(), batch, n, heads, d, dp, float(scale), _stream()), "attention_bwd_heads_bf16")
    return dq, dk, dv


def attention_heads_supported(n: int, d: int) -> bool:
    return n <= 288 and d % 8 == 0 and (d + 31) // 32 * 32 in (64, 96)


def scale_residual_layernorm_(x: torch.Tensor, r: torch.Tensor, ls_gamma: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """x (f32, in place) += ls_gamma * r (bf16), then LayerNorm(x) -> bf16: `scale_residual_` + `layernorm_bf16` in one pass over the residual stream."""
    x = _req(x, f32, "x")
    r = _req(r, bf16, "r")
    assert x.shape == r.shape
    c = x.shape[-1]
    y = torch.empty(x.shape, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_scale_residual_layernorm(x.data_ptr(), r.data_ptr(), _req(ls_gamma, f32, "ls_gamma").data_ptr(), _req(gamma, f32, "gamma").data_ptr(),
                                                    _req(beta, f32, "beta").data_ptr(), y.data_ptr(), x.numel() // c, c, float(eps), _stream()),
          "scale_residual_layernorm")
    return y


def scale_residual_(x: torch.Tensor, y: torch.Tensor, gamma: torch.Tensor) -> torch.Tensor:
    """x (f32, in place) += gamma * y (bf16): LayerScale + residual add."""
    x = _req(x, f32, "x")
    y = _req(y, bf16, "y")
    assert x.shape == y.shape
    c = x.shape[-1]
    check(_lib.lib().dmvae_scale_residual_f32(x.data_ptr(), y.data_ptr(), _req(gamma, f32, "gamma").data_ptr(), x.numel() // c, c, _stream()),
          "scale_residual_f32")
    return x


def layernorm_bwd_(dx_io: torch.Tensor, dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, eps: float = 1e-6, need_param_grads: bool = True,
                   dg_out: Optional[torch.Tensor] = None, db_out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """dx_io (f32, in place) += LayerNorm backward of dy (bf16) at input x (f32); returns (dgamma, dbeta) or (None, None)."""
    dx_io = _req(dx_io, f32, "dx_io")
    dy = _req(dy, bf16, "dy")
    x = _req(x, f32, "x")
    c = x.shape[-1]
    rows = x.numel() // c
    L = _lib.lib()
    ws = workspace(L.dmvae_vit_bwd_workspace(c), x.device)
    dg = (dg_out if dg_out is not None else torch.empty(c, dtype=f32, device=x.device)) if need_param_grads else None
    db = (db_out if db_out is not None else torch.empty(c, dtype=f32, device=x.device)) if need_param_grads else None
    check(L.dmvae_layernorm_bwd_f32(dy.data_ptr(), x.data_ptr(), _req(gamma, f32, "gamma").data_ptr(), dx_io.data_ptr(), _ptr(dg), _ptr(db),
                                    ws.data_ptr(), ws.numel(), rows, c, float(eps), int(accumulate), _stream()), "layernorm_bwd_f32")
    return dg, db


def layerscale_bwd(dt: torch.Tensor, y: torch.Tensor, gamma: torch.Tensor, dg_out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """For r = x + gamma * y: returns (dy = gamma * dt as bf16, dgamma = sum_rows dt * y)."""
    dt = _req(dt, f32, "dt")
    y = _req(y, bf16, "y")
    c = y.shape[-1]
    rows = y.numel() // c
    L = _lib.lib()
    ws = workspace(L.dmvae_vit_bwd_workspace(c), y.device)
    dy = torch.empty_like(y)
    dg = dg_out if dg_out is not None else torch.empty(c, dtype=f32, device=y.device)
    check(L.dmvae_layerscale_bwd(dt.data_ptr(), y.data_ptr(), _req(gamma, f32, "gamma").data_ptr(), dy.data_ptr(), dg.data_ptr(), ws.data_ptr(),
                                 ws.numel(), rows, c, int(accumulate), _stream()), "layerscale_bwd")
    return dy, dg


def gelu(x: torch.Tensor) -> torch.Tensor:
    x = _req(x, bf16, "x")
    y = torch.empty_like(x)
    check(_lib.lib().dmvae_gelu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "gelu_fwd")
    return y


def gelu_bwd(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    dy = _req(dy, bf16, "dy")
    x = _req(x, bf16, "x")
    dx = torch.empty_like(x)
    check(_lib.lib().dmvae_gelu_bwd(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "gelu_bwd")
    return dx


# ---- LightningDiT inference path (csrc/dit.hip) -------------------------------------------------------
def rmsnorm_modulate(x: torch.Tensor, w: torch.Tensor, mod: torch.Tensor, shift_off: int, scale_off: int, eps: float = 1e-6) -> torch.Tensor:
    """bf16( RMSNorm(x [B,N,C] f32; w) * bf16(1 + scale[b]) + shift[b] ); mod [B, k*C] bf16 holds the adaLN chunks at the given element offsets."""
    x = _req(x, f32, "x")
    mod = _req(mod, bf16, "mod")
    b, n, c = x.shape
    y = torch.empty(b, n, c, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_rmsnorm_modulate_bf16(x.data_ptr(), _req(w, f32, "w").data_ptr(), mod.data_ptr(), y.data_ptr(), b * n, n, c, mod.shape[1],
                                                 int(shift_off), int(scale_off), float(eps), _stream()), "rmsnorm_modulate_bf16")
    return y


def gated_residual_rmsnorm_modulate_(x: torch.Tensor, r: torch.Tensor, gate_mod: torch.Tensor, gate_off: int, w: torch.Tensor, mod: torch.Tensor,
                                     shift_off: int, scale_off: int, eps: float = 1e-6) -> torch.Tensor:
    """x [B,N,C] (f32, in place) += bf16(gate_mod[b, gate_off:gate_off+C] * r); returns rmsnorm_modulate(x, w, mod, shift_off, scale_off) -- one pass."""
    x = _req(x, f32, "x"); r = _req(r, bf16, "r"); gate_mod = _req(gate_mod, bf16, "gate_mod"); mod = _req(mod, bf16, "mod")
    b, n, c = x.shape
    y = torch.empty(b, n, c, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_gated_residual_rmsnorm_modulate(x.data_ptr(), r.data_ptr(), gate_mod.data_ptr(), gate_mod.shape[1], int(gate_off),
                                                           _req(w, f32, "w").data_ptr(), mod.data_ptr(), y.data_ptr(), b * n, n, c, mod.shape[1],
                                                           int(shift_off), int(scale_off), float(eps), _stream()), "gated_residual_rmsnorm_modulate")
    return y


def gated_residual_out(x: torch.Tensor, r: torch.Tensor, gate_mod: torch.Tensor, gate_off: int, w: Optional[torch.Tensor] = None,
                       mod: Optional[torch.Tensor] = None, shift_off: int = -1, scale_off: int = 0, eps: float = 1e-6):
    """-> (x_new = x + bf16(gate * r) as a NEW tensor, rmsnorm_modulate(x_new, w, mod, ...) or None when w is None): the out-of-place form of
    `gated_residual_` / `gated_residual_rmsnorm_modulate_` for the training route (x is kept for the backward pass)."""
    x = _req(x, f32, "x"); r = _req(r, bf16, "r"); gate_mod = _req(gate_mod, bf16, "gate_mod")
    b, n, c = x.shape
    xo = torch.empty_like(x)
    y = None
    if w is not None:
        mod = _req(mod, bf16, "mod")
        y = torch.empty(b, n, c, dtype=bf16, device=x.device)
    check(_lib.lib().dmvae_gated_residual_out(x.data_ptr(), xo.data_ptr(), r.data_ptr(), gate_mod.data_ptr(), gate_mod.shape[1], int(gate_off),
                                              _ptr(_req(w, f32, "w") if w is not None else None), _ptr(mod), _ptr(y), b * n, n, c,
                                              mod.shape[1] if mod is not None else 0, int(shift_off), int(scale_off), float(eps), _stream()), "gated_residual_out")
    return xo, y


def qknorm_rope(qkv: torch.Tensor, qw: torch.Tensor, kw: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, eps: float = 1e-6, padded: bool = True):
    """qkv [B,N,3*H*D] bf16 -> (q, k [B*H, N, Dp] with Dp = D rounded up to 32, v [B*H, N, D]): per-head RMSNorm * weight + 2-D RoPE on q and k.  padded=False
    (D a multiple of 8): q, k rows of D channels -- the fused attention kernels then load and store D of their 96 channels (a quarter fewer q / k bytes at D = 72)."""
    qkv = _req(qkv, bf16, "qkv")
    b, n, c3 = qkv.shape
    d = c3 // 3 // heads
    dp = (d + 31) // 32 * 32 if (padded or d % 8) else d
    q = torch.empty(b * heads, n, dp, dtype=bf16, device=qkv.device)
    k = torch.empty_like(q)
    v = torch.empty(b * heads, n, d, dtype=bf16, device=qkv.device)
    check(_lib.lib().dmvae_qknorm_rope_bf16(qkv.data_ptr(), _req(qw, f32, "qw").data_ptr(), _req(kw, f32, "kw").data_ptr(), _req(cos, f32, "cos").data_ptr(),
                                            _req(sin, f32, "sin").data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), b, n, heads, d, dp, float(eps),
                                            _stream()), "qknorm_rope_bf16")
    return q, k, v


def swiglu(x12: torch.Tensor) -> torch.Tensor:
    x12 = _req(x12, bf16, "x12")
    hid = x12.shape[-1] // 2
    out = torch.empty(*x12.shape[:-1], hid, dtype=bf16, device=x12.device)
    check(_lib.lib().dmvae_swiglu_bf16(x12.data_ptr(), out.data_ptr(), x12.numel() // (2 * hid), hid, _stream()), "swiglu_bf16")
    return out


def gated_residual_(x: torch.Tensor, y: torch.Tensor, mod: torch.Tensor, gate_off: int) -> torch.Tensor:
    """x [B,N,C] (f32, in place) += bf16(gate[b] * y); gate = mod[:, gate_off : gate_off + C]."""
    x = _req(x, f32, "x")
    y = _req(y, bf16, "y")
    mod = _req(mod, bf16, "mod")
    b, n, c = x.shape
    check(_lib.lib().dmvae_gated_residual_f32(x.data_ptr(), y.data_ptr(), mod.data_ptr(), b * n, n, c, mod.shape[1], int(gate_off), _stream()),
          "gated_residual_f32")
    return x


def gated_residual_bwd(dx: torch.Tensor, y: torch.Tensor, mod: torch.Tensor, dmod: torch.Tensor, gate_off: int) -> torch.Tensor:
    """-> dy = bf16(gate[b] * dx); fills dmod[:, gate_off : gate_off + C] (f32) with sum_n dx * y."""
    dx = _req(dx, f32, "dx"); y = _req(y, bf16, "y"); mod = _req(mod, bf16, "mod"); dmod = _req(dmod, f32, "dmod")
    b, n, c = dx.shape
    dy = torch.empty_like(y)
    check(_lib.lib().dmvae_gated_residual_bwd(dx.data_ptr(), y.data_ptr(), mod.data_ptr(), dy.data_ptr(), dmod.data_ptr(), b, n, c, mod.shape[1],
                                              int(gate_off), _stream()), "gated_residual_bwd")
    return dy


def swiglu_bwd(dh: torch.Tensor, x12: torch.Tensor) -> torch.Tensor:
    dh = _req(dh, bf16, "dh"); x12 = _req(x12, bf16, "x12")
    hid = x12.shape[-1] // 2
    dx12 = torch.empty_like(x12)
    check(_lib.lib().dmvae_swiglu_bwd(dh.data_ptr(), x12.data_ptr(), dx12.data_ptr(), x12.numel() // (2 * hid), hid, _stream()), "swiglu_bwd")
    return dx12


def rmsnorm_modulate_bwd_(dx_io: torch.Tensor, da: torch.Tensor, x: torch.Tensor, w: torch.Tensor, mod: torch.Tensor, dmod: torch.Tensor,
                          shift_off: int, scale_off: int, eps: float = 1e-6, dw_out: Optional[torch.Tensor] = None, accumulate: bool = False):
    """dx_io (f32, in place) += backward of rmsnorm_modulate at x; fills the shift / scale chunks of dmod; returns dw [C] f32."""
    dx_io = _req(dx_io, f32, "dx_io"); da = _req(da, bf16, "da"); x = _req(x, f32, "x"); mod = _req(mod, bf16, "mod"); dmod = _req(dmod, f32, "dmod")
    b, n, c = x.shape
    L = _lib.lib()
    ws = workspace(L.dmvae_dit_bwd_workspace(b, c), x.device)
    dw = dw_out if dw_out is not None else torch.empty(c, dtype=f32, device=x.device)
    check(L.dmvae_rmsnorm_modulate_bwd(da.data_ptr(), x.data_ptr(), _req(w, f32, "w").data_ptr(), mod.data_ptr(), dx_io.data_ptr(), dmod.data_ptr(),
                                       dw.data_ptr(), ws.data_ptr(), ws.numel(), b, n, c, mod.shape[1], int(shift_off), int(scale_off), float(eps),
                                       int(accumulate), _stream()), "rmsnorm_modulate_bwd")
    return dw


def qknorm_rope_bwd(dq, dk, dv, qkv, qw, kw, cos, sin, heads: int, eps: float = 1e-6, dqw_out=None, dkw_out=None, accumulate: bool = False):
    """-> (dqkv [B,N,3*H*D] bf16, dq_weight [D], dk_weight [D])."""
    dq = _req(dq, bf16, "dq"); dk = _req(dk, bf16, "dk"); dv = _req(dv, bf16, "dv"); qkv = _req(qkv, bf16, "qkv")
    b, n, c3 = qkv.shape
    d = c3 // 3 // heads
    dp = dq.shape[-1]
    L = _lib.lib()
    ws = workspace(L.dmvae_dit_bwd_workspace(b, c3 // 3), qkv.device)
    dqkv = torch.empty_like(qkv)
    dqw = dqw_out if dqw_out is not None else torch.empty(d, dtype=f32, device=qkv.device)
    dkw = dkw_out if dkw_out is not None else torch.empty(d, dtype=f32, device=qkv.device)
    check(L.dmvae_qknorm_rope_bwd(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), qkv.data_ptr(), _req(qw, f32, "qw").data_ptr(), _req(kw, f32, "kw").data_ptr(),
                                  _req(cos, f32, "cos").data_ptr(), _req(sin, f32, "sin").data_ptr(), dqkv.data_ptr(), dqw.data_ptr(), dkw.data_ptr(),
                                  ws.data_ptr(), ws.numel(), b, n, heads, d, dp, float(eps), int(accumulate), _stream()), "qknorm_rope_bwd")
    return dqkv, dqw, dkw


# ---- whole-stack LightningDiT backward + batched per-sample Linears (csrc/dit_stack.hip, linear_rows.hip) ---------------------------
class _TableCache(dict):
    """key -> device table(s) that LAUNCHED KERNELS READ (pointer tables of the batched / grouped entry points).  Dropping an entry frees device memory a kernel in
    flight -- or a captured hipGraph, which has the table's address baked in (sample.GraphedInference with batched adaLN) -- may still read, so eviction is
    restricted:
      * an entry is PINNED, never evicted, once it is looked up a second time (tables over parameter / flat-buffer / cached-operand pointers recur every step;
        the ones an unsettled allocator produces from activation pointers are used once) or when it was built or used during stream capture;
      * when more than LIMIT unpinned entries have piled up, they leave the dict but their tensors move to `grave`, which is only released at the NEXT eviction
        -- thousands of table builds later, each of them a blocking host-to-device copy behind every kernel that could have read the old table.
    `clear()` (tests) keeps dict semantics."""
    LIMIT = 4096

    def __init__(self):
        super().__init__()
        self.pinned = set()
        self.grave = []

    def lookup(self, key):
        hit = self.get(key)
        if hit is not None and key not in self.pinned:
            self.pinned.add(key)
        return hit

    def store(self, key, value):
        if len(self) - len(self.pinned) > self.LIMIT:
            victims = [k for k in self if k not in self.pinned]
            self.grave = [self.pop(k) for k in victims]          # the previous generation is released here; this one stays allocated until the next eviction
        self[key] = value
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self.pinned.add(key)

    def discard(self, key):
        v = self.pop(key, None)
        self.pinned.discard(key)
        if v is not None:
            self.grave.append(v)

    def clear(self):
        super().clear()
        self.pinned.clear()
        self.grave = []


_PTR_TABLES = _TableCache()
TABLE_BUILDS = [0]      # device tables built so far (each build is a small synchronous host-to-device copy): stays flat once the allocator's address pattern has settled


def ptr_table(tensors) -> torch.Tensor:
    """Device array of the tensors' data pointers (int64), cached per pointer tuple: the per-layer weight / destination tables of the batched entry points.
    The pointers of a model's parameters, of an optimiser's flat-buffer views and of cached operands are stable across steps, so a table is uploaded once."""
    key = tuple(0 if t is None else t.data_ptr() for t in tensors)
    dev = next(t for t in tensors if t is not None).device
    hit = _PTR_TABLES.lookup((key, dev))
    if hit is None:
        hit = torch.tensor(key, dtype=torch.int64).to(dev)
        TABLE_BUILDS[0] += 1
        _PTR_TABLES.store((key, dev), hit)
    return hit


def linear_rows_batched(x: torch.Tensor, ws, biases=None, act: int = ACT_NONE, out_f32: bool = False) -> torch.Tensor:
    """y [L, M, N] = act(x_l @ ws[l]^T + biases[l]) for L per-sample Linears of one shape in ONE launch (include/dmvae_hip.h dmvae_linear_rows_batched_bf16): x [M, K]
    (one input for all layers) or [L, M, K] bf16; ws: L bf16 weights, all [N, K] row-major or all K-tile-major [K / 32, N, 32]; biases: L f32 / bf16 vectors or None."""
    nl = len(ws)
    if x.dim() == 2:
        xs, x2 = 0, _req2d(x, "x")
        m, k = x2.shape
        ldx = x2.stride(0)
    else:
        x = _req(x, bf16, "x")
        assert x.dim() == 3 and x.shape[0] == nl
        m, k = x.shape[1], x.shape[2]
        xs, ldx, x2 = m * k, k, x
    w0 = ws[0]
    if w0.dim() == 3:
        n, layout, ldw = w0.shape[1], 1, 0
        assert all(w.dtype == bf16 and w.is_contiguous() and w.shape == w0.shape for w in ws) and w0.shape[0] * 32 == k and w0.shape[2] == 32
    else:
        n, layout, ldw = w0.shape[0], 0, w0.stride(0)
        assert all(w.dtype == bf16 and w.shape == w0.shape and w.stride() == w0.stride() and w.stride(1) == 1 for w in ws) and w0.shape[1] == k
    bias_bf16 = 0
    if biases is not None:
        assert len(biases) == nl and all(b.is_contiguous() and b.numel() == n and b.dtype == biases[0].dtype for b in biases) and biases[0].dtype in (bf16, f32)
        bias_bf16 = int(biases[0].dtype == bf16)
    y = torch.empty(nl, m, n, dtype=f32 if out_f32 else bf16, device=x.device)
    check(_lib.lib().dmvae_linear_rows_batched_bf16(x2.data_ptr(), xs, ptr_table(ws).data_ptr(), None if biases is None else ptr_table(biases).data_ptr(), y.data_ptr(),
                                                    m * n, nl, m, n, k, ldx, ldw, n, int(act), bias_bf16, int(out_f32), layout, _stream()), "linear_rows_batched_bf16")
    return y


def rows_transposed(x: torch.Tensor) -> torch.Tensor:
    """x [M, K] (M <= 64) -> bf16 [K, mp] with mp = 32 or 64 and zeros beyond M: the shared-input operand of `linear_rows_wgrad_batched`."""
    m, k = x.shape
    mp = 32 if m <= 32 else 64
    xt = torch.zeros(k, mp, dtype=bf16, device=x.device)
    xt[:, :m] = x.t()
    return xt


def linear_rows_wgrad_batched(dy: torch.Tensor, xt: torch.Tensor, dws, dbs=None, accumulate: bool = False) -> None:
    """dws[l] [N, K] f32 (+)= dy[l]^T @ x, dbs[l] [N] f32 (+)= column sums of dy[l]: the weight / bias gradients of L per-sample Linears sharing their input x, in one
    launch on the matrix cores.  dy [L, M, N] bf16; xt = rows_transposed(x)."""
    dy = _req(dy, bf16, "dy")
    nl, m, n = dy.shape
    k, mp = xt.shape
    assert xt.dtype == bf16 and xt.is_contiguous() and mp in (32, 64) and mp >= m and len(dws) == nl
    assert all(w.dtype == f32 and w.is_contiguous() and w.numel() == n * k for w in dws)
    if dbs is not None:
        assert len(dbs) == nl and all(b.dtype == f32 and b.is_contiguous() and b.numel() == n for b in dbs)
    check(_lib.lib().dmvae_linear_rows_wgrad_batched(dy.data_ptr(), m * n, xt.data_ptr(), mp, ptr_table(dws).data_ptr(), None if dbs is None else ptr_table(dbs).data_ptr(),
                                                     None, None, nl, m, n, k, n, int(accumulate), _stream()), "linear_rows_wgrad_batched")


def linear_weight_t_kmajor_batched(pairs) -> None:
    """`linear_weight_t_kmajor` for a list of (src bf16 [N, K], dst bf16 [N / 32, K, 32]) in ONE launch; the table lives on the device, cached per pointer set."""
    key = ("wt",) + tuple((s_.data_ptr(), d_.data_ptr(), s_.shape[0], s_.shape[1]) for s_, d_ in pairs)
    dev = pairs[0][0].device
    hit = _PTR_TABLES.lookup((key, dev))
    if hit is None:
        import ctypes
        assert _lib.lib().dmvae_wt_entry_bytes() == ctypes.sizeof(_lib.WtEntry)
        arr = (_lib.WtEntry * len(pairs))()
        start = 0
        for i, (src, dst) in enumerate(pairs):
            n, k = src.shape
            assert src.dtype == bf16 and dst.dtype == bf16 and src.is_contiguous() and dst.is_contiguous() and n % 32 == 0 and k % 8 == 0 and dst.numel() == n * k
            tx = (k + 63) // 64
            arr[i] = _lib.WtEntry(src.data_ptr(), dst.data_ptr(), n, k, start, tx)
            start += tx * (n // 32)
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        TABLE_BUILDS[0] += 1
        hit = (tab, len(pairs), start)
        _PTR_TABLES.store((key, dev), hit)
    tab, cnt, total = hit
    check(_lib.lib().dmvae_linear_weight_t_kmajor_batched(tab.data_ptr(), cnt, total, _stream()), "linear_weight_t_kmajor_batched")


def linear_wgrad_grouped_supported(m: int, cout: int, cin: int) -> bool:
    return bool(_lib.lib().dmvae_linear_wgrad_grouped_supported(int(m), int(cout), int(cin)))


WGRAD_GROUPED_XCD_MIN = 16     # problems: below it the first form (a ViT block's four problems are 11 chunks for 8 XCDs -- measured 8 % slower placed, profiles/r5_grouped_wgrad_placed_ab.txt)
WGRAD_GROUPED_XCD = os.environ.get("DMVAE_WGRAD_GROUPED_XCD", "1") != "0"      # grouped weight gradients: tiles placed on the XCDs (0: the first form, every problem spread over all eight)


def linear_wgrad_grouped(problems) -> None:
    """Weight (+ bias) gradients of a LIST of independent Linears in one launch (include/dmvae_hip.h dmvae_linear_wgrad_grouped): problems = [(dy [M, cout] bf16,
    x [M, cin] bf16, dw f32 [cout, cin] (written), db f32 [cout] or None), ...]; each problem unsplit, results written straight to dw / db."""
    import ctypes
    L = _lib.lib()
    dev = problems[0][0].device
    key = ("wg",) + tuple((dy.data_ptr(), x.data_ptr(), dw.data_ptr(), 0 if db is None else db.data_ptr(), dy.shape[0], dy.shape[1], x.shape[1]) for dy, x, dw, db in problems)
    hit = _PTR_TABLES.lookup((key, dev))
    if hit is None:
        eb, bb = L.dmvae_linear_wgrad_grouped_entry_bytes(), L.dmvae_linear_wgrad_grouped_bias_entry_bytes()
        n = len(problems)
        tab = ctypes.create_string_buffer(eb * n)
        btab = ctypes.create_string_buffer(bb * n)
        start, bstart = ctypes.c_uint(0), ctypes.c_uint(0)
        nb, ragged, part_floats = 0, 0, 0
        for dy, x, dw, db in problems:
            if db is not None:
                part_floats += L.dmvae_linear_wgrad_grouped_bias_parts(x.shape[1]) * dy.shape[1]
        part = workspace(max(part_floats, 1) * 4, dev, "wgrad_grouped_bias")
        off = 0
        for i, (dy, x, dw, db) in enumerate(problems):
            dy, x = _req2d(dy, "dy"), _req2d(x, "x")
            m, cout = dy.shape
            cin = x.shape[1]
            assert x.shape[0] == m and dy.is_contiguous() and x.is_contiguous() and dw.dtype == f32 and dw.is_contiguous() and dw.numel() == cout * cin
            assert db is None or (db.dtype == f32 and db.is_contiguous() and db.numel() == cout)
            ragged |= int(m % 32 != 0)
            check(L.dmvae_linear_wgrad_grouped_fill(ctypes.addressof(tab) + i * eb, ctypes.addressof(btab) + nb * bb if db is not None else None, dy.data_ptr(), x.data_ptr(),
                                                    dw.data_ptr(), part.data_ptr() + off * 4 if db is not None else None, _ptr(db), m, cout, cin, ctypes.byref(start),
                                                    ctypes.byref(bstart)), "linear_wgrad_grouped_fill")
            if db is not None:
                off += L.dmvae_linear_wgrad_grouped_bias_parts(cin) * cout
                nb += 1
        TABLE_BUILDS[0] += 1
        dtab = torch.frombuffer(bytearray(tab.raw), dtype=torch.uint8).to(dev)
        dbtab = torch.frombuffer(bytearray(btab.raw[:max(nb, 1) * bb]), dtype=torch.uint8).to(dev)
        plan = None
        if WGRAD_GROUPED_XCD and n >= WGRAD_GROUPED_XCD_MIN:       # every tile placed on an XCD (include/dmvae_hip.h dmvae_linear_wgrad_grouped_plan): chunk records + the XCDs' ranges
            cb = L.dmvae_linear_wgrad_grouped_chunk_bytes()
            chunks = ctypes.create_string_buffer(cb * n * 64)
            nch, grid = ctypes.c_int(0), ctypes.c_uint(0)
            xoff = (ctypes.c_uint * 9)()
            check(L.dmvae_linear_wgrad_grouped_plan(ctypes.addressof(tab), n, ctypes.addressof(chunks), n * 64, ctypes.byref(nch), xoff, ctypes.byref(grid)), "linear_wgrad_grouped_plan")
            plan = (torch.frombuffer(bytearray(chunks.raw[:nch.value * cb]), dtype=torch.uint8).to(dev), xoff, grid.value)
        hit = (dtab, n, start.value, ragged, dbtab, nb, bstart.value, part.data_ptr(), plan)
        _PTR_TABLES.store((key, dev), hit)
    dtab, n, total, ragged, dbtab, nb, btotal, part_ptr, plan = hit
    if nb and workspace(1, dev, "wgrad_grouped_bias").data_ptr() != part_ptr:      # the workspace slot grew since this table was built: its bias-partial pointers are stale
        _PTR_TABLES.discard((key, dev))
        return linear_wgrad_grouped(problems)
    if plan is not None:
        check(L.dmvae_linear_wgrad_grouped_xcd(dtab.data_ptr(), plan[0].data_ptr(), plan[1], plan[2], ragged, dbtab.data_ptr() if nb else None, nb, btotal, _stream()),
              "linear_wgrad_grouped_xcd")
        return
    check(L.dmvae_linear_wgrad_grouped(dtab.data_ptr(), n, total, ragged, dbtab.data_ptr() if nb else None, nb, btotal, _stream()), "linear_wgrad_grouped")


class DitStackBwd:
    """Scratch of one whole-stack backward pass (functional.DitStackFn.backward): the boundary slots' partial sums, the deferred norm-weight partials of the QK-norm,
    the row statistics -- see include/dmvae_hip.h (dmvae_dit_boundary_bwd, dmvae_dit_stack_finalize, dmvae_colsum2_batched)."""

    def __init__(self, layers: int, batch: int, seq: int, c: int, heads: int, device):
        L = _lib.lib()
        self.layers, self.batch, self.seq, self.c, self.heads = layers, batch, seq, c, heads
        self.d = c // heads
        self.dp = (self.d + 31) // 32 * 32
        self.bps = L.dmvae_dit_stack_bps(batch)
        self.slot_elems = batch * self.bps * 4 * c
        self.part = workspace(L.dmvae_dit_stack_part_bytes(layers, batch, c), device, "dit_stack_part").view(torch.float32)
        self.ws_bytes = L.dmvae_dit_stack_workspace(layers, batch, seq, c)
        self.ws = workspace(self.ws_bytes, device, "dit_stack_ws")
        self.rowstat_ptr = self.ws.data_ptr() + 2 * layers * batch * c * 4
        self.nblk = L.dmvae_qknorm_rope_bwd_nblk(batch, seq, heads, self.d, self.dp)
        self.qk_part = workspace(layers * self.nblk * 2 * self.d * 4, device, "dit_stack_qk").view(torch.float32)

    def boundary(self, slot: int, dt: torch.Tensor, da=None, x=None, w=None, mod=None, scale_off: int = 0, eps: float = 1e-6, y=None, gate_mod=None, gate_off: int = 0):
        """The pass over dt at boundary `slot`: norm half when `da` is given, gate half (-> dy) when `y` is given."""
        dy = torch.empty_like(y) if y is not None else None
        check(_lib.lib().dmvae_dit_boundary_bwd(_ptr(da), _ptr(x), _ptr(w), _ptr(mod), mod.shape[-1] if mod is not None else 0, int(scale_off), float(eps), dt.data_ptr(),
                                                _ptr(y), _ptr(gate_mod), gate_mod.shape[-1] if gate_mod is not None else 0, int(gate_off), _ptr(dy),
                                                self.part.data_ptr() + slot * self.slot_elems * 4, self.rowstat_ptr, self.batch, self.seq, self.c, _stream()),
              "dit_boundary_bwd")
        return dy

    def qknorm_rope_bwd(self, layer: int, dq, dk, dv, qkv, qw, kw, cos, sin, eps: float) -> torch.Tensor:
        dqkv = torch.empty_like(qkv)
        nbytes = self.nblk * 2 * self.d * 4
        check(_lib.lib().dmvae_qknorm_rope_bwd_partial(dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), qkv.data_ptr(), qw.data_ptr(), kw.data_ptr(), cos.data_ptr(),
                                                       sin.data_ptr(), dqkv.data_ptr(), self.qk_part.data_ptr() + layer * nbytes, nbytes, self.batch, self.seq,
                                                       self.heads, self.d, dq.shape[-1], float(eps), _stream()), "qknorm_rope_bwd_partial")
        return dqkv

    def finalize(self, dmod: torch.Tensor, norm_dws, qn_dws, kn_dws) -> None:
        """dmod bf16 [L, B, 6C] <- every boundary's sums; norm_dws: 2 L f32 [C] destinations (norm1, norm2 of block 0, 1, ...); qn_dws / kn_dws: L f32 [D] each."""
        L = _lib.lib()
        assert dmod.dtype == bf16 and dmod.is_contiguous() and dmod.shape == (self.layers, self.batch, 6 * self.c) and len(norm_dws) == 2 * self.layers
        check(L.dmvae_dit_stack_finalize(self.part.data_ptr(), dmod.data_ptr(), self.ws.data_ptr(), self.ws_bytes, ptr_table(norm_dws).data_ptr(), self.layers,
                                         self.batch, self.seq, self.c, 0, _stream()), "dit_stack_finalize")
        check(L.dmvae_colsum2_batched(self.qk_part.data_ptr(), ptr_table(qn_dws).data_ptr(), ptr_table(kn_dws).data_ptr(), self.layers, self.nblk, self.d, 0,
                                      _stream()), "colsum2_batched")


# ---- downstream consumers (sampler state update, image -> uint8) ---------------------------------------
def sde_euler_step(x: torch.Tensor, v: torch.Tensor, w: Optional[torch.Tensor], rar: float, var: float, diff: float, dt: float, sqrt_2diff: float,
                   sqrt_dt: float, need_mean: bool = False):
    """One Euler-Maruyama step (integrators.py:27-35 with transport.py:254-257 / path.py:74-89 folded in) -> (x_new, mean_x or None); w = None:
    x_new = x + drift * dt (the sampler's last step).  x, w f32; v bf16 or f32; all the same shape."""
    x = _req(x, f32, "x")
    assert v.is_cuda and v.is_contiguous() and v.dtype in (bf16, f32) and v.shape == x.shape
    if w is not None:
        w = _req(w, f32, "w")
        assert w.shape == x.shape
    out = torch.empty_like(x)
    mean = torch.empty_like(x) if need_mean else None
    check(_lib.lib().dmvae_sde_euler_step(x.data_ptr(), v.data_ptr(), int(v.dtype == bf16), _ptr(w), out.data_ptr(), _ptr(mean), x.numel(), float(rar),
                                          float(var), float(diff), float(dt), float(sqrt_2diff), float(sqrt_dt), _stream()), "sde_euler_step")
    return out, mean


def image_to_u8(y: torch.Tensor, channels: int, round_bf16: bool = False) -> torch.Tensor:
    """y [N,H,W,Cs] f32 (NHWC, first `channels` used) -> [N,H,W,channels] uint8 = clamp(127.5 y + 128, 0, 255) truncated (sample_50k.py:151)."""
    y = _req(y, f32, "y")
    n, h, w_, cs = y.shape
    out = torch.empty(n, h, w_, channels, dtype=torch.uint8, device=y.device)
    check(_lib.lib().dmvae_image_to_u8(y.data_ptr(), out.data_ptr(), n * h * w_, int(channels), cs, int(round_bf16), _stream()), "image_to_u8")
    return out


# ---- losses ---------------------------------------------------------------------------------------
def _loss_ws(device) -> torch.Tensor:
    return workspace(_lib.lib().dmvae_loss_workspace(), device, slot="loss")


def l1_mse(recon: torch.Tensor, images: torch.Tensor, w1: float = 1.0, w2: float = 0.0, need_grad: bool = True):
    """-> (out2 = [L1, MSE] device tensor, grad wrt recon of w1*L1 + w2*MSE or None)."""
    recon = _req(recon, f32, "recon")
    images = _req(images, f32, "images")
    assert recon.shape == images.shape
    ws = _loss_ws(recon.device)
    out = torch.empty(2, dtype=f32, device=recon.device)
    grad = torch.empty_like(recon) if need_grad else None
    check(_lib.lib().dmvae_l1_mse(recon.data_ptr(), images.data_ptr(), _ptr(grad), out.data_ptr(), ws.data_ptr(), ws.numel(), recon.numel(),
                                  float(w1), float(w2), _stream()), "l1_mse")
    return out, grad


def lpips_diff(f0: torch.Tensor, f1: torch.Tensor, lin_w: torch.Tensor, out: torch.Tensor, gscale: float, need_grad: bool, accumulate: bool):
    """One LPIPS level on NHWC bf16 features; accumulates the level value into out[0]; returns d/d f1 (bf16) or None."""
    if f0.dtype == f32 and parity.on():
        return parity.lpips_diff(f0, f1, _req(lin_w, f32, "lin_w"), out, gscale, need_grad, accumulate)
    f0 = _req(f0, bf16, "f0")
    f1 = _req(f1, bf16, "f1")
    n, c = f0.shape[0], f0.shape[-1]
    hw = f0.numel() // (n * c)
    ws = _loss_ws(f0.device)
    df1 = torch.empty_like(f1) if need_grad else None
    check(_lib.lib().dmvae_lpips_diff(f0.data_ptr(), f1.data_ptr(), _req(lin_w, f32, "lin_w").data_ptr(), _ptr(df1), out.data_ptr(), ws.data_ptr(),
                                      ws.numel(), n, hw, c, float(gscale), int(accumulate), _stream()), "lpips_diff")
    return df1


def lpips_diff_pool(h: torch.Tensor, b: int, lin_w: torch.Tensor, out: torch.Tensor, gscale: float, need_grad: bool, accumulate: bool):
    """lpips_diff(h[:b], h[b:]) and maxpool2x2(h) from one read of the 2b-image feature tensor h [2b, H, W, C] (H, W even) -> (d/d h[b:] or None, pooled)."""
    h = _req(h, bf16, "h")
    n2, hh, ww, c = h.shape
    assert n2 == 2 * b and hh % 2 == 0 and ww % 2 == 0, h.shape
    ws = _loss_ws(h.device)
    f0, f1 = h[:b], h[b:]
    df1 = torch.empty_like(f1) if need_grad else None
    pooled = torch.empty(n2, hh // 2, ww // 2, c, dtype=bf16, device=h.device)
    check(_lib.lib().dmvae_lpips_diff_pool(f0.data_ptr(), f1.data_ptr(), _req(lin_w, f32, "lin_w").data_ptr(), _ptr(df1), out.data_ptr(), pooled[:b].data_ptr(),
                                           pooled[b:].data_ptr(), ws.data_ptr(), ws.numel(), b, hh, ww, c, float(gscale), int(accumulate), _stream()), "lpips_diff_pool")
    return df1, pooled


def dmd_pre(x1: torch.Tensor, x0: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    x1 = _req(x1, f32, "x1")
    x0 = _req(x0, f32, "x0")
    t = _req(t, f32, "t")
    xt = torch.empty_like(x1)
    b = x1.shape[0]
    check(_lib.lib().dmvae_dmd_pre(x1.data_ptr(), x0.data_ptr(), t.data_ptr(), xt.data_ptr(), b, x1.numel() // b, _stream()), "dmd_pre")
    return xt


def dmd_post(x1, xt, t, v_teacher, v_student, v_teacher_u=None, v_student_u=None, cfg: float = 1.0, weight_factor: bool = True):
    """-> (out2 = [loss, dmd_gradient_norm], dlatents = grad/numel)."""
    for name, v in (("x1", x1), ("xt", xt), ("t", t), ("v_teacher", v_teacher), ("v_student", v_student)):
        _req(v, f32, name)
    b = x1.shape[0]
    ws = _loss_ws(x1.device)
    out = torch.empty(2, dtype=f32, device=x1.device)
    dl = torch.empty_like(x1)
    check(_lib.lib().dmvae_dmd_post(x1.data_ptr(), xt.data_ptr(), t.data_ptr(), v_teacher.data_ptr(), _ptr(v_teacher_u), v_student.data_ptr(),
                                    _ptr(v_student_u), dl.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), b, x1.numel() // b, float(cfg),
                                    int(weight_factor), _stream()), "dmd_post")
    return out, dl


def kl_mmd(z: torch.Tensor, y: Optional[torch.Tensor], w_kl: float = 1.0, w_mmd: float = 1.0, need_grad: bool = True):
    """z [G,n,32], y [G,m,32] f32 -> (kl [33] (32 per-latent + mean), mmd [G], dz or None).  y=None: KL pass only (mmd is None)."""
    z = _req(z, f32, "z")
    g, n, d = z.shape
    m = 0
    if y is not None:
        y = _req(y, f32, "y")
        m = y.shape[1]
    L = _lib.lib()
    ws = workspace(L.dmvae_kl_mmd_workspace(g, n, m), z.device, slot="klmmd")
    kl = torch.empty(d + 1, dtype=f32, device=z.device)
    mmd = torch.empty(g, dtype=f32, device=z.device) if y is not None else None
    dz = torch.empty_like(z) if need_grad else None
    check(L.dmvae_kl_mmd(z.data_ptr(), _ptr(y), kl.data_ptr(), _ptr(mmd), _ptr(dz), ws.data_ptr(), ws.numel(), g, n, m, d,
                         float(w_kl), float(w_mmd), _stream()), "kl_mmd")
    return kl, mmd, dz


def reparam_kl_fwd(moments: torch.Tensor, eps: Optional[torch.Tensor], need_z: bool = True):
    """moments [rows, 2C] (mu | logvar; f32 or bf16), eps [rows, C] f32 or None (posterior mode) -> (z [rows, C] or None, kl [C+1] f32: per-latent + mean).
    Build-defined (models/vae.py VAE(reparameterize=True)); the reference's forward is deterministic (models/vae.py:90-98)."""
    if moments.dtype not in (f32, bf16):
        raise TypeError(f"moments: expected float32 or bfloat16, got {moments.dtype}")
    moments = _req(moments, moments.dtype, "moments")
    rows, c2 = moments.shape
    c = c2 // 2
    if eps is not None:
        eps = _req(eps, f32, "eps")
        assert eps.shape == (rows, c), f"eps: expected {(rows, c)}, got {tuple(eps.shape)}"
    L = _lib.lib()
    ws = workspace(max(L.dmvae_reparam_kl_workspace(rows, c), 16), moments.device, slot="reparam")
    z = torch.empty(rows, c, dtype=moments.dtype, device=moments.device) if need_z else None
    kl = torch.empty(c + 1, dtype=f32, device=moments.device)
    check(L.dmvae_reparam_kl_fwd(moments.data_ptr(), _ptr(eps), _ptr(z), kl.data_ptr(), ws.data_ptr(), ws.numel(), rows, c, int(moments.dtype == bf16),
                                 _stream()), "reparam_kl_fwd")
    return z, kl


def reparam_kl_bwd(moments: torch.Tensor, eps: Optional[torch.Tensor], dz: Optional[torch.Tensor], g_kl: Optional[torch.Tensor], w_kl: float = 1.0):
    """-> d moments [rows, 2C]; dz [rows, C] (dtype of moments) or None, g_kl a one-element f32 device tensor (gradient of kl[C]) or None (= 1)."""
    moments = _req(moments, moments.dtype, "moments")
    rows, c2 = moments.shape
    c = c2 // 2
    if dz is not None:
        dz = _req(dz, moments.dtype, "dz")
    if g_kl is not None:
        g_kl = _req(g_kl, f32, "g_kl")
    out = torch.empty_like(moments)
    check(_lib.lib().dmvae_reparam_kl_bwd(moments.data_ptr(), _ptr(eps), _ptr(dz), _ptr(g_kl), float(w_kl), out.data_ptr(), rows, c,
                                          int(moments.dtype == bf16), _stream()), "reparam_kl_bwd")
    return out


# ---- optimiser tail -----------------------------------------------------------------------------
def grad_norm(flat_grads: torch.Tensor, max_norm: float, norm_out: Optional[torch.Tensor] = None, accumulate_prev: bool = False) -> torch.Tensor:
    g = _req(flat_grads, f32, "grads")
    ws = workspace(8192, g.device, slot="opt")
    out = norm_out if norm_out is not None else torch.zeros(3, dtype=f32, device=g.device)
    check(_lib.lib().dmvae_grad_norm(g.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), g.numel(), float(max_norm), int(accumulate_prev), _stream()),
          "grad_norm")
    return out


def adamw_ema_step(p, g, m, v, ema, norm_out, lr, beta1, beta2, eps, wd, step, ema_decay, shadow=None):
    for name, t in (("p", p), ("g", g), ("m", m), ("v", v)):
        _req(t, f32, name)
    if shadow is not None:
        _req(shadow, bf16, "shadow")
        assert shadow.numel() == p.numel()
        check(_lib.lib().dmvae_adamw_ema_step_shadow(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(ema), shadow.data_ptr(), _ptr(norm_out),
                                                     p.numel(), float(lr), float(beta1), float(beta2), float(eps), float(wd), int(step), float(ema_decay),
                                                     _stream()), "adamw_ema_step_shadow")
        return
    check(_lib.lib().dmvae_adamw_ema_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(ema), _ptr(norm_out), p.numel(), float(lr),
                                          float(beta1), float(beta2), float(eps), float(wd), int(step), float(ema_decay), _stream()), "adamw_ema_step")
