"""LightningDiT in the fp32 parity mode (dmvae_amd/parity.py): forward AND backward at north_star's 1e-4 against the reference's f32 captures
(tests/test_gpu_parity_fp32.py; fixtures of oracle/capture_golden_dit.py, oracle/capture_golden_diffusion.py).

The production route (lightningdit_fast.forward_train -> functional.DitStackFn) stores bf16 where autocast(bf16) does and can therefore only be held to the
reference through the bf16-site oracle (tests/test_gpu_dit.py).  Here activations stay f32 end to end:
  * every Linear (patch embedding, timestep MLP, adaLN modulations, qkv / proj / w12 / w3, output layer) and both attention contractions, forward, input
    gradient and weight gradient, run on the PRODUCTION MFMA GEMM kernels over exactly split bf16 operands (ops.gemm_nt / gemm_tn / conv2d_nhwc_wgrad dispatch
    to dmvae_amd.parity: three bf16 terms per f32 value, six partial products along the reduction of one launch);
  * RMSNorm + modulate, the gated residual, QK-norm + RoPE, SwiGLU, SiLU, softmax -- and the backward of each -- on the f32 kernels of csrc/parity_dit.hip /
    parity.hip, their per-sample and per-parameter sums by a row-ordered f64 column sum;
  * what is left to torch is layout (patchify / unpatchify, head split), the sinusoidal timestep features, the embedding lookup and additions of f32 tensors --
    as on the production route.
One autograd Function per step of lightningdit.py:241-250, in the reference's order; the call sequence of a block is the one functional.DitBlockFn schedules.
A verification mode: ~6x the matrix work, no fusion.

Reference: diffusion/lightningdit/lightningdit.py:34-93 (Attention), :96-171 (embedders), :173-252 (block), :255-274 (final layer), :393-421 (forward);
rms_norm.py:52-76; swiglu_ffn.py:15-36; pos_embed.py:37-41,135."""
from __future__ import annotations

import torch

from .. import _lib, ops, parity
from .._lib import check
from ..functional import _c, _dst, packed

f32 = torch.float32


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _pad_cols(x: torch.Tensor, mult: int) -> torch.Tensor:
    pad = (-x.shape[-1]) % mult
    return x if pad == 0 else torch.nn.functional.pad(x, (0, pad))


def colsum_groups(x: torch.Tensor, groups: int, out: torch.Tensor = None) -> torch.Tensor:
    """[groups * rows, C] f32 -> [groups, C]: row-ordered f64 sums (csrc/parity_dit.hip)."""
    x = _c(x)
    c = x.shape[-1]
    rows = x.numel() // (c * groups)
    res = out if out is not None else torch.empty(groups, c, dtype=f32, device=x.device)
    check(_lib.lib().dmvae_colsum_groups_f32(x.data_ptr(), res.data_ptr(), groups, rows, c, 0, _stream()), "colsum_groups_f32")
    return res


class PLinearFn(torch.autograd.Function):
    """y = x W^T + b, f32: forward, input gradient and weight gradient on the split-operand MFMA GEMMs.  x [..., K]; K and N need not be multiples of the kernels'
    granules (zero-padded here: reduced fixture models)."""

    @staticmethod
    def forward(ctx, x, w, b):
        k, n = x.shape[-1], w.shape[0]
        x2 = _c(x.reshape(-1, k).float())
        wf = w.detach().float().reshape(n, -1)
        kp, np_ = (-k) % 16, (-n) % 8
        if kp or np_:
            xg, wg = torch.nn.functional.pad(x2, (0, kp)), torch.nn.functional.pad(wf, (0, kp, 0, np_))
            bg = None if b is None else torch.nn.functional.pad(b.detach().float(), (0, np_))
            y = ops.gemm_nt(xg, parity.split_channels(wg.contiguous(), parity.W_SIDE), bg)[:, :n].contiguous()
        else:
            ws = packed(w).view(n, -1) if (isinstance(w, torch.nn.Parameter) and w.dim() == 2) else parity.split_channels(wf.contiguous(), parity.W_SIDE)
            y = ops.gemm_nt(x2, ws, None if b is None else b.detach().float())
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        ctx.bias = b
        ctx.shape = x.shape
        return y.view(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        n, k = w.shape[0], x2.shape[1]
        rows = x2.shape[0]
        dy2 = _c(dy.reshape(rows, n).float())
        wf = w.detach().float().reshape(n, -1)
        # d x = dy W: the NT kernel against W^T [K, N]; reduction over N (granule 16), output columns K (granule 8)
        dx = None
        if ctx.needs_input_grad[0]:
            wt = _pad_cols(torch.nn.functional.pad(wf.t().contiguous(), (0, 0, 0, (-k) % 8)), 16)
            dx = ops.gemm_nt(_pad_cols(dy2, 16), parity.split_channels(wt.contiguous(), parity.W_SIDE))[:, :k].contiguous().view(ctx.shape)
        # d W = dy^T x: the TN kernel (reduction over the rows); M = N and N = K padded to its granule of 8
        a, bm = _pad_cols(dy2, 8), _pad_cols(x2, 8)
        dw = ops.gemm_tn(a.view(1, rows, -1), bm.view(1, rows, -1), out_f32=True)[0, :n, :k].contiguous().view(w.shape)
        db = colsum_groups(dy2, 1).view(n) if ctx.has_bias else None
        dst_w, dst_b = _dst(w), (_dst(ctx.bias) if ctx.has_bias else None)
        if dst_w is not None:
            dst_w.copy_(dw)
            dw = dst_w
        if dst_b is not None:
            dst_b.copy_(db)
            db = dst_b
        return dx, dw, db


class PSiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x.float())
        ctx.save_for_backward(x)
        return parity.eltwise(1, x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return parity.eltwise(2, x, _c(dy.float()))


class PRmsModFn(torch.autograd.Function):
    """RMSNorm(h) * w * (1 + scale) + shift with the adaLN chunks of mod [B, L] at the given offsets (None: absent); h [B, N, C] f32."""

    @staticmethod
    def forward(ctx, h, w, mod, shift_off, scale_off, eps):
        h = _c(h.float())
        b, n, c = h.shape
        wf = w.detach().float().contiguous()
        y = torch.empty_like(h)
        rstd = torch.empty(b * n, dtype=f32, device=h.device)
        mod = None if mod is None else _c(mod.float())
        ld = 0 if mod is None else mod.shape[1]
        ptr = lambda off: None if (mod is None or off is None) else mod.data_ptr() + 4 * off
        check(_lib.lib().dmvae_rms_modulate_fwd_f32(h.data_ptr(), wf.data_ptr(), ptr(shift_off), ptr(scale_off), y.data_ptr(), rstd.data_ptr(), b * n, c, n, ld,
                                                    float(eps), _stream()), "rms_modulate_fwd_f32")
        ctx.save_for_backward(h, w, mod, rstd)
        ctx.cfg = (shift_off, scale_off)
        return y

    @staticmethod
    def backward(ctx, dy):
        h, w, mod, rstd = ctx.saved_tensors
        shift_off, scale_off = ctx.cfg
        b, n, c = h.shape
        dy = _c(dy.float())
        wf = w.detach().float().contiguous()
        dx, gw = torch.empty_like(h), torch.empty_like(h)
        has_scale = mod is not None and scale_off is not None
        gs = torch.empty_like(h) if has_scale else None
        ld = 0 if mod is None else mod.shape[1]
        check(_lib.lib().dmvae_rms_modulate_bwd_f32(dy.data_ptr(), h.data_ptr(), wf.data_ptr(), (mod.data_ptr() + 4 * scale_off) if has_scale else None, rstd.data_ptr(),
                                                    dx.data_ptr(), gw.data_ptr(), None if gs is None else gs.data_ptr(), b * n, c, n, ld, _stream()),
              "rms_modulate_bwd_f32")
        dw = colsum_groups(gw, 1).view(w.shape)
        dst = _dst(w)
        if dst is not None:
            dst.copy_(dw)
            dw = dst
        dmod = None
        if mod is not None:
            dmod = torch.zeros_like(mod)
            if has_scale:
                dmod[:, scale_off:scale_off + c] = colsum_groups(gs, b)
            if shift_off is not None:
                dmod[:, shift_off:shift_off + c] = colsum_groups(dy, b)
        return dx, dw, dmod, None, None, None


class PGateResFn(torch.autograd.Function):
    """h + gate * y with gate = mod[:, off : off + C] per sample (lightningdit.py:245,249)."""

    @staticmethod
    def forward(ctx, h, y, mod, gate_off):
        h, y, mod = _c(h.float()), _c(y.float()), _c(mod.float())
        b, n, c = h.shape
        out = torch.empty_like(h)
        check(_lib.lib().dmvae_bcast_rows_f32(0, h.data_ptr(), y.data_ptr(), mod.data_ptr() + 4 * gate_off, out.data_ptr(), b * n, c, n, mod.shape[1], _stream()),
              "bcast_rows_f32")
        ctx.save_for_backward(y, mod)
        ctx.off = gate_off
        return out

    @staticmethod
    def backward(ctx, dout):
        y, mod = ctx.saved_tensors
        b, n, c = y.shape
        dout = _c(dout.float())
        L = _lib.lib()
        dy, prod = torch.empty_like(y), torch.empty_like(y)
        check(L.dmvae_bcast_rows_f32(1, dout.data_ptr(), None, mod.data_ptr() + 4 * ctx.off, dy.data_ptr(), b * n, c, n, mod.shape[1], _stream()), "bcast_rows_f32")
        check(L.dmvae_bcast_rows_f32(2, dout.data_ptr(), y.data_ptr(), None, prod.data_ptr(), b * n, c, n, 0, _stream()), "bcast_rows_f32")
        dmod = torch.zeros_like(mod)
        dmod[:, ctx.off:ctx.off + c] = colsum_groups(prod, b)
        return dout, dy, dmod, None


class PSwigluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x12):
        x12 = _c(x12.float())
        hid = x12.shape[-1] // 2
        rows = x12.numel() // (2 * hid)
        g = torch.empty(*x12.shape[:-1], hid, dtype=f32, device=x12.device)
        check(_lib.lib().dmvae_swiglu_fwd_f32(x12.data_ptr(), g.data_ptr(), rows, hid, _stream()), "swiglu_fwd_f32")
        ctx.save_for_backward(x12)
        return g

    @staticmethod
    def backward(ctx, dg):
        (x12,) = ctx.saved_tensors
        hid = x12.shape[-1] // 2
        dx = torch.empty_like(x12)
        check(_lib.lib().dmvae_swiglu_bwd_f32(_c(dg.float()).data_ptr(), x12.data_ptr(), dx.data_ptr(), x12.numel() // (2 * hid), hid, _stream()), "swiglu_bwd_f32")
        return dx


class PAttentionFn(torch.autograd.Function):
    """Attention of one block from its qkv Linear's output: QK RMSNorm + RoPE (f32 kernel), S = q k^T, softmax(S / sqrt(d)), O = P v on the split-operand GEMMs;
    backward: dV = P^T dO, dP = dO v^T, dS (softmax backward), dq = dS k, dk = dS^T q, then the QK-norm / RoPE backward.  qkv [B, N, 3C] -> o [B, N, C]."""

    @staticmethod
    def forward(ctx, qkv, wq, wk, cos, sin, heads, eps):
        qkv = _c(qkv.float())
        b, n, c3 = qkv.shape
        c = c3 // 3
        d = c // heads
        dp = (d + 15) // 16 * 16
        wqf, wkf = wq.detach().float().contiguous(), wk.detach().float().contiguous()
        cosf, sinf = _c(cos.float()), _c(sin.float())
        q = torch.empty(b * heads, n, dp, dtype=f32, device=qkv.device)
        k, v = torch.empty_like(q), torch.empty_like(q)
        rstd = torch.empty(2, b * n * heads, dtype=f32, device=qkv.device)
        check(_lib.lib().dmvae_qknorm_rope_fwd_f32(qkv.data_ptr(), wqf.data_ptr(), wkf.data_ptr(), cosf.data_ptr(), sinf.data_ptr(), q.data_ptr(), k.data_ptr(),
                                                   v.data_ptr(), rstd.data_ptr(), b, n, heads, d, dp, float(eps), _stream()), "qknorm_rope_fwd_f32")
        assert n % 16 == 0, "parity attention: tokens must be a multiple of 16 (the P.V reduction granule)"
        s = ops.gemm_nt(q, k, out_f32=True)                                       # [B*H, N, N]
        p = parity.softmax_rows(s, d ** -0.5)
        o = ops.gemm_nt(p, parity.transpose_last2(v), out_f32=True)               # [B*H, N, dp]
        ctx.save_for_backward(qkv, wq, wk, cosf, sinf, q, k, v, p, rstd)
        ctx.cfg = (b, n, heads, d, dp)
        return o[..., :d].reshape(b, heads, n, d).permute(0, 2, 1, 3).reshape(b, n, c)

    @staticmethod
    def backward(ctx, do):
        qkv, wq, wk, cosf, sinf, q, k, v, p, rstd = ctx.saved_tensors
        b, n, heads, d, dp = ctx.cfg
        do = do.float().reshape(b, n, heads, d).permute(0, 2, 1, 3).reshape(b * heads, n, d)
        do = _c(torch.nn.functional.pad(do, (0, dp - d)) if dp != d else do)
        dv = ops.gemm_nt(parity.transpose_last2(p), parity.transpose_last2(do), out_f32=True)      # P^T dO: [N, N] x [dp, N]^T -> [N, dp]
        dpm = ops.gemm_nt(do, v, out_f32=True)                                                      # dO v^T: [N, N]
        ds = parity.softmax_rows_bwd(dpm, p, d ** -0.5)
        dq = ops.gemm_nt(ds, parity.transpose_last2(k), out_f32=True)                               # dS k: [N, dp]
        dk = ops.gemm_nt(parity.transpose_last2(ds), parity.transpose_last2(q), out_f32=True)       # dS^T q
        dqkv = torch.empty_like(qkv)
        gwq = torch.empty(b * n * heads, d, dtype=f32, device=qkv.device)
        gwk = torch.empty_like(gwq)
        wqf, wkf = wq.detach().float().contiguous(), wk.detach().float().contiguous()
        check(_lib.lib().dmvae_qknorm_rope_bwd_f32(_c(dq).data_ptr(), _c(dk).data_ptr(), _c(dv).data_ptr(), qkv.data_ptr(), wqf.data_ptr(), wkf.data_ptr(),
                                                   cosf.data_ptr(), sinf.data_ptr(), rstd.data_ptr(), dqkv.data_ptr(), gwq.data_ptr(), gwk.data_ptr(), b, n, heads, d,
                                                   dp, _stream()), "qknorm_rope_bwd_f32")
        dwq, dwk = colsum_groups(gwq, 1).view(wq.shape), colsum_groups(gwk, 1).view(wk.shape)
        for w_, g_ in ((wq, dwq), (wk, dwk)):
            dst = _dst(w_)
            if dst is not None:
                dst.copy_(g_)
        return dqkv, (_dst(wq) if _dst(wq) is not None else dwq), (_dst(wk) if _dst(wk) is not None else dwk), None, None, None, None


def _block(blk, h, sc, rope, heads):
    """lightningdit.py:241-250 in its order: adaLN chunks -> (norm1, modulate) -> attention -> gated residual -> (norm2, modulate) -> SwiGLU MLP -> gated residual."""
    c = h.shape[-1]
    lin = blk.adaLN_modulation[1]
    mod = PLinearFn.apply(sc, lin.weight, lin.bias)                               # [B, 6C]: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp
    a1 = PRmsModFn.apply(h, blk.norm1.weight, mod, 0, c, blk.norm1.eps)
    qkv = PLinearFn.apply(a1, blk.attn.qkv.weight, blk.attn.qkv.bias)
    o = PAttentionFn.apply(qkv, blk.attn.q_norm.weight, blk.attn.k_norm.weight, rope.freqs_cos, rope.freqs_sin, heads, blk.attn.q_norm.eps)
    h = PGateResFn.apply(h, PLinearFn.apply(o, blk.attn.proj.weight, blk.attn.proj.bias), mod, 2 * c)
    a2 = PRmsModFn.apply(h, blk.norm2.weight, mod, 3 * c, 4 * c, blk.norm2.eps)
    g = PSwigluFn.apply(PLinearFn.apply(a2, blk.mlp.w12.weight, blk.mlp.w12.bias))
    return PGateResFn.apply(h, PLinearFn.apply(g, blk.mlp.w3.weight, blk.mlp.w3.bias), mod, 5 * c)


def structurally_supported(model) -> bool:
    from .lightningdit_fast import structurally_supported as fast_ok
    return fast_ok(model)


def forward_parity(model, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """LightningDiT.forward (lightningdit.py:393-421), f32, differentiable w.r.t. x and every parameter.  Label dropout as in the module."""
    b, cin, hh, ww = x.shape
    ps, c, heads = model.patch_size, model.hidden_size, model.num_heads
    w = model.x_embedder.proj.weight
    patches = x.float().reshape(b, cin, hh // ps, ps, ww // ps, ps).permute(0, 2, 4, 1, 3, 5).reshape(b, -1, cin * ps * ps)
    h = PLinearFn.apply(patches, w.view(w.shape[0], -1), model.x_embedder.proj.bias) + model.pos_embed.float()
    te = model.t_embedder
    emb = te.timestep_embedding(t, te.frequency_embedding_size).float()
    temb = PLinearFn.apply(PSiluFn.apply(PLinearFn.apply(emb, te.mlp[0].weight, te.mlp[0].bias)), te.mlp[2].weight, te.mlp[2].bias)
    cvec = temb + model.y_embedder(y, model.training).float()
    sc = PSiluFn.apply(cvec)                                                       # adaLN_modulation[0] of every block and of the final layer
    for blk in model.blocks:
        h = _block(blk, h, sc, model.feat_rope, heads)
    fl = model.final_layer
    modf = PLinearFn.apply(sc, fl.adaLN_modulation[1].weight, fl.adaLN_modulation[1].bias)      # [B, 2C]: shift, scale
    a = PRmsModFn.apply(h, fl.norm_final.weight, modf, 0, c, fl.norm_final.eps)
    out = model.unpatchify(PLinearFn.apply(a, fl.linear.weight, fl.linear.bias))
    if model.learn_sigma:
        out, _ = out.chunk(2, dim=1)
    return out
