"""The trainable DINOv2 / timm ViT encoder in the fp32 parity mode: forward AND backward at 1e-4 against the reference's f32 capture (tests/golden/vit_w256.npz from
oracle/capture_golden_vit.py; tests/test_gpu_parity_fp32.py).  The DMD stage trains the encoder (train_dmd.py:349,518-520); the production route
(models/vit_fast.trainable_forward_features -> functional.VitBlockFn) stores bf16 where autocast does and is held to the reference through the bf16-site oracle
(tests/test_gpu_vit_pin.py).  Here, as in models/lightningdit_parity.py: f32 activations; every Linear (patch embedding included) and both attention contractions
-- forward, input gradient, weight gradient -- on the production MFMA GEMMs over exactly split bf16 operands; LayerNorm, GELU, LayerScale, softmax and their
backward on the f32 kernels of csrc/parity.hip / parity_dit.hip; layout, the class / position embeddings and f32 additions left to torch.

Reference: models/dinov2.py + dino_layers/{block.py:89-115, attention.py:56-69, layer_scale.py:15-26, mlp.py:34-39, patch_embed.py:68-81} -- the in-repo restatement
of the timm model reached through models/vae.py:47-53."""
from __future__ import annotations

import torch

from .. import _lib, ops, parity
from .._lib import check
from ..functional import _c, _dst
from .lightningdit_parity import PLinearFn, _stream, colsum_groups

f32 = torch.float32


def _give(param, grad):
    """`grad` into the parameter's flat-buffer slot when it has one (optim.FlatParams direct gradients), else returned as it is."""
    dst = _dst(param)
    if dst is None:
        return grad
    dst.copy_(grad.view(dst.shape))
    return dst


class PLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = _c(x.float())
        ctx.save_for_backward(x, w, b)
        ctx.eps = eps
        return parity.layernorm(x, w.detach().float().contiguous(), b.detach().float().contiguous(), eps)

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        c = x.shape[-1]
        rows = x.numel() // c
        dy = _c(dy.float())
        dx, gw = torch.empty_like(x), torch.empty_like(x)
        check(_lib.lib().dmvae_layernorm_bwd_full_f32(dy.data_ptr(), x.data_ptr(), w.detach().float().contiguous().data_ptr(), dx.data_ptr(), gw.data_ptr(), rows, c,
                                                      float(ctx.eps), _stream()), "layernorm_bwd_full_f32")
        return dx, _give(w, colsum_groups(gw, 1).view(w.shape)), _give(b, colsum_groups(dy, 1).view(b.shape)), None


class PLayerScaleResFn(torch.autograd.Function):
    """t + o * gamma (layer_scale.py:18-27 inside block.py:89-115's residual)."""

    @staticmethod
    def forward(ctx, t, o, gamma):
        t, o = _c(t.float()), _c(o.float())
        ctx.save_for_backward(o, gamma)
        return parity.eltwise(6, t, o, g=gamma.detach().float().contiguous())

    @staticmethod
    def backward(ctx, dout):
        o, gamma = ctx.saved_tensors
        c = o.shape[-1]
        rows = o.numel() // c
        dout = _c(dout.float())
        gf = gamma.detach().float().contiguous().view(1, c)
        L = _lib.lib()
        do, prod = torch.empty_like(o), torch.empty_like(o)
        check(L.dmvae_bcast_rows_f32(1, dout.data_ptr(), None, gf.data_ptr(), do.data_ptr(), rows, c, rows, c, _stream()), "bcast_rows_f32")      # one "sample": gamma for every row
        check(L.dmvae_bcast_rows_f32(2, dout.data_ptr(), o.data_ptr(), None, prod.data_ptr(), rows, c, rows, 0, _stream()), "bcast_rows_f32")
        return dout, do, _give(gamma, colsum_groups(prod, 1).view(gamma.shape))


class PGeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x.float())
        ctx.save_for_backward(x)
        return parity.eltwise(4, x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return parity.eltwise(5, x, _c(dy.float()))


class PAttentionPlainFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v from the qkv Linear's output [B, S, 3C] (attention.py:56-69: no QK-norm, no RoPE).  The keys are padded to a multiple of 16 --
    the granule of the P.V reduction over split operands -- with -inf scores, so the padded columns of P are exactly zero, forward and backward."""

    @staticmethod
    def forward(ctx, qkv, heads):
        qkv = _c(qkv.float())
        b, s, c3 = qkv.shape
        c = c3 // 3
        d = c // heads
        assert d % 16 == 0, "parity attention: head dim must be a multiple of 16"
        sp = (s + 15) // 16 * 16
        buf = torch.zeros(3, b * heads, sp, d, dtype=f32, device=qkv.device)
        buf[:, :, :s] = qkv.view(b, s, 3, heads, d).permute(2, 0, 3, 1, 4).reshape(3, b * heads, s, d)
        sc = ops.gemm_nt(buf[0], buf[1], out_f32=True)                            # [B*H, sp, sp]
        if sp != s:
            sc[:, :, s:] = float("-inf")
        p = parity.softmax_rows(sc, d ** -0.5)
        o = ops.gemm_nt(p, parity.transpose_last2(buf[2]), out_f32=True)           # [B*H, sp, d]
        ctx.save_for_backward(buf, p)
        ctx.cfg = (b, s, sp, heads, d)
        return o[:, :s].reshape(b, heads, s, d).permute(0, 2, 1, 3).reshape(b, s, c)

    @staticmethod
    def backward(ctx, do):
        buf, p = ctx.saved_tensors
        b, s, sp, heads, d = ctx.cfg
        q, k, v = buf[0], buf[1], buf[2]
        dof = torch.zeros(b * heads, sp, d, dtype=f32, device=p.device)
        dof[:, :s] = do.float().reshape(b, s, heads, d).permute(0, 2, 1, 3).reshape(b * heads, s, d)
        dv = ops.gemm_nt(parity.transpose_last2(p), parity.transpose_last2(dof), out_f32=True)        # P^T dO  (rows >= s of P are the padded queries: dO = 0 there)
        dp = ops.gemm_nt(dof, v, out_f32=True)                                                         # dO v^T
        ds = parity.softmax_rows_bwd(dp, p, d ** -0.5)                                                 # zero where P is zero (the padded keys)
        dq = ops.gemm_nt(ds, parity.transpose_last2(k), out_f32=True)
        dk = ops.gemm_nt(parity.transpose_last2(ds), parity.transpose_last2(q), out_f32=True)
        dqkv = torch.stack([dq[:, :s], dk[:, :s], dv[:, :s]]).reshape(3, b, heads, s, d).permute(1, 3, 0, 2, 4).reshape(b, s, 3 * heads * d)
        return dqkv.contiguous(), None


def forward_features_parity(vit, x: torch.Tensor) -> torch.Tensor:
    """DinoV2ViT.forward_features, f32, differentiable w.r.t. the image and every parameter: [B, 1 + N, C] (class + patch tokens after the final norm)."""
    w = vit.patch_embed.proj.weight
    b_, c_in, hh, ww = x.shape
    p = w.shape[-1]
    patches = x.float().reshape(b_, c_in, hh // p, p, ww // p, p).permute(0, 2, 4, 1, 3, 5).reshape(b_, (hh // p) * (ww // p), c_in * p * p)
    t = PLinearFn.apply(patches, w.view(w.shape[0], -1), vit.patch_embed.proj.bias)
    t = torch.cat([vit.cls_token.expand(b_, -1, -1).float(), t], dim=1) + vit.pos_embed.float()
    for blk in vit.blocks:
        hn = PLayerNormFn.apply(t, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        o = PAttentionPlainFn.apply(PLinearFn.apply(hn, blk.attn.qkv.weight, blk.attn.qkv.bias), blk.attn.num_heads)
        t = PLayerScaleResFn.apply(t, PLinearFn.apply(o, blk.attn.proj.weight, blk.attn.proj.bias), blk.ls1.gamma)
        hn = PLayerNormFn.apply(t, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        o = PLinearFn.apply(PGeluFn.apply(PLinearFn.apply(hn, blk.mlp.fc1.weight, blk.mlp.fc1.bias)), blk.mlp.fc2.weight, blk.mlp.fc2.bias)
        t = PLayerScaleResFn.apply(t, o, blk.ls2.gamma)
    return PLayerNormFn.apply(t, vit.norm.weight, vit.norm.bias, vit.norm.eps)
