"""Inference-only forward of the frozen DINOv2 ViT encoder (models/vae.py:52-53 in the tokenizer stage, where the encoder is
frozen and runs under no_grad, train_tokenizer.py:295-297): same arithmetic as `vit.DinoV2ViT.forward_features` under
autocast(bf16), with the elementwise chain on the HIP kernels of csrc/vit.hip -- LayerNorm straight to bf16, LayerScale +
residual add fused on the f32 residual stream, multi-head attention as one fused MFMA kernel -- and the four Linear GEMMs
per block (and the patch embedding) on the hand-written GEMM of csrc/gemm_pp.hip through the bf16 weight shadow, fc1 with its
GELU in the epilogue (`functional.linear`; tools/bench_gemm.py has the per-call comparison with the vendor library it replaced).
SURVEY.md 8(f) rank 3 ("next")."""
import torch
import torch.nn.functional as F

from .. import ops


def patch_embed_gemm(vit, x: torch.Tensor) -> torch.Tensor:
    """PatchEmbed (Conv2d(kernel = stride = patch), dino_layers/patch_embed.py) as what it is -- one GEMM over non-overlapping patches:
    [B,3,H,W] -> [B, N, 3*p*p] (channel, row, column order = the conv weight's) @ W^T + b.  Differentiable (views + `functional.LinearFn`); avoids
    MIOpen's convolution search (minutes on the first backward call) and its atomically-accumulated, run-to-run different weight gradient."""
    w = vit.patch_embed.proj.weight
    b_, c, hh, ww = x.shape
    p = w.shape[-1]
    assert hh % p == 0 and ww % p == 0, "image size must be a multiple of the patch size"
    patches = x.view(b_, c, hh // p, p, ww // p, p).permute(0, 2, 4, 1, 3, 5).reshape(b_, (hh // p) * (ww // p), c * p * p)
    if not torch.is_grad_enabled():          # frozen use: the cached bf16 copies (functional._bf), not a conversion per call, on this build's GEMM
        from ..functional import linear
        return linear(patches.to(torch.bfloat16), _w(w).view(w.shape[0], -1), _w(vit.patch_embed.proj.bias))
    from ..functional import LinearFn        # with gradients: the same GEMM forward, the weight gradient from the split-K kernel
    return LinearFn.apply(patches, w.view(w.shape[0], -1), vit.patch_embed.proj.bias)


def _w(p: torch.Tensor) -> torch.Tensor:
    """A Linear operand in bf16: the parameter itself when the module is a bf16 shadow (train.frozen_bf16_shadow), else its cached bf16 copy."""
    if p.dtype == torch.bfloat16:
        return p
    from ..functional import _bf
    return _bf(p)


@torch.no_grad()
def frozen_forward_features(vit, x: torch.Tensor) -> torch.Tensor:
    """vit: a DinoV2ViT -- either a bf16 shadow (train.frozen_bf16_shadow: Linear / Conv2d weights already bf16) or the f32 module itself, whose
    Linear weights are then served as cached bf16 copies (`functional._bf`, refreshed when a parameter changes); x: [B,3,H,W] f32, already normalised.
    Returns the final-norm tokens [B, 1+N, C] in bf16 (what the bottleneck MLP consumes)."""
    from ..functional import linear
    t = patch_embed_gemm(vit, x).float()
    t = torch.cat([vit.cls_token.expand(t.shape[0], -1, -1).float(), t], dim=1) + vit.pos_embed.float()
    t = t.contiguous()                                   # f32 residual stream [B, S, C]
    b, s, c = t.shape
    hn = ops.layernorm_bf16(t, vit.blocks[0].norm1.weight, vit.blocks[0].norm1.bias, vit.blocks[0].norm1.eps)
    nblk = len(vit.blocks)
    for i, blk in enumerate(vit.blocks):
        nh = blk.attn.num_heads
        hd = c // nh
        # the PARAMETERS go to `linear`, not their bf16 copies: it serves a frozen one from the K-tile-major pack and one that an optimiser of this build
        # owns (DMDTrainer's encoder between its VAE turns: vae.encode under no_grad) from the live bf16 shadow that optimiser maintains
        qkv = linear(hn, blk.attn.qkv.weight, blk.attn.qkv.bias)                       # [b, s, 3*c] = [b, s, 3, heads, hd]
        if hd == 64 and s <= 288:
            o = ops.attention_qkv(qkv, nh, hd ** -0.5)                             # fused: nothing of size s x s reaches HBM
        else:
            # head dims / sequence lengths the fused kernel does not take (hip_path_supported keeps the reference's shapes off this branch): library batched
            # GEMMs, only behind the explicit opt-in
            from .._stock import require_opt_in
            require_opt_in("vit_fast.forward_features (attention)", f"head dim {hd}, {s} tokens: the fused attention kernel takes head dim 64 and <= 288 tokens")
            qkv = qkv.reshape(b, s, 3, nh, hd).permute(2, 0, 3, 1, 4)
            att = ops.softmax_rows_bf16(qkv[0] @ qkv[1].transpose(-2, -1), hd ** -0.5)   # scale, f32 softmax and the casts in one pass
            o = (att @ qkv[2]).transpose(1, 2).reshape(b, s, c)
        o = linear(o, blk.attn.proj.weight, blk.attn.proj.bias)
        # every LayerScale + residual add is followed by a LayerNorm (this block's norm2, the next block's norm1, the final norm): one pass over the stream
        hn = ops.scale_residual_layernorm_(t, o.contiguous(), blk.ls1.gamma, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        h = linear(hn, blk.mlp.fc1.weight, blk.mlp.fc1.bias, act=ops.ACT_GELU)   # GELU in the GEMM's epilogue (bit-identical to the two kernels)
        o = linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias)
        nxt = vit.blocks[i + 1].norm1 if i + 1 < nblk else vit.norm
        hn = ops.scale_residual_layernorm_(t, o, blk.ls2.gamma, nxt.weight, nxt.bias, nxt.eps)
    return hn


def hip_path_supported(vit, seq_len: int) -> bool:
    """Shapes the encoder kernels cover: width 256 / 512 / 768 / 1024 / 1280 / 1536 (LayerNorm kernels; ViT-B = 768 and ViT-L = 1024 are the
    reference's two sizes, models/vae.py:41-48), head dim 64, at most 288 tokens (fused attention)."""
    c = vit.embed_dim
    nh = vit.blocks[0].attn.num_heads
    return c in (256, 512, 768, 1024, 1280, 1536) and c // nh == 64 and seq_len <= 288


NOGRAD_FUSED = True      # trainable_forward_features under no_grad takes the frozen route's fused kernels (False: the block Functions' forward; tests compare)


def trainable_forward_features(vit, x: torch.Tensor) -> torch.Tensor:
    """`DinoV2ViT.forward_features` with gradients, for the stages where the encoder trains (train_dmd.py:349,519): patch embedding
    (one GEMM over patches), class token and position embedding through stock autograd, every transformer block as one `VitBlockFn` on the f32
    residual stream, the final LayerNorm as `LayerNormBf16Fn`.  Same arithmetic as the module under autocast(bf16)."""
    from ..functional import LayerNormBf16Fn, VitBlockFn
    if not torch.is_grad_enabled() and NOGRAD_FUSED:
        # graph-free call (the DMD stage's student-only steps, train_dmd.py:520-523: four of five steps): the frozen route's fused launches -- GELU in the fc1
        # GEMM's epilogue, LayerScale + residual + the next LayerNorm as one pass -- on the live parameters (`functional.linear` reads the bf16 shadow their
        # optimiser maintains); the same bits as the block Functions' forward (tests/test_gpu_vit_train.py)
        return frozen_forward_features(vit, x)
    t = patch_embed_gemm(vit, x)
    t = torch.cat([vit.cls_token.expand(t.shape[0], -1, -1).float(), t.float()], dim=1) + vit.pos_embed.float()
    t = t.contiguous()
    for blk in vit.blocks:
        t = VitBlockFn.apply(t, blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias, blk.attn.proj.weight, blk.attn.proj.bias,
                             blk.ls1.gamma, blk.norm2.weight, blk.norm2.bias, blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight,
                             blk.mlp.fc2.bias, blk.ls2.gamma, blk.attn.num_heads, blk.norm1.eps)
    return LayerNormBf16Fn.apply(t, vit.norm.weight, vit.norm.bias, vit.norm.eps)


@torch.no_grad()
def parity_forward_features(vit, x: torch.Tensor) -> torch.Tensor:
    """The frozen encoder in the fp32 parity mode (dmvae_amd/parity.py): f32 residual stream and f32 activations throughout, every Linear (patch
    embedding included) and both attention contractions on the MFMA GEMM kernel over exactly-split bf16 operands, LayerNorm / GELU / LayerScale /
    softmax on the f32 kernels of csrc/parity.hip.  Any width / head count (the reduced ViT stand-ins of the golden fixtures included): the
    production route's shape limits come from its fused kernels, which this mode does not use.  Same block algebra as
    `vit.DinoV2ViT.forward_features_stock` (reference: models/dinov2.py + dino_layers, reached through models/vae.py:47-53)."""
    from .. import parity
    from ..functional import packed

    def linear(a2, lin):                       # [rows, in] f32 -> [rows, out] f32
        w = lin.weight
        return ops.gemm_nt(a2, packed(w, frozen=not w.requires_grad).view(w.shape[0], -1), lin.bias.detach().float())

    w = vit.patch_embed.proj.weight
    b_, c_in, hh, ww = x.shape
    p = w.shape[-1]
    patches = x.float().view(b_, c_in, hh // p, p, ww // p, p).permute(0, 2, 4, 1, 3, 5).reshape(b_ * (hh // p) * (ww // p), c_in * p * p).contiguous()
    kpad = (-patches.shape[1]) % 16                      # the split operand's reduction length 6*K must be a multiple of 32
    wp = w.detach().float().reshape(w.shape[0], -1)
    if kpad:
        patches = torch.nn.functional.pad(patches, (0, kpad))
        wp = torch.nn.functional.pad(wp, (0, kpad))
    t = ops.gemm_nt(patches, parity.pack_conv_weight(wp.contiguous()).view(w.shape[0], -1), vit.patch_embed.proj.bias.detach().float())
    t = t.view(b_, -1, w.shape[0])
    t = (torch.cat([vit.cls_token.expand(b_, -1, -1).float(), t], dim=1) + vit.pos_embed.float()).contiguous()
    b, s, c = t.shape
    sp = (s + 15) // 16 * 16                             # keys padded (and masked) so that the P.V reduction length 6*sp is a multiple of 32
    for blk in vit.blocks:
        nh = blk.attn.num_heads
        hd = c // nh
        hn = parity.layernorm(t, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        qkv = linear(hn.view(b * s, c), blk.attn.qkv).view(b, s, 3, nh, hd)
        buf = torch.zeros(3, b * nh, sp, hd, dtype=torch.float32, device=t.device)
        buf[:, :, :s] = qkv.permute(2, 0, 3, 1, 4).reshape(3, b * nh, s, hd)
        sc = ops.gemm_nt(buf[0], buf[1], out_f32=True)                              # [b*nh, sp, sp]
        if sp != s:
            sc[:, :, s:] = float("-inf")
        pr = ops.softmax_rows(sc, hd ** -0.5)
        o = ops.gemm_nt(pr, ops.transpose_last2(buf[2]), out_f32=True)[:, :s]        # [b*nh, s, hd]
        o = o.reshape(b, nh, s, hd).permute(0, 2, 1, 3).reshape(b * s, c).contiguous()
        o = linear(o, blk.attn.proj).view(b, s, c)
        t = parity.eltwise(6, t, o, g=blk.ls1.gamma.detach().float(), out=t)
        hn = parity.layernorm(t, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        h = parity.eltwise(4, linear(hn.view(b * s, c), blk.mlp.fc1))
        o = linear(h, blk.mlp.fc2).view(b, s, c)
        t = parity.eltwise(6, t, o, g=blk.ls2.gamma.detach().float(), out=t)
    return parity.layernorm(t, vit.norm.weight, vit.norm.bias, vit.norm.eps)
