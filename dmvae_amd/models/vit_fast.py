"""Inference-only forward of the frozen DINOv2 ViT encoder (models/vae.py:52-53 in the tokenizer stage, where the encoder is
frozen and runs under no_grad, train_tokenizer.py:295-297): same arithmetic as `vit.DinoV2ViT.forward_features` under
autocast(bf16), with the elementwise chain on the HIP kernels of csrc/vit.hip -- LayerNorm straight to bf16, LayerScale +
residual add fused on the f32 residual stream, multi-head attention as one fused MFMA kernel -- and the four Linear GEMMs
per block through the bf16 weight shadow (hipBLASLt: measured faster than conv_pp on these M = 8224 shapes,
tools/bench_gemm.py).  SURVEY.md 8(f) rank 3 ("next") -- forward / frozen case only."""
import torch
import torch.nn.functional as F

from .. import ops


@torch.no_grad()
def frozen_forward_features(vit, x: torch.Tensor) -> torch.Tensor:
    """vit: a DinoV2ViT whose Linear / Conv2d weights are bf16 (train.frozen_bf16_shadow); x: [B,3,H,W] f32, already normalised.
    Returns the final-norm tokens [B, 1+N, C] in bf16 (what the bottleneck MLP consumes)."""
    bf = torch.bfloat16
    t = vit.patch_embed(x.to(bf)).float()
    t = torch.cat([vit.cls_token.expand(t.shape[0], -1, -1).float(), t], dim=1) + vit.pos_embed.float()
    t = t.contiguous()                                   # f32 residual stream [B, S, C]
    b, s, c = t.shape
    for blk in vit.blocks:
        hn = ops.layernorm_bf16(t, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps)
        nh = blk.attn.num_heads
        hd = c // nh
        qkv = F.linear(hn, blk.attn.qkv.weight, blk.attn.qkv.bias)              # [b, s, 3*c] = [b, s, 3, heads, hd]
        if hd == 64 and s <= 288:
            o = ops.attention_qkv(qkv, nh, hd ** -0.5)                             # fused: nothing of size s x s reaches HBM
        else:
            qkv = qkv.reshape(b, s, 3, nh, hd).permute(2, 0, 3, 1, 4)
            att = ops.softmax_rows_bf16(qkv[0] @ qkv[1].transpose(-2, -1), hd ** -0.5)   # scale, f32 softmax and the casts in one pass
            o = (att @ qkv[2]).transpose(1, 2).reshape(b, s, c)
        o = F.linear(o, blk.attn.proj.weight, blk.attn.proj.bias)
        ops.scale_residual_(t, o.contiguous(), blk.ls1.gamma)
        hn = ops.layernorm_bf16(t, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        h = F.gelu(F.linear(hn, blk.mlp.fc1.weight, blk.mlp.fc1.bias))
        o = F.linear(h, blk.mlp.fc2.weight, blk.mlp.fc2.bias)
        ops.scale_residual_(t, o, blk.ls2.gamma)
    return ops.layernorm_bf16(t, vit.norm.weight, vit.norm.bias, vit.norm.eps)
