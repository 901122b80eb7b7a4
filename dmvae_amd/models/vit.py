"""DINOv2 ViT encoder with timm's `vit_{base,large}_patch14_dinov2` parameter names, in stock PyTorch-ROCm.

The reference builds this through the third-party `timm.create_model(..., pretrained=True)` (models/vae.py:47-50),
which is not installed here and needs the network.  The block algebra follows the reference's vendored
models/dinov2.py + models/dino_layers (LayerNorm eps 1e-6, MHA, LayerScale, GELU MLP); sub-module names match timm
(blocks.N.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,mlp.fc1,mlp.fc2,ls2.gamma}) so reference `vae.pt` files load.
SURVEY.md section-8(f) rank 3.  Three ways through it: frozen (tokenizer stage, train_tokenizer.py:295-297) -> the step harness runs
`vit_fast.frozen_forward_features` on a bf16 shadow; trainable on the GPU at a width the kernels cover -> `vit_fast.trainable_forward_features`
(autograd Functions over csrc/vit.hip + vit_bwd.hip, library GEMMs for the Linear layers); anything else (CPU construction / state_dict
work, unusual widths) -> the stock PyTorch modules below, which also define the parameter names.
"""
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn


class _Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)

    def forward(self, x):
        b, n, c = x.shape
        hd = c // self.num_heads
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, hd).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1), dim=-1)
        return self.proj((att.to(qkv[2].dtype) @ qkv[2]).transpose(1, 2).reshape(b, n, c))


class _LayerScale(nn.Module):
    def __init__(self, dim, init_values=1e-5):
        super().__init__()
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, init_values=1e-5):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _Attention(dim, num_heads)
        self.ls1 = _LayerScale(dim, init_values)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.ls2 = _LayerScale(dim, init_values)

    def forward(self, x):
        x = x + self.ls1(self.attn(self.norm1(x)))
        return x + self.ls2(self.mlp(self.norm2(x)))


class _PatchEmbed(nn.Module):
    def __init__(self, patch_size, in_chans, embed_dim):
        super().__init__()
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class DinoV2ViT(nn.Module):
    num_prefix_tokens = 1

    def __init__(self, embed_dim=1024, depth=24, num_heads=16, patch_size=16, img_size=256, in_chans=3):
        super().__init__()
        self.embed_dim = embed_dim
        self.patch_embed = _PatchEmbed(patch_size, in_chans, embed_dim)
        n = (img_size // patch_size) ** 2
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward_features(self, x):
        if x.is_cuda:
            from .vit_fast import frozen_forward_features, hip_path_supported, trainable_forward_features
            from .. import parity
            if parity.on():
                # fp32 parity mode: f32 activations, every contraction on the MFMA kernels over exactly-split operands (any width)
                if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
                    from .vit_parity import forward_features_parity        # the trainable encoder of the DMD stage (train_dmd.py:518-520): forward + backward in f32
                    return forward_features_parity(self, x)
                from .vit_fast import parity_forward_features
                return parity_forward_features(self, x)
            if hip_path_supported(self, self.pos_embed.shape[1]):
                if torch.is_grad_enabled() and self.pos_embed.requires_grad:
                    return trainable_forward_features(self, x)    # trainable encoder on the HIP kernels (csrc/vit.hip, vit_bwd.hip)
                needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
                if not needs_grad and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
                    # frozen / no-grad use under autocast(bf16) -- `vae.encode`, `vae(x, freeze_encoder=True)`: the fused inference route on cached
                    # bf16 weights.  Tokens come back in bf16 (the stock modules' final LayerNorm returns f32 under autocast; the bottleneck's
                    # Linear casts to bf16 either way).
                    return frozen_forward_features(self, x.float())
                why = "frozen / no-grad use outside autocast(bfloat16) (the HIP route implements the reference's autocast arithmetic; f32: DMVAE_PARITY=1)"
            else:
                nh = self.blocks[0].attn.num_heads
                why = (f"width {self.embed_dim} with {nh} heads at {self.pos_embed.shape[1]} tokens is outside the encoder kernels' range "
                       "(width 256/512/768/1024/1280/1536, head dim 64, <= 288 tokens)")
        else:
            why = "CPU tensor"
        from .._stock import require_opt_in
        require_opt_in("DinoV2ViT.forward_features", why)
        return self.forward_features_stock(x)

    def forward_features_stock(self, x):
        """The stock PyTorch route (parameter-name definition; CPU; widths the kernels do not cover; the parity reference in the tests)."""
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1).to(x.dtype), x], dim=1) + self.pos_embed.to(x.dtype)
        for blk in self.blocks:
            x = blk(x)
        return self.norm(x)


_ARCH = {
    "vit_base_patch14_dinov2.lvd142m": dict(embed_dim=768, depth=12, num_heads=12),
    "vit_large_patch14_dinov2.lvd142m": dict(embed_dim=1024, depth=24, num_heads=16),
}


def create_model(name, pretrained=True, patch_size=16, img_size=256, **kw):
    """Offline stand-in for timm.models.create_model (vae.py:48-50).  No network here, so `pretrained` weights cannot be
    fetched: the model is randomly initialised; load a reference `vae.pt` (which contains encoder.model.*) for real weights."""
    if name not in _ARCH:
        raise ValueError(f"unknown encoder {name!r}; known: {sorted(_ARCH)}")
    if pretrained:
        warnings.warn("timm pretrained DINOv2 weights are not available offline; encoder is randomly initialised "
                      "(load a vae.pt checkpoint for trained weights)", stacklevel=2)
    cfg = dict(_ARCH[name])
    cfg.update(kw)
    return DinoV2ViT(patch_size=patch_size, img_size=img_size, **cfg)
