"""Drop-in counterpart of the reference's diffusion/lightningdit/{lightningdit,rms_norm,swiglu_ffn,pos_embed}.py: the LightningDiT velocity
model that train_dmd.py uses as the frozen teacher (`base_model`) and as the trainable student (`sit`) -- same constructor arguments,
`state_dict` keys (407 for XL/1, incl. the frozen `pos_embed` parameter and the `feat_rope.freqs_{cos,sin}` buffers), `forward(x, t, y)`
and `LightningDiT_models` table.

Two routes through `forward`:
  * no gradients needed (the four teacher / student evaluations inside the DMD loss, train_dmd.py:211-217; sampling) on a GPU at a shape
    the kernels cover -> `lightningdit_fast.forward_inference`: RMSNorm + adaLN modulate, QK-norm + RoPE, SwiGLU gate and the gated
    residual on HIP kernels (csrc/dit.hip), GEMMs through the library, attention on the conv kernel's batched-GEMM path;
  * gradients needed (the student's own training turn, train_dmd.py:565-575) on such a GPU shape -> `lightningdit_fast.forward_train`: one autograd
    Function per block (`functional.DitBlockFn`) over the same kernels and their backward counterparts;
  * otherwise (CPU, no autocast, unusual shapes) -> the stock PyTorch modules below, which also define the parameters.
SURVEY.md 8(f) rank 3."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def modulate(x, shift, scale):
    """adaLN modulation (reference lightningdit.py:27-31): x * (1 + scale) [+ shift], per-sample vectors broadcast over tokens."""
    if shift is None:
        return x * (1 + scale.unsqueeze(1))
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


class RMSNorm(nn.Module):
    """rms_norm.py:34-76: x * rsqrt(mean(x^2) + eps) computed in f32, cast back to the input dtype, times a learnable weight."""

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x) * self.weight


class SwiGLUFFN(nn.Module):
    """swiglu_ffn.py:15-36: w3(silu(x1) * x2) with [x1, x2] = w12(x)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, bias=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.w12 = nn.Linear(in_features, 2 * hidden_features, bias=bias)
        self.w3 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        x1, x2 = self.w12(x).chunk(2, dim=-1)
        return self.w3(F.silu(x1) * x2)


class _Mlp(nn.Module):
    """timm Mlp with tanh-GELU (the non-SwiGLU variant, lightningdit.py:218-224)."""

    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = nn.GELU(approximate="tanh")
        self.fc2 = nn.Linear(hidden_features, in_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


def rotate_half(x):
    """pos_embed.py:37-41: (x0, x1, x2, x3, ...) -> (-x1, x0, -x3, x2, ...)."""
    x = x.reshape(*x.shape[:-1], -1, 2)
    return torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(*x.shape[:-2], -1)


class VisionRotaryEmbeddingFast(nn.Module):
    """2-D rotary embedding of EVA-02 (pos_embed.py:96-135): per token (row r, column c of a pt_seq_len grid) the first `dim` features rotate
    with angle r * f_j, the next `dim` with c * f_j, f_j = theta^(-2j/dim), each frequency repeated for a feature pair."""

    def __init__(self, dim, pt_seq_len=16, ft_seq_len=None, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        ft_seq_len = pt_seq_len if ft_seq_len is None else ft_seq_len
        pos = torch.arange(ft_seq_len) / ft_seq_len * pt_seq_len
        ang = (pos[:, None] * freqs[None, :]).repeat_interleave(2, dim=-1)                       # [L, dim]
        full = torch.cat([ang[:, None, :].expand(-1, ft_seq_len, -1), ang[None, :, :].expand(ft_seq_len, -1, -1)], dim=-1)
        self.register_buffer("freqs_cos", full.cos().reshape(-1, full.shape[-1]))
        self.register_buffer("freqs_sin", full.sin().reshape(-1, full.shape[-1]))

    def forward(self, t):
        return t * self.freqs_cos + rotate_half(t) * self.freqs_sin


class Attention(nn.Module):
    """lightningdit.py:34-91: qkv Linear, optional per-head q/k normalisation, optional RoPE, softmax attention, proj."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, use_rmsnorm=False):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        norm = RMSNorm if use_rmsnorm else nn.LayerNorm
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = norm(self.head_dim) if qk_norm else nn.Identity()
        self.k_norm = norm(self.head_dim) if qk_norm else nn.Identity()
        self.proj = nn.Linear(dim, dim)

    def forward(self, x, rope=None):
        b, n, c = x.shape
        q, k, v = self.qkv(x).reshape(b, n, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        q, k = self.q_norm(q), self.k_norm(k)
        if rope is not None:
            q, k = rope(q), rope(k)
        x = F.scaled_dot_product_attention(q, k, v)
        return self.proj(x.transpose(1, 2).reshape(b, n, c))


class TimestepEmbedder(nn.Module):
    """lightningdit.py:94-139: sinusoidal features (cos first, then sin) -> Linear, SiLU, Linear."""

    def __init__(self, hidden_size, frequency_embedding_size=256):
        super().__init__()
        self.frequency_embedding_size = frequency_embedding_size
        self.mlp = nn.Sequential(nn.Linear(frequency_embedding_size, hidden_size), nn.SiLU(), nn.Linear(hidden_size, hidden_size))

    _FREQS = {}     # (half, max_period, device) -> table: computed on the CPU like the reference (:122-124), moved once (hipGraph-capturable afterwards)

    @staticmethod
    def timestep_embedding(t, dim, max_period=10000):
        half = dim // 2
        key = (half, max_period, str(t.device))
        freqs = TimestepEmbedder._FREQS.get(key)
        if freqs is None:
            freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
            TimestepEmbedder._FREQS[key] = freqs
        args = t[:, None].float() * freqs[None]
        emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
        if dim % 2:
            emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
        return emb

    def forward(self, t):
        return self.mlp(self.timestep_embedding(t, self.frequency_embedding_size))


class LabelEmbedder(nn.Module):
    """lightningdit.py:142-173: class embedding with one extra row for the dropped / unconditional label."""

    def __init__(self, num_classes, hidden_size, dropout_prob):
        super().__init__()
        self.embedding_table = nn.Embedding(num_classes + (dropout_prob > 0), hidden_size)
        self.num_classes, self.dropout_prob = num_classes, dropout_prob

    def token_drop(self, labels, force_drop_ids=None):
        drop = torch.rand(labels.shape[0], device=labels.device) < self.dropout_prob if force_drop_ids is None else force_drop_ids == 1
        return torch.where(drop, self.num_classes, labels)

    def forward(self, labels, train, force_drop_ids=None):
        if (train and self.dropout_prob > 0) or force_drop_ids is not None:
            labels = self.token_drop(labels, force_drop_ids)
        return self.embedding_table(labels)


class LightningDiTBlock(nn.Module):
    """lightningdit.py:175-250: x += gate_msa * attn(modulate(norm1(x))); x += gate_mlp * mlp(modulate(norm2(x)))."""

    def __init__(self, hidden_size, num_heads, mlp_ratio=4.0, use_qknorm=False, use_swiglu=False, use_rmsnorm=False, wo_shift=False):
        super().__init__()
        if use_rmsnorm:
            self.norm1, self.norm2 = RMSNorm(hidden_size), RMSNorm(hidden_size)
        else:
            self.norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
            self.norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.attn = Attention(hidden_size, num_heads=num_heads, qkv_bias=True, qk_norm=use_qknorm, use_rmsnorm=use_rmsnorm)
        mlp_hidden = int(hidden_size * mlp_ratio)
        self.mlp = SwiGLUFFN(hidden_size, int(2 / 3 * mlp_hidden)) if use_swiglu else _Mlp(hidden_size, mlp_hidden)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, (4 if wo_shift else 6) * hidden_size, bias=True))
        self.wo_shift = wo_shift

    def forward(self, x, c, feat_rope=None):
        if self.wo_shift:
            scale_msa, gate_msa, scale_mlp, gate_mlp = self.adaLN_modulation(c).chunk(4, dim=1)
            shift_msa = shift_mlp = None
        else:
            shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.adaLN_modulation(c).chunk(6, dim=1)
        x = x + gate_msa.unsqueeze(1) * self.attn(modulate(self.norm1(x), shift_msa, scale_msa), rope=feat_rope)
        return x + gate_mlp.unsqueeze(1) * self.mlp(modulate(self.norm2(x), shift_mlp, scale_mlp))


class FinalLayer(nn.Module):
    """lightningdit.py:252-273."""

    def __init__(self, hidden_size, patch_size, out_channels, use_rmsnorm=False):
        super().__init__()
        self.norm_final = RMSNorm(hidden_size) if use_rmsnorm else nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, patch_size * patch_size * out_channels, bias=True)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size, bias=True))

    def forward(self, x, c):
        shift, scale = self.adaLN_modulation(c).chunk(2, dim=1)
        return self.linear(modulate(self.norm_final(x), shift, scale))


class _PatchEmbed(nn.Module):
    """timm PatchEmbed as the reference uses it (lightningdit.py:305): Conv2d(kernel = stride = patch), tokens row-major."""

    def __init__(self, img_size, patch_size, in_chans, embed_dim, bias=True):
        super().__init__()
        self.img_size, self.patch_size = (img_size, img_size), (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


def get_1d_sincos_pos_embed_from_grid(embed_dim, pos):
    omega = 1.0 / 10000 ** (np.arange(embed_dim // 2, dtype=np.float64) / (embed_dim / 2.0))
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size):
    """lightningdit.py:467-499 (no class token): first half of the features encodes grid[0] (the column index -- `w goes first`), second the row."""
    grid = np.stack(np.meshgrid(np.arange(grid_size, dtype=np.float32), np.arange(grid_size, dtype=np.float32)), axis=0)
    grid = grid.reshape([2, 1, grid_size, grid_size])
    return np.concatenate([get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[0]), get_1d_sincos_pos_embed_from_grid(embed_dim // 2, grid[1])], axis=1)


class LightningDiT(nn.Module):
    """lightningdit.py:276-421."""

    def __init__(self, input_size=32, patch_size=2, in_channels=32, hidden_size=1152, depth=28, num_heads=16, mlp_ratio=4.0,
                 class_dropout_prob=0.1, num_classes=1000, learn_sigma=False, use_qknorm=True, use_swiglu=True, use_rope=True, use_rmsnorm=True,
                 wo_shift=False, use_checkpoint=False):
        super().__init__()
        self.learn_sigma, self.in_channels = learn_sigma, in_channels
        self.out_channels = in_channels * 2 if learn_sigma else in_channels
        self.patch_size, self.num_heads, self.use_rope, self.use_rmsnorm = patch_size, num_heads, use_rope, use_rmsnorm
        self.depth, self.hidden_size, self.use_checkpoint = depth, hidden_size, use_checkpoint
        self.x_embedder = _PatchEmbed(input_size, patch_size, in_channels, hidden_size, bias=True)
        self.t_embedder = TimestepEmbedder(hidden_size)
        self.y_embedder = LabelEmbedder(num_classes, hidden_size, class_dropout_prob)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.x_embedder.num_patches, hidden_size), requires_grad=False)
        self.feat_rope = VisionRotaryEmbeddingFast(dim=hidden_size // num_heads // 2, pt_seq_len=input_size // patch_size) if use_rope else None
        self.blocks = nn.ModuleList([LightningDiTBlock(hidden_size, num_heads, mlp_ratio=mlp_ratio, use_qknorm=use_qknorm, use_swiglu=use_swiglu,
                                                       use_rmsnorm=use_rmsnorm, wo_shift=wo_shift) for _ in range(depth)])
        self.final_layer = FinalLayer(hidden_size, patch_size, self.out_channels, use_rmsnorm=use_rmsnorm)
        self.initialize_weights()

    def initialize_weights(self):
        """lightningdit.py:342-376: xavier Linear weights / zero biases, fixed sin-cos pos_embed, normal(0.02) embeddings, zeroed adaLN and output."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        pe = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.x_embedder.num_patches ** 0.5))
        self.pos_embed.data.copy_(torch.from_numpy(pe).float().unsqueeze(0))
        w = self.x_embedder.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.constant_(self.x_embedder.proj.bias, 0)
        nn.init.normal_(self.y_embedder.embedding_table.weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[0].weight, std=0.02)
        nn.init.normal_(self.t_embedder.mlp[2].weight, std=0.02)
        for block in self.blocks:
            nn.init.constant_(block.adaLN_modulation[-1].weight, 0)
            nn.init.constant_(block.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].weight, 0)
        nn.init.constant_(self.final_layer.adaLN_modulation[-1].bias, 0)
        nn.init.constant_(self.final_layer.linear.weight, 0)
        nn.init.constant_(self.final_layer.linear.bias, 0)

    def unpatchify(self, x):
        c, p = self.out_channels, self.x_embedder.patch_size[0]
        h = w = int(x.shape[1] ** 0.5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, c)
        return x.permute(0, 5, 1, 3, 2, 4).reshape(x.shape[0], c, h * p, h * p)

    def forward(self, x, t=None, y=None):
        if x.is_cuda:
            from .. import parity
            if parity.on():          # fp32 parity mode: f32 activations, split-operand GEMMs, f32 elementwise kernels -- forward and backward (lightningdit_parity.py)
                from . import lightningdit_parity
                if lightningdit_parity.structurally_supported(self):
                    return lightningdit_parity.forward_parity(self, x, t, y)
            from . import lightningdit_fast
            if lightningdit_fast.tokens1_supported(self, x):        # config C1: one token per sample (toy_example_2d/dmd.py:436-454)
                return lightningdit_fast.forward_tokens1(self, x, t, y)
            if lightningdit_fast.supported(self, x):
                if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                    return lightningdit_fast.forward_train(self, x, t, y)        # autograd Functions over csrc/dit.hip
                return lightningdit_fast.forward_inference(self, x, t, y)
            why = ("call outside autocast(bfloat16) or a configuration the DiT kernels do not cover (RoPE + RMSNorm + SwiGLU blocks, width <= 2048, "
                   "even head dim <= 128, tokens a multiple of 32)")
        else:
            why = "CPU tensor"
        from .._stock import require_opt_in
        require_opt_in("LightningDiT.forward", why)
        return self.forward_stock(x, t, y)

    def forward_stock(self, x, t=None, y=None):
        """The stock PyTorch route (training of the student, CPU, shapes the kernels do not cover; the parity reference in the tests)."""
        x = self.x_embedder(x) + self.pos_embed
        c = self.t_embedder(t) + self.y_embedder(y, self.training)
        for block in self.blocks:
            x = block(x, c, self.feat_rope)
        x = self.unpatchify(self.final_layer(x, c))
        if self.learn_sigma:
            x, _ = x.chunk(2, dim=1)
        return x

    def forward_with_cfg(self, x, t, y, cfg_scale, cfg_interval=None, cfg_interval_start=None, standard_cfg=False):
        """lightningdit.py:423-448: both halves of the batch share the latent; guidance on the first 3 (or all in_channels) channels."""
        half = x[: len(x) // 2]
        out = self.forward(torch.cat([half, half], dim=0), t, y)
        k = self.in_channels if standard_cfg else 3
        eps, rest = out[:, :k], out[:, k:]
        cond, uncond = torch.split(eps, len(eps) // 2, dim=0)
        half_eps = uncond + cfg_scale * (cond - uncond)
        if cfg_interval is True and t[0] < cfg_interval_start:
            half_eps = cond
        return torch.cat([torch.cat([half_eps, half_eps], dim=0), rest], dim=1)


def _cfg(depth, hidden_size, patch_size, num_heads):
    return lambda **kw: LightningDiT(depth=depth, hidden_size=hidden_size, patch_size=patch_size, num_heads=num_heads, **kw)


LightningDiT_models = {                                    # lightningdit.py:519-565
    "LightningDiT-Mini/1": _cfg(6, 256, 1, 4), "LightningDiT-S/1": _cfg(12, 384, 1, 6),
    "LightningDiT-B/1": _cfg(12, 768, 1, 12), "LightningDiT-B/2": _cfg(12, 768, 2, 12),
    "LightningDiT-L/2": _cfg(24, 1024, 2, 16), "LightningDiT-L/1": _cfg(24, 1024, 1, 16),
    "LightningDiT-XL/1": _cfg(28, 1152, 1, 16), "LightningDiT-XL/2": _cfg(28, 1152, 2, 16),
    "LightningDiT-1p0B/1": _cfg(24, 1536, 1, 24), "LightningDiT-1p0B/2": _cfg(24, 1536, 2, 24),
    "LightningDiT-1p6B/1": _cfg(28, 1792, 1, 28), "LightningDiT-1p6B/2": _cfg(28, 1792, 2, 28),
}
