"""CPU oracle: a functional restatement of the reference's DMVAE hot path (PyTorch CPU, fp32/fp64).

TEST INFRASTRUCTURE ONLY.  Nothing under ``dmvae_amd/`` imports this file; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may.  The product path has no
CPU fallback.

Pinning: every function here is checked against golden vectors captured by importing the
reference's own modules in the build container (``oracle/capture_golden.py`` ->
``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).  The reference ships no tests or
fixtures of its own, so those captured outputs are the pin.  The KL / MMD
functions have NO reference counterpart: they are this build's own
specification and their parity is *unpinned*.

All functions take parameters as a flat ``dict[str, Tensor]`` keyed exactly like the reference's
``state_dict()`` and use the reference's NCHW layout.  ``q`` is an optional rounding hook applied
at the points where the HIP path stores bf16 (conv/GEMM operands and outputs); ``q=None`` is the
plain fp32 algorithm.

Reference files followed (all under /root/reference):
  models/flux_ae.py:21-107,184-278   models/vae.py:10-98          utils/lpips.py:81-162
  train_tokenizer.py:134-150,179-204  train_dmd.py:204-230,408-416
  diffusion/transport/transport.py:105-164, path.py:5-136, utils.py:12-16
  models/dinov2.py + models/dino_layers/{block,attention,layer_scale,mlp,patch_embed}.py
  toy_example_2d/dmd.py:320-360, toy_example_2d/sshpae.py:29-71
  models/patchgan.py:99-151  utils/diffaug.py:43-114  train_tokenizer.py:190-227 (discriminator branch)
  diffusion/lightningdit/{lightningdit.py:27-421, rms_norm.py:34-76, swiglu_ffn.py:15-36, pos_embed.py:37-41,96-135}
  diffusion/transport/{integrators.py:8-77, transport.py:75-102,236-354, path.py:42-89}  sample_50k.py:128-157  train_diffusion.py:276-290
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
P = Dict[str, Tensor]
Q = Optional[Callable[[Tensor], Tensor]]


# --------------------------------------------------------------------------------------------
# rounding hooks
# --------------------------------------------------------------------------------------------
class _Bf16RoundSTE(torch.autograd.Function):
    """Round to bf16 in forward AND round the incoming gradient in backward (the HIP path stores
    activations and activation-gradients as bf16 at the same tensor sites)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _Bf16RoundFwdOnly(torch.autograd.Function):
    """Round to bf16 in forward, identity gradient (weights: f32 master, bf16 shadow copy)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


def bf16_round(x: Tensor) -> Tensor:
    return _Bf16RoundSTE.apply(x)


def bf16_round_weight(x: Tensor) -> Tensor:
    return _Bf16RoundFwdOnly.apply(x)


def _q(q: Q, x: Tensor) -> Tensor:
    return x if q is None else q(x)


def _qw(q: Q, w: Tensor) -> Tensor:
    return w if q is None else bf16_round_weight(w)


# --------------------------------------------------------------------------------------------
# models/flux_ae.py
# --------------------------------------------------------------------------------------------
def swish(x: Tensor) -> Tensor:
    """flux_ae.py:21-22"""
    return x * torch.sigmoid(x)


def group_norm(x: Tensor, weight: Tensor, bias: Tensor, groups: int = 32, eps: float = 1e-6) -> Tensor:
    """nn.GroupNorm(32, C, eps=1e-6, affine=True) (flux_ae.py:28,62,64,236), biased variance."""
    n, c = x.shape[:2]
    xg = x.reshape(n, groups, -1)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    xh = ((xg - mean) / torch.sqrt(var + eps)).reshape(x.shape)
    shape = [1, c] + [1] * (x.dim() - 2)
    return xh * weight.reshape(shape) + bias.reshape(shape)


def conv2d(x: Tensor, p: P, name: str, q: Q = None, stride: int = 1, padding: int = 1, round_out: bool = True) -> Tensor:
    y = F.conv2d(_q(q, x), _qw(q, p[name + ".weight"]), p[name + ".bias"], stride=stride, padding=padding)
    return _q(q, y) if round_out else y


def resnet_block(x: Tensor, p: P, pre: str, q: Q = None) -> Tensor:
    """flux_ae.py:69-82.  bf16 sites of the HIP path: conv inputs (after GN+swish), conv1 output,
    shortcut output, block output (conv2 + bias + residual rounded once)."""
    cin, cout = p[pre + "conv1.weight"].shape[1], p[pre + "conv1.weight"].shape[0]
    h = swish(group_norm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]))
    h = conv2d(h, p, pre + "conv1", q)
    h = swish(group_norm(h, p[pre + "norm2.weight"], p[pre + "norm2.bias"]))
    h = conv2d(h, p, pre + "conv2", q, round_out=False)
    if cin != cout:
        x = conv2d(x, p, pre + "nin_shortcut", q, padding=0)
    return _q(q, x + h)


def attn_block(x: Tensor, p: P, pre: str, q: Q = None) -> Tensor:
    """flux_ae.py:37-52: GN -> q,k,v 1x1 -> single-head SDPA over (h w) tokens with d=C -> proj -> +x."""
    b, c, hh, ww = x.shape
    h_ = group_norm(x, p[pre + "norm.weight"], p[pre + "norm.bias"])
    qq = conv2d(h_, p, pre + "q", q, padding=0)
    kk = conv2d(h_, p, pre + "k", q, padding=0)
    vv = conv2d(h_, p, pre + "v", q, padding=0)
    qt = qq.reshape(b, c, hh * ww).transpose(1, 2)  # b (h w) c
    kt = kk.reshape(b, c, hh * ww).transpose(1, 2)
    vt = vv.reshape(b, c, hh * ww).transpose(1, 2)
    s = torch.matmul(qt, kt.transpose(1, 2)) * (1.0 / math.sqrt(c))
    pr = torch.softmax(s, dim=-1)
    o = torch.matmul(_q(q, pr), vt)  # HIP path stores P as bf16 before P.V
    o = _q(q, o).transpose(1, 2).reshape(b, c, hh, ww)
    o = conv2d(o, p, pre + "proj_out", q, padding=0, round_out=False)
    return _q(q, x + o)


_SUBPIX = torch.tensor([[0., 0., 1.], [0., 1., 1.], [1., 1., 0.], [1., 0., 0.]])      # [r][k] = 1 when 2 <= r + k <= 3


def subpixel_weight(w: Tensor) -> Tensor:
    """W [cout, cin, 3, 3] -> WD [cin, cout, 4, 4] with conv2d(nearest_x2(x), W, padding=1) == conv_transpose2d(x, WD, stride=2, padding=1):
    output pixel (2y+py, 2x+px) of flux_ae.py:103-107 only sees the 2x2 source pixels around (y, x), so the taps of W that read the same source
    pixel are added up front (WD[ci][co][r][s] = sum over ky, kx with 2 <= r+ky <= 3, 2 <= s+kx <= 3).  This is how the HIP path evaluates the
    layer (include/dmvae_hip.h: dmvae_subpixel_weight); the plain-f32 oracle below stays the reference's own two-step form."""
    e = _SUBPIX.to(w.dtype)
    return torch.einsum("rk,sl,oikl->iors", e, e, w)


def upsample(x: Tensor, p: P, pre: str, q: Q = None) -> Tensor:
    """flux_ae.py:103-107: nearest x2 then conv3x3.  With q (bf16 rounding at the HIP path's storage sites) the layer is evaluated the way the HIP path
    evaluates it -- sub-pixel form, the bf16 rounding on the pre-added 4x4 weights -- which is the same function of (x, W) up to that rounding."""
    if q is not None:
        wd = _qw(q, subpixel_weight(p[pre + "conv.weight"]))
        return _q(q, F.conv_transpose2d(_q(q, x), wd, p[pre + "conv.bias"], stride=2, padding=1))
    x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    return conv2d(x, p, pre + "conv", q)


def downsample(x: Tensor, p: P, pre: str, q: Q = None) -> Tensor:
    """flux_ae.py:91-95: pad (0,1,0,1) then conv3x3 stride 2 pad 0."""
    x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0.0)
    return conv2d(x, p, pre + "conv", q, stride=2, padding=0)


def decoder_forward(z: Tensor, p: P, pre: str = "", q: Q = None, num_resolutions: int = 4, num_res_blocks: int = 2,
                    final_f32: bool = True) -> Tensor:
    """flux_ae.py:239-269 with post_init's conv_in = Sequential(Upsample(z), Conv3x3) (:271-275).
    z: [B,256,C] tokens (hard-coded 16x16, :244-245) or [B,C,h,w]."""
    if z.dim() == 3:
        b, t, c = z.shape
        assert t == 256, "reference hard-codes a 16x16 token grid (flux_ae.py:245)"
        z = z.reshape(b, 16, 16, c).permute(0, 3, 1, 2)
    h = upsample(_q(q, z), p, pre + "conv_in.0.", q)
    h = conv2d(h, p, pre + "conv_in.1", q)
    h = resnet_block(h, p, pre + "mid.block_1.", q)
    h = attn_block(h, p, pre + "mid.attn_1.", q)
    h = resnet_block(h, p, pre + "mid.block_2.", q)
    for lvl in reversed(range(num_resolutions)):
        for blk in range(num_res_blocks + 1):
            h = resnet_block(h, p, f"{pre}up.{lvl}.block.{blk}.", q)
        if lvl != 0:
            h = upsample(h, p, f"{pre}up.{lvl}.upsample.", q)
    h = swish(group_norm(h, p[pre + "norm_out.weight"], p[pre + "norm_out.bias"]))
    return conv2d(h, p, pre + "conv_out", q, round_out=not final_f32)


def encoder_forward(x: Tensor, p: P, pre: str = "", q: Q = None, num_resolutions: int = 4, num_res_blocks: int = 2) -> Tensor:
    """flux_ae.py:160-181 (defined but never instantiated by the reference scripts; same kernels)."""
    h = conv2d(x, p, pre + "conv_in", q)
    for lvl in range(num_resolutions):
        for blk in range(num_res_blocks):
            h = resnet_block(h, p, f"{pre}down.{lvl}.block.{blk}.", q)
        if lvl != num_resolutions - 1:
            h = downsample(h, p, f"{pre}down.{lvl}.downsample.", q)
    h = resnet_block(h, p, pre + "mid.block_1.", q)
    h = attn_block(h, p, pre + "mid.attn_1.", q)
    h = resnet_block(h, p, pre + "mid.block_2.", q)
    h = swish(group_norm(h, p[pre + "norm_out.weight"], p[pre + "norm_out.bias"]))
    return conv2d(h, p, pre + "conv_out", q)


# --------------------------------------------------------------------------------------------
# models/vae.py
# --------------------------------------------------------------------------------------------
def linear(x: Tensor, p: P, name: str, q: Q = None, round_out: bool = True) -> Tensor:
    y = F.linear(_q(q, x), _qw(q, p[name + ".weight"]), p.get(name + ".bias"))
    return _q(q, y) if round_out else y


def mlp_forward(x: Tensor, p: P, pre: str = "bottle_neck.", q: Q = None) -> Tensor:
    """vae.py:56-65: Linear -> SiLU -> Linear."""
    h = linear(x, p, pre + "mlp.0", q)
    return linear(F.silu(h), p, pre + "mlp.2", q)


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-6) -> Tensor:
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def vit_forward_features(x: Tensor, p: P, pre: str = "encoder.model.", num_heads: int = 16, patch: int = 16, q: Q = None) -> Tensor:
    """DINOv2 ViT forward_features -> [B, 1+N, C] (cls first), following models/dinov2.py:224-262 and
    dino_layers/{patch_embed.py:68-81, block.py:89-115, attention.py:56-69, layer_scale.py:26, mlp.py:34-39}.
    timm's `vit_*_patch14_dinov2` uses the same algebra and the same parameter sub-names."""
    w = p[pre + "patch_embed.proj.weight"]
    t = F.conv2d(_q(q, x), _qw(q, w), p[pre + "patch_embed.proj.bias"], stride=patch)
    b, c = t.shape[:2]
    t = _q(q, t.flatten(2).transpose(1, 2))
    t = torch.cat([p[pre + "cls_token"].expand(b, -1, -1), t], dim=1) + p[pre + "pos_embed"]
    depth = 1 + max(int(k[len(pre) + 7:].split(".")[0]) for k in p if k.startswith(pre + "blocks."))
    hd = c // num_heads
    for i in range(depth):
        bp = f"{pre}blocks.{i}."
        h = layer_norm(t, p[bp + "norm1.weight"], p[bp + "norm1.bias"])
        qkv = linear(h, p, bp + "attn.qkv", q).reshape(b, -1, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        att = torch.softmax((qkv[0] * hd ** -0.5) @ qkv[1].transpose(-2, -1), dim=-1)
        h = (att @ qkv[2]).transpose(1, 2).reshape(b, -1, c)
        h = linear(h, p, bp + "attn.proj", q)
        t = t + h * p[bp + "ls1.gamma"]
        h = layer_norm(t, p[bp + "norm2.weight"], p[bp + "norm2.bias"])
        h = linear(F.gelu(linear(h, p, bp + "mlp.fc1", q)), p, bp + "mlp.fc2", q)
        t = t + h * p[bp + "ls2.gamma"]
    return layer_norm(t, p[pre + "norm.weight"], p[pre + "norm.bias"])


def dino_encoder_forward(x: Tensor, p: P, pre: str = "encoder.", num_heads: int = 16, q: Q = None) -> Tensor:
    """vae.py:52-53 with Denormalize/Normalize (:10-31): x in [-1,1] -> ImageNet-normalised -> ViT -> drop cls."""
    x = x * p[pre + "de_scale.std"] + p[pre + "de_scale.mean"]
    x = (x - p[pre + "scale.mean"]) / p[pre + "scale.std"]
    return vit_forward_features(x, p, pre + "model.", num_heads=num_heads, q=q)[:, 1:]


def vae_forward(x: Tensor, p: P, num_heads: int = 16, q: Q = None, return_latent: bool = False):
    """vae.py:90-98."""
    tok = dino_encoder_forward(x, p, num_heads=num_heads, q=q)
    lat = mlp_forward(tok, p, q=q)
    rec = decoder_forward(lat, p, pre="decoder.", q=q).float()
    return (rec, lat) if return_latent else rec


# --------------------------------------------------------------------------------------------
# utils/lpips.py
# --------------------------------------------------------------------------------------------
LPIPS_SHIFT = (-0.030, -0.088, -0.188)
LPIPS_SCALE = (0.458, 0.448, 0.450)
VGG_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
VGG_TAPS = (3, 8, 15, 22, 29)  # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 (lpips.py:126-135)


def vgg16_features(x: Tensor, p: P, pre: str = "net.", q: Q = None) -> List[Tensor]:
    """torchvision VGG16 'D' features sliced as lpips.py:116-153; returns the five tapped activations.
    Parameter names follow the reference's slices: net.slice{1..5}.{idx}.{weight,bias}."""
    outs, idx, h = [], 0, x
    bounds = (4, 9, 16, 23, 30)
    for v in VGG_CFG:
        sl = 1 + sum(idx >= bnd for bnd in bounds)
        if v == "M":
            h = F.max_pool2d(h, 2, 2)
            idx += 1
        else:
            name = f"{pre}slice{sl}.{idx}"
            h = F.relu(F.conv2d(_q(q, h), _qw(q, p[name + ".weight"]), p[name + ".bias"], padding=1))
            h = _q(q, h)
            idx += 2
        if idx - 1 in VGG_TAPS:
            outs.append(h)
    return outs


def lpips_from_feats(f0: Sequence[Tensor], f1: Sequence[Tensor], lin_w: Sequence[Tensor], eps: float = 1e-10) -> Tensor:
    """lpips.py:86-94,156-162: sum_l mean_hw( w_l . (f0/|f0| - f1/|f1|)^2 ), then mean over batch -> scalar."""
    val = 0.0
    for a, b, w in zip(f0, f1, lin_w):
        na = a / (torch.sqrt((a ** 2).sum(dim=1, keepdim=True)) + eps)
        nb = b / (torch.sqrt((b ** 2).sum(dim=1, keepdim=True)) + eps)
        d = (na - nb) ** 2
        val = val + (d * w.reshape(1, -1, 1, 1)).sum(dim=1, keepdim=True).mean(dim=(2, 3), keepdim=True)
    return val.mean()


def lpips_forward(inp: Tensor, tgt: Tensor, p: P, q: Q = None) -> Tensor:
    """lpips.py:81-94 (ScalingLayer :97-104)."""
    shift = torch.tensor(LPIPS_SHIFT, dtype=inp.dtype).reshape(1, 3, 1, 1)
    scale = torch.tensor(LPIPS_SCALE, dtype=inp.dtype).reshape(1, 3, 1, 1)
    f0 = vgg16_features((inp - shift) / scale, p, q=q)
    f1 = vgg16_features((tgt - shift) / scale, p, q=q)
    lin = [p[f"lin{k}.model.1.weight"].reshape(-1) for k in range(5)]
    return lpips_from_feats(f0, f1, lin)


# --------------------------------------------------------------------------------------------
# train_tokenizer.py / train_dmd.py losses
# --------------------------------------------------------------------------------------------
def l1_mse(recon: Tensor, images: Tensor) -> Tuple[Tensor, Tensor]:
    """train_tokenizer.py:180-181."""
    d = recon - images
    return d.abs().mean(), (d * d).mean()


def forward_generator(images: Tensor, recon: Tensor, lpips_p: P, l1_w: float = 1.0, l2_w: float = 0.0,
                      lpips_w: float = 1.0, q: Q = None):
    """train_tokenizer.py:179-190 with the discriminator branch off (step < disc_start_step)."""
    l1, l2 = l1_mse(recon, images)
    lp = lpips_forward(images, recon, lpips_p, q=q)
    rec = l1 * l1_w + l2 * l2_w + lp * lpips_w
    return rec, {"L1": l1, "L2": l2, "LPIPS": lp, "rec_loss": rec}


# --------------------------------------------------------------------------------------------
# diffusion/lightningdit: the LightningDiT velocity model (teacher / student of train_dmd.py)
# --------------------------------------------------------------------------------------------
def rms_norm(x: Tensor, w: Tensor, eps: float = 1e-6, q: Q = None) -> Tensor:
    """rms_norm.py:52-76: normalise in f32, cast back to the input's dtype (a bf16 rounding site when the input is bf16), then scale."""
    xf = x.float()
    return _q(q, xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)) * w


def rope_2d(t: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """pos_embed.py:37-41,135: t * cos + rotate_half(t) * sin with (x0, x1) -> (-x1, x0) on consecutive feature pairs."""
    r = t.reshape(*t.shape[:-1], -1, 2)
    rot = torch.stack((-r[..., 1], r[..., 0]), dim=-1).reshape(t.shape)
    return t * cos + rot * sin


def lightningdit_forward(x: Tensor, t: Tensor, y: Tensor, p: P, num_heads: int, patch_size: int = 1, q: Q = None) -> Tensor:
    """LightningDiT.forward in eval mode (lightningdit.py:390-421) for the configuration train_dmd.py builds (RMSNorm, QK-norm, RoPE, SwiGLU,
    shift + scale + gate adaLN; :289-294 defaults).  p: state_dict.  bf16 sites (`q`) = where autocast(bf16) rounds in the reference and where
    the HIP inference path stores bf16: every Linear's input and output, the per-head q/k after normalisation, the SwiGLU gate and product,
    the gated branch output before it joins the f32 residual stream."""
    b, c_in, hh, ww = x.shape
    hid = p["pos_embed"].shape[-1]
    hd = hid // num_heads
    lin = lambda v, name: _q(q, F.linear(_q(q, v), _qw(q, p[name + ".weight"]), p[name + ".bias"]))
    # patch embedding: Conv2d(kernel = stride = patch) == Linear over (c, ky, kx) patches, tokens row-major
    w = p["x_embedder.proj.weight"]
    pt = x.reshape(b, c_in, hh // patch_size, patch_size, ww // patch_size, patch_size).permute(0, 2, 4, 1, 3, 5).reshape(b, -1, c_in * patch_size ** 2)
    h = _q(q, F.linear(_q(q, pt), _qw(q, w.reshape(w.shape[0], -1)), p["x_embedder.proj.bias"])) + p["pos_embed"]
    half = 128
    freqs = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    temb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    temb = lin(F.silu(lin(temb, "t_embedder.mlp.0")), "t_embedder.mlp.2")
    c = temb + p["y_embedder.embedding_table.weight"][y]
    cos, sin = p["feat_rope.freqs_cos"], p["feat_rope.freqs_sin"]
    nblk = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("blocks."))
    for i in range(nblk):
        pre = f"blocks.{i}."
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = lin(F.silu(c), pre + "adaLN_modulation.1").chunk(6, dim=1)
        a = rms_norm(h, p[pre + "norm1.weight"]) * _q(q, 1 + sc_a.unsqueeze(1)) + sh_a.unsqueeze(1)      # `1 + scale` is itself a bf16 tensor under autocast
        qkv = lin(a, pre + "attn.qkv").reshape(b, -1, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        qq = rope_2d(rms_norm(qkv[0], p[pre + "attn.q_norm.weight"], q=q), cos, sin)
        kk = rope_2d(rms_norm(qkv[1], p[pre + "attn.k_norm.weight"], q=q), cos, sin)
        att = torch.softmax((_q(q, qq) @ _q(q, kk).transpose(-2, -1)) * hd ** -0.5, dim=-1)
        o = _q(q, _q(q, att) @ qkv[2]).transpose(1, 2).reshape(b, -1, hid)
        h = h + _q(q, g_a.unsqueeze(1) * lin(o, pre + "attn.proj"))
        a = rms_norm(h, p[pre + "norm2.weight"]) * _q(q, 1 + sc_m.unsqueeze(1)) + sh_m.unsqueeze(1)
        x1, x2 = lin(a, pre + "mlp.w12").chunk(2, dim=-1)
        h = h + _q(q, g_m.unsqueeze(1) * lin(_q(q, _q(q, F.silu(x1)) * x2), pre + "mlp.w3"))
    sh, sc = lin(F.silu(c), "final_layer.adaLN_modulation.1").chunk(2, dim=1)
    a = rms_norm(h, p["final_layer.norm_final.weight"]) * _q(q, 1 + sc.unsqueeze(1)) + sh.unsqueeze(1)
    o = lin(a, "final_layer.linear")
    gh = hh // patch_size
    c_out = o.shape[-1] // patch_size ** 2
    return o.reshape(b, gh, gh, patch_size, patch_size, c_out).permute(0, 5, 1, 3, 2, 4).reshape(b, c_out, hh, ww)


# --------------------------------------------------------------------------------------------
# models/patchgan.py, utils/diffaug.py and the discriminator branch of train_tokenizer.py
# --------------------------------------------------------------------------------------------
def batch_norm(x: Tensor, w: Tensor, b: Tensor, running_mean: Tensor, running_var: Tensor, training: bool, momentum: float = 0.1,
               eps: float = 1e-5):
    """nn.BatchNorm2d == nn.SyncBatchNorm on one rank (patchgan.py:113-115).  Returns (y, running_mean', running_var')."""
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        n = x.numel() // x.shape[1]
        new_rm = (1 - momentum) * running_mean + momentum * mean.detach()
        new_rv = (1 - momentum) * running_var + momentum * var.detach() * (n / max(n - 1, 1))
    else:
        mean, var, new_rm, new_rv = running_mean, running_var, running_mean, running_var
    sh = (1, -1, 1, 1)
    y = (x - mean.reshape(sh)) / torch.sqrt(var.reshape(sh) + eps) * w.reshape(sh) + b.reshape(sh)
    return y, new_rm, new_rv


def patchgan_forward(x: Tensor, p: P, pre: str = "main.", q: Q = None, training: bool = True, n_layers: int = 3):
    """NLayerDiscriminator.forward (patchgan.py:99-151): conv4x4 s2 + LeakyReLU; (n_layers-1) x [conv4x4 s2, BN, LeakyReLU];
    [conv4x4 s1, BN, LeakyReLU]; conv4x4 s1 -> 1 channel.  bf16 sites of the HIP path: every conv input / weight, every conv
    output except the f32 logits, every BN+LeakyReLU output.  Returns (logits, {buffer name: updated running statistic})."""
    new_buf: Dict[str, Tensor] = {}
    h = F.conv2d(_q(q, x), _qw(q, p[pre + "0.weight"]), p[pre + "0.bias"], stride=2, padding=1)
    h = _q(q, F.leaky_relu(h, 0.2))
    idx = 2
    for layer in range(1, n_layers + 1):
        stride = 2 if layer < n_layers else 1
        h = _q(q, F.conv2d(_q(q, h), _qw(q, p[f"{pre}{idx}.weight"]), p.get(f"{pre}{idx}.bias"), stride=stride, padding=1))
        bn = f"{pre}{idx + 1}."
        h, rm, rv = batch_norm(h, p[bn + "weight"], p[bn + "bias"], p[bn + "running_mean"], p[bn + "running_var"], training)
        new_buf[bn + "running_mean"], new_buf[bn + "running_var"] = rm, rv
        h = _q(q, F.leaky_relu(h, 0.2))
        idx += 3
    return F.conv2d(_q(q, h), _qw(q, p[f"{pre}{idx}.weight"]), p[f"{pre}{idx}.bias"], stride=1, padding=1), new_buf


def diffaug(x: Tensor, rand01: Tensor, trans: bool = True, color: bool = True, cut: bool = True, cutout: float = 0.2) -> Tensor:
    """DiffAug.aug (diffaug.py:43-114) with the blur warm-up off (every reference call site passes schedule 0) and the random draws
    injected: rand01 [7, B] are the values of torch.rand(7, B, 1, 1) (:69), (trans, color, cut) the outcome of torch.rand(3) <= prob
    (:66).  Translation by up to 1/8 of the size with zero fill; brightness, per-pixel saturation, per-image contrast; one zeroed
    rectangle of `cutout` x size, clamped at the border."""
    x = x.float()
    B, C, H, W = x.shape
    r = rand01.reshape(7, B).to(x.dtype)
    if trans:
        dh, dw = round(H * 0.125), round(W * 0.125)
        th = torch.floor(r[0] * (2 * dh + 1)).long() - dh
        tw = torch.floor(r[1] * (2 * dw + 1)).long() - dw
        ys = torch.arange(H).view(1, H, 1) + th.view(B, 1, 1)            # source row of output row h
        xs = torch.arange(W).view(1, 1, W) + tw.view(B, 1, 1)
        ok = ((ys >= 0) & (ys < H) & (xs >= 0) & (xs < W)).unsqueeze(1)
        bi = torch.arange(B).view(B, 1, 1).expand(B, H, W)
        g = x.permute(0, 2, 3, 1)[bi, ys.clamp(0, H - 1).expand(B, H, W), xs.clamp(0, W - 1).expand(B, H, W)].permute(0, 3, 1, 2)
        x = torch.where(ok, g, torch.zeros_like(g))
    if color:
        x = x + (r[2].view(B, 1, 1, 1) - 0.5)
        m = x.mean(dim=1, keepdim=True)
        x = (x - m) * (r[3].view(B, 1, 1, 1) * 2) + m
        m = x.mean(dim=(1, 2, 3), keepdim=True)
        x = (x - m) * (r[4].view(B, 1, 1, 1) + 0.5) + m
    if cut:
        ch, cw = round(H * cutout), round(W * cutout)
        oh = torch.floor(r[5] * (H + (1 - ch % 2))).long()
        ow = torch.floor(r[6] * (W + (1 - cw % 2))).long()
        mask = torch.ones(B, H, W, dtype=x.dtype)
        for b in range(B):
            hh = (torch.arange(ch) + oh[b] - ch // 2).clamp(0, H - 1)
            ww = (torch.arange(cw) + ow[b] - cw // 2).clamp(0, W - 1)
            mask[b][hh.view(-1, 1), ww.view(1, -1)] = 0
        x = x * mask.unsqueeze(1)
    return x


def hinge_d_loss(logits_real: Tensor, logits_fake: Tensor) -> Tensor:
    """train_tokenizer.py:214."""
    return 0.5 * (F.relu(1.0 - logits_real).mean() + F.relu(1.0 + logits_fake).mean())


def forward_discriminator(images: Tensor, recon: Tensor, disc_p: P, rand01_a: Tensor, rand01_b: Tensor, bcr_weight: float,
                          cutout_a: float = 0.2, cutout_b: float = 0.2, q: Q = None):
    """VAELossFunction.forward_discriminator (train_tokenizer.py:207-227): hinge loss on D(aug([images; recon])) plus the balanced
    consistency term bcr * mse(D(strong_aug(.)), D(aug(.))); the discriminator is in train mode for both passes (BatchNorm uses batch
    statistics and updates its running estimates twice).  Returns (loss, log, buffers after both passes)."""
    bs = images.shape[0]
    both = torch.cat([images, recon], dim=0)
    logits, buf1 = patchgan_forward(diffaug(both, rand01_a, cutout=cutout_a), disc_p, q=q, training=True)
    logits = logits.float()
    lr, lf = logits[:bs], logits[bs:]
    d_loss = hinge_d_loss(lr, lf)
    p2 = dict(disc_p)
    p2.update(buf1)
    logits2, buf2 = patchgan_forward(diffaug(both, rand01_b, cutout=cutout_b), p2, q=q, training=True)
    bcr = F.mse_loss(logits2.float(), logits) * bcr_weight
    acc_real = (lr.detach() > 0).float().mean() * 100
    acc_fake = (lf.detach() < 0).float().mean() * 100
    log = {"d_loss": d_loss.detach(), "bcr_loss": bcr.detach(), "acc_real": acc_real, "acc_fake": acc_fake, "acc_mean": (acc_real + acc_fake) * 0.5}
    return d_loss + bcr, log, buf2


def adaptive_disc_weight(rec_loss: Tensor, g_loss: Tensor, last_layer: Tensor, disc_weight: float) -> Tensor:
    """train_tokenizer.py:194-198: disc_weight * clamp(|d rec / d last| / (|d g / d last| + 1e-6), 0, 1e4)."""
    gr = torch.autograd.grad(rec_loss, last_layer, retain_graph=True)[0]
    gd = torch.autograd.grad(g_loss, last_layer, retain_graph=True)[0]
    return disc_weight * (gr.norm() / (gd.norm() + 1e-6)).clamp(0.0, 1e4).detach()


def forward_generator_gan(images: Tensor, recon: Tensor, lpips_p: P, disc_p: P, last_layer: Tensor, rand01: Tensor, disc_weight: float = 0.5,
                          l1_w: float = 1.0, l2_w: float = 0.0, lpips_w: float = 1.0, cutout: float = 0.2, q: Q = None):
    """train_tokenizer.py:179-204 with the discriminator branch on (step >= disc_start_step): the discriminator is in eval mode
    (running statistics), g = -mean D(aug(recon)), total = rec + adaptive_weight * g."""
    rec, log = forward_generator(images, recon, lpips_p, l1_w, l2_w, lpips_w, q)
    logits, _ = patchgan_forward(diffaug(recon, rand01, cutout=cutout), disc_p, q=q, training=False)
    g_loss = -logits.float().mean()
    w = adaptive_disc_weight(rec, g_loss, last_layer, disc_weight)
    log = dict(log, d_weight=w)
    return rec + g_loss * w, log


def expand_t(t: Tensor, x: Tensor) -> Tensor:
    return t.reshape(t.shape[0], *([1] * (x.dim() - 1)))


def transport_plan(t: Tensor, x0: Tensor, x1: Tensor) -> Tuple[Tensor, Tensor]:
    """ICPlan (path.py:18-27,114-136): alpha=t, sigma=1-t -> xt = t*x1 + (1-t)*x0, ut = x1 - x0."""
    te = expand_t(t, x1)
    return te * x1 + (1 - te) * x0, x1 - x0


def transport_loss(model_out: Tensor, t: Tensor, x0: Tensor, x1: Tensor) -> Tensor:
    """transport.py:134-143: mean_flat((model_output - ut)^2) per sample."""
    _, ut = transport_plan(t, x0, x1)
    return ((model_out - ut) ** 2).flatten(1).mean(dim=1)


def dmd_loss(latents: Tensor, t: Tensor, x0: Tensor, v_teacher: Tensor, v_student: Tensor,
             v_teacher_u: Optional[Tensor] = None, v_student_u: Optional[Tensor] = None, cfg: float = 1.0,
             weight_factor: bool = True):
    """train_dmd.py:204-230 with (t, x0) and the four model outputs injected.  `t` is already mapped
    into [t0, t1].  weight_factor=False gives the toy variant (toy_example_2d/dmd.py:349-360).
    Returns loss, dmd_gradient_norm, grad (the detached score-gradient; dloss/dlatents = grad/numel)."""
    xt, _ = transport_plan(t, x0, latents)
    vt, vs = v_teacher, v_student
    if cfg > 1:
        vt = vt + (cfg - 1) * (vt - v_teacher_u)
        vs = vs + (cfg - 1) * (vs - v_student_u)
    om = expand_t(1 - t, xt)
    pred_t = xt + vt * om
    pred_s = xt + vs * om
    p_real = latents - pred_t
    p_student = latents - pred_s
    grad = p_real - p_student
    if weight_factor:
        grad = grad / p_real.abs().mean(dim=tuple(range(1, latents.dim())), keepdim=True)
    grad = torch.nan_to_num(grad).detach()
    loss = 0.5 * F.mse_loss(latents, (latents - grad).detach(), reduction="mean")
    gnorm = torch.norm(grad.flatten(1), dim=1).mean()
    return loss, gnorm, grad


def latents_to_spatial(tokens: Tensor) -> Tensor:
    """train_dmd.py:408-416 (p=1): [B, h*w, C] -> [B, C, h, w], a pure permutation (bit-exact)."""
    b, t, c = tokens.shape
    h = int(t ** 0.5)
    assert h * h == t
    return tokens.reshape(b, h, h, c).permute(0, 3, 1, 2).contiguous()


# --------------------------------------------------------------------------------------------
# optimiser tail (train_tokenizer.py:140-150,382-392,415-419)
# --------------------------------------------------------------------------------------------
def clip_grad_norm(grads: Sequence[Tensor], max_norm: float = 1.0, eps: float = 1e-6) -> Tuple[Tensor, List[Tensor]]:
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + eps), max=1.0)
    return total, [g * coef for g in grads]


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, beta1: float = 0.9, beta2: float = 0.95,
               eps: float = 1e-8, wd: float = 0.005):
    """torch.optim.AdamW semantics (decoupled decay, bias correction, eps added after sqrt(v_hat))."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * m / denom
    return p, m, v


def ema_update(ema: Tensor, p: Tensor, decay: float = 0.9999) -> Tensor:
    return ema * decay + p * (1 - decay)


def tokenizer_train_steps(images: Tensor, p: P, lpips_p: P, trainable: Sequence[str], steps: int, base_lr: float = 1e-4,
                          warmup_steps: int = 1000, num_heads: int = 16, max_norm: float = 1.0, ema_decay: float = 0.9999, q: Q = None,
                          on_grads: Optional[Callable[[int, Dict[str, Tensor]], None]] = None):
    """The loop body of train_tokenizer.py:403-437 with the discriminator branch off, `steps` times on the same batch:
    VAE.forward (frozen encoder, :411) -> forward_generator (:412) -> backward (:414) -> clip_grad_norm_ over every parameter that
    received a gradient (:415) -> AdamW(betas (0.9, 0.95), eps 1e-8, wd 0.005; :382) at LambdaLR's rate for this step (:385-392) ->
    update_ema (:437; the frozen encoder's EMA copy equals the encoder and is not tracked here).
    p is updated functionally; returns (per-step logs, parameters, EMA of the trainable parameters).  Pinned by tests/golden/step_small.npz."""
    p = dict(p)
    m = {k: torch.zeros_like(p[k]) for k in trainable}
    v = {k: torch.zeros_like(p[k]) for k in trainable}
    ema = {k: p[k].detach().clone() for k in trainable}
    logs = []
    for step in range(steps):
        leaves = {k: p[k].detach().clone().requires_grad_(True) for k in trainable}
        pp = {**{k: t.detach() for k, t in p.items()}, **leaves}
        with torch.no_grad():
            tok = dino_encoder_forward(images, pp, num_heads=num_heads, q=q)
        rec = decoder_forward(mlp_forward(tok, pp, q=q), pp, pre="decoder.", q=q).float()
        loss, log = forward_generator(images, rec, lpips_p, q=q)
        grads = dict(zip(trainable, torch.autograd.grad(loss, [leaves[k] for k in trainable])))
        if on_grads is not None:
            on_grads(step, grads)
        total, clipped = clip_grad_norm([grads[k] for k in trainable], max_norm)
        lr = warmup_lr(step, base_lr, warmup_steps)
        for k, g in zip(trainable, clipped):
            p[k], m[k], v[k] = adamw_step(p[k].detach(), g, m[k], v[k], step + 1, lr)
            ema[k] = ema_update(ema[k], p[k], ema_decay)
        logs.append({**{kk: float(vv.detach()) for kk, vv in log.items()}, "vae_norm": float(total), "lr": lr})
    return logs, p, ema


def warmup_lr(step: int, base_lr: float, warmup_steps: int = 1000) -> float:
    """LambdaLR of train_tokenizer.py:385-392, lr_lambda(step) = step / warmup if step < warmup else 1: the learning rate in force
    for optimiser step number `step` (0-based) -- the very first step runs at lr 0."""
    return base_lr * (step / warmup_steps if step < warmup_steps else 1.0)


# --------------------------------------------------------------------------------------------
# Build-defined distribution-matching statistics (NO reference counterpart: parity unpinned)
# --------------------------------------------------------------------------------------------
def kl_moment(z: Tensor) -> Tuple[Tensor, Tensor]:
    """Per-latent-channel KL( N(mu_c, sigma_c^2) || N(0,1) ) with batch moments over (B*T):
    0.5*(mu^2 + var - 1 - ln var), biased variance.  z: [B,T,C] -> (per-channel [C], mean scalar)."""
    zz = z.reshape(-1, z.shape[-1]).double()
    mu = zz.mean(dim=0)
    var = ((zz - mu) ** 2).mean(dim=0)
    kl = 0.5 * (mu * mu + var - 1.0 - torch.log(var))
    return kl.to(z.dtype), kl.mean().to(z.dtype)


def reparam_kl(moments: Tensor, eps: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor]:
    """Build-defined reparameterise hook + posterior-form KL (the reference's forward is deterministic, models/vae.py:90-98 --
    PARITY UNPINNED).  moments [..., 2C] = torch.chunk -> (mu, logvar); z = mu + exp(logvar / 2) * eps (eps None: z = mu);
    kl_c = mean over rows of 0.5 * (mu^2 + exp(logvar) - 1 - logvar).  -> (z [..., C], per-latent kl [C], mean scalar)."""
    mu, lv = moments.chunk(2, dim=-1)
    z = mu if eps is None else mu + torch.exp(0.5 * lv) * eps.reshape(mu.shape)
    kl = (0.5 * (mu * mu + torch.exp(lv) - 1.0 - lv)).reshape(-1, mu.shape[-1]).mean(dim=0)
    return z, kl, kl.mean()


MMD_BANDWIDTH_MULTS = (0.5, 1.0, 2.0, 4.0, 8.0)


def mmd_rbf(x: Tensor, y: Tensor) -> Tensor:
    """Biased RBF-mixture MMD^2 per group.  x: [G,n,d], y: [G,m,d] -> [G].
    k(a,b) = mean_j exp(-|a-b|^2 / (2 * mult_j * d));  MMD^2 = mean k(x,x) + mean k(y,y) - 2 mean k(x,y)."""
    d = x.shape[-1]

    def kmean(a, b):
        d2 = (a.double().unsqueeze(2) - b.double().unsqueeze(1)).pow(2).sum(-1)
        k = sum(torch.exp(-d2 / (2.0 * m * d)) for m in MMD_BANDWIDTH_MULTS) / len(MMD_BANDWIDTH_MULTS)
        return k.mean(dim=(1, 2))

    return (kmean(x, x) + kmean(y, y) - 2 * kmean(x, y)).to(x.dtype)


# --------------------------------------------------------------------------------------------
# toy_example_2d/sshpae.py
# --------------------------------------------------------------------------------------------
def sshape_sample(n: int, seed: int = 42, thickness: float = 0.06, diffusion: float = 0.03, amplitude: float = 0.85,
                  vertical_scale: float = 0.85, skew: float = 0.15) -> np.ndarray:
    """SShapeDistribution2D(random_state=seed, flip_y=True).sample(n)[0] (sshpae.py:29-71)."""
    rng = np.random.default_rng(seed)
    t = rng.uniform(-1.0, 1.0, size=n)
    pts = np.stack([amplitude * np.sin(np.pi * t), vertical_scale * t - skew * np.sin(2 * np.pi * t)], axis=1)
    tang = np.stack([amplitude * np.pi * np.cos(np.pi * t), vertical_scale - 2 * np.pi * skew * np.cos(2 * np.pi * t)], axis=1)
    tang /= np.linalg.norm(tang, axis=1, keepdims=True) + 1e-8
    ang = np.arctan2(tang[:, 0], -tang[:, 1])
    rad = rng.normal(0.0, thickness, size=n)
    pts[:, 0] += rad * np.cos(ang)
    pts[:, 1] += rad * np.sin(ang)
    pts += rng.normal(0.0, (diffusion * (0.4 + 0.6 * np.abs(t)))[:, None])
    pts[:, 1] *= -1
    return np.clip(pts, -1.0, 1.0)


# --------------------------------------------------------------------------------------------
# downstream consumers: SDE sampler (diffusion/transport/{integrators.py:8-77, transport.py:236-354, path.py:18-89}),
# image -> uint8 and the work split of sample_50k.py:128-157, the diffusion trainer's latent handling (train_diffusion.py:276-290)
# --------------------------------------------------------------------------------------------
def icplan_score_coeffs(t: Tensor) -> Tuple[Tensor, Tensor]:
    """(reverse_alpha_ratio, var) of ICPlan.get_score_from_velocity (path.py:82-88): alpha = t, d_alpha = 1, sigma = 1 - t, d_sigma = -1."""
    sigma = 1 - t
    rar = t / 1
    return rar, sigma ** 2 - rar * -1 * sigma


def icplan_diffusion(t: Tensor, form: str = "sigma", norm: float = 1.0) -> Tensor:
    """ICPlan.compute_diffusion (path.py:42-72) for the tensor-valued forms."""
    if form == "sigma":
        return norm * (1 - t)
    if form == "linear":
        return norm * (1 - t)
    if form == "SBDM":
        sigma = 1 - t
        return norm * ((1 / t) * (sigma ** 2) - sigma * -1)
    if form == "decreasing":
        return 0.25 * (norm * torch.cos(np.pi * t) + 1) ** 2
    if form == "inccreasing-decreasing":
        return norm * torch.sin(np.pi * t) ** 2
    raise NotImplementedError(form)


def sde_drift_from_velocity(v: Tensor, x: Tensor, t: Tensor, form: str = "sigma", norm: float = 1.0) -> Tensor:
    """Sampler's sde_drift for a velocity model (transport.py:254-257): v + diffusion(t) * score(v, x, t)."""
    te = expand_t(t, x)
    rar, var = icplan_score_coeffs(te)
    return v + icplan_diffusion(te, form, norm) * ((rar * v - x) / var)


def sde_euler_step(x: Tensor, v: Tensor, w: Tensor, t: Tensor, dt: Tensor, form: str = "sigma", norm: float = 1.0) -> Tuple[Tensor, Tensor]:
    """One Euler-Maruyama step (integrators.py:27-35) given the model output v: -> (x_new, mean_x)."""
    te = expand_t(t, x)
    mean = x + sde_drift_from_velocity(v, x, t, form, norm) * dt
    return mean + torch.sqrt(2 * icplan_diffusion(te, form, norm)) * (w * torch.sqrt(dt)), mean


def sde_interval(last_step: Optional[str], last_step_size: float, sample_eps: float = 0.0, form: str = "sigma") -> Tuple[float, float]:
    """Transport.check_interval(sde=True, eval=True) for the Linear path with a velocity model (transport.py:75-102)."""
    if last_step is None:
        last_step_size = 0.0
    t0 = sample_eps if form == "SBDM" else 0
    t1 = 1 - sample_eps if last_step_size == 0 else 1 - last_step_size
    return t0, t1


def sde_sample(init: Tensor, model: Callable, *, num_steps: int = 250, method: str = "Euler", form: str = "sigma", norm: float = 1.0,
               last_step: Optional[str] = "Mean", last_step_size: float = 0.04, sample_eps: float = 0.0) -> List[Tensor]:
    """Sampler.sample_sde(...)(init, model) (transport.py:298-354 + integrators.py:62-77): `num_steps` states.  `model(x, t)` -> velocity.
    Noise comes from the global CPU generator, one `torch.randn(x.size())` per step, like the reference."""
    t0, t1 = sde_interval(last_step, last_step_size, sample_eps, form)
    ts = torch.linspace(t0, t1, num_steps)
    dt = ts[1] - ts[0]
    drift = lambda x_, t_: sde_drift_from_velocity(model(x_, t_), x_, t_, form, norm)
    x, xs = init, []
    for ti in ts[:-1]:
        w = torch.randn(x.size()).to(x)
        t = torch.ones(x.size(0)).to(x) * ti
        if method == "Euler":
            x, _ = sde_euler_step(x, model(x, t), w, t, dt, form, norm)
        elif method == "Heun":
            xhat = x + torch.sqrt(2 * icplan_diffusion(expand_t(t, x), form, norm)) * (w * torch.sqrt(dt))
            k1 = drift(xhat, t)
            k2 = drift(xhat + dt * k1, t + dt)
            x = xhat + 0.5 * dt * (k1 + k2)
        else:
            raise NotImplementedError(method)
        xs.append(x)
    t = torch.ones(init.size(0)) * t1
    if last_step is None:
        pass
    elif last_step == "Mean":
        x = x + drift(x, t) * last_step_size
    elif last_step == "Euler":
        x = x + model(x, t) * last_step_size
    elif last_step == "Tweedie":
        score = (icplan_score_coeffs(expand_t(t, x))[0] * model(x, t) - x) / icplan_score_coeffs(expand_t(t, x))[1]
        x = x / t[0] + ((1 - t)[0] ** 2) / t[0] * score
    else:
        raise NotImplementedError(last_step)
    xs.append(x)
    return xs


def image_to_uint8(img: Tensor) -> Tensor:
    """sample_50k.py:151: NCHW float image in [-1, 1] -> [B, H, W, C] uint8 = trunc(clamp(127.5 x + 128, 0, 255))."""
    return torch.clamp(127.5 * img + 128.0, 0, 255).permute(0, 2, 3, 1).to(torch.uint8)


def sample50k_plan(num_fid_samples: int, num_classes: int, world_size: int, rank: int, n: int) -> Tuple[List[List[int]], List[List[int]]]:
    """Work split of sample_50k.py:128-157: per iteration the class labels of this rank's batch and the file indices its images are saved under
    (`f"{index:06d}.png"`; `total` is incremented BEFORE use, so indices start at n * world_size)."""
    labels = list(range(num_classes)) * (num_fid_samples // num_classes)
    per_rank = len(labels) // world_size
    mine = labels[per_rank * rank: per_rank * (rank + 1)]
    assert per_rank % n == 0
    ys, idx, total = [], [], 0
    for it in range(int(math.ceil(per_rank / n))):
        total += n * world_size
        ys.append(mine[it * n: (it + 1) * n])
        idx.append([j * world_size + rank + total for j in range(n)])
    return ys, idx


def latents_to_dit_input(tokens: Tensor, latent_mean: float, latent_scale: float) -> Tensor:
    """train_diffusion.py:279-287: [B, h*w, C] tokens -> (x - mean) * scale -> [B, C, h, w]."""
    x = (tokens - latent_mean) * latent_scale
    b, n, c = x.shape
    h = int(n ** 0.5)
    return x.reshape(b, h, h, c).permute(0, 3, 1, 2)


def diffusion_train_steps(latents: Tensor, labels: Tensor, p: P, trainable: Sequence[str], draws: Sequence[Tuple[Tensor, Tensor, Tensor]], num_heads: int,
                          num_classes: int, lr: float = 1e-4, max_norm: float = 1.0, ema_decay: float = 0.9999, q: Q = None,
                          on_grads: Optional[Callable[[int, Dict[str, Tensor]], None]] = None):
    """The loop body of train_diffusion.py:288-297 on given latents (the frozen encode + normalisation of :276-287 is `latents_to_dit_input`), once per entry of
    `draws` = (t, x0, dropped): transport.training_losses (transport.py:119-164: ICPlan xt / ut, the model in train mode -- a dropped label becomes the
    unconditional class `num_classes`, lightningdit.py:156-163 --, per-sample mean squared error) -> mean over the batch (:290) -> backward -> clip_grad_norm_ (:293)
    -> AdamW(betas (0.9, 0.95), eps 1e-8, weight_decay 0; :209) at a constant rate -> update_ema (:137-147).  p is updated functionally; returns
    (per-step (loss, gradient norm), parameters, EMA of the trainable parameters).  Pinned by tests/golden/diffusion_step_small.npz."""
    p = dict(p)
    m = {k: torch.zeros_like(p[k]) for k in trainable}
    v = {k: torch.zeros_like(p[k]) for k in trainable}
    ema = {k: p[k].detach().clone() for k in trainable}
    logs = []
    for step, (t, x0, dropped) in enumerate(draws):
        pp = {k: (val.detach().clone().requires_grad_(True) if k in trainable else val) for k, val in p.items()}
        y = torch.where(dropped, torch.full_like(labels, num_classes), labels)
        xt, _ = transport_plan(t, x0, latents)
        out = lightningdit_forward(xt, t, y, pp, num_heads, 1, q=q)
        loss = transport_loss(out, t, x0, latents).mean()
        grads = torch.autograd.grad(loss, [pp[k] for k in trainable])
        if on_grads is not None:
            on_grads(step, dict(zip(trainable, grads)))
        total, clipped = clip_grad_norm(grads, max_norm)
        for k, g_ in zip(trainable, clipped):
            p[k], m[k], v[k] = adamw_step(p[k], g_, m[k], v[k], step + 1, lr, wd=0.0)
            ema[k] = ema_update(ema[k], p[k], ema_decay)
        logs.append((float(loss.detach()), float(total)))
    return logs, p, ema


def dmd_train_steps(images: Tensor, labels: Tensor, p_vae: P, lpips_p: P, p_teacher: P, p_student: P, vae_trainable: Sequence[str],
                    student_trainable: Sequence[str], draws: Sequence[dict], *, vit_heads: int, dit_heads: int, num_classes: int, cfg: float, dmd_weight: float,
                    latent_mean: float, latent_scale: float, vae_train_every: int, lr: float, diff_lr: float, wd: float, warmup_steps: int,
                    max_norm: float = 1.0, q: Q = None, on_vae_grads=None, on_student_grads=None):
    """The loop body of train_dmd.py:506-575 (discriminator off), once per entry of `draws` = {"dmd": (t, x0) on VAE turns, "student": (t, x0, dropped)}:
    VAE turn every `vae_train_every`-th step -- VAE.forward with the encoder TRAINABLE (:518-520), latents_to_spatial((z - mean) * scale) (:525-526),
    forward_generator = L1 + LPIPS + dmd_weight * compute_distribution_matching_loss (:204-262: the four velocity evaluations under no_grad, so the gradient reaches
    the latents through the surrogate's mse alone), backward, clip_grad_norm_, AdamW(lr, wd) over every VAE parameter that received a gradient, LambdaLR counted in VAE
    turns (:541-544) -- else latents from the no-grad encode (:521-523); then the student's turn (:558-575): flow-matching loss on the detached latents with its
    label drop-out, clip, AdamW(diff_lr, wd), LambdaLR counted in steps.  Parameters are updated functionally; returns (per-step logs, VAE parameters, student
    parameters).  Pinned by tests/golden/dmd_step_small.npz."""
    pv, ps = dict(p_vae), dict(p_student)
    mv = {k: torch.zeros_like(pv[k]) for k in vae_trainable}
    vv = {k: torch.zeros_like(pv[k]) for k in vae_trainable}
    ms = {k: torch.zeros_like(ps[k]) for k in student_trainable}
    vs_ = {k: torch.zeros_like(ps[k]) for k in student_trainable}
    null = torch.full_like(labels, num_classes)
    logs, vae_steps = [], 0
    for step, d in enumerate(draws):
        log = {}
        if step % vae_train_every == 0:
            leaves = {k: pv[k].detach().clone().requires_grad_(True) for k in vae_trainable}
            pp = {**{k: t_.detach() for k, t_ in pv.items()}, **leaves}
            rec, z = vae_forward(images, pp, num_heads=vit_heads, q=q, return_latent=True)
            lat = latents_to_spatial((z - latent_mean) * latent_scale)
            rec_loss, rlog = forward_generator(images, rec, lpips_p, q=q)
            t, x0 = d["dmd"]
            with torch.no_grad():
                xt, _ = transport_plan(t, x0, lat.detach())
                vel = lambda prm, y: lightningdit_forward(xt, t, y, prm, dit_heads, 1, q=q)
                vt, vst = vel(p_teacher, labels), vel(ps, labels)
                vtu, vsu = (vel(p_teacher, null), vel(ps, null)) if cfg > 1 else (None, None)
            dl, gnorm, _ = dmd_loss(lat, t, x0, vt, vst, vtu, vsu, cfg=cfg)
            loss = rec_loss + dl * dmd_weight
            got = torch.autograd.grad(loss, [leaves[k] for k in vae_trainable], allow_unused=True)
            grads = {k: g_ for k, g_ in zip(vae_trainable, got) if g_ is not None}
            if on_vae_grads is not None:
                on_vae_grads(step, grads)
            total, clipped = clip_grad_norm(list(grads.values()), max_norm)
            vae_steps += 1
            rate = warmup_lr(vae_steps - 1, lr, warmup_steps)
            for k, g_ in zip(grads, clipped):
                pv[k], mv[k], vv[k] = adamw_step(pv[k].detach(), g_, mv[k], vv[k], vae_steps, rate, wd=wd)
            log.update({kk: float(val.detach()) for kk, val in rlog.items()})
            log.update({"dmd_loss": float(dl.detach()), "dmd_gradient_norm": float(gnorm), "vae_norm": float(total)})
            lat = lat.detach()
        else:
            with torch.no_grad():
                z = mlp_forward(dino_encoder_forward(images, pv, num_heads=vit_heads, q=q), pv, q=q)
                lat = latents_to_spatial((z - latent_mean) * latent_scale)
        t, x0, dropped = d["student"]
        y = torch.where(dropped, null, labels)
        sl = {k: ps[k].detach().clone().requires_grad_(True) for k in student_trainable}
        pps = {**{k: t_.detach() for k, t_ in ps.items()}, **sl}
        xt, _ = transport_plan(t, x0, lat)
        sloss = transport_loss(lightningdit_forward(xt, t, y, pps, dit_heads, 1, q=q), t, x0, lat).mean()
        sg = dict(zip(student_trainable, torch.autograd.grad(sloss, [sl[k] for k in student_trainable])))
        if on_student_grads is not None:
            on_student_grads(step, sg)
        stotal, sclipped = clip_grad_norm([sg[k] for k in student_trainable], max_norm)
        rate = warmup_lr(step, diff_lr, warmup_steps)
        for k, g_ in zip(student_trainable, sclipped):
            ps[k], ms[k], vs_[k] = adamw_step(ps[k].detach(), g_, ms[k], vs_[k], step + 1, rate, wd=wd)
        log.update({"diffusion_loss": float(sloss.detach()), "sit_norm": float(stotal)})
        logs.append(log)
    return logs, pv, ps


def dit_output_to_latents(samples: Tensor, latent_mean: float, latent_scale: float) -> Tensor:
    """sample_50k.py:143-148: [B, C, h, w] -> [B, h*w, C] tokens / scale + mean."""
    b, c, h, w = samples.shape
    return samples.permute(0, 2, 3, 1).reshape(b, h * w, c) / latent_scale + latent_mean


# --------------------------------------------------------------------------------------------
# train_tokenizer.py:324-367 eval() without the FID (evaluation/fid.py needs Inception weights that are not in the repo)
# --------------------------------------------------------------------------------------------
def psnr_sum(img1: Tensor, img2: Tensor) -> Tensor:
    """evaluation/metrics.py:6-13 with reduce="sum": images [B, C, H, W] in [0, 1]; -10 log10 of the per-sample mean squared error, summed over the batch."""
    mse = torch.mean((img1 - img2) ** 2, dim=(1, 2, 3))
    return torch.sum(-10 * torch.log10(mse))


def eval_metrics(encode: Callable, decode: Callable, batches, num_samples: int) -> dict:
    """The numbers eval() logs (train_tokenizer.py:340-360) at world size 1: per batch `latent = encode(x)`, `x_rec = decode(latent)`;
    latent_mean += latent.float().mean(); latent_scale += 1 / (latent.float().std() + 1e-8) (std over every element of the batch's latent, unbiased);
    psnr += PSNR((x_rec + 1) / 2, (x + 1) / 2, "sum"); then psnr / num_samples (the dataset's size, not the images seen) and the two latent sums / batches."""
    psnr = torch.zeros((), dtype=torch.float64)
    mean = torch.zeros((), dtype=torch.float64)
    scale = torch.zeros((), dtype=torch.float64)
    nb = 0
    for x in batches:
        lat = encode(x)
        rec = decode(lat)
        lf = lat.float()
        mean += lf.mean().double().cpu()
        scale += (1 / (lf.std() + 1e-8)).double().cpu()
        psnr += psnr_sum((rec + 1) / 2, (x + 1) / 2).double().cpu()
        nb += 1
    return {"PSNR": (psnr / num_samples).item(), "latent_mean": (mean / nb).item(), "latent_scale": (scale / nb).item()}
