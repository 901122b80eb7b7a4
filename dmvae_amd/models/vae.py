"""Host-side VAE of the MI355X build: the module surface of the reference's models/vae.py (VAE :71-121, DINOEncoder :34-53,
MLP :56-68, Normalize / Denormalize :10-31) over the HIP kernels.

What is kept from the reference is the CONTRACT, because its scripts and checkpoints depend on it: class names, constructor
arguments, ``forward(x, freeze_encoder=False, return_latent=False)``, ``encode`` / ``decode`` under inference mode,
``load_pretrained(path, ema)`` with its warn-and-return on a missing file, ``get_last_layer`` and the state_dict keys
(``encoder.{scale,de_scale}.{mean,std}``, ``encoder.model.*``, ``bottle_neck.mlp.{0,2}.*``, ``decoder.*``), so a reference
``vae.pt`` loads with strict=True.  What differs is everything underneath: the bottleneck runs as one autograd Function over
two MFMA GEMMs (functional.MLPFn), the decoder is NHWC bf16 end to end (models/flux_ae.py), and ``decode_uint8`` goes
straight from the output conv to bytes.

Beyond the reference (BUILD-DEFINED, off by default): ``VAE(..., reparameterize=True)`` gives the bottleneck a (mu | logvar) head and samples
z = mu + exp(logvar / 2) * eps between encoder and decoder, with the posterior-form KL 0.5 * (mu^2 + exp(logvar) - 1 - logvar) per latent (SURVEY.md 0 and
8a row a15; BASELINE.json's "encoder -> reparameterise -> decoder").  The reference's forward is deterministic (vae.py:90-98); with the keyword at its
default nothing of this is constructed or run and forward() is the reference's, bit for bit.
"""
import os

import torch
from torch import nn

from .. import functional as Fn
from .flux_ae import Decoder
from .init_param import init_weights
from .vit import create_model

# ImageNet statistics the DINOv2 encoder was trained with, and the [-1, 1] <-> [0, 1] map of the data pipeline (reference vae.py:39-40)
_IMAGENET_MEAN, _IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
_HALF = (0.5, 0.5, 0.5)
# model_size -> (token width, timm checkpoint name); the name only matters to the ViT factory (models/vit.py)
_ENCODERS = {
    "base": (768, "vit_base_patch14_dinov2.lvd142m"),
    "large": (1024, "vit_large_patch14_dinov2.lvd142m"),
}
# the reference hard-codes the decoder (vae.py:82): 128 base channels, multipliers (1, 2, 4, 4), two res-blocks per level, 256 x 256 RGB
_DECODER_CFG = dict(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, resolution=256, z_channels=16)


class _PerChannelStats(nn.Module):
    """Holds the per-channel `mean` / `std` buffers ([1, C, 1, 1], part of the checkpoint) shared by the two affine maps below."""

    def __init__(self, mean, std):
        super().__init__()
        for name, values in (("mean", mean), ("std", std)):
            self.register_buffer(name, torch.as_tensor(values, dtype=torch.float32).reshape(1, len(values), 1, 1))


class Normalize(_PerChannelStats):
    """x -> (x - mean) / std"""

    def forward(self, x):
        return (x - self.mean) / self.std


class Denormalize(_PerChannelStats):
    """x -> x * std + mean (inverse of Normalize with the same statistics)"""

    def forward(self, x):
        return x * self.std + self.mean


class DINOEncoder(nn.Module):
    """[-1, 1] images -> patch tokens of a DINOv2 ViT (prefix tokens dropped).  `vit_kw` (embed_dim, depth, num_heads ...) builds reduced
    encoders for tests; the reference always instantiates the checkpoint geometry."""

    def __init__(self, model_size="base", patch_size=16, image_size=256, pretrained=True, **vit_kw):
        super().__init__()
        width, ckpt_name = _ENCODERS[model_size]
        self.dim = vit_kw.get("embed_dim", width)
        self.de_scale = Denormalize(mean=_HALF, std=_HALF)                      # [-1, 1] -> [0, 1]
        self.scale = Normalize(mean=_IMAGENET_MEAN, std=_IMAGENET_STD)          # [0, 1] -> ImageNet-normalised
        self.model = create_model(ckpt_name, pretrained=pretrained, patch_size=patch_size, img_size=image_size, **vit_kw)

    def preprocess(self, x):
        return self.scale(self.de_scale(x))

    def forward(self, x):
        tokens = self.model.forward_features(self.preprocess(x))
        return tokens[:, self.model.num_prefix_tokens:]


class MLP(nn.Module):
    """Linear -> SiLU -> Linear bottleneck from encoder tokens to latent channels; both GEMMs, the activation and their backward are one
    autograd Function on the HIP GEMM kernels (bf16 operands, f32 accumulation)."""

    def __init__(self, in_dim, out_dim, hidden_dim=2048):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.SiLU(), nn.Linear(hidden_dim, out_dim))

    def forward(self, x):
        fc1, fc2 = self.mlp[0], self.mlp[2]
        lead = x.shape[:-1]
        rows = x.reshape(-1, x.shape[-1]).to(Fn.parity.act_dtype()).contiguous()
        return Fn.MLPFn.apply(rows, fc1.weight, fc1.bias, fc2.weight, fc2.bias).reshape(*lead, -1)

    def get_last_layer(self):
        return self.mlp[2].weight


def _read_checkpoint(path):
    """torch.load with the reference's fallback (vae.py:114-117): retry with weights_only=False when the safe loader refuses the file."""
    try:
        return torch.load(path, map_location="cpu")
    except Exception:
        return torch.load(path, map_location="cpu", weights_only=False)


class VAE(nn.Module):
    def __init__(self, z_channels: int = 16, image_size: int = 256, model_size: str = "base", patch_size: int = 16,
                 conv_std_or_gain: float = 0.02, encoder_kwargs=None, reparameterize: bool = False):
        super().__init__()
        # build-defined hook, NOT in the reference (module docstring): False = the reference's deterministic forward, same modules, same generator consumption
        self.reparameterize = bool(reparameterize)
        self.z_channels = z_channels
        self.posterior_kl = None               # scalar (differentiable) and per-latent [z + 1] KL of the last training-mode forward with the hook on
        self.posterior_kl_per_latent = None
        self.reparam_generator = None          # optional torch.Generator (device) for eps
        # construction order = the reference's (encoder, decoder, post_init, bottleneck, three init_weights calls): under a fixed seed the random
        # initialisation then consumes the generator identically.  As there, `image_size` does not reach the encoder.
        self.encoder = DINOEncoder(model_size, patch_size=patch_size, **(encoder_kwargs or {}))
        self.decoder = Decoder(**_DECODER_CFG)
        self.decoder.post_init(z_channels=z_channels)
        self.bottle_neck = MLP(in_dim=self.encoder.dim, out_dim=2 * z_channels if self.reparameterize else z_channels)
        for part in (self.decoder.conv_in, self.bottle_neck, self.decoder):
            init_weights(part, conv_std_or_gain)

    # ---- training-time forward --------------------------------------------------------------------------------------------------
    def tokens(self, x, freeze_encoder=False):
        """Latent tokens [B, 256, z]; with `freeze_encoder` the ViT runs without a graph (train_tokenizer.py keeps it frozen)."""
        with torch.set_grad_enabled(torch.is_grad_enabled() and not freeze_encoder):
            feats = self.encoder(x)
        return self.latent(feats)

    def latent(self, feats, eps=None):
        """Encoder tokens [B, T, width] -> latent tokens [B, T, z].  Hook off (default): the bottleneck's output, as in the reference (vae.py:94).  Hook on:
        the bottleneck emits (mu | logvar); in training mode z = mu + exp(logvar / 2) * eps (eps drawn here unless given) and `posterior_kl` /
        `posterior_kl_per_latent` hold the KL of this call; in eval mode z = mu (the posterior mode) and eps is not drawn."""
        out = self.bottle_neck(feats)
        if not self.reparameterize:
            return out
        lead, zc = out.shape[:-1], self.z_channels
        moments = out.reshape(-1, 2 * zc)
        if self.training and eps is None:
            eps = torch.randn(moments.shape[0], zc, device=moments.device, dtype=torch.float32, generator=self.reparam_generator)
        elif eps is not None:
            eps = eps.reshape(-1, zc).float().contiguous()
        z, kl, per_latent = Fn.ReparamKLFn.apply(moments, eps)
        self.posterior_kl, self.posterior_kl_per_latent = kl, per_latent
        return z.reshape(*lead, zc)

    def forward(self, x, freeze_encoder=False, return_latent=False):
        latent_tokens = self.tokens(x, freeze_encoder)
        recon = self.decoder(latent_tokens).float()
        return (recon, latent_tokens) if return_latent else recon

    # ---- inference entry points --------------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def encode(self, x):
        return self.tokens(x)

    @torch.inference_mode()
    def decode(self, latent_tokens):
        return self.decoder(latent_tokens)

    @torch.inference_mode()
    def decode_uint8(self, latent_tokens, round_bf16: bool = True):
        """decode + the uint8 conversion of sample_50k.py:149-151 in one go -> [B, H, W, 3] uint8 on the device (not in the reference's API)."""
        return self.decoder.forward_uint8(latent_tokens, round_bf16)

    # ---- checkpoints ------------------------------------------------------------------------------------------------------------
    def load_pretrained(self, state_dict_path, ema=False):
        """`vae.pt` of the reference: {'vae_wo_ddp': sd, 'vae_ema': sd (optional), ...}; a missing file is a warning, not an error."""
        if not os.path.exists(state_dict_path):
            print(f"[WARNING] VAE state_dict_path {state_dict_path} not found, skip loading")
            return
        ckpt = _read_checkpoint(state_dict_path)
        which = "vae_ema" if (ema and "vae_ema" in ckpt) else "vae_wo_ddp"
        self.load_state_dict(ckpt[which], strict=True)
