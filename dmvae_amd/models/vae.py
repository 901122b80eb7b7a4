"""Drop-in counterpart of the reference's models/vae.py (VAE :71-121, DINOEncoder :34-53, MLP :56-68, Normalize /
Denormalize :10-31): same constructor, ``forward(x, freeze_encoder=False, return_latent=False)``, ``encode`` /
``decode``, ``load_pretrained`` and state_dict keys, so reference `vae.pt` checkpoints load with strict=True."""
import os
from contextlib import nullcontext

import torch
from torch import nn

from .. import functional as Fn
from .flux_ae import Decoder
from .init_param import init_weights
from .vit import create_model


class Normalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean).view(1, -1, 1, 1))
        self.register_buffer("std", torch.tensor(std).view(1, -1, 1, 1))

    def forward(self, x):
        return (x - self.mean) / self.std


class Denormalize(nn.Module):
    def __init__(self, mean, std):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean).view(1, -1, 1, 1))
        self.register_buffer("std", torch.tensor(std).view(1, -1, 1, 1))

    def forward(self, x):
        return x * self.std + self.mean


class DINOEncoder(nn.Module):
    def __init__(self, model_size="base", patch_size=16, image_size=256, pretrained=True, **vit_kw):
        super().__init__()
        self.dim = {"base": 768, "large": 1024}[model_size]
        self.de_scale = Denormalize(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5])
        self.scale = Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])
        name = {"base": "vit_base_patch14_dinov2.lvd142m", "large": "vit_large_patch14_dinov2.lvd142m"}[model_size]
        self.model = create_model(name, pretrained=pretrained, patch_size=patch_size, img_size=image_size, **vit_kw)
        if "embed_dim" in vit_kw:
            self.dim = vit_kw["embed_dim"]

    def forward(self, x):
        return self.model.forward_features(self.scale(self.de_scale(x)))[:, self.model.num_prefix_tokens:]


class MLP(nn.Module):
    def __init__(self, in_dim, out_dim, hidden_dim=2048):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.SiLU(), nn.Linear(hidden_dim, out_dim))

    def forward(self, x):
        shp = x.shape
        y = Fn.MLPFn.apply(x.reshape(-1, shp[-1]).to(torch.bfloat16).contiguous(), self.mlp[0].weight, self.mlp[0].bias,
                           self.mlp[2].weight, self.mlp[2].bias)
        return y.reshape(*shp[:-1], -1)

    def get_last_layer(self):
        return self.mlp[-1].weight


class VAE(nn.Module):
    def __init__(self, z_channels: int = 16, image_size: int = 256, model_size: str = "base", patch_size: int = 16,
                 conv_std_or_gain: float = 0.02, encoder_kwargs=None):
        super().__init__()
        # as in the reference (vae.py:81-82) the encoder ignores image_size and the decoder hyper-parameters are fixed
        self.encoder = DINOEncoder(model_size, patch_size=patch_size, **(encoder_kwargs or {}))
        self.decoder = Decoder(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, resolution=256, z_channels=16)
        self.decoder.post_init(z_channels=z_channels)
        self.bottle_neck = MLP(in_dim=self.encoder.dim, out_dim=z_channels)
        init_weights(self.decoder.conv_in, conv_std_or_gain)
        init_weights(self.bottle_neck, conv_std_or_gain)
        init_weights(self.decoder, conv_std_or_gain)

    def forward(self, x, freeze_encoder=False, return_latent=False):
        ctx = torch.no_grad() if freeze_encoder else nullcontext()
        with ctx:
            latent_tokens = self.encoder(x)
        latent_tokens = self.bottle_neck(latent_tokens)
        x_rec = self.decoder(latent_tokens)
        if return_latent:
            return x_rec.float(), latent_tokens
        return x_rec.float()

    @torch.inference_mode()
    def encode(self, x):
        return self.bottle_neck(self.encoder(x))

    @torch.inference_mode()
    def decode(self, latent_tokens):
        return self.decoder(latent_tokens)

    @torch.inference_mode()
    def decode_uint8(self, latent_tokens, round_bf16: bool = True):
        """decode + the uint8 conversion of sample_50k.py:149-151 in one go -> [B, H, W, 3] uint8 on the device (not in the reference's API)."""
        return self.decoder.forward_uint8(latent_tokens, round_bf16)

    def load_pretrained(self, state_dict_path, ema=False):
        if not os.path.exists(state_dict_path):
            print(f"[WARNING] VAE state_dict_path {state_dict_path} not found, skip loading")
            return
        try:
            ckpt = torch.load(state_dict_path, map_location="cpu")
        except Exception:
            ckpt = torch.load(state_dict_path, map_location="cpu", weights_only=False)
        if ema and "vae_ema" in ckpt:
            self.load_state_dict(ckpt["vae_ema"], strict=True)
        else:
            self.load_state_dict(ckpt["vae_wo_ddp"], strict=True)
