"""Host-side mirror of the reference's `models` package for the hot path (models/vae.py, models/flux_ae.py)."""
from .vae import VAE, DINOEncoder, MLP, Normalize, Denormalize  # noqa: F401
from .flux_ae import (AttnBlock, AutoEncoderParams, Decoder, Downsample, Encoder, ResnetBlock, Upsample, swish)  # noqa: F401
