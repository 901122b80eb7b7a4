"""Deterministic, construction-order-independent parameter fill used by the golden capture script and
by the tests (TEST INFRASTRUCTURE ONLY).  Each tensor is drawn from its own CPU generator seeded by
crc32(name) + base_seed, so the reference modules (in the build container) and this build's modules
(on the GPU box) receive bit-identical weights without shipping them."""
from __future__ import annotations

import math
import zlib

import torch


def det_tensor(name: str, shape, base_seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + base_seed) % (2 ** 31))
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if len(shape) <= 1 or leaf in ("cls_token", "pos_embed", "mask_token", "gamma"):
        if leaf == "weight" and ("norm" in name):           # norm scales
            return 1.0 + 0.2 * torch.randn(shape, generator=g)
        if leaf == "gamma":                                   # LayerScale
            return 0.5 + 0.1 * torch.randn(shape, generator=g)
        if leaf in ("cls_token", "pos_embed", "mask_token"):
            return 0.02 * torch.randn(shape, generator=g)
        return 0.1 * torch.randn(shape, generator=g)        # biases, norm shifts
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    return torch.randn(shape, generator=g) / math.sqrt(fan_in)


@torch.no_grad()
def det_fill_(module_or_sd, base_seed: int, prefix: str = "", skip=()):
    """In-place fill of every floating parameter/buffer-free parameter of a module (or dict)."""
    items = module_or_sd.named_parameters() if hasattr(module_or_sd, "named_parameters") else module_or_sd.items()
    for name, p in items:
        if any(name.startswith(s) for s in skip):
            continue
        p.copy_(det_tensor(prefix + name, p.shape, base_seed).to(p.dtype))
    return module_or_sd


@torch.no_grad()
def det_fill_patchgan_(sd, base_seed: int):
    """In-place fill of an NLayerDiscriminator state_dict (models/patchgan.py key layout main.{idx}.*): conv weights / biases and
    BatchNorm shifts from det_tensor, BatchNorm scales around 1, running variances in [0.5, ...), counters untouched."""
    for k, v in sd.items():
        if k.endswith("num_batches_tracked"):
            continue
        if k.endswith("running_var"):
            v.copy_(0.5 + det_tensor(k, v.shape, base_seed).abs() * 5)
        elif v.dim() == 1 and k.endswith("weight"):                  # BatchNorm scale (conv weights are 4-D)
            v.copy_(1.0 + det_tensor(k, v.shape, base_seed))
        else:
            v.copy_(det_tensor(k, v.shape, base_seed))
    return sd
