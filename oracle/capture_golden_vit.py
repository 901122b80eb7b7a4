"""Golden vectors for the ViT encoder at widths the HIP encoder kernels accept (TEST INFRASTRUCTURE; run in the build container only).

The reference builds its encoder through timm (models/vae.py:47-53), which is not installed here; its vendored models/dinov2.py has the same block
algebra and sub-module names.  This script imports THAT module from /root/reference with the import stubs of capture_golden.py,
fills it with name-seeded weights (oracle/detweights.py: the GPU box regenerates them bit for bit) and records

  vit_w256 : embed 256, 4 heads x 64, 2 blocks, 257 tokens (patch 16 @ 256^2), batch 2 -- forward_features output (class + patch tokens after the
             final norm, f32), and the backward of sum(out * dy): input-image gradient slice, gradient norm of EVERY parameter, the full gradient
             of a handful of small tensors;
  vit_w768 : embed 768, 12 heads x 64 (ViT-B, the reference's `model_size='base'`, models/vae.py:41-48), 1 block, batch 1 -- forward only.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_vit.py"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import capture_golden as cg  # noqa: E402
from oracle.detweights import det_fill_  # noqa: E402

FULL = ("cls_token", "norm.weight", "norm.bias", "blocks.0.ls1.gamma", "blocks.1.ls2.gamma", "blocks.0.attn.qkv.bias", "blocks.1.norm2.weight",
        "blocks.0.attn.proj.bias", "blocks.1.mlp.fc2.bias", "blocks.0.norm1.bias")


def tokens(vit, x):
    d = vit.forward_features(x)
    return torch.cat([d["x_norm_clstoken"][:, None], d["x_norm_patchtokens"]], 1)


def main():
    cg.install_stubs()
    dinov2 = importlib.import_module("models.dinov2")
    torch.set_grad_enabled(True)
    kw = dict(patch_size=16, img_size=256, init_values=1e-5, block_chunks=0)

    vit = dinov2.DinoVisionTransformer(embed_dim=256, depth=2, num_heads=4, mlp_ratio=4, **kw)
    det_fill_(vit, 77)
    x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(78)) * 2 - 1).requires_grad_(True)
    out = tokens(vit, x)
    dy = torch.randn(out.shape, generator=torch.Generator().manual_seed(79))
    (out * dy).sum().backward()
    grads = {n: p.grad for n, p in vit.named_parameters() if p.grad is not None}
    cg.save("vit_w256", seed=np.array(77), x_seed=np.array(78), dy_seed=np.array(79), heads=np.array(4), out=out.detach(),
            dx_slice=x.grad[:, :, ::16, ::16], dx_norm=np.array(x.grad.double().norm().item()),
            names=np.array(list(grads)), gnorm=np.array([grads[n].double().norm().item() for n in grads]),
            **{"g." + n: grads[n] for n in FULL})

    vit = dinov2.DinoVisionTransformer(embed_dim=768, depth=1, num_heads=12, mlp_ratio=4, **kw)
    det_fill_(vit, 87)
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(88)) * 2 - 1
    with torch.no_grad():
        out = tokens(vit, x)
    cg.save("vit_w768", seed=np.array(87), x_seed=np.array(88), heads=np.array(12), out=out)


if __name__ == "__main__":
    main()
