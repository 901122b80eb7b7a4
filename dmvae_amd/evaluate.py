"""The tokenizer's `eval()` (train_tokenizer.py:324-367, train_dmd.py:419-461) without the FID: reconstruction PSNR and the latent statistics
`latent_mean` / `latent_scale` that the later stages normalise the latents with (`(z - latent_mean) * latent_scale`: train_dmd.py:525-526,
train_diffusion.py:279-286, sample_50k.py:143-148; readme.md:31 quotes 0.0685 / 0.1763 for the released tokenizer).  `latent_stats(result)` turns the
result into the keyword arguments of `train.DMDTrainer`, `train.DiffusionTrainer` and `sample.SamplePipeline`.

What is computed, exactly as the reference does (it is NOT the mean / std over the dataset): per batch `latent.float().mean()` and
`1 / (latent.float().std() + 1e-8)` (unbiased std over every element of that batch's latent), summed over the batches of every rank (all-reduce) and
divided by the number of batches; PSNR per image on [0, 1]-scaled images (evaluation/metrics.py:6-13), summed over every rank's images and divided by
`num_samples` -- the DATASET's size, as the reference divides, whatever the loaders delivered.  The model's `encode` / `decode` run the HIP path under
autocast(bf16) like the reference's (`train_tokenizer.py:342`); the reductions are a handful of small device ops per batch, every 10 000 steps.
FID needs torchmetrics' Inception weights, which neither the reference repo nor this image holds: `fid` takes any object with the
reference's `update(images01, real: bool)` / `compute()` protocol (evaluation/fid.py:51-65) and is skipped when None."""
from __future__ import annotations

from typing import Iterable, Optional

import torch

from . import dist


def psnr(img1: torch.Tensor, img2: torch.Tensor, reduce: str = "sum") -> torch.Tensor:
    """evaluation/metrics.py:6-13: [B, C, H, W] images scaled to [0, 1] -> -10 log10(per-sample MSE), summed or averaged over the batch."""
    mse = torch.mean((img1 - img2) ** 2, dim=(1, 2, 3))
    v = -10 * torch.log10(mse)
    if reduce == "sum":
        return torch.sum(v)
    if reduce == "mean":
        return torch.mean(v)
    raise ValueError(f"reduce must be 'sum' or 'mean', got {reduce!r}")


@torch.inference_mode()
def evaluate(vae, loader: Iterable, num_samples: int, device=None, fid=None, autocast: Optional[bool] = None) -> dict:
    """`loader` yields `(images, labels)` pairs (the reference's eval_data.dataloader) or bare image batches, images [B, 3, H, W] in [-1, 1];
    `num_samples` is the size of the whole evaluation set (eval_data.num_samples).  Every rank runs its shard of the loader; the four sums are
    all-reduced like train_tokenizer.py:357.  Leaves `vae` in train mode, like the reference (:367).  Returns python floats:
    {"PSNR", "latent_mean", "latent_scale", "FID" (None without `fid`), "batches", "images"}."""
    vae.eval()
    if device is None:
        p = next(vae.parameters(), None)
        device = p.device if p is not None else torch.device("cpu")
    device = torch.device(device)
    if autocast is None:
        autocast = device.type == "cuda"
    acc = torch.zeros(4, dtype=torch.float64, device=device)        # psnr sum, latent_mean sum, latent_scale sum, batches (+ images, local only)
    images = 0
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bool(autocast)):
        for sample in loader:
            if isinstance(sample, (tuple, list)):
                sample = sample[0]
            sample = sample.to(device, non_blocking=True)
            latent = vae.encode(sample)
            x_rec = vae.decode(latent)
            lf = latent.float()
            acc[1] += lf.mean().double()
            acc[2] += (1 / (lf.std() + 1e-8)).double()
            acc[3] += 1
            x01, s01 = (x_rec + 1) / 2, (sample + 1) / 2
            acc[0] += psnr(x01, s01, reduce="sum").double()
            images += int(sample.shape[0])
            if fid is not None:
                fid.update(x01, False)
                fid.update(s01, True)
    if dist.initialized() and dist.get_world_size() > 1:
        dist.allreduce(acc)
        dist.barrier()
    v = acc.tolist()                                                # the single D2H copy
    nb = max(v[3], 1.0)
    out = {"PSNR": v[0] / num_samples, "latent_mean": v[1] / nb, "latent_scale": v[2] / nb, "FID": None, "batches": int(v[3]), "images": images}
    if fid is not None:
        f = fid.compute()
        out["FID"] = float(f.item() if hasattr(f, "item") else f)
    vae.train()
    return out


def latent_stats(result: dict) -> dict:
    """`evaluate`'s result -> `dict(latent_mean=..., latent_scale=...)`, the constructor arguments of train.DMDTrainer / train.DiffusionTrainer /
    sample.SamplePipeline (the reference passes them on the command line: scripts/train_dmd.sh, scripts/sample50k.sh:14-15)."""
    return {"latent_mean": float(result["latent_mean"]), "latent_scale": float(result["latent_scale"])}
