"""G-eval (`eval_metrics`): the reference's tokenizer `eval()` (train_tokenizer.py:324-367) run on the CPU with a stub FID and a cheap deterministic stand-in
for the VAE, recorded as its logged numbers.  It pins the arithmetic this build restates in `dmvae_amd/evaluate.py` and `oracle/ref_cpu.py::eval_metrics`:
PSNR summed per sample over [0, 1]-scaled images (evaluation/metrics.py:6-13) and divided by `num_samples`, `latent_mean` = mean over batches of
`latent.mean()`, `latent_scale` = mean over batches of `1 / (latent.std() + 1e-8)` (unbiased std over ALL elements of the batch's latent) -- the two numbers
readme.md:31 / scripts/sample50k.sh:14-15 quote and train_dmd.py / train_diffusion.py / sample_50k.py consume.

The VAE itself is pinned elsewhere (vae_forward_tiny, decoder_*, step_small); `eval()` only calls `.eval() / .encode() / .decode() / .train()` on it, so the
stand-in (`StandInVAE`, restated in tests/test_oracle_eval.py from the seeds stored here) keeps the fixture at a few hundred bytes.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_eval.py     (CPU, seconds)  -> tests/golden/eval_metrics.npz
"""
from __future__ import annotations

import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle.capture_golden import install_stubs, save  # noqa: E402

SEED_VAE, SEED_IMG = 91, 92
BATCHES = (3, 3, 2)            # a ragged last batch: per-batch means are NOT the mean over samples
NUM_SAMPLES = 8


class StandInVAE(torch.nn.Module):
    """encode: 16 x 16 average-pooled patches through a fixed [3, 32] map (+ offset) -> [B, 256, 32]; decode: tanh of a fixed [32, 768] map of the tokens,
    un-patchified to [B, 3, 256, 256].  Deterministic from `seed`; nothing about it matters except that both sides of the comparison run the same one."""

    def __init__(self, seed: int):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("m1", torch.randn(3, 32, generator=g) * 0.8)
        self.register_buffer("m2", torch.randn(32, 768, generator=g) * 0.6)

    def encode(self, x):
        t = torch.nn.functional.avg_pool2d(x, 16).flatten(2).transpose(1, 2)          # [B, 256, 3]
        return t @ self.m1 + 0.07

    def decode(self, lat):
        b = lat.shape[0]
        p = torch.tanh(lat @ self.m2).view(b, 16, 16, 3, 16, 16)                      # [B, gh, gw, c, ph, pw]
        return p.permute(0, 3, 1, 4, 2, 5).reshape(b, 3, 256, 256)


def batches(seed: int, sizes):
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(b, 3, 256, 256, generator=g) * 2 - 1, torch.zeros(b, dtype=torch.long)) for b in sizes]


class StubFID(torch.nn.Module):
    """FID needs Inception weights that are not in the repo (out of scope); eval() only needs .to / .update / .compute from it."""

    def update(self, imgs, real):
        pass

    def compute(self):
        return torch.tensor(float("nan"))


def main():
    install_stubs()
    import train_tokenizer
    logged = {}
    train_tokenizer.FID = StubFID
    train_tokenizer.wandb_log = lambda data, **kw: logged.update(data)
    vae = StandInVAE(SEED_VAE)
    data = SimpleNamespace(dataloader=batches(SEED_IMG, BATCHES), num_samples=NUM_SAMPLES)
    train_tokenizer.eval(SimpleNamespace(device="cpu"), vae, data, ema=False)
    assert vae.training, "eval() leaves the model in train mode (train_tokenizer.py:367)"
    out = {"seed_vae": np.array(SEED_VAE), "seed_img": np.array(SEED_IMG), "batches": np.array(BATCHES), "num_samples": np.array(NUM_SAMPLES),
           "PSNR": np.array(float(logged["PSNR"]), dtype=np.float64), "latent_mean": np.array(logged["latent_mean"], dtype=np.float64),
           "latent_scale": np.array(logged["latent_scale"], dtype=np.float64)}
    print({k: float(v) for k, v in out.items() if v.ndim == 0})
    save("eval_metrics", **out)


if __name__ == "__main__":
    main()
