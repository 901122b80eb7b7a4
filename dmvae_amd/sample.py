"""The sampling loop of the reference's sample_50k.py (`sample`, :62-164) as a reusable pipeline: class labels split over ranks, noise -> SDE / ODE sampler
on LightningDiT -> tokens -> `vae.decode` -> uint8 -> `"{index:06d}.png"`.  FID / Inception (torch_fidelity, :171-209) are evaluation and not built.

Device work per batch: `num_sampling_steps` DiT forwards on the HIP kernels + one fused state update each (transport.py), then the flux decoder forward and
ONE conversion kernel to channels-last uint8 (`VAE.decode_uint8`), so the host receives 196 KB per image instead of the f32 NCHW tensor.  PNG encoding
(PIL, host) runs on a writer thread so it overlaps the next batch's sampling.  Ranks own disjoint label slices and file indices: no collective on the
data path."""
from __future__ import annotations

import math
import os
import queue
import threading
from typing import Iterator, List, Optional, Tuple

import torch

from .transport import Sampler, create_transport


def labels_and_indices(num_fid_samples: int, num_classes: int, world_size: int, rank: int, n: int) -> Tuple[List[List[int]], List[List[int]]]:
    """sample_50k.py:128-157: per iteration, this rank's class labels and the indices its images are saved under (the counter is advanced by
    `n * world_size` BEFORE it is used, like the script)."""
    label_list = list(range(num_classes)) * (num_fid_samples // num_classes)
    per_rank = len(label_list) // world_size
    mine = label_list[per_rank * rank: per_rank * (rank + 1)]
    assert per_rank % n == 0, "num_samples_per_rank must be divisible by per_proc_batch_size"
    ys, idx, total = [], [], 0
    for it in range(int(math.ceil(per_rank / n))):
        total += n * world_size
        ys.append(mine[it * n: (it + 1) * n])
        idx.append([j * world_size + rank + total for j in range(n)])
    return ys, idx


def dit_output_to_tokens(samples: torch.Tensor, latent_mean: float, latent_scale: float) -> torch.Tensor:
    """sample_50k.py:143-148 (patch 1): [B, C, h, w] -> [B, h*w, C] / latent_scale + latent_mean."""
    b, c, h, w = samples.shape
    return samples.permute(0, 2, 3, 1).reshape(b, h * w, c) / latent_scale + latent_mean


def tokens_to_dit_input(tokens: torch.Tensor, latent_mean: float, latent_scale: float) -> torch.Tensor:
    """train_diffusion.py:279-287: [B, h*w, C] tokens -> (x - mean) * scale -> [B, C, h, w]."""
    x = (tokens - latent_mean) * latent_scale
    b, n, c = x.shape
    h = int(n ** 0.5)
    assert h * h == n
    return x.reshape(b, h, h, c).permute(0, 3, 1, 2).contiguous()


class _PngWriter:
    """Encodes and writes PNGs on a host thread (PIL), bounded queue."""

    def __init__(self, depth: int = 4):
        self.q: "queue.Queue" = queue.Queue(maxsize=depth)
        self.err: Optional[BaseException] = None
        self.th = threading.Thread(target=self._run, daemon=True)
        self.th.start()

    def _run(self):
        from PIL import Image
        while True:
            item = self.q.get()
            if item is None:
                return
            try:
                arr, paths, ev = item
                if ev is not None:
                    ev.synchronize()
                for a, p in zip(arr.numpy(), paths):
                    Image.fromarray(a).save(p)
            except BaseException as e:          # surfaced by close()
                self.err = e

    def put(self, arr, paths, ev=None):
        self.q.put((arr, paths, ev))

    def close(self):
        self.q.put(None)
        self.th.join()
        if self.err is not None:
            raise self.err


class SamplePipeline:
    """`model`: LightningDiT (eval); `vae`: dmvae_amd.models.vae.VAE (eval).  Keyword names follow sample_50k.Args."""

    def __init__(self, model, vae, *, mode="SDE", sampling_method="Euler", num_sampling_steps=250, diffusion_form="sigma", diffusion_norm=1.0,
                 last_step="Mean", last_step_size=0.04, atol=1e-6, rtol=1e-3, reverse=False, cfg_scale=1.0, latent_mean=0.0, latent_scale=1.0,
                 path_type="Linear", prediction="velocity", loss_weight=None, train_eps=0.0, sample_eps=0.0, time_dist_shift=1.0, use_graph=True):
        assert cfg_scale >= 1.0, "In almost all cases, cfg_scale be >= 1.0"
        self.model, self.vae = model, vae
        self.latent_mean, self.latent_scale, self.cfg_scale = latent_mean, latent_scale, cfg_scale
        self.use_graph, self._graphed = use_graph, None      # the frozen DiT forward as one hipGraph replay per sampler step (models/lightningdit_fast.GraphedInference)
        transport = create_transport(path_type, prediction, loss_weight, train_eps, sample_eps, time_dist_shift=time_dist_shift)
        sampler = Sampler(transport)
        if mode == "ODE":
            self.sample_fn = sampler.sample_ode(sampling_method=sampling_method, num_steps=num_sampling_steps, atol=atol, rtol=rtol, reverse=reverse)
        else:
            self.sample_fn = sampler.sample_sde(sampling_method=sampling_method, diffusion_form=diffusion_form, diffusion_norm=diffusion_norm,
                                                last_step=last_step, last_step_size=last_step_size, num_steps=num_sampling_steps)

    @torch.no_grad()
    def latents(self, z: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        """noise [n, C, h, w] + labels [n] -> latent tokens [n, h*w, C] (sample_50k.py:138-148), under autocast(bf16) like the script."""
        with torch.autocast("cuda", dtype=torch.bfloat16):
            samples = self.sample_fn(z, self._model_fn(z, y), y=y)[-1]        # the script never enables guidance (cfg_scale stays 1.0, :79)
        return dit_output_to_tokens(samples.float(), self.latent_mean, self.latent_scale)

    def _model_fn(self, z, y):
        """`model.forward`, or its hipGraph replay when the model takes the HIP inference route at this shape (called under autocast)."""
        from .models import lightningdit_fast as fast
        if not (self.use_graph and z.is_cuda and hasattr(self.model, "blocks") and fast.supported(self.model, z)):
            return self.model.forward
        t = torch.zeros(z.shape[0], device=z.device, dtype=z.dtype)
        if self._graphed is None or not self._graphed.matches(z, t, y):
            self._graphed = fast.GraphedInference(self.model, z, t, y)
        return self._graphed

    @torch.no_grad()
    def images_uint8(self, z: torch.Tensor, y: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> ([n, H, W, 3] uint8 on the device, latent tokens)."""
        tok = self.latents(z, y)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return self.vae.decode_uint8(tok), tok

    def run(self, sample_dir: str, *, per_proc_batch_size=25, num_fid_samples=50000, num_classes=1000, rank=0, world_size=1, device="cuda",
            max_iterations: Optional[int] = None) -> int:
        """The loop of sample_50k.py:126-157 for this rank; returns the number of images written."""
        os.makedirs(sample_dir, exist_ok=True)
        n = per_proc_batch_size
        ys, idx = labels_and_indices(num_fid_samples, num_classes, world_size, rank, n)
        latent_size = int(round(self.model.x_embedder.num_patches ** 0.5)) * self.model.patch_size
        writer, done = _PngWriter(), 0
        try:
            for it, (yl, il) in enumerate(zip(ys, idx)):
                if max_iterations is not None and it >= max_iterations:
                    break
                z = torch.randn(n, self.model.in_channels, latent_size, latent_size, device=device)
                y = torch.tensor(yl, device=device)
                u8, _ = self.images_uint8(z, y)
                host = torch.empty(u8.shape, dtype=torch.uint8, pin_memory=True)
                host.copy_(u8, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                writer.put(host, [os.path.join(sample_dir, f"{i:06d}.png") for i in il], ev)
                done += n
        finally:
            writer.close()
        return done
