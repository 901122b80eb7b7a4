"""Drop-in counterpart of the reference's utils/diffaug.py (DiffAug :22-114): same constructor and ``aug(BCHW, warmup_blur_schedule)``,
same consumption of the global RNG streams (``torch.rand(3)`` on the CPU generator for the three coin flips, then
``torch.rand(7, B, 1, 1, device=...)`` on the device generator, :66-69), so a seeded run draws the same augmentations.

Translation, colour and cut-out run in one fused HIP pass each way (csrc/diffaug.hip) instead of ~25 ATen launches.  The Gaussian
warm-up blur (:47-63) is not built: every call site in the reference's scripts passes schedule 0 (train_tokenizer.py:192,212,217)."""
from __future__ import annotations

import torch

from .. import ops


class _DiffAugFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rand01, flags, cutout):
        ctx.save_for_backward(rand01)
        ctx.cfg = (flags, cutout)
        return ops.diffaug(x.contiguous(), rand01, flags, cutout)

    @staticmethod
    def backward(ctx, dy):
        (rand01,) = ctx.saved_tensors
        flags, cutout = ctx.cfg
        return ops.diffaug_bwd(dy.contiguous().float(), rand01, flags, cutout), None, None, None


class DiffAug(object):
    def __init__(self, prob=1.0, cutout=0.2):
        self.prob = abs(prob)
        self.using_cutout = prob > 0
        self.cutout = cutout

    def aug(self, BCHW: torch.Tensor, warmup_blur_schedule: float = 0) -> torch.Tensor:
        if BCHW.dtype != torch.float32:
            BCHW = BCHW.float()
        if warmup_blur_schedule > 0:
            raise NotImplementedError("DiffAug warm-up blur is not built (the reference never enables it: fade_blur_schedule is always 0)")
        if self.prob < 1e-6:
            return BCHW
        trans, color, cut = torch.rand(3) <= self.prob
        trans, color, cut = trans.item(), color.item(), cut.item()
        if not (trans or color or cut):
            return BCHW
        B = BCHW.shape[0]
        rand01 = torch.rand(7, B, 1, 1, device=BCHW.device)
        flags = (1 if trans else 0) | (2 if color else 0) | (4 if (self.using_cutout and cut) else 0)
        return _DiffAugFn.apply(BCHW, rand01.view(7, B), flags, self.cutout)
