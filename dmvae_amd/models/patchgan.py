"""Drop-in counterpart of the reference's models/patchgan.py (NLayerDiscriminator :99-151, weights_init :90-96):
same constructor, ``state_dict`` keys (``main.{0,2,3,5,6,8,9,11}.*`` incl. the BatchNorm running statistics) and
``forward(input, grad_ckpt=False) -> [B, 1, 30, 30]`` logits for 256x256 inputs.

The five 4x4 convolutions (stride 2,2,2,1,1, padding 1) are strided 16-tap gathers in the implicit-GEMM conv kernel (forward, weight
gradient, and the transposed gather for the input gradient) with the bias and the first LeakyReLU(0.2) in the epilogue; BatchNorm / SyncBatchNorm + LeakyReLU run on the GroupNorm kernels (one
"image" of N*H*W pixels, one channel per group) with the cross-rank combination of the statistics done the way
torch.nn.SyncBatchNorm does it (all-gather of per-rank mean / var / count forward, all-reduce of the two per-channel sums
backward).  ``ActNorm`` (reference :5-87) is never enabled by the reference's scripts (`use_actnorm=False` everywhere) and is
not provided."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import ops


def weights_init(m):
    """Reference :90-96."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif classname.find("BatchNorm") != -1:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)


def sync_batch_stats(mean: torch.Tensor, var: torch.Tensor, count: int, group=None):
    """Combine per-rank per-channel (mean, biased variance, element count) into the statistics of the union of all ranks' batches, the
    way torch.nn.SyncBatchNorm does (ranks may hold different counts): one all-reduce of [sum x, sum x^2, n].  Returns (mean, var, n)."""
    c = mean.numel()
    packed = torch.cat([mean * count, (var + mean * mean) * count, torch.full((1,), float(count), device=mean.device, dtype=mean.dtype)])
    torch.distributed.all_reduce(packed, group=group)
    total = packed[-1]
    g_mean = packed[:c] / total
    g_var = (packed[c:2 * c] / total - g_mean * g_mean).clamp_min_(0.0)
    return g_mean, g_var, int(round(total.item()))


def _bn_stats(bn: nn.modules.batchnorm._BatchNorm, x: torch.Tensor):
    """Per-channel (mean, rstd) [1,C,2], whether they are batch statistics, the global element count and the process group to
    reduce the backward sums over.  Training: batch statistics (over all ranks for SyncBatchNorm) and the running-estimate update
    of nn.BatchNorm2d (momentum, unbiased variance).  Eval: the running estimates."""
    c = x.shape[-1]
    count = x.numel() // c
    use_batch = bn.training or not bn.track_running_stats
    if not use_batch:
        st = torch.stack([bn.running_mean.float(), torch.rsqrt(bn.running_var.float() + bn.eps)], dim=1).view(1, c, 2).contiguous()
        return st, False, count, None
    st = ops.groupnorm_stats(x.view(1, -1, c), groups=c, eps=bn.eps)
    group = None
    synced = isinstance(bn, nn.SyncBatchNorm) and torch.distributed.is_available() and torch.distributed.is_initialized() \
        and torch.distributed.get_world_size(bn.process_group) > 1
    if not synced and bn.running_mean is not None and bn.running_mean.dtype == torch.float32 and bn.momentum is not None:
        # single rank: the whole running-estimate update as ONE launch (was nine 4-us ATen launches per layer and discriminator pass)
        if bn.training and bn.track_running_stats:
            with torch.no_grad():
                bn.num_batches_tracked += 1
                ops.batchnorm_running_update(st, bn.running_mean, bn.running_var, bn.eps, bn.momentum, count / max(count - 1, 1))
        return st, True, count, None
    with torch.no_grad():
        mean, rstd = st[0, :, 0], st[0, :, 1]
        var = (1.0 / (rstd * rstd) - bn.eps).clamp_min_(0.0)           # biased variance of this rank's batch
        total = count
        if isinstance(bn, nn.SyncBatchNorm) and torch.distributed.is_available() and torch.distributed.is_initialized() \
                and torch.distributed.get_world_size(bn.process_group) > 1:
            group = bn.process_group if bn.process_group is not None else torch.distributed.group.WORLD
            mean, var, total = sync_batch_stats(mean, var, count, group)
            st = torch.stack([mean, torch.rsqrt(var + bn.eps)], dim=1).view(1, c, 2).contiguous()
        if bn.training and bn.track_running_stats:
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
            bn.running_var.mul_(1 - mom).add_(var * (total / max(total - 1, 1)), alpha=mom)
    return st, True, total, group


class NLayerDiscriminator(nn.Module):
    """PatchGAN discriminator as in Pix2Pix (reference :99-151)."""

    def __init__(self, input_nc=3, ndf=64, n_layers=3, use_actnorm=False, use_syncbn=True):
        super().__init__()
        if use_actnorm:
            raise NotImplementedError("ActNorm is not built (the reference never enables it: train_tokenizer.py:316, train_dmd.py:398)")
        norm_layer = nn.SyncBatchNorm if use_syncbn else nn.BatchNorm2d
        use_bias = norm_layer != nn.BatchNorm2d          # reference :119-122
        kw, padw = 4, 1
        sequence = [nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev, nf_mult = nf_mult, min(2 ** n, 8)
            sequence += [nn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias),
                         norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        nf_mult_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        sequence += [nn.Conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                     norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        sequence += [nn.Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.main = nn.Sequential(*sequence)

    def forward(self, input, grad_ckpt=False):
        mods = list(self.main)
        h = Fn.ImageToNhwcFn.apply(input, (input.shape[1] + 31) // 32 * 32)      # channels zero-padded to the conv kernel's 32-channel K step
        i = 0
        while i < len(mods):
            conv = mods[i]
            assert isinstance(conv, nn.Conv2d)
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            last = nxt is None
            fuse_act = isinstance(nxt, nn.LeakyReLU)
            assert conv.kernel_size == (4, 4) and conv.padding == (1, 1) and conv.stride[0] in (1, 2)
            h = Fn.ConvK4Fn.apply(h, conv.weight, conv.bias, conv.stride[0], ops.ACT_LEAKY if fuse_act else ops.ACT_NONE, last)
            i += 2 if fuse_act else 1
            if isinstance(nxt, nn.modules.batchnorm._BatchNorm):
                act = 2 if isinstance(mods[i + 1] if i + 1 < len(mods) else None, nn.LeakyReLU) else 0
                st, batch_stats, count, group = _bn_stats(nxt, h)
                h = Fn.BatchNormActFn.apply(h, nxt.weight, nxt.bias, st, act, batch_stats, count, group)
                i += 2 if act else 1
        return h.permute(0, 3, 1, 2).contiguous()          # [B, 1, Ho, Wo] f32 logits
