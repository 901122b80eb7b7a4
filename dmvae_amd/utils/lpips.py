"""Drop-in counterpart of the reference's utils/lpips.py (LPIPS :51-94, ScalingLayer :97-104, NetLinLayer :107-113,
vgg16 :116-153): same class names, state_dict keys (net.slice{1..5}.{idx}.*, lin{0..4}.model.1.weight) and
``forward(input, target) -> scalar``.

The feature-difference reduction (normalize_tensor, diff^2, 1x1 lin, spatial mean, level sum, batch mean;
:86-94,156-162) runs in the fused HIP kernel dmvae_lpips_diff (forward value + gradient w.r.t. the second
argument's features in one pass over NHWC bf16 features).  The VGG16 trunk itself is a SURVEY.md 8(f) "next" row:
stock PyTorch-ROCm convs in channels-last bf16 (torchvision is not installed here, so the 'D' layer list is built
locally; pretrained trunk weights must be loaded from a checkpoint -- they are not downloadable offline)."""
import torch
import torch.nn as nn

from .. import ops

_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512)
_SLICE_BOUNDS = (4, 9, 16, 23, 30)


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.Tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, inp):
        return (inp - self.shift) / self.scale


class NetLinLayer(nn.Module):
    """A single linear layer which does a 1x1 conv (weights only; applied inside the fused kernel)."""

    def __init__(self, chn_in, chn_out=1, use_dropout=False):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, chn_out, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class vgg16(nn.Module):
    """VGG16 'D' features split at relu1_2 / relu2_2 / relu3_3 / relu4_3 / relu5_3 like the reference's slices."""

    def __init__(self, requires_grad=False, pretrained=True):
        super().__init__()
        self.N_slices = 5
        slices = [nn.Sequential() for _ in range(5)]
        idx, c = 0, 3
        for v in _CFG:
            sl = sum(idx >= b for b in _SLICE_BOUNDS)
            if v == "M":
                slices[sl].add_module(str(idx), nn.MaxPool2d(2, 2))
                idx += 1
            else:
                slices[sl].add_module(str(idx), nn.Conv2d(c, v, 3, padding=1))
                slices[sl].add_module(str(idx + 1), nn.ReLU(inplace=True))
                c = v
                idx += 2
        self.slice1, self.slice2, self.slice3, self.slice4, self.slice5 = slices
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, x):
        outs = []
        h = x
        for sl in (self.slice1, self.slice2, self.slice3, self.slice4, self.slice5):
            h = sl(h)
            outs.append(h)
        return outs


class _LpipsDiff(torch.autograd.Function):
    """sum_l mean_{n,hw} sum_c w_c (f0_hat - f1_hat)^2 over the five levels; gradient flows to feats1 only
    (the reference calls lpips(images, recon): the first argument carries no graph)."""

    @staticmethod
    def forward(ctx, lin_ws, *feats):
        k = len(feats) // 2
        f0s, f1s = feats[:k], feats[k:]
        out = torch.zeros(1, dtype=torch.float32, device=f0s[0].device)
        need = any(f.requires_grad for f in f1s)
        grads = []
        for i in range(k):
            a = f0s[i].detach().permute(0, 2, 3, 1)
            b = f1s[i].detach().permute(0, 2, 3, 1)
            a = a.to(torch.bfloat16).contiguous()
            b = b.to(torch.bfloat16).contiguous()
            n, hw = a.shape[0], a.shape[1] * a.shape[2]
            g = ops.lpips_diff(a, b, lin_ws[i], out, 1.0 / (hw * n), need, accumulate=i > 0)
            grads.append(g)
        ctx.k = k
        ctx.dtypes = [f.dtype for f in f1s]
        ctx.save_for_backward(*[g for g in grads if g is not None])
        ctx.need = need
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        k = ctx.k
        if not ctx.need:
            return (None,) * (1 + 2 * k)
        grads = [(g.permute(0, 3, 1, 2) * gout).to(dt) for g, dt in zip(ctx.saved_tensors, ctx.dtypes)]
        return (None,) + (None,) * k + tuple(grads)


class LPIPS(nn.Module):
    def __init__(self, ckpt_path=None, use_dropout=True):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        self.net = vgg16(pretrained=True, requires_grad=False)
        self.lin0 = NetLinLayer(self.chns[0], use_dropout=use_dropout)
        self.lin1 = NetLinLayer(self.chns[1], use_dropout=use_dropout)
        self.lin2 = NetLinLayer(self.chns[2], use_dropout=use_dropout)
        self.lin3 = NetLinLayer(self.chns[3], use_dropout=use_dropout)
        self.lin4 = NetLinLayer(self.chns[4], use_dropout=use_dropout)
        if ckpt_path is not None:
            self.load_from_pretrained(ckpt_path)
        for param in self.parameters():
            param.requires_grad = False

    def load_from_pretrained(self, ckpt_path=None, name="vgg_lpips"):
        self.load_state_dict(torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=True), strict=False)

    def forward(self, input, target):
        in0, in1 = self.scaling_layer(input), self.scaling_layer(target)
        in0 = in0.contiguous(memory_format=torch.channels_last)
        in1 = in1.contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            outs0 = self.net(in0)
        outs1 = self.net(in1)
        lins = [self.lin0, self.lin1, self.lin2, self.lin3, self.lin4]
        ws = [l.model[-1].weight.detach().reshape(-1).float().contiguous() for l in lins]
        return _LpipsDiff.apply(ws, *outs0, *outs1)
