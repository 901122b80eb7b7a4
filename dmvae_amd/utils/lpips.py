"""Drop-in counterpart of the reference's utils/lpips.py (LPIPS :51-94, ScalingLayer :97-104, NetLinLayer :107-113,
vgg16 :116-153): same class names, state_dict keys (net.slice{1..5}.{idx}.*, lin{0..4}.model.1.weight) and
``forward(input, target) -> scalar``.

The feature-difference reduction (normalize_tensor, diff^2, 1x1 lin, spatial mean, level sum, batch mean;
:86-94,156-162) runs in the fused HIP kernel dmvae_lpips_diff (forward value + gradient w.r.t. the second
argument's features in one pass over NHWC bf16 features).  The VGG16 trunk runs on the same
implicit-GEMM conv kernel as the decoder (ReLU epilogue, NHWC bf16, both LPIPS branches batched as one pass), with HIP
2x2 max pools; its backward is hand-scheduled (`_VggLpips`).  torchvision is not installed here, so the 'D' layer list is built
locally; pretrained trunk weights must be loaded from a checkpoint -- they are not downloadable offline."""
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


class _VggLpips(torch.autograd.Function):
    """LPIPS(input, target) on the HIP kernels: ScalingLayer -> VGG16 trunk (both branches batched as one N = 2B pass through
    the implicit-GEMM conv kernel with a ReLU epilogue, 2x2 max pools) -> fused feature-diff reduction per tapped level.
    Backward (w.r.t. `target` only -- the reference calls lpips(images, recon), the first argument carries no graph) walks
    the trunk in reverse on the target half: conv input-gradients with the ReLU gate fused into the producing conv's epilogue
    or into the max-pool backward.  Replaces utils/lpips.py:81-94,116-153 (MIOpen convs / ATen pools in the reference)."""

    @staticmethod
    def forward(ctx, inp, tgt, mod):
        from .. import functional as Fn
        b = inp.shape[0]
        shift, scale = mod.scaling_layer.shift, mod.scaling_layer.scale
        x = torch.cat([(inp.detach().float() - shift) / scale, (tgt.detach().float() - shift) / scale], 0).contiguous()
        h = ops.nchw_to_nhwc_bf16(x, c_pad=32)                       # [2B, H, W, 32] (3 real channels)
        need = tgt.requires_grad
        lin_ws = [l.model[-1].weight.detach().reshape(-1).float().contiguous() for l in (mod.lin0, mod.lin1, mod.lin2, mod.lin3, mod.lin4)]
        out = torch.zeros(1, dtype=torch.float32, device=inp.device)
        convs = [m for sl in (mod.net.slice1, mod.net.slice2, mod.net.slice3, mod.net.slice4, mod.net.slice5) for m in sl if isinstance(m, nn.Conv2d)]
        tape, ci, level = [], 0, 0                                    # tape: ("conv", conv, y_tgt_half, df1 | None) / ("pool",)
        taps = {1, 3, 6, 9, 12}                                        # conv indices whose ReLU output is an LPIPS feature
        for v in _CFG:
            if v == "M":
                h = ops.maxpool2x2(h)
                tape.append(("pool",))
                continue
            conv = convs[ci]
            wp = Fn.packed(conv.weight, False, 0, 32 if conv.weight.shape[1] < 32 else 0, frozen=True)
            h = ops.conv2d_nhwc(h, wp, conv.bias.detach().float(), ks=3, act=ops.ACT_RELU)
            df1 = None
            if ci in taps:
                n, hh, ww, _ = h.shape
                df1 = ops.lpips_diff(h[:b], h[b:], lin_ws[level], out, 1.0 / (hh * ww * b), need, accumulate=level > 0)
                level += 1
            tape.append(("conv", conv, h[b:] if need else None, df1))
            ci += 1
        ctx.tape, ctx.need, ctx.scale = tape, need, scale
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        if not ctx.need:
            return None, None, None
        from .. import functional as Fn
        tape = ctx.tape
        d = None            # gradient w.r.t. the input of the layer above (bf16 NHWC, target half), already ReLU-gated where due
        pending_pool = False
        for k in range(len(tape) - 1, -1, -1):
            e = tape[k]
            if e[0] == "pool":
                pending_pool = True
                continue
            _, conv, y, df1 = e
            if d is None:                                   # topmost conv (relu5_3): only the feature gradient arrives
                dpre = ops.relu_bwd(df1, y)
            elif pending_pool:                              # conv -> relu -> [tap] -> pool: un-pool + feature gradient + ReLU gate
                dpre = ops.maxpool2x2_relu_bwd(d, y, df1)
            else:
                dpre = d                                    # gate was fused into the dgrad conv that produced d
            pending_pool = False
            # input gradient of this conv; if the layer below is conv+ReLU directly, gate by its saved output in the epilogue
            below = tape[k - 1] if k > 0 else None
            cin = conv.weight.shape[1]
            wd = Fn.packed(conv.weight, True, 4 if cin < 4 else 0, 0, frozen=True)
            if below is None:
                dimg = ops.conv2d_nhwc(dpre, wd, ks=3, out_f32=True)                      # [B, H, W, 4]
                g = ops.nhwc_to_nchw_f32(dimg, cin) / ctx.scale * gout
                return None, g, None
            if below[0] == "conv":
                d = ops.conv2d_nhwc(dpre, wd, residual=below[2], ks=3, act=ops.ACT_RELU_GATE)
            else:
                d = ops.conv2d_nhwc(dpre, wd, ks=3)
        raise AssertionError("unreachable")


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
        if not (input.is_cuda and target.is_cuda):
            raise ops._lib.DmvaeHipError("LPIPS: expected GPU tensors; dmvae_amd has no CPU path")
        return _VggLpips.apply(input, target, self)
