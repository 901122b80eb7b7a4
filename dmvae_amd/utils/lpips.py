"""Drop-in counterpart of the reference's utils/lpips.py (LPIPS :51-94, ScalingLayer :97-104, NetLinLayer :107-113,
vgg16 :116-153): same class names, state_dict keys (net.slice{1..5}.{idx}.*, lin{0..4}.model.1.weight) and
``forward(input, target) -> scalar``.

The feature-difference reduction (normalize_tensor, diff^2, 1x1 lin, spatial mean, level sum, batch mean;
:86-94,156-162) runs in the fused HIP kernel dmvae_lpips_diff (forward value + gradient w.r.t. the second
argument's features in one pass over NHWC bf16 features).  The VGG16 trunk runs on the same
implicit-GEMM conv kernel as the decoder (ReLU epilogue, NHWC bf16, both LPIPS branches batched as one pass), with HIP
2x2 max pools; its backward is hand-scheduled (`_VggLpips`).  torchvision is not installed here, so the 'D' layer list is built
locally; pretrained trunk weights must be loaded from a checkpoint -- they are not downloadable offline."""
import warnings

import torch
import torch.nn as nn

from .. import ops, parity

FIRST_LAYER_FUSED = True      # _VggLpips.forward: csrc/conv_in3.hip for ScalingLayer + cat + conv1_1 + ReLU (tests compare it with the padded 32-channel route)

DIFF_POOL_FUSED = True        # a tapped level's feature diff and the max pool behind it in one pass (tests compare it with the two launches)

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
        # The reference takes the trunk from torchvision's `vgg16(pretrained=True)` (utils/lpips.py:119) -- its ckpt_vae/vgg.pth holds only the five
        # lin*.model.1.weight tensors.  torchvision is not installed in this image and there is no network: `trunk_loaded` stays False until
        # `load_trunk()` (or a state_dict that carries net.slice*) supplies the weights, and LPIPS warns / raises on use accordingly.
        self.trunk_loaded = False
        if pretrained:
            try:
                import torchvision                                                            # noqa: F401
                tv = torchvision.models.vgg16(weights=torchvision.models.VGG16_Weights.IMAGENET1K_V1)
                self.load_torchvision_features(tv.features.state_dict())
            except Exception as e:                                                            # ImportError offline, URLError without network, ...
                import os
                if os.environ.get("DMVAE_LPIPS_RANDOM_TRUNK", "0") in ("", "0"):
                  warnings.warn("LPIPS: torchvision's pretrained VGG16 trunk is not available here (%s: %s); the trunk is RANDOMLY initialised until "
                              "LPIPS.load_trunk(path) loads torchvision-layout (`features.N.weight/bias`) or net.slice* weights"
                              % (type(e).__name__, str(e)[:80]), stacklevel=3)
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def load_torchvision_features(self, sd: dict) -> None:
        """torchvision VGG16 `features` state_dict ({"N.weight", "N.bias"} or {"features.N.weight", ...}, N = the layer index 0..28 the reference's
        slices also use, utils/lpips.py:126-135) -> net.slice{k}.{N}.*; every one of the 13 convolutions must be present."""
        own = {}
        for k in range(1, 6):
            for idx, m in getattr(self, f"slice{k}").named_children():
                if isinstance(m, nn.Conv2d):
                    own[int(idx)] = m
        got = {}
        for name, t in sd.items():
            parts = name.split(".")
            if parts[0] == "features":
                parts = parts[1:]
            if len(parts) == 2 and parts[0].isdigit() and parts[1] in ("weight", "bias"):
                got[(int(parts[0]), parts[1])] = t
        missing = [f"features.{i}.{w}" for i in sorted(own) for w in ("weight", "bias") if (i, w) not in got]
        if missing:
            raise KeyError(f"VGG16 trunk weights incomplete: missing {missing[:4]}{'...' if len(missing) > 4 else ''}")
        with torch.no_grad():
            for i, m in own.items():
                m.weight.copy_(got[(i, "weight")])
                m.bias.copy_(got[(i, "bias")])
        self.trunk_loaded = True

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
        convs = [m for sl in (mod.net.slice1, mod.net.slice2, mod.net.slice3, mod.net.slice4, mod.net.slice5) for m in sl if isinstance(m, nn.Conv2d)]
        # first layer: ScalingLayer, the concatenation of the two branches and conv1_1 + ReLU in one kernel on the three real channels (csrc/conv_in3.hip) where the
        # shape allows; else the ScalingLayer in ATen and the image zero-padded to one 32-channel K step of the general kernel
        first = None
        if FIRST_LAYER_FUSED and not parity.on() and inp.shape[1] == 3 and ops.conv_in3_supported(2 * b, inp.shape[2], inp.shape[3], convs[0].weight.shape[0]) \
                and convs[0].weight.dtype == torch.float32:
            first = ops.conv_in3(inp.detach().float().contiguous(), tgt.detach().float().contiguous(), convs[0].weight.detach().contiguous(),
                                 convs[0].bias.detach().float(), shift.float(), scale.float(), act=ops.ACT_RELU)
            h = None
        else:
            x = torch.cat([(inp.detach().float() - shift) / scale, (tgt.detach().float() - shift) / scale], 0).contiguous()
            h = ops.nchw_to_nhwc_bf16(x, c_pad=32)                   # [2B, H, W, 32] (3 real channels)
        need = tgt.requires_grad
        lin_ws = [l.model[-1].weight.detach().reshape(-1).float().contiguous() for l in (mod.lin0, mod.lin1, mod.lin2, mod.lin3, mod.lin4)]
        out = torch.zeros(1, dtype=torch.float32, device=inp.device)
        tape, ci, level = [], 0, 0                                    # tape: ("conv", conv, y_tgt_half, df1 | None) / ("pool",)
        taps = {1, 3, 6, 9, 12}                                        # conv indices whose ReLU output is an LPIPS feature
        pooled = None
        for pos, v in enumerate(_CFG):
            if v == "M":
                h = pooled if pooled is not None else ops.maxpool2x2(h)      # the tapped level in front of it pooled on the way (ops.lpips_diff_pool)
                pooled = None
                tape.append(("pool",))
                continue
            conv = convs[ci]
            if ci == 0 and first is not None:
                h = first
            else:
                wp = Fn.packed(conv.weight, False, 0, 32 if conv.weight.shape[1] < 32 else 0, frozen=True)
                h = ops.conv2d_nhwc(h, wp, conv.bias.detach().float(), ks=3, act=ops.ACT_RELU)
            df1 = None
            if ci in taps:
                n, hh, ww, _ = h.shape
                if DIFF_POOL_FUSED and not parity.on() and pos + 1 < len(_CFG) and _CFG[pos + 1] == "M" and hh % 2 == 0 and ww % 2 == 0 and h.dtype == torch.bfloat16:
                    df1, pooled = ops.lpips_diff_pool(h, b, lin_ws[level], out, 1.0 / (hh * ww * b), need, accumulate=level > 0)
                else:
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
                if FIRST_LAYER_FUSED and not parity.on() and dpre.dtype == torch.bfloat16 and tuple(wd.shape[:2]) == (4, 9) \
                        and ops.conv_to_image_supported(dpre.shape[0], dpre.shape[1], dpre.shape[2], dpre.shape[3], cin):
                    # the image gradient straight out of the conv's epilogue: NCHW f32, x gout / scale per channel (one launch for four)
                    mul = (gout.float().reshape(1) / ctx.scale.reshape(-1).float()).contiguous()
                    return None, ops.conv_to_image(dpre, wd, cin, mul=mul), None
                dimg = ops.conv2d_nhwc(dpre, wd, ks=3, out_f32=True)                      # [B, H, W, 4]
                g = ops.nhwc_to_nchw_f32(dimg, cin) / ctx.scale * gout
                return None, g, None
            if below[0] == "conv":
                d = ops.conv2d_nhwc(dpre, wd, residual=below[2], ks=3, act=ops.ACT_RELU_GATE)
            else:
                d = ops.conv2d_nhwc(dpre, wd, ks=3)
        raise AssertionError("unreachable")


class LPIPS(nn.Module):
    def __init__(self, ckpt_path=None, use_dropout=True, trunk_path=None):
        """ckpt_path: the reference's vgg.pth (lin layers; may also carry net.slice*).  trunk_path (not in the reference's signature, optional): a
        torchvision VGG16 checkpoint / `features` state_dict for offline use, see `load_trunk`."""
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.chns = [64, 128, 256, 512, 512]
        with warnings.catch_warnings():
            if trunk_path is not None:
                warnings.simplefilter("ignore")                  # the trunk is about to be loaded explicitly
            self.net = vgg16(pretrained=True, requires_grad=False)
        self.lin0 = NetLinLayer(self.chns[0], use_dropout=use_dropout)
        self.lin1 = NetLinLayer(self.chns[1], use_dropout=use_dropout)
        self.lin2 = NetLinLayer(self.chns[2], use_dropout=use_dropout)
        self.lin3 = NetLinLayer(self.chns[3], use_dropout=use_dropout)
        self.lin4 = NetLinLayer(self.chns[4], use_dropout=use_dropout)
        if trunk_path is not None:
            self.load_trunk(trunk_path)
        if ckpt_path is not None:
            self.load_from_pretrained(ckpt_path)
        for param in self.parameters():
            param.requires_grad = False

    def load_trunk(self, path_or_state_dict) -> None:
        """VGG16 trunk weights from a torchvision checkpoint (`vgg16-*.pth`: keys features.N.*, classifier.* ignored), a bare `features` state_dict, or
        a dict in this module's own layout (net.slice{k}.{N}.* / slice{k}.{N}.*)."""
        sd = path_or_state_dict
        if not isinstance(sd, dict):
            sd = torch.load(sd, map_location="cpu", weights_only=True)
        own_layout = {k.split("net.", 1)[-1]: v for k, v in sd.items() if "slice" in k}
        if own_layout:
            res = self.net.load_state_dict(own_layout, strict=False)
            if res.missing_keys:
                raise KeyError(f"VGG16 trunk weights incomplete: missing {res.missing_keys[:4]}")
            self.net.trunk_loaded = True
        else:
            self.net.load_torchvision_features(sd)

    def load_from_pretrained(self, ckpt_path=None, name="vgg_lpips"):
        """utils/lpips.py:75-78 (`strict=False`).  The reference's file carries the lin layers only; what `strict=False` skipped is inspected instead of
        ignored: missing lin weights are an error, a missing trunk is reported once (it must come from torchvision / `load_trunk`)."""
        sd = torch.load(ckpt_path, map_location=torch.device("cpu"), weights_only=True)
        res = self.load_state_dict(sd, strict=False)
        lin_missing = [k for k in res.missing_keys if k.startswith("lin")]
        if lin_missing:
            raise KeyError(f"LPIPS checkpoint {ckpt_path} lacks the linear layers {lin_missing}")
        if not any(k.startswith("net.slice") for k in res.missing_keys):
            self.net.trunk_loaded = True
        elif not self.net.trunk_loaded:
            import os
            if os.environ.get("DMVAE_LPIPS_RANDOM_TRUNK", "0") in ("", "0"):
              warnings.warn(f"LPIPS: {ckpt_path} holds no VGG16 trunk ({sum(k.startswith('net.slice') for k in res.missing_keys)} net.slice* tensors missing) and no "
                          "pretrained trunk was loaded: the perceptual loss would run on RANDOM features. Call LPIPS.load_trunk(<torchvision vgg16 .pth>) "
                          "or pass trunk_path=...; set DMVAE_LPIPS_RANDOM_TRUNK=1 to accept a random trunk (benchmarks, synthetic tests)", stacklevel=2)

    def forward(self, input, target):
        if not (input.is_cuda and target.is_cuda):
            raise ops._lib.DmvaeHipError("LPIPS: expected GPU tensors; dmvae_amd has no CPU path")
        return _VggLpips.apply(input, target, self)

    @property
    def trunk_loaded(self) -> bool:
        """True once the VGG16 trunk holds loaded (not randomly initialised) weights."""
        return bool(self.net.trunk_loaded)
