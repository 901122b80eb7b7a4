"""Drop-in counterparts of the reference's models/flux_ae.py modules, computing on MI355X HIP kernels.

Same class names, constructor arguments, ``forward`` signatures and ``state_dict`` keys as the
reference (AttnBlock :25-52, ResnetBlock :55-82, Downsample :85-95, Upsample :98-107, Encoder
:110-181, Decoder :184-278).  Parameters live in ordinary nn.Conv2d / nn.GroupNorm holders created in
the reference's order (so a fixed seed gives the reference's initial weights and checkpoints load
with strict=True), but ``forward`` never calls them: it dispatches to dmvae_amd.functional.

Public ``forward`` takes/returns NCHW like the reference; ``forward_nhwc`` is the internal
channels-last bf16 path the Decoder chains without layout round-trips.
"""
from dataclasses import dataclass

import torch
from torch import Tensor, nn

from .. import functional as Fn
from .. import ops


@dataclass
class AutoEncoderParams:
    resolution: int
    in_channels: int
    ch: int
    out_ch: int
    ch_mult: list
    num_res_blocks: int
    z_channels: int
    scale_factor: float
    shift_factor: float


def swish(x: Tensor) -> Tensor:
    return x * torch.sigmoid(x)


def _gn(c):
    return nn.GroupNorm(num_groups=32, num_channels=c, eps=1e-6, affine=True)


def _mid_stage(c: int) -> nn.Module:
    """ResnetBlock -> AttnBlock -> ResnetBlock at constant width (registered as block_1 / attn_1 / block_2, the checkpoint's names)."""
    stage = nn.Module()
    stage.block_1 = ResnetBlock(c, c)
    stage.attn_1 = AttnBlock(c)
    stage.block_2 = ResnetBlock(c, c)
    return stage


def _level(c_in: int, c_out: int, n_blocks: int, resample=None, resample_name: str = "") -> nn.Module:
    """One resolution level: `n_blocks` ResnetBlocks (the first one changes the width), an `attn` list the reference registers but never fills, and
    optionally the level's resampling module, registered last under `resample_name`."""
    lvl = nn.Module()
    lvl.block = nn.ModuleList(ResnetBlock(c_in if i == 0 else c_out, c_out) for i in range(n_blocks))
    lvl.attn = nn.ModuleList()
    if resample is not None:
        setattr(lvl, resample_name, resample(c_out))
    return lvl


class _NCHWAdapter(nn.Module):
    def forward(self, x: Tensor) -> Tensor:
        return Fn.to_nchw(self.forward_nhwc(Fn.to_nhwc_bf16(x)), x.dtype if x.dtype != torch.float64 else torch.float32)


class AttnBlock(_NCHWAdapter):
    def __init__(self, in_channels: int):
        super().__init__()
        self.in_channels = in_channels
        self.norm = _gn(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.k = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.v = nn.Conv2d(in_channels, in_channels, kernel_size=1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, kernel_size=1)

    def forward_nhwc(self, x: Tensor) -> Tensor:
        return Fn.AttnBlockFn.apply(x, self.norm.weight, self.norm.bias, self.q.weight, self.q.bias, self.k.weight, self.k.bias,
                                    self.v.weight, self.v.bias, self.proj_out.weight, self.proj_out.bias)


class ResnetBlock(_NCHWAdapter):
    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels = in_channels
        out_channels = in_channels if out_channels is None else out_channels
        self.out_channels = out_channels
        self.norm1 = _gn(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.norm2 = _gn(out_channels)
        self.conv2 = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        if self.in_channels != self.out_channels:
            self.nin_shortcut = nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)

    def forward_nhwc(self, x: Tensor) -> Tensor:
        sc = getattr(self, "nin_shortcut", None)
        return Fn.ResnetBlockFn.apply(x, self.norm1.weight, self.norm1.bias, self.conv1.weight, self.conv1.bias, self.norm2.weight,
                                      self.norm2.bias, self.conv2.weight, self.conv2.bias, None if sc is None else sc.weight,
                                      None if sc is None else sc.bias)


class Downsample(_NCHWAdapter):
    """pad (0,1,0,1) + conv3x3 stride 2 (reference :85-95): one strided gather in the HIP conv kernel."""

    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=2, padding=0)

    def forward_nhwc(self, x: Tensor) -> Tensor:
        return Fn.DownsampleFn.apply(x, self.conv.weight, self.conv.bias)


class Upsample(_NCHWAdapter):
    def __init__(self, in_channels: int):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=1, padding=1)

    def forward_nhwc(self, x: Tensor) -> Tensor:
        return Fn.ConvFn.apply(x, self.conv.weight, self.conv.bias, 3, True)


class _Conv3x3(nn.Conv2d):
    """nn.Conv2d parameter holder whose forward runs the HIP kernel on NHWC bf16 (used inside Sequential conv_in)."""

    def forward_nhwc(self, x: Tensor) -> Tensor:
        return Fn.ConvFn.apply(x, self.weight, self.bias, self.kernel_size[0], False)

    def forward(self, x: Tensor) -> Tensor:
        return Fn.to_nchw(self.forward_nhwc(Fn.to_nhwc_bf16(x)), x.dtype)


class Encoder(nn.Module):
    def __init__(self, resolution: int, in_channels: int, ch: int, ch_mult: list, num_res_blocks: int, z_channels: int):
        super().__init__()
        self.ch, self.resolution, self.in_channels = ch, resolution, in_channels
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        self.in_ch_mult = (1,) + tuple(ch_mult)
        widths = [ch * m for m in self.in_ch_mult]          # widths[i] -> widths[i + 1] across level i
        last = self.num_resolutions - 1
        self.conv_in = _Conv3x3(in_channels, ch, kernel_size=3, stride=1, padding=1)
        self.down = nn.ModuleList(
            _level(widths[i], widths[i + 1], num_res_blocks, Downsample if i != last else None, "downsample") for i in range(self.num_resolutions))
        self.mid = _mid_stage(widths[-1])
        self.norm_out = _gn(widths[-1])
        self.conv_out = _Conv3x3(widths[-1], 2 * z_channels, kernel_size=3, stride=1, padding=1)

    def forward(self, x: Tensor) -> Tensor:
        """Reference :160-181 (downsampling stack -> mid -> norm_out/swish/conv_out), NCHW in, NCHW [B, 2*z, H/2^(L-1), W/2^(L-1)] out."""
        if self.in_channels % 32:
            h = Fn.ConvInFn.apply(x, self.conv_in.weight, self.conv_in.bias)
        else:
            h = self.conv_in.forward_nhwc(Fn.to_nhwc_bf16(x))
        for i_level in range(self.num_resolutions):
            for i_block in range(self.num_res_blocks):
                h = self.down[i_level].block[i_block].forward_nhwc(h)
                if len(self.down[i_level].attn) > 0:      # never populated by the reference ctor (:139 builds an empty list)
                    h = self.down[i_level].attn[i_block].forward_nhwc(h)
            if i_level != self.num_resolutions - 1:
                h = self.down[i_level].downsample.forward_nhwc(h)
        h = self.mid.block_1.forward_nhwc(h)
        h = self.mid.attn_1.forward_nhwc(h)
        h = self.mid.block_2.forward_nhwc(h)
        h = Fn.NormSwishConvFn.apply(h, self.norm_out.weight, self.norm_out.bias, self.conv_out.weight, self.conv_out.bias)
        return Fn.to_nchw(h, x.dtype if x.dtype != torch.float64 else torch.float32)


class Decoder(nn.Module):
    def __init__(self, ch: int, out_ch: int, ch_mult: list, num_res_blocks: int, in_channels: int, resolution: int, z_channels: int):
        super().__init__()
        self.ch, self.resolution, self.in_channels = ch, resolution, in_channels
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        top = self.num_resolutions - 1
        self.ffactor = 1 << top
        widths = [ch * m for m in ch_mult]                  # level i runs at widths[i]; it is entered from level i + 1 (the top level from `block_in`)
        self.block_in = widths[top]
        low_res = resolution // self.ffactor
        self.z_shape = (1, z_channels, low_res, low_res)
        self.conv_in = _Conv3x3(z_channels, self.block_in, kernel_size=3, stride=1, padding=1)
        self.mid = _mid_stage(self.block_in)
        # `up[i]` is level i of the checkpoint (the reference builds top-down and inserts at the front; the registration order is the same)
        self.up = nn.ModuleList(
            _level(widths[min(i + 1, top)], widths[i], num_res_blocks + 1, Upsample if i != 0 else None, "upsample") for i in range(self.num_resolutions))
        self.norm_out = _gn(widths[0])
        self.conv_out = nn.Conv2d(widths[0], out_ch, kernel_size=3, stride=1, padding=1)

    def forward(self, z: Tensor, grad_ckpt=False) -> Tensor:
        """z: [B, 256, C] tokens (the reference hard-codes the 16x16 grid, :244-245) or NCHW [B, C, h, w].
        Returns the NCHW image; f32 (the reference returns bf16 under autocast and VAE.forward casts to float)."""
        h = self._body_nhwc(z)
        return Fn.NormConvOutFn.apply(h, self.norm_out.weight, self.norm_out.bias, self.conv_out.weight, self.conv_out.bias)

    @torch.no_grad()
    def forward_uint8(self, z: Tensor, round_bf16: bool = True) -> Tensor:
        """Decode straight to the [B, H, W, 3] uint8 image sample_50k.py:149-151 builds (`clamp(127.5 * decode(z).float() + 128, 0, 255)`, channels
        last, uint8): the output conv's f32 NHWC result goes through one conversion kernel instead of NHWC->NCHW f32, clamp, permute and a cast.
        round_bf16: round the decoded value to bf16 first -- what `.float()` of the reference's autocast decoder output holds."""
        h = self._body_nhwc(z)
        cout = self.conv_out.weight.shape[0]
        _, a = Fn._gn_swish(h, self.norm_out.weight, self.norm_out.bias)
        cbp = torch.zeros(4, dtype=torch.float32, device=h.device)
        cbp[:cout] = self.conv_out.bias
        y4 = ops.conv2d_nhwc(a, Fn.packed(self.conv_out.weight, False, rows_pad=4), cbp, ks=3, out_f32=True, flop_channels=(self.conv_out.weight.shape[1], self.conv_out.weight.shape[0]))
        return ops.image_to_u8(y4, cout, round_bf16)

    def _body_nhwc(self, z: Tensor) -> Tensor:
        if z.ndim == 3:
            b, t, c = z.shape
            if t != 256:
                raise ValueError("Decoder expects 256 tokens (16x16) like the reference (flux_ae.py:245)")
            h = z.reshape(b, 16, 16, c).to(Fn.parity.act_dtype()).contiguous()      # tokens are already channels-last (bf16; f32 in the parity mode)
        else:
            h = Fn.to_nhwc_bf16(z)
        if isinstance(self.conv_in, nn.Sequential):
            for m in self.conv_in:
                h = m.forward_nhwc(h)
        else:
            h = self.conv_in.forward_nhwc(h)
        h = self.mid.block_1.forward_nhwc(h)
        h = self.mid.attn_1.forward_nhwc(h)
        h = self.mid.block_2.forward_nhwc(h)
        for i_level in reversed(range(self.num_resolutions)):
            for i_block in range(self.num_res_blocks + 1):
                h = self.up[i_level].block[i_block].forward_nhwc(h)
            if i_level != 0:
                h = self.up[i_level].upsample.forward_nhwc(h)
        return h

    def post_init(self, z_channels):
        self.conv_in = nn.Sequential(
            Upsample(z_channels),
            _Conv3x3(z_channels, self.block_in, kernel_size=3, stride=1, padding=1),
        )

    def get_last_layer(self):
        return self.conv_out.weight
