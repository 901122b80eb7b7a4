"""Parameter initialisation with the same random-number consumption as the reference's models/init_param.py:4-33, so that
`torch.manual_seed(s)` followed by the reference's constructor order yields the reference's parameters bit for bit (pinned by the
`ck.*` checksums in tests/golden/{vae_forward_tiny,decoder_full_b1}.npz).

Rule per module, in `model.modules()` order (one generator draw per weight tensor, none for constants):
  Linear / Embedding         weight ~ trunc_normal(std = other_std) (padding row zeroed), bias = 0
  Conv* / ConvTranspose*     weight ~ trunc_normal(std = v) if v > 0 else xavier_normal(gain = -v), bias = 0      (v = conv_std_or_gain)
  *Norm                      weight = 1, bias = 0
|v| > 10 leaves the model untouched.
"""
import torch
import torch.nn as nn

_DENSE = (nn.Linear, nn.Embedding)
_CONV = (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d)
_NORM = (nn.LayerNorm, nn.GroupNorm, nn.SyncBatchNorm, nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d,
         nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d)


def _fill(t, value):
    if t is not None:
        nn.init.constant_(t.data, value)


@torch.no_grad()
def init_weights(model: nn.Module, conv_std_or_gain: float = 0.02, other_std: float = 0.02, verbose: bool = False):
    v = conv_std_or_gain
    if abs(v) > 10:
        return
    if verbose:
        print(f"[init_weights] {type(model).__name__}: conv {'std' if v > 0 else 'gain'} {abs(v):g}, dense std {other_std:g}")

    def draw_conv(w):
        return nn.init.trunc_normal_(w, std=v) if v > 0 else nn.init.xavier_normal_(w, gain=-v)

    for m in model.modules():
        if isinstance(m, _DENSE):
            nn.init.trunc_normal_(m.weight.data, std=other_std)
            if isinstance(m, nn.Embedding):
                if m.padding_idx is not None:
                    m.weight.data[m.padding_idx].zero_()
            else:
                _fill(m.bias, 0.0)
        elif isinstance(m, _CONV):
            draw_conv(m.weight.data)
            _fill(getattr(m, "bias", None), 0.0)
        elif isinstance(m, _NORM):
            _fill(m.bias, 0.0)
            _fill(m.weight, 1.0)
