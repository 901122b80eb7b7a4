"""Weight initialisation matching the reference's models/init_param.py:4-33 (needed so a fixed seed
reproduces the reference's parameters: trunc-normal(std) convs/linears, zero biases, unit norms)."""
import torch.nn as nn

_CONVS = (nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.ConvTranspose1d, nn.ConvTranspose2d, nn.ConvTranspose3d)
_NORMS = (nn.LayerNorm, nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm, nn.GroupNorm,
          nn.InstanceNorm1d, nn.InstanceNorm2d, nn.InstanceNorm3d)


def init_weights(model: nn.Module, conv_std_or_gain: float = 0.02, other_std: float = 0.02, verbose: bool = False):
    """conv_std_or_gain > 0: trunc_normal_(std); < 0: xavier_normal_(gain=-v); |v| > 10: skip entirely."""
    if abs(conv_std_or_gain) > 10:
        return
    if verbose:
        print(f'[init_weights] {type(model).__name__} with {"std" if conv_std_or_gain > 0 else "gain"}={abs(conv_std_or_gain):g}')
    for m in model.modules():
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight.data, std=other_std)
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif isinstance(m, nn.Embedding):
            nn.init.trunc_normal_(m.weight.data, std=other_std)
            if m.padding_idx is not None:
                m.weight.data[m.padding_idx].zero_()
        elif isinstance(m, _CONVS):
            if conv_std_or_gain > 0:
                nn.init.trunc_normal_(m.weight.data, std=conv_std_or_gain)
            else:
                nn.init.xavier_normal_(m.weight.data, gain=-conv_std_or_gain)
            if getattr(m, "bias", None) is not None:
                nn.init.constant_(m.bias.data, 0.0)
        elif isinstance(m, _NORMS):
            if m.bias is not None:
                nn.init.constant_(m.bias.data, 0.0)
            if m.weight is not None:
                nn.init.constant_(m.weight.data, 1.0)
