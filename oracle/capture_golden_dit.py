"""Generate the LightningDiT fixtures (tests/golden/dit_*.npz) by importing the reference's own diffusion/lightningdit package (read-only at
/root/reference) in THIS container.  Separate script so that the generator streams of the older fixtures stay untouched.

Run:  TORCHDYNAMO_DISABLE=1 python oracle/capture_golden_dit.py            (CPU, under a minute; the @torch.compile decorators are no-ops then)

timm (PatchEmbed, Mlp) is satisfied by capture_golden.install_stubs' stand-ins (a Conv2d patch embedding and a two-layer MLP: no arithmetic of
the reference is replaced -- timm is third-party).  Parameters are oracle.detweights.det_tensor values regenerated from names + seed in the tests
(the reference zero-initialises the adaLN and output layers, which would make every fixture trivially zero)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import capture_golden as cg  # noqa: E402
from oracle.detweights import det_fill_  # noqa: E402


def main():
    cg.install_stubs()
    torch.set_grad_enabled(True)
    from diffusion.lightningdit.lightningdit import LightningDiT, LightningDiT_models
    g = torch.Generator().manual_seed(777)
    for tag, kw, seed in (("dit_small_hd64", dict(input_size=8, patch_size=1, in_channels=8, hidden_size=128, depth=2, num_heads=2, num_classes=10), 71),
                          ("dit_small_hd72", dict(input_size=8, patch_size=1, in_channels=8, hidden_size=144, depth=2, num_heads=2, num_classes=10), 72),
                          ("dit_small_p2", dict(input_size=8, patch_size=2, in_channels=4, hidden_size=128, depth=1, num_heads=2, num_classes=10), 73),
                          # head dim 64 again, at a width whose SwiGLU inner size int(2/3 * 4 * 192) = 512 is a multiple of 8: hidden 128 gives 341, which
                          # the HIP DiT kernels (16-byte rows) do not take, so `dit_small_hd64` can only pin the CPU oracle / stock module
                          ("dit_small_hd64w", dict(input_size=8, patch_size=1, in_channels=8, hidden_size=192, depth=2, num_heads=3, num_classes=10), 74)):
        m = LightningDiT(**kw).eval()
        fixed = {k: v.clone() for k, v in m.state_dict().items() if k == "pos_embed" or k.startswith("feat_rope")}
        det_fill_(m, seed, skip=("pos_embed",))
        b = 3
        x = torch.randn(b, kw["in_channels"], kw["input_size"], kw["input_size"], generator=g).requires_grad_(True)
        t = torch.rand(b, generator=g)
        y = torch.tensor([1, 10, 4])                      # 10 = the unconditional row of the embedding table
        out = m(x, t, y)
        dy = torch.randn(out.shape, generator=g)
        out.backward(dy)
        grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
        cg.save(tag, seed=np.array(seed), x=x.detach(), t=t, y=y, out=out.detach(), dy=dy, dx=x.grad,
                keys=np.array(list(m.state_dict().keys())), **{"fix." + k: v for k, v in fixed.items()},
                **{"gn." + n: np.array([v.double().norm().item(), v.double().sum().item()]) for n, v in grads.items()},
                **{"g." + n: grads[n] for n in ("blocks.0.adaLN_modulation.1.weight", "blocks.0.attn.q_norm.weight", "blocks.0.mlp.w12.bias",
                                                 "final_layer.linear.weight", "x_embedder.proj.weight", "t_embedder.mlp.0.weight")})
    # ---- config C1's velocity model: LightningDiT-Mini/1 at ONE token per sample (toy_example_2d/dmd.py:436-454: input_size 1, in_channels = z_channels 2,
    # num_classes 1), its own generator so that the fixtures above keep their streams.  SwiGLU width int(2/3 * 1024) = 682.
    g2 = torch.Generator().manual_seed(778)
    m = LightningDiT_models["LightningDiT-Mini/1"](input_size=1, in_channels=2, num_classes=1).eval()
    fixed = {k: v.clone() for k, v in m.state_dict().items() if k == "pos_embed" or k.startswith("feat_rope")}
    det_fill_(m, 76, skip=("pos_embed",))
    b = 48
    x = (torch.rand(b, 2, 1, 1, generator=g2) * 3.0 - 1.5).requires_grad_(True)
    t = torch.rand(b, generator=g2)
    y = (torch.rand(b, generator=g2) < 0.25).long()        # 0 = the toy's single class, 1 = the unconditional row
    out = m(x, t, y)
    dy = torch.randn(out.shape, generator=g2)
    out.backward(dy)
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    cg.save("dit_toy_mini1", seed=np.array(76), x=x.detach(), t=t, y=y, out=out.detach(), dy=dy, dx=x.grad,
            keys=np.array(list(m.state_dict().keys())), **{"fix." + k: v for k, v in fixed.items()},
            **{"gn." + n: np.array([v.double().norm().item(), v.double().sum().item()]) for n, v in grads.items()},
            **{"g." + n: grads[n] for n in ("blocks.0.adaLN_modulation.1.bias", "blocks.5.norm2.weight", "blocks.2.mlp.w3.bias", "final_layer.linear.weight",
                                             "x_embedder.proj.weight", "t_embedder.mlp.2.bias", "blocks.3.attn.q_norm.weight", "blocks.1.attn.proj.bias")})
    xl = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000)
    sd = xl.state_dict()
    cg.save("dit_xl1_manifest", keys=np.array(list(sd.keys())), shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
            n_params=np.array(sum(p.numel() for p in xl.parameters())),
            **{"ck." + k: np.array([sd[k].double().sum().item(), sd[k].double().abs().sum().item()]) for k in ("pos_embed", "feat_rope.freqs_cos", "feat_rope.freqs_sin")},
            pos_embed_slice=sd["pos_embed"][0, ::37, ::97], rope_cos_slice=sd["feat_rope.freqs_cos"][::29, ::7])


if __name__ == "__main__":
    main()
