"""Bit-identical reruns of the kernels whose LDS fragment reads are inline asm (csrc/conv_wgrad_pp.hip, wgrad_thin.hip: the compiler inserts no waits between the
LDS-DMA pieces and those reads -- the counted vmcnt + barrier protocol alone orders them, DESIGN_HISTORY.md 8.12) and of the other kernels added with them.  A race would show
as a run-to-run difference; tools/probes/stress_wgrad.py is the long form."""
import pytest
import torch

pytestmark = pytest.mark.gpu
REPS = 12


def _same(fn):
    ref = [t.clone() for t in fn() if t is not None]
    junk = torch.randn(2048, 2048, device="cuda")
    for r in range(REPS):
        if r % 3 == 0:
            junk.mul_(1.0001)
        out = [t for t in fn() if t is not None]
        assert all(torch.equal(a, b) for a, b in zip(ref, out)), f"rerun {r} differs"


@pytest.mark.parametrize("case", [(8, 64, 64, 512, 512, 3, 0, 1), (4, 128, 128, 256, 128, 3, 0, 1), (6, 32, 32, 128, 128, 3, 0, 1), (4, 32, 32, 512, 512, 3, 1, 1),
                                  (4, 64, 64, 256, 256, 4, 0, 2), (8, 32, 32, 512, 512, 1, 0, 1)],
                         ids=["plain256", "halo_k64", "halo_k32", "ups", "s2", "1x1"])
def test_wgrad_pp_reruns_are_bit_identical(case):
    from dmvae_amd import ops
    n, h, w, cin, cout, ks, ups, stride = case
    g = torch.Generator().manual_seed(5)
    a = torch.randn(n, h, w, cin, generator=g).to("cuda").to(torch.bfloat16)
    ho, wo = (2 * h, 2 * w) if ups else ((h // 2, w // 2) if stride == 2 else (h, w))
    dy = torch.randn(n, ho, wo, cout, generator=g).to("cuda").to(torch.bfloat16)
    _same(lambda: ops.conv2d_nhwc_wgrad(dy, a, ks, upsample=bool(ups), stride=stride))


def test_thin_kernels_and_attention_reruns_are_bit_identical():
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(6)
    a = torch.randn(2, 64, 64, 128, generator=g).to("cuda").to(torch.bfloat16)
    dyi = torch.randn(2, 3, 64, 64, generator=g).to("cuda")
    _same(lambda: (ops.conv_out_wgrad(dyi, a),))
    qkv = torch.randn(8, 257, 3 * 16 * 64, generator=g).to("cuda").to(torch.bfloat16)
    _same(lambda: (ops.attention_qkv(qkv, 16, 0.125),))
