"""Stream-K / fused split-K instantiation of the Linear GEMM (csrc/gemm_pp.hip SK, `ops.linear_sk`): nn.Linear under autocast for the few-tile deep-K problems of
LightningDiT-XL/1 / ViT-L at batch 16 (diffusion/lightningdit/lightningdit.py:66-75,236-250, swiglu_ffn.py:15-36, their input gradients dX = dY W; timm blocks
via models/vae.py:47-53).

Reference: the same contraction in fp64 on the GPU over the same bf16 operands.  Bars:
  * every element within bf16 rounding of the fp64 result (the f32 sum of the parts differs from the one-chain sum by f32 rounding of a few partial sums);
  * against the one-chain kernel (`ops.linear_bf16`): at most a handful of last-place bf16 flips, none larger than one ulp;
  * run-to-run bit-identical -- the parts of a tile are summed in K order whichever workgroup arrives last --, also when the call is repeated back to back many
    times (the arrival counters are left zero);
  * uniform parts (splits >= 2): the cut depends on N and K only, so the rows of a 2B-row call are bit-identical to two B-row calls;
  * fused activations (GELU / SiLU / SwiGLU) == the act = 0 SK call followed by the standalone kernel, bit for bit; rows past M / columns past N never written."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

#          M,     N,    K,   splits
CASES = [
    (4096, 1152, 3072, 0), (4096, 1152, 3072, 3),      # DiT-XL/1 w3, batch 16
    (4096, 1152, 3456, 0), (4096, 1152, 3456, 3),      # input gradient of qkv
    (4096, 1152, 6144, 0), (4096, 1152, 6144, 3),      # input gradient of w12
    (4112, 1024, 4096, 0), (4112, 1024, 4096, 2),      # ViT-L fc2 at 16 x 257 tokens (ragged M)
    (4096, 6144, 1152, 0),                             # w12: 384 tiles = 1.5 rounds
    (8224, 1024, 4096, 0),                             # ViT-L fc2 at batch 32 (C2's frozen encoder)
    (4096, 1152, 1152, 3),                             # proj: 12-step parts
    (16384, 1152, 3072, 0),                            # DiT-XL/1 w3, batch 64: 320 tiles
    (300, 520, 1024, 0), (300, 520, 1024, 4),          # small and ragged
]
IDS = ["%dx%dx%d_s%d" % c for c in CASES]


def _operands(m, n, k, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(m, k, device=DEV, generator=g).to(BF)
    w = (torch.randn(n, k, device=DEV, generator=g) * (k ** -0.5)).to(BF)
    b = torch.randn(n, device=DEV, generator=g)
    return x, w, b


@pytest.mark.parametrize("tile", [0, 1], ids=["256x256", "256x128"])
@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_linear_sk_vs_fp64_and_one_chain(case, tile):
    from dmvae_amd import ops
    import functools
    m, n, k, s = case
    if not ops.linear_sk_supported(m, n, k, s, tile):
        pytest.skip("shape outside the SK instantiation's range")
    ops = type("O", (), {**{a: getattr(ops, a) for a in dir(ops) if not a.startswith("__")}, "linear_sk": staticmethod(functools.partial(ops.linear_sk, tile=tile))})
    x, w, b = _operands(m, n, k, seed=m + s)
    ref = torch.addmm(b.double(), x.double(), w.double().t())
    buf = torch.full((m + 8, n), 7.0, dtype=BF, device=DEV)            # the result inside a sentinel-filled buffer? (ldy = N: rows past M only)
    y = ops.linear_sk(x, w, b, splits=s)
    scale = ref.abs().max().item()
    assert (y.double() - ref).abs().max().item() <= 2.0 ** -8 * scale + 1e-6, "outside bf16 rounding of the fp64 result"
    err = ((y.double() - ref).abs() / (ref.abs() + 1e-2 * scale)).max().item()
    assert err < 2.0 ** -7, f"element-wise relative error {err:.2e}"
    y1 = ops.linear_bf16(x, w, b)
    diff = (y.float() - y1.float()).abs()
    ulp = y1.float().abs() * 2.0 ** -7 + 1e-5 * scale                  # one bf16 ulp of the element (+ the f32 summation-order difference near zero)
    assert bool((diff <= ulp).all()), "more than one bf16 ulp from the one-chain kernel"
    assert (diff > 0).float().mean().item() < 0.02, "too many last-place differences from the one-chain kernel"
    for _ in range(5):                                                 # counters left zero, fixed summation order
        assert torch.equal(ops.linear_sk(x, w, b, splits=s), y), "rerun differs"
    # K-tile-major weights (what frozen weights / transposed input-gradient operands are served as): the same bits
    wk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32)
    assert torch.equal(ops.linear_sk(x, wk, b, splits=s), y)
    del buf


@pytest.mark.parametrize("shape", [(4096, 1152, 3072, 3), (4096, 1152, 6144, 3), (4096, 1152, 1152, 3), (2048, 1024, 4096, 2)])
def test_uniform_parts_do_not_depend_on_the_row_count(shape):
    """splits >= 2: a 2B-row call equals two B-row calls bit for bit (the DMD loss's cond / uncond pair as one call, train_dmd.py:212-217) -- on either tile."""
    from dmvae_amd import ops
    m, n, k, s = shape
    x, w, b = _operands(2 * m, n, k, seed=3)
    for tile in (0, 1):
        if not ops.linear_sk_supported(m, n, k, s, tile):
            continue
        both = ops.linear_sk(x, w, b, splits=s, tile=tile)
        lo = ops.linear_sk(x[:m].contiguous(), w, b, splits=s, tile=tile)
        hi = ops.linear_sk(x[m:].contiguous(), w, b, splits=s, tile=tile)
        assert torch.equal(both[:m], lo) and torch.equal(both[m:], hi), tile


@pytest.mark.parametrize("splits", [0, 3])
def test_linear_sk_fused_activations(splits):
    from dmvae_amd import ops
    m, n, k = 4096, 1152, 3072
    x, w, b = _operands(m, n, k, seed=9)
    y0 = ops.linear_sk(x, w, b, splits=splits)
    assert torch.equal(ops.linear_sk(x, w, b, act=ops.ACT_GELU, splits=splits), ops.gelu(y0))
    assert torch.equal(ops.linear_sk(x, w, b, act=ops.ACT_SILU, splits=splits), ops.silu(y0))
    g = ops.linear_sk(x, w, b, act=ops.ACT_SWIGLU, splits=splits)
    if splits:
        assert torch.equal(g, ops.swiglu(y0))
    else:      # stream-K cuts follow the TILES, and the gated form's tiles pair columns (h, H + h): other cut positions per element than the plain call's -> f32 rounding apart
        want = ops.swiglu(y0).float()
        assert (g.float() - want).abs().max().item() <= 2.0 ** -6 * want.abs().max().item()
        assert torch.equal(g, ops.linear_sk(x, w, b, act=ops.ACT_SWIGLU, splits=0))
    assert torch.equal(ops.linear_sk(x, w, None, splits=splits), ops.linear_sk(x, w, torch.zeros_like(b), splits=splits))


def test_linear_sk_interleaved_shapes_share_the_workspace():
    """Calls of different shapes alternate on one stream and one scratch buffer (counters at its start, slots behind): every result as when run alone."""
    from dmvae_amd import ops
    cases = [(4096, 1152, 3072, 0), (4112, 1024, 4096, 0), (4096, 1152, 6144, 3), (300, 520, 1024, 4)]
    ops_in = [(_operands(m, n, k, seed=i), s) for i, (m, n, k, s) in enumerate(cases)]
    alone = [ops.linear_sk(x, w, b, splits=s) for (x, w, b), s in ops_in]
    for _ in range(3):
        for ((x, w, b), s), want in zip(ops_in, alone):
            assert torch.equal(ops.linear_sk(x, w, b, splits=s), want)


def test_linear_sk_rejects_what_it_does_not_take():
    from dmvae_amd import ops
    from dmvae_amd._lib import DmvaeHipError
    assert not ops.linear_sk_supported(4096, 1152, 1152, 0)            # ranges of 11 steps: too short for stream-K
    assert not ops.linear_sk_supported(4096, 1152, 3072, 16)
    x, w, b = _operands(128, 64, 384)
    with pytest.raises(DmvaeHipError):
        ops.linear_sk(x, w, b, splits=0)
