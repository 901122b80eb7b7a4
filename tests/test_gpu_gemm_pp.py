"""Parity of the hand-written Linear GEMM (csrc/gemm_pp.hip, `ops.linear_bf16`) at the shapes of the transformer blocks it serves: ViT-L at batch 32
(M = 32 x 257 = 8224 tokens: qkv / proj / fc1 + GELU / fc2; models/vae.py:47-53 -> timm blocks), the patch embedding as a GEMM (M = 8192, K = 768), and
LightningDiT-XL/1 (width 1152: qkv / proj / w12 / w3 at M = 4096 and 16384; diffusion/lightningdit/lightningdit.py:34-93, swiglu_ffn.py:15-36) -- every one of
them a tile-quantisation edge: M = 8224 leaves a 32-row remainder whatever the tile, N = 1152 is 4.5 tiles of 256 columns.

Reference: the same contraction in fp64 ON THE GPU over the same bf16-rounded operands (rocBLAS dgemm: independent of this build's kernels).  Bars, per assert:
  * f32 result: max |y - ref| / max |ref| < 1e-5 over EVERY element (f32 accumulation order is all that differs: products of bf16 operands are exact in f32),
    and on a random 1 % sample, element by element, |y_i - ref_i| <= 1e-4 |ref_i| + 2e-5 rms(ref);
  * the bf16 result is bit-for-bit the round-to-nearest-even of the f32 result (same kernel, same accumulation chain);
  * every tile of the menu gives the bit-identical result (one K-ordered accumulation chain per element, whatever the tile);
  * fused GELU / SiLU epilogue == the act = 0 call followed by the standalone kernel, bit for bit;
  * rows past M / columns past N are never written (the output sits inside a sentinel-filled buffer);
  * two runs are bit-identical."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

#          M,     N,    K      what
SHAPES = [
    (8224, 3072, 1024),    # ViT-L qkv
    (8224, 1024, 1024),    # ViT-L proj
    (8224, 4096, 1024),    # ViT-L fc1
    (8224, 1024, 4096),    # ViT-L fc2
    (8192, 1024, 768),     # patch embedding over 16 x 16 x 3 patches
    (4096, 1152, 1152),    # DiT-XL/1 proj, batch 16
    (4096, 3456, 1152),    # DiT-XL/1 qkv, batch 16
    (16384, 1152, 1152),   # DiT-XL/1 proj, batch 64
    (16384, 6144, 1152),   # DiT-XL/1 w12, batch 64
    (16384, 1152, 3072),   # DiT-XL/1 w3, batch 64
    (8224, 32, 2048),      # bottleneck MLP's second Linear (N = 32 output columns)
    (200, 72, 416),        # small and ragged in every dimension
]
IDS = ["%dx%dx%d" % s for s in SHAPES]


def _operands(m, n, k, seed=0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    x = torch.randn(m, k, device=DEV, generator=g).to(BF)
    w = (torch.randn(n, k, device=DEV, generator=g) * (k ** -0.5)).to(BF)
    b = torch.randn(n, device=DEV, generator=g)
    return x, w, b


def _ref64(x, w, b):
    return torch.addmm(b.double(), x.double(), w.double().t())


def _check(y, ref, what, seed=0):
    scale = ref.abs().max().item()
    err = (y.double() - ref).abs().max().item() / scale
    assert err < 1e-5, f"{what}: max|d|/max|ref| = {err:.2e}"
    fy, fr = y.reshape(-1), ref.reshape(-1)
    g = torch.Generator(device=DEV).manual_seed(seed)
    idx = torch.randint(0, fr.numel(), (max(1000, fr.numel() // 100),), device=DEV, generator=g)
    rms = ref.pow(2).mean().sqrt().item()
    d = (fy[idx].double() - fr[idx]).abs()
    bound = 1e-4 * fr[idx].abs() + 2e-5 * rms
    assert bool((d <= bound).all()), f"{what}: {(d > bound).sum().item()} of {idx.numel()} sampled elements outside 1e-4 |ref| + 2e-5 rms"


@pytest.mark.parametrize("shape", SHAPES, ids=IDS)
def test_linear_vs_fp64(shape):
    from dmvae_amd import ops
    m, n, k = shape
    x, w, b = _operands(m, n, k)
    ref = _ref64(x, w, b)
    y32 = ops.linear_bf16(x, w, b, out_f32=True)
    _check(y32, ref, "f32 result %s" % (shape,))
    y16 = ops.linear_bf16(x, w, b)
    assert torch.equal(y16, y32.to(BF)), "bf16 result is not RNE(f32 result)"
    assert torch.equal(y16, ops.linear_bf16(x, w, b)), "rerun differs"
    # bf16 bias, the operand autocast hands the library: the bias is widened to f32 and added to the f32 accumulator
    yb = ops.linear_bf16(x, w, b.to(BF), out_f32=True)
    _check(yb, _ref64(x, w, b.to(BF).float()), "bf16-bias result %s" % (shape,))
    # no bias
    yn = ops.linear_bf16(x, w, None, out_f32=True)
    _check(yn, x.double() @ w.double().t(), "no-bias result %s" % (shape,))


@pytest.mark.parametrize("shape", [(8224, 4096, 1024), (4096, 1152, 1152), (200, 72, 416)], ids=["fc1", "dit_proj", "ragged"])
def test_fused_activation_is_the_two_kernel_route(shape):
    from dmvae_amd import ops
    m, n, k = shape
    x, w, b = _operands(m, n, k, seed=1)
    h = ops.linear_bf16(x, w, b.to(BF))
    g = ops.linear_bf16(x, w, b.to(BF), act=ops.ACT_GELU)
    assert torch.equal(g, ops.gelu(h.view(-1)).view_as(h)), "fused GELU differs from Linear -> gelu kernel"
    s = ops.linear_bf16(x, w, b.to(BF), act=ops.ACT_SILU)
    assert torch.equal(s, ops.silu(h.view(-1)).view_as(h)), "fused SiLU differs from Linear -> silu kernel"
    # and against fp64 on the bf16-rounded pre-activation: erf-form GELU, one bf16 rounding of the result (2^-8 relative) + the f32 evaluation's
    # absolute error where 1 + erf cancels (x < -2)
    ref = torch.nn.functional.gelu(h.double())
    d = (g.double() - ref).abs()
    bound = 2.0 ** -8 * ref.abs() * 1.01 + 1e-6
    assert bool((d <= bound).all()), f"fused GELU vs fp64: {(d > bound).sum().item()} elements outside half a bf16 ulp + 1e-6, worst {((d - bound).max().item()):.3e}"


@pytest.mark.parametrize("shape", [(4096, 6144, 1152), (16384, 6144, 1152), (1000, 6144, 1152), (200, 80, 416), (300, 528, 384)],
                         ids=["dit16_w12", "dit64_w12", "ragged_rows", "ragged_small", "half_tile_columns"])
@pytest.mark.parametrize("bias", ["bf16", "f32", "none"])
def test_fused_swiglu_is_the_two_kernel_route(shape, bias):
    """act = ACT_SWIGLU (swiglu_ffn.py:32-35: x1, x2 = w12(x).chunk(2); silu(x1) * x2): the epilogue pairs column h with column H + h of the bf16-rounded
    Linear output -- bit-identical to `swiglu(linear(.))`, row-major and K-tile-major weights, and against fp64 within one bf16 rounding of silu and one of the product."""
    from dmvae_amd import ops
    m, n, k = shape
    x, w, b = _operands(m, n, k, seed=7)
    bb = None if bias == "none" else (b.to(BF) if bias == "bf16" else b)
    h = ops.linear_bf16(x, w, bb)
    want = ops.swiglu(h)
    got = ops.linear_bf16(x, w, bb, act=ops.ACT_SWIGLU)
    assert got.shape == (m, n // 2)
    assert torch.equal(got, want), f"fused SwiGLU differs from Linear -> swiglu kernel in {(got != want).sum().item()} elements"
    wk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32)
    assert torch.equal(ops.linear_bf16(x, wk, bb, act=ops.ACT_SWIGLU), want), "K-tile-major weights"
    x1, x2 = h.double().chunk(2, dim=-1)
    ref = torch.nn.functional.silu(x1) * x2
    d = (got.double() - ref).abs()
    bound = 2.0 ** -7 * ref.abs() * 1.01 + 1e-6
    assert bool((d <= bound).all()), f"fused SwiGLU vs fp64: {(d > bound).sum().item()} elements outside two bf16 roundings"


def test_fused_swiglu_writes_nothing_past_its_rows_and_columns():
    from dmvae_amd import _lib, ops
    m, n, k = 200, 80, 416
    x, w, b = _operands(m, n, k, seed=8)
    want = ops.swiglu(ops.linear_bf16(x, w, b.to(BF)))
    ldy = 56                                      # H = 40 columns of a 56-wide row: the pad columns and the rows after M keep their sentinel
    buf = torch.full((m + 64, ldy), -7.0, device=DEV, dtype=BF)
    bbf = b.to(BF)
    ops.check(_lib.lib().dmvae_linear_bf16(x.data_ptr(), w.data_ptr(), bbf.data_ptr(), buf.data_ptr(), m, n, k, k, k, ldy, ops.ACT_SWIGLU, 1, 0, 0, ops._stream()), "linear_bf16")
    torch.cuda.synchronize()
    assert torch.equal(buf[:m, : n // 2], want)
    assert bool((buf[:m, n // 2:] == -7.0).all()) and bool((buf[m:] == -7.0).all())


@pytest.mark.parametrize("shape", [(8224, 1024, 4096), (4096, 1152, 1152), (200, 72, 416)], ids=["fc2", "dit_proj", "ragged"])
def test_kmajor_weight_layout_is_bit_identical(shape):
    """w_layout = 1: the K-tile-major copy [K / 32][N][32] written by the pack kernel gives the same bits as the row-major operand (same values, same K order);
    and it is the input-gradient operand: against the K-tile-major copy of the TRANSPOSED weight the call computes dY . W."""
    from dmvae_amd import ops
    m, n, k = shape
    x, w, b = _operands(m, n, k, seed=4)
    w32 = w.float()                                   # bf16-exact f32 master
    wp = ops.pack_conv_weight(w32, kmajor=True)
    wk = wp._dmvae_kmajor.view(k // 32, n, 32)
    assert torch.equal(wp.view(n, k), w)
    y0 = ops.linear_bf16(x, w, b, out_f32=True)
    assert torch.equal(ops.linear_bf16(x, wk, b, out_f32=True), y0), "K-tile-major operand differs from the row-major one"
    if n % 32 == 0:
        g = torch.Generator(device=DEV).manual_seed(5)
        dy = torch.randn(m, n, device=DEV, generator=g).to(BF)
        wt = ops.pack_conv_weight(w32, for_dgrad=True, kmajor=True)        # [k, 1, n] = W^T, and its K-tile-major copy [n / 32, 1, k, 32]
        dx = ops.linear_bf16(dy, wt._dmvae_kmajor.view(n // 32, k, 32), out_f32=True)
        assert torch.equal(dx, ops.linear_bf16(dy, wt.view(k, n), out_f32=True))
        assert torch.equal(ops.linear_weight_t_kmajor(w), wt._dmvae_kmajor.view(n // 32, k, 32)), "tiled transpose != the element-wise pack's K-tile-major copy"
        _check(dx, dy.double() @ w.double(), "input gradient %s" % (shape,))


def test_rows_and_columns_past_the_end_are_not_written():
    """The output sits in the middle of a sentinel-filled buffer; M and N are not multiples of any tile (ragged rows AND columns), and leading dimensions
    larger than the rows are honoured through the C ABI."""
    from dmvae_amd import _lib, ops
    m, n, k = 1000, 200, 512
    x, w, b = _operands(m, n, k, seed=2)
    ldy = 208
    pad = 4096
    buf = torch.full((pad + m * ldy + pad,), -7.0, device=DEV, dtype=BF)
    y = buf[pad:pad + m * ldy].view(m, ldy)
    rc = _lib.lib().dmvae_linear_bf16(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), m, n, k, k, k, ldy, 0, 0, 0, 0,
                                      torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert bool((buf[:pad] == -7.0).all()) and bool((buf[pad + m * ldy:] == -7.0).all()), "wrote outside the output"
    assert bool((y[:, n:] == -7.0).all()), "wrote columns past N"
    assert torch.equal(y[:, :n], ops.linear_bf16(x, w, b)), "strided result differs from the dense one"


def test_argument_errors():
    from dmvae_amd import _lib
    L = _lib.lib()
    x = torch.zeros(64, 512, device=DEV, dtype=BF)
    st = torch.cuda.current_stream().cuda_stream
    assert L.dmvae_linear_bf16(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 64, 64, 200, 200, 200, 64, 0, 0, 0, 0, st) != 0      # K % 32
    assert b"K" in L.dmvae_last_error()
    assert L.dmvae_linear_bf16(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 64, 64, 352, 352, 352, 64, 0, 0, 0, 0, st) != 0      # K < 384
    assert L.dmvae_linear_bf16(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 64, 60, 512, 512, 512, 64, 0, 0, 0, 0, st) != 0      # N % 8
    assert L.dmvae_linear_bf16(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 64, 64, 512, 512, 512, 64, 2, 0, 0, 0, st) != 0      # act
    assert L.dmvae_linear_bf16(None, x.data_ptr(), None, x.data_ptr(), 64, 64, 512, 512, 512, 64, 0, 0, 0, 0, st) != 0


_TILE_SCRIPT = r"""
import sys, torch
sys.path.insert(0, %r)
from dmvae_amd import ops, _lib
_lib.lib().dmvae_debug_gemm_cfg(int(sys.argv[2]))       # force one menu entry for the whole process
g = torch.Generator(device="cuda").manual_seed(3)
out = {}
for m, n, k in [(8224, 1024, 1024), (4096, 1152, 1152), (777, 520, 384), (777, 544, 384), (300, 264, 416)]:
    x = torch.randn(m, k, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(torch.bfloat16)
    b = torch.randn(n, device="cuda", generator=g)
    out[(m, n, k)] = ops.linear_bf16(x, w, b, out_f32=True).cpu()
    if n %% 16 == 0:
        out[("swiglu", m, n, k)] = ops.linear_bf16(x, w, b.to(torch.bfloat16), act=ops.ACT_SWIGLU).cpu()
torch.save(out, sys.argv[1])
"""


def test_every_tile_of_the_menu_gives_the_same_bits(tmp_path):
    """dmvae_debug_gemm_cfg forces one menu entry; each runs in a fresh process (first-launch attributes, caches)."""
    res = []
    for cfg in range(12):
        f = tmp_path / ("cfg%d.pt" % cfg)
        subprocess.run([sys.executable, "-c", _TILE_SCRIPT % ROOT, str(f), str(cfg)], check=True, timeout=600)
        res.append(torch.load(f))
    for cfg in range(1, 12):
        for key in res[0]:
            assert torch.equal(res[0][key], res[cfg][key]), f"tile {cfg} differs from tile 0 at {key}"


@pytest.mark.parametrize("batch,m,n,k,out_f32", [(4, 1024, 1024, 512, True), (3, 1024, 512, 1024, False), (2, 300, 136, 384, True), (5, 256, 128, 512, False),
                                                 (33, 512, 512, 512, False)])
def test_batched_products_on_the_large_tile_kernel(batch, m, n, k, out_f32, monkeypatch):
    """ops.gemm_nt on csrc/gemm_pp.hip's BATCHED instantiations (a batch index in the tile decode: the decoder attention's per-sample GEMMs, flux_ae.py:37-49)
    against f64 on the same bf16 operands and against the 128 x 128-tile kernel it replaces; ragged rows / columns per product; rows past M of one product must not
    leak into the next one's result (sentinel check: each product against ITS OWN operands only)."""
    from conftest import rel_err
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(batch * 1000 + m + n + k)
    a = torch.randn(batch, m, k, generator=g).to("cuda").to(torch.bfloat16)
    b = (torch.randn(batch, n, k, generator=g) * 0.1).to("cuda").to(torch.bfloat16)
    monkeypatch.setattr(ops, "GEMM_NT_LARGE_TILES", True)
    c1 = ops.gemm_nt(a, b, out_f32=out_f32)
    monkeypatch.setattr(ops, "GEMM_NT_LARGE_TILES", False)
    c0 = ops.gemm_nt(a, b, out_f32=out_f32)
    ref = a.double() @ b.double().transpose(1, 2)
    assert c1.shape == (batch, m, n) and c1.dtype == (torch.float32 if out_f32 else torch.bfloat16)
    tol = 1e-5 if out_f32 else 2 ** -8
    for i in range(batch):
        assert rel_err(c1[i].double(), ref[i]) < tol, i
    assert rel_err(c1.double(), c0.double()) < (1e-5 if out_f32 else 2 ** -7)
    monkeypatch.setattr(ops, "GEMM_NT_LARGE_TILES", True)
    assert torch.equal(ops.gemm_nt(a, b, out_f32=out_f32), c1)


@pytest.mark.parametrize("shape", [(4096, 6144, 1152), (16384, 6144, 1152), (4112, 1024, 768), (300, 64, 416)], ids=lambda s: "%dx%dx%d" % s)
def test_swiglu_epilogue_that_also_stores_the_pre_activation(shape):
    """dmvae_linear_bf16_swiglu_pre (swiglu_ffn.py:31-36 in one launch): x12 and g bit-identical to the Linear followed by the swiglu pass; row-major and
    K-tile-major weights; f32 and bf16 bias; nothing written past M."""
    from dmvae_amd import ops
    m, n, k = shape
    x, w, b = _operands(m, n, k, seed=5)
    x12_ref = ops.linear_bf16(x, w, b)
    g_ref = ops.swiglu(x12_ref)
    g, x12 = ops.linear_swiglu_pre(x, w, b)
    assert torch.equal(x12, x12_ref) and torch.equal(g, g_ref)
    g2, x122 = ops.linear_swiglu_pre(x, w, b.to(BF))
    assert torch.equal(x122, ops.linear_bf16(x, w, b.to(BF))) and torch.equal(g2, ops.swiglu(x122))
    if k % 32 == 0:
        wk = ops.pack_conv_weight(w.float(), kmajor=True)._dmvae_kmajor.view(k // 32, n, 32)
        g3, x123 = ops.linear_swiglu_pre(x, wk, b)
        assert torch.equal(g3, g) and torch.equal(x123, x12)
    g4, x124 = ops.linear_swiglu_pre(x, w, None)
    assert torch.equal(x124, ops.linear_bf16(x, w, None)) and torch.equal(g4, ops.swiglu(x124))
