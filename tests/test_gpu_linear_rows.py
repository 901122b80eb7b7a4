"""csrc/linear_rows.hip (-m gpu): nn.Linear on one row per SAMPLE -- the per-sample conditioning Linears of LightningDiT (adaLN modulations, timestep embedder;
diffusion/lightningdit/lightningdit.py:96-139,236-240,266-268) and their input gradients -- against fp64 on the same bf16 operands, the fused SiLU against the
two-kernel route, rows / columns past the end untouched, reruns bit-identical, and `functional.linear` / `LinearFn` routing rows <= 64 to it."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

SHAPES = [(16, 6912, 1152), (32, 6912, 1152), (64, 2304, 1152), (16, 1152, 256), (5, 1152, 6912), (64, 1152, 6912), (1, 36, 64), (33, 200, 96), (48, 1152, 3072)]


def _ops(m, n, k, seed=0, wstd=0.05):
    g = torch.Generator().manual_seed(1000 * m + n + k + seed)
    x = torch.randn(m, k, generator=g).to(DEV).to(BF)
    w = (torch.randn(n, k, generator=g) * wstd).to(DEV).to(BF)
    b = torch.randn(n, generator=g).to(DEV)
    return x, w, b


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_linear_rows_vs_fp64(m, n, k):
    from dmvae_amd import ops
    assert ops.linear_rows_supported(m, n, k)
    x, w, b = _ops(m, n, k)
    ref = x.double() @ w.double().t() + b.double()
    y32 = ops.linear_rows(x, w, b, out_f32=True)
    assert y32.dtype == torch.float32 and rel_err(y32, ref) < 1e-5                     # f32 accumulation of exact bf16 products
    y = ops.linear_rows(x, w, b)
    assert y.dtype == BF and torch.equal(y, y32.to(BF))                               # the bf16 result is RNE of the f32 one
    yb = ops.linear_rows(x, w, b.to(BF))                                              # bf16 bias (what autocast hands the library), added in f32
    assert rel_err(yb.float(), x.double() @ w.double().t() + b.to(BF).double()) < 2 ** -8
    assert torch.equal(ops.linear_rows(x, w, None, out_f32=True) + b, y32) or rel_err(ops.linear_rows(x, w, None, out_f32=True) + b, ref) < 1e-5
    for _ in range(2):                                                                # fixed-order reduction: reruns give the same bits
        assert torch.equal(ops.linear_rows(x, w, b, out_f32=True), y32)


@pytest.mark.parametrize("m,n,k", [(16, 1152, 6912), (40, 1152, 1152), (3, 64, 256)])
def test_linear_rows_k_tile_major_weight_gives_the_same_bits(m, n, k):
    """w_layout = 1: the K-tile-major copy [K / 32][N][32] (what the input gradient reads: `functional._bf_t`) against the row-major operand."""
    from dmvae_amd import ops
    x, w, b = _ops(m, n, k, seed=9)
    wk = w.view(n, k // 32, 32).permute(1, 0, 2).contiguous()
    assert torch.equal(ops.linear_rows(x, wk, b, out_f32=True), ops.linear_rows(x, w, b, out_f32=True))
    wt = ops.linear_weight_t_kmajor(w) if n % 32 == 0 else None          # the library's own transposed K-tile-major pack of w [n, k]: [n / 32][k][32]
    if wt is not None:                                                   # = the operand of y2 [m, k] = x2 [m, n] @ w   (the input-gradient form: reduction over n)
        x2 = torch.randn(m, n, generator=torch.Generator().manual_seed(2)).to(DEV).to(BF)
        assert rel_err(ops.linear_rows(x2, wt, None, out_f32=True), x2.double() @ w.double()) < 1e-5


@pytest.mark.parametrize("m,n,k", [(16, 6912, 1152), (64, 2304, 1152), (5, 1152, 256), (33, 40, 2056), (1, 17, 8)])
def test_linear_rows_wgrad_vs_fp64(m, n, k):
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(m * n + k)
    dy = torch.randn(m, n, generator=g).to(DEV).to(BF)
    x = torch.randn(m, k, generator=g).to(DEV).to(BF)
    dw, db = ops.linear_rows_wgrad(dy, x)
    assert rel_err(dw, dy.double().t() @ x.double()) < 1e-5 and rel_err(db, dy.double().sum(0)) < 1e-5
    dw2, db2 = ops.linear_rows_wgrad(dy, x, dw_out=dw.clone(), db_out=db.clone(), accumulate=True)
    assert rel_err(dw2, 2 * (dy.double().t() @ x.double())) < 1e-5 and rel_err(db2, 2 * dy.double().sum(0)) < 1e-5
    dw3, none = ops.linear_rows_wgrad(dy, x, need_bias=False)
    assert none is None and torch.equal(dw3, dw)


def test_linear_rows_fused_silu_is_the_two_kernels():
    from dmvae_amd import ops
    x, w, b = _ops(16, 1152, 256, seed=3)
    y0 = ops.linear_rows(x, w, b.to(BF))
    assert torch.equal(ops.linear_rows(x, w, b.to(BF), act=ops.ACT_SILU), ops.silu(y0))


def test_linear_rows_strided_operands_and_bounds():
    """Row strides larger than K (a slice of a wider tensor) and a destination with rows / columns the call must not touch."""
    from dmvae_amd import _lib, ops
    m, n, k = 13, 40, 96
    g = torch.Generator().manual_seed(5)
    xw = torch.randn(m, k + 64, generator=g).to(DEV).to(BF)
    ww = (torch.randn(n, k + 32, generator=g) * 0.1).to(DEV).to(BF)
    x, w = xw[:, :k], ww[:, :k]
    ref = x.double() @ w.double().t()
    assert rel_err(ops.linear_rows(x, w, None, out_f32=True), ref) < 1e-5
    # sentinel rows / columns around the result: ldy = n + 8, two spare rows
    buf = torch.full((m + 2, n + 8), 7.0, device=DEV, dtype=BF)
    L = _lib.lib()
    ops.check(L.dmvae_linear_rows_bf16(x.data_ptr(), w.data_ptr(), None, buf.data_ptr(), m, n, k, x.stride(0), w.stride(0), n + 8, 0, 0, 0, 0,
                                       torch.cuda.current_stream().cuda_stream), "linear_rows_bf16")
    assert rel_err(buf[:m, :n].float(), ref) < 2 ** -8
    assert (buf[m:] == 7).all() and (buf[:, n:] == 7).all()
    with pytest.raises(_lib.DmvaeHipError):
        ops.linear_rows(torch.zeros(65, 64, device=DEV, dtype=BF), torch.zeros(8, 64, device=DEV, dtype=BF))      # more than 64 rows: not this kernel's shape


@pytest.mark.parametrize("rows,cin,cout", [(16, 1152, 6912), (32, 256, 1152), (7, 192, 1152)])
def test_linear_fn_per_sample_rows_forward_backward(rows, cin, cout):
    """functional.LinearFn with <= 64 rows: forward and input gradient on csrc/linear_rows.hip, weight / bias gradient on the split-K kernel -- against fp64
    autograd on the same bf16-rounded operands (adaLN_modulation[1] of a DiT block: 16 samples x 1152 -> 6912)."""
    from dmvae_amd.functional import LinearFn
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(cout, cin, generator=g) * 0.03).to(DEV).requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).to(DEV).requires_grad_(True)
    dy = torch.randn(rows, cout, generator=g).to(DEV).to(BF)
    y = LinearFn.apply(x, w, b)
    y.backward(dy)
    xr = x.detach().to(BF).double().requires_grad_(True)
    wr = w.detach().to(BF).double().requires_grad_(True)
    br = b.detach().to(BF).double().requires_grad_(True)
    yr = xr @ wr.t() + br
    yr.backward(dy.double())
    assert y.dtype == BF and rel_err(y.float(), yr.detach()) < 2 ** -8
    assert rel_err(x.grad, xr.grad) < 2 ** -8                      # dx is a bf16 result (autocast's backward), returned in x's dtype
    assert rel_err(w.grad, wr.grad) < 1e-5 and rel_err(b.grad, br.grad) < 1e-5
