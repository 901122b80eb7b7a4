"""Kernel-level parity (-m gpu): each C-ABI entry point vs the CPU oracle / an fp64 evaluation of the same op on
the SAME bf16-rounded inputs.  Bars: f32 outputs 1e-4 relative-to-max (f32 accumulation-order only);
bf16 outputs within bf16 rounding (2^-8 relative); integer/index ops bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_err
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _ops():
    from dmvae_amd import ops
    return ops


def _conv_ref(x, w, b, r, ks, ups, act):
    xr = x.float().cpu().double().permute(0, 3, 1, 2)
    if ups:
        xr = xr.repeat_interleave(2, 2).repeat_interleave(2, 3)
    y = F.conv2d(xr, w.float().cpu().double(), None if b is None else b.cpu().double(), padding=ks // 2)
    if r is not None:
        y = y + r.float().cpu().double().permute(0, 3, 1, 2)
    if act == 1:
        y = y * torch.sigmoid(y)
    if act == 2:
        y = y.relu()
    return y.permute(0, 2, 3, 1)


CONV_CASES = [  # N, H, W, Cin, Cout, ks, ups
    (2, 8, 8, 64, 64, 3, 0), (2, 8, 8, 64, 128, 3, 0), (1, 16, 16, 128, 64, 3, 0), (3, 5, 7, 32, 32, 3, 0),
    (2, 8, 8, 64, 64, 1, 0), (1, 32, 32, 512, 512, 3, 0), (2, 4, 4, 32, 96, 3, 1), (1, 8, 8, 64, 256, 1, 0),
    (1, 1, 1, 32, 4, 3, 0), (1, 2, 130, 64, 36, 3, 0),
    # a handful of output channels on >= 4096 pixels: the 32 x 256 tile of the general kernel
    (2, 64, 64, 128, 4, 3, 0), (1, 70, 90, 64, 12, 3, 0), (1, 64, 64, 64, 32, 3, 1), (1, 64, 65, 64, 4, 1, 0),
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("act", [0, 1, 2])
def test_conv_fwd(case, act):
    ops = _ops()
    n, h, w_, cin, cout, ks, ups = case
    g = torch.Generator(device="cpu").manual_seed(hash(case) % 1000)
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, ks, ks, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    ho, wo = (2 * h, 2 * w_) if ups else (h, w_)
    r = torch.randn(n, ho, wo, cout, generator=g).to(DEV).to(BF)
    wp = ops.pack_conv_weight(w)
    ref = _conv_ref(x, w.to(BF), b, r, ks, ups, act)
    y32 = ops.conv2d_nhwc(x, wp, b, r, ks=ks, upsample=bool(ups), act=act, out_f32=True)
    assert rel_err(y32.cpu(), ref) < 1e-5
    y16 = ops.conv2d_nhwc(x, wp, b, r, ks=ks, upsample=bool(ups), act=act)
    assert torch.equal(y16, y32.to(BF))        # bf16 store == RNE of the f32 result, bit-exact


S2_CASES = [(2, 6, 6, 32, 32), (1, 20, 28, 96, 72), (3, 16, 16, 64, 128), (1, 64, 64, 128, 128), (1, 2, 2, 32, 8),
            (4, 128, 128, 64, 128), (2, 192, 100, 96, 64)]  # N, H, W, Cin, Cout; the last two are large enough for the ping-pong kernel (>= 16384 output pixels)


@pytest.mark.parametrize("case", S2_CASES)
def test_conv_stride2_fwd_dgrad_wgrad(case):
    """Downsample conv (flux_ae.py:85-95): F.pad(x, (0,1,0,1)) + conv3x3 stride 2, its input gradient (the stride-1 dgrad conv
    over dy zero-inserted at odd positions, desc.upsample = 2) and its weight / bias gradient (desc.stride = 2), against fp64
    autograd on the same bf16-rounded operands."""
    ops = _ops()
    n, h, w_, cin, cout = case
    g = torch.Generator().manual_seed(31 + cin + h)
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    dy = torch.randn(n, h // 2, w_ // 2, cout, generator=g).to(DEV).to(BF)
    xr = x.float().cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.to(BF).float().cpu().double().requires_grad_(True)
    br = b.cpu().double().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (0, 1, 0, 1)), wr, br, stride=2)
    yr.backward(dy.float().cpu().double().permute(0, 3, 1, 2))
    y32 = ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, ks=3, stride=2, out_f32=True)
    assert y32.shape == (n, h // 2, w_ // 2, cout)
    assert rel_err(y32.cpu(), yr.detach().permute(0, 2, 3, 1)) < 1e-5
    assert torch.equal(ops.conv2d_nhwc(x, ops.pack_conv_weight(w), b, ks=3, stride=2), y32.to(BF))
    if cout % 32 == 0:     # dy is the dgrad conv's Cin
        dx = ops.conv2d_nhwc(dy, ops.pack_conv_weight(w, for_dgrad=True), ks=3, upsample=2, out_f32=True)
        assert dx.shape == (n, h, w_, cin)
        assert rel_err(dx.cpu(), xr.grad.permute(0, 2, 3, 1)) < 1e-5
    dw, db = ops.conv2d_nhwc_wgrad(dy, x, 3, stride=2)
    assert rel_err(dw.cpu(), wr.grad) < 1e-5
    assert rel_err(db.cpu(), br.grad) < 1e-5
    dw2, db2 = ops.conv2d_nhwc_wgrad(dy, x, 3, stride=2)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)          # deterministic split-K


K4_CASES = [(2, 8, 8, 32, 64, 2), (1, 9, 11, 32, 32, 1), (3, 16, 16, 64, 128, 2), (1, 7, 10, 96, 8, 1), (2, 64, 64, 64, 128, 2), (2, 33, 32, 256, 512, 1),
            (1, 12, 10, 32, 4, 1), (4, 130, 128, 64, 128, 2), (2, 96, 97, 64, 192, 1), (5, 66, 64, 32, 64, 1),   # N, H, W, Cin, Cout, stride; these three reach the ping-pong kernel
            (2, 128, 128, 128, 256, 2), (1, 66, 128, 256, 128, 2), (3, 64, 64, 128, 128, 2)]  # stride 2 with 128-multiples and 32-pixel output rows: wgrad_pp's S2 instantiation (both tile configurations)


@pytest.mark.parametrize("case", K4_CASES)
def test_conv_k4_fwd_dgrad_wgrad(case):
    """4x4 conv, padding 1, stride 1 | 2 (models/patchgan.py:125-147) as a gather in the conv kernels: forward, the transposed gather for
    the input gradient, and the weight / bias gradient, against fp64 autograd on the same bf16-rounded operands; forward also against the
    im2col + GEMM formulation (an independent route through other kernels)."""
    ops = _ops()
    n, h, w_, cin, cout, stride = case
    g = torch.Generator().manual_seed(41 + cin + h)
    x = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, 4, 4, generator=g) * 0.05).to(DEV)
    b = torch.randn(cout, generator=g).to(DEV)
    ho, wo = (h - 2) // stride + 1, (w_ - 2) // stride + 1
    dy = torch.randn(n, ho, wo, cout, generator=g).to(DEV).to(BF)
    xr = x.float().cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.to(BF).float().cpu().double().requires_grad_(True)
    br = b.cpu().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=stride, padding=1)
    yr.backward(dy.float().cpu().double().permute(0, 3, 1, 2))
    wp = ops.pack_conv_weight(w)
    assert wp.shape == (cout, 16, cin)
    y32 = ops.conv2d_nhwc(x, wp, b, ks=4, stride=stride, out_f32=True)
    assert y32.shape == (n, ho, wo, cout)
    assert rel_err(y32.cpu(), yr.detach().permute(0, 2, 3, 1)) < 1e-5
    for act in (0, 4):
        ya = ops.conv2d_nhwc(x, wp, b, ks=4, stride=stride, act=act)
        want = y32 if act == 0 else torch.where(y32 > 0, y32, 0.2 * y32)
        assert torch.equal(ya, want.to(BF))
    col = ops.im2col(x, 4, stride, 1)
    y_col = ops.conv2d_nhwc(col.view(1, 1, n * ho * wo, 16 * cin), wp.view(cout, 1, 16 * cin), b, ks=1, out_f32=True).view(n, ho, wo, cout)
    assert rel_err(y32, y_col) < 1e-5
    if cout % 32 == 0:
        dx = ops.conv2d_nhwc(dy, ops.pack_conv_weight(w, for_dgrad=True), ks=4, stride=stride, transposed=True, out_f32=True)
        hh, ww = (ho - 1) * stride + 2, (wo - 1) * stride + 2
        assert dx.shape == (n, hh, ww, cin)
        ref = xr.grad.permute(0, 2, 3, 1)
        assert rel_err(dx.cpu(), ref[:, :hh, :ww]) < 1e-5
        assert ref[:, hh:].abs().max() == 0 if hh < h else True           # rows the strided conv never read
    if cout % 8 == 0:
        dw, db = ops.conv2d_nhwc_wgrad(dy, x, 4, stride=stride)
        assert dw.shape == (cout, cin, 4, 4)
        assert rel_err(dw.cpu(), wr.grad) < 1e-5
        assert rel_err(db.cpu(), br.grad) < 1e-5


def test_conv_stride2_rejects_odd_sizes():
    ops = _ops()
    from dmvae_amd._lib import DmvaeHipError
    x = torch.zeros(1, 5, 6, 32, device=DEV, dtype=BF)
    with pytest.raises(DmvaeHipError):
        ops.conv2d_nhwc(x, torch.zeros(32, 9, 32, device=DEV, dtype=BF), ks=3, stride=2)


@pytest.mark.parametrize("case", CONV_CASES[:8] + [(2, 16, 16, 32, 512, 3, 0), (2, 3, 3, 40, 24, 3, 0)])
def test_conv_wgrad_and_dgrad(case):
    ops = _ops()
    n, h, w_, cin, cout, ks, ups = case
    g = torch.Generator().manual_seed(7)
    ho, wo = (2 * h, 2 * w_) if ups else (h, w_)
    a = torch.randn(n, h, w_, cin, generator=g).to(DEV).to(BF)
    dy = torch.randn(n, ho, wo, cout, generator=g).to(DEV).to(BF)
    w = (torch.randn(cout, cin, ks, ks, generator=g) * 0.05).to(DEV)
    xr = a.float().cpu().double().permute(0, 3, 1, 2)
    if ups:
        xr = xr.repeat_interleave(2, 2).repeat_interleave(2, 3)
    xr.requires_grad_(True)
    wr = w.to(BF).float().cpu().double().requires_grad_(True)
    br = torch.zeros(cout, dtype=torch.double, requires_grad=True)
    F.conv2d(xr, wr, br, padding=ks // 2).backward(dy.float().cpu().double().permute(0, 3, 1, 2))
    dw, db = ops.conv2d_nhwc_wgrad(dy, a, ks, upsample=bool(ups))
    assert rel_err(dw.cpu(), wr.grad) < 1e-5 and rel_err(db.cpu(), br.grad) < 1e-5
    dw2, db2 = ops.conv2d_nhwc_wgrad(dy, a, ks, upsample=bool(ups), dw_out=dw.clone(), db_out=db.clone(), accumulate=True)
    assert rel_err(dw2.cpu(), 2 * wr.grad) < 1e-5 and rel_err(db2.cpu(), 2 * br.grad) < 1e-5
    if cout % 32 == 0 and cin % 4 == 0:
        dx = ops.conv2d_nhwc(dy, ops.pack_conv_weight(w, for_dgrad=True), ks=ks, out_f32=True)
        assert rel_err(dx.cpu(), xr.grad.permute(0, 2, 3, 1)) < 1e-5
        if ups:  # nearest-x2 backward == 2x2 sum pool, checked on the bf16 path
            dxl = ops.sumpool2x2(dx.to(BF)).float().cpu()
            ref = dx.to(BF).float().cpu().reshape(n, h, 2, w_, 2, cin).sum(dim=(2, 4))
            assert rel_err(dxl, ref) < 8e-3


@pytest.mark.parametrize("shape", [(2, 8, 8, 64), (2, 8, 8, 32), (1, 16, 16, 128), (3, 5, 7, 96), (2, 32, 32, 512), (1, 64, 64, 256)])
@pytest.mark.parametrize("swish", [True, False])
def test_groupnorm(shape, swish):
    ops = _ops()
    n, h, w_, c = shape
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(n, h, w_, c, generator=g) * 1.5 + 0.3).to(DEV).to(BF)
    gamma = (torch.randn(c, generator=g) * 0.5 + 1).to(DEV)
    beta = (torch.randn(c, generator=g) * 0.2).to(DEV)
    da = torch.randn(n, h, w_, c, generator=g).to(DEV).to(BF)
    dres = torch.randn(n, h, w_, c, generator=g).to(DEV).to(BF)
    st = ops.groupnorm_stats(x)
    y = ops.groupnorm_apply(x, st, gamma, beta, swish)
    dx, dg, db = ops.groupnorm_bwd(da, x, st, gamma, beta, swish, dres=dres)
    xr = x.float().cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = gamma.cpu().double().requires_grad_(True), beta.cpu().double().requires_grad_(True)
    yr = R.group_norm(xr, gr, br)
    if swish:
        yr = R.swish(yr)
    yr.backward(da.float().cpu().double().permute(0, 3, 1, 2))
    xg = xr.detach().reshape(n, 32, -1)
    assert rel_err(st[..., 0].cpu(), xg.mean(-1)) < 1e-5
    assert rel_err(st[..., 1].cpu(), 1 / torch.sqrt(xg.var(-1, unbiased=False) + 1e-6)) < 1e-5
    assert rel_err(y.float().cpu(), yr.detach().permute(0, 2, 3, 1)) < 4e-3          # bf16 store
    assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1) + dres.float().cpu().double()) < 4e-3
    assert rel_err(dg.cpu(), gr.grad) < 1e-5 and rel_err(db.cpu(), br.grad) < 1e-5


def test_gemm_nt_tn_softmax_transpose():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    a = torch.randn(3, 128, 96, generator=g).to(DEV).to(BF)
    b = torch.randn(3, 72, 96, generator=g).to(DEV).to(BF)
    c = ops.gemm_nt(a, b, out_f32=True)
    assert rel_err(c.cpu(), a.float().cpu().double() @ b.float().cpu().double().transpose(1, 2)) < 1e-5
    shared = torch.randn(72, 96, generator=g).to(DEV).to(BF)
    c2 = ops.gemm_nt(a, shared, out_f32=True)
    assert rel_err(c2.cpu(), a.float().cpu().double() @ shared.float().cpu().double().t()) < 1e-5
    p = torch.randn(3, 200, 64, generator=g).to(DEV).to(BF)
    q = torch.randn(3, 200, 40, generator=g).to(DEV).to(BF)
    d = ops.gemm_tn(p, q, alpha=0.5, out_f32=True)
    assert rel_err(d.cpu(), 0.5 * p.float().cpu().double().transpose(1, 2) @ q.float().cpu().double()) < 1e-5
    s = torch.randn(2, 50, 130, generator=g).to(DEV) * 3
    pr = ops.softmax_rows(s, 0.25)
    ref = torch.softmax(s.cpu().double() * 0.25, -1)
    assert rel_err(pr.float().cpu(), ref) < 4e-3
    dp = torch.randn(2, 50, 130, generator=g).to(DEV)
    ds = ops.softmax_rows_bwd(dp, pr, 0.25)
    pd = pr.float().cpu().double()
    refds = 0.25 * pd * (dp.cpu().double() - (dp.cpu().double() * pd).sum(-1, keepdim=True))
    assert rel_err(ds.float().cpu(), refds) < 4e-3
    t = ops.transpose_last2(a)
    assert torch.equal(t, a.transpose(1, 2).contiguous())                     # pure index op: bit-exact


@pytest.mark.parametrize("cols", [256, 512, 1024, 2048])
def test_softmax_rows_register_resident_forms(cols):
    """cols = 256 NV: the row in registers (16-B loads, one exponential per element, 8-B stores: the decoder attention's S = 1024) against f64 -- every element
    within bf16 rounding of the f64 softmax, rows summing to 1 within their 2^-9 elements' worth -- and against the three-pass form on a neighbouring width."""
    ops = _ops()
    g = torch.Generator().manual_seed(cols)
    s = (torch.randn(3, 37, cols, generator=g) * 4).to(DEV)
    pr = ops.softmax_rows(s, 0.3)
    ref = torch.softmax(s.cpu().double() * 0.3, -1)
    assert pr.dtype == BF and ((pr.float().cpu().double() - ref).abs() <= 2.0 ** -8 * ref + 1e-30).all()
    assert (pr.float().sum(-1) - 1).abs().max().item() < 3e-3
    dp = torch.randn(3, 37, cols, generator=g).to(DEV)
    ds = ops.softmax_rows_bwd(dp, pr, 0.3)
    pd = pr.float().cpu().double()
    refds = 0.3 * pd * (dp.cpu().double() - (dp.cpu().double() * pd).sum(-1, keepdim=True))
    assert ((ds.float().cpu().double() - refds).abs() <= 2.0 ** -8 * refds.abs() + 1e-6 * refds.abs().max()).all()
    # the generic (three-pass) kernel on the same rows with eight extra columns of very negative scores: the same probabilities on the first `cols` columns
    s2 = torch.cat([s, torch.full((3, 37, 8), -1e4, device=DEV)], -1).contiguous()
    p2 = ops.softmax_rows(s2, 0.3)
    assert not p2[..., cols:].any() and (p2[..., :cols].float() - pr.float()).abs().max().item() <= 2.0 ** -8 * pr.float().max().item()
    for _ in range(2):
        assert torch.equal(ops.softmax_rows(s, 0.3), pr) and torch.equal(ops.softmax_rows_bwd(dp, pr, 0.3), ds)


def test_layout_ops_bit_exact():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 3, 10, 12, generator=g).to(DEV)
    nhwc = ops.nchw_to_nhwc_bf16(x, c_pad=8)
    assert torch.equal(nhwc[..., :3], x.permute(0, 2, 3, 1).to(BF)) and (nhwc[..., 3:] == 0).all()
    assert torch.equal(ops.nhwc_to_nchw_f32(nhwc, 3), x.to(BF).float())
    w = torch.randn(24, 40, 3, 3, generator=g).to(DEV)
    pw = ops.pack_conv_weight(w)
    assert torch.equal(pw, w.permute(0, 2, 3, 1).reshape(24, 9, 40).to(BF))
    pd = ops.pack_conv_weight(w, for_dgrad=True)
    assert torch.equal(pd, w.flip(2, 3).permute(1, 2, 3, 0).reshape(40, 9, 24).to(BF))
    tok = load_golden("latents_to_spatial")
    from dmvae_amd.losses import latents_to_spatial
    assert torch.equal(latents_to_spatial(tok.t("tokens").to(DEV)).cpu(), tok.t("spatial"))


def test_l1_mse():
    ops = _ops()
    g = load_golden("gen_loss")
    rec, img = g.t("recon").to(DEV), g.t("images").to(DEV)
    out, grad = ops.l1_mse(rec, img, 1.0, 0.5)
    assert abs(out[0].item() - float(g["L1"])) < 1e-6 and abs(out[1].item() - float(g["L2"])) < 1e-6
    r = g.t("recon").double().requires_grad_(True)
    l1, l2 = R.l1_mse(r, g.t("images").double())
    (l1 + 0.5 * l2).backward()
    assert rel_err(grad.cpu(), r.grad) < 1e-5


def test_lpips_diff_golden():
    ops = _ops()
    g = load_golden("lpips_diff")
    out = torch.zeros(1, device=DEV)
    # golden features are f32; the kernel consumes bf16, so the oracle is evaluated on the same rounded features
    f0 = [g.t(f"f0_{k}").to(BF).float() for k in range(5)]
    f1 = [g.t(f"f1_{k}").to(BF).float().requires_grad_(True) for k in range(5)]
    ws = [g.t(f"w_{k}") for k in range(5)]
    val = R.lpips_from_feats(f0, f1, ws)
    grads = torch.autograd.grad(val, f1)
    n = f0[0].shape[0]
    for k in range(5):
        a = f0[k].permute(0, 2, 3, 1).contiguous().to(DEV).to(BF)
        b = f1[k].detach().permute(0, 2, 3, 1).contiguous().to(DEV).to(BF)
        hw = a.shape[1] * a.shape[2]
        d = ops.lpips_diff(a, b, ws[k].to(DEV), out, 1.0 / (hw * n), True, accumulate=k > 0)
        assert rel_err(d.float().cpu().permute(0, 3, 1, 2), grads[k]) < 4e-3, k
    assert abs(out.item() - val.item()) < 1e-5 * val.item()
    assert abs(out.item() - float(g["value"])) < 2e-2 * float(g["value"])      # vs the reference's f32-feature value


@pytest.mark.parametrize("name", ["dmd_loss_cfg5", "dmd_loss_cfg1"])
def test_dmd_golden(name):
    ops = _ops()
    g = load_golden(name)
    t = (g.t("t_raw") * (float(g["t1"]) - float(g["t0"])) + float(g["t0"])).to(DEV)
    lat, x0 = g.t("latents").to(DEV), g.t("x0").to(DEV)
    xt = ops.dmd_pre(lat, x0, t)
    xt_ref, _ = R.transport_plan(t.cpu(), g.t("x0"), g.t("latents"))
    assert rel_err(xt.cpu(), xt_ref) < 1e-6
    out, dl = ops.dmd_post(lat, xt, t, g.t("v_teacher").to(DEV), g.t("v_student").to(DEV), g.t("v_teacher_u").to(DEV),
                           g.t("v_student_u").to(DEV), cfg=float(g["cfg"]))
    assert abs(out[0].item() - float(g["loss"])) < 1e-4 * float(g["loss"])
    assert abs(out[1].item() - float(g["dmd_gradient_norm"])) < 1e-4 * float(g["dmd_gradient_norm"])
    assert rel_err(dl.cpu(), g.t("dlatents")) < 1e-4


def test_dmd_toy_and_nan_to_num():
    ops = _ops()
    g = load_golden("dmd_loss_toy")
    pts = g.t("points")[:, :, None, None].contiguous().to(DEV)
    t = g.t("t_raw").to(DEV)
    xt = ops.dmd_pre(pts, g.t("x0").to(DEV), t)
    out, dl = ops.dmd_post(pts, xt, t, g.t("v_teacher").to(DEV), g.t("v_student").to(DEV), weight_factor=False)
    assert abs(out[0].item() - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    assert rel_err(dl.cpu().flatten(1), g.t("dpoints")) < 1e-4
    # weight_factor == 0 for a sample (p_real == 0) -> inf/nan -> nan_to_num like torch (train_dmd.py:224)
    z = torch.zeros(1, 8, 2, 2, device=DEV)
    tt = torch.full((1,), 0.5, device=DEV)
    out, dl = ops.dmd_post(z, z, tt, z, z + 1.0)
    ref = torch.nan_to_num((0.5 * torch.ones(1)) / torch.zeros(1))   # p_real - p_student = +0.5, weight_factor = 0
    assert torch.isfinite(dl).all() and dl.flatten()[0].item() == pytest.approx((ref / 32).item(), rel=1e-6)


def test_kl_mmd_vs_spec():
    ops = _ops()
    gen = torch.Generator().manual_seed(11)
    z = (torch.randn(6, 256, 32, generator=gen) * 0.7 + 0.2)
    y = torch.randn(6, 200, 32, generator=gen)
    kl, mmd, dz = ops.kl_mmd(z.to(DEV), y.to(DEV), w_kl=0.3, w_mmd=2.0)
    zr = z.double().requires_grad_(True)
    klr, klm = R.kl_moment(zr)
    mr = R.mmd_rbf(zr, y.double())
    assert rel_err(kl[:32].cpu(), klr) < 1e-4 and abs(kl[32].item() - klm.item()) < 1e-4 * klm.item()
    assert rel_err(mmd.cpu(), mr) < 1e-4
    (0.3 * klm + 2.0 * mr.mean()).backward()
    assert rel_err(dz.cpu(), zr.grad) < 1e-4
    # closed forms: MMD(X,X) = 0, symmetry
    _, m0, _ = ops.kl_mmd(z.to(DEV), z.to(DEV), need_grad=False)
    assert m0.abs().max().item() < 1e-5
    _, mxy, _ = ops.kl_mmd(z[:, :200].contiguous().to(DEV), y.to(DEV), need_grad=False)
    _, myx, _ = ops.kl_mmd(y.to(DEV), z[:, :200].contiguous().to(DEV), need_grad=False)
    assert rel_err(mxy.cpu(), myx.cpu()) < 1e-5


@pytest.mark.parametrize("shape", [(3, 300, 520), (5, 64, 8), (2, 129, 257)])
def test_kl_mmd_ragged_tiles_and_chunks(shape):
    """Row tiles (128) and column chunks (256) with ragged edges; KL-only mode (m = 0)."""
    ops = _ops()
    g, n, m = shape
    gen = torch.Generator().manual_seed(5)
    z = torch.randn(g, n, 32, generator=gen) * 1.3 - 0.1
    y = torch.randn(g, m, 32, generator=gen)
    kl, mmd, dz = ops.kl_mmd(z.to(DEV), y.to(DEV), w_kl=1.0, w_mmd=0.5)
    zr = z.double().requires_grad_(True)
    klr, klm = R.kl_moment(zr)
    mr = R.mmd_rbf(zr, y.double())
    (klm + 0.5 * mr.mean()).backward()
    assert rel_err(kl[:32].cpu(), klr) < 1e-4 and rel_err(mmd.cpu(), mr) < 1e-4 and rel_err(dz.cpu(), zr.grad) < 1e-4
    kl2, none, dz2 = ops.kl_mmd(z.to(DEV), None, w_kl=1.0)
    zr2 = z.double().requires_grad_(True)
    R.kl_moment(zr2)[1].backward()
    assert none is None and rel_err(kl2, kl) < 1e-6 and rel_err(dz2.cpu(), zr2.grad) < 1e-5      # (KL-only: streaming moment pass; with y: moments folded into the pair kernel)
    kl3, mmd3, dz3 = ops.kl_mmd(z.to(DEV), y.to(DEV), w_kl=1.0, w_mmd=0.5)
    assert torch.equal(dz3, dz) and torch.equal(mmd3, mmd)          # deterministic


@pytest.fixture
def kl_debug_reset():
    """dmvae_debug_kl_mmd forces a path for the whole process: back to the planner's choice whatever the test did."""
    yield
    from dmvae_amd import _lib
    _lib.lib().dmvae_debug_kl_mmd(-1, -1, -1)


def test_kl_mmd_two_launch_path_matches_five_launch_path(kl_debug_reset):
    """At the training step's own shape (32 images x 256 tokens vs 256 prior samples) the op runs as TWO launches (pair kernel with the KL moment partials
    folded in + one finishing kernel); dmvae_debug_kl_mmd(0, -1, -1) selects the five-launch path (streaming moment pass, single-block final, pair kernel, MMD
    final, gradient pass).  Same results to f32 summation order (the two-launch path shares the columns of a row tile out over up to four workgroups at this size, so its
    fixed summation order is a different one), both deterministic, both vs the f64 spec; without the column split the MMD part agrees to the same bar."""
    ops = _ops()
    gen = torch.Generator().manual_seed(3)
    z = (torch.randn(32, 256, 32, generator=gen) * 0.8 + 0.1).to(DEV)
    y = torch.randn(32, 256, 32, generator=gen).to(DEV)
    from dmvae_amd import _lib
    dbg = _lib.lib().dmvae_debug_kl_mmd          # (fused, csplit, mfma), -1 = the planner's choice
    dbg(1, -1, -1)
    kl_a, mmd_a, dz_a = ops.kl_mmd(z, y, w_kl=0.7, w_mmd=1.5)
    kl_a2, mmd_a2, dz_a2 = ops.kl_mmd(z, y, w_kl=0.7, w_mmd=1.5)
    assert torch.equal(kl_a, kl_a2) and torch.equal(mmd_a, mmd_a2) and torch.equal(dz_a, dz_a2)
    dbg(0, -1, -1)
    kl_b, mmd_b, dz_b = ops.kl_mmd(z, y, w_kl=0.7, w_mmd=1.5)
    # MMD^2 = kxx/n^2 + kyy/m^2 - 2 kxy/(n m) cancels ~two digits: 1e-7 on the three f32 sums shows as ~1e-5 on the difference
    assert rel_err(mmd_a, mmd_b) < 5e-5 and rel_err(kl_a, kl_b) < 1e-6 and rel_err(dz_a, dz_b) < 1e-6
    zr = z.cpu().double().requires_grad_(True)
    klr, klm = R.kl_moment(zr)
    mr = R.mmd_rbf(zr, y.cpu().double())
    (0.7 * klm + 1.5 * mr.mean()).backward()
    assert rel_err(kl_a[:32].cpu(), klr) < 1e-4 and rel_err(mmd_a.cpu(), mr) < 1e-4 and rel_err(dz_a.cpu(), zr.grad) < 1e-4
    _, mmd_v, none = ops.kl_mmd(z, y, need_grad=False)                     # value-only call on the two-launch path
    dbg(1, -1, -1)
    kl_v, mmd_v2, none2 = ops.kl_mmd(z, y, need_grad=False)
    assert none is None and none2 is None and rel_err(mmd_v, mmd_v2) < 5e-5 and rel_err(kl_v, kl_a) < 1e-6
    try:
        dbg(1, 1, -1)                                                     # no column split: same per-tile order as the five-launch path
        _, mmd_c, dz_c = ops.kl_mmd(z, y, w_kl=0.7, w_mmd=1.5)            # (pair sums on the matrix cores: another fixed order, same bar as above)
        assert rel_err(mmd_c, mmd_b) < 5e-5 and rel_err(dz_c, dz_b) < 1e-6
        assert rel_err(mmd_c.cpu(), mr) < 1e-4 and rel_err(dz_c.cpu(), zr.grad) < 1e-4
        dbg(1, -1, 0)                                                     # the scalar-FMA pair kernel instead of the matrix-core one
        _, mmd_s, dz_s = ops.kl_mmd(z, y, w_kl=0.7, w_mmd=1.5)
        assert rel_err(mmd_s, mmd_b) < 5e-5 and rel_err(dz_s, dz_b) < 1e-6
    finally:
        dbg(-1, -1, -1)


def test_adamw_ema_golden():
    ops = _ops()
    g = load_golden("opt_tail")
    names = ["0.weight", "0.bias", "2.weight", "2.bias"]
    sizes = [g["p0." + n].size for n in names]
    flat = torch.cat([g.t("p0." + n).flatten() for n in names]).to(DEV)
    ema, m, v = flat.clone(), torch.zeros_like(flat), torch.zeros_like(flat)
    flat0 = flat.clone()
    for it in range(6):
        gr = torch.cat([g.t(f"g{it}.{i}").flatten() for i in range(4)]).to(DEV)
        norm = ops.grad_norm(gr, 1.0)
        assert abs(norm[0].item() - float(g["norms"][it])) < 1e-5 * float(g["norms"][it])
        ops.adamw_ema_step(flat, gr, m, v, ema, norm, R.warmup_lr(it, 1e-4, int(g["warmup_steps"])), 0.9, 0.95, 1e-8, 0.005, it + 1, 0.9999)
        # the shadow-writing variant: identical f32 state, plus bf16(new weight) -- autocast's per-forward `weight.to(bfloat16)` made once per step
        if it == 0:
            flat2, ema2, m2, v2, sh = flat0.clone(), flat0.clone(), torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros(flat.numel(), dtype=BF, device=DEV)
        ops.adamw_ema_step(flat2, gr, m2, v2, ema2, norm, R.warmup_lr(it, 1e-4, int(g["warmup_steps"])), 0.9, 0.95, 1e-8, 0.005, it + 1, 0.9999, shadow=sh)
        assert torch.equal(flat2, flat) and torch.equal(ema2, ema) and torch.equal(m2, m) and torch.equal(v2, v)
        assert torch.equal(sh, flat.to(BF))
    off = 0
    for n, sz in zip(names, sizes):
        assert rel_err(flat[off:off + sz].cpu(), g.t("p6." + n).flatten()) < 1e-6
        assert rel_err(ema[off:off + sz].cpu(), g.t("ema6." + n).flatten()) < 1e-6
        off += sz


def test_silu():
    ops = _ops()
    x = torch.randn(4, 64, device=DEV).to(BF)
    dy = torch.randn(4, 64, device=DEV).to(BF)
    xr = x.float().cpu().double().requires_grad_(True)
    F.silu(xr).backward(dy.float().cpu().double())
    assert rel_err(ops.silu(x).float().cpu(), F.silu(xr.detach())) < 4e-3
    assert rel_err(ops.silu_bwd(x, dy).float().cpu(), xr.grad) < 4e-3
