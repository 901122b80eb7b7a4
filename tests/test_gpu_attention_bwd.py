"""csrc/attention_bwd.hip: backward of the fused multi-head self-attention (timm Attention, dino_layers/attention.py:56-69; LightningDiT Attention,
lightningdit.py:76-88) against fp64 autograd on the same bf16 operands -- both layouts (qkv-interleaved head dim 64; head-major with head dim 64 and
72 padded to 96), ragged token counts around the 32-token blocks, the 257 / 256 tokens of the two models.  The kernel rounds P and dS to bf16 where they
enter the matrix cores (as the GEMM-composed backward it replaces did, and as autocast SDPA does), so the bar is the bf16 operand floor; the composed
route is held to the same reference next to it."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _ref(q, k, v, do, scale):
    """q, k, v, do: [B, H, S, D] fp64 -> o, dq, dk, dv"""
    q, k, v = (t.clone().requires_grad_(True) for t in (q, k, v))
    p = torch.softmax(scale * q @ k.transpose(-1, -2), dim=-1)
    o = p @ v
    o.backward(do)
    return o.detach(), q.grad, k.grad, v.grad


def _rl2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("b,s,h", [(2, 257, 4), (1, 288, 2), (3, 64, 2), (2, 33, 1), (1, 1, 2), (1, 200, 16)])
def test_attention_bwd_qkv_layout(b, s, h, monkeypatch):
    from dmvae_amd import ops, functional as Fn
    d, c = 64, h * 64
    g = torch.Generator().manual_seed(s + h)
    qkv = (torch.randn(b, s, 3, h, d, generator=g) * 1.5).to(BF)
    do = torch.randn(b, s, c, generator=g).to(BF)
    scale = d ** -0.5
    q, k, v = (qkv[:, :, i].double().permute(0, 2, 1, 3) for i in range(3))
    o_r, dq_r, dk_r, dv_r = _ref(q, k, v, do.double().view(b, s, h, d).permute(0, 2, 1, 3), scale)
    want = torch.stack([dq_r, dk_r, dv_r], 0).permute(1, 3, 0, 2, 4).reshape(b, s, 3 * c)      # [3,B,H,S,D] -> [B,S,3,H,D]
    qkv_g, do_g = qkv.view(b, s, 3 * c).to(DEV), do.to(DEV)
    o = ops.attention_qkv(qkv_g, h, scale)
    assert _rl2(o.cpu(), o_r.permute(0, 2, 1, 3).reshape(b, s, c)) < 6e-3
    dqkv = ops.attention_bwd_qkv(qkv_g, o, do_g, h, scale)
    assert dqkv.shape == (b, s, 3 * c) and torch.isfinite(dqkv.float()).all()
    for i, name in enumerate("qkv"):
        got = dqkv.view(b, s, 3, c)[:, :, i].float().cpu()
        ref = want.view(b, s, 3, c)[:, :, i]
        if s == 1 and name != "v":         # a single key: the softmax is constant, dq = dk = 0 exactly; the kernel leaves rounding noise of P - 1
            assert got.abs().max() < 1e-5
            continue
        assert _rl2(got, ref) < 1.2e-2, (name, _rl2(got, ref))
        assert rel_err(got, ref) < 3e-2, name
    assert torch.equal(dqkv, ops.attention_bwd_qkv(qkv_g, o, do_g, h, scale))                    # fixed summation order
    # the eight-wave form on the forward's row statistics: lse against fp64, the same bars on the gradients, rerun-identical, and the same output bits as the plain forward
    o2, lse = ops.attention_qkv(qkv_g, h, scale, need_lse=True)
    assert torch.equal(o2, o)
    lse_r = torch.logsumexp(scale * q @ k.transpose(-1, -2), dim=-1).reshape(b * h, s)
    assert (lse.double().cpu() - lse_r).abs().max() < 2e-3
    dqkv2 = ops.attention_bwd_qkv(qkv_g, o, do_g, h, scale, lse=lse)
    assert torch.isfinite(dqkv2.float()).all() and torch.equal(dqkv2, ops.attention_bwd_qkv(qkv_g, o, do_g, h, scale, lse=lse))
    for i, name in enumerate("qkv"):
        got = dqkv2.view(b, s, 3, c)[:, :, i].float().cpu()
        ref = want.view(b, s, 3, c)[:, :, i]
        if s == 1 and name != "v":
            assert got.abs().max() < 1e-5
            continue
        assert _rl2(got, ref) < 1.2e-2 and rel_err(got, ref) < 3e-2, ("lse", name, _rl2(got, ref))
    comp = Fn._attention_bwd(qkv_g, do_g, h, scale)                                               # the GEMM-composed route: same bar, and close to the fused one
    if s > 1:
        assert _rl2(comp.float().cpu(), want) < 1.2e-2 and _rl2(dqkv.float(), comp.float()) < 1.2e-2


@pytest.mark.parametrize("b,n,h,d", [(2, 256, 4, 72), (2, 256, 3, 64), (1, 100, 2, 72), (1, 288, 1, 40)])
def test_attention_bwd_heads_layout(b, n, h, d):
    from dmvae_amd import ops
    dp = (d + 31) // 32 * 32
    g = torch.Generator().manual_seed(n + d)
    q = torch.zeros(b * h, n, dp, dtype=BF); k = torch.zeros(b * h, n, dp, dtype=BF)
    q[..., :d] = (torch.randn(b * h, n, d, generator=g)).to(BF)
    k[..., :d] = (torch.randn(b * h, n, d, generator=g)).to(BF)
    v = torch.randn(b * h, n, d, generator=g).to(BF)
    do = torch.randn(b, n, h * d, generator=g).to(BF)
    scale = d ** -0.5
    f = lambda t: t[..., :d].double().view(b, h, n, d)
    o_r, dq_r, dk_r, dv_r = _ref(f(q), f(k), f(v), do.double().view(b, n, h, d).permute(0, 2, 1, 3), scale)
    qg, kg, vg, dog = q.to(DEV), k.to(DEV), v.to(DEV), do.to(DEV)
    o = ops.attention_heads(qg, kg, vg, b, scale)
    assert _rl2(o.cpu(), o_r.permute(0, 2, 1, 3).reshape(b, n, h * d)) < 6e-3
    dq, dk, dv = ops.attention_bwd_heads(qg, kg, vg, o, dog, b, scale)
    assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == v.shape
    o2, lse = ops.attention_heads(qg, kg, vg, b, scale, need_lse=True)
    assert torch.equal(o2, o)
    assert (lse.double().cpu() - torch.logsumexp(scale * f(q) @ f(k).transpose(-1, -2), dim=-1).reshape(b * h, n)).abs().max() < 2e-3
    dq2, dk2, dv2 = ops.attention_bwd_heads(qg, kg, vg, o, dog, b, scale, lse=lse)
    for got, ref, name in ((dq, dq_r, "dq"), (dk, dk_r, "dk"), (dv, dv_r, "dv"), (dq2, dq_r, "dq (lse)"), (dk2, dk_r, "dk (lse)"), (dv2, dv_r, "dv (lse)")):
        gf = got.float().cpu()
        assert _rl2(gf[..., :d], ref.reshape(b * h, n, d)) < 1.2e-2, name
        if gf.shape[-1] > d:
            assert gf[..., d:].abs().max() == 0, name + ": padded channels"
    again = ops.attention_bwd_heads(qg, kg, vg, o, dog, b, scale, lse=lse)
    assert all(torch.equal(x, y) for x, y in zip((dq2, dk2, dv2), again))


@pytest.mark.parametrize("n,d", [(256, 72), (200, 72), (272, 72), (256, 64)])      # 272 tokens: nine query blocks for eight waves (a second round inside an item)
def test_attention_kernels_give_the_same_bits_in_every_launch_form(n, d):
    """The launch forms chosen by the NUMBER of (sample, head) blocks -- the three-image backward at <= 512 blocks, the two-image one above; the XCD-aware block order
    at any grid -- compute the same sums in the same order: samples 0..7 of a 40-sample call (640 blocks) must equal the 8-sample call (128 blocks) bit for bit,
    forward output, row statistics and all three gradients.  LightningDiT-XL/1's head geometry (16 x 72, padded to 96) and a 64-wide one."""
    from dmvae_amd import ops
    h, dp = 16, (d + 31) // 32 * 32
    g = torch.Generator().manual_seed(n + d)
    big, small = 40, 8
    q = torch.zeros(big * h, n, dp); k = torch.zeros(big * h, n, dp)
    q[..., :d] = torch.randn(big * h, n, d, generator=g); k[..., :d] = torch.randn(big * h, n, d, generator=g)
    v = torch.randn(big * h, n, d, generator=g)
    q, k, v = (t.to(DEV).to(BF) for t in (q, k, v))
    do = torch.randn(big, n, h * d, generator=g).to(DEV).to(BF)
    scale = d ** -0.5
    outs = []
    for b in (big, small):
        qq, kk, vv, dd = q[:b * h].contiguous(), k[:b * h].contiguous(), v[:b * h].contiguous(), do[:b].contiguous()
        o, lse = ops.attention_heads(qq, kk, vv, b, scale, need_lse=True)
        dq, dk, dv = ops.attention_bwd_heads(qq, kk, vv, o, dd, b, scale, lse=lse)
        outs.append((o, lse, dq, dk, dv))
    for a, c, rows in zip(outs[0], outs[1], (small, small * h, small * h, small * h, small * h)):
        assert torch.equal(a[:rows], c), "launch form changed the bits"
    # every sample of the large call, forward: at 640 blocks the 96-wide forward is the persistent form (one workgroup per CU walking two or three (sample, head)
    # items, the next item's K / V loaded under the current one's sweeps -- csrc/vit.hip, PIPE); 128-block calls are the one-item-per-workgroup form
    o_big, lse_big = outs[0][0], outs[0][1]
    for s0 in range(0, big, small):
        sl = slice(s0 * h, (s0 + small) * h)
        o_c, lse_c = ops.attention_heads(q[sl].contiguous(), k[sl].contiguous(), v[sl].contiguous(), small, scale, need_lse=True)
        assert torch.equal(o_big[s0:s0 + small], o_c) and torch.equal(lse_big[sl], lse_c), f"samples {s0}..{s0 + small - 1}"


@pytest.mark.parametrize("big,n", [(8, 256), (40, 256), (40, 200), (40, 272)])
def test_attention_heads_with_unpadded_q_k_rows_gives_the_same_bits(big, n):
    """q / k rows of D = 72 channels (144 B apart; `ops.qknorm_rope(padded=False)`, what `DitStackFn` hands the kernels) instead of rows zero-padded to 96: the kernels
    neither load nor store the chunks past D, and every product sees the same zeros -- forward output, row statistics and all three gradients are bit-identical to
    the padded call, in every launch form (128 blocks: three-image backward; 640 blocks: the pipelined forward and the two-workgroup backward, or the wide 8-wave
    backward at 272 tokens)."""
    from dmvae_amd import ops
    h, d, dp = 16, 72, 96
    g = torch.Generator().manual_seed(big + n)
    q = torch.zeros(big * h, n, dp); k = torch.zeros(big * h, n, dp)
    q[..., :d] = torch.randn(big * h, n, d, generator=g); k[..., :d] = torch.randn(big * h, n, d, generator=g)
    v = torch.randn(big * h, n, d, generator=g)
    q, k, v = (t.to(DEV).to(BF) for t in (q, k, v))
    do = torch.randn(big, n, h * d, generator=g).to(DEV).to(BF)
    qu, ku = q[..., :d].contiguous(), k[..., :d].contiguous()
    scale = d ** -0.5
    o, lse = ops.attention_heads(q, k, v, big, scale, need_lse=True)
    ou, lseu = ops.attention_heads(qu, ku, v, big, scale, need_lse=True)
    assert torch.equal(o, ou) and torch.equal(lse, lseu)
    dq, dk, dv = ops.attention_bwd_heads(q, k, v, o, do, big, scale, lse=lse)
    dqu, dku, dvu = ops.attention_bwd_heads(qu, ku, v, o, do, big, scale, lse=lse)
    assert dqu.shape[-1] == d and torch.equal(dq[..., :d], dqu) and torch.equal(dk[..., :d], dku) and torch.equal(dv, dvu)
    assert dq[..., d:].abs().max() == 0 and dk[..., d:].abs().max() == 0
    if big == 8:      # the kernel without row statistics (first form), too
        dq0, dk0, dv0 = ops.attention_bwd_heads(qu, ku, v, o, do, big, scale)
        dq1, dk1, dv1 = ops.attention_bwd_heads(q, k, v, o, do, big, scale)
        assert torch.equal(dq1[..., :d], dq0) and torch.equal(dk1[..., :d], dk0) and torch.equal(dv1, dv0)
