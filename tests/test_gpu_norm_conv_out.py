"""csrc/groupnorm.hip::convout_bwd_kernel (-m gpu): the backward of the decoder's tail conv_out(swish(norm_out(x))) (models/flux_ae.py:266-268) with the 3 -> C
input-gradient conv evaluated inside both GroupNorm backward passes -- against the stored-operand route (input-gradient conv -> dmvae_groupnorm_bwd: the same
arithmetic up to the summation order inside one bf16 rounding of the intermediate), against f64 autograd of the oracle's group_norm / swish / conv on the same
operands, at image borders, reruns bit-identical, and through `NormConvOutFn` with the switch on and off."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16

SHAPES = [(2, 32, 48), (3, 16, 16), (1, 64, 32), (2, 8, 112)]      # (n, h, w): w % 16 == 0; 16-pixel runs touch both borders at w = 16


def _case(n, h, w, c=128, seed=0, dy_scale=1.0):
    g = torch.Generator().manual_seed(100 * n + h + w + seed)
    x = (torch.randn(n, h, w, c, generator=g) * 1.5 + 0.3).to(DEV).to(BF)
    gamma = (1.0 + 0.2 * torch.randn(c, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(c, generator=g)).to(DEV)
    cw = (torch.randn(3, c, 3, 3, generator=g) * 0.05).to(DEV)
    dy = (torch.randn(n, 3, h, w, generator=g) * dy_scale).to(DEV)
    return x, gamma, beta, cw, dy


def _stored_route(x, gamma, beta, cw, dy, st):
    from dmvae_amd import ops
    from dmvae_amd.functional import packed
    dyp = ops.nchw_to_nhwc_bf16(dy, c_pad=32)
    da = ops.conv2d_nhwc(dyp, packed(cw, True, cols_pad=32), ks=3)
    return ops.groupnorm_bwd(da, x, st, gamma, beta, True), da


@pytest.mark.parametrize("n,h,w", SHAPES)
def test_fused_tail_backward_equals_the_stored_operand_route(n, h, w):
    from dmvae_amd import ops
    assert ops.norm_conv_out_bwd_supported(n, h, w, 128, 3)
    x, gamma, beta, cw, dy = _case(n, h, w)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    (dx0, dg0, db0), da = _stored_route(x, gamma, beta, cw, dy, st)
    dx, dg, db = ops.norm_conv_out_bwd(dy, cw, x, st, gamma, beta)
    assert dx.dtype == BF and dx.shape == x.shape
    # the only difference is the order in which the 27 products of one `da` element are added in f32 before its bf16 rounding: a few elements of da land on
    # the other side of a rounding boundary (one bf16 spacing), which moves dx there by a comparable amount and the group sums by next to nothing
    d = (dx.float() - dx0.float()).abs()
    scale = dx0.float().abs().max().item()
    assert d.max().item() <= 2.0 ** -6 * scale, (d.max().item(), scale)
    assert (d > 0).float().mean().item() < 0.05
    assert rel_err(dx.float(), dx0.float()) < 2e-3
    assert rel_err(dg, dg0) < 1e-3 and rel_err(db, db0) < 1e-3
    for _ in range(2):
        dx2, dg2, db2 = ops.norm_conv_out_bwd(dy, cw, x, st, gamma, beta)
        assert torch.equal(dx2, dx) and torch.equal(dg2, dg) and torch.equal(db2, db)


@pytest.mark.parametrize("n,h,w", SHAPES[:3])
def test_fused_tail_backward_vs_f64_autograd_of_the_oracle(n, h, w):
    """f64 autograd through oracle.ref_cpu's group_norm -> swish -> conv on the HIP path's operands (bf16 x, bf16-rounded dy and conv weight): what is left is
    the bf16 rounding of the activation gradient (2^-9 relative per element) and of dx itself."""
    from dmvae_amd import ops
    from oracle import ref_cpu
    x, gamma, beta, cw, dy = _case(n, h, w, seed=5)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    dx, dg, db = ops.norm_conv_out_bwd(dy, cw, x, st, gamma, beta)
    xd = x.double().permute(0, 3, 1, 2).cpu().requires_grad_(True)
    gd, bd = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
    a = ref_cpu.swish(ref_cpu.group_norm(xd, gd, bd, 32, 1e-6))
    y = F.conv2d(a, cw.to(BF).double().cpu(), padding=1)
    (y * dy.to(BF).double().cpu()).sum().backward()
    ref_dx = xd.grad.permute(0, 2, 3, 1)
    assert rel_err(dx.double().cpu(), ref_dx) < 6e-3
    assert rel_err(dg.double().cpu(), gd.grad) < 5e-3 and rel_err(db.double().cpu(), bd.grad) < 5e-3
    # and it is no further from f64 than the stored-operand route is
    (dx0, dg0, db0), _ = _stored_route(x, gamma, beta, cw, dy, st)
    assert rel_err(dx.double().cpu(), ref_dx) <= 1.1 * rel_err(dx0.double().cpu(), ref_dx) + 1e-6
    assert rel_err(dg.double().cpu(), gd.grad) <= 1.1 * rel_err(dg0.double().cpu(), gd.grad) + 1e-6


def test_fused_tail_backward_borders_and_accumulate():
    """A gradient that is non-zero on the image border only: every contribution crosses the zero padding of the bordered copy; and parameter gradients written
    into caller-owned buffers."""
    from dmvae_amd import ops
    n, h, w = 2, 16, 16
    x, gamma, beta, cw, dy = _case(n, h, w, seed=11)
    m = torch.zeros_like(dy)
    m[:, :, 0, :] = 1; m[:, :, -1, :] = 1; m[:, :, :, 0] = 1; m[:, :, :, -1] = 1
    dy = dy * m
    st = ops.groupnorm_stats(x, 32, 1e-6)
    (dx0, dg0, db0), _ = _stored_route(x, gamma, beta, cw, dy, st)
    dgo, dbo = torch.full((128,), 7.0, device=DEV), torch.full((128,), -3.0, device=DEV)
    dx, dg, db = ops.norm_conv_out_bwd(dy, cw, x, st, gamma, beta, dg_out=dgo, db_out=dbo)
    assert dg.data_ptr() == dgo.data_ptr() and db.data_ptr() == dbo.data_ptr()
    assert rel_err(dx.float(), dx0.float()) < 2e-3 and rel_err(dg, dg0) < 1e-3 and rel_err(db, db0) < 1e-3


def test_unsupported_shapes_are_refused():
    from dmvae_amd import ops
    assert not ops.norm_conv_out_bwd_supported(2, 16, 24, 128, 3)      # w % 16
    assert not ops.norm_conv_out_bwd_supported(2, 16, 16, 256, 3)      # c
    assert not ops.norm_conv_out_bwd_supported(2, 16, 16, 128, 4)      # cout
    x, gamma, beta, cw, dy = _case(1, 16, 16)
    st = ops.groupnorm_stats(x, 32, 1e-6)
    with pytest.raises((ValueError, AssertionError)):
        ops.norm_conv_out_bwd(dy[:, :, :, :8].contiguous(), cw, x[:, :, :8].contiguous(), st, gamma, beta)


def test_norm_conv_out_fn_backward_with_and_without_the_fused_route(monkeypatch):
    """NormConvOutFn (the decoder's tail as the model runs it): forward identical, backward equal to the stored-operand route within the bars above; the weight
    and bias gradients of conv_out do not depend on the switch at all."""
    from dmvae_amd import functional as Fn
    n, h, w = 2, 32, 32
    x, gamma, beta, cw, dy = _case(n, h, w, seed=3)
    cb = torch.randn(3, device=DEV) * 0.1
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(Fn, "NORM_CONV_OUT_FUSED_BWD", fused)
        leaves = [t.clone().requires_grad_(True) for t in (x.float(), gamma, beta, cw, cb)]
        y = Fn.NormConvOutFn.apply(leaves[0].to(BF), *leaves[1:])
        y.backward(dy)
        outs.append((y.detach(), [t.grad for t in leaves]))
    (y1, g1), (y0, g0) = outs
    assert torch.equal(y1, y0)
    assert rel_err(g1[0], g0[0]) < 2e-3 and rel_err(g1[1], g0[1]) < 1e-3 and rel_err(g1[2], g0[2]) < 1e-3
    assert torch.equal(g1[3], g0[3]) and torch.equal(g1[4], g0[4])


def test_tail_node_asked_for_the_last_layers_weight_gradient_alone(monkeypatch):
    """functional.tail_weight_only (losses.generator_gan_backward's two adaptive-weight norms, train_tokenizer.py:190-203): inside the context NormConvOutFn's
    backward returns conv_out's weight gradient -- the bits of the full backward's -- and launches neither the input-gradient / norm backward nor the bias sum."""
    from dmvae_amd import functional as Fn, ops
    n, h, w = 2, 32, 64
    x, gamma, beta, cw, dy = _case(n, h, w, seed=5)
    if not ops.conv_out_wgrad_supported(n, h, w, x.shape[-1], 3):
        pytest.skip("shape not on the thin weight-gradient kernel")
    cb = torch.randn(3, device=DEV) * 0.1
    leaves = [t.clone().requires_grad_(True) for t in (x.float(), gamma, beta, cw, cb)]
    y = Fn.NormConvOutFn.apply(leaves[0].to(BF), *leaves[1:])
    full = torch.autograd.grad(y, leaves[3], grad_outputs=dy, retain_graph=True)[0].clone()

    def boom(*a, **k):
        raise AssertionError("the input gradient was computed for a weight-gradient-only request")
    monkeypatch.setattr(ops, "norm_conv_out_bwd", boom)
    monkeypatch.setattr(ops, "groupnorm_bwd", boom)
    with Fn.tail_weight_only():
        only = torch.autograd.grad(y, leaves[3], grad_outputs=dy, retain_graph=True)[0]
    assert torch.equal(only, full)
    assert not Fn._TAIL_WEIGHT_ONLY[0]


@pytest.mark.parametrize("n,h,w", [(2, 32, 64), (3, 8, 32), (1, 64, 32)])
def test_fused_tail_forward_gives_the_bits_of_the_five_launch_route(n, h, w, monkeypatch):
    """csrc/conv_thin.hip, NORM instantiation: GroupNorm + swish on the way into the conv's halo tile, the NCHW image out of its epilogue -- the saved activation
    and the image must be the bits of groupnorm_apply -> conv2d_nhwc (f32 result) -> nhwc_to_nchw (same arithmetic, same summation order), with zero padding
    applied after the activation (a beta that makes swish(GroupNorm(0)) non-zero would show a mistake there)."""
    from dmvae_amd import functional as Fn, ops
    assert ops.norm_conv_out_fwd_supported(n, h, w, 128, 3)
    x, gamma, beta, cw, _ = _case(n, h, w, seed=21)
    beta = beta + 0.7
    cb = torch.tensor([0.3, -0.2, 0.1], device=DEV)
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(Fn, "NORM_CONV_OUT_FUSED_FWD", fused)

        class Ctx:
            def save_for_backward(self, *t):
                self.saved = t
        ctx = Ctx()
        y = Fn.NormConvOutFn.forward(ctx, x, gamma, beta, cw, cb)
        outs.append((y, ctx.saved[2], ctx.saved[1]))
    (y1, a1, st1), (y0, a0, st0) = outs
    assert y1.shape == (n, 3, h, w) and y1.dtype == torch.float32
    assert torch.equal(st1, st0) and torch.equal(a1, a0) and torch.equal(y1, y0)
    # and the module-level function end to end (forward + backward) with both switches on
    monkeypatch.setattr(Fn, "NORM_CONV_OUT_FUSED_FWD", True)
    leaves = [t.clone().requires_grad_(True) for t in (x.float(), gamma, beta, cw, cb)]
    yy = Fn.NormConvOutFn.apply(leaves[0].to(BF), *leaves[1:])
    assert torch.equal(yy.detach(), y0)
    yy.backward(torch.ones_like(yy))
    assert all(t.grad is not None and torch.isfinite(t.grad).all() for t in leaves)
