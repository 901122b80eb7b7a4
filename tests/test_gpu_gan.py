"""Discriminator branch on the HIP kernels (-m gpu): the drop-in NLayerDiscriminator (im2col + MFMA GEMM convs, BatchNorm on the
GroupNorm kernels), DiffAug and the GAN loss terms, against the fixtures captured from the reference and the CPU oracle evaluated
with bf16 rounding at the HIP path's storage points.  Tolerances as in test_gpu_modules.py."""
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from test_oracle_gan import patchgan_params

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16
Q = R.bf16_round
TOL_Q, TOL_REF = 1e-2, 3e-2


def _disc(params):
    from dmvae_amd.models.patchgan import NLayerDiscriminator
    d = NLayerDiscriminator()
    sd = d.state_dict()
    for k in sd:
        if k in params:
            sd[k] = params[k].clone()
    d.load_state_dict(sd, strict=True)
    return d.to(DEV)


@pytest.mark.parametrize("case", [(2, 8, 8, 16, 64, 4, 2, 1), (1, 9, 11, 8, 32, 4, 1, 1), (3, 16, 16, 8, 64, 4, 2, 1), (1, 7, 7, 24, 8, 3, 1, 0),
                                  (2, 64, 64, 64, 128, 4, 2, 1)])
def test_im2col_col2im(case):
    """im2col against F.unfold (bit-exact: pure data movement) and col2im against its autograd adjoint (f32 sum of <= 16 bf16 values,
    one rounding)."""
    from dmvae_amd import ops
    n, h, w, c, _, ks, stride, pad = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, h, w, c, generator=g).to(DEV).to(BF)
    col = ops.im2col(x, ks, stride, pad)
    xr = x.float().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    u = F.unfold(xr, ks, padding=pad, stride=stride)                                  # [n, c*ks*ks, L], channel-major rows
    ho, wo = col.shape[1], col.shape[2]
    ref = u.view(n, c, ks * ks, ho, wo).permute(0, 3, 4, 2, 1).reshape(n, ho, wo, ks * ks * c)
    assert torch.equal(col.float().cpu(), ref.detach())
    dcol = torch.randn(col.shape, generator=g).to(DEV).to(BF)
    dx = ops.col2im(dcol, h, w, ks, stride, pad)
    ref.backward(dcol.float().cpu(), retain_graph=True)
    assert rel_err(dx.float().cpu(), xr.grad.permute(0, 2, 3, 1)) < 4e-3              # one bf16 rounding of the f32 sum
    dcol32 = torch.randn(col.shape, generator=g).to(DEV)                               # f32 input: the GEMM's unrounded result
    xr.grad = None
    ref.backward(dcol32.cpu())
    assert rel_err(ops.col2im(dcol32, h, w, ks, stride, pad).float().cpu(), xr.grad.permute(0, 2, 3, 1)) < 4e-3


def test_patchgan_eval_and_train():
    g = load_golden("patchgan_small")
    p = patchgan_params(g, int(g["seed"]))
    disc = _disc(p)
    assert [k for k in disc.state_dict()] == [str(k) for k in g["keys"]]
    x = g.t("x").to(DEV)
    disc.eval()
    with torch.no_grad():
        y_eval = disc(x)
    assert y_eval.shape == (2, 1, 6, 6) and y_eval.dtype == torch.float32
    with torch.no_grad():
        yo_eval, _ = R.patchgan_forward(g.t("x"), p, q=Q, training=False)
    assert rel_err(y_eval.cpu(), yo_eval) < TOL_Q
    assert rel_err(y_eval.cpu(), g.t("y_eval")) < TOL_REF
    # training step: batch statistics, running-estimate update, all gradients
    disc.train()
    xg = x.clone().requires_grad_(True)
    y = disc(xg)
    y.backward(g.t("dy").to(DEV))
    po = {k: (v.clone().requires_grad_(True) if "running" not in k else v.clone()) for k, v in p.items()}
    xo = g.t("x").requires_grad_(True)
    yo, buf = R.patchgan_forward(xo, po, q=Q, training=True)
    yo.backward(Q(g.t("dy")))                    # the logit gradient enters the HIP path as a bf16 GEMM operand
    assert rel_err(y.cpu(), yo.detach()) < TOL_Q
    assert rel_err(y.detach().cpu(), g.t("y")) < TOL_REF
    assert rel_err(xg.grad.cpu(), xo.grad) < 3 * TOL_Q          # 5 convs / 3 BatchNorms deep (cf. test_flux_encoder_small_fwd_bwd)
    # against the reference's f32 result: 5 convs / 3 small-batch BatchNorms / LeakyReLU kinks amplify bf16 rounding (both for this
    # path and for the bf16-site oracle), so the criterion is test_gpu_modules.py::test_decoder_small_fwd_bwd's -- the HIP path must be
    # as close to the f32 reference as the bf16-site oracle is
    def floor(hip, orc, ref, what, slack=1.5, abs_floor=1e-3):
        rl2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
        e_hip, e_orc = rl2(hip, ref), rl2(orc, ref)
        print(f"patchgan {what}: rel-L2 to f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
        assert e_hip < slack * e_orc + abs_floor, what
    floor(xg.grad.cpu(), xo.grad, g.t("dx"), "dx")
    sd = disc.state_dict()
    for k, v in buf.items():
        assert rel_err(sd[k].cpu(), g.t("buf1." + k)) < 5e-3, k
    assert int(sd["main.3.num_batches_tracked"]) == int(g["buf1.main.3.num_batches_tracked"]) == 1
    for n, prm in disc.named_parameters():
        ref_norm = g["gn." + n][0]
        if ref_norm < 1e-3:                                   # main.2/5/8.bias: exact zeros in f32, bf16 noise here
            assert prm.grad.abs().max() < 2e-2, n
            continue
        # rel-L2, not max-abs: a pre-activation within bf16 noise of the LeakyReLU kink takes slope 1 on one side and 0.2 on the other,
        # which moves single gradient entries by ~10 % of the maximum while the tensor as a whole agrees to 1.5 % (tools/probes/dbg_patchgan.py)
        assert ((prm.grad.cpu().double() - po[n].grad.double()).norm() / po[n].grad.double().norm()).item() < 3 * TOL_Q, n
        assert abs(prm.grad.double().norm().item() - ref_norm) < 3e-2 * ref_norm, n
        if "g." + n in g:
            floor(prm.grad.cpu(), po[n].grad, g.t("g." + n), "grad " + n)


def test_patchgan_full_size_shapes_and_determinism():
    """256x256 inputs, batch 8 (the discriminator step sees [images; recon]): logits [B,1,30,30]; two identical steps give identical
    logits and gradients (split-K reductions are ordered)."""
    from dmvae_amd.models.patchgan import NLayerDiscriminator, weights_init
    torch.manual_seed(0)
    disc = NLayerDiscriminator().apply(weights_init).to(DEV)
    x = (torch.rand(8, 3, 256, 256, generator=torch.Generator().manual_seed(1)) * 2 - 1).to(DEV)
    outs = []
    for _ in range(2):
        disc.zero_grad(set_to_none=True)
        xg = x.clone().requires_grad_(True)
        y = disc(xg)
        assert y.shape == (8, 1, 30, 30)
        F.relu(1.0 - y).mean().backward()
        outs.append((y.detach().clone(), xg.grad.clone(), disc.main[0].weight.grad.clone(), disc.main[8].weight.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.isfinite(outs[0][1]).all() and outs[0][1].abs().max() > 0


@pytest.mark.parametrize("tag", ["a", "b"])
def test_diffaug_kernels(tag):
    """Fused DiffAug against the fixtures captured from the reference's DiffAug.aug (translation / cut-out bit-exact, colour to f32
    summation order) and against its captured input gradient."""
    from dmvae_amd import ops
    g = load_golden("diffaug")
    x = g.t(f"{tag}.x").to(DEV)
    for name, flags in (("trans", 1), ("color", 2), ("cut", 4), ("all", 7)):
        y = ops.diffaug(x, g.t(f"{tag}.{name}.rand01").to(DEV), flags, 0.2)
        ref = g.t(f"{tag}.{name}.y")
        if name in ("trans", "cut"):
            assert torch.equal(y.cpu(), ref), name
        else:
            assert rel_err(y.cpu(), ref) < 2e-6, name
            assert torch.equal(y.cpu() == 0, ref == 0) or name == "color"
    dx = ops.diffaug_bwd(g.t(f"{tag}.all.dy").to(DEV), g.t(f"{tag}.all.rand01").to(DEV), 7, 0.2)
    assert rel_err(dx.cpu(), g.t(f"{tag}.all.dx")) < 1e-5
    assert torch.equal(dx.cpu() == 0, g.t(f"{tag}.all.dx") == 0)
    if tag == "a":
        y = ops.diffaug(x, g.t("a.edge.rand01").to(DEV), 7, 0.2)
        assert rel_err(y.cpu(), g.t("a.edge.y")) < 2e-6 and torch.equal(y.cpu() == 0, g.t("a.edge.y") == 0)


def test_diffaug_module_rng_and_adjoint():
    """The drop-in DiffAug consumes the RNG streams like the reference (torch.rand(3) on the CPU generator, torch.rand(7,B,1,1) on the
    device generator) and its backward is the exact adjoint of its forward: <aug(x), dy> == <x, aug^T(dy)> up to the constant offset the
    brightness term adds (checked through a difference of two inputs)."""
    from dmvae_amd.utils.diffaug import DiffAug
    x1 = (torch.rand(6, 3, 64, 48, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(DEV)
    x2 = (torch.rand(6, 3, 64, 48, generator=torch.Generator().manual_seed(6)) * 2 - 1).to(DEV)
    dy = torch.randn(6, 3, 64, 48, generator=torch.Generator().manual_seed(7)).to(DEV)
    outs = []
    for x in (x1, x2):
        torch.manual_seed(123)
        xg = x.clone().requires_grad_(True)
        y = DiffAug(prob=1.0, cutout=0.3).aug(xg)
        y.backward(dy)
        outs.append((y.detach(), xg.grad))
    assert torch.equal(outs[0][1], outs[1][1])                       # the map is affine: the adjoint does not depend on x
    lhs = ((outs[0][0] - outs[1][0]).double() * dy.double()).sum()
    rhs = ((x1 - x2).double() * outs[0][1].double()).sum()
    assert abs(lhs.item() - rhs.item()) < 1e-6 * max(1.0, abs(lhs.item()))
    torch.manual_seed(123)
    cpu_flags = torch.rand(3)                                         # same draws as the module made
    dev_draws = torch.rand(7, 6, 1, 1, device=DEV)
    y_ref = R.diffaug(x1.cpu(), dev_draws.view(7, 6).cpu(), cutout=0.3)
    assert bool((cpu_flags <= 1.0).all())
    assert rel_err(outs[0][0].cpu(), y_ref) < 2e-6
    assert DiffAug(prob=0.0).aug(x1) is x1
    with pytest.raises(NotImplementedError):
        DiffAug().aug(x1, 0.5)


class _RandQueue:
    """torch.rand returns the queued tensors (moved to the requested device) instead of drawing: replays the fixture's draws through the
    drop-in DiffAug, which consumes torch.rand(3) then torch.rand(7, B, 1, 1, device=...) like the reference."""

    def __init__(self, queued):
        self.queue = list(queued)

    def __enter__(self):
        self._orig = torch.rand

        def rand(*size, **kw):
            t = self.queue.pop(0)
            return t.to(kw["device"]) if "device" in kw else t
        torch.rand = rand
        return self

    def __exit__(self, *a):
        torch.rand = self._orig
        assert not self.queue


def test_gan_loss_terms_vs_reference_fixture():
    """losses.generator_gan_term / discriminator_loss against train_tokenizer.VAELossFunction.forward_generator (discriminator branch on)
    and forward_discriminator, captured with a conv stand-in for the decoder's last layer (oracle/capture_golden_gan.py)."""
    from dmvae_amd import losses
    from dmvae_amd.utils.diffaug import DiffAug
    from dmvae_amd.utils.lpips import LPIPS
    from test_oracle_golden import lpips_params
    g = load_golden("gan_losses")
    lp = LPIPS().eval().requires_grad_(False)
    sd = lp.state_dict()
    for k, v in lpips_params(g).items():
        sd[k] = v.reshape(sd[k].shape)
    lp.load_state_dict(sd)
    lp = lp.to(DEV)
    disc_p = patchgan_params(g, int(g["disc_seed"]))
    disc = _disc(disc_p)
    img, feat = g.t("images").to(DEV), g.t("feat").to(DEV)
    last = g.t("last").to(DEV).requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        recon = (F.conv2d(feat, last, padding=1) + 0.9 * img).float()
        l1, l2 = losses.l1_mse(recon, img, 1.0, 0.0)
        rec_loss = l1 + lp(img, recon)
        B = img.shape[0]
        with _RandQueue([torch.zeros(3), g.t("gen_rand01").view(7, B, 1, 1)]):
            total, d_weight = losses.generator_gan_term(rec_loss, recon, disc, DiffAug(prob=1.0, cutout=0.2), last, 0.5)
    assert not disc.training and all(not p.requires_grad for p in disc.parameters())
    assert abs(rec_loss.item() - float(g["gen_rec_loss"])) < 2e-2 * float(g["gen_rec_loss"])
    assert abs(d_weight.item() - float(g["d_weight"])) < 8e-2 * float(g["d_weight"])          # ratio of two bf16-path gradient norms
    assert abs(total.item() - float(g["gen_loss"])) < 3e-2 * abs(float(g["gen_loss"]))
    total.backward()
    rl2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    assert rl2(last.grad.cpu(), g.t("g_last")) < 8e-2
    # bf16-site oracle on the same draws: tighter
    lpo = lpips_params(g)
    lo = g.t("last").requires_grad_(True)
    ro = F.conv2d(g.t("feat"), lo, padding=1) + 0.9 * g.t("images")
    to, logo = R.forward_generator_gan(g.t("images"), ro, lpo, disc_p, lo, g.t("gen_rand01"), 0.5, q=Q)
    assert abs(d_weight.item() - logo["d_weight"].item()) < 5e-2 * logo["d_weight"].item()
    assert abs(total.item() - to.item()) < 2e-2 * abs(to.item())
    # discriminator step terms (bcr weight 4, strong cut-out 0.5 as captured); buffers continue from the generator pass
    draws = [torch.zeros(3), g.t("d_rand01_a").view(7, 2 * B, 1, 1), torch.zeros(3), g.t("d_rand01_b").view(7, 2 * B, 1, 1)]
    with torch.autocast("cuda", dtype=torch.bfloat16), _RandQueue(draws):
        d_total, dlog = losses.discriminator_loss(img, g.t("recon").to(DEV), disc, DiffAug(prob=1.0, cutout=0.2), DiffAug(prob=1, cutout=0.5), 4.0)
    assert disc.training
    assert abs(d_total.item() - float(g["d_total"])) < 3e-2 * abs(float(g["d_total"]))
    assert abs(dlog["d_loss"].item() - float(g["dlog.d_loss"])) < 2e-2 * float(g["dlog.d_loss"])
    assert abs(dlog["bcr_loss"].item() - float(g["dlog.bcr_loss"])) < 8e-2 * float(g["dlog.bcr_loss"]) + 1e-3
    # 72 logits per half: one sign flip of a near-zero logit under bf16 = 1.4 points
    assert abs(dlog["acc_real"].item() - float(g["dlog.acc_real"])) <= 3.0 and abs(dlog["acc_fake"].item() - float(g["dlog.acc_fake"])) <= 3.0
    d_total.backward()
    sd = disc.state_dict()
    for k, v in g.sub("buf2.").items():
        if "num_batches" in k:
            assert int(sd[k]) == int(v) == 2, k                 # two training-mode passes
        else:
            assert rel_err(sd[k].cpu(), v) < 1e-2, k
    for n, prm in disc.named_parameters():
        gn = g["dgn." + n][0]
        if gn > 1e-3:
            assert abs(prm.grad.double().norm().item() - gn) < 6e-2 * gn, n
    for k, v in g.sub("dg.").items():
        assert rl2(dict(disc.named_parameters())[k].grad.cpu(), v) < 0.12, k


@pytest.mark.parametrize("n,hw,cin,cout", [(32, 14, 64, 128), (32, 32, 256, 512)])
def test_k4_stride1_weight_gradient_on_the_im2col_form(n, hw, cin, cout, monkeypatch):
    """functional.ConvK4Fn.backward: the 4x4 stride-1 conv's weight gradient as the 1x1 weight gradient of the im2col operand on the large kernel (the PatchGAN's
    256 -> 512 layer, models/patchgan.py:125-147) against the gather route through the small-shape kernel, and (small case) against fp64 autograd."""
    import torch.nn.functional as F
    from dmvae_amd import functional as Fn, ops
    g = torch.Generator().manual_seed(cin + hw)
    x0 = torch.randn(n, hw, hw, cin, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(cout, cin, 4, 4, generator=g) * 0.05).cuda().requires_grad_(True)
    b = torch.randn(cout, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(n, hw - 1, hw - 1, cout, generator=g).cuda().to(torch.bfloat16)
    res = {}
    for route in ((1,), ()):
        monkeypatch.setattr(Fn, "K4_WGRAD_AS_GEMM", route)
        w.grad = b.grad = None
        x = x0.clone().requires_grad_(True)
        y = Fn.ConvK4Fn.apply(x, w, b, 1, ops.ACT_LEAKY, False)
        y.backward(dy)
        res[route] = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
    for a, c in zip(res[(1,)][:2], res[()][:2]):
        assert torch.equal(a, c)                                  # forward and input gradient do not depend on the route
    assert rel_err(res[(1,)][2], res[()][2]) < 1e-5 and rel_err(res[(1,)][3], res[()][3]) < 1e-5
    if n * (hw - 1) ** 2 * cout * cin < 10 ** 9:
        xr = x0.float().cpu().double().permute(0, 3, 1, 2)
        wr = w.detach().to(torch.bfloat16).float().cpu().double().requires_grad_(True)
        br = b.detach().cpu().double().requires_grad_(True)
        yr = F.conv2d(xr, wr, br, stride=1, padding=1)
        gy = (dy.float() * torch.where(res[(1,)][0].float() > 0, 1.0, 0.2)).to(torch.bfloat16).float().cpu().double().permute(0, 3, 1, 2)      # the gate's bf16 result
        yr.backward(gy)
        assert rel_err(res[(1,)][2].cpu(), wr.grad) < 1e-4 and rel_err(res[(1,)][3].cpu(), br.grad) < 1e-4


def test_k4_first_layer_weight_gradient_on_the_eight_channel_im2col_form(monkeypatch):
    """ConvK4Fn.backward for the PatchGAN's first layer (3 real input channels in a 32-channel tensor, stride 2): the im2col route on an 8-channel copy through the
    large kernel against the gather route through the small-shape kernel and against fp64 autograd."""
    from dmvae_amd import functional as Fn, ops
    g = torch.Generator().manual_seed(11)
    n, hw, cin, cout = 16, 64, 3, 64
    x0 = torch.zeros(n, hw, hw, 32)
    x0[..., :cin] = torch.randn(n, hw, hw, cin, generator=g)
    x0 = x0.cuda().to(torch.bfloat16)
    w = (torch.randn(cout, cin, 4, 4, generator=g) * 0.1).cuda().requires_grad_(True)
    b = torch.randn(cout, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(n, hw // 2, hw // 2, cout, generator=g).cuda().to(torch.bfloat16)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(Fn, "K4_WGRAD_THIN_CIN", on)
        w.grad = b.grad = None
        y = Fn.ConvK4Fn.apply(x0, w, b, 2, ops.ACT_NONE, False)
        y.backward(dy)
        res[on] = (y.detach().clone(), w.grad.clone(), b.grad.clone())
    assert torch.equal(res[True][0], res[False][0])
    assert rel_err(res[True][1], res[False][1]) < 1e-5 and rel_err(res[True][2], res[False][2]) < 1e-5
    xr = x0[..., :cin].float().cpu().double().permute(0, 3, 1, 2)
    wr = w.detach().to(torch.bfloat16).float().cpu().double().requires_grad_(True)
    br = b.detach().cpu().double().requires_grad_(True)
    F.conv2d(xr, wr, br, stride=2, padding=1).backward(dy.float().cpu().double().permute(0, 3, 1, 2))
    assert rel_err(res[True][1].cpu(), wr.grad) < 1e-5 and rel_err(res[True][2].cpu(), br.grad) < 1e-5


def test_k4_conv_computes_no_weight_gradient_for_a_frozen_discriminator(monkeypatch):
    """losses.generator_gan_term runs the generator's adversarial term through a frozen discriminator (train_tokenizer.py:190-203), twice per step: ConvK4Fn.backward
    must then launch no weight-gradient kernel (it did: 2 x 1.1 ms per step) and return the same input gradient."""
    from dmvae_amd import functional as Fn, ops
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(4, 16, 16, 64, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(128, 64, 4, 4, generator=g) * 0.05).cuda()
    b = torch.randn(128, generator=g).cuda()
    dy = torch.randn(4, 8, 8, 128, generator=g).cuda().to(torch.bfloat16)
    xa = x0.clone().requires_grad_(True)
    wa, ba = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    Fn.ConvK4Fn.apply(xa, wa, ba, 2, ops.ACT_LEAKY, False).backward(dy)
    assert wa.grad is not None and ba.grad is not None

    def boom(*a, **k):
        raise AssertionError("weight gradient computed for frozen parameters")
    monkeypatch.setattr(ops, "conv2d_nhwc_wgrad", boom)
    monkeypatch.setattr(ops, "im2col", boom)
    xb = x0.clone().requires_grad_(True)
    Fn.ConvK4Fn.apply(xb, w, b, 2, ops.ACT_LEAKY, False).backward(dy)
    assert torch.equal(xb.grad, xa.grad)


def test_single_pass_generator_backward_equals_the_three_sweeps():
    """losses.generator_gan_backward (one backward through LPIPS and one through the frozen discriminator, the adaptive weight from the last layer's own backward
    node) against losses.generator_gan_term + backward (the reference's autograd.grad twice + backward, train_tokenizer.py:190-203,414) on the captured inputs and
    DiffAug draws: same loss value and adaptive weight, the same gradients up to the bf16 rounding of a scaled gradient inside the discriminator's backward."""
    from dmvae_amd import losses
    from dmvae_amd.utils.diffaug import DiffAug
    from dmvae_amd.utils.lpips import LPIPS
    from test_oracle_golden import lpips_params
    g = load_golden("gan_losses")
    lp = LPIPS().eval().requires_grad_(False)
    sd = lp.state_dict()
    for k, v in lpips_params(g).items():
        sd[k] = v.reshape(sd[k].shape)
    lp.load_state_dict(sd)
    lp = lp.to(DEV)
    disc = _disc(patchgan_params(g, int(g["disc_seed"])))
    img = g.t("images").to(DEV)
    B = img.shape[0]
    res = {}
    for single in (False, True):
        feat = g.t("feat").to(DEV).requires_grad_(True)
        last = g.t("last").to(DEV).requires_grad_(True)
        side = torch.zeros(3, device=DEV, requires_grad=True)             # a term that does not run through recon (the KL / MMD stand-in)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            recon = (F.conv2d(feat, last, padding=1) + 0.9 * img).float()
            l1, l2 = losses.l1_mse(recon, img, 1.0, 0.0)
            extra = (side * torch.tensor([1.0, -2.0, 0.5], device=DEV)).sum()
            rec_loss = l1 + lp(img, recon) + extra
            with _RandQueue([torch.zeros(3), g.t("gen_rand01").view(7, B, 1, 1)]):
                if single:
                    total, d_weight = losses.generator_gan_backward(rec_loss, recon, disc, DiffAug(prob=1.0, cutout=0.2), last, 0.5, extra=extra)
                else:
                    total, d_weight = losses.generator_gan_term(rec_loss, recon, disc, DiffAug(prob=1.0, cutout=0.2), last, 0.5)
        if not single:
            total.backward()
        res[single] = (total.detach().item(), d_weight.item(), last.grad.clone(), feat.grad.clone(), side.grad.clone())
    (t0, w0, gl0, gf0, gs0), (t1, w1, gl1, gf1, gs1) = res[False], res[True]
    rl2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
    assert abs(t1 - t0) <= 1e-5 * abs(t0) and abs(w1 - w0) <= 1e-4 * abs(w0), (t0, t1, w0, w1)        # same forward, same two norms
    # the discriminator's backward rounds its gradients to bf16 at five layers; the two evaluations feed it the same gradient at two scales (1 and d_weight), so its
    # roundings differ: 3e-3 / 7e-3 measured -- a tenth of what either is from the reference's fp32 gradient (the 8e-2 bar of the fixture test above, held here too)
    assert rl2(gl1, gl0) < 2e-2 and rl2(gf1, gf0) < 2e-2, (rl2(gl1, gl0), rl2(gf1, gf0))
    assert rl2(gl1.cpu(), g.t("g_last")) < 8e-2 and rl2(gl1.cpu(), g.t("g_last")) < 1.1 * rl2(gl0.cpu(), g.t("g_last")) + 1e-3
    assert torch.equal(gs1, gs0)                                          # the term outside recon takes part in the final backward
    assert abs(w1 - float(g["d_weight"])) < 8e-2 * float(g["d_weight"])  # and it is the reference's adaptive weight


@pytest.mark.gpu
@pytest.mark.parametrize("n,h,c", [(3, 31, 512), (2, 9, 1024), (64, 31, 512)])
def test_k4_logits_layer_on_the_vector_unit_kernels(n, h, c, monkeypatch):
    """functional.ConvK4Fn for the one-output-channel 4x4 stride-1 conv (the PatchGAN's logits layer, models/patchgan.py:146) on csrc/conv_c1.hip: forward, input
    gradient, weight and bias gradient against fp64 autograd on the bf16-rounded operands and against the matrix-core route it replaces; frozen parameters get no
    weight gradient; two runs give the same bits."""
    from dmvae_amd import functional as Fn, ops
    g = torch.Generator().manual_seed(n + h + c)
    x0 = torch.randn(n, h, 31, c, generator=g).cuda().to(BF)
    w = (torch.randn(1, c, 4, 4, generator=g) * 0.02).cuda().requires_grad_(True)
    b = torch.randn(1, generator=g).cuda().requires_grad_(True)
    dy = torch.randn(n, h - 1, 30, 1, generator=g).cuda()
    res = {}
    for on in (True, False, True):
        monkeypatch.setattr(Fn, "K4_COUT1", on)
        w.grad = b.grad = None
        x = x0.clone().requires_grad_(True)
        y = Fn.ConvK4Fn.apply(x, w, b, 1, ops.ACT_NONE, True)
        assert y.dtype == torch.float32 and y.shape == (n, h - 1, 30, 1)
        y.backward(dy)
        cur = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
        if on and True in res:
            assert all(torch.equal(a, c_) for a, c_ in zip(cur, res[True]))
        res[on] = cur
    xr = x0.float().cpu().double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.detach().to(BF).float().cpu().double().requires_grad_(True)
    br = b.detach().cpu().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, stride=1, padding=1)
    yr.backward(dy.cpu().double().permute(0, 3, 1, 2))
    new, old = res[True], res[False]
    assert rel_err(new[0].cpu().permute(0, 3, 1, 2), yr.detach()) < 2e-6
    assert rel_err(new[1].float().cpu().permute(0, 3, 1, 2), xr.grad) < 5e-3          # bf16 result: 2^-9 per element
    assert rel_err(new[2].cpu(), wr.grad) < 2e-6 and rel_err(new[3].cpu(), br.grad) < 2e-6
    assert rel_err(new[0], old[0]) < 1e-5 and rel_err(new[1].float(), old[1].float()) < 8e-3 and rel_err(new[2], old[2]) < 4e-3 and rel_err(new[3], old[3]) < 2e-2      # the old route rounds dy to bf16 on its way into the matrix kernels
    # frozen parameters: no weight-gradient launch, the same input gradient
    monkeypatch.setattr(Fn, "K4_COUT1", True)

    def boom(*a, **k):
        raise AssertionError("weight gradient computed for frozen parameters")
    monkeypatch.setattr(ops, "conv_k4c1_wgrad", boom)
    xb = x0.clone().requires_grad_(True)
    Fn.ConvK4Fn.apply(xb, w.detach(), b.detach(), 1, ops.ACT_NONE, True).backward(dy)
    assert torch.equal(xb.grad, new[1])


@pytest.mark.gpu
def test_k4_logits_layer_kernels_are_adjoint_at_full_size():
    """Size-independent property at the training shape (B = 64, 31 x 31 x 512): the three kernels of csrc/conv_c1.hip compute one bilinear form,
    <conv(x, w), dy> = <x, dgrad(dy, w)> = <w, wgrad(x, dy)>, up to the bf16 rounding of dgrad's result and f32 summation."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(77)
    x = torch.randn(64, 31, 31, 512, generator=g).cuda().to(BF)
    w = (torch.randn(1, 512, 4, 4, generator=g) * 0.02).cuda()
    dy = torch.randn(64, 30, 30, 1, generator=g).cuda()
    y = ops.conv_k4c1_fwd(x, w, None)
    dx = ops.conv_k4c1_dgrad(dy, w, 31)
    dw, db = ops.conv_k4c1_wgrad(x, dy)
    wb = w.to(BF).double()                                   # the kernels multiply by the bf16-rounded weight
    a = (y.double() * dy.double()).sum().item()
    b = (x.double() * dx.double()).sum().item()
    c = (wb * dw.double()).sum().item()
    scale = (y.double().abs() * dy.double().abs()).sum().item()
    assert abs(a - c) < 1e-5 * scale, (a, c)
    assert abs(a - b) < 3e-4 * scale, (a, b)                  # dx is rounded to bf16 element by element
    assert abs(db.item() - dy.double().sum().item()) < 1e-5 * dy.double().abs().sum().item()
