"""Module-level parity (-m gpu): the drop-in nn.Modules (dmvae_amd.models) vs (a) the golden vectors captured from the
reference's own modules and (b) the CPU oracle evaluated with bf16 rounding at the HIP path's storage points.

Tolerances (relative to the tensor's max-abs):
  * vs oracle-with-bf16-sites: per module, 2-3 x what MI355X measures (round 4: ResnetBlock y <= 6e-5 / dx <= 6.5e-4 / parameter gradients <= 9e-6; Upsample and
    Downsample exact; AttnBlock y 2e-4 / dx 4e-3 / q.weight 3e-3 -- its softmax probabilities are one more bf16 site the S x S scores pass through) -- both
    sides round activations to bf16 at the same sites, so differences are f32 accumulation order plus rare 1-spacing bf16 flips (2^-8) that propagate
    (`test_resnet_block_stage_by_stage_on_the_production_kernels` counts them: 2e-5 ... 2e-4 of the elements per stage);
  * vs the reference's pure-f32 goldens: 3e-2 -- bf16 autocast-level agreement (the reference itself trains under
    autocast(bf16), train_tokenizer.py:410).
The 1e-4 f32 bar is enforced per kernel on identical inputs in test_gpu_kernels.py."""
import warnings

import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from oracle.detweights import det_tensor

pytestmark = pytest.mark.gpu
DEV = "cuda"
Q = R.bf16_round
TOL_Q, TOL_REF = 1e-2, 3e-2        # TOL_Q: the deep stacks (flux Encoder: 9 convs / 8 GroupNorms); single modules pass their own bar to _run_block


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _load(mod, params):
    sd = mod.state_dict()
    for k in sd:
        if k in params:
            sd[k] = params[k].to(sd[k].dtype)
    mod.load_state_dict(sd, strict=True)
    return mod.to(DEV)


def _run_block(mod, g, oracle_fn, tol_q=TOL_Q):
    p = g.sub("p.")
    _load(mod, p)
    x = g.t("x").to(DEV).requires_grad_(True)
    y = mod(x)
    y.backward(g.t("dy").to(DEV))
    # oracle with bf16 storage sites
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xo = g.t("x").requires_grad_(True)
    yo = oracle_fn(Q(xo), po, "", Q)
    yo.backward(Q(g.t("dy")))
    pg = [(n, rel_err(prm.grad.cpu(), po[n].grad)) for n, prm in mod.named_parameters() if g.t("g." + n).abs().max() >= 1e-4]
    print(f"{type(mod).__name__}: vs bf16-site oracle: y {rel_err(y.float().cpu(), yo.detach()):.2e}  dx {rel_err(x.grad.cpu(), xo.grad):.2e}  "
          f"worst parameter gradient {max(pg, key=lambda t: t[1])}")
    assert rel_err(y.float().cpu(), yo.detach()) < tol_q
    assert rel_err(y.float().cpu(), g.t("y")) < TOL_REF
    assert rel_err(x.grad.cpu(), xo.grad) < tol_q
    assert rel_err(x.grad.cpu(), g.t("dx")) < TOL_REF
    for n, prm in mod.named_parameters():
        ref, orc = g.t("g." + n), po[n].grad
        if ref.abs().max() < 1e-4:      # exactly-zero gradients (see test_oracle_golden): noise on both sides
            assert prm.grad.abs().max() < 1e-2
            continue
        assert rel_err(prm.grad.cpu(), orc) < tol_q, n
        assert rel_err(prm.grad.cpu(), ref) < TOL_REF, n


@pytest.mark.parametrize("name,cin,cout", [("resblock_same", 64, 64), ("resblock_short", 128, 64)])
def test_resnet_block(name, cin, cout):
    from dmvae_amd.models.flux_ae import ResnetBlock
    _run_block(ResnetBlock(cin, cout), load_golden(name), R.resnet_block, tol_q=2e-3)      # measured: y 0 / 6e-5, dx 1e-4 / 6.5e-4, parameter gradients 6e-7 / 9e-6 (same / short)


def test_attn_block():
    from dmvae_amd.models.flux_ae import AttnBlock
    _run_block(AttnBlock(64), load_golden("attnblock"), R.attn_block, tol_q=8e-3)


def test_upsample():
    from dmvae_amd.models.flux_ae import Upsample
    _run_block(Upsample(32), load_golden("upsample"), R.upsample, tol_q=1e-5)


def test_downsample():
    from dmvae_amd.models.flux_ae import Downsample
    _run_block(Downsample(32), load_golden("downsample"), R.downsample, tol_q=1e-5)


def test_flux_encoder_small_fwd_bwd():
    """flux_ae.Encoder (reference :110-181; never instantiated by its scripts, so the golden is its only pin): forward against the
    captured output, forward + every gradient against the bf16-site oracle."""
    from dmvae_amd.models.flux_ae import Encoder
    g = load_golden("flux_encoder_small")
    p = g.sub("p.")
    enc = _load(Encoder(resolution=16, in_channels=32, ch=32, ch_mult=[1, 2], num_res_blocks=1, z_channels=16), p)
    x = g.t("x").to(DEV).requires_grad_(True)
    y = enc(x)
    assert y.shape == (1, 32, 8, 8)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    y.backward(dy.to(DEV))
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xo = g.t("x").requires_grad_(True)
    yo = R.encoder_forward(Q(xo), po, q=Q, num_resolutions=2, num_res_blocks=1)
    yo.backward(Q(dy))
    # exact (f32, no rounding sites) gradients tell which parameters have identically-zero gradients: with ch = 32 every GroupNorm
    # group is a single channel, so a conv bias feeding a GroupNorm is removed by the mean subtraction
    pe = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    R.encoder_forward(g.t("x"), pe, num_resolutions=2, num_res_blocks=1).backward(dy)
    assert rel_err(y.float().cpu(), g.t("y")) < TOL_REF
    assert rel_err(y.float().cpu(), yo.detach()) < 2 * TOL_Q      # 9 convs / 8 GroupNorms deep
    assert rel_err(x.grad.cpu(), xo.grad) < 3 * TOL_Q
    checked = 0
    for n, prm in enc.named_parameters():
        assert prm.grad is not None, n
        if pe[n].grad.abs().max() < 1e-4:
            assert prm.grad.abs().max() < 5e-2, n                 # bf16 noise on both sides
            continue
        assert rel_err(prm.grad.cpu(), po[n].grad) < 3 * TOL_Q, n
        checked += 1
    assert checked >= 40


def test_flux_encoder_rgb_input():
    """in_channels = 3 (the reference's AutoEncoderParams default): conv_in pads the image to 32 channels on the way to NHWC."""
    from dmvae_amd.models.flux_ae import Encoder
    enc = Encoder(resolution=32, in_channels=3, ch=32, ch_mult=[1, 2], num_res_blocks=1, z_channels=16)
    params = {k: det_tensor(k, v.shape, 17) for k, v in enc.state_dict().items()}
    _load(enc, params)
    x = (torch.rand(2, 3, 32, 32, generator=torch.Generator().manual_seed(3)) * 2 - 1).to(DEV).requires_grad_(True)
    y = enc(x)
    assert y.shape == (2, 32, 16, 16) and y.dtype == torch.float32
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(4))
    y.backward(dy.to(DEV))
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    xo = x.detach().cpu().requires_grad_(True)
    yo = R.encoder_forward(xo, po, q=Q, num_resolutions=2, num_res_blocks=1)
    yo.backward(Q(dy))
    assert rel_err(y.cpu(), yo.detach()) < 2 * TOL_Q
    assert rel_err(x.grad.cpu(), xo.grad) < 3 * TOL_Q
    for n in ("conv_in.weight", "conv_in.bias", "down.0.downsample.conv.weight", "conv_out.weight", "norm_out.weight"):
        assert rel_err(dict(enc.named_parameters())[n].grad.cpu(), po[n].grad) < 3 * TOL_Q, n


def test_mlp():
    from dmvae_amd.models.vae import MLP
    g = load_golden("mlp")
    mod = _load(MLP(64, 32, hidden_dim=128), g.sub("p."))
    x = g.t("x").to(DEV).requires_grad_(True)
    y = mod(x)
    y.backward(g.t("dy").to(DEV))
    assert rel_err(y.float().cpu(), g.t("y")) < TOL_REF
    assert rel_err(x.grad.cpu(), g.t("dx")) < TOL_REF
    for n, prm in mod.named_parameters():
        assert rel_err(prm.grad.cpu(), g.t("g." + n)) < TOL_REF, n


def _decoder(ch, seed):
    from dmvae_amd.models.flux_ae import Decoder
    dec = Decoder(ch=ch, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, resolution=256, z_channels=16)
    dec.post_init(z_channels=32)
    params = {k: det_tensor(k, v.shape, seed) for k, v in dec.state_dict().items()}
    return _load(dec, params), params


def test_decoder_small_fwd_bwd():
    g = load_golden("decoder_small")
    dec, params = _decoder(32, 12)
    z = g.t("z").to(DEV).requires_grad_(True)
    y = dec(z)
    assert y.dtype == torch.float32 and y.shape == (2, 3, 64, 64)
    y.backward(g.t("dy").to(DEV))
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    zo = g.t("z").requires_grad_(True)
    yo = R.decoder_forward(zo, po, q=Q)
    yo.backward(g.t("dy"))
    # 40 convs / 30 GroupNorms deep: a single 1-ulp bf16 flip early on perturbs thousands of downstream sums and
    # re-rolls their rounding, so two correct bf16 pipelines decorrelate to the bf16 noise floor (measured stage by
    # stage with tools/probes/t_dec.py; single blocks match the bf16-site oracle bit-for-bit, see _run_block).
    # Criterion: the HIP path must be as close to the reference's f32 result as the bf16-site oracle is.
    def floor(hip, orc, ref, what, slack=1.15, abs_floor=1e-3):      # measured on MI355X (round 4): e_hip / e_orc = 0.89 ... 1.04 over y, dz and the captured parameter gradients
        e_hip, e_orc = rel_l2(hip, ref), rel_l2(orc, ref)
        print(f"decoder_small {what}: rel-L2 to f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
        assert e_hip < slack * e_orc + abs_floor, what
    floor(y.cpu(), yo.detach(), g.t("y"), "y")
    floor(z.grad.cpu(), zo.grad, g.t("dz"), "dz")
    for n, prm in dec.named_parameters():
        if "g." + n in g and g["gn." + n][0] > 1e-3:
            floor(prm.grad.cpu(), po[n].grad, g.t("g." + n), "grad " + n)
        gn = g["gn." + n][0]
        if gn > 1e-3:   # every parameter: gradient norm within 5% of the reference's
            assert abs(prm.grad.double().norm().item() - gn) < 5e-2 * gn, n


def test_decoder_full_b1_tokens():
    g = load_golden("decoder_full_b1")
    dec, _ = _decoder(128, 22)
    with torch.no_grad():
        y = dec(g.t("z").to(DEV))
    assert y.shape == (1, 3, 256, 256)
    assert rel_err(y[0, :, ::8, ::8].cpu(), g.t("y_slice")) < TOL_REF
    assert abs(y.double().abs().sum().item() - g["y_sum"][1]) < 2e-2 * g["y_sum"][1]


def test_decoder_full_b1_backward_production_path():
    """The FULL-width decoder's backward on the PRODUCTION bf16 path against the reference's f32 capture (oracle/capture_golden_bwd.py), by the same criterion as
    `test_decoder_small_fwd_bwd`: as close to the reference as the bf16-site oracle is (d z, the six captured gradient slices), every parameter-gradient norm
    within 5 % -- the large-tile conv / weight-gradient / GroupNorm instantiations through a whole module."""
    from oracle.capture_golden_bwd import SLICES
    g = load_golden("decoder_full_b1_bwd")
    dec, params = _decoder(128, 22)
    z0 = load_golden("decoder_full_b1").t("z")
    dy = torch.randn(1, 3, 256, 256, generator=torch.Generator().manual_seed(23))
    z = z0.to(DEV).requires_grad_(True)
    y = dec(z)
    y.backward(dy.to(DEV))
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    zo = z0.clone().requires_grad_(True)
    yo = R.decoder_forward(zo, po, q=Q)
    yo.backward(dy)

    def floor(hip, orc, ref, what, slack=1.15, abs_floor=1e-3):
        e_hip, e_orc = rel_l2(hip, ref), rel_l2(orc, ref)
        print(f"decoder_full_b1 {what}: rel-L2 to f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
        assert e_hip < slack * e_orc + abs_floor, what
    floor(z.grad.cpu(), zo.grad, g.t("dz"), "dz")
    for n, prm in dec.named_parameters():
        if "g." + n in g:
            floor(prm.grad[SLICES[n]].cpu(), po[n].grad[SLICES[n]], g.t("g." + n), "grad " + n)
        gn = g["gn." + n][0]
        if gn > 1e-3:
            assert abs(prm.grad.double().norm().item() - gn) < 5e-2 * gn, n


def test_vae_forward_tiny_and_api(allow_stock):
    # allow_stock: the capture's reduced ViT (width 64, 4 heads) is outside the bf16 encoder kernels' range; the fp32 parity mode runs this fixture with
    # the encoder on the MFMA GEMM route (tests/test_gpu_parity_fp32.py) and tests/test_gpu_vit_pin.py pins the bf16 encoder routes to the reference
    from test_oracle_golden import vae_tiny_params
    g = load_golden("vae_forward_tiny")
    p, vae = vae_tiny_params()
    vae.load_state_dict(p, strict=True)
    vae = vae.to(DEV)
    x = (torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(32)) * 2 - 1).to(DEV)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        rec, lat = vae(x, return_latent=True)
    assert rec.dtype == torch.float32 and lat.shape == (1, 256, 32)
    assert rel_err(lat.float().cpu(), g.t("latent")) < TOL_REF
    assert rel_err(rec[0, :, ::8, ::8].cpu(), g.t("rec_slice")) < TOL_REF
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert torch.equal(vae.encode(x), lat)
        assert torch.equal(vae.decode(lat).float(), rec)
    # get_last_layer must be a leaf usable with autograd.grad(retain_graph=True) (train_tokenizer.py:194-196)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        rec = vae(x)
    last = vae.decoder.get_last_layer()
    g1 = torch.autograd.grad(rec.abs().mean(), last, retain_graph=True)[0]
    g2 = torch.autograd.grad((rec ** 2).mean(), last, retain_graph=True)[0]
    assert g1.shape == last.shape and g2.shape == last.shape and g1.norm() > 0 and g2.norm() > 0


def test_cpu_tensor_fails_loudly():
    from dmvae_amd import _lib, ops
    with pytest.raises(_lib.DmvaeHipError):
        ops.groupnorm_stats(torch.zeros(1, 4, 4, 32, dtype=torch.bfloat16))


def test_conv_out_weight_gradient_via_gradient_im2col():
    """NormConvOutFn's weight / bias gradient for >= 16384 pixels comes from im2col of the output gradient + one 1x1 weight-gradient GEMM
    (functional.py); it must equal the direct 3x3 weight-gradient kernel on the same operands (f32 accumulation on both sides)."""
    from dmvae_amd import functional as Fn, ops
    g = torch.Generator().manual_seed(9)
    n, h, w_, c = 2, 96, 100, 128
    x = torch.randn(n, h, w_, c, generator=g).to(DEV).to(torch.bfloat16).requires_grad_(True)
    nw, nb = (1 + 0.2 * torch.randn(c, generator=g)).to(DEV).requires_grad_(True), (0.1 * torch.randn(c, generator=g)).to(DEV).requires_grad_(True)
    cw = (0.05 * torch.randn(3, c, 3, 3, generator=g)).to(DEV).requires_grad_(True)
    cb = (0.1 * torch.randn(3, generator=g)).to(DEV).requires_grad_(True)
    y = Fn.NormConvOutFn.apply(x, nw, nb, cw, cb)
    assert y.shape == (n, 3, h, w_) and y.dtype == torch.float32
    dy = torch.randn(y.shape, generator=g).to(DEV)
    y.backward(dy)
    st = ops.groupnorm_stats(x.detach())
    a = ops.groupnorm_apply(x.detach(), st, nw.detach(), nb.detach(), True)
    dyp = ops.nchw_to_nhwc_bf16(dy, c_pad=32)
    dw_ref, db_ref = ops.conv2d_nhwc_wgrad(dyp, a, 3)
    assert rel_err(cw.grad, dw_ref[:3]) < 1e-5
    assert rel_err(cb.grad, db_ref[:3]) < 1e-5


def _ulp_bf16(v):
    """bf16 spacing at |v| (8 significant bits): 2^(floor(log2 |v|) - 7), with the normal-range floor."""
    return torch.exp2(torch.floor(torch.log2(v.abs().clamp_min(2.0 ** -120))) - 7)


def _assert_bf16_of(hip, ref64, what, max_flip_frac=2e-3, slack=1e-4):
    """`hip` (bf16, from a production kernel) must be the bf16 rounding of the oracle's f64 result `ref64` computed from the SAME inputs: equal to RNE(ref64)
    except for a small fraction of elements (f32 accumulation order puts a value on the other side of a rounding boundary), and those within one bf16 spacing of
    ref64 -- i.e. every element is a correct rounding of a value within f32 accumulation error of the oracle's."""
    h = hip.double().cpu()
    r = ref64.to(torch.float32).to(torch.bfloat16).double()
    flips = h != r
    frac = flips.double().mean().item()
    worst = ((h - ref64).abs() / (_ulp_bf16(ref64) + slack * ref64.abs().max())).max().item()
    assert frac <= max_flip_frac, f"{what}: {frac:.2e} of the elements differ from RNE(oracle)"
    assert worst <= 1.0, f"{what}: an element is {worst:.2f} bf16 spacings from the oracle"
    return frac


def test_resnet_block_stage_by_stage_on_the_production_kernels():
    """Where the module-level tolerance comes from, and the only place where the PRODUCTION bf16 GroupNorm / swish / conv-epilogue kernels meet the oracle inside a
    module: ResnetBlock(256 -> 128) at 4 x 64 x 64 (large enough for the kernels the C2 step runs: conv_pp with the GroupNorm-statistics epilogue, wgrad_pp, the
    chunked GroupNorm passes), forward and backward replayed call by call exactly as functional.ResnetBlockFn issues them, each call's result compared with the
    oracle's arithmetic (oracle/ref_cpu.py: group_norm, swish, conv2d; flux_ae.py:55-82) evaluated in f64 on THAT call's actual inputs.  Every bf16 result must be
    the correct rounding of the oracle's value (<= 1 spacing, and != RNE(oracle) for at most 0.2 % of the elements), every f32 result within 1e-5.  The block-level
    comparison (`test_resnet_block`) then differs from the oracle only through those rare one-spacing flips propagating -- which is what its bar measures."""
    import torch.nn.functional as F
    from dmvae_amd import ops
    from dmvae_amd.functional import packed
    from conftest import elem_err
    n, hh, ww, cin, cout = 4, 64, 64, 256, 128
    g = torch.Generator().manual_seed(77)
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (rn(n, hh, ww, cin) * 1.3 + 0.2).to(torch.bfloat16)
    dy = rn(n, hh, ww, cout, sc=0.7).to(torch.bfloat16)
    P = {"n1w": 1 + 0.3 * rn(cin), "n1b": 0.2 * rn(cin), "c1w": rn(cout, cin, 3, 3, sc=0.03), "c1b": 0.1 * rn(cout),
         "n2w": 1 + 0.3 * rn(cout), "n2b": 0.2 * rn(cout), "c2w": rn(cout, cout, 3, 3, sc=0.04), "c2b": 0.1 * rn(cout),
         "sw": rn(cout, cin, 1, 1, sc=0.06), "sb": 0.1 * rn(cout)}
    D = {k: v.to(DEV) for k, v in P.items()}
    xd, dyd = x.to(DEV), dy.to(DEV)
    nchw = lambda t: t.double().cpu().permute(0, 3, 1, 2)            # NHWC device tensor -> NCHW f64 on the CPU (the oracle's layout)
    wq = lambda k: P[k].to(torch.bfloat16).double()                  # the bf16 operand the kernels read (functional.packed rounds the f32 master RNE)

    def stats64(t):                                                  # (mean, rstd) per (sample, group) like the kernels' [n, 32, 2]
        v = nchw(t).reshape(n, 32, -1)
        m = v.mean(2)
        return torch.stack([m, 1.0 / torch.sqrt(((v - m[..., None]) ** 2).mean(2) + 1e-6)], 2)

    def gn_swish64(t, w, b):                                         # the oracle's own functions, f64
        return R.swish(R.group_norm(nchw(t), P[w].double(), P[b].double()))

    # ---- forward, as ResnetBlockFn.forward
    st1 = ops.groupnorm_stats(xd)
    assert rel_err(st1.cpu(), stats64(xd)) < 1e-5
    a1 = ops.groupnorm_apply(xd, st1, D["n1w"], D["n1b"], True)
    f1 = _assert_bf16_of(a1.permute(0, 3, 1, 2), gn_swish64(xd, "n1w", "n1b"), "a1 = swish(GN(x))")
    h1, st2 = ops.conv2d_nhwc_gnstats(a1, packed(D["c1w"]), D["c1b"], ks=3)
    f2 = _assert_bf16_of(h1.permute(0, 3, 1, 2), F.conv2d(nchw(a1), wq("c1w"), P["c1b"].double(), padding=1), "h1 = conv1(a1)")
    assert rel_err(st2.cpu(), stats64(h1)) < 1e-5                     # the conv epilogue's statistics are those of the bf16 tensor it stored
    a2 = ops.groupnorm_apply(h1, st2, D["n2w"], D["n2b"], True)
    _assert_bf16_of(a2.permute(0, 3, 1, 2), gn_swish64(h1, "n2w", "n2b"), "a2 = swish(GN(h1))")
    xs = ops.conv2d_nhwc(xd, packed(D["sw"]), D["sb"], ks=1)
    _assert_bf16_of(xs.permute(0, 3, 1, 2), F.conv2d(nchw(xd), wq("sw"), P["sb"].double()), "xs = nin_shortcut(x)")
    y, sty = ops.conv2d_nhwc_gnstats(a2, packed(D["c2w"]), D["c2b"], residual=xs, ks=3)
    _assert_bf16_of(y.permute(0, 3, 1, 2), F.conv2d(nchw(a2), wq("c2w"), P["c2b"].double(), padding=1) + nchw(xs), "y = conv2(a2) + xs")
    assert rel_err(sty.cpu(), stats64(y)) < 1e-5

    # ---- backward, as ResnetBlockFn.backward
    def conv_grads64(inp, wkey, dout, pad):                          # f64 autograd of the oracle's conv on this call's operands
        i = nchw(inp).requires_grad_(True)
        w = wq(wkey).requires_grad_(True)
        b = P[wkey[:-1] + "b"].double().requires_grad_(True)
        return torch.autograd.grad(F.conv2d(i, w, b, padding=pad), (i, w, b), nchw(dout))

    def gn_grads64(inp, w, b, dout, dres=None):
        i = nchw(inp).requires_grad_(True)
        ww_, bb_ = P[w].double().requires_grad_(True), P[b].double().requires_grad_(True)
        gi, gw, gb = torch.autograd.grad(R.swish(R.group_norm(i, ww_, bb_)), (i, ww_, bb_), nchw(dout))
        return (gi if dres is None else gi + nchw(dres)), gw, gb

    def f32_close(a, b, what):
        assert rel_err(a.cpu(), b) < 1e-5 and elem_err(a.cpu(), b) < 1e-4, what

    dc2w, dc2b = ops.conv2d_nhwc_wgrad(dyd, a2, 3)
    gi, gw, gb = conv_grads64(a2, "c2w", dyd, 1)
    f32_close(dc2w, gw, "dW(conv2)"); f32_close(dc2b, gb, "db(conv2)")
    da2 = ops.conv2d_nhwc(dyd, packed(D["c2w"], True), ks=3)
    _assert_bf16_of(da2.permute(0, 3, 1, 2), gi, "da2 = dgrad(conv2)")
    dh1, dn2w, dn2b = ops.groupnorm_bwd(da2, h1, st2, D["n2w"], D["n2b"], True)
    gi, gw, gb = gn_grads64(h1, "n2w", "n2b", da2)
    f3 = _assert_bf16_of(dh1.permute(0, 3, 1, 2), gi, "dh1 = GN2 backward")
    f32_close(dn2w, gw, "dgamma(norm2)"); f32_close(dn2b, gb, "dbeta(norm2)")
    dc1w, dc1b = ops.conv2d_nhwc_wgrad(dh1, a1, 3)
    gi, gw, gb = conv_grads64(a1, "c1w", dh1, 1)
    f32_close(dc1w, gw, "dW(conv1)"); f32_close(dc1b, gb, "db(conv1)")
    da1 = ops.conv2d_nhwc(dh1, packed(D["c1w"], True), ks=3)
    _assert_bf16_of(da1.permute(0, 3, 1, 2), gi, "da1 = dgrad(conv1)")
    dsw, dsb = ops.conv2d_nhwc_wgrad(dyd, xd, 1)
    gi, gw, gb = conv_grads64(xd, "sw", dyd, 0)
    f32_close(dsw, gw, "dW(nin_shortcut)"); f32_close(dsb, gb, "db(nin_shortcut)")
    dxs = ops.conv2d_nhwc(dyd, packed(D["sw"], True), ks=1)
    _assert_bf16_of(dxs.permute(0, 3, 1, 2), gi, "dxs = dgrad(nin_shortcut)")
    dx, dn1w, dn1b = ops.groupnorm_bwd(da1, xd, st1, D["n1w"], D["n1b"], True, dres=dxs)
    gi, gw, gb = gn_grads64(xd, "n1w", "n1b", da1, dres=dxs)
    _assert_bf16_of(dx.permute(0, 3, 1, 2), gi, "dx = GN1 backward + dxs")
    f32_close(dn1w, gw, "dgamma(norm1)"); f32_close(dn1b, gb, "dbeta(norm1)")
    print(f"stage-by-stage: fraction of elements != RNE(oracle): a1 {f1:.1e}, h1 {f2:.1e}, dh1 {f3:.1e}")

    # ---- and the module as a whole (autograd Function, same kernels) equals this replay bit for bit
    from dmvae_amd.models.flux_ae import ResnetBlock
    blk = ResnetBlock(cin, cout).to(DEV)
    with torch.no_grad():
        for k, t in (("norm1.weight", "n1w"), ("norm1.bias", "n1b"), ("conv1.weight", "c1w"), ("conv1.bias", "c1b"), ("norm2.weight", "n2w"), ("norm2.bias", "n2b"),
                     ("conv2.weight", "c2w"), ("conv2.bias", "c2b"), ("nin_shortcut.weight", "sw"), ("nin_shortcut.bias", "sb")):
            dict(blk.named_parameters())[k].copy_(D[t])
    xin = xd.permute(0, 3, 1, 2).float().requires_grad_(True)
    ym = blk(xin)
    ym.backward(dyd.permute(0, 3, 1, 2).float())
    assert torch.equal(ym.to(torch.bfloat16), y.permute(0, 3, 1, 2)) and torch.equal(xin.grad.to(torch.bfloat16), dx.permute(0, 3, 1, 2))
    assert torch.equal(blk.conv1.weight.grad, dc1w) and torch.equal(blk.norm2.weight.grad, dn2w)


@pytest.mark.parametrize("n,hh,ww,cin,cout", [(2, 128, 128, 512, 256), (1, 256, 256, 256, 128)])
def test_shortcut_in_norm_routes_stage_by_stage_on_the_production_kernels(n, hh, ww, cin, cout):
    """The stage-by-stage comparison of `test_resnet_block_stage_by_stage_on_the_production_kernels` for the two ResnetBlocks whose 1 x 1 shortcut rides on the
    GroupNorm passes (csrc/norm_short.hip: 512 -> 256 at 128 x 128 with the weight resident in the accumulation registers, `coop`; 256 -> 128 at 256 x 256 with
    the weight in LDS; flux_ae.py:67,71,77-82): `groupnorm_apply_short` (a1 AND xs from one read of x) and `groupnorm_bwd_short` (dx = GN1' + dy . Ws) against the
    oracle's arithmetic in f64 on the calls' own inputs -- every bf16 result a correct rounding (<= 1 spacing; != RNE(oracle) for <= 0.2 % / 5 % of the elements:
    the shortcut sums run in another f32 order inside one bf16 rounding, tests/test_gpu_norm_short.py), dgamma / dbeta within 1e-5."""
    import torch.nn.functional as F
    from dmvae_amd import ops
    from dmvae_amd.functional import packed
    from conftest import elem_err
    assert ops.groupnorm_short_supported(n, hh * ww, cin, cout)
    g = torch.Generator().manual_seed(cin + hh)
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (rn(n, hh, ww, cin) * 1.3 + 0.2).to(torch.bfloat16)
    dy = rn(n, hh, ww, cout, sc=0.7).to(torch.bfloat16)
    da1 = rn(n, hh, ww, cin, sc=0.5).to(torch.bfloat16)
    P = {"n1w": 1 + 0.3 * rn(cin), "n1b": 0.2 * rn(cin), "sw": rn(cout, cin, 1, 1, sc=0.06), "sb": 0.1 * rn(cout)}
    D = {k: v.to(DEV) for k, v in P.items()}
    xd, dyd, da1d = x.to(DEV), dy.to(DEV), da1.to(DEV)
    nchw = lambda t: t.double().cpu().permute(0, 3, 1, 2)
    wq = P["sw"].to(torch.bfloat16).double()
    st1 = ops.groupnorm_stats(xd)
    a1, xs = ops.groupnorm_apply_short(xd, st1, D["n1w"], D["n1b"], packed(D["sw"]), D["sb"])
    _assert_bf16_of(a1.permute(0, 3, 1, 2), R.swish(R.group_norm(nchw(xd), P["n1w"].double(), P["n1b"].double())), "a1 = swish(GN(x)) [short]")
    _assert_bf16_of(xs.permute(0, 3, 1, 2), F.conv2d(nchw(xd), wq, P["sb"].double()), "xs = nin_shortcut(x) [short]", max_flip_frac=5e-2)
    # the stored-operand route gives the same a1 bits
    assert torch.equal(a1, ops.groupnorm_apply(xd, st1, D["n1w"], D["n1b"], True))
    dx, dn1w, dn1b = ops.groupnorm_bwd_short(da1d, xd, dyd, packed(D["sw"], True), st1, D["n1w"], D["n1b"], True)
    i = nchw(xd).requires_grad_(True)
    gw_, gb_ = P["n1w"].double().requires_grad_(True), P["n1b"].double().requires_grad_(True)
    gi, gw, gb = torch.autograd.grad(R.swish(R.group_norm(i, gw_, gb_)), (i, gw_, gb_), nchw(da1d))
    dxs_exact = F.conv_transpose2d(nchw(dyd), wq)                                       # dy . Ws, the 1 x 1 conv's input gradient
    dxs64 = dxs_exact.to(torch.float32).to(torch.bfloat16).double()                     # rounded to bf16 before it is added (the stored route's tensor)
    # dx = bf16(GN1' + bf16(dy . Ws)): the shortcut term is rounded on its own first, in the kernel from an f32 sum in another order -- where that flips (a few % of the
    # elements) dx moves by one spacing OF THE SHORTCUT TERM, which is several spacings of dx wherever the two terms nearly cancel.  So: every element within one
    # spacing of dx plus one spacing of the shortcut term, and all but 10 % exactly RNE of the oracle's value
    ref = gi + dxs64
    h = dx.permute(0, 3, 1, 2).double().cpu()
    err = (h - ref).abs()
    allowed = _ulp_bf16(ref) + _ulp_bf16(dxs_exact) + 1e-4 * ref.abs().max()
    assert bool((err <= allowed).all()), float((err / allowed).max())
    assert (h != ref.to(torch.float32).to(torch.bfloat16).double()).double().mean().item() < 0.10
    assert rel_err(dn1w.cpu(), gw) < 1e-5 and rel_err(dn1b.cpu(), gb) < 1e-5 and elem_err(dn1w.cpu(), gw) < 1e-4


def test_decoder_tail_stage_by_stage_on_the_production_kernels():
    """norm_out -> swish -> conv_out (flux_ae.py:266-268) at the C2 width on the fused production kernels (`norm_conv_out_fwd` / `norm_conv_out_bwd`: the 128 -> 3
    conv inside the GroupNorm passes) against the oracle in f64 on the same inputs: a = swish(GN(x)) a correct bf16 rounding, the f32 image within 2e-3 of the f64
    conv of the bf16 activation (bf16 operands, f32 accumulation), dx a correct bf16 rounding of GN'(conv_out^T dy), dgamma / dbeta within 1e-4."""
    import torch.nn.functional as F
    from dmvae_amd import ops
    from dmvae_amd.functional import packed
    n, hh, ww, c, cout = 2, 256, 256, 128, 3
    if not (ops.norm_conv_out_fwd_supported(n, hh, ww, c, cout) and ops.norm_conv_out_bwd_supported(n, hh, ww, c, cout)):
        pytest.skip("fused decoder tail not supported at this shape")
    g = torch.Generator().manual_seed(5)
    rn = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    x = (rn(n, hh, ww, c) * 1.2 + 0.1).to(torch.bfloat16)
    dy = rn(n, cout, hh, ww, sc=0.8)
    P = {"nw": 1 + 0.3 * rn(c), "nb": 0.2 * rn(c), "cw": rn(cout, c, 3, 3, sc=0.05), "cb": 0.1 * rn(cout)}
    D = {k: v.to(DEV) for k, v in P.items()}
    xd, dyd = x.to(DEV), dy.to(DEV)
    nchw = lambda t: t.double().cpu().permute(0, 3, 1, 2)
    wq = P["cw"].to(torch.bfloat16).double()
    st = ops.groupnorm_stats(xd)
    a, y = ops.norm_conv_out_fwd(xd, st, D["nw"], D["nb"], packed(D["cw"], False, rows_pad=4), D["cb"], cout)
    a64 = R.swish(R.group_norm(nchw(xd), P["nw"].double(), P["nb"].double()))
    _assert_bf16_of(a.permute(0, 3, 1, 2), a64, "a = swish(norm_out(x)) [tail]")
    y64 = F.conv2d(nchw(a), wq, P["cb"].double(), padding=1)
    assert rel_err(y.cpu(), y64) < 1e-5, rel_err(y.cpu(), y64)
    dx, dnw, dnb = ops.norm_conv_out_bwd(dyd, D["cw"], xd, st, D["nw"], D["nb"])
    da64 = F.conv_transpose2d(dy.double(), wq, padding=1)                               # conv_out's input gradient, never materialised by the kernel
    i = nchw(xd).requires_grad_(True)
    gw_, gb_ = P["nw"].double().requires_grad_(True), P["nb"].double().requires_grad_(True)
    gi, gw, gb = torch.autograd.grad(R.swish(R.group_norm(i, gw_, gb_)), (i, gw_, gb_), da64)
    # the kernel forms d a from bf16 operands (dy rounded to bf16 at the matrix cores) in f32: dx is a bf16 rounding of a value within that operand noise of the oracle's
    e = ((dx.permute(0, 3, 1, 2).double().cpu() - gi).norm() / gi.norm()).item()
    assert e < 6e-3, e
    assert rel_err(dnw.cpu(), gw) < 5e-3 and rel_err(dnb.cpu(), gb) < 5e-3
