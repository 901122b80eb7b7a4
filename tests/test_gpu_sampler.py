"""Downstream consumers on the MI355X (-m gpu): the fused SDE state update and the uint8 conversion kernel (csrc/sampler.hip) bit-exact
against the CPU oracle, the sampler on the HIP LightningDiT path against the trajectories captured from the reference, decode-to-uint8, the
sample_50k loop and the diffusion trainer's step."""
import copy
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from test_oracle_sampler import CASES, DIT_KW, _kw, small_dit

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def _rl2(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("form,norm,tval", [("sigma", 1.0, 0.0), ("sigma", 1.0, 0.5139), ("linear", 0.7, 0.9599), ("decreasing", 1.0, 0.25),
                                            ("inccreasing-decreasing", 1.3, 0.77), ("SBDM", 1.0, 0.3)])
@pytest.mark.parametrize("vdtype", [torch.float32, BF])
def test_sde_euler_step_kernel_bit_exact(form, norm, tval, vdtype):
    """ops.sde_euler_step == the reference's f32 elementwise graph (oracle.sde_euler_step on the CPU) for the same model output, to the bit."""
    from dmvae_amd import ops
    from dmvae_amd.transport import ICPlan
    g = torch.Generator().manual_seed(int(tval * 1e4) + len(form))
    shape = (5, 8, 6, 6)
    x, w = torch.randn(shape, generator=g) * 1.7, torch.randn(shape, generator=g)
    v = (torch.randn(shape, generator=g) * 2).to(vdtype)
    t = torch.tensor(tval, dtype=torch.float32)
    dt = torch.linspace(0, 0.96, 250)[1] - torch.linspace(0, 0.96, 250)[0]
    tv = torch.ones(shape[0]) * t
    x_ref, mean_ref = R.sde_euler_step(x, v.float(), w, tv, dt, form, norm)
    ps = ICPlan()
    te = t.view(1, 1)
    rar, var = ps._score_coeffs(te)
    diff = ps.compute_diffusion(te, te.view(1), form=form, norm=norm)
    out, mean = ops.sde_euler_step(x.to(DEV), v.to(DEV), w.to(DEV), float(rar), float(var), float(diff), float(dt), float(torch.sqrt(2 * diff)),
                                   float(torch.sqrt(dt)), need_mean=True)
    assert torch.equal(mean.cpu(), mean_ref) and torch.equal(out.cpu(), x_ref)
    # last step ("Mean", transport.py:283-286): x + drift * last_step_size, no noise
    last = ops.sde_euler_step(x.to(DEV), v.to(DEV), None, float(rar), float(var), float(diff), 0.04, 0.0, 0.0)[0]
    assert torch.equal(last.cpu(), x + R.sde_drift_from_velocity(v.float(), x, tv, form, norm) * 0.04)


def test_image_to_u8_kernel_bit_exact_vs_reference_fixture():
    from dmvae_amd import ops
    g = load_golden("image_u8")
    s = g.t("s")                                                       # NCHW
    y4 = torch.zeros(*s.permute(0, 2, 3, 1).shape[:3], 4)
    y4[..., :3] = s.permute(0, 2, 3, 1)
    y4[..., 3] = 1e9                                                   # the padding channel is never read
    assert np.array_equal(ops.image_to_u8(y4.to(DEV), 3).cpu().numpy(), np.asarray(g["u8"]))
    assert np.array_equal(ops.image_to_u8(y4.to(DEV), 3, round_bf16=True).cpu().numpy(), np.asarray(g["u8_bf16"]))
    big = (torch.rand(3, 64, 64, 4, generator=torch.Generator().manual_seed(1)) * 3 - 1.5)
    assert np.array_equal(ops.image_to_u8(big.to(DEV), 3).cpu().numpy(), R.image_to_uint8(big[..., :3].permute(0, 3, 1, 2)).numpy())


@pytest.mark.parametrize("tag", CASES)
def test_sampler_on_hip_dit_vs_reference_trajectory(tag):
    """`transport.Sampler.sample_sde` with LightningDiT on the HIP kernels under autocast(bf16) -- the fused state update for Euler -- against the f32
    trajectory captured from the reference (same noise stream): as close as the stock modules under autocast, and bit-identical to the same sampler
    composed of tensor ops (the fused kernel changes no bit)."""
    from dmvae_amd import transport as T
    g = load_golden(tag)
    kw = _kw(g)
    m = small_dit(g["dit_seed"]).to(DEV)
    z, y, ref = g.t("z").to(DEV), torch.from_numpy(np.asarray(g["y"])).to(DEV), g.t("xs")
    sampler = T.Sampler(T.create_transport("Linear", "velocity", None, None, None, time_dist_shift=2.5))

    def run(model_fn, fused=True):
        fn = sampler.sample_sde(**kw)
        T.FUSED_STATE_UPDATE = fused
        try:
            torch.manual_seed(int(g["seed"]))
            with torch.no_grad(), torch.autocast("cuda", dtype=BF):
                return torch.stack(fn(z, model_fn, y=y)).float().cpu()
        finally:
            T.FUSED_STATE_UPDATE = True

    hip = run(m.forward)
    stock = run(m.forward_stock)
    e_hip, e_stock = rel_err(hip, ref), rel_err(stock, ref)
    assert e_hip < max(2 * e_stock, 2e-2), (e_hip, e_stock)
    assert rel_err(hip[-1], ref[-1]) < max(2 * rel_err(stock[-1], ref[-1]), 2e-2)
    if kw["sampling_method"] == "Euler":
        plain = run(m.forward, fused=False)
        if kw["diffusion_form"] in ("sigma", "linear"):
            assert torch.equal(hip[:-1], plain[:-1])                   # every Euler-Maruyama state
            if kw["last_step"] in ("Mean", None):
                assert torch.equal(hip[-1], plain[-1])
        else:       # cos / sin of the diffusion coefficient: evaluated on the host (= the CPU reference's libm) in the fused route, by the device's cosf / sinf in the composed one
            assert rel_err(hip, plain) < 5e-3


def test_graphed_dit_forward_replays_bit_exactly():
    """lightningdit_fast.GraphedInference (the frozen DiT forward captured in a hipGraph, one replay per sampler step) returns exactly what the kernel-by-kernel
    forward returns, for new inputs each call; SamplePipeline produces the same latents with and without it."""
    from dmvae_amd.models import lightningdit_fast as fast
    from dmvae_amd.models.lightningdit import LightningDiT
    from dmvae_amd.sample import SamplePipeline
    torch.manual_seed(8)
    dit = LightningDiT(input_size=16, patch_size=1, in_channels=32, hidden_size=144, depth=3, num_heads=2, num_classes=10).to(DEV).eval().requires_grad_(False)
    with torch.no_grad():
        for blk in dit.blocks:
            blk.adaLN_modulation[1].weight.normal_(0, 0.02)
        dit.final_layer.linear.weight.normal_(0, 0.05)
    g = torch.Generator(device=DEV).manual_seed(1)
    mk = lambda: (torch.randn(5, 32, 16, 16, device=DEV, generator=g), torch.rand(5, device=DEV, generator=g), torch.randint(0, 11, (5,), device=DEV, generator=g))
    with torch.no_grad(), torch.autocast("cuda", dtype=BF):
        gi = fast.GraphedInference(dit, *mk())
        for _ in range(3):
            x, t, y = mk()
            want = dit(x, t, y).clone()
            got = gi(x, t, y)
            assert torch.equal(got, want)
        with pytest.raises(AssertionError):
            gi(x[:2], t[:2], y[:2])
    z, y = torch.randn(5, 32, 16, 16, device=DEV, generator=g), torch.tensor([0, 3, 5, 7, 9], device=DEV)
    outs = []
    for use_graph in (True, False):
        pipe = SamplePipeline(dit, None, num_sampling_steps=7, latent_mean=0.0685, latent_scale=0.1763, use_graph=use_graph)
        torch.manual_seed(77)
        outs.append(pipe.latents(z, y))
        assert (pipe._graphed is not None) == use_graph
    assert torch.equal(outs[0], outs[1])


def test_decode_to_uint8_matches_decode_then_convert():
    from dmvae_amd.models.vae import VAE
    torch.manual_seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).to(DEV).eval()
    tok = torch.randn(2, 256, 32, device=DEV) * 0.7
    with torch.autocast("cuda", dtype=BF):
        img = vae.decode(tok).float()
        u8 = vae.decode_uint8(tok, round_bf16=False)
        u8b = vae.decode_uint8(tok, round_bf16=True)
    assert u8.shape == (2, 256, 256, 3) and u8.dtype == torch.uint8
    assert torch.equal(u8.cpu(), R.image_to_uint8(img.cpu()))
    assert torch.equal(u8b.cpu(), R.image_to_uint8(img.cpu().to(BF).float()))
    assert 0 < u8.float().std() and u8.min() < 100 and u8.max() > 150     # not a degenerate image


def test_sample_50k_loop_writes_the_reference_file_layout(tmp_path):
    from PIL import Image
    from dmvae_amd.models.lightningdit import LightningDiT
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.sample import SamplePipeline, labels_and_indices
    torch.manual_seed(4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).to(DEV).eval()
    dit = LightningDiT(input_size=16, patch_size=1, in_channels=32, hidden_size=192, depth=2, num_heads=3, num_classes=10).to(DEV).eval()
    with torch.no_grad():
        for blk in dit.blocks:
            blk.adaLN_modulation[1].weight.normal_(0, 0.02)
        dit.final_layer.linear.weight.normal_(0, 0.05)
    pipe = SamplePipeline(dit, vae, num_sampling_steps=6, latent_mean=0.0685, latent_scale=0.1763, time_dist_shift=2.5)
    for rank in (0, 1):
        n = pipe.run(str(tmp_path), per_proc_batch_size=5, num_fid_samples=40, num_classes=10, rank=rank, world_size=2, max_iterations=2)
        assert n == 10
    names = sorted(os.listdir(tmp_path))
    want = sorted(f"{i:06d}.png" for r in (0, 1) for row in labels_and_indices(40, 10, 2, r, 5)[1][:2] for i in row)
    assert names == want and names[0] == "000010.png"
    # one batch again with the same seeds: the PNG holds exactly the uint8 image of the pipeline
    torch.manual_seed(9)
    z = torch.randn(5, 32, 16, 16, device=DEV)
    y = torch.tensor([0, 1, 2, 3, 4], device=DEV)
    torch.manual_seed(10)
    u8, tok = pipe.images_uint8(z, y)
    torch.manual_seed(10)
    u8b, _ = pipe.images_uint8(z, y)
    assert torch.equal(u8, u8b) and tok.shape == (5, 256, 32) and torch.isfinite(tok).all()
    p = tmp_path / "x.png"
    Image.fromarray(u8[0].cpu().numpy()).save(p)
    assert np.array_equal(np.asarray(Image.open(p)), u8[0].cpu().numpy())


def test_diffusion_trainer_step_vs_stock_autocast_step():
    """train_diffusion.py's step (frozen encode -> latents -> flow-matching loss -> clip -> AdamW -> EMA) on the HIP path against the same step written with
    the stock modules, torch.optim.AdamW, clip_grad_norm_ and the reference's update_ema; same random draws.  Forward / backward: loss and gradient norm
    of the stock route at the same weights.  Optimiser tail: the stock optimiser fed the HIP path's gradients must land on the same weights (Adam's
    g / sqrt(v) turns bf16 noise on near-zero gradients into full-size steps, so two independent bf16 runs cannot be compared weight by weight)."""
    from dmvae_amd.models.lightningdit import LightningDiT
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import DiffusionTrainer
    torch.manual_seed(6)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).to(DEV).eval()
    dit = LightningDiT(input_size=16, patch_size=1, in_channels=32, hidden_size=192, depth=2, num_heads=3, num_classes=10).to(DEV)
    with torch.no_grad():
        for blk in dit.blocks:
            blk.adaLN_modulation[1].weight.normal_(0, 0.02)
        dit.final_layer.linear.weight.normal_(0, 0.05)
    ref_m = copy.deepcopy(dit)
    ema = copy.deepcopy(dit).eval().requires_grad_(False)
    tr = DiffusionTrainer(dit, vae, lr=1e-3, latent_mean=0.05, latent_scale=0.8)
    names = {id(p): n for n, p in dit.named_parameters()}
    ref_p = dict(ref_m.named_parameters())
    opt = torch.optim.AdamW([ref_p[names[id(p)]] for p in tr.fp.params], lr=1e-3, betas=(0.9, 0.95), weight_decay=0)
    g = torch.Generator(device=DEV).manual_seed(0)
    images = torch.rand(4, 3, 256, 256, device=DEV, generator=g) * 2 - 1
    labels = torch.tensor([3, 7, 1, 9], device=DEV)
    with torch.autocast("cuda", dtype=BF):              # train_diffusion.py:276-287 runs the frozen encode under autocast: the HIP encoder route
        x = tr.latents(images)
    assert x.shape == (4, 32, 16, 16) and not x.requires_grad
    for step in range(3):
        ema_before = tr.fp.ema.clone()
        # the stock forward / backward at the current weights (train_diffusion.py:288-293)
        ref_m.train()
        torch.manual_seed(100 + step)
        with torch.autocast("cuda", dtype=BF):
            _, terms = tr.transport.training_losses(ref_m.forward_stock, x, dict(y=labels))
        rloss = terms["loss"].mean().float()
        opt.zero_grad()
        rloss.backward()
        rnorm = torch.nn.utils.clip_grad_norm_(ref_m.parameters(), 1.0)
        torch.manual_seed(100 + step)
        loss = tr.step(images, labels)
        log = tr.read_log()
        assert abs(loss.item() - log["loss"]) < 1e-6
        assert abs(log["loss"] - rloss.item()) < 2e-2 * abs(rloss.item()), (step, log, rloss.item())
        assert abs(log["grad_norm"] - rnorm.item()) < 5e-2 * rnorm.item(), (step, log, rnorm.item())
        # the stock tail (:294-297) on the HIP path's gradients
        for p_, off in zip(tr.fp.params, tr.fp.offsets):
            ref_p[names[id(p_)]].grad = tr.fp.grad[off:off + p_.numel()].view(p_.shape).clone()
        torch.nn.utils.clip_grad_norm_([ref_p[names[id(p_)]] for p_ in tr.fp.params], 1.0)
        opt.step()
        with torch.no_grad():
            for (n_, pe), (_, pm) in zip(ema.named_parameters(), ref_m.named_parameters()):
                pe.mul_(0.9999).add_(pm.data, alpha=1 - 0.9999)
        for p_, e_ in zip(tr.fp.params, tr.fp.ema_state()):
            n_ = names[id(p_)]
            assert torch.allclose(p_.detach(), ref_p[n_].detach(), rtol=1e-5, atol=2e-6), (step, n_)
            assert torch.allclose(e_, dict(ema.named_parameters())[n_], rtol=1e-5, atol=1e-6), (step, n_)
        assert torch.allclose(tr.fp.ema, ema_before * 0.9999 + tr.fp.flat * (1 - 0.9999), rtol=1e-6, atol=1e-7)      # update_ema's recurrence
    sd = tr.ema_state_dict()
    assert list(sd.keys()) == list(dit.state_dict().keys()) and torch.equal(sd["pos_embed"], dit.pos_embed)
    assert not torch.equal(sd["blocks.1.mlp.w12.weight"], dit.state_dict()["blocks.1.mlp.w12.weight"])


def test_diffusion_trainer_vs_reference_capture_c4():
    """Config C4's step pinned to the REFERENCE: `DiffusionTrainer` (frozen encode on the bf16 encoder kernels -> latents -> transport.training_losses with
    LightningDiT's HIP training route in train mode -> clip -> fused AdamW + EMA) replays the two steps oracle/capture_golden_diffusion.py recorded from
    train_diffusion.py:268-297 run with the reference's own modules, with the capture's draws injected (t, x0, dropped labels).  Bars: the bf16-site criterion --
    as close to the reference's f32 numbers as the CPU oracle with bf16 rounding at the autocast sites (oracle.ref_cpu.diffusion_train_steps(q=bf16_round)) is,
    x 1.15 + a floor -- for the latents, the first step's loss and the nine fully captured gradients; the gradient norms of both steps; and the optimiser
    tail exactly: torch.optim.AdamW + clip_grad_norm_ + update_ema fed the HIP path's own gradients must land on the HIP path's weights / EMA (two bf16
    pipelines cannot be compared weight by weight through Adam's g / sqrt(v): tests/test_oracle_diffusion.py holds the f32 oracle to the capture's updates)."""
    from conftest import load_golden
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import DiffusionTrainer
    from oracle import ref_cpu as R
    from test_oracle_diffusion import DIT_KW as KW, SMALL, diffusion_step_inputs
    g = load_golden("diffusion_step_small")
    pv, dit, images, labels, draws = diffusion_step_inputs(g)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=2, num_heads=4))
    vae.load_state_dict(pv, strict=True)
    vae, dit = vae.to(DEV).eval(), dit.to(DEV)
    tr = DiffusionTrainer(dit, vae, lr=float(g["lr"]), latent_mean=float(g["latent_mean"]), latent_scale=float(g["latent_scale"]))
    names = {id(p): n for n, p in dit.named_parameters()}
    # ---- the frozen encode + normalisation (train_diffusion.py:276-287) ----
    with torch.autocast("cuda", dtype=BF):
        x = tr.latents(images.to(DEV))
    p_cpu = {k: v.detach().cpu().clone() for k, v in dit.state_dict().items()}
    with torch.no_grad():
        tok_q = R.mlp_forward(R.dino_encoder_forward(images, pv, num_heads=4, q=R.bf16_round), pv, q=R.bf16_round)
    x_q = R.latents_to_dit_input(tok_q, float(g["latent_mean"]), float(g["latent_scale"]))
    e_hip, e_orc = _rl2(x.float().cpu(), g.t("latents")), _rl2(x_q, g.t("latents"))
    print(f"C4 latents: rel-L2 to the f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
    assert e_hip < 1.15 * e_orc + 1e-3
    # ---- the bf16-site oracle's two steps on the reference's latents (what "as close as bf16 allows" means for loss and gradients) ----
    trainable = [n for n in (str(s) for s in g["names"]) if n != "pos_embed"]
    og = {}
    ologs, _, _ = R.diffusion_train_steps(g.t("latents"), labels, p_cpu, trainable, draws, KW["num_heads"], KW["num_classes"], lr=float(g["lr"]), q=R.bf16_round,
                                          on_grads=lambda s, gr: og.update({k: v.clone() for k, v in gr.items()}) if s == 0 else None)
    # ---- the HIP steps with the capture's draws injected ----
    ref_m = copy.deepcopy(dit)
    ref_p = dict(ref_m.named_parameters())
    ema = copy.deepcopy(dit).eval().requires_grad_(False)
    opt = torch.optim.AdamW([ref_p[names[id(p)]] for p in tr.fp.params], lr=float(g["lr"]), betas=(0.9, 0.95), weight_decay=0)
    orig_sample, orig_drop = tr.transport.sample, dit.y_embedder.token_drop
    for step, (t, x0, dropped) in enumerate(draws):
        tr.transport.sample = lambda x1, t=t, x0=x0: (t.to(x1), x0.to(x1), x1)
        dit.y_embedder.token_drop = lambda lab, force_drop_ids=None, d=dropped: torch.where(d.to(lab.device), torch.full_like(lab, KW["num_classes"]), lab)
        loss = tr.step(images.to(DEV), labels.to(DEV))
        log = tr.read_log()
        want_loss, want_norm = float(g["loss"][step]), float(g["grad_norm"][step])
        o_loss, o_norm = ologs[step] if step == 0 else (None, None)
        if step == 0:      # same weights on all three sides: the oracle's distance is the bar
            assert abs(log["loss"] - want_loss) < 1.15 * abs(o_loss - want_loss) + 2e-3 * want_loss, (log, want_loss, o_loss)
            assert abs(log["grad_norm"] - want_norm) < 1.15 * abs(o_norm - want_norm) + 1e-2 * want_norm, (log, want_norm, o_norm)
            grads = {names[id(p)]: tr.fp.grad[off:off + p.numel()].view(p.shape).float().cpu() for p, off in zip(tr.fp.params, tr.fp.offsets)}
            for k in SMALL:
                e_hip, e_orc = _rl2(grads[k], g.t("g0." + k)), _rl2(og[k], g.t("g0." + k))
                print(f"C4 step 0 grad {k}: rel-L2 to the f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
                assert e_hip < 1.15 * e_orc + 2e-3, (k, e_hip, e_orc)
            for k in trainable:
                want = float(g["gn0." + k][0])
                if want > 1e-4:
                    e_hip = abs(grads[k].double().norm().item() - want) / want
                    e_orc = abs(og[k].double().norm().item() - want) / want
                    assert e_hip < 1.15 * e_orc + 2e-2, (k, e_hip, e_orc)
        else:              # the weights have moved by one bf16-noisy Adam step on each side: the trajectory, loosely
            assert abs(log["loss"] - want_loss) < 2e-2 * want_loss and abs(log["grad_norm"] - want_norm) < 8e-2 * want_norm, (log, want_loss, want_norm)
        assert abs(loss.item() - log["loss"]) < 1e-6
        # the reference's tail (:293-297) on the HIP path's gradients
        for p_, off in zip(tr.fp.params, tr.fp.offsets):
            ref_p[names[id(p_)]].grad = tr.fp.grad[off:off + p_.numel()].view(p_.shape).clone()
        torch.nn.utils.clip_grad_norm_([ref_p[names[id(p_)]] for p_ in tr.fp.params], 1.0)
        opt.step()
        with torch.no_grad():
            for (_, pe), (_, pm) in zip(ema.named_parameters(), ref_m.named_parameters()):
                pe.mul_(0.9999).add_(pm.data, alpha=1 - 0.9999)
        for p_, e_ in zip(tr.fp.params, tr.fp.ema_state()):
            n_ = names[id(p_)]
            assert torch.allclose(p_.detach(), ref_p[n_].detach(), rtol=1e-5, atol=2e-6), (step, n_)
            assert torch.allclose(e_, dict(ema.named_parameters())[n_], rtol=1e-5, atol=1e-6), (step, n_)
    tr.transport.sample, dit.y_embedder.token_drop = orig_sample, orig_drop
