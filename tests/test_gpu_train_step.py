"""Step-harness checks on the GPU: direct flat-buffer gradients == autograd-accumulated gradients (bit-exact), the step
is deterministic, and the loss goes down."""
import copy
import os
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def _trainer(direct, seed=3, with_lpips=True):
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import TokenizerTrainer
    from dmvae_amd.utils.lpips import LPIPS
    torch.manual_seed(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).cuda()
    lp = LPIPS().eval().requires_grad_(False).cuda()
    with torch.no_grad():
        for lin in (lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4):
            lin.model[-1].weight.fill_(1.0 / lin.model[-1].weight.shape[1])
    tr = TokenizerTrainer(vae, lp if with_lpips else None, warmup_steps=2)
    if not direct:
        tr.fp.direct = False
        for p, off in zip(tr.fp.params, tr.fp.offsets):
            del p._dmvae_grad_view
            p.grad = tr.fp.grad[off:off + p.numel()].view(p.shape)
    return tr


def test_direct_flat_grads_match_autograd_accumulation():
    g = torch.Generator(device="cuda").manual_seed(0)
    images = torch.rand(2, 3, 256, 256, device="cuda", generator=g) * 2 - 1
    # without the LPIPS trunk every kernel on the path is ours and deterministic: bit-exact equality
    # (the stock MIOpen VGG backward is not run-to-run deterministic, so the LPIPS variant is compared to tolerance)
    a, b = _trainer(True, with_lpips=False), _trainer(False, with_lpips=False)
    assert torch.equal(a.fp.flat, b.fp.flat)
    for _ in range(3):
        la, lb = a.step(images), b.step(images)
        assert torch.equal(la, lb)
        assert torch.equal(a.fp.grad, b.fp.grad)
        assert torch.equal(a.fp.flat, b.fp.flat) and torch.equal(a.fp.ema, b.fp.ema)
    assert all(p.grad is not None for p in a.fp.params)
    c, d = _trainer(True), _trainer(False)
    losses = []
    for _ in range(6):          # warm-up of 2: the reference's schedule runs step 0 at lr 0, step 1 at half the rate
        lc, ld = c.step(images), d.step(images)
        assert abs(lc.item() - ld.item()) < 1e-3 * abs(ld.item())
        losses.append(lc.item())
    assert (c.fp.flat - d.fp.flat).abs().max().item() < 2e-3
    assert losses[-1] < losses[0]
    log = c.read_log()
    assert log["L1"] > 0 and log["LPIPS"] > 0 and log["vae_norm"] > 0


def test_frozen_vit_fast_path_matches_stock_module():
    """csrc/vit.hip kernels (LayerNorm -> bf16, LayerScale + residual) inside the frozen-encoder forward vs the stock module under
    autocast, and each kernel vs its fp64 definition."""
    from dmvae_amd import ops
    from dmvae_amd.models.vit import DinoV2ViT
    from dmvae_amd.models.vit_fast import frozen_forward_features
    from dmvae_amd.train import frozen_bf16_shadow
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(37, 1024, generator=g) * 2 + 0.5).cuda()
    gam, bet = torch.randn(1024, generator=g).cuda(), torch.randn(1024, generator=g).cuda()
    y = ops.layernorm_bf16(x, gam, bet, 1e-6)
    ref = torch.nn.functional.layer_norm(x.double(), (1024,), gam.double(), bet.double(), 1e-6)
    assert torch.equal(y, ref.float().to(torch.bfloat16)) or (y.float() - ref.float()).abs().max() < 2e-2
    assert ((y.double() - ref).abs() / (ref.abs() + 1)).max() < 5e-3
    r = torch.randn(37, 1024, generator=g).cuda()
    yb = torch.randn(37, 1024, generator=g).cuda().to(torch.bfloat16)
    want = r.double() + gam.double() * yb.double()
    ops.scale_residual_(r, yb, gam)
    assert (r.double() - want).abs().max() < 1e-5
    torch.manual_seed(0)
    vit = DinoV2ViT(embed_dim=256, depth=2, num_heads=4, patch_size=16, img_size=64).cuda().eval()
    with torch.no_grad():
        for blk in vit.blocks:
            blk.ls1.gamma.fill_(0.5); blk.ls2.gamma.fill_(0.5)
    img = torch.randn(3, 3, 64, 64, generator=g).cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        want = vit.forward_features_stock(img).float()
    got = frozen_forward_features(frozen_bf16_shadow(vit), img).float()
    assert ((got - want).norm() / want.norm()).item() < 2e-2
    # the module itself takes the same route when nothing needs a gradient under autocast(bf16) -- `vae.encode`, `vae(x, freeze_encoder=True)` -- on cached bf16
    # copies of its f32 weights: identical to the shadow-module route, and refreshed when a weight changes
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        via_module = vit.forward_features(img)
    assert via_module.dtype == torch.bfloat16 and torch.equal(via_module.float(), got)
    with torch.no_grad():
        vit.blocks[1].mlp.fc2.weight.mul_(1.25)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        again = vit.forward_features(img)
        want2 = vit.forward_features_stock(img).float()
    assert not torch.equal(again, via_module) and ((again.float() - want2).norm() / want2.norm()).item() < 2e-2
    # no autocast: the bf16 HIP route does not apply and nothing falls back silently -- it raises (f32 arithmetic on the HIP kernels: DMVAE_PARITY=1)
    from dmvae_amd._lib import DmvaeHipError
    with torch.no_grad(), pytest.raises(DmvaeHipError, match="DMVAE_ALLOW_STOCK"):
        vit.forward_features(img)


@pytest.mark.parametrize("shape", [(2, 257, 16), (3, 65, 4), (1, 288, 2), (2, 17, 3)])
def test_fused_encoder_attention_vs_reference(shape):
    from dmvae_amd import ops
    b, s, h = shape
    c = h * 64
    g = torch.Generator().manual_seed(s)
    qkv = (torch.randn(b, s, 3 * c, generator=g) * 1.5).cuda().to(torch.bfloat16)
    out = ops.attention_qkv(qkv, h, 64 ** -0.5)
    t = qkv.double().reshape(b, s, 3, h, 64).permute(2, 0, 3, 1, 4)
    att = torch.softmax(t[0] @ t[1].transpose(-2, -1) * 64 ** -0.5, dim=-1)
    ref = (att @ t[2]).transpose(1, 2).reshape(b, s, c)
    err = (out.double() - ref).abs().max().item()
    assert err < 2e-2 * ref.abs().max().item() + 1e-3, err     # P and O are rounded to bf16 once each
    assert torch.equal(out, ops.attention_qkv(qkv, h, 64 ** -0.5))


def test_step_with_discriminator_branch():
    """Steady-state step of train_tokenizer.py (global_step >= disc_start_step): generator term with the adaptive weight, then the
    discriminator update.  Checks the wiring -- the branch switches on at disc_start_step, both optimisers move their parameters, the
    discriminator is left in train mode with trainable parameters, BatchNorm counters advance by two per step, the log entries are
    finite and the checkpoint carries disc_wo_ddp -- and run-to-run determinism given the same RNG seeds."""
    from dmvae_amd.models.init_param import init_weights
    from dmvae_amd.models.patchgan import NLayerDiscriminator

    def run():
        tr = _trainer(True)
        torch.manual_seed(11)
        disc = NLayerDiscriminator()
        init_weights(disc, 0.02)
        from dmvae_amd.train import TokenizerTrainer
        tr = TokenizerTrainer(tr.vae, tr.lpips, warmup_steps=2, disc=disc.cuda(), disc_start_step=1, disc_weight=0.5)
        g = torch.Generator(device="cuda").manual_seed(0)
        images = torch.rand(2, 3, 256, 256, device="cuda", generator=g) * 2 - 1
        d0 = tr.dfp.flat.clone()
        tr.step(images)                                   # step 0: branch off
        assert torch.equal(tr.dfp.flat, d0) and int(disc.main[3].num_batches_tracked) == 0 and tr.read_log()["d_weight"] == 0.0
        torch.manual_seed(5)
        out = []
        for _ in range(3):                                # steps 1..3: branch on (lr: 0.5x at step 1 of the warm-up, then 1x)
            out.append(tr.step(images).item())
        return tr, disc, d0, out

    tr, disc, d0, out = run()
    log, dlog = tr.read_log(), tr.read_disc_log()
    assert all(map(lambda v: v == v and abs(v) < 1e6, list(log.values()) + list(dlog.values())))
    assert log["d_weight"] > 0 and dlog["d_loss"] > 0 and dlog["disc_norm"] > 0 and 0 <= dlog["acc_mean"] <= 100
    assert not torch.equal(tr.dfp.flat, d0)                                    # the discriminator optimiser moved its parameters
    assert disc.training and all(p.requires_grad for p in disc.parameters())
    assert int(disc.main[3].num_batches_tracked) == 6                           # two training-mode passes per discriminator step
    ck = tr.checkpoint()
    assert set(ck["disc_wo_ddp"].keys()) == set(disc.state_dict().keys()) and "vae_ema" in ck
    tr2, _, _, out2 = run()
    assert out == out2 and torch.equal(tr.fp.flat, tr2.fp.flat) and torch.equal(tr.dfp.flat, tr2.dfp.flat)


class _TinyVelocity(torch.nn.Module):
    """Stand-in for the reference's LightningDiT (f(xt, t, labels) -> velocity): per-pixel MLP over channels with a label / time embedding."""

    def __init__(self, c=32, ncls=10):
        super().__init__()
        self.emb = torch.nn.Embedding(ncls + 1, c)
        self.f1, self.f2 = torch.nn.Linear(c, 64), torch.nn.Linear(64, c)

    def forward(self, xt, t, y):
        h = xt.permute(0, 2, 3, 1) + self.emb(y)[:, None, None, :] * t.view(-1, 1, 1, 1).to(xt.dtype)
        return self.f2(torch.nn.functional.silu(self.f1(h))).permute(0, 3, 1, 2)


def test_dmd_stage_step_harness():
    """train_dmd.py's step structure on the HIP path with a pluggable velocity model: VAE turn every `vae_train_every` steps (whole VAE trainable,
    encoder included: gradients through functional.VitBlockFn; DMD loss with CFG through losses.dmd_loss), student turn every step; turn pattern,
    finiteness, every parameter group moves, run-to-run determinism under fixed seeds."""
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import DMDTrainer
    from dmvae_amd.utils.lpips import LPIPS

    def run(direct_grads=True):
        torch.manual_seed(21)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).cuda()
        with torch.no_grad():
            vae.encoder.model.blocks[0].ls1.gamma.fill_(1.0); vae.encoder.model.blocks[0].ls2.gamma.fill_(1.0)
        lp = LPIPS().eval().requires_grad_(False).cuda()
        with torch.no_grad():
            for lin in (lp.lin0, lp.lin1, lp.lin2, lp.lin3, lp.lin4):
                lin.model[-1].weight.fill_(1.0 / lin.model[-1].weight.shape[1])
        teacher, student = _TinyVelocity().cuda().requires_grad_(False), _TinyVelocity().cuda()
        tr = DMDTrainer(vae, lp, teacher, student, dmd_weight=5.0, dmd_cfg_scale=2.0, num_classes=10, vae_train_every=2, warmup_steps=2, direct_grads=direct_grads)
        g = torch.Generator(device="cuda").manual_seed(0)
        images = torch.rand(2, 3, 256, 256, device="cuda", generator=g) * 2 - 1
        labels = torch.tensor([3, 7], device="cuda")
        torch.manual_seed(5)
        snaps = []
        for _ in range(5):
            tr.step(images, labels)
            tr.wait_optimizers()                       # the two updates run on side streams (optim.FlatAdamWEMA.enable_overlap): reading the buffers comes after them
            snaps.append((tr.fp.flat.clone(), torch.cat([p.detach().flatten() for p in student.parameters()]).clone()))
        return tr, snaps

    tr, snaps = run()
    log = tr.read_log()
    assert all(v == v and abs(v) < 1e6 for v in log.values()), log
    assert log["dmd_loss"] > 0 and log["dmd_gradient_norm"] > 0 and log["diffusion_loss"] > 0 and log["vae_norm"] > 0
    # lr warm-up: step 0 runs at lr 0 for both optimisers; VAE turns are steps 0, 2, 4; the student trains every step
    vae_moved = [not torch.equal(snaps[i][0], snaps[i - 1][0]) for i in range(1, 5)]
    stu_moved = [not torch.equal(snaps[i][1], snaps[i - 1][1]) for i in range(1, 5)]
    assert vae_moved == [False, True, False, True], vae_moved
    assert stu_moved == [True, True, True, True], stu_moved
    enc_w = tr.vae.encoder.model.blocks[0].attn.qkv.weight
    assert enc_w.grad is not None and enc_w.grad.abs().max() > 0                       # the encoder trains in this stage
    tr2, snaps2 = run()
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(snaps, snaps2))
    # direct gradient writes (decoder, bottleneck, encoder blocks) change no bit against plain autograd accumulation into the zeroed flat buffer
    tr3, snaps3 = run(direct_grads=False)
    assert getattr(tr, "fp").direct and not getattr(tr3.fp, "direct", False)
    assert all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(snaps, snaps3))
    # the bf16 shadows the fused optimiser step maintains are what a fresh conversion would give, and functional._bf serves them
    from dmvae_amd import functional as Fn
    tr.wait_optimizers()
    assert torch.equal(tr.fp.shadow, tr.fp.flat.to(torch.bfloat16))
    w = tr.vae.encoder.model.blocks[0].attn.qkv.weight
    assert Fn._bf(w).data_ptr() == w._dmvae_shadow.data_ptr() and torch.equal(Fn._bf(w), w.detach().to(torch.bfloat16))
    with torch.no_grad():
        w.mul_(1.5)                                      # changed behind the optimiser's back: converted again on next use
    assert torch.equal(Fn._bf(w), w.detach().to(torch.bfloat16))


def test_step_small_vs_reference_capture():
    """G12: four steps of TokenizerTrainer ENTIRELY on the HIP path -- the encoder too: the capture's reduced ViT is 256 wide (4 heads of
    64 channels, oracle/capture_golden_step.py --width 256), inside the bf16 encoder kernels' range, and no stock-module route is allowed -- against four
    steps of the REFERENCE's own modules and optimiser (tests/golden/step_small_w256.npz, captured in fp32 on the CPU; tests/test_oracle_step.py holds the
    oracle to the same fixture at f32 tolerances, tests/test_gpu_parity_fp32.py the fp32 parity mode at 1e-4).  Same name-seeded weights, same two images,
    same lr schedule.  The HIP path computes in bf16 where the reference's CUDA autocast would, the capture is fp32, so the bars are the bf16 floor of a
    60-layer forward + backward: losses 2 %, gradient norm 5 %; the optimiser tail is then checked through what it did to the weights -- per-tensor
    sum|update| (lr x Adam direction + decay) for all 140 trainable tensors, and the direction of the complete update of eight small tensors."""
    from conftest import load_golden
    from test_oracle_golden import lpips_params
    from test_oracle_step import check_step_small, step_small_inputs
    from dmvae_amd.train import TokenizerTrainer
    from dmvae_amd.utils.lpips import LPIPS
    g = load_golden("step_small_w256")
    p, vae, names, images = step_small_inputs(g)
    vae.load_state_dict(p, strict=True)
    vae = vae.cuda()
    lp = LPIPS().eval().requires_grad_(False)
    missing = lp.load_state_dict(lpips_params(g, "lp."), strict=False)
    assert not missing.unexpected_keys and all("scaling_layer" in k for k in missing.missing_keys), missing
    tr = TokenizerTrainer(vae, lp.cuda(), lr=float(g["base_lr"]), warmup_steps=int(g["warmup_steps"]))
    p0 = {k: p[k].clone() for k in names}
    x = images.cuda()
    steps = len(g["lr"])
    logs = []
    for s in range(steps):
        lr = tr.opt.current_lr()
        tr.step(x)
        logs.append({**tr.read_log(), "lr": lr})
    assert logs[0]["rec_loss"] == logs[1]["rec_loss"]            # the first optimiser step runs at lr 0
    assert logs[3]["rec_loss"] < logs[2]["rec_loss"] < logs[1]["rec_loss"]
    by_id = {id(q): n for n, q in vae.named_parameters()}
    p1 = {k: q.detach().cpu() for k, q in vae.named_parameters() if k in p0}
    ema = {by_id[id(q)]: e.detach().cpu() for q, e in zip(tr.fp.params, tr.fp.ema_state())}
    assert sorted(ema) == sorted(names)
    check_step_small(g, logs, p0, p1, ema, names, steps - 1, tol_loss=2e-2, tol_norm=5e-2, tol_abs_delta=5e-2, tol_signed=0.25, min_cos=0.9, tol_ema=0.5)


class _ModeRecordingVelocity(_TinyVelocity):
    """_TinyVelocity that records (module.training, torch.is_grad_enabled()) at every call."""

    def __init__(self):
        super().__init__()
        self.calls = []

    def forward(self, xt, t, y):
        self.calls.append((self.training, torch.is_grad_enabled()))
        return super().forward(xt, t, y)


def test_dmd_trainer_student_modes_and_adversarial_branch():
    """train_dmd.py:534 / :561-562: the student is in EVAL mode (no label dropout) for the no-grad velocity evaluations of the DMD loss and in TRAIN mode
    for its own flow-matching turn, whatever mode the caller left it in; :244-256 / :546-556: with a discriminator attached the VAE turn carries the
    adaptive-weight adversarial term and the discriminator takes its own step, from `disc_start_step` on."""
    from dmvae_amd.models.init_param import init_weights
    from dmvae_amd.models.patchgan import NLayerDiscriminator
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import DMDTrainer
    torch.manual_seed(31)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).cuda()
    teacher, student = _TinyVelocity().cuda().requires_grad_(False), _ModeRecordingVelocity().cuda()
    disc = NLayerDiscriminator()
    init_weights(disc, 0.02)
    disc = disc.cuda()
    d0 = torch.cat([p.detach().flatten() for p in disc.parameters()]).clone()
    tr = DMDTrainer(vae, None, teacher, student, dmd_weight=5.0, dmd_cfg_scale=2.0, num_classes=10, vae_train_every=2, warmup_steps=1, disc=disc,
                    disc_weight=0.5, disc_start_step=2)
    images = torch.rand(2, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)) * 2 - 1
    labels = torch.tensor([3, 7], device="cuda")
    student.eval()                                     # the reference's pre-loop state (train_dmd.py:501)
    assert tr.batch_cfg is None                        # the default: automatic -- a velocity model that is not this build's per-sample HIP LightningDiT
    # (here: a plain nn.Module) gets the reference's call structure, conditional and unconditional evaluation as two B-sized calls (train_dmd.py:211-217)
    tr.step(images, labels)                            # step 0: VAE turn (DMD: two student evaluations with CFG), then the student's turn
    assert student.calls == [(False, False), (False, False), (True, True)], student.calls
    assert tr.read_log()["d_weight"] == 0.0 and torch.equal(torch.cat([p.detach().flatten() for p in disc.parameters()]), d0)    # before disc_start_step
    student.calls.clear()
    tr.step(images, labels)                            # step 1: student only
    assert student.calls == [(True, True)]
    student.calls.clear()
    tr.batch_cfg = True                                # the caller vouches its models are per-sample: both evaluations as ONE call on 2B samples
    tr.step(images, labels)                            # step 2: VAE turn again, adversarial branch active now
    assert student.calls == [(False, False), (True, True)], student.calls
    log, dlog = tr.read_log(), tr.read_disc_log()
    assert log["d_weight"] > 0 and dlog["d_loss"] > 0 and dlog["disc_norm"] > 0
    assert all(v == v and abs(v) < 1e6 for v in log.values())
    tr.step(images, labels)
    tr.step(images, labels)                            # step 4: second discriminator step (the first ran at warm-up lr 0)
    assert not torch.equal(torch.cat([p.detach().flatten() for p in disc.parameters()]), d0)
    ck = tr.checkpoint()
    assert set(ck) >= {"model", "vae_wo_ddp", "disc_wo_ddp", "opt_sit", "opt_vae", "opt_disc", "steps"} and ck["steps"] == 5


def test_tokenizer_trainer_checkpoint_resume_is_bit_exact():
    """checkpoint() -> load() into a trainer built from OTHER weights: the continued run equals the uninterrupted one bit for bit (weights, EMA, both
    Adam moments, warm-up position), and the optimiser entry loads into torch.optim.AdamW over vae.parameters() like the reference's opt_vae."""
    images = torch.rand(2, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)) * 2 - 1
    a = _trainer(True, seed=3, with_lpips=False)
    for _ in range(3):
        a.step(images)
    ck = a.checkpoint()
    assert set(ck) >= {"vae_wo_ddp", "vae_ema", "opt_vae", "scheduler_vae", "steps"} and ck["steps"] == 3 and ck["scheduler_vae"]["last_epoch"] == 3
    for _ in range(2):
        a.step(images)
    b = _trainer(True, seed=11, with_lpips=False)
    assert not torch.equal(a.fp.flat, b.fp.flat)
    b.load(ck)
    assert b.global_step == 3 and b.opt.t == 3
    for _ in range(2):
        b.step(images)
    assert torch.equal(a.fp.flat, b.fp.flat) and torch.equal(a.fp.ema, b.fp.ema)
    assert torch.equal(a.opt.exp_avg, b.opt.exp_avg) and torch.equal(a.opt.exp_avg_sq, b.opt.exp_avg_sq)
    # the reference builds optimizer_vae over the TRAINABLE parameters (train_tokenizer.py:381-382): its resume path accepts the entry, and every state lands
    # on the parameter it belongs to
    trainable = [p for p in a.vae.parameters() if p.requires_grad]
    assert len(trainable) < len(list(a.vae.parameters())) and len(ck["opt_vae"]["param_groups"][0]["params"]) == len(trainable)
    ref_opt = torch.optim.AdamW(trainable, lr=1e-4, betas=(0.9, 0.95), weight_decay=0.005)
    ref_opt.load_state_dict(ck["opt_vae"])
    assert float(ref_opt.state[trainable[-1]]["step"]) == 3.0
    assert all(ref_opt.state[p]["exp_avg"].shape == p.shape for p in trainable)
    with pytest.raises(ValueError):                              # an entry taken over another parameter list is refused, not silently mis-assigned
        b.opt.load_state_dict(ck["opt_vae"], list(b.vae.parameters()))


def test_tokenizer_step_calls_no_library_gemm(monkeypatch):
    """nothing on the tokenizer step's path goes through a vendor GEMM any more (the encoder's Linear layers run csrc/gemm_pp.hip, the
    bottleneck MLP and the decoder this build's conv / GEMM kernels).  Every torch entry point that would reach hipBLASLt / rocBLAS raises for the duration of two
    full steps (forward, losses, backward, optimiser) of the width-256 model -- the ViT stand-in inside the bf16 encoder kernels' range."""
    import warnings
    import torch.nn.functional as F
    from test_oracle_golden import vae_tiny_params
    from dmvae_amd.train import TokenizerTrainer
    from dmvae_amd.utils.lpips import LPIPS
    p, vae = vae_tiny_params(seed=71, width=256)
    vae.load_state_dict(p, strict=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lp = LPIPS().eval().requires_grad_(False)
    tr = TokenizerTrainer(vae.cuda(), lp.cuda(), lr=2e-6, warmup_steps=1)
    x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(3)) * 2 - 1).cuda()
    tr.step(x)                                    # first step outside the guard: lazy one-time set-up may do what it likes
    called = []

    def guard(name):
        def f(*a, **k):
            called.append(name)
            raise AssertionError(f"{name} called inside the tokenizer step: a library GEMM on the measured path")
        return f
    for mod, names in ((torch, ("matmul", "mm", "bmm", "addmm", "baddbmm", "einsum")), (F, ("linear", "bilinear", "scaled_dot_product_attention", "conv2d", "conv_transpose2d"))):
        for n in names:
            monkeypatch.setattr(mod, n, guard(f"{mod.__name__}.{n}"))
    monkeypatch.setattr(torch.Tensor, "matmul", guard("Tensor.matmul"))
    monkeypatch.setattr(torch.Tensor, "__matmul__", guard("Tensor.__matmul__"))
    for _ in range(2):
        tr.step(x)
    log = tr.read_log()
    assert not called and log["rec_loss"] == log["rec_loss"]          # finite, and nothing tripped the guard


def test_evaluate_on_the_hip_path_matches_the_oracle_reductions():
    """dmvae_amd.evaluate.evaluate (train_tokenizer.py:324-367 without FID) with the real VAE on the HIP kernels under autocast: PSNR / latent_mean / latent_scale
    equal the oracle's reductions (oracle/ref_cpu.py::eval_metrics) over the same encode / decode calls; the result feeds DMDTrainer / SamplePipeline."""
    from dmvae_amd import evaluate as E
    from dmvae_amd.models.vae import VAE
    from oracle import ref_cpu as R
    torch.manual_seed(41)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).cuda()
    g = torch.Generator().manual_seed(3)
    data = [(torch.rand(b, 3, 256, 256, generator=g) * 2 - 1, torch.zeros(b, dtype=torch.long)) for b in (2, 1)]
    r = E.evaluate(vae, data, num_samples=3)
    assert vae.training and r["batches"] == 2 and r["images"] == 3

    def enc(x):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return vae.encode(x)

    def dec(z):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            return vae.decode(z)
    ro = R.eval_metrics(enc, dec, [x.cuda() for x, _ in data], 3)
    for k in ("PSNR", "latent_mean", "latent_scale"):
        assert abs(r[k] - ro[k]) <= 1e-5 * abs(ro[k]) + 1e-7, (k, r[k], ro[k])
    assert r["PSNR"] == r["PSNR"] and 0 < r["latent_scale"] < 1e4
    kw = E.latent_stats(r)
    from dmvae_amd.sample import SamplePipeline  # noqa: F401  (the keywords are SamplePipeline's / DMDTrainer's constructor arguments: tests/test_oracle_eval.py)
    assert set(kw) == {"latent_mean", "latent_scale"}


def _small_dit(seed):
    """LightningDiT inside the HIP kernels' range at the DMD stage's latent shape (32 channels x 16 x 16, patch 1): width 192 = 3 heads x 64, depth 2."""
    from dmvae_amd.models.lightningdit import LightningDiT
    torch.manual_seed(seed)
    m = LightningDiT(input_size=16, patch_size=1, in_channels=32, hidden_size=192, depth=2, num_heads=3, num_classes=10)
    with torch.no_grad():
        for p in m.parameters():            # the reference zero-initialises the adaLN and output layers (lightningdit.py:367-376): perturb, or v == 0
            if p.abs().max() == 0:
                p.normal_(0, 0.02)
    return m.cuda()


def _gemm_guards(monkeypatch, what):
    import torch.nn.functional as F
    called = []

    def guard(name):
        def f(*a, **k):
            called.append(name)
            raise AssertionError(f"{name} called inside {what}: a library GEMM")
        return f
    for mod, names in ((torch, ("matmul", "mm", "bmm", "addmm", "baddbmm", "einsum")), (F, ("linear", "bilinear", "scaled_dot_product_attention", "conv2d", "conv_transpose2d"))):
        for n in names:
            monkeypatch.setattr(mod, n, guard(f"{mod.__name__}.{n}"))
    monkeypatch.setattr(torch.Tensor, "matmul", guard("Tensor.matmul"))
    monkeypatch.setattr(torch.Tensor, "__matmul__", guard("Tensor.__matmul__"))
    return called


def test_dmd_and_diffusion_steps_call_no_library_gemm(monkeypatch):
    """The twin of test_tokenizer_step_calls_no_library_gemm for configs C3 / C4 (train_dmd.py:506-575, train_diffusion.py:268-297): with the per-sample
    conditioning Linears of LightningDiT (adaLN modulations, timestep embedder) on csrc/linear_rows.hip nothing in a DMD cycle -- VAE turn with the trainable
    ViT encoder, four no-grad teacher / student evaluations (one 2B call each), the student's own training turn -- nor in a latent-diffusion train step reaches
    hipBLASLt / rocBLAS / MIOpen: every torch entry point that would raises for two full cycles."""
    from dmvae_amd.models.vae import VAE
    from dmvae_amd.train import DiffusionTrainer, DMDTrainer
    from dmvae_amd.utils.lpips import LPIPS
    torch.manual_seed(21)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).cuda()
        lp = LPIPS().eval().requires_grad_(False).cuda()
    teacher, student = _small_dit(1).eval().requires_grad_(False), _small_dit(2)
    tr = DMDTrainer(vae, lp, teacher, student, dmd_weight=5.0, dmd_cfg_scale=2.0, num_classes=10, vae_train_every=2, warmup_steps=1)
    images = torch.rand(2, 3, 256, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)) * 2 - 1
    labels = torch.tensor([3, 7], device="cuda")
    for _ in range(2):
        tr.step(images, labels)                   # one whole cycle outside the guard: lazy one-time set-up may do what it likes
    called = _gemm_guards(monkeypatch, "the DMD cycle")
    for _ in range(4):
        tr.step(images, labels)
    log = tr.read_log()
    assert not called and all(v == v for v in log.values()) and log["dmd_loss"] > 0 and log["diffusion_loss"] > 0, (called, log)
    monkeypatch.undo()
    # config C4: the latent-diffusion trainer on the same pieces
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vae2 = VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4)).cuda()
    dt = DiffusionTrainer(_small_dit(3), vae2, lr=1e-4)
    dt.step(images, labels)
    called = _gemm_guards(monkeypatch, "the latent-diffusion step")
    for _ in range(2):
        dt.step(images, labels)
    assert not called and dt.read_log()["loss"] == dt.read_log()["loss"]


@pytest.mark.gpu
def test_transport_times_on_the_device_equal_the_pageable_copy():
    """The pinned, non-blocking hand-over of the CPU-drawn times (transport.cpu_rand_like_batch) gives the device the values `th.rand((B,)).to(x1)` gives it, draw after
    draw (more draws than staging buffers), and leaves the CPU generator where th.rand leaves it."""
    from dmvae_amd import transport as T
    x1 = torch.zeros(16, 32, 16, 16, device="cuda")
    torch.manual_seed(11)
    ref = [torch.rand((16,)).to(x1) for _ in range(7)]
    end_ref = torch.rand(3)
    torch.manual_seed(11)
    got = [T.cpu_rand_like_batch(x1).to(x1) for _ in range(7)]
    end = torch.rand(3)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(ref, got)) and torch.equal(end, end_ref)


def _dmd_capture_trainer(g):
    """DMDTrainer over the capture's models (tests/test_oracle_dmd_step.py::dmd_step_inputs), on the GPU, with the capture's hyper-parameters."""
    from dmvae_amd.train import DMDTrainer
    from dmvae_amd.utils.lpips import LPIPS
    from test_oracle_dmd_step import dmd_step_inputs
    pv, vae, lp_w, teacher, student, images, labels, draws = dmd_step_inputs(g)
    vae.load_state_dict(pv, strict=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lp = LPIPS().eval().requires_grad_(False)
    lp.load_state_dict(lp_w, strict=False)
    # the capture's VGG trunk: name-seeded (oracle/capture_golden_dmd_step.py), rebuilt the way tests/test_oracle_golden.py::lpips_params does for the oracle
    from test_oracle_golden import lpips_params
    full = lpips_params(g, "lp.")
    lp.load_state_dict({k: v for k, v in full.items() if k in lp.state_dict()}, strict=False)
    tr = DMDTrainer(vae.cuda(), lp.cuda(), teacher.cuda().eval().requires_grad_(False), student.cuda(), lr=float(g["lr"]), diff_lr=float(g["diff_lr"]), wd=float(g["wd"]),
                    dmd_weight=float(g["dmd_weight"]), dmd_cfg_scale=float(g["cfg"]), num_classes=10, latent_mean=float(g["latent_mean"]),
                    latent_scale=float(g["latent_scale"]), vae_train_every=int(g["vae_train_every"]), warmup_steps=int(g["warmup_steps"]))
    return tr, pv, lp_w, teacher, student, images, labels, draws


def _run_dmd_capture_steps(tr, images, labels, draws, on_step):
    """Four steps with the capture's draws injected: DMDTrainer._sample hands out the recorded (t, x0) in the order the step asks for them (the DMD loss's on VAE
    turns, then the student's), the student's label drop-out is the recorded mask."""
    student = tr.student
    orig_sample, orig_drop = tr._sample, student.y_embedder.token_drop
    try:
        for step, d in enumerate(draws):
            queue = ([d["dmd"]] if "dmd" in d else []) + [d["student"][:2]]
            tr._sample = lambda x1, q=queue: tuple(v.to(x1) for v in q.pop(0))
            student.y_embedder.token_drop = lambda lab, force_drop_ids=None, m=d["student"][2]: torch.where(m.to(lab.device), torch.full_like(lab, 10), lab)
            tr.step(images.cuda(), labels.cuda())
            tr.wait_optimizers()
            assert not queue
            on_step(step, tr.read_log())
    finally:
        tr._sample, student.y_embedder.token_drop = orig_sample, orig_drop


def test_dmd_trainer_vs_reference_capture_c3():
    """Config C3's step pinned to the REFERENCE: `DMDTrainer` (trainable ViT encoder, decoder, LPIPS, the DMD loss over teacher / student LightningDiT with CFG as one
    2B call, then the student's flow-matching turn) replays the four steps oracle/capture_golden_dmd_step.py recorded from train_dmd.py:506-575 run with the
    reference's own modules and VAELossFunction, with the capture's draws injected.  Bars: the bf16-site criterion -- as close to the reference's f32 numbers as the
    CPU oracle with bf16 rounding at the autocast sites (oracle.ref_cpu.dmd_train_steps(q=bf16_round)) is, x 1.15 + a floor -- for every logged scalar of the first
    step, the fifteen fully captured gradients and every parameter's gradient norm (VAE: decoder, bottleneck, the TRAINABLE encoder; student); the later steps'
    scalars to the bf16 floor of a trajectory (2 % losses, 8 % norms)."""
    from conftest import load_golden
    from oracle import ref_cpu as R
    from test_oracle_dmd_step import DIT_KW as KW, REF_NAME, SMALL_SIT, SMALL_VAE, hyper
    g = load_golden("dmd_step_small")
    tr, pv, lp_w, teacher, student, images, labels, draws = _dmd_capture_trainer(g)
    vnames = [n for n, _ in tr.vae.named_parameters()]
    snames = [str(n) for n in g["student_names"]]
    # ---- the bf16-site oracle's first step (CPU): what "as close as bf16 allows" means for each number ----
    pt = {k: v.detach().cpu().clone() for k, v in teacher.state_dict().items()}
    ps = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
    ovg, osg = {}, {}
    ologs, _, _ = R.dmd_train_steps(images, labels, {k: v.clone() for k, v in pv.items()}, lp_w, pt, ps, vnames, snames, draws[:1], **hyper(g), q=R.bf16_round,
                                    on_vae_grads=lambda s, gr: ovg.update({k: v.clone() for k, v in gr.items()}),
                                    on_student_grads=lambda s, gr: osg.update({k: v.clone() for k, v in gr.items()}))
    rl2 = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()
    vid = {id(p): n for n, p in tr.vae.named_parameters()}
    sid = {id(p): n for n, p in student.named_parameters()}

    def on_step(step, log):
        for k in [k for k in ("L1", "LPIPS", "rec_loss", "dmd_loss", "dmd_gradient_norm", "vae_norm", "diffusion_loss", "sit_norm") if f"log{step}.{k}" in g]:
            want, got = float(g[f"log{step}.{k}"]), log[k]
            norm_like = k in ("vae_norm", "sit_norm", "dmd_gradient_norm")
            if step == 0:
                e_orc = abs(ologs[0][k] - want) / abs(want)
                print(f"C3 step 0 {k}: reference {want:.6f}  HIP {got:.6f}  bf16-site oracle {ologs[0][k]:.6f}")
                assert abs(got - want) / abs(want) < 1.15 * e_orc + (2e-2 if norm_like else 3e-3), (k, got, want, ologs[0][k])
            else:
                assert abs(got - want) < (8e-2 if norm_like else 2e-2) * abs(want), (step, k, got, want)
        if step == 0:
            vg = {vid[id(p)]: tr.fp.grad[off:off + p.numel()].view(p.shape).float().cpu() for p, off in zip(tr.fp.params, tr.fp.offsets)}
            sg = {sid[id(p)]: tr.sfp.grad[off:off + p.numel()].view(p.shape).float().cpu() for p, off in zip(tr.sfp.params, tr.sfp.offsets)}
            for k in SMALL_VAE:
                e_hip, e_orc = rl2(vg[k], g.t("vg0." + REF_NAME(k))), rl2(ovg[k], g.t("vg0." + REF_NAME(k)))
                print(f"C3 step 0 VAE grad {k}: rel-L2 to the f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
                assert e_hip < 1.15 * e_orc + 3e-3, (k, e_hip, e_orc)
            for k in SMALL_SIT:
                e_hip, e_orc = rl2(sg[k], g.t("sg0." + k)), rl2(osg[k], g.t("sg0." + k))
                print(f"C3 step 0 student grad {k}: rel-L2 to the f32 reference -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
                assert e_hip < 1.15 * e_orc + 3e-3, (k, e_hip, e_orc)
            for grads, ograds, key, ren in ((vg, ovg, "vgn0.", REF_NAME), (sg, osg, "sgn0.", lambda k: k)):
                for k, gr in grads.items():
                    if key + ren(k) not in g or k not in ograds:
                        continue
                    want = float(g[key + ren(k)][0])
                    if want < 1e-4 or k.endswith("attn_1.k.bias"):
                        continue
                    e_hip = abs(gr.double().norm().item() - want) / want
                    e_orc = abs(ograds[k].double().norm().item() - want) / want
                    assert e_hip < 1.15 * e_orc + 3e-2, (k, e_hip, e_orc)
    _run_dmd_capture_steps(tr, images, labels, draws, on_step)
