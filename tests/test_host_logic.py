"""CPU: host-side mirror of the reference interface -- constructor/RNG parity, state_dict layout, checkpoint layout,
flat-buffer optimiser plumbing, loud failure without a GPU."""
import os
import warnings

import numpy as np
import pytest
import torch

from conftest import load_golden


def _ck_equal(mod, g):
    for k, v in mod.state_dict().items():
        ck = g["ck." + k]
        assert abs(v.double().sum().item() - ck[0]) < 1e-9 and abs(v.double().abs().sum().item() - ck[1]) < 1e-9, k


def test_fixed_seed_init_reproduces_reference_decoder():
    """torch.manual_seed + the reference's constructor / post_init / init_weights order (models/vae.py:81-88) gives the
    reference's initial weights bit-for-bit (checksums captured from the reference modules)."""
    from dmvae_amd.models import flux_ae
    from dmvae_amd.models.init_param import init_weights
    g = load_golden("decoder_small")
    torch.manual_seed(11)
    dec = flux_ae.Decoder(ch=32, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, resolution=64, z_channels=16)
    dec.post_init(z_channels=32)
    init_weights(dec, 0.02)
    _ck_equal(dec, g)
    g = load_golden("decoder_full_b1")
    torch.manual_seed(21)
    dec = flux_ae.Decoder(ch=128, out_ch=3, ch_mult=(1, 2, 4, 4), num_res_blocks=2, in_channels=3, resolution=256, z_channels=16)
    dec.post_init(z_channels=32)
    init_weights(dec.conv_in, 0.02)
    init_weights(dec, 0.02)
    _ck_equal(dec, g)
    assert list(dec.state_dict().keys()) == list(g["keys"])
    assert sum(p.numel() for p in dec.parameters()) == int(g["n_params"]) == 49628451


def _tiny_vae():
    from dmvae_amd.models.vae import VAE
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=64, depth=1, num_heads=4))


def test_vae_checkpoint_layout_roundtrip(tmp_path):
    """vae.pt = {'vae_wo_ddp': sd, 'vae_ema': sd} (train_tokenizer.py:440-450); load_pretrained picks by `ema` (vae.py:110-121)."""
    a, b = _tiny_vae(), _tiny_vae()
    sd = a.state_dict()
    ema = {k: v + 1 for k, v in sd.items()}
    path = tmp_path / "vae.pt"
    torch.save({"vae_wo_ddp": sd, "vae_ema": ema, "steps": 3}, path)
    b.load_pretrained(str(path), ema=False)
    assert all(torch.equal(b.state_dict()[k], sd[k]) for k in sd)
    b.load_pretrained(str(path), ema=True)
    assert all(torch.equal(b.state_dict()[k], ema[k]) for k in sd)
    b.load_pretrained(str(tmp_path / "missing.pt"))          # reference behaviour: warn and return silently
    import copy
    c = copy.deepcopy(a)                                       # EMA copies (train_tokenizer.py:397)
    assert all(torch.equal(c.state_dict()[k], sd[k]) for k in sd)
    assert a.decoder.get_last_layer() is a.decoder.conv_out.weight and a.bottle_neck.get_last_layer() is a.bottle_neck.mlp[-1].weight


def test_flat_params_and_backward_order():
    from dmvae_amd.optim import FlatParams
    from dmvae_amd.train import backward_order_params
    vae = _tiny_vae()
    for p in vae.encoder.parameters():
        p.requires_grad_(False)
    before = {k: v.clone() for k, v in vae.state_dict().items()}
    params = backward_order_params(vae)
    assert sum(p.numel() for p in params) == sum(p.numel() for p in vae.parameters() if p.requires_grad)
    assert params[0] is vae.decoder.conv_out.weight and params[-1] is vae.bottle_neck.mlp[2].bias
    fp = FlatParams(params)
    assert all(torch.equal(vae.state_dict()[k], before[k]) for k in before)       # values preserved
    fp.flat.mul_(2.0)                                                              # params are views of the flat buffer
    assert torch.equal(vae.decoder.conv_out.weight, before["decoder.conv_out.weight"] * 2)
    vae.decoder.conv_out.weight.grad.add_(1.0)
    assert fp.grad[: vae.decoder.conv_out.weight.numel()].eq(1).all()
    vae.load_state_dict(before)                                                    # load_state_dict copies into the views
    assert torch.equal(fp.flat[: params[0].numel()].view(params[0].shape), before["decoder.conv_out.weight"])
    assert all(o % 4 == 0 for o in fp.offsets)


def test_partial_direct_gradients_zero_only_what_accumulates():
    """FlatParams.enable_direct_grads(only=...): begin_step() clears the slices of the parameters that still accumulate through autograd -- as contiguous runs, not
    the whole buffer -- and hands the direct ones to autograd as None (their kernels overwrite the flat views)."""
    from dmvae_amd.optim import FlatParams
    ps = [torch.nn.Parameter(torch.randn(n)) for n in (5, 16, 7, 8, 3)]          # offsets 0, 8, 24, 32, 40 (16-B alignment)
    fp = FlatParams(ps, with_ema=False)
    fp.enable_direct_grads(only=[ps[1], ps[2]])
    assert fp.partial and fp.accum_runs == [(0, 8), (32, 44)]
    fp.grad.fill_(3.0)
    fp.begin_step()
    assert fp.grad[:8].eq(0).all() and fp.grad[32:].eq(0).all() and fp.grad[8:32].eq(3).all()        # the direct slices are left to their kernels
    assert ps[1].grad is None and ps[2].grad is None
    for i in (0, 3, 4):
        assert ps[i].grad.data_ptr() == fp.grad.data_ptr() + fp.offsets[i] * 4
    (ps[0].sum() * 2 + ps[3].sum()).backward()
    assert fp.grad[:5].eq(2).all() and fp.grad[32:40].eq(1).all()
    fp2 = FlatParams([torch.nn.Parameter(torch.randn(4)) for _ in range(40)], with_ema=False)
    fp2.enable_direct_grads(only=fp2.params[1::2])                                  # many runs: one fill of the whole buffer instead
    assert len(fp2.accum_runs) == 20
    fp2.grad.fill_(1.0)
    fp2.begin_step()
    assert fp2.grad.eq(0).all()


def test_direct_gradient_invariant_is_checked_and_can_be_opted_out_of():
    """The direct-gradient mode leaves a direct parameter's slice to the kernel that writes it; a backward that skips one (an unused block, a Function returning None)
    would feed AdamW the previous step's gradient.  `FlatParams.check_direct_writes` (DMVAE_CHECK_DIRECT_GRADS=1) raises on exactly that, does not judge a step in
    which none of this buffer's parameters took part, and `zero_all` falls back to zeroing the whole buffer; the all-ranks RNG default follows the world size."""
    from dmvae_amd import functional as Fn
    from dmvae_amd.optim import FlatParams
    ps = [torch.nn.Parameter(torch.randn(8)) for _ in range(3)]
    fp = FlatParams(ps, with_ema=False)
    fp.enable_direct_grads()
    old = Fn.DIRECT_GRAD_WRITES
    try:
        Fn.DIRECT_GRAD_WRITES = None
        fp.check_direct_writes()                              # first call: starts counting
        assert Fn.DIRECT_GRAD_WRITES == {}
        for p in ps:
            assert Fn._dst(p) is not None                     # a backward that writes all three
        fp.check_direct_writes()                              # fine
        fp.check_direct_writes()                              # a step in which this buffer took no part: not judged
        Fn._dst(ps[0]); Fn._dst(ps[2])                        # a backward that skips ps[1]
        with pytest.raises(RuntimeError, match="not written"):
            fp.check_direct_writes()
    finally:
        Fn.DIRECT_GRAD_WRITES = old
    fp.zero_all = True
    fp.grad.fill_(5.0)
    fp.begin_step()
    assert fp.grad.eq(0).all()


def test_lambda_lr_warmup_matches_reference_schedule():
    from dmvae_amd.optim import FlatAdamWEMA, FlatParams
    p = torch.nn.Parameter(torch.zeros(8))
    opt = FlatAdamWEMA(FlatParams([p]), lr=1e-4, warmup_steps=1000)
    sgd = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(sgd, lambda s: s / 1000 if s < 1000 else 1.0)     # train_tokenizer.py:386-391
    for t in range(1003):
        opt.t = t
        assert opt.current_lr() == pytest.approx(sgd.param_groups[0]["lr"], rel=1e-12, abs=0), t
        sgd.step()
        sched.step()
    assert opt.current_lr() == 1e-4
    opt.t = 0
    assert opt.current_lr() == 0.0                       # the reference's first step runs at lr 0


def test_forward_without_gpu_raises_not_falls_back():
    from dmvae_amd import _lib
    from dmvae_amd.models.flux_ae import ResnetBlock
    with pytest.raises((_lib.DmvaeHipError, RuntimeError)):
        ResnetBlock(32, 32)(torch.zeros(1, 32, 4, 4))


def test_product_does_not_import_oracle():
    import os
    import re
    from conftest import ROOT
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dmvae_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_flat_adamw_state_dict_is_torch_adamw_layout():
    """FlatAdamWEMA.state_dict(param_order) is what torch.optim.AdamW.state_dict() holds for the same parameters (the reference's opt_vae / opt_disc /
    opt_sit / opt checkpoint entries): indices follow `module.parameters()`, frozen parameters carry no state, and it round-trips."""
    import torch
    from dmvae_amd.optim import FlatAdamWEMA, FlatParams
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(4, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 2))
    for p in net[1].parameters():
        p.requires_grad_(False)                                  # frozen block in the middle
    trainable = [p for p in net.parameters() if p.requires_grad]
    fp = FlatParams(list(reversed(trainable)), with_ema=False)   # the flat buffer's order differs from the module's
    opt = FlatAdamWEMA(fp, lr=3e-4, weight_decay=0.01, warmup_steps=10)
    opt.exp_avg.normal_(); opt.exp_avg_sq.uniform_(0.1, 1.0); opt.t = 4
    sd = opt.state_dict(list(net.parameters()))
    assert sorted(sd["state"]) == [0, 1, 4, 5] and sd["param_groups"][0]["params"] == list(range(6))
    ref = torch.optim.AdamW(list(net.parameters()), lr=3e-4, betas=(0.9, 0.95), weight_decay=0.01)
    ref.load_state_dict(sd)
    w0 = list(net.parameters())[0]
    assert float(ref.state[w0]["step"]) == 4.0 and ref.state[w0]["exp_avg"].shape == w0.shape
    off = fp.offsets[[id(q) for q in fp.params].index(id(w0))]
    assert torch.equal(ref.state[w0]["exp_avg"].reshape(-1), opt.exp_avg[off:off + w0.numel()])
    # round trip through the reference optimiser's own state_dict
    opt2 = FlatAdamWEMA(FlatParams(list(reversed(trainable)), with_ema=False), lr=1.0, warmup_steps=10)
    opt2.load_state_dict(ref.state_dict(), list(net.parameters()))
    assert opt2.t == 4
    for q, o in zip(fp.params, fp.offsets):               # (the 16-byte alignment padding between tensors is not state)
        assert torch.equal(opt2.exp_avg[o:o + q.numel()], opt.exp_avg[o:o + q.numel()]) and torch.equal(opt2.exp_avg_sq[o:o + q.numel()], opt.exp_avg_sq[o:o + q.numel()])
    assert opt.scheduler_state_dict()["last_epoch"] == 4 and abs(opt.current_lr() - 3e-4 * 0.4) < 1e-12


def test_lpips_reports_a_missing_trunk_and_loads_torchvision_layout(tmp_path, monkeypatch):
    """The reference's vgg.pth holds only the five lin layers; its trunk comes from torchvision's pretrained VGG16 (utils/lpips.py:119), which is not
    available offline.  `strict=False` must not hide that: a warning names the missing trunk, `trunk_loaded` says so, `load_trunk` maps
    torchvision's features.N.* keys onto net.slice{k}.{N}.*, and a checkpoint without the lin layers is an error."""
    import warnings
    import pytest
    import torch
    from dmvae_amd.utils.lpips import LPIPS
    monkeypatch.delenv("DMVAE_LPIPS_RANDOM_TRUNK", raising=False)
    lin_only = {f"lin{i}.model.1.weight": torch.rand(1, c, 1, 1) for i, c in enumerate((64, 128, 256, 512, 512))}
    ck = tmp_path / "vgg.pth"
    torch.save(lin_only, ck)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lp = LPIPS(ckpt_path=str(ck))
    msgs = " | ".join(str(w.message) for w in rec)
    assert "RANDOM" in msgs and "net.slice" in msgs and not lp.trunk_loaded
    assert torch.equal(lp.lin2.model[1].weight, lin_only["lin2.model.1.weight"])
    tv = {}
    for k in range(1, 6):
        for idx, m in getattr(lp.net, f"slice{k}").named_children():
            if isinstance(m, torch.nn.Conv2d):
                tv[f"features.{idx}.weight"], tv[f"features.{idx}.bias"] = torch.randn_like(m.weight), torch.randn_like(m.bias)
    tv["classifier.0.weight"] = torch.zeros(2, 2)
    assert len(tv) == 27
    tvp = tmp_path / "vgg16-397923af.pth"
    torch.save(tv, tvp)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        lp2 = LPIPS(ckpt_path=str(ck), trunk_path=str(tvp))
    assert lp2.trunk_loaded and not rec
    assert torch.equal(dict(lp2.net.slice3.named_children())["14"].weight, tv["features.14.weight"])
    assert torch.equal(lp2.net.slice1[0].bias, tv["features.0.bias"])
    del tv["features.28.bias"]
    with pytest.raises(KeyError, match="features.28.bias"):
        lp.load_trunk(tv)
    torch.save({k: v for k, v in lin_only.items() if not k.startswith("lin4")}, ck)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(KeyError, match="lin4"):
            LPIPS(ckpt_path=str(ck))


def test_toy_sshape_sampler_and_points_match_the_reference():
    """dmvae_amd.toy's host-side pieces of config C1: SShapeDistribution2D reproduces the reference's samples bit for bit (golden G13, captured from
    toy_example_2d/sshpae.py with random_state=42) and create_learnable_points its initialiser (toy_example_2d/dmd.py:139-145: global seed, U[-1.5, 1.5])."""
    from dmvae_amd.toy import SShapeDistribution2D, create_learnable_points
    g = load_golden("sshape")
    pts, comp = SShapeDistribution2D(random_state=42).sample(1536)
    assert np.array_equal(pts, g["samples"]) and comp.shape == (1536,) and not comp.any()
    assert np.abs(pts).max() <= 1.0
    unflipped, _ = SShapeDistribution2D(random_state=42, flip_y=False).sample(1536)
    assert np.array_equal(unflipped[:, 0], g["samples"][:, 0]) and np.array_equal(unflipped[:, 1], -g["samples"][:, 1])     # flip_y mirrors the ordinate only
    p = create_learnable_points(1536, 2, "cpu", seed=42)
    torch.manual_seed(42)
    assert torch.equal(p.data, torch.rand(1536, 2) * 3.0 - 1.5) and isinstance(p, torch.nn.Parameter)
    assert p.min() >= -1.5 and p.max() <= 1.5


def test_run_on_mi355x_launcher_shadows_the_reference_import_paths():
    """run_on_mi355x.py: in a fresh interpreter, after install_shadow() every module path the reference's drivers import their model
    code from (train_tokenizer.py:10-17, train_dmd.py:10-17, train_diffusion.py:15-17, sample_50k.py:6-14) resolves to this build's mirror, the names they
    import exist, and a module this build does not shadow still resolves below the reference's directory when one is given."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, os, tempfile
sys.path.insert(0, %r)
import run_on_mi355x as L
ref = tempfile.mkdtemp()
os.makedirs(os.path.join(ref, "models")); os.makedirs(os.path.join(ref, "utils"))
open(os.path.join(ref, "models", "dinodisc.py"), "w").write("class DinoDisc: marker = 'reference file'\n")
open(os.path.join(ref, "utils", "dist.py"), "w").write("WHO = 'reference utils.dist'\n")
L.install_shadow(ref)
from models.vae import VAE
from models import VAE as V2, NLayerDiscriminator, DinoDisc
from models.flux_ae import Decoder, Encoder
from models.init_param import init_weights
from models.patchgan import NLayerDiscriminator as N2
from utils.lpips import LPIPS
from utils.diffaug import DiffAug
import utils.dist as dist
from diffusion.lightningdit.lightningdit import LightningDiT_models, LightningDiT
from diffusion.transport import create_transport, Transport, Sampler
import dmvae_amd.models.vae, dmvae_amd.models.flux_ae, dmvae_amd.utils.lpips, dmvae_amd.transport, dmvae_amd.models.lightningdit
assert VAE is dmvae_amd.models.vae.VAE is V2 and Decoder is dmvae_amd.models.flux_ae.Decoder and LPIPS is dmvae_amd.utils.lpips.LPIPS
assert NLayerDiscriminator is N2 and create_transport is dmvae_amd.transport.create_transport
assert LightningDiT is dmvae_amd.models.lightningdit.LightningDiT and "LightningDiT-XL/1" in LightningDiT_models
assert DinoDisc.marker == 'reference file' and dist.WHO == 'reference utils.dist'
print("shadow ok")
""" % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "shadow ok" in r.stdout, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(root, "run_on_mi355x.py"), "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "models.vae" in r.stdout and "dmvae_amd.transport" in r.stdout, r.stderr[-2000:]


def test_transport_times_consume_the_cpu_generator_like_the_reference():
    """transport.cpu_rand_like_batch (the pinned, non-blocking form of `th.rand((B,)).to(x1)`, transport.py:110-111) draws from the CPU generator exactly what
    th.rand draws -- two consecutive draws, and Transport.sample's t on a CPU tensor equals the reference expression."""
    from dmvae_amd import transport as T
    torch.manual_seed(5)
    a, b = torch.rand((7,)), torch.rand((7,))
    torch.manual_seed(5)
    a2, b2 = T.cpu_rand_like_batch(torch.zeros(7, 3)), T.cpu_rand_like_batch(torch.zeros(7, 3))
    assert torch.equal(a, a2) and torch.equal(b, b2)
    tr = T.create_transport("Linear", "velocity", None, 0.0, 0.0)
    x1 = torch.zeros(5, 2, 4)
    torch.manual_seed(9)
    t, x0, _ = tr.sample(x1)
    torch.manual_seed(9)
    x0_ref = torch.randn_like(x1)
    t_ref = torch.rand((5,)).to(x1)
    assert torch.equal(t, t_ref) and torch.equal(x0, x0_ref)


def test_every_entry_point_is_mapped_to_reference_lines_in_integration_md():
    """tools/integration_index.py: every declaration of include/dmvae_hip.h occurs in INTEGRATION.md (Appendix A is generated from the header's comments) and only the two
    plumbing entries (error string, ABI version) are without a reference citation."""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "integration_index.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 not in INTEGRATION.md, 0 without a citation" in r.stdout, r.stdout


def test_every_profiles_file_is_cited_once_in_design_md():
    """tools/design_profiles_index.py: every file under profiles/ occurs exactly once in DESIGN.md (Appendix P) and every [P:key] of the text names a file."""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_profiles_index.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_pointer_table_cache_never_drops_recurring_or_captured_tables():
    """ops._TableCache: device tables that launched kernels / captured graphs read are evicted only when they were used ONCE (an unsettled
    allocator's activation pointers), and even those stay allocated for one more generation; a table looked up twice (parameter / flat-buffer pointers) stays."""
    from dmvae_amd import ops
    c = ops._TableCache()
    c.LIMIT = 8
    c.store("param", "P")
    assert c.lookup("param") == "P" and "param" in c.pinned           # second use: pinned
    for i in range(40):
        c.store(("act", i), f"A{i}")
    assert "param" in c and c.lookup("param") == "P"
    assert len(c) - len(c.pinned) <= c.LIMIT + 1                      # the single-use entries were trimmed ...
    assert c.grave and all(isinstance(g, str) for g in c.grave)       # ... but the last trimmed generation is still referenced (deferred free)
    assert ("act", 39) in c
    c.discard(("act", 39))
    assert ("act", 39) not in c and "A39" in c.grave
    c.clear()
    assert not c and not c.pinned and not c.grave


def test_reparam_kl_spec_closed_forms():
    """oracle.ref_cpu.reparam_kl -- the build's own spec of the reparameterise hook (parity unpinned: no reference counterpart, models/vae.py:90-98 is
    deterministic): N(0, 1) posterior has KL 0 and z = eps; constant (m, s^2) has KL 0.5 (m^2 + s^2 - 1 - ln s^2); eps None is the mode; gradient of the KL."""
    from oracle import ref_cpu as R
    eps = torch.randn(50, 8, dtype=torch.float64)
    z, kl, klm = R.reparam_kl(torch.zeros(50, 16, dtype=torch.float64), eps)
    assert torch.equal(z, eps) and kl.abs().max() == 0 and klm == 0
    m, s2 = 0.3, 1.7
    mom = torch.cat([torch.full((50, 8), m, dtype=torch.float64), torch.full((50, 8), s2, dtype=torch.float64).log()], 1).requires_grad_(True)
    z, kl, klm = R.reparam_kl(mom, None)
    want = 0.5 * (m * m + s2 - 1 - np.log(s2))
    assert torch.equal(z, mom[:, :8]) and abs(klm.item() - want) < 1e-12 and (kl - want).abs().max() < 1e-12
    klm.backward()
    g = mom.grad
    assert (g[:, :8] - m / 400).abs().max() < 1e-15 and (g[:, 8:] - 0.5 * (s2 - 1) / 400).abs().max() < 1e-15


def test_integration_md_fits_a_160_column_view():
    """tools/wrap_md.py --check: outside code fences no line of INTEGRATION.md is longer than 160 characters (long tables are wrapped bullet lists)."""
    import subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "wrap_md.py"), os.path.join(ROOT, "INTEGRATION.md"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


def test_every_environment_switch_the_package_reads_is_listed_in_integration_md():
    """INTEGRATION.md ("Run-time switches") claims to list every environment variable the package reads: every DMVAE_* name that appears in a getenv / os.environ
    read of dmvae_amd/ (Python and HIP sources) or bench.py must appear in that file."""
    import glob, re
    from conftest import ROOT
    pat = re.compile(r'(?:getenv\(|environ\.get\(|environ\[|_env_int\(|_flag\()\s*"(DMVAE_[A-Z0-9_]+)"')
    names = set()
    for f in glob.glob(os.path.join(ROOT, "dmvae_amd", "**", "*.py"), recursive=True) + glob.glob(os.path.join(ROOT, "dmvae_amd", "csrc", "*.hip")) + [os.path.join(ROOT, "bench.py")]:
        names |= set(pat.findall(open(f, encoding="utf-8").read()))
    assert len(names) >= 20, names
    doc = open(os.path.join(ROOT, "INTEGRATION.md"), encoding="utf-8").read()
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
