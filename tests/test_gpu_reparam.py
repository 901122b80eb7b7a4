"""The build-defined reparameterise hook and posterior-form KL (csrc/reparam.hip, models/vae.py VAE(reparameterize=True)) -- PARITY UNPINNED: the reference's
VAE.forward is deterministic (models/vae.py:90-98), so the oracle is the build's own spec `oracle.ref_cpu.reparam_kl` (+ closed forms), exactly as for the
batch-moment KL / MMD.  What IS pinned to the reference: with the keyword at its default the VAE is the reference's -- same modules, same
state_dict, same bits as before the hook existed."""
import pytest
import torch

from conftest import rel_err
from oracle import ref_cpu as R

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-4


def _ops():
    from dmvae_amd import ops
    return ops


@pytest.mark.parametrize("rows,c", [(8192, 32), (777, 16), (5, 4), (3000, 64), (70000, 32), (33, 256)])
def test_reparam_kl_f32_vs_spec(rows, c):
    ops = _ops()
    gen = torch.Generator().manual_seed(rows + c)
    mom = torch.randn(rows, 2 * c, generator=gen) * 0.8 + 0.1
    eps = torch.randn(rows, c, generator=gen)
    z, kl = ops.reparam_kl_fwd(mom.to(DEV), eps.to(DEV))
    mr = mom.double().requires_grad_(True)
    zr, klr, klm = R.reparam_kl(mr, eps.double())
    assert rel_err(z.cpu(), zr.detach()) < TOL
    assert rel_err(kl[:c].cpu(), klr.detach()) < TOL and abs(kl[c].item() - klm.item()) < TOL * abs(klm.item())
    dz = torch.randn(rows, c, generator=gen)
    g = torch.tensor([0.37])
    ((zr * dz.double()).sum() + 0.37 * klm).backward()
    dm = ops.reparam_kl_bwd(mom.to(DEV), eps.to(DEV), dz.to(DEV), g.to(DEV), 1.0)
    assert rel_err(dm.cpu(), mr.grad) < TOL
    # deterministic: same bits on a second call
    z2, kl2 = ops.reparam_kl_fwd(mom.to(DEV), eps.to(DEV))
    assert torch.equal(z, z2) and torch.equal(kl, kl2)


def test_reparam_kl_closed_forms():
    ops = _ops()
    c, rows = 32, 4096
    zero = torch.zeros(rows, 2 * c, device=DEV)
    eps = torch.randn(rows, c, device=DEV)
    z, kl = ops.reparam_kl_fwd(zero, eps)
    assert torch.equal(z, eps) and kl.abs().max().item() == 0.0            # N(0, 1) posterior: z = eps, KL = 0 exactly
    dm = ops.reparam_kl_bwd(zero, eps, None, None, 1.0)
    assert dm.abs().max().item() == 0.0                                    # ... and it is the KL's stationary point
    mom = torch.randn(rows, 2 * c, device=DEV)
    zm, klm = ops.reparam_kl_fwd(mom, None)                                # posterior mode: z = mu, bit for bit; KL does not depend on eps
    assert torch.equal(zm, mom[:, :c].contiguous())
    _, kle = ops.reparam_kl_fwd(mom, eps)
    assert torch.equal(klm, kle)
    # mu = m, logvar = log s^2 constant: KL = 0.5 (m^2 + s^2 - 1 - log s^2)
    m, s2 = 0.7, 2.5
    const = torch.cat([torch.full((rows, c), m), torch.full((rows, c), s2).log()], dim=1).to(DEV)
    _, klc = ops.reparam_kl_fwd(const, eps)
    want = 0.5 * (m * m + s2 - 1.0 - torch.tensor(s2).log().item())
    assert (klc.cpu() - want).abs().max().item() < 1e-5 * want


def test_reparam_kl_bf16_storage():
    """Production storage (the bottleneck's bf16 output): z within one bf16 rounding of the spec on the same bf16 moments, KL at f32 accuracy."""
    ops = _ops()
    rows, c = 8192, 32
    gen = torch.Generator().manual_seed(3)
    mom = (torch.randn(rows, 2 * c, generator=gen) * 0.5).bfloat16()
    eps = torch.randn(rows, c, generator=gen)
    z, kl = ops.reparam_kl_fwd(mom.to(DEV), eps.to(DEV))
    zr, klr, klm = R.reparam_kl(mom.double(), eps.double())
    assert z.dtype == torch.bfloat16
    assert (z.float().cpu() - zr.float()).abs().max().item() <= 2.0 ** -8 * zr.abs().max().item()
    assert rel_err(kl[:c].cpu(), klr) < TOL
    dz = (torch.randn(rows, c, generator=gen)).bfloat16()
    dm = ops.reparam_kl_bwd(mom.to(DEV), eps.to(DEV), dz.to(DEV), None, 0.0)
    mu, lv = mom.double().chunk(2, dim=-1)
    want = torch.cat([dz.double(), dz.double() * 0.5 * torch.exp(0.5 * lv) * eps.double()], dim=1)
    assert (dm.float().cpu() - want.float()).abs().max().item() <= 2.0 ** -8 * want.abs().max().item()


def test_reparam_kl_rejects_bad_arguments():
    from dmvae_amd._lib import DmvaeHipError
    ops = _ops()
    with pytest.raises(DmvaeHipError):
        ops.reparam_kl_fwd(torch.zeros(4, 2 * 12, device=DEV), None)       # width not a power of two
    with pytest.raises(DmvaeHipError):
        ops.reparam_kl_fwd(torch.zeros(4, 64), None)                       # CPU tensor: no CPU path


def _tiny_vae(reparameterize, seed=0):
    from dmvae_amd.models.vae import VAE
    torch.manual_seed(seed)
    return VAE(z_channels=32, model_size="base", encoder_kwargs=dict(embed_dim=256, depth=1, num_heads=4), reparameterize=reparameterize).to(DEV)


def test_vae_hook_off_is_the_reference_module():
    """Default keyword: the module tree, the state_dict and the forward are what they were (models/vae.py:72-98): the bottleneck emits z channels,
    nothing is sampled, posterior_kl stays None, and two forwards agree bit for bit."""
    vae = _tiny_vae(False)
    assert vae.bottle_neck.mlp[2].weight.shape == (32, 2048) and vae.reparameterize is False
    x = torch.rand(2, 3, 256, 256, device=DEV) * 2 - 1
    with torch.autocast("cuda", dtype=torch.bfloat16):
        r0, z0 = vae(x, return_latent=True)
        r1, z1 = vae(x, return_latent=True)
    assert z0.shape == (2, 256, 32) and torch.equal(r0, r1) and torch.equal(z0, z1) and vae.posterior_kl is None
    # the hook's entry point is the bottleneck itself when off
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        feats = vae.encoder(x)
        assert torch.equal(vae.latent(feats), vae.bottle_neck(feats))


def test_vae_hook_on_forward_backward_vs_spec():
    """Hook on: z = mu + sigma * eps with the given eps, KL as the spec, and the gradients reaching the bottleneck are the spec's (recon path + KL path)."""
    from dmvae_amd import parity
    with parity.enabled():                                                 # f32 activations: the comparison is at 1e-4, not at bf16 level
        vae = _tiny_vae(True, seed=1)
        assert vae.bottle_neck.mlp[2].weight.shape == (64, 2048)
        vae.train()
        feats = torch.randn(2, 256, 256, device=DEV) * 0.5
        eps = torch.randn(2, 256, 32, device=DEV)
        z = vae.latent(feats, eps=eps)
        mom = vae.bottle_neck(feats).detach()
        zr, klr, klm = R.reparam_kl(mom.double().cpu(), eps.double().cpu())
        assert z.shape == (2, 256, 32) and rel_err(z.detach().cpu(), zr) < TOL
        assert abs(vae.posterior_kl.item() - klm.item()) < TOL * abs(klm.item())
        assert rel_err(vae.posterior_kl_per_latent[:32].cpu(), klr) < TOL
        w = torch.randn_like(z)
        vae.zero_grad(set_to_none=True)
        ((z * w).sum() + 3.0 * vae.posterior_kl).backward()
        got = vae.bottle_neck.mlp[2].bias.grad.detach().cpu().double()
        mr = mom.double().cpu().requires_grad_(True)
        zr, _, klm = R.reparam_kl(mr, eps.double().cpu())
        ((zr * w.double().cpu()).sum() + 3.0 * klm).backward()
        want = mr.grad.reshape(-1, 64).sum(0)                              # d / d bias of the last Linear = column sums of d moments
        assert rel_err(got, want) < TOL
        # eval mode: the posterior mode, no draw
        vae.eval()
        st = torch.cuda.get_rng_state()
        zm = vae.latent(feats)
        assert torch.equal(torch.cuda.get_rng_state(), st)
        assert rel_err(zm.detach().cpu(), mom[..., :32].cpu()) < 1e-6


def test_tokenizer_step_with_hook_trains():
    """TokenizerTrainer over a VAE(reparameterize=True): the posterior KL joins the loss, every parameter of the (mu | logvar) head gets a gradient, the
    loss stays finite and the KL falls under a large weight; posterior_kl_w without the hook is refused."""
    from dmvae_amd.train import TokenizerTrainer
    vae = _tiny_vae(True, seed=2)
    tr = TokenizerTrainer(vae, None, lr=2e-3, warmup_steps=1, posterior_kl_w=50.0)
    gen = torch.Generator(device=DEV).manual_seed(0)
    x = torch.rand(2, 3, 256, 256, device=DEV, generator=gen) * 2 - 1
    kls = []
    for _ in range(6):
        loss = tr.step(x)
        kls.append(float(vae.posterior_kl.detach()))
        assert torch.isfinite(loss)
    assert kls[-1] < kls[1], kls
    with pytest.raises(ValueError):
        TokenizerTrainer(_tiny_vae(False), None, posterior_kl_w=1.0)
