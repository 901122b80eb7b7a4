"""Loss assembly of the DMVAE train step on the HIP kernels: the build's counterpart of
VAELossFunction.forward_generator (train_tokenizer.py:179-204, train_dmd.py:233-262) and
compute_distribution_matching_loss (train_dmd.py:204-230; toy_example_2d/dmd.py:349-360), plus the build-defined
KL / MMD statistics (no reference counterpart, weights default to 0)."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def latents_to_spatial(tokens: torch.Tensor) -> torch.Tensor:
    """[B, h*w, C] -> [B, C, h, w] (train_dmd.py:408-416, p = 1): a pure permutation, bit-exact."""
    b, t, c = tokens.shape
    h = int(t ** 0.5)
    assert h * h == t
    return tokens.reshape(b, h, h, c).permute(0, 3, 1, 2).contiguous()


class _L1MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, recon, images, w1, w2):
        out, grad = ops.l1_mse(recon.contiguous(), images.contiguous(), w1, w2, need_grad=True)
        ctx.save_for_backward(grad)
        ctx.w = (w1, w2)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g1, g2):
        # grad was built for w1*L1 + w2*L2; the caller combines exactly that, so upstream grads are (w1*g, w2*g)
        (grad,) = ctx.saved_tensors
        w1, w2 = ctx.w
        g = g1 / w1 if w1 != 0 else (g2 / w2 if w2 != 0 else torch.zeros_like(g1))
        return grad * g, None, None, None


def l1_mse(recon: torch.Tensor, images: torch.Tensor, w1: float = 1.0, w2: float = 0.0):
    """(L1, L2) as in F.l1_loss / F.mse_loss; differentiable w.r.t. recon for the combination w1*L1 + w2*L2."""
    return _L1MSE.apply(recon, images, float(w1), float(w2))


class _DMDLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, latents, xt, t, vt, vs, vtu, vsu, cfg, weight_factor):
        out, dl = ops.dmd_post(latents.contiguous(), xt, t, vt, vs, vtu, vsu, cfg=cfg, weight_factor=weight_factor)
        ctx.save_for_backward(dl)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, g, _):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None, None, None, None, None, None, None


def dmd_make_xt(latents: torch.Tensor, x0: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """xt = t*x1 + (1-t)*x0 (ICPlan.plan, diffusion/transport/path.py:114-136)."""
    return ops.dmd_pre(latents.detach().float().contiguous(), x0.float().contiguous(), t.float().contiguous())


def dmd_loss(latents, xt, t, v_teacher, v_student, v_teacher_u=None, v_student_u=None, cfg: float = 1.0, weight_factor: bool = True):
    """-> (loss, log tensor [dmd_loss, dmd_gradient_norm]); gradient reaches `latents` only (dL/dlatents = grad/numel)."""
    f = lambda v: None if v is None else v.detach().float().contiguous()
    loss, log = _DMDLoss.apply(latents.float(), f(xt), f(t), f(v_teacher), f(v_student), f(v_teacher_u), f(v_student_u), float(cfg),
                               bool(weight_factor))
    return loss, log


class _KLMMD(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, y, w_kl, w_mmd):
        kl, mmd, dz = ops.kl_mmd(z.contiguous(), y.contiguous(), w_kl, w_mmd, need_grad=True)
        ctx.save_for_backward(dz)
        ctx.mark_non_differentiable(kl, mmd)
        return w_kl * kl[-1] + w_mmd * mmd.mean(), kl, mmd

    @staticmethod
    def backward(ctx, g, _a, _b):
        (dz,) = ctx.saved_tensors
        return dz * g, None, None, None


def kl_mmd_loss(latent_tokens: torch.Tensor, prior: Optional[torch.Tensor] = None, w_kl: float = 1.0, w_mmd: float = 1.0,
                generator: Optional[torch.Generator] = None):
    """Build-defined distribution-matching statistics on latent tokens [B, T, 32] (per-rank statistics under DP):
    returns (w_kl*mean_c KL_c + w_mmd*mean_g MMD^2_g, per-latent KL [33], per-image MMD^2 [B])."""
    z = latent_tokens.float()
    if prior is None:
        prior = torch.randn(z.shape, device=z.device, dtype=torch.float32, generator=generator)
    return _KLMMD.apply(z, prior.float(), float(w_kl), float(w_mmd))


# ---- adversarial branch (train_tokenizer.py:190-227) -------------------------------------------------------------------------------
def generator_gan_term(rec_loss: torch.Tensor, recon: torch.Tensor, disc: torch.nn.Module, daug, last_layer: torch.Tensor,
                       disc_weight: float = 0.5):
    """VAELossFunction.forward_generator's discriminator branch (train_tokenizer.py:190-203): the discriminator is frozen and in eval
    mode (BatchNorm running statistics), g = -mean D(aug(recon)); the adaptive weight is disc_weight * clamp(|d rec/d last| /
    (|d g/d last| + 1e-6), 0, 1e4) at the decoder's last layer.  Returns (rec_loss + d_weight * g, d_weight).

    With direct flat-buffer gradients (optim.FlatParams.enable_direct_grads) each autograd.grad call leaves its result in
    `last_layer`'s gradient slot, so each is reduced to its norm before the next call overwrites the slot (same stream: the order
    holds); the caller's final backward rewrites the slot with the full gradient."""
    disc.eval()
    for p in disc.parameters():
        p.requires_grad_(False)
    g_loss = -disc(daug.aug(recon, 0)).float().mean()
    n_rec = torch.autograd.grad(rec_loss, last_layer, retain_graph=True)[0].detach().norm()
    n_gan = torch.autograd.grad(g_loss, last_layer, retain_graph=True)[0].detach().norm()
    d_weight = (n_rec / (n_gan + 1e-6)).clamp_(0.0, 1e4) * disc_weight
    return rec_loss + g_loss * d_weight, d_weight


GAN_SINGLE_PASS = True      # TokenizerTrainer.step: generator_gan_backward instead of generator_gan_term + loss.backward() (tests compare the two)


def generator_gan_backward(rec_loss: torch.Tensor, recon: torch.Tensor, disc: torch.nn.Module, daug, last_layer: torch.Tensor, disc_weight: float = 0.5,
                           extra: Optional[torch.Tensor] = None):
    """The generator's step of the adversarial phase, BACKWARD INCLUDED, with one pass through each loss network: what `generator_gan_term` followed by
    `loss.backward()` computes (train_tokenizer.py:190-203 + :414) -- gradients of rec_loss + d_weight * g with d_weight = disc_weight * clamp(|d rec / d last| /
    (|d g / d last| + 1e-6), 0, 1e4) held constant -- evaluated as

        g_rec = d rec_loss / d recon          (ONE backward through LPIPS / L1 / MSE)
        g_gan = d g / d recon                 (ONE backward through the frozen discriminator and DiffAug)
        |d . / d last| from recon's own backward node, fed g_rec and g_gan in turn        (the decoder's last layer: two launches of its weight gradient)
        backward of the decoder from  g_rec + d_weight * g_gan  (+ `extra`, a scalar term that does not run through recon: KL / MMD)

    instead of the reference's three backward sweeps over the loss networks (autograd.grad twice, then backward): the gradients are linear in the image
    gradient, so the sums are the same -- up to where d_weight multiplies (the image gradient here, the scalar loss there: the discriminator's bf16 backward
    rounds a scaled copy of the same numbers).  `extra` joins the final backward.  Returns (rec_loss + d_weight * g, d_weight), both detached."""
    disc.eval()
    for p in disc.parameters():
        p.requires_grad_(False)
    g_loss = -disc(daug.aug(recon, 0)).float().mean()
    g_rec = torch.autograd.grad(rec_loss, recon, retain_graph=True)[0]
    g_gan = torch.autograd.grad(g_loss, recon)[0]
    # with direct flat-buffer gradients each call leaves its result in `last_layer`'s gradient slot: reduced to its norm before the next call overwrites it
    from . import functional as _Fn
    with _Fn.tail_weight_only():      # recon's node computes the last layer's weight gradient and nothing else for these two calls
        n_rec = torch.autograd.grad(recon, last_layer, grad_outputs=g_rec, retain_graph=True)[0].detach().norm()
        n_gan = torch.autograd.grad(recon, last_layer, grad_outputs=g_gan, retain_graph=True)[0].detach().norm()
    d_weight = (n_rec / (n_gan + 1e-6)).clamp_(0.0, 1e4) * disc_weight
    total = g_rec + g_gan * d_weight
    if extra is not None:
        torch.autograd.backward([recon, extra], [total, torch.ones_like(extra)])
    else:
        recon.backward(total)
    return rec_loss.detach() + g_loss.detach() * d_weight, d_weight


def discriminator_loss(images: torch.Tensor, recon: torch.Tensor, disc: torch.nn.Module, daug, bcr_strong_aug, bcr_weight: float = 1.0):
    """VAELossFunction.forward_discriminator (train_tokenizer.py:207-227): hinge loss on D(aug([images; recon])) plus
    bcr_weight * mse(D(strong_aug(.)), D(aug(.))), the discriminator in train mode for both passes.  Returns (loss, log) with the log
    entries as device scalars (no host sync): d_loss, bcr_loss, acc_real, acc_fake."""
    for p in disc.parameters():
        p.requires_grad_(True)
    disc.train()
    bs = images.shape[0]
    both = torch.cat([images, recon], dim=0)
    logits = disc(daug.aug(both, 0.0)).float()
    lr_, lf_ = logits[:bs], logits[bs:]
    d_loss = 0.5 * (torch.relu(1.0 - lr_).mean() + torch.relu(1.0 + lf_).mean())
    logits2 = disc(bcr_strong_aug.aug(both, 0.0)).float()
    l_bcr = torch.nn.functional.mse_loss(logits2, logits) * bcr_weight
    log = {"d_loss": d_loss.detach(), "bcr_loss": l_bcr.detach(), "acc_real": (lr_.detach() > 0).float().mean() * 100,
           "acc_fake": (lf_.detach() < 0).float().mean() * 100}
    return d_loss + l_bcr, log
