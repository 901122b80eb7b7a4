"""C3 `dmd_step_small` (oracle/capture_golden_dmd_step.py: four steps of the reference's train_dmd.py:506-575 with its own modules and VAELossFunction, CPU, fp32):
the oracle's restatement of that loop (oracle/ref_cpu.py::dmd_train_steps) against the capture -- every logged scalar of every turn (L1, L2, LPIPS, rec_loss,
dmd_loss, dmd_gradient_norm, vae_norm, diffusion_loss, sit_norm), the latents, the first VAE turn's gradient of EVERY VAE parameter (the trainable encoder
included) and the first student gradient of every student parameter (norm; fifteen tensors element by element), and per-tensor parameter checksums + the complete
update of the fifteen tensors after the last step.  fp32 on both sides: tolerances are f32 summation order and Adam's amplification of it."""
import numpy as np
import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from oracle.detweights import det_fill_
from test_oracle_golden import lpips_params, vae_tiny_params

DIT_KW = dict(input_size=16, patch_size=1, in_channels=32, hidden_size=192, depth=2, num_heads=3, num_classes=10)
SMALL_VAE = ("decoder.conv_out.weight", "decoder.norm_out.weight", "bottle_neck.mlp.2.bias", "bottle_neck.mlp.0.bias", "decoder.conv_in.1.bias",
             "encoder.model.norm.weight", "encoder.model.blocks.1.ls2.gamma", "encoder.model.blocks.0.attn.proj.bias", "encoder.model.cls_token")
SMALL_SIT = ("final_layer.linear.weight", "blocks.0.norm1.weight", "blocks.1.attn.q_norm.weight", "t_embedder.mlp.2.bias", "y_embedder.embedding_table.weight",
             "blocks.1.adaLN_modulation.1.bias")
REF_NAME = lambda k: k.replace("encoder.model.", "encoder.model.vit.", 1) if k.startswith("encoder.model.") else k      # the capture's adapter inserts 'vit.'


def dmd_step_inputs(g):
    """(VAE parameter dict + module, LPIPS lin weights, teacher / student modules with the capture's weights, images, labels, per-step draws) of the capture."""
    from dmvae_amd.models.lightningdit import LightningDiT
    pv, vae = vae_tiny_params(seed=int(g["vae_seed"]), width=256)
    mk = lambda seed: det_fill_(LightningDiT(class_dropout_prob=float(g["class_dropout_prob"]), **DIT_KW), int(seed), skip=("pos_embed",))
    teacher, student = LightningDiT(class_dropout_prob=float(g["class_dropout_prob"]), **DIT_KW), LightningDiT(class_dropout_prob=float(g["class_dropout_prob"]), **DIT_KW)
    det_fill_(teacher, int(g["teacher_seed"]), skip=("pos_embed",))
    det_fill_(student, int(g["student_seed"]), skip=("pos_embed",))
    images = torch.rand(int(g["batch"]), 3, 256, 256, generator=torch.Generator().manual_seed(int(g["images_seed"]))) * 2 - 1
    labels = torch.from_numpy(np.asarray(g["labels"]))
    draws = []
    for s in range(4):
        d = {"student": (g.t(f"sit_t_{s}"), g.t(f"sit_x0_{s}"), torch.from_numpy(np.asarray(g[f"sit_drop_{s}"])))}
        if f"dmd_t_{s}" in g:
            d["dmd"] = (g.t(f"dmd_t_{s}"), g.t(f"dmd_x0_{s}"))
        draws.append(d)
    return pv, vae, lpips_params(g, "lp."), teacher, student, images, labels, draws


def hyper(g):
    return dict(vit_heads=4, dit_heads=DIT_KW["num_heads"], num_classes=DIT_KW["num_classes"], cfg=float(g["cfg"]), dmd_weight=float(g["dmd_weight"]),
                latent_mean=float(g["latent_mean"]), latent_scale=float(g["latent_scale"]), vae_train_every=int(g["vae_train_every"]), lr=float(g["lr"]),
                diff_lr=float(g["diff_lr"]), wd=float(g["wd"]), warmup_steps=int(g["warmup_steps"]))


def check_logs(g, logs, tol, tol_norm, later=1.0):
    """`later`: factor on both bars from step 2 on -- the first forward passes that see weights moved by an Adam step at full rate: Adam's first update is
    +-lr per element whatever the gradient's size, so entries whose gradient is rounding noise move by +-lr with a sign that is not reproducible; the student
    differs by that much between two f32 implementations, and the DMD loss (a ratio of differences of its velocities) shows it at the 1e-4 level."""
    for s, lg in enumerate(logs):
        for k, v in lg.items():
            want = float(g[f"log{s}.{k}"])
            bar = (tol_norm if k in ("vae_norm", "sit_norm", "dmd_gradient_norm") else tol) * (later if s >= 2 else 1.0)
            assert abs(v - want) < bar * abs(want), (s, k, v, want)
        assert ("dmd_loss" in lg) == (s % int(g["vae_train_every"]) == 0)


def check_checksums(g, which, p0, p1, names, small, tol_abs_delta, tol_signed, min_cos):
    ck = g[which + "ck"]
    row = {str(n): i for i, n in enumerate(g["vae_names" if which == "v" else "student_names"])}
    for k in names:
        i = row[REF_NAME(k) if which == "v" else k]
        if k not in p0 or ck[i][3] == 0:                 # a parameter the loop never moves
            continue
        if k.endswith("attn_1.k.bias"):                  # analytically zero gradient: Adam turns rounding noise into +-lr steps (tests/test_oracle_step.py)
            continue
        d = p1[k].double() - p0[k].double()
        assert abs(d.abs().sum().item() - ck[i][3]) < tol_abs_delta * ck[i][3], (k, d.abs().sum().item(), ck[i][3])
        assert abs(d.sum().item() - ck[i][2]) < tol_signed * ck[i][3], (k, d.sum().item(), ck[i][2])
    for k in small:
        d, ref = (p1[k] - p0[k]).double().flatten(), g.t(which + "d." + REF_NAME(k)).double().flatten()
        assert (d @ ref) / (d.norm() * ref.norm()) > min_cos, k


def test_dmd_step_small_oracle_vs_reference_capture():
    g = load_golden("dmd_step_small")
    pv, vae, lp, teacher, student, images, labels, draws = dmd_step_inputs(g)
    vnames_ref = [str(n) for n in g["vae_names"]]
    vnames = [n for n, _ in vae.named_parameters()]
    # the reference optimises EVERY VAE parameter in this stage, in this order; its ViT stand-in (models/dinov2.py) carries one more, `mask_token`, which is not on
    # the forward path (no gradient, never updated) and which timm's model -- and this build's -- does not have
    assert [REF_NAME(n) for n in vnames] == [n for n in vnames_ref if not n.endswith("mask_token")]
    snames = [str(n) for n in g["student_names"]]
    pt = {k: v.detach().clone() for k, v in teacher.state_dict().items()}
    ps = {k: v.detach().clone() for k, v in student.state_dict().items()}
    v0, s0 = {k: pv[k].clone() for k in vnames}, {k: ps[k].clone() for k in snames}
    vg0, sg0 = {}, {}
    logs, pv1, ps1 = R.dmd_train_steps(images, labels, pv, lp, pt, ps, vnames, snames, draws, **hyper(g),
                                       on_vae_grads=lambda s, gr: vg0.update({k: v.clone() for k, v in gr.items()}) if s == 0 else None,
                                       on_student_grads=lambda s, gr: sg0.update({k: v.clone() for k, v in gr.items()}) if s == 0 else None)
    check_logs(g, logs, tol=3e-5, tol_norm=3e-4, later=10.0)
    for k, gr in vg0.items():
        want = g["vgn0." + REF_NAME(k)]
        if want[0] > 1e-7:
            assert abs(gr.double().norm().item() - want[0]) < 3e-4 * want[0], k
    assert len(vg0) == sum(("vgn0." + REF_NAME(k)) in g for k in vnames)                   # the same parameters received a gradient (not mask_token)
    for k in SMALL_VAE:
        assert rel_err(vg0[k], g.t("vg0." + REF_NAME(k))) < 3e-4, k
    for k, gr in sg0.items():
        want = g["sgn0." + k]
        if want[0] > 1e-7:
            assert abs(gr.double().norm().item() - want[0]) < 3e-4 * want[0], k
    for k in SMALL_SIT:
        assert rel_err(sg0[k], g.t("sg0." + k)) < 3e-4, k
    check_checksums(g, "v", v0, {k: pv1[k].detach() for k in vnames}, vnames, SMALL_VAE, tol_abs_delta=3e-3, tol_signed=3e-2, min_cos=0.999)
    check_checksums(g, "s", s0, {k: ps1[k].detach() for k in snames}, snames, SMALL_SIT, tol_abs_delta=3e-3, tol_signed=3e-2, min_cos=0.999)
