"""C4 `diffusion_step_small` (oracle/capture_golden_diffusion.py: two steps of the reference's train_diffusion.py:268-297 with its own modules, CPU, fp32): the
oracle's restatement of that loop (oracle/ref_cpu.py::diffusion_train_steps over dino_encoder_forward / mlp_forward / latents_to_dit_input / transport_plan /
lightningdit_forward / clip_grad_norm / adamw_step / ema_update) against the capture -- the normalised latents of the frozen encode, the per-sample losses, the
gradient of every parameter (norm + sum; nine small tensors element by element), both steps' losses and gradient norms, per-tensor parameter / EMA checksums and
the complete update of the nine tensors.  fp32 on both sides: tolerances are f32 summation order (and Adam's amplification of it on near-zero gradients)."""
import numpy as np
import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from oracle.detweights import det_fill_
from test_oracle_golden import vae_tiny_params

DIT_KW = dict(input_size=16, patch_size=1, in_channels=32, hidden_size=192, depth=2, num_heads=3, num_classes=10)
SMALL = ("final_layer.linear.weight", "final_layer.linear.bias", "blocks.1.attn.q_norm.weight", "blocks.0.norm1.weight", "blocks.0.mlp.w3.bias",
         "t_embedder.mlp.2.bias", "x_embedder.proj.weight", "y_embedder.embedding_table.weight", "blocks.1.adaLN_modulation.1.bias")


def diffusion_step_inputs(g):
    """(VAE parameters, the DiT module with the capture's weights, images, labels, draws) of the capture."""
    from dmvae_amd.models.lightningdit import LightningDiT
    pv, _ = vae_tiny_params(seed=int(g["vae_seed"]), width=256)
    dit = LightningDiT(class_dropout_prob=float(g["class_dropout_prob"]), **DIT_KW)
    det_fill_(dit, int(g["dit_seed"]), skip=("pos_embed",))
    assert [str(n) for n in g["names"]] == [n for n, _ in dit.named_parameters()]
    images = torch.rand(int(g["batch"]), 3, 256, 256, generator=torch.Generator().manual_seed(int(g["images_seed"]))) * 2 - 1
    labels = torch.from_numpy(np.asarray(g["labels"]))
    draws = [(g.t(f"t_{s}"), g.t(f"x0_{s}"), torch.from_numpy(np.asarray(g[f"drop_{s}"]))) for s in range(2)]
    return pv, dit, images, labels, draws


def check_diffusion_steps(g, logs, p0, p1, ema, names, tol_loss, tol_norm, tol_abs_delta, tol_signed, min_cos, tol_ema):
    """A replay of both steps against the capture.  p0 / p1 / ema: name -> CPU tensor (initial, after the last step, EMA after the last step)."""
    last = len(logs) - 1
    for s, (loss, norm) in enumerate(logs):
        assert abs(loss - float(g["loss"][s])) < tol_loss * float(g["loss"][s]), (s, loss, float(g["loss"][s]))
        assert abs(norm - float(g["grad_norm"][s])) < tol_norm * float(g["grad_norm"][s]), (s, norm, float(g["grad_norm"][s]))
    ck, eck = g[f"ck{last}"], g[f"ema_ck{last}"]       # per tensor: [sum, sum|.|, sum(delta), sum|delta|] in f64
    for i, k in enumerate(names):
        if k == "pos_embed":
            continue
        d = p1[k].double() - p0[k].double()
        assert abs(p1[k].double().abs().sum().item() - ck[i][1]) < 1e-5 * ck[i][1], k
        assert abs(d.abs().sum().item() - ck[i][3]) < tol_abs_delta * ck[i][3], (k, d.abs().sum().item(), ck[i][3])
        assert abs(d.sum().item() - ck[i][2]) < tol_signed * ck[i][3], (k, d.sum().item(), ck[i][2])
        assert abs(ema[k].double().abs().sum().item() - eck[i][1]) < 1e-5 * eck[i][1], k
    for k in SMALL:
        d, ref = (p1[k] - p0[k]).double().flatten(), g.t(f"d{last}." + k).double().flatten()
        cos = (d @ ref) / (d.norm() * ref.norm())
        assert cos > min_cos, (k, cos.item())
        de, eref = (ema[k] - p0[k]).double().flatten(), g.t(f"dema{last}." + k).double().flatten()
        assert (de - eref).abs().max() < 3e-7 + tol_ema * eref.abs().max(), k


def test_diffusion_step_small_oracle_vs_reference_capture():
    g = load_golden("diffusion_step_small")
    pv, dit, images, labels, draws = diffusion_step_inputs(g)
    with torch.no_grad():
        tokens = R.mlp_forward(R.dino_encoder_forward(images, pv, num_heads=4), pv)             # vae.encode (models/vae.py:100-103), frozen
    x = R.latents_to_dit_input(tokens, float(g["latent_mean"]), float(g["latent_scale"]))
    assert rel_err(x, g.t("latents")) < 2e-5
    p = {k: v.detach().clone() for k, v in dit.state_dict().items()}
    names = [str(n) for n in g["names"]]
    trainable = [n for n in names if n != "pos_embed"]
    p0 = {k: p[k].clone() for k in trainable}
    grads0 = {}

    def on_grads(step, grads):
        if step == 0:
            grads0.update({k: v.clone() for k, v in grads.items()})
    logs, p1, ema = R.diffusion_train_steps(g.t("latents"), labels, p, trainable, draws, DIT_KW["num_heads"], DIT_KW["num_classes"], lr=float(g["lr"]),
                                            on_grads=on_grads)
    for k in trainable:                                  # the first step's gradient of EVERY parameter: norm and sum; nine tensors in full
        want = g["gn0." + k]
        assert abs(grads0[k].double().norm().item() - want[0]) < 2e-4 * want[0] + 1e-9, k
    for k in SMALL:
        assert rel_err(grads0[k], g.t("g0." + k)) < 2e-4, k
    check_diffusion_steps(g, logs, p0, {k: p1[k].detach() for k in trainable}, ema, names, tol_loss=2e-5, tol_norm=2e-4, tol_abs_delta=2e-3, tol_signed=2e-2,
                          min_cos=0.999, tol_ema=5e-3)
    assert bool(np.asarray(g["drop_0"]).any()) and logs[1][0] < logs[0][0]        # a dropped label is in the capture; the loss falls
