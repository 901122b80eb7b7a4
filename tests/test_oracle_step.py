"""G12 `step_small`: the oracle's restatement of the tokenizer train loop (oracle/ref_cpu.py::tokenizer_train_steps) against the
reference's own VAE / LPIPS / forward_generator / clip_grad_norm_ / AdamW / LambdaLR / update_ema run for four steps by
oracle/capture_golden_step.py -- loss trajectory, gradient norms, learning rates, per-tensor parameter and EMA checksums, and the complete
update of eight small tensors after every step.  CPU, fp32 both sides: tolerances are f32 summation order.  Three of the four captured steps
are replayed here (each costs ~30 s on 8 cores); tests/test_gpu_train_step.py replays all four on the HIP path."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import ref_cpu as R
from test_oracle_golden import lpips_params, vae_tiny_params

SMALL = ("decoder.conv_out.weight", "decoder.conv_out.bias", "decoder.norm_out.weight", "decoder.norm_out.bias",
         "bottle_neck.mlp.2.weight", "bottle_neck.mlp.2.bias", "decoder.conv_in.1.weight", "decoder.mid.attn_1.norm.weight")


def step_small_inputs(g):
    """Weights (name-seeded fill), trainable names in the reference's order, and the batch of the capture."""
    p, vae = vae_tiny_params(seed=int(g["vae_seed"]), width=int(g["width"]) if "width" in g else 64)
    names = [str(n) for n in g["names"]]
    assert names == [n for n, _ in vae.named_parameters() if not n.startswith("encoder.")]       # same trainable set, same order as the reference
    images = torch.rand(int(g["batch"]), 3, 256, 256, generator=torch.Generator().manual_seed(int(g["images_seed"]))) * 2 - 1
    return p, vae, names, images


def check_step_small(g, logs, p0, p1, ema, names, last, tol_loss, tol_norm, tol_abs_delta, tol_signed, min_cos, tol_ema):
    """Compare a replay of steps 0..last with the capture.  p0 / p1 / ema: name -> CPU tensor (initial, after step `last`, EMA after step `last`)."""
    for s in range(last + 1):
        assert abs(logs[s]["lr"] - float(g["lr"][s])) < 1e-12
        for k in ("L1", "L2", "LPIPS", "rec_loss"):
            assert abs(logs[s][k] - float(g[k][s])) < tol_loss * abs(float(g[k][s])), (s, k, logs[s][k], float(g[k][s]))
        assert abs(logs[s]["vae_norm"] - float(g["vae_norm"][s])) < tol_norm * float(g["vae_norm"][s]), (s, logs[s]["vae_norm"], float(g["vae_norm"][s]))
    ck, eck = g[f"ck{last}"], g[f"ema_ck{last}"]       # per tensor: [sum, sum|.|, sum(delta), sum|delta|] in f64
    for i, k in enumerate(names):
        if k.endswith("attn_1.k.bias"):
            # a bias on the keys shifts every score of a softmax row by the same q.b: its gradient is analytically ZERO, numerically ~1e-12 of
            # rounding noise, and Adam turns the sign of that noise into +-lr steps -- not a reproducible quantity on either side
            continue
        d = p1[k].double() - p0[k].double()
        assert abs(p1[k].double().abs().sum().item() - ck[i][1]) < 1e-6 * ck[i][1], k
        assert abs(d.abs().sum().item() - ck[i][3]) < tol_abs_delta * ck[i][3], (k, d.abs().sum().item(), ck[i][3])    # lr x |Adam direction + decay|
        assert abs(d.sum().item() - ck[i][2]) < tol_signed * ck[i][3], (k, d.sum().item(), ck[i][2])                    # sign flips of near-zero gradients move it
        # EMA: a 1e-4 blend of updates of ~1e-6 is below one f32 ulp of the weights, so its DELTA is rounding on both sides; what is pinned is the value
        assert abs(ema[k].double().abs().sum().item() - eck[i][1]) < 1e-6 * eck[i][1], k
    for k in SMALL:          # the whole update, element by element
        d, ref = (p1[k] - p0[k]).double().flatten(), g.t(f"d{last}." + k).double().flatten()
        cos = (d @ ref) / (d.norm() * ref.norm())
        assert cos > min_cos, (k, cos.item())
        de, eref = (ema[k] - p0[k]).double().flatten(), g.t(f"dema{last}." + k).double().flatten()
        assert (de - eref).abs().max() < 3e-7 + tol_ema * eref.abs().max(), k       # within an ulp or two of O(0.1 .. 1.6) weights


@pytest.mark.parametrize("fixture", ["step_small", "step_small_w256"])
def test_step_small_trajectory_matches_reference_capture(fixture):
    """Both captures of oracle/capture_golden_step.py: the ViT stand-in at embed 64 and at embed 256 (--width 256: the width the bf16 step test needs)."""
    g = load_golden(fixture)
    p, _, names, images = step_small_inputs(g)
    lp = lpips_params(g, "lp.")
    p0 = {k: p[k].clone() for k in names}
    grads0 = {}

    def on_grads(step, grads):
        if step == 0:
            grads0.update({k: grads[k].clone() for k in SMALL})

    last = 2
    logs, p1, ema = R.tokenizer_train_steps(images, p, lp, names, last + 1, base_lr=float(g["base_lr"]), warmup_steps=int(g["warmup_steps"]),
                                            num_heads=4, on_grads=on_grads)
    # the reference's update_ema also walks the frozen encoder: (p * 0.9999 + p * 0.0001) in f32 is p up to one rounding, nothing else
    assert float(g["encoder_moved"]) < 1e-6
    assert logs[0]["rec_loss"] == logs[1]["rec_loss"]            # LambdaLR: the first optimiser step runs at lr 0 (train_tokenizer.py:385-392)
    assert logs[2]["rec_loss"] < logs[1]["rec_loss"]
    for k in SMALL:
        assert rel_err(grads0[k], g.t("g0." + k)) < 2e-4, k
    check_step_small(g, logs, p0, {k: p1[k].detach() for k in names}, ema, names, last, tol_loss=3e-5, tol_norm=2e-4, tol_abs_delta=2e-3,
                     tol_signed=2e-2, min_cos=0.999, tol_ema=5e-3)
