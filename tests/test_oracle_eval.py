"""The tokenizer's eval() without FID (train_tokenizer.py:324-367): the oracle's restatement and the product's `dmvae_amd.evaluate.evaluate` against the numbers
the REFERENCE function logged (tests/golden/eval_metrics.npz, oracle/capture_golden_eval.py: the reference's eval() run with a stub FID over a deterministic
stand-in VAE that is rebuilt here from the stored seeds).  f32 reductions of O(1e5..1e6) elements: 1e-5 relative."""
import torch

from conftest import load_golden
from oracle import ref_cpu as R
from oracle.capture_golden_eval import StandInVAE, batches

TOL = 1e-5


def _setup():
    g = load_golden("eval_metrics")
    vae = StandInVAE(int(g["seed_vae"]))
    data = batches(int(g["seed_img"]), [int(b) for b in g["batches"]])
    return g, vae, data


def _close(a, b):
    return abs(a - b) <= TOL * abs(b)


def test_oracle_eval_metrics_vs_reference_capture():
    g, vae, data = _setup()
    r = R.eval_metrics(vae.encode, vae.decode, [x for x, _ in data], int(g["num_samples"]))
    for k in ("PSNR", "latent_mean", "latent_scale"):
        assert _close(r[k], float(g[k])), (k, r[k], float(g[k]))


def test_evaluate_vs_reference_capture_and_latent_stats_wiring():
    from dmvae_amd import evaluate as E
    g, vae, data = _setup()
    r = E.evaluate(vae, data, int(g["num_samples"]))                  # (images, labels) pairs, like eval_data.dataloader
    for k in ("PSNR", "latent_mean", "latent_scale"):
        assert _close(r[k], float(g[k])), (k, r[k], float(g[k]))
    assert r["batches"] == 3 and r["images"] == 8 and r["FID"] is None
    assert vae.training                                               # train_tokenizer.py:367
    # bare image batches work too; a fid object with the reference's protocol is driven with [0, 1] images, fake first (train_tokenizer.py:353-354)
    calls = []

    class Fid:
        def update(self, imgs, real):
            calls.append((bool(real), float(imgs.min()), float(imgs.max())))

        def compute(self):
            return torch.tensor(12.5)
    r2 = E.evaluate(vae, [x for x, _ in data], int(g["num_samples"]), fid=Fid())
    assert r2["FID"] == 12.5 and [c[0] for c in calls] == [False, True] * 3 and all(-1e-6 <= lo and hi <= 1 + 1e-6 for _, lo, hi in calls)
    assert _close(r2["PSNR"], r["PSNR"])
    # what the later stages take: the constructor keywords of DMDTrainer / DiffusionTrainer / SamplePipeline
    import inspect
    from dmvae_amd import sample, train
    kw = E.latent_stats(r)
    assert set(kw) == {"latent_mean", "latent_scale"} and _close(kw["latent_scale"], float(g["latent_scale"]))
    for ctor in (train.DMDTrainer.__init__, train.DiffusionTrainer.__init__, sample.SamplePipeline.__init__):
        assert set(kw) <= set(inspect.signature(ctor).parameters), ctor


def test_psnr_matches_the_reference_formula():
    from dmvae_amd.evaluate import psnr
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(3, 3, 8, 8, generator=g), torch.rand(3, 3, 8, 8, generator=g)
    want = sum(-10 * torch.log10(((a[i] - b[i]) ** 2).mean()) for i in range(3))
    assert torch.allclose(psnr(a, b, "sum"), want, rtol=1e-6) and torch.allclose(psnr(a, b, "mean"), want / 3, rtol=1e-6)
