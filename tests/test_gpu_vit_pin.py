"""The HIP encoder routes (models/vit_fast.py over csrc/vit.hip, vit_bwd.hip) held to outputs and gradients captured from the REFERENCE's own
models/dinov2.py at widths the kernels accept (tests/golden/vit_w256.npz: embed 256, 4 heads x 64, 2 blocks, 257 tokens; vit_w768.npz: ViT-B
width) -- round 1 compared these routes with the stock PyTorch-ROCm module only, and the reduced-width reference fixtures bypassed them.

bf16 routes (what the production step runs: the reference's autocast arithmetic): as close to the reference's f32 result as the CPU oracle with
bf16 rounding at the same sites is (x 1.5 + 1e-3: two correct bf16 pipelines decorrelate to the rounding floor), and within TOL_Q of that oracle.
fp32 parity mode: 1e-4 against the reference capture."""
import pytest
import torch

from conftest import rel_err
from oracle import ref_cpu as R
from test_oracle_vit import vit_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda"
Q = R.bf16_round
TOL_Q = 1e-2


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _as_close_as_oracle(hip, orc, ref, what, slack=1.5, floor=1e-3):
    e_hip, e_orc = rel_l2(hip, ref), rel_l2(orc, ref)
    print(f"{what}: rel-L2 to the reference's f32 -- HIP {e_hip:.2e}, bf16-site oracle {e_orc:.2e}")
    assert e_hip < slack * e_orc + floor, what


@pytest.mark.parametrize("name", ["vit_w256", "vit_w768"])
def test_frozen_route_vs_reference_capture(name):
    from dmvae_amd.models import vit_fast
    g, vit, p, x = vit_fixture(name)
    vit = vit.to(DEV).eval().requires_grad_(False)
    assert vit_fast.hip_path_supported(vit, 257)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = vit.forward_features(x.to(DEV))                      # module entry point -> frozen_forward_features
        direct = vit_fast.frozen_forward_features(vit, x.to(DEV))
    assert out.dtype == torch.bfloat16 and torch.equal(out, direct)
    with torch.no_grad():
        orc = R.vit_forward_features(x, p, pre="", num_heads=int(g["heads"]), q=Q)
    _as_close_as_oracle(out.float().cpu(), orc, g.t("out"), name + " frozen route")
    assert rel_err(out.float().cpu(), orc) < 2 * TOL_Q


def test_trainable_route_forward_and_gradients_vs_reference_capture():
    from dmvae_amd.models import vit_fast
    g, vit, p, x = vit_fixture("vit_w256")
    vit = vit.to(DEV).train()
    dy = torch.randn(g["out"].shape, generator=torch.Generator().manual_seed(int(g["dy_seed"])))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = vit.forward_features(x.to(DEV))                      # parameters require grad -> trainable_forward_features (VitBlockFn)
    (out.float() * dy.to(DEV)).sum().backward()
    po = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    oo = R.vit_forward_features(x, po, pre="", num_heads=int(g["heads"]), q=Q)
    (oo * dy).sum().backward()
    _as_close_as_oracle(out.float().detach().cpu(), oo.detach(), g.t("out"), "trainable route output")
    grads = dict(vit.named_parameters())
    for n, gn in zip(g["names"], g["gnorm"]):
        n, gn = str(n), float(gn)
        got = grads[n].grad
        assert got is not None, n
        assert abs(got.double().norm().item() - gn) < 5e-2 * gn, (n, got.double().norm().item(), gn)      # bf16 floor of a 2-block backward
    for k in [k for k in g if k.startswith("g.")]:
        _as_close_as_oracle(grads[k[2:]].grad.cpu(), po[k[2:]].grad, g.t(k), "gradient " + k[2:], slack=2.0, floor=5e-3)


@pytest.mark.parametrize("name", ["vit_w256", "vit_w768"])
def test_parity_mode_vs_reference_capture(name):
    from dmvae_amd import parity
    g, vit, p, x = vit_fixture(name)
    vit = vit.to(DEV).eval().requires_grad_(False)
    with parity.enabled(True), torch.no_grad():
        out = vit.forward_features(x.to(DEV))
    assert out.dtype == torch.float32
    assert rel_err(out.cpu(), g.t("out")) < 1e-4


def test_uncovered_width_raises_instead_of_running_stock(monkeypatch):
    import warnings
    from dmvae_amd import _stock
    from dmvae_amd._lib import DmvaeHipError
    from dmvae_amd.models.vit import DinoV2ViT
    monkeypatch.delenv("DMVAE_ALLOW_STOCK", raising=False)
    monkeypatch.setattr(_stock, "_warned", set())          # the warning is issued once per call site per process
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vit = DinoV2ViT(embed_dim=64, depth=1, num_heads=2, patch_size=16, img_size=64).to(DEV).eval().requires_grad_(False)
    x = torch.zeros(1, 3, 64, 64, device=DEV)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        with pytest.raises(DmvaeHipError, match="outside the encoder kernels' range"):
            vit.forward_features(x)
        monkeypatch.setenv("DMVAE_ALLOW_STOCK", "1")
        with pytest.warns(UserWarning, match="STOCK"):
            vit.forward_features(x)


@pytest.mark.parametrize("c", [256, 768, 1024])
def test_scale_residual_layernorm_is_the_two_kernels_in_one_pass(c):
    """dmvae_scale_residual_layernorm (LayerScale + residual add fused into the LayerNorm that follows it; dino_layers/block.py:89-115) against
    scale_residual_ followed by layernorm_bf16: the updated residual stream and the normalised tokens bit for bit, and against f64."""
    from dmvae_amd import ops
    g = torch.Generator().manual_seed(c)
    x = torch.randn(3, 257, c, generator=g).to(DEV)
    r = torch.randn(3, 257, c, generator=g).to(torch.bfloat16).to(DEV)
    ls = (torch.randn(c, generator=g) * 0.1).to(DEV)
    w, b = (torch.rand(c, generator=g) + 0.5).to(DEV), torch.randn(c, generator=g).to(DEV)
    x1 = x.clone()
    y1 = ops.scale_residual_layernorm_(x1, r, ls, w, b, 1e-6)
    x2 = ops.scale_residual_(x.clone(), r, ls)
    y2 = ops.layernorm_bf16(x2, w, b, 1e-6)
    assert torch.equal(x1, x2) and torch.equal(y1, y2)
    xr = x.double() + ls.double() * r.double()
    yr = torch.nn.functional.layer_norm(xr, (c,), w.double(), b.double(), 1e-6)
    assert ((y1.double() - yr).abs().max() / yr.abs().max()).item() < 2.0 ** -8
