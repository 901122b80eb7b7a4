"""The committed hipBLASLt solution table (dmvae_amd/tuned/, dmvae_amd/gemm_select.py) for the frozen encoder's Linear GEMMs (models/vae.py:47-53)."""
import glob
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tables():
    return sorted(glob.glob(os.path.join(ROOT, "dmvae_amd", "tuned", "*.csv")))


def test_table_is_wellformed_and_covers_the_vit_l_shapes():
    files = _tables()
    assert files
    rows = [l.strip().split(",") for f in files for l in open(f) if l.strip()]
    validators = {r[1] for r in rows if r[0] == "Validator"}
    assert {"PT_VERSION", "HIPBLASLT_VERSION", "GCN_ARCH_NAME"} <= validators
    shapes = {r[1] for r in rows if r[0] != "Validator"}
    assert all(len(r) == 4 and float(r[3]) > 0 for r in rows if r[0] != "Validator")
    # qkv, proj, fc1, fc2 of ViT-L at 32 x 257 tokens: (N, M, K)
    for n, k in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        assert f"tn_{n}_8224_{k}_ld_{k}_{k}_{n}" in shapes


def test_enable_is_a_noop_without_a_gpu(monkeypatch):
    from dmvae_amd import gemm_select
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    monkeypatch.setattr(gemm_select, "_done", False)
    assert gemm_select.enable() is False


@pytest.mark.gpu
def test_table_loads_on_the_stack_it_was_measured_on():
    from dmvae_amd import gemm_select
    tun = torch.cuda.tunable
    torch.zeros(1, device="cuda")
    mine = dict(tun.get_validators())
    want = {l.split(",")[1]: l.strip().split(",", 2)[2] for l in open(_tables()[0]) if l.startswith("Validator")}
    ok = gemm_select.enable()
    if os.environ.get("DMVAE_GEMM_SELECT", "1") == "0" or os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1":
        pytest.skip("table switched off / user-driven TunableOp")
    if all(mine.get(k) == v for k, v in want.items()):
        assert ok and len(tun.get_results()) >= 4 and not tun.tuning_is_enabled()
    else:
        assert not ok          # another stack: TunableOp rejects the table, the library's own picks stay in force
