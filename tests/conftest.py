import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def allow_stock(monkeypatch):
    """Opt in to the stock-PyTorch routes of the ViT / LightningDiT mirrors (dmvae_amd/_stock.py) for a test whose SUBJECT is host logic around the
    model on the CPU (sampler / transport mirrors, state_dict layout) or a reduced reference fixture whose width the HIP kernels do not cover.
    Without it an implicit fallback raises."""
    import warnings
    monkeypatch.setenv("DMVAE_ALLOW_STOCK", "1")
    with warnings.catch_warnings():
        warnings.filterwarnings("ignore", message=".*STOCK PyTorch modules.*")
        yield


class Golden(dict):
    def t(self, k, dtype=torch.float32):
        return torch.from_numpy(np.asarray(self[k])).to(dtype)

    def sub(self, prefix):
        return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in self.items() if k.startswith(prefix)}


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return Golden({k: z[k] for k in z.files})


@pytest.fixture
def golden():
    return load_golden


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def elem_err(a, b):
    """Element-wise form of rel_err: max over the elements of |a - b| / (|b| + rms(b)) -- every element is held to the tolerance at ITS OWN magnitude
    (floored by the tensor's RMS, so that elements near zero are not asked for more digits than f32 accumulation has), not at the tensor's largest one."""
    a, b = a.double(), b.double()
    rms = b.pow(2).mean().sqrt().clamp_min(1e-30)
    return ((a - b).abs() / (b.abs() + rms)).max().item()
