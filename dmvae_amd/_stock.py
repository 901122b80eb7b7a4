"""Gate for the stock-PyTorch routes of the two transformer mirrors (models/vit.py, models/lightningdit.py).

The HIP routes implement the arithmetic the reference runs these models with -- autocast(bf16) on the GPU -- at the shapes its scripts use.  A call
they do not cover (CPU tensors, no autocast, a width / head size outside the kernels' range) used to drop silently onto the stock nn.Module
forward, i.e. ATen / hipBLASLt kernels, and a test could pass without touching a HIP kernel.  Now such a call raises, unless the caller opts in with
DMVAE_ALLOW_STOCK=1 (checked at call time; one warning per call site).  `forward_stock` / `forward_features_stock` stay callable directly: the
tests use them as the PyTorch reference."""
import os
import warnings

from ._lib import DmvaeHipError

_warned = set()


def require_opt_in(what: str, why: str) -> None:
    if os.environ.get("DMVAE_ALLOW_STOCK", "0") in ("", "0"):
        raise DmvaeHipError(f"{what}: {why}; the HIP route does not cover this call and dmvae_amd does not fall back silently. "
                            "Set DMVAE_ALLOW_STOCK=1 to run the stock PyTorch modules instead (ATen / library kernels, not the HIP path).")
    if what not in _warned:
        _warned.add(what)
        warnings.warn(f"{what}: {why} -- running the STOCK PyTorch modules (DMVAE_ALLOW_STOCK=1), not the HIP kernels", stacklevel=3)
