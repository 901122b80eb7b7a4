#!/usr/bin/env python
"""Run one of the reference's scripts on the MI355X kernels without editing it.

    python run_on_mi355x.py /path/to/dmvae/train_tokenizer.py --local_bs 32 ...
    torchrun --nproc-per-node 8 run_on_mi355x.py /path/to/dmvae/train_dmd.py ...
    python run_on_mi355x.py --check            # print what is shadowed and exit

The reference's drivers import their model code by module path (`from models.vae import VAE`, `from utils.lpips import LPIPS`,
`from diffusion.lightningdit.lightningdit import LightningDiT_models`, `from diffusion.transport import create_transport`; train_tokenizer.py:10-17,
train_dmd.py:10-17, train_diffusion.py:15-17, sample_50k.py:6-14).  `install_shadow()` registers this build's mirrors under exactly those module paths --
same class names, constructor arguments, forward signatures and state_dict keys, HIP kernels underneath -- before the script runs.  Everything else the
scripts import (utils.dist, utils.build_dataset, evaluation.*, models.dinodisc, models.init_param's callers ...) still resolves to the reference's own
files: the shadow packages keep the reference's directories on their __path__."""
from __future__ import annotations

import importlib
import os
import runpy
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

# module path the reference imports -> this build's module
SHADOWS = {
    "models.vae": "dmvae_amd.models.vae",                                  # VAE, DINOEncoder, MLP            (models/vae.py)
    "models.flux_ae": "dmvae_amd.models.flux_ae",                          # Decoder, Encoder, ResnetBlock...  (models/flux_ae.py)
    "models.init_param": "dmvae_amd.models.init_param",                    # init_weights                      (models/init_param.py)
    "models.patchgan": "dmvae_amd.models.patchgan",                        # NLayerDiscriminator               (models/patchgan.py)
    "utils.lpips": "dmvae_amd.utils.lpips",                                # LPIPS                             (utils/lpips.py)
    "utils.diffaug": "dmvae_amd.utils.diffaug",                            # DiffAug                           (utils/diffaug.py)
    "diffusion.lightningdit.lightningdit": "dmvae_amd.models.lightningdit",   # LightningDiT, LightningDiT_models (diffusion/lightningdit/lightningdit.py)
    "diffusion.transport": "dmvae_amd.transport",                          # create_transport, Transport, Sampler (diffusion/transport/__init__.py)
}


def _package(name: str, ref_dir: str | None) -> types.ModuleType:
    """A namespace-like package module whose __path__ is the reference's directory of that name (so that sub-modules this build does not shadow still
    import from the reference), registered in sys.modules."""
    mod = sys.modules.get(name)
    if mod is None or not hasattr(mod, "__path__"):
        mod = types.ModuleType(name)
        mod.__path__ = []
        sys.modules[name] = mod
    if ref_dir:
        d = os.path.join(ref_dir, *name.split("."))
        if os.path.isdir(d) and d not in mod.__path__:
            mod.__path__.append(d)
    return mod


def install_shadow(ref_dir: str | None = None) -> dict:
    """Register the mirrors; returns {shadowed module path: module}.  `ref_dir`: root of the reference checkout (None: only the shadowed modules resolve)."""
    out = {}
    for path, target in SHADOWS.items():
        parts = path.split(".")
        for i in range(1, len(parts)):
            _package(".".join(parts[:i]), ref_dir)
        mod = importlib.import_module(target)
        sys.modules[path] = mod
        setattr(sys.modules[".".join(parts[:-1])], parts[-1], mod)
        out[path] = mod
    models = sys.modules["models"]
    # models/__init__.py re-exports three names (train_dmd.py:10: `from models import VAE, DinoDisc, NLayerDiscriminator`)
    models.VAE = out["models.vae"].VAE
    models.NLayerDiscriminator = out["models.patchgan"].NLayerDiscriminator

    def _lazy(name):          # DinoDisc stays the reference's own (out of scope here): imported from its file on first use
        if name == "DinoDisc":
            return importlib.import_module("models.dinodisc").DinoDisc
        raise AttributeError(f"module 'models' has no attribute {name!r}")
    models.__getattr__ = _lazy
    return out


def main(argv) -> int:
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 0
    if argv[0] == "--check":
        ref = argv[1] if len(argv) > 1 else None
        for path, mod in install_shadow(ref).items():
            print(f"{path:40s} -> {mod.__name__}")
        return 0
    script = os.path.abspath(argv[0])
    ref = os.path.dirname(script)
    install_shadow(ref)
    if ref not in sys.path:
        sys.path.insert(0, ref)                     # what `python script.py` would have put first
    sys.argv = [script] + list(argv[1:])
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    raise SystemExit(main(sys.argv[1:]))
