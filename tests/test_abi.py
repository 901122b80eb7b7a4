"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/dmvae_hip.h declares; argument
validation returns errno-style codes without touching a GPU."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="session")
def lib():
    from dmvae_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.run(["make", "-C", os.path.join(ROOT, "dmvae_amd", "csrc"), "-j8"], check=True)
    return _lib.lib()


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "dmvae_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(dmvae_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from dmvae_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dmvae_hip.h but not exported by libdmvae_hip.so"
    assert sorted(_lib.SIGNATURES) == syms, "dmvae_amd/_lib.py SIGNATURES out of sync with include/dmvae_hip.h"


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "dmvae_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)            # declarations only
    assert "torch" not in code.lower() and "at::" not in code and "#include <hip" not in code and "std::" not in code


def test_argument_validation_without_gpu(lib):
    from dmvae_amd._lib import ConvDesc
    d = ConvDesc(1, 8, 8, 48, 64, 3, 0, 0, 0)        # Cin not a multiple of 32
    buf = ctypes.create_string_buffer(16)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.dmvae_conv2d_nhwc_fwd(p, p, None, None, p, ctypes.byref(d), None)
    assert rc == -22 and b"multiple of 32" in lib.dmvae_last_error()
    assert lib.dmvae_conv2d_nhwc_fwd(None, p, None, None, p, ctypes.byref(d), None) == -22
    d2 = ConvDesc(1, 8, 8, 64, 64, 5, 0, 0, 0)
    assert lib.dmvae_conv2d_nhwc_fwd(p, p, None, None, p, ctypes.byref(d2), None) == -22
    assert lib.dmvae_groupnorm_workspace(2, 64, 100, 32) == 0        # c % groups != 0 -> unsupported
    assert lib.dmvae_groupnorm_workspace(2, 64, 128, 32) > 0
    assert lib.dmvae_kl_mmd(p, p, p, p, None, p, 1 << 20, 2, 16, 16, 16, 1.0, 1.0, None) == -22   # d != 32
    assert lib.dmvae_adamw_ema_step(p, p, p, p, None, None, 4, 1e-4, 0.9, 0.95, 1e-8, 0.0, 0, 0.999, None) == -22  # step 0
    assert lib.dmvae_abi_version() == 2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dmvae_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DmvaeHipError):
        _lib.lib()
