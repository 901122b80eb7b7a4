"""CPU: the C-ABI library builds, loads, and exports exactly the symbols include/dmvae_hip.h declares; argument
validation returns errno-style codes without touching a GPU."""
import ctypes
import os
import re
import subprocess
import sys

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
    assert lib.dmvae_abi_version() == 8


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dmvae_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DmvaeHipError):
        _lib.lib()


_ASAN_SCRIPT = r"""
import ctypes, sys
sys.path.insert(0, %r)
from dmvae_amd import _lib
from dmvae_amd._lib import ConvDesc
lib = _lib.lib()
assert "asan" in _lib.LIB_PATH
maps = open("/proc/self/maps").read()
assert "libclang_rt.asan" in maps and "libdmvae_hip_asan.so" in maps      # the runtime is in the process and the instrumented library is the one bound
buf = ctypes.create_string_buffer(64)
p = ctypes.cast(buf, ctypes.c_void_p)
# argument validation: every call returns before any launch
for d in (ConvDesc(1, 8, 8, 48, 64, 3, 0, 0, 0), ConvDesc(1, 8, 8, 64, 64, 5, 0, 0, 0)):
    assert lib.dmvae_conv2d_nhwc_fwd(p, p, None, None, p, ctypes.byref(d), None) == -22 and len(lib.dmvae_last_error()) > 0
assert lib.dmvae_conv2d_nhwc_fwd(None, p, None, None, p, ctypes.byref(ConvDesc(1, 8, 8, 64, 64, 3, 0, 0, 0)), None) == -22
assert lib.dmvae_kl_mmd(p, p, p, p, None, p, 1 << 20, 2, 16, 16, 16, 1.0, 1.0, None) == -22
assert lib.dmvae_adamw_ema_step(p, p, p, p, None, None, 4, 1e-4, 0.9, 0.95, 1e-8, 0.0, 0, 0.999, None) == -22
assert lib.dmvae_linear_bf16(p, p, None, p, 64, 64, 100, 100, 100, 64, 0, 0, 0, 0, None) == -22       # K not a multiple of 32
assert lib.dmvae_linear_bf16(p, p, None, p, 64, 72, 384, 384, 384, 72, 6, 0, 0, 0, None) == -22       # SwiGLU needs N %% 16 == 0
assert lib.dmvae_swiglu_bf16(p, p, 4, 12, None) == -22
# host-side planning / sizing over a sweep of shapes: the tile menus, workspace formulas and the halo predicate
tc, tr = ctypes.c_int(), ctypes.c_int()
for m in (64, 257, 4096, 8224, 16384, 524288):
    for n in (8, 72, 1024, 1152, 3456, 6144):
        for k in (384, 1024, 3072):
            c = lib.dmvae_linear_bf16_plan(m, n, k, ctypes.byref(tc), ctypes.byref(tr))
            assert 0 <= c < 12 and tc.value in (128, 192, 256) and tr.value %% 32 == 0
for (n, h, w, ci, co, ks) in [(32, 16, 16, 512, 512, 3), (32, 256, 256, 128, 128, 3), (2, 64, 64, 256, 512, 1), (1, 8, 8, 32, 64, 3), (32, 128, 128, 512, 256, 3)]:
    d = ConvDesc(n, h, w, ci, co, ks, 0, 0, 0)
    assert lib.dmvae_conv_halo_applies(ctypes.byref(d)) in (0, 1)
    assert lib.dmvae_conv2d_nhwc_wgrad_workspace(ctypes.byref(d)) > 0
    assert lib.dmvae_conv2d_nhwc_fwd_gnstats_workspace(ctypes.byref(d), 32) >= 0
assert lib.dmvae_groupnorm_workspace(2, 64, 100, 32) == 0 and lib.dmvae_groupnorm_workspace(32, 65536, 128, 32) > 0
assert lib.dmvae_conv_out_wgrad_workspace(32, 256, 256, 128, 3) > 0
assert lib.dmvae_gemm_tn_batched_workspace(64, 64, 4096, 8) >= 0
assert lib.dmvae_vit_bwd_workspace(1024) > 0
# round 5: the whole-stack DiT backward's sizing functions and the batched entry points' argument validation
for bsz in (1, 16, 64, 2048):
    assert 4 <= lib.dmvae_dit_stack_bps(bsz) <= 32
    assert lib.dmvae_dit_stack_part_bytes(28, bsz, 1152) == (2 * 28 + 1) * bsz * lib.dmvae_dit_stack_bps(bsz) * 4 * 1152 * 4
    assert lib.dmvae_dit_stack_workspace(28, bsz, 256, 1152) > 0
assert 0 < lib.dmvae_qknorm_rope_bwd_nblk(16, 256, 16, 72, 96) <= 2048
assert lib.dmvae_dit_boundary_bwd(None, None, None, None, 0, 0, 1e-6, p, None, None, 0, 0, None, p, p, 2, 8, 64, None) == -22      # neither half requested
assert lib.dmvae_dit_boundary_bwd(p, p, p, p, 384, 65, 1e-6, p, None, None, 0, 0, None, p, p, 2, 8, 64, None) == -22                # scale offset not a multiple of 4
assert lib.dmvae_linear_rows_wgrad_batched(p, 0, p, 48, p, None, None, None, 2, 16, 64, 64, 64, 0, None) == -22                      # mp must be 32 or 64
assert lib.dmvae_linear_rows_batched_bf16(p, 0, p, None, p, 0, 2, 65, 64, 64, 64, 64, 64, 0, 0, 0, 0, None) == -22                   # more than 64 rows
assert lib.dmvae_linear_weight_t_kmajor_batched(None, 0, 0, None) == -22
assert lib.dmvae_wt_entry_bytes() == 32
assert lib.dmvae_abi_version() == 8
print("asan-ok")
"""


def test_host_side_under_address_sanitizer():
    """`make -C dmvae_amd/csrc asan` (the sanitizer runs on the CPU build only): the host pass of every source instrumented, loaded under the ASAN
    runtime; argument validation, the tile planners, the workspace formulas and the halo predicate are driven through it -- any out-of-bounds access or
    use-after-free in the host code aborts the child with an AddressSanitizer report."""
    csrc = os.path.join(ROOT, "dmvae_amd", "csrc")
    subprocess.run(["make", "-C", csrc, "asan", "-j8"], check=True, capture_output=True)
    rt = subprocess.run(["make", "-s", "-C", csrc, "asan-runtime"], check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1]
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66", DMVAE_LIB=os.path.join(ROOT, "dmvae_amd", "libdmvae_hip_asan.so"))
    r = subprocess.run([sys.executable, "-c", _ASAN_SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "asan-ok" in r.stdout and "AddressSanitizer" not in r.stderr, (r.returncode, r.stdout[-500:], r.stderr[-3000:])
