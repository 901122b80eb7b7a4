"""Build-time check of the weight-gradient kernels' K loops (CPU: hipcc cross-compiles without a GPU).

The loops keep a counted LDS-DMA prefetch queue in flight (s_waitcnt vmcnt(N > 0)); a compiler-inserted `s_waitcnt vmcnt(0)` inside them drains it once per
K tile -- that happened silently when the fragment reads were the ds_read_tr16 builtin (DESIGN_HISTORY.md 8.12: the wait-count pass orders every LDS read it can see
behind every earlier LDS-DMA) and cost the 128-row tile 20 %.  tools/loop_waits.py extracts each MFMA loop's wait / barrier / DMA skeleton from the listing."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def _skeleton(src, tmp_path):
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "dmvae_amd", "csrc"),
           "-Wno-unused-value", "-S", "--cuda-device-only", os.path.join(ROOT, "dmvae_amd", "csrc", src), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "loop_waits.py"), str(out)], check=True, capture_output=True, text=True)
    return r.stdout


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("src,kernels", [("conv_wgrad_pp.hip", 8), ("wgrad_thin.hip", 1)])
def test_k_loops_keep_their_dma_queue_in_flight(src, kernels, tmp_path):
    text = _skeleton(src, tmp_path)
    loops = [b for b in text.split("\n_Z") if " mfma" in b.split("\n")[0]]
    assert len(loops) >= kernels, f"expected at least {kernels} MFMA loops in {src}, found {len(loops)}:\n{text[:2000]}"
    for blk in loops:
        head, body = blk.split("\n", 1)
        assert "dma" in body, head                                   # the loop issues LDS-DMA ...
        assert "wait vmcnt(0)" not in body, f"{head}\n{body}"        # ... and never drains the queue
        assert "br_execnz" not in body, f"waterfall loop (readfirstlane ... s_cbranch_execnz) around a DMA issue: {head}\n{body}"
