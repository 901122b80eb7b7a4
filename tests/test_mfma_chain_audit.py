"""tools/check_mfma_chain.py on the two sources that issue MFMAs of more than one opcode (csrc/groupnorm.hip's decoder-tail backward, csrc/wgrad_thin.hip): no
accumulate chain through two different MFMA opcodes with fewer than 8 wait states in between -- the sequence hipcc emits without padding and the hardware does
not interlock (profiles/r4_mfma_mixed_chain_probe.txt).  Cross-compiles to assembly only: runs without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_mixed_opcode_mfma_chain_in_the_kernels_that_mix_opcodes():
    files = [os.path.join(ROOT, "dmvae_amd", "csrc", f) for f in ("groupnorm.hip", "wgrad_thin.hip")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mfma_chain.py")] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "groupnorm.hip: 32 MFMAs audited" in r.stdout, r.stdout
