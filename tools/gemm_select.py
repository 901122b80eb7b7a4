"""NOT on the product path any more (round 3: every Linear layer runs on csrc/gemm_pp.hip through `functional.linear`).  Kept for tools/bench_gemm.py, whose
vendor-library arm should be the library at its best (tools/tuned/*.csv): hipBLASLt solution selection for the four Linear shapes per block of the ViT encoder
(models/vae.py:47-53) through PyTorch's TunableOp with tuning OFF and a committed results file, i.e. a fixed shape -> solution table measured on MI355X
(`PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=...`; tools/tuned/*.csv).  The library's heuristic pick for
fc1 (M = 8224, N = 4096, K = 1024) runs at 0.80 PFLOP/s, the table's at 0.94; the step gains 0.5-0.7 ms.  A table whose validator lines (PyTorch /
hipBLASLt / rocBLAS versions, gfx arch) do not match the running stack is rejected by TunableOp and the heuristic picks stay in force.
DMVAE_GEMM_SELECT=0 turns the table off."""
import glob
import os

import torch

_done = False


def enable() -> bool:
    """Idempotent; returns whether a table is in force."""
    global _done
    if _done or os.environ.get("DMVAE_GEMM_SELECT", "1") == "0" or not torch.cuda.is_available():
        return _done
    _done = True
    if os.environ.get("PYTORCH_TUNABLEOP_ENABLED") == "1":      # the user is tuning / bringing a table of their own: leave TunableOp to them
        return True
    try:
        tun = torch.cuda.tunable
        tun.enable(True)
        tun.tuning_enable(False)
        if hasattr(tun, "write_file_on_exit"):
            tun.write_file_on_exit(False)
        else:                                  # this PyTorch writes its table at exit unconditionally: point it away from the working directory
            import tempfile
            tun.set_filename(os.path.join(tempfile.gettempdir(), "dmvae_tunableop_unused_%d.csv" % os.getpid()))     # one per process (rank)
        ok = False
        for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "*.csv"))):
            ok = bool(tun.read_file(f)) or ok
        if not ok:
            tun.enable(False)
        return ok
    except (AttributeError, RuntimeError):
        return False
