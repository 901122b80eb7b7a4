"""Same-box A/B of a module-level switch under the unmodified bench: python tools/probes/ab_flag.py <module> <ATTR> <0|1> [bench.py arguments...]"""
import importlib
import os
import runpy
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
mod, attr, val = sys.argv[1], sys.argv[2], sys.argv[3]
setattr(importlib.import_module(mod), attr, bool(int(val)))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[4:]
runpy.run_path(sys.argv[0], run_name="__main__")
