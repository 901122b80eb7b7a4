#!/usr/bin/env python
"""Appendix P of DESIGN.md: one line per file under profiles/ (description by name pattern), and a check that every file is cited exactly once in DESIGN.md.
    python tools/design_profiles_index.py --write    regenerate the appendix in place (everything after the '## Appendix P' heading's blank line)
    python tools/design_profiles_index.py            check only (exit 1 on a missing / duplicated citation or a [P:key] without a file)"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RULES = [
    (r"_bench_line.*\.json$", "the bench.py JSON line of that tree / box"),
    (r"kernel_stats_rocprofv3.*\.csv$", "rocprofv3 --kernel-trace --stats summary of the named command"),
    (r"trace_summary.*\.txt$", "per-step kernel / family table cut from the rocprofv3 kernel trace"),
    (r"traffic.*\.json$", "HBM bytes per launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes)"),
    (r"traffic.*\.txt$|_pmc.*\.txt$", "PMC counter summary (HBM traffic / LDS conflicts) of the named kernel(s)"),
    (r"step_shapes\.txt$", "per-shape time and TFLOP/s of every conv / wgrad / GEMM call of one step"),
    (r"_ab\.txt$", "same-box A/B of the named change (alternating runs)"),
    (r"gpu_suite\.txt$", "tail of `pytest -m gpu` on that tree"),
    (r"secondary_benches\.txt$", "wall-clock lines of the secondary bench tools"),
    (r"gemm.*\.txt$", "Linear GEMM shapes: gemm_pp tiles / plan vs hipBLASLt"),
    (r"power_ceiling\.txt$", "hipBLASLt 8192^3 bf16, random vs zero data: the practical MFMA ceiling"),
    (r"probe|microbench|bound|per_launch|_us\.txt$", "kernel micro-benchmark / probe output"),
    (r"sample50k", "sampler loop (250-step SDE + decode) timings"),
    (r"soak", "long run of a stage: losses finite, memory and device-table count flat"),
    (r"after_|passes", "kernel micro-benchmark after the named change (tools/bench_*.py)"),
]
def describe(name):
    for pat, d in RULES:
        if re.search(pat, name):
            return d
    return "evidence file"
def main():
    files = sorted(os.listdir(os.path.join(ROOT, "profiles")))
    path = os.path.join(ROOT, "DESIGN.md")
    s = open(path).read()
    head = "## Appendix P"
    i = s.index(head)
    j = s.index("\n\n", i) + 2
    if "--write" in sys.argv:
        lines = ["| file (key = name without extension) | what |", "|---|---|"]
        for f in files:
            lines.append(f"| `profiles/{f}` | {describe(f)} |")
        s = s[:j] + "\n".join(lines) + "\n"
        open(path, "w").write(s)
    bad = 0
    for f in files:
        n = s.count("profiles/" + f)
        if n != 1:
            print(f"profiles/{f}: cited {n} times"); bad += 1
    keys = {os.path.splitext(f)[0] for f in files}
    for k in set(re.findall(r"\[P:([\w.\-]+?)\]", s)) | set(re.findall(r"P:([\w\-]+)\]", s)):
        if k not in keys:
            print(f"[P:{k}] has no file under profiles/"); bad += 1
    print(f"{len(files)} files, DESIGN.md {len(s.encode())} bytes, longest line {max(len(l) for l in s.splitlines())}")
    sys.exit(1 if bad else 0)
main()
