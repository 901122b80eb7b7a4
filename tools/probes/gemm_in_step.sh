# per-call durations of the encoder's GEMM kernels inside the bench step (kernel trace of 3 steps).  usage: bash tools/probes/gemm_in_step.sh <tag>
R=$GRAFT_REPO_ROOT; TAG=${1:-gis}; OUT=$R/gpurun_out/$TAG
mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o step -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_prof.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/prof/**/step_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
sel = rows[ends[-2] + 1: ends[-1] + 1]
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in sel]
# the encoder: from the first layernorm to the last gemm / Cijk before the decoder
idx = [i for i, s in enumerate(seq) if "gemm_pp" in s[0] or "Cijk" in s[0]]
per = collections.defaultdict(list)
k = 0
for i in idx:
    name = seq[i][0]
    short = name[name.find("MT"):name.find("MT") + 12] if "Cijk" in name else name[name.find("<"):name.find(">") + 1]
    gap = (seq[i][2] - seq[i - 1][3]) / 1e3
    per[(k % 4 if k >= 1 else -1, short)].append((seq[i][1], gap))
    k += 1
for key in sorted(per):
    v = per[key]
    print(key, "n", len(v), "avg_us %.1f" % (sum(x[0] for x in v) / len(v)), "min %.1f max %.1f" % (min(x[0] for x in v), max(x[0] for x in v)), "gap before avg %.1f" % (sum(x[1] for x in v) / len(v)))
PY
rm -rf $OUT/prof
