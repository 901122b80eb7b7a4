cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r6_names; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p -o n -- python $GRAFT_REPO_ROOT/tools/probes/hipblaslt_names.py > $O/log.txt 2>&1
python - <<'PY'
import csv, glob, os, re
o = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6_names"
f = (glob.glob(o + "/p/*/n_kernel_trace.csv") + glob.glob(o + "/p/n_kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
pats = {"MT": r"MT(\d+x\d+x\d+)", "GSU": r"GSU(\d+)", "SK": r"_SK(\d+)", "MIWT": r"MIWT(\d+_\d+)", "WG": r"_WG(\d+_\d+_\d+)", "PGR": r"PGR(\d+)", "PLR": r"PLR(\d+)", "LDSB": r"LDSB(\d+)"}
out, prev = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "Cijk" not in n:
        continue
    d = {k: (re.search(p, n).group(1) if re.search(p, n) else "-") for k, p in pats.items()}
    key = (n, r["Grid_Size_X"])
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if key != prev:
        out.append([d, r["Grid_Size_X"], r["Workgroup_Size_X"], "Custom" if n.startswith("Custom") else "", [us]])
        prev = key
    else:
        out[-1][4].append(us)
shapes = ["w3 4096x1152x3072", "d_qkv 4096x1152x3456", "d_w12 4096x1152x6144", "fc2@16x257 4112x1024x4096", "w12 4096x6144x1152", "qkv 4096x3456x1152"]
with open(o + "/names.txt", "w") as fo:
    for s, (d, grid, wg, cust, us) in zip(shapes, out):
        line = f"{s:28s} {min(us):6.1f} us  MT{d['MT']} MIWT{d['MIWT']} GSU{d['GSU']} SK{d['SK']} PGR{d['PGR']} PLR{d['PLR']} grid {int(grid)//int(wg)} x {wg} threads {cust}"
        print(line); fo.write(line + "\n")
PY
rm -rf $O/p
