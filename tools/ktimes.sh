# usage: bash tools/ktimes.sh <script.py> <pattern>  -> mean duration per kernel name matching pattern
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o k -- python $R/$1 > /dev/null 2>&1
python - "$2" <<PY
import csv,collections,sys,glob
rows=list(csv.DictReader(open(glob.glob("/tmp/kt/**/k_kernel_trace.csv",recursive=True)[0])))
agg=collections.OrderedDict()
for r in rows:
    if sys.argv[1] in r["Kernel_Name"]:
        k=(r["Kernel_Name"][:70], r["Grid_Size_X"], r["Grid_Size_Y"])
        a=agg.setdefault(k,[]); a.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in agg.items(): print(f"{sum(v[len(v)//2:])/len(v[len(v)//2:])/1e3:9.2f} us (last half of {len(v)})", k)
PY
