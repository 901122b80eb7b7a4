R=$GRAFT_REPO_ROOT; cd $R; OUT=$R/gpurun_out/r5_full; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench.log 2>&1; echo "bench rc=$?" >> $OUT/bench.log
tail -2 $OUT/bench.log | cut -c1-300
python - <<'PY'
import json,sys
ln=[l for l in open(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r5_full/bench.log") if l.startswith("{")]
d=json.loads(ln[-1])
for k in ("value","ms_per_step","c3_dmd_cycle","c4_diffusion_step","gan_step","secondary_error"):
    print(k, json.dumps(d.get(k))[:700])
print("roofline.frac", d["roofline"]["frac"], "wgrad", d["roofline_wgrad"]["frac"], "linear", d["linear_gemm"]["frac"], "cpu", d.get("cpu_baseline",{}).get("value"))
PY
