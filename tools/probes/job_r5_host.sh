cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_host
python tools/probes/host_ahead.py 2>&1 | grep -v amdgpu | tee gpurun_out/r5_host/host_ahead.txt
python bench.py --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/r5_host/bench_line.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_host/bench_line.json").read())
print("C2", d["ms_per_step"], "c3", {k: d["c3_dmd_cycle"][k] for k in ("ms_per_step", "vae_turn_ms", "student_ms")}, "c4", d["c4_diffusion_step"]["ms"], "gan", d["gan_step"]["ms"])
PY
