mkdir -p gpurun_out/r4c
export LBMPM_DIST_BACKEND=gloo LBMPM_TRANSPORT=ipc
for t in fine coarse fine coarse; do
  LBMPM_IPC_LAND=$t timeout 500 python bench.py --gpus 8 --steps 40 --warmup 5 --no-calibration > gpurun_out/r4c/n8l_$t.json 2> gpurun_out/r4c/n8l_$t.err
  python - $t <<'PY'
import json, sys
t = sys.argv[1]
try:
    d = json.loads([l for l in open("gpurun_out/r4c/n8l_%s.json" % t) if l.startswith("{")][-1])
    m = d["multi_gpu"]
    print(t, d["value"], d["ms_per_step"], m["transport"][:50], m["host_enqueue_us_per_step"], [round(r["step_ms"], 3) for r in m["per_rank"]])
except Exception as e:
    print(t, "failed", e); print(open("gpurun_out/r4c/n8l_%s.err" % t).read()[-1500:])
PY
done
