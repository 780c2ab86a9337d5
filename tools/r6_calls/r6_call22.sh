mkdir -p gpurun_out/r6
bash tools/evidence.sh r06z > gpurun_out/r6/call22_evidence.log 2>&1
tail -4 gpurun_out/r6/call22_evidence.log | cut -c1-200
python - <<'PY'
import json
d = json.loads(open('gpurun_out/ev_r06z/bench_line.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "device_state", d["device_state"])
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["trainer_surface"].items() if k not in ("loop", "optimizer", "launch")})
PY
