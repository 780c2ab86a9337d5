mkdir -p gpurun_out/r6
S=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6/call27_bench.json 2> gpurun_out/r6/call27_bench.err; rc=$?
E=$(date +%s); echo "bench rc=$rc wall=$((E-S)) s"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6/call27_bench.json').read().strip().splitlines()[-1])
print("ms_per_step", round(d["ms_per_step"], 3), "value", round(d["value"]), "frac", round(d["roofline"]["frac"], 3), "traffic", d["roofline"]["traffic"])
print("top3", [(t["kernel"], round(t["ms_per_step"], 2), round(t.get("mfma_busy") or 0, 3)) for t in d["roofline"]["top3"]])
print("trainer", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d["trainer_surface"].items() if k.endswith("ms_per_step")})
print("attention", round(d["attention_decoder"]["ms_per_step"], 2), "render", round(d["render_1080p"]["ms_per_frame"]), d["render_1080p"].get("hbm"))
print("device", d["device_state"]); print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
