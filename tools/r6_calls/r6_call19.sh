mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_slab_adam.py tests/test_gpu_trainer_surface.py -m gpu -x -q > gpurun_out/r6/call19_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call19_pytest.log)"; grep -a "Error\|assert" gpurun_out/r6/call19_pytest.log | head -10 | cut -c1-250
timeout 900 python bench.py > gpurun_out/r6/call19_bench.json 2> gpurun_out/r6/call19_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r6/call19_bench.json').read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"]); print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["trainer_surface"].items() if k not in ("loop", "optimizer", "launch")})
PY
