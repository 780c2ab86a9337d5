mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_illumination_attention.py tests/test_gpu_eval_methods.py -m gpu -x -q > gpurun_out/r6/call23_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call23_pytest.log)"
for i in 1 2; do timeout 300 python tools/bench_attention.py 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attention ms_per_step', round(d['ms_per_step'],2), d.get('final_loss'))"; done
