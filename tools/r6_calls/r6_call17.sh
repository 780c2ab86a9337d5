mkdir -p gpurun_out/r6
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r6/call17_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call17_pytest.log)"
for v in "" "NSKY_FIT_STREAM=0" "" "NSKY_FIT_STREAM=0"; do
  env $v timeout 300 python tools/bench_step.py 30 2>/dev/null | tail -1
done > gpurun_out/r6/call17_ab.log 2>&1
cat gpurun_out/r6/call17_ab.log
FLAKE_RUNS=4 timeout 900 bash tools/flake_seq.sh > gpurun_out/r6/call17_guard.log 2>&1
grep "guard" gpurun_out/r6/call17_guard.log | cut -c1-200
