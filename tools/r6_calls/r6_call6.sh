mkdir -p gpurun_out/r6
for v in "" "NSKY_FIT_STREAM=0" "" "NSKY_FIT_STREAM=0"; do
  env $v timeout 300 python tools/bench_step.py 30 2>/dev/null | tail -1
done > gpurun_out/r6/call6_ab.log 2>&1
cat gpurun_out/r6/call6_ab.log
timeout 1500 bash tools/flake_seq.sh > gpurun_out/r6/call6_guard.log 2>&1
tail -15 gpurun_out/r6/call6_guard.log
bash tools/trace_gaps.sh r06b > /dev/null 2>&1; head -5 gpurun_out/trace_r06b.txt
timeout 900 python -m pytest tests/test_gpu_trainer_surface.py tests/test_gpu_graph.py tests/test_gpu_two_ranks.py tests/test_gpu_step.py -m gpu -x -q > gpurun_out/r6/call6_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call6_pytest.log)"
timeout 900 python bench.py > gpurun_out/r6/call6_bench.json 2> gpurun_out/r6/call6_bench.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/r6/call6_bench.json
