mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_trainer_surface.py tests/test_gpu_graph.py tests/test_gpu_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_trajectory.py tests/test_gpu_step.py tests/test_gpu_full_size.py tests/test_gpu_film_chain.py -m gpu -x -q > gpurun_out/r6/call3_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call3_pytest.log)"
for v in "" "NSKY_FIT_STREAM=0" "NSKY_FILM_ASYNC=0" "" "NSKY_FIT_STREAM=0" "NSKY_FILM_ASYNC=0"; do
  env $v timeout 300 python tools/bench_step.py 30 2>/dev/null | tail -1
done > gpurun_out/r6/call3_ab.log 2>&1
cat gpurun_out/r6/call3_ab.log
timeout 1500 bash tools/flake_seq.sh > gpurun_out/r6/call3_guard.log 2>&1
tail -15 gpurun_out/r6/call3_guard.log
