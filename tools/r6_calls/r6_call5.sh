mkdir -p gpurun_out/r6
timeout 1200 bash tools/flake_seq.sh > gpurun_out/r6/call5_guard.log 2>&1
tail -15 gpurun_out/r6/call5_guard.log
bash tools/trace_gaps.sh r06a > /dev/null 2>&1; head -5 gpurun_out/trace_r06a.txt
timeout 1500 python -m pytest tests/test_gpu_full_size_trajectory.py tests/test_gpu_engine_grads.py tests/test_gpu_trajectory.py tests/test_gpu_step.py tests/test_gpu_full_size.py tests/test_gpu_film_chain.py -m gpu -x -q > gpurun_out/r6/call5_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call5_pytest.log)"
