mkdir -p gpurun_out/r6
T="tests/test_checkpoints.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_eval_latents.py tests/test_gpu_eval_methods.py tests/test_gpu_field_chain.py tests/test_gpu_field_paths.py tests/test_gpu_film_chain.py tests/test_gpu_full_size.py tests/test_gpu_full_size_trajectory.py tests/test_gpu_gemm.py tests/test_gpu_graph.py"
timeout 900 python tools/pytest_lab.py $T -m gpu -x -q > gpurun_out/r6/call15_A.log 2>&1; echo "A (as is) rc=$? $(tail -1 gpurun_out/r6/call15_A.log | cut -c1-80)"
NSKY_RETIRE_SECONDS=1e9 timeout 900 python tools/pytest_lab.py $T -m gpu -x -q > gpurun_out/r6/call15_B.log 2>&1; echo "B (never destroy) rc=$? $(tail -1 gpurun_out/r6/call15_B.log | cut -c1-80)"
NSKY_ORDER_BY_STREAM=1 timeout 900 python tools/pytest_lab.py $T -m gpu -x -q > gpurun_out/r6/call15_C.log 2>&1; echo "C (stream waits) rc=$? $(tail -1 gpurun_out/r6/call15_C.log | cut -c1-80)"
grep -a "Fatal\|File \"/tmp/code\|File \"/root" gpurun_out/r6/call15_A.log | head -12 | cut -c1-200
