mkdir -p gpurun_out/r6
T="tests/test_gpu_full_size_trajectory.py tests/test_gpu_graph.py"
timeout 900 python tools/pytest_lab.py $T -m gpu -x -q > gpurun_out/r6/call14_A.log 2>&1; echo "A (as is) rc=$? $(tail -1 gpurun_out/r6/call14_A.log | cut -c1-80)"
NSKY_RETIRE_SECONDS=1e9 timeout 900 python tools/pytest_lab.py $T -m gpu -x -q > gpurun_out/r6/call14_B.log 2>&1; echo "B (never destroy) rc=$? $(tail -1 gpurun_out/r6/call14_B.log | cut -c1-80)"
NSKY_ORDER_BY_STREAM=1 timeout 900 python tools/pytest_lab.py $T -m gpu -x -q > gpurun_out/r6/call14_C.log 2>&1; echo "C (stream waits) rc=$? $(tail -1 gpurun_out/r6/call14_C.log | cut -c1-80)"
NSKY_FIT_STREAM=0 timeout 900 python tools/pytest_lab.py $T -m gpu -x -q > gpurun_out/r6/call14_D.log 2>&1; echo "D (no fit stream) rc=$? $(tail -1 gpurun_out/r6/call14_D.log | cut -c1-80)"
grep -a "Fatal\|File \"/tmp/code\|File \"/root" gpurun_out/r6/call14_A.log | head -12 | cut -c1-200
