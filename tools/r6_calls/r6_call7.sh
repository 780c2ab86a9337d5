mkdir -p gpurun_out/r6
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
P="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}"
for m in fork line fork; do
  LD_PRELOAD="$P" timeout 300 python tools/hip_graph_destroy_uaf.py $m 60 > gpurun_out/r6/call7_uaf_$m.log 2>&1; echo "uaf $m rc=$?"; grep -a "heap_guard:\|HEAP DAMAGE\|clean\|torch " gpurun_out/r6/call7_uaf_$m.log | head -5
done
timeout 600 python -m pytest tests/test_gpu_two_ranks.py -m gpu -x -q > gpurun_out/r6/call7_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call7_pytest.log)"
timeout 600 python tools/op_sites.py > gpurun_out/r6/call7_op_sites.log 2>&1; tail -50 gpurun_out/r6/call7_op_sites.log
