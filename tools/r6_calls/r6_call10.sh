mkdir -p gpurun_out/r6
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
P="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}"
for v in "fork inflight" "fork idle" "line inflight" "fork inflight"; do
  set -- $v
  HEAP_GUARD_FENCE_SIZE=920 LD_PRELOAD="$P" timeout 300 python tools/hip_graph_destroy_uaf.py $1 $2 60 > gpurun_out/r6/call10_uaf_$1_$2.log 2>&1; echo "uaf $v rc=$?"
  grep -a "signal 11\|HEAP DAMAGE\|clean\|torch " gpurun_out/r6/call10_uaf_$1_$2.log | head -4; grep -a -c "allocated by libamdhip64.so+0x3c2872) FREED" gpurun_out/r6/call10_uaf_$1_$2.log
done
timeout 900 python -m pytest tests/test_gpu_soak.py tests/test_gpu_eval_methods.py tests/test_gpu_eval_latents.py -m gpu -x -q > gpurun_out/r6/call10_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call10_pytest.log)"
for i in 1 2; do
  HEAP_GUARD_FENCE_SIZE=920 LD_PRELOAD="$P" timeout 900 python tools/flake_seq.py test_gpu_eval_latents.py > gpurun_out/r6/call10_fence_$i.log 2>&1; echo "fence run $i rc=$?"
  grep -a "signal 11\|signal 7" gpurun_out/r6/call10_fence_$i.log | head -3; grep -a -c "allocated by libamdhip64.so+0x3c2872) FREED" gpurun_out/r6/call10_fence_$i.log
done
timeout 1500 bash tools/flake_seq.sh > gpurun_out/r6/call10_guard.log 2>&1
grep "guard" gpurun_out/r6/call10_guard.log | cut -c1-200
