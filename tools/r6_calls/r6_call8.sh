mkdir -p gpurun_out/r6
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
P="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}"
for i in 1 2; do
  HEAP_GUARD_FENCE_SIZE=920 LD_PRELOAD="$P" timeout 900 python tools/flake_seq.py test_gpu_eval_latents.py > gpurun_out/r6/call8_fence_$i.log 2>&1; echo "fence run $i rc=$?"
  grep -a -c "FREED by" gpurun_out/r6/call8_fence_$i.log; grep -a "signal 11\|signal 7" gpurun_out/r6/call8_fence_$i.log | head -3
done
for v in "" "NSKY_FIT_STREAM=0" "" "NSKY_FIT_STREAM=0"; do
  env $v timeout 300 python tools/bench_step.py 30 2>/dev/null | tail -1
done > gpurun_out/r6/call8_ab.log 2>&1
cat gpurun_out/r6/call8_ab.log
