mkdir -p gpurun_out/r6
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
P="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}"
for v in "fork inflight" "fork idle" "line inflight" "fork inflight"; do
  set -- $v
  HEAP_GUARD_FENCE_SIZE=920 LD_PRELOAD="$P" timeout 300 python tools/hip_graph_destroy_uaf.py $1 $2 60 > gpurun_out/r6/call9_uaf_$1_$2.log 2>&1; echo "uaf $v rc=$?"
  grep -a "signal 11\|HEAP DAMAGE\|clean\|torch " gpurun_out/r6/call9_uaf_$1_$2.log | head -4; grep -a -c "allocated by libamdhip64.so+0x3c2872) FREED" gpurun_out/r6/call9_uaf_$1_$2.log
done
