mkdir -p gpurun_out/r6
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
P="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}"
for v in "fork inflight 40 3 8" "fork inflight 40 4 30" "fork idle 40 3 8" "fork inflight 40 2 4"; do
  set -- $v
  HEAP_GUARD_FENCE_SIZE=920 LD_PRELOAD="$P" timeout 300 python tools/hip_graph_destroy_uaf.py $1 $2 $3 $4 $5 > gpurun_out/r6/call11_uaf_$1_$2_$4_$5.log 2>&1; echo "uaf $v rc=$?"
  grep -a "signal 11\|HEAP DAMAGE\|clean\|torch " gpurun_out/r6/call11_uaf_$1_$2_$4_$5.log | head -4; grep -a -c "allocated by libamdhip64.so+0x3c2872) FREED" gpurun_out/r6/call11_uaf_$1_$2_$4_$5.log
  grep -a "FREED by libamdhip64" -A14 gpurun_out/r6/call11_uaf_$1_$2_$4_$5.log | grep -a -c "0x3d59b5"
done
