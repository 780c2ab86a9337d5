mkdir -p gpurun_out/r6
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl
P="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}"
HEAP_GUARD_FENCE_SIZE=920 LD_PRELOAD="$P" timeout 1500 python -m pytest tests -m gpu -x -q -s -p no:faulthandler -p no:cacheprovider > gpurun_out/r6/call13_pytest_guard.log 2>&1
echo "guarded pytest rc=$?"; grep -a "signal 11\|heap_guard: sweep\|passed\|failed" gpurun_out/r6/call13_pytest_guard.log | tail -5 | cut -c1-250
grep -a -n "signal 11" -A14 gpurun_out/r6/call13_pytest_guard.log | cut -c1-200 | head -30
