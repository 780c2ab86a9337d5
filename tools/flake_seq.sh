# NEXT STEP of the flake hunt (not run yet -- the round's GPU minutes were spent): the sequence that aborts one time in four,
# under tools/heap_guard.c, with a sweep of every host block after each test and around every graph replay / Adam launch of the fits.
# (LD_PRELOAD is PREPENDED: the GPU box preloads its own exec guard, and a python that drops it exits 77 without a word at the first GPU call.)
# Exit code 77 of a run = the guard saw the damage; its log names the block's allocator and the step it was first seen after.
mkdir -p gpurun_out/flake
gcc -O1 -g -shared -fPIC -o /tmp/heap_guard.so tools/heap_guard.c -ldl || exit 1
for i in $(seq 1 ${FLAKE_RUNS:-12}); do
  LD_PRELOAD="/tmp/heap_guard.so${LD_PRELOAD:+:$LD_PRELOAD}" NSKY_FLAKE_SWEEP_STEPS=1 python tools/flake_seq.py test_gpu_eval_latents.py > /tmp/fs.log 2>&1; rc=$?
  echo "[guard] rc=$rc $(grep -c 'eval methods' /tmp/fs.log) $(grep -a -m1 'heap_guard:' /tmp/fs.log | cut -c1-200)"
  [ $rc -ne 0 ] && { grep -a -B2 -A40 -m1 'heap_guard:' /tmp/fs.log > gpurun_out/flake/guard_$i.log; tail -c 4000 /tmp/fs.log >> gpurun_out/flake/guard_$i.log; }
done
