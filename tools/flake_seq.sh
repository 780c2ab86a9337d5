# does the abort move into an explicit collection of what tests/test_gpu_eval_latents.py left behind?
mkdir -p gpurun_out/flake
run() { NSKY_FLAKE_COLLECT=$1 python tools/flake_seq.py test_gpu_eval_latents.py > /tmp/fs.log 2>&1; rc=$?; echo "[collect=$1] rc=$rc $(grep -c 'eval methods' /tmp/fs.log) last: $(grep -a 'collect:' /tmp/fs.log | tail -1)"; [ $rc -ne 0 ] && tail -c 6000 /tmp/fs.log > gpurun_out/flake/fail_$2_$1.log; }
for i in 1 2 3 4 5 6 7 8 9 10; do
  run 1 $i
  run "" $i
done
