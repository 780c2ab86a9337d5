# which preceding tests does the abort need?  baseline first (all four files); the variants only if the box shows the flake at all
ALL=test_checkpoints.py,test_gpu_bench_two_ranks.py,test_gpu_engine_grads.py,test_gpu_eval_latents.py
run() { python tools/flake_seq.py "$1" > /tmp/fs.log 2>&1; rc=$?; echo "[$2] rc=$rc $(grep -c 'eval methods' /tmp/fs.log)"; return $rc; }
fails=0
for i in 1 2 3 4 5 6 7 8; do run $ALL all || fails=$((fails+1)); done
echo "baseline failures: $fails of 8"
if [ $fails -eq 0 ]; then echo "box not flaky: stop"; exit 0; fi
for i in 1 2 3 4 5 6; do
  run test_gpu_eval_latents.py latents
  run test_checkpoints.py checkpoints
  run test_gpu_bench_two_ranks.py bench
  run test_gpu_engine_grads.py engine
  run "" none
done
