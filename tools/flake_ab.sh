# same-box alternation of the flaky subset with autograd's worker threads on (1) and off (0): tools/flake_ab.sh [pairs]
N=${1:-12}
for i in $(seq 1 $N); do for v in 1 0; do
  NSKY_BACKWARD_THREADS=$v python -m pytest tests/test_checkpoints.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_eval_latents.py tests/test_gpu_eval_methods.py -m gpu -q -x > /tmp/flake_$i.log 2>&1
  echo "threads=$v run $i rc=$? $(tail -1 /tmp/flake_$i.log | cut -c1-50)"
done; done
