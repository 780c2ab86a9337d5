# repeat the subset of the GPU suite in which the round-5 intermittent abort showed (host heap corruption after record_stream'd
# side-stream operands met the captured graphs' pools): tools/flake.sh [runs]
N=${1:-10}
for i in $(seq 1 $N); do
  python -m pytest tests/test_checkpoints.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_eval_latents.py tests/test_gpu_eval_methods.py -m gpu -q -x > /tmp/flake_$i.log 2>&1
  echo "run $i rc=$? $(tail -1 /tmp/flake_$i.log | cut -c1-60)"
done
