# repeat the subset of the GPU suite in which the round-5 intermittent abort showed: tools/flake.sh [runs]; failing logs -> gpurun_out/flake_fail_<i>.log
N=${1:-10}
for i in $(seq 1 $N); do
  python -m pytest tests/test_checkpoints.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_eval_latents.py tests/test_gpu_eval_methods.py -m gpu -q -x > /tmp/flake_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(tail -1 /tmp/flake_$i.log | cut -c1-60)"
  if [ $rc -ne 0 ]; then grep -v "^Extension" /tmp/flake_$i.log | tail -120 | cut -c1-400 > gpurun_out/flake_fail_$i.log; fi
done
