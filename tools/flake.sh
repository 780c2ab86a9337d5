# the flaky subset under glibc's heap checking: where is the corruption detected?
for i in 1 2 3 4 5 6; do
  MALLOC_CHECK_=3 PYTHONFAULTHANDLER=1 python -m pytest tests/test_checkpoints.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_eval_latents.py tests/test_gpu_eval_methods.py -m gpu -q -x > /tmp/flake_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -c PASSED /tmp/flake_$i.log)"
  if [ $rc -ne 0 ]; then
    grep -v "^Extension" /tmp/flake_$i.log | grep -B 5 -A 45 "Fatal Python\|free()\|malloc\|corrupt" | grep -v "site-packages/_pytest\|pluggy" | head -90 > gpurun_out/flake_ctx_$i.log
  fi
done
