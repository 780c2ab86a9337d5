# run the flaky subset until it aborts, with ROCclr logging; keep the tail of the failing run
for i in 1 2 3 4 5 6; do
  AMD_LOG_LEVEL=3 python -m pytest tests/test_checkpoints.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_eval_latents.py tests/test_gpu_eval_methods.py -m gpu -q -x > /tmp/flake_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc"
  if [ $rc -ne 0 ]; then
    grep -n "Fatal Python" /tmp/flake_$i.log | head -2
    grep -v "^  File\|^Extension" /tmp/flake_$i.log | grep -B 400 "Fatal Python" | grep -i "ShaderName\|error\|fault\|abort\|fail" | tail -60 > gpurun_out/flake_tail.log
    grep -B 60 "Fatal Python" /tmp/flake_$i.log | head -80 > gpurun_out/flake_ctx.log
    break
  fi
done
