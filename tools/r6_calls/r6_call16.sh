mkdir -p gpurun_out/r6
for c in distinct self_wait alias_side; do timeout 120 python tools/stream_alias_probe.py $c > gpurun_out/r6/call16_alias_$c.log 2>&1; echo "alias probe $c rc=$? $(tail -1 gpurun_out/r6/call16_alias_$c.log | cut -c1-100)"; done
T="tests/test_checkpoints.py tests/test_gpu_bench_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_eval_latents.py tests/test_gpu_eval_methods.py tests/test_gpu_field_chain.py tests/test_gpu_field_paths.py tests/test_gpu_film_chain.py tests/test_gpu_full_size.py tests/test_gpu_full_size_trajectory.py tests/test_gpu_gemm.py tests/test_gpu_graph.py"
NSKY_PRINT_STREAMS=1 timeout 900 python tools/pytest_lab.py $T -m gpu -x -q -s > gpurun_out/r6/call16_streams.log 2>&1; echo "streams run rc=$?"
grep -a "NSKY streams" gpurun_out/r6/call16_streams.log | tail -6 | cut -c1-250
