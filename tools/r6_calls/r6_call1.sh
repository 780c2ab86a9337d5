# round 6, GPU call 1: the refactored boundary (slab + exchange + replay behind the pipeline) and the heap-guard run of the aborting sequence
mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_trainer_surface.py tests/test_gpu_graph.py tests/test_gpu_two_ranks.py tests/test_gpu_engine_grads.py tests/test_gpu_trajectory.py tests/test_gpu_losses.py tests/test_gpu_step.py -m gpu -x -q > gpurun_out/r6/call1_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call1_pytest.log)"
timeout 1500 bash tools/flake_seq.sh > gpurun_out/r6/call1_guard.log 2>&1
echo "guard done"; cat gpurun_out/r6/call1_guard.log | tail -20
