mkdir -p gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_slab_adam.py tests/test_gpu_trainer_surface.py tests/test_gpu_trajectory.py tests/test_gpu_full_size_trajectory.py tests/test_gpu_eval_latents.py tests/test_gpu_graph.py tests/test_checkpoints.py -m gpu -x -q > gpurun_out/r6/call21_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call21_pytest.log)"; grep -a "Error\|assert " gpurun_out/r6/call21_pytest.log | head -10 | cut -c1-250
