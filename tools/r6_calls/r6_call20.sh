mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_slab_adam.py tests/test_gpu_trainer_surface.py -m gpu -x -q > gpurun_out/r6/call20_pytest.log 2>&1
echo "pytest rc=$? $(tail -1 gpurun_out/r6/call20_pytest.log)"; grep -a "Error\|assert" gpurun_out/r6/call20_pytest.log | head -10 | cut -c1-250
