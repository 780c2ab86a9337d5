timeout 900 python -m pytest tests/test_gpu_trainer_surface.py tests/test_gpu_two_ranks.py -m gpu -x -q 2>&1 | tail -3
