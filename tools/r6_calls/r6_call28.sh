timeout 600 python -m pytest tests/test_retire_and_streams.py -m gpu -x -q 2>&1 | tail -3
