python -m pytest tests/test_gpu_hpr_paths.py -x -q 2>&1 | tail -15
