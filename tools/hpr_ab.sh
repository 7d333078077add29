python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_concurrency.py -x -q 2>&1 | tail -2
timeout 400 python tools/time_c2_lanes.py 1 2 4 6 8 6 2>&1 | grep -v amdgpu
