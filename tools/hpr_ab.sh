python -m pytest tests/test_gpu_hpr.py tests/test_gpu_scans.py -x -q 2>&1 | tail -2
python tools/time_hpr_1024.py 2>&1 | grep -v amdgpu
