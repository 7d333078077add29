python -m pytest tests/test_gpu_hpr.py -x -q 2>&1 | tail -5
echo "== LP on"; python tools/time_hpr_1024.py 2>&1 | grep -v amdgpu
echo "== LP off"; GENPC_HPR_NOCULL=128 python tools/time_hpr_1024.py 2>&1 | grep -v amdgpu
echo "== LP on"; python tools/time_hpr.py 2>&1 | grep -v amdgpu
echo "== LP off"; GENPC_HPR_NOCULL=128 python tools/time_hpr.py 2>&1 | grep -v amdgpu
