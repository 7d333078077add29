python -m pytest tests/test_gpu_hpr.py -x -q 2>&1 | tail -5
for ht in 3 2 1; do echo "== home tiles $ht"; GENPC_HPR_HOME_TILES=$ht python tools/time_hpr_1024.py 2>&1 | grep -v amdgpu; done
echo "== time_hpr"; python tools/time_hpr.py 2>&1 | grep -v amdgpu
GENPC_HPR_HOME_TILES=1 python -m pytest tests/test_gpu_hpr.py -x -q 2>&1 | tail -3
