python -m pytest tests/test_gpu_hpr.py -x -q 2>&1 | tail -3
python tools/time_hpr_1024.py 2>&1 | grep -v amdgpu
python tools/time_hpr.py 2>&1 | grep -v amdgpu | head -4
GENPC_LIB=$PWD/tools/_hprprof/libgenpc_hip.so python tools/hpr_phases.py 2>&1 | grep walk
