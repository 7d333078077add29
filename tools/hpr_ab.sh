STRESS_SOAK=1 timeout 900 python tools/stress_concurrent.py 3 8 2>&1 | grep -v amdgpu | cut -c1-260
