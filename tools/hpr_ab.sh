timeout 600 python tools/soak_lanes.py 6 180 2>&1 | grep -v amdgpu | tail -3
timeout 600 python tools/soak_lanes.py 8 160 2>&1 | grep -v amdgpu | tail -3
