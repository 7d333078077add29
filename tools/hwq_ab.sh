#!/bin/bash
# Scans in flight against the number of hardware queues the HIP runtime multiplexes streams onto (GPU_MAX_HW_QUEUES, default 4).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for q in 4 8 16 24; do
  echo "GPU_MAX_HW_QUEUES=$q: $(GPU_MAX_HW_QUEUES=$q python3 tools/time_c2_lanes.py 1 6 8 2>&1 | grep lanes | tr '\n' ';')"
done
