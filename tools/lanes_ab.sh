#!/bin/bash
# Scans in flight with and without the shared farthest-point-sampling launches.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in 1 0 1 0; do
  echo "GENPC_FPS_COMBINER=$v: $(GENPC_FPS_COMBINER=$v python3 tools/time_c2_lanes.py 1 4 6 8 2>&1 | grep lanes | tr '\n' ';')"
done
