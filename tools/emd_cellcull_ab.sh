#!/bin/bash
# 13 bundled scans, EMD through the launch-per-round path: cells culled by their smallest price or not.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in 0 1 2 4; do
  echo "GENPC_EMD_CELLCULL=$v: $(GENPC_EMD_CELLCULL=$v python3 tools/emd_scan_sweep.py child 2>&1 | tail -1)"
done
