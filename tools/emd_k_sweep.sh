#!/bin/bash
# one-launch EMD: lanes per point (GENPC_EMD_AUCTION_K) x shapes, a process per setting (the switch is read once)
for K in 1 2 4 8 16; do
  echo "=== K=$K"
  GENPC_EMD_AUCTION_K=$K timeout 200 python3 tools/time_emd_auction.py quick 2>&1 | grep -v amdgpu.ids
done
