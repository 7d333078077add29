#!/bin/bash
# one-launch EMD: lanes per point (GENPC_EMD_AUCTION_K) x barrier form (GENPC_EMD_AUCTION_FLAT: workgroups per cloud up to
# which the one-word barrier is used) x shapes, a process per setting (the switches are read once)
for FLAT in 0 128 1024; do
for K in 1 2 4 8; do
  echo "=== FLAT=$FLAT K=$K"
  GENPC_EMD_AUCTION_FLAT=$FLAT GENPC_EMD_AUCTION_K=$K timeout 200 python3 tools/time_emd_auction.py quick 2>&1 | grep -v "amdgpu.ids\|scan"
done
done
