#!/bin/bash
# SQ counters of one kernel of any driver (run on the GPU box): tools/pmc_any.sh <tag> <kernel substring> <script> [args...]
set -u
TAG=$1; KER=$2; shift; shift
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/p$i -o p -- python3 "$@" > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$KER" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-24s last call %16.0f  (calls %d)" % (k, v[-1], len(v)))
PY
