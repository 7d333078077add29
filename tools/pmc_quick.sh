#!/bin/bash
# Quick SQ counter passes for one Chamfer size (run on the GPU box):
#   tools/pmc_quick.sh B N [env assignments exported beforehand]
set -u
B=${1:-13}; N=${2:-16384}
OUT=gpurun_out/pmcq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K="timeout -s KILL 90"
$K rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -o c -- python3 tools/prof_chamfer.py $B $N 3 > $OUT/a.log 2>&1
$K rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/b -o c -- python3 tools/prof_chamfer.py $B $N 3 > $OUT/b.log 2>&1
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob("gpurun_out/pmcq/*/*counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        per[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, d in per.items():
        if "nn_" not in k: continue
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s %.4g" % (c, sum(v) / len(v)))
PY
