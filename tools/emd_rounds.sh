#!/bin/bash
# Per-round kernel durations of one EMD forward (run on the GPU box): tools/emd_rounds.sh B N
set -u
B=${1:-1}; N=${2:-16384}
OUT=gpurun_out/emdr
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python3 tools/prof_emd.py $B $N 2 > $OUT/log.txt 2>&1
python3 - <<'PY'
import csv, glob
rows = [r for r in csv.DictReader(open(glob.glob("gpurun_out/emdr/*kernel_trace.csv")[0])) if "emd_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-152:]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
bids = [dur(r) for r in last if "bid" in r["Kernel_Name"]]
gm = [dur(r) for r in last if "getmax" in r["Kernel_Name"]]
asg = [dur(r) for r in last if "assign" in r["Kernel_Name"]]
print("span %.1f us; bid %.1f getmax %.1f assign %.1f" % ((int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3, sum(bids), sum(gm), sum(asg)))
print("bid per round:", [round(x, 1) for x in bids])
PY
