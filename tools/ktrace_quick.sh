#!/bin/bash
# Per-kernel durations of a few Chamfer forward calls (run on the GPU box):
#   tools/ktrace_quick.sh B N
set -u
B=${1:-1}; N=${2:-16384}; KIND=${3:-uniform}
OUT=gpurun_out/ktq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 90 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 tools/prof_chamfer.py $B $N 20 $KIND > $OUT/t.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/ktq/t/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "nn_" in r["Name"] or "grid_" in r["Name"]:
            print("%-70s calls %s avg %.2f us min %.2f max %.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
