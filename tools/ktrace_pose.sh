#!/bin/bash
# Per-kernel durations of a short alignment loop (run on the GPU box): tools/ktrace_pose.sh
set -u
OUT=gpurun_out/ktp
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 tools/prof_pose.py ${POSE_ARGS:-} > $OUT/t.log 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/ktp/t/*kernel_stats.csv"):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:14]:
        print("%-64s calls %6s avg %9.2f us total %8.2f ms" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
