#!/bin/bash
# Per-kernel durations of any driver (run on the GPU box): tools/ktrace_any.sh <tag> <script> [args...]
set -u
TAG=$1; shift
OUT=gpurun_out/kt_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 180 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python3 "$@" > $OUT/t.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/t/*kernel_stats.csv"):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:12]:
        print("%-70s calls %6s avg %9.2f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
