#!/bin/bash
# kernel trace of one completed scan (tools/prof_c2_scan.py) under a few switches -> gpurun_out/kt_c2_<name>.csv (kernel stats)
# usage: tools/kt_c2.sh name [ENV=VAL ...]     (run on the GPU box)
set -u
NAME=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kt_c2_$NAME
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 tools/prof_c2_scan.py > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$NAME" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== %s: total kernel time %.2f ms (two scans)" % (sys.argv[2], tot / 1e6))
for r in rows[:26]:
    print("%-70s calls %5s avg_us %9.2f total_ms %8.3f" % (r["Name"].replace("void ", "").replace("genpc::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
cp "$f" gpurun_out/kt_c2_$NAME.csv
rm -rf $OUT
