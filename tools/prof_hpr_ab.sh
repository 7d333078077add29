#!/bin/bash
# Per-kernel times of the hidden-point removal at 1024 x 10000 under a few launch shapes.   bash tools/prof_hpr_ab.sh [VAR=val ...]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
run() {
    tag=$1; shift
    rm -rf /tmp/hp_$tag
    ( export "$@"; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hp_$tag -o t -- python3 $R/tools/prof_hpr1024.py > /dev/null 2>&1 )
    echo "== $tag ($*)"
    python3 - "$tag" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/hp_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
for r in csv.DictReader(open(f[0])):
    if "hpr" in r["Name"] and float(r["AverageNs"]) > 20000:
        print("  %-44s calls %3s avg %10.1f us" % (r["Name"][:44].replace("genpc::", ""), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
}
run default GENPC_X=0


