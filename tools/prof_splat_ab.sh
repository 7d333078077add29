cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 1 0; do
  export GENPC_RENDER_BLEND=$B
  rm -rf gpurun_out/ps_$B
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ps_$B -o t -- python3 tools/prof_pose.py 16384 8192 20 1 > gpurun_out/ps_$B.log 2>&1
  echo "== blend $B"
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/ps_$B/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mask" in r["Name"] or "pose" in r["Name"]:
            print("%-60s calls %4s avg %8.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
  rm -rf gpurun_out/ps_$B
done
