#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box):
#   tools/collect_profiles.sh <tag>        e.g. r01
# kernel-trace/stats of the bench command, then FETCH_SIZE / WRITE_SIZE / SQ counter
# passes (separate runs: TCC has 4 slots, FETCH_SIZE takes 3) of the headline launch.
# Every profiled program is `python3 <script>` directly after `--` (no wrappers).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K="timeout -s KILL 90"
$K rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline > $OUT/bench_trace.log 2>&1
$K rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o c -- python3 tools/prof_chamfer.py 1 16384 5 > $OUT/fetch.log 2>&1
$K rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o c -- python3 tools/prof_chamfer.py 1 16384 5 > $OUT/write.log 2>&1
$K rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq -o c -- python3 tools/prof_chamfer.py 1 16384 5 > $OUT/sq.log 2>&1
$K rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch13 -o c -- python3 tools/prof_chamfer.py 13 16384 3 > $OUT/fetch13.log 2>&1
$K rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/sq13 -o c -- python3 tools/prof_chamfer.py 13 16384 3 > $OUT/sq13.log 2>&1
ls -R $OUT | head -40
