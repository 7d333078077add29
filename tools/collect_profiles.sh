#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box):
#   tools/collect_profiles.sh <tag>        e.g. r02
# For the bench command and for every kernel family: a kernel-trace/stats pass, then FETCH_SIZE /
# WRITE_SIZE / SQ counter passes (separate runs: TCC has 4 slots, FETCH_SIZE takes 3; never combined
# with a trace).  Every profiled program is `python3 <script>` directly after `--` (no wrappers).
set -u
TAG=${1:-r06}
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K="timeout -s KILL 240"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"
$K rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python3 bench.py --steps 50 --warmup 5 --no-extra --no-cpu-baseline > $OUT/bench_trace.log 2>&1
family() {   # family <name> <script> [args...]
    local name=$1; shift
    $K rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -o t -- python3 "$@" > $OUT/$name.trace.log 2>&1
    $K rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name/fetch -o c -- python3 "$@" > $OUT/$name.fetch.log 2>&1
    $K rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/$name/write -o c -- python3 "$@" > $OUT/$name.write.log 2>&1
    $K rocprofv3 --pmc $SQ --output-format csv -d $OUT/$name/sq -o c -- python3 "$@" > $OUT/$name.sq.log 2>&1
}
family chamfer_B1_16384 tools/prof_chamfer.py 1 16384 5
family chamfer_B13_16384 tools/prof_chamfer.py 13 16384 3
family chamfer_B13_16384_scans tools/prof_chamfer.py 13 16384 3 scan
family emd_B1_16384 tools/prof_emd.py 1 16384 2
family emd_B13_16384 tools/prof_emd.py 13 16384 1
family emd_B13_16384_scans tools/prof_emd.py 13 16384 1 scan
family emd_B1_16384_scan tools/prof_emd.py 1 16384 2 scan
family get_uvs_1024x71372 tools/prof_uvs.py 3
family streaming_64x32768 tools/prof_streaming.py
family pose_loop_16384x8192 tools/prof_pose.py 16384 8192 20 1
family scale_search_icp tools/prof_scale_search.py
family fps_voxel tools/prof_fps_voxel.py
family fps_scan_24576 tools/prof_fps_scan.py
family hpr_64x10000 tools/prof_hpr.py small
family hpr_2x165546 tools/prof_hpr.py big
family hpr_1024x10000 tools/prof_hpr1024.py
family reg_8192_vs_16384 tools/prof_reg.py
family c2_chain_8192 tools/prof_c2.py
family c2_chain_8192_scan tools/prof_c2_scan.py
family reg8_lockstep tools/time_reg8.py
ls $OUT | head -60
