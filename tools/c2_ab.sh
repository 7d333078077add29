#!/bin/bash
# One completed scan (bench.py's C2 input) under the round-5 switches, a process each:
#   what the FPS verification, Pulsar's blend and the one-launch auction cost or buy in the chain
run() { echo -n "$1: "; env $2 python3 - <<'PY' 2>&1 | grep "one scan"
import os, sys, time, runpy
sys.argv = ["prof_c2_scan.py"]
os.environ["GENPC_POSE_SEEDED"] = os.environ.get("GENPC_POSE_SEEDED", "2")
import io, contextlib
src = open("tools/prof_c2_scan.py").read().replace('os.environ.setdefault("GENPC_POSE_SEEDED", "0")', "pass")
src += '''
import torch, time
ts = []
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print("one scan: best %.2f ms, median %.2f ms of 5" % (min(ts), sorted(ts)[2]))
'''
exec(compile(src, "c2_ab", "exec"), {"__name__": "__main__", "__file__": os.path.abspath("tools/prof_c2_scan.py")})
PY
}
run "default                 " "X=1"
run "GENPC_FPS_VERIFY=0      " "GENPC_FPS_VERIFY=0"
run "GENPC_RENDER_BLEND=0    " "GENPC_RENDER_BLEND=0"
run "GENPC_EMD_AUCTION=0     " "GENPC_EMD_AUCTION=0"
run "all three off           " "GENPC_FPS_VERIFY=0 GENPC_RENDER_BLEND=0 GENPC_EMD_AUCTION=0"
