"""bench.py's 'three groups of eight scans in flight' number, five times in one process.   python3 tools/time_groups_in_flight.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
from genpc_amd.metric import evaluate_scans
from genpc_amd import pipeline as pl
dev = torch.device("cuda")
gen = torch.Generator(device="cuda"); gen.manual_seed(0)
n = 16384
C8 = (torch.rand(8, n, 3, device=dev, generator=gen) - 0.5)
P8b = (C8[:, :8192] * 0.9).contiguous()
X8, Y8 = C8 + 0.5, (C8.flip(0) + 0.5).contiguous()


def group8(li, _):
    object_pose_optimization(C8, P8b, radius=0.02, lr=0.01, iters=200, render_size=224)
    return evaluate_scans(X8, Y8)


group8(0, 0); torch.cuda.synchronize()
t0 = time.perf_counter(); group8(0, 0); torch.cuda.synchronize()
print("one group: %.1f scans/s" % (8 / (time.perf_counter() - t0)))
for lanes in (2, 3, 3, 3, 4):
    pl.run_in_lanes(group8, range(lanes), lanes, dev); torch.cuda.synchronize()
    t0 = time.perf_counter(); pl.run_in_lanes(group8, range(2 * lanes), lanes, dev); torch.cuda.synchronize()
    print("%d groups in flight: %.1f scans/s" % (lanes, 16 * lanes / (time.perf_counter() - t0)), flush=True)
