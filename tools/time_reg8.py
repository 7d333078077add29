"""Alignment loop, 8 scans in lock-step only (A/B of launch shapes).   python3 tools/time_reg8.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
b = 8
C = torch.rand(b, 16384, 3, device="cuda", generator=gen) - 0.5
P = (C[:, :8192] * 0.9).contiguous()
object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=20, render_size=224); torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=200, render_size=224); torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print("b 8: %.1f ms per call (min of 3)" % (min(ts) * 1e3))
