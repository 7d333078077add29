"""Alignment loop throughput (full objective): single scan and 8 scans in lock-step.   python3 tools/time_reg.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
for b in (1, 8):
    C = torch.rand(b, 16384, 3, device="cuda", generator=gen) - 0.5
    P = (C[:, :8192] * 0.9).contiguous()
    object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=20, render_size=224); torch.cuda.synchronize()
    t0 = time.perf_counter(); object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=200, render_size=224); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("b %d: %.1f ms per call, %.1f us per Adam step, %.1f scans/s" % (b, dt * 1e3, dt / 804 * 1e6, b / dt))
