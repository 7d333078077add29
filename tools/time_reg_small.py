"""Alignment loop on a SMALL object (the cloud covers ~1/10 of the image: a few crowded tiles).   python3 tools/time_reg_small.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
for b in (1, 8):
    C = (torch.rand(b, 16384, 3, device="cuda", generator=gen) - 0.5) * 0.3
    P = (C[:, :8192] * 0.9).contiguous()
    object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=20, render_size=224); torch.cuda.synchronize()
    t0 = time.perf_counter(); object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=200, render_size=224); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("b %d: %.1f ms per call" % (b, dt * 1e3))
