"""rocprofv3 driver: one alignment loop (starts x (iters + 1) steps, full objective).
   python3 tools/prof_pose.py [NC NP [ITERS [B]]]"""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization

nc = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
npart = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
b = int(sys.argv[4]) if len(sys.argv) > 4 else 1
rng = np.random.default_rng(1)
C = torch.from_numpy(rng.random((b, nc, 3), dtype=np.float32) - np.float32(0.5)).cuda()
P = (C[:, :npart] * 0.9).contiguous()
T = object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=iters, render_size=224)
torch.cuda.synchronize()
print("done", T[0, 0, 0])
