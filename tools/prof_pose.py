"""rocprofv3 driver: one alignment loop (4 starts x 201 steps).  python3 tools/prof_pose.py NC NP"""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization

nc, npart = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1)
C = torch.from_numpy(rng.random((nc, 3), dtype=np.float32) - np.float32(0.5)).cuda()
P = (C[:npart] * 0.9).contiguous()
T = object_pose_optimization(C, P, lr=0.01, iters=200)
torch.cuda.synchronize()
print("done", T[0, 0])
