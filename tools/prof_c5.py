"""rocprofv3 driver: a short alignment loop on BASELINE config 5's per-rank shape (8 synthetic scans x 32768
points, bench.synth_scan: partial vs complete shapes).   python3 tools/prof_c5.py [ITERS]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc = [bench.synth_scan(k, 32768) for k in range(8)]
C5 = torch.from_numpy(np.stack([x[0] for x in sc])).cuda()
P5 = torch.from_numpy(np.stack([x[1] for x in sc])).cuda()
T = object_pose_optimization(C5, P5, radius=0.02, lr=0.01, iters=iters, render_size=224)
torch.cuda.synchronize()
print("done", T[0, 0, 0])
