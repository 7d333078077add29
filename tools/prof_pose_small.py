"""rocprofv3 driver: the alignment loop on a small object (cloud scaled by SCALE, default 0.3).
   python3 tools/prof_pose_small.py [B [SCALE]]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
C = (torch.rand(b, 16384, 3, device="cuda", generator=gen) - 0.5) * scale
P = (C[:, :8192] * 0.9).contiguous()
object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=30, render_size=224)
torch.cuda.synchronize()
print("done")
