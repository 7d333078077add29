"""rocprofv3 driver: farthest point sampling (4 x 165546 -> 16384) and voxel down-sampling (163840 points, 0.03)."""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.fps import fps_sampling
from genpc_amd.reg_xyz import voxel_down_sample

g = torch.Generator(device="cuda")
g.manual_seed(20250101)
big = torch.rand(4, 165546, 3, device="cuda", generator=g)
idx = fps_sampling(big, 16384)
v = voxel_down_sample(big[0, :163840].contiguous() - 0.5, 0.03)
torch.cuda.synchronize()
print("done", int(idx[0, 5]), v.shape)
