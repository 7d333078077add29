"""rocprofv3 driver: getUvs for the reference's 1024 cameras x 71372 points.  python3 tools/prof_uvs.py [reps]"""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from types import SimpleNamespace
import torch
from genpc_amd.DepthPrompting import DepthPrompting

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, padding=0.15, rescale=True, point_size=1, mask_pixel_rate=3,
                      view_num=1024, distance=1.6)
dp = DepthPrompting(cfg)
g = torch.Generator(device="cuda")
g.manual_seed(20250101)
pts = (torch.rand(71372, 3, device="cuda", generator=g) - 0.5) * 0.8
for _ in range(reps):
    uv, depth, _ = dp.getUvs(dp.cameras, pts, want_transformed=False)
torch.cuda.synchronize()
print("done", float(uv[3, 5, 0]))
