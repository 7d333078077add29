"""rocprofv3 driver: hidden-point removal at getDepth's shape (2 views x 165546 points) and at
viewpoint_select's (64 x 10000).   python3 tools/prof_hpr.py [big|small]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from types import SimpleNamespace
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.fps import fps_sampling

which = sys.argv[1] if len(sys.argv) > 1 else "big"
rng = np.random.default_rng(5)
v = rng.normal(size=(165546, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
cloud = torch.from_numpy((v * (0.3 + 0.2 * np.abs(np.sin(3 * v[:, :1])))).astype(np.float32)).cuda()
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                      mask_pixel_rate=3, view_num=64, distance=1.6, downsample_num=10000, removal_radius=10000)
dp = DepthPrompting(cfg)
if which == "big":
    pts, eyes = cloud, dp.viewpoints[:2]
else:
    pts, eyes = cloud[fps_sampling(cloud, 10000).long()].contiguous(), dp.viewpoints
for _ in range(2):
    vis, cnt, second = dp.hidden_point_removal(pts, eyes, 10000.0, want_second=True)
torch.cuda.synchronize()
print("visible", cnt.tolist()[:4], "second pass", second)
