"""viewpoint_select's shape for rocprofv3: 1024 viewpoints x 10000 points of scan 0.   python3 tools/prof_hpr1024.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from types import SimpleNamespace
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.fps import fps_sampling
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                      mask_pixel_rate=3, view_num=1024, distance=1.6, downsample_num=10000, removal_radius=10000)
dp = DepthPrompting(cfg)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
pts = torch.from_numpy(g["partial"][0]).cuda()
sub = pts[fps_sampling(pts, 10000).long()].contiguous()
for _ in range(2):
    dp.hidden_point_removal(sub, dp.viewpoints, 10000.0)
torch.cuda.synchronize()
