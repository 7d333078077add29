"""Hidden-point removal on the bundled real scans: 64 viewpoints x 10000 FPS-ordered points, radius 10000.
   python3 tools/time_hpr_scans.py        (GENPC_HPR_NOCULL=16: without the early accept)"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from types import SimpleNamespace
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.fps import fps_sampling
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                      mask_pixel_rate=3, view_num=64, distance=1.6, downsample_num=10000, removal_radius=10000)
dp = DepthPrompting(cfg)
for name in ("partial", "gt"):
    for k in (0, 5, 9):
        pts = torch.from_numpy(g[name][k]).cuda()
        sub = pts[fps_sampling(pts, 10000).long()].contiguous()
        dp.hidden_point_removal(sub, dp.viewpoints, 10000.0); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): vis, cnt, second = dp.hidden_point_removal(sub, dp.viewpoints, 10000.0)
        torch.cuda.synchronize()
        print("%-8s scan %d: %6.2f ms  visible %.3f  second pass %d" % (name, k, (time.perf_counter() - t0) / 3 * 1e3, float(cnt.float().mean()) / 10000, second))
