"""viewpoint_select's shape: 1024 viewpoints x 10000 points (synthetic blob and one real scan).  python3 tools/time_hpr_1024.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from types import SimpleNamespace
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.fps import fps_sampling
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                      mask_pixel_rate=3, view_num=1024, distance=1.6, downsample_num=10000, removal_radius=10000)
dp = DepthPrompting(cfg)
rng = np.random.default_rng(5)
v = rng.normal(size=(165546, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
blob = torch.from_numpy((v * (0.3 + 0.2 * np.abs(np.sin(3 * v[:, :1])))).astype(np.float32)).cuda()
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
for name, pts in (("blob", blob), ("scan partial 0", torch.from_numpy(g["partial"][0]).cuda()), ("scan gt 5", torch.from_numpy(g["gt"][5]).cuda())):
    sub = pts[fps_sampling(pts, 10000).long()].contiguous()
    dp.hidden_point_removal(sub, dp.viewpoints, 10000.0); torch.cuda.synchronize()
    t0 = time.perf_counter(); vis, cnt, second = dp.hidden_point_removal(sub, dp.viewpoints, 10000.0); torch.cuda.synchronize()
    print("%-16s 1024 x 10000: %7.2f ms  second pass %d" % (name, (time.perf_counter() - t0) * 1e3, second))
