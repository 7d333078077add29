"""viewpoint_select's pass (best_only) vs the full pass, 1024 viewpoints.   python3 tools/time_vsel.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from types import SimpleNamespace
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.fps import fps_sampling
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                      mask_pixel_rate=3, view_num=1024, distance=1.6, downsample_num=10000, removal_radius=10000)
dp = DepthPrompting(cfg)
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
uni = ((torch.rand(8192, 3, device="cuda", generator=gen) - 0.5) * 0.9 + 0.01).contiguous()
for name, pts in (("uniform volume 8192", uni), ("scan partial 0", torch.from_numpy(g["partial"][0]).cuda()), ("scan gt 5", torch.from_numpy(g["gt"][5]).cuda())):
    sub = pts[fps_sampling(pts, 10000).long()].contiguous() if pts.shape[0] > 10000 else pts
    for best in (False, True):
        dp.hidden_point_removal(sub, dp.viewpoints, 10000.0, best_only=best); torch.cuda.synchronize()
        t0 = time.perf_counter(); vis, cnt, second = dp.hidden_point_removal(sub, dp.viewpoints, 10000.0, best_only=best); torch.cuda.synchronize()
        print("%-20s best_only=%d: %7.2f ms  argmax %d" % (name, best, (time.perf_counter() - t0) * 1e3, int(torch.argmax(cnt))))
