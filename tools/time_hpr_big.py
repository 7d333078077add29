"""Hidden-point removal at the two shapes bench.py quotes (1024 views x 10000 FPS-ordered points; 2 views x 165546), wall
time per call with the stream drained once at the end of five calls.   python3 tools/time_hpr_big.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from types import SimpleNamespace
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.fps import fps_sampling

rng = np.random.default_rng(5)
v = rng.normal(size=(165546, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
scan = torch.from_numpy((v * (0.3 + 0.2 * np.abs(np.sin(3 * v[:, :1])))).astype(np.float32)).cuda()
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                      mask_pixel_rate=3, view_num=1024, distance=1.6, downsample_num=10000, removal_radius=10000)
dp = DepthPrompting(cfg)
sub = scan[fps_sampling(scan, 10000).long()].contiguous()
for name, pts, eyes, best in (("1024 x 10000", sub, dp.viewpoints, False), ("1024 x 10000 best view", sub, dp.viewpoints, True),
                              ("64 x 10000", sub, dp.viewpoints[:64], False), ("2 x 165546", scan, dp.viewpoints[:2], False)):
    dp.hidden_point_removal(pts, eyes, 10000.0, best_only=best); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): vis, cnt, _ = dp.hidden_point_removal(pts, eyes, 10000.0, best_only=best)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    second = dp.hidden_point_removal(pts, eyes, 10000.0, best_only=best, want_second=True)[2]
    print("%-24s %8.2f ms per call (host returned after %.2f ms per call)  visible %.4f  wave-per-point points %d"
          % (name, (t2 - t0) / 5 * 1e3, (t1 - t0) / 5 * 1e3, float(cnt.float().mean()) / pts.shape[0], second))
