"""Times genpc_hpr_visibility at viewpoint_select's shape (64 views x 10000 FPS-ordered points) and at
getDepth's (2 views x the whole scan), against qhull on the host.   python3 tools/time_hpr.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from types import SimpleNamespace
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.fps import fps_sampling
from oracle import hpr

rng = np.random.default_rng(5)
v = rng.normal(size=(165546, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
cloud = torch.from_numpy((v * (0.3 + 0.2 * np.abs(np.sin(3 * v[:, :1])))).astype(np.float32)).cuda()
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                      mask_pixel_rate=3, view_num=64, distance=1.6, downsample_num=10000, removal_radius=10000)
dp = DepthPrompting(cfg)
sub = cloud[fps_sampling(cloud, 10000).long()].contiguous()
for name, pts, eyes in (("64 x 10000 (FPS order)", sub, dp.viewpoints), ("2 x 165546 (input order)", cloud, dp.viewpoints[:2]),
                        ("64 x 10000 (shuffled)", sub[torch.randperm(10000, device="cuda")].contiguous(), dp.viewpoints)):
    for radius in (10000.0, 100.0):
        dp.hidden_point_removal(pts, eyes, radius); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): vis, cnt, second = dp.hidden_point_removal(pts, eyes, radius)
        second = dp.hidden_point_removal(pts, eyes, radius, want_second=True)[2]
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        t0 = time.perf_counter()
        ref = hpr.visible_counts(pts.cpu().numpy(), np.asarray(eyes)[:2], radius)
        q = (time.perf_counter() - t0) / 2 * len(eyes) * 1e3
        print("%-26s R=%-7g %8.2f ms  (qhull, 1 core, extrapolated from 2 views: %8.0f ms)  visible %.3f second pass %d  counts equal: %s"
              % (name, radius, ms, q, float(cnt.float().mean()) / pts.shape[0], second, bool((cnt[:2].cpu().numpy() == ref).all())))
