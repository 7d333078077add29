"""Completed scans per second with several scans in flight (pipeline.complete_scans), bench.py's C2 input.   python3 tools/time_c2_lanes.py"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import pipeline
from genpc_amd.DepthPrompting import DepthPrompting
z13 = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
gt0 = z13["gt"][0]
cc = (gt0.max(0) + gt0.min(0)) / 2
th = np.deg2rad(9.0)
ax = np.array([0.2, 1.0, 0.1]) / np.linalg.norm([0.2, 1.0, 0.1])
Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
Rg = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
gen_np = (((gt0 - cc) / (gt0.max(0) - gt0.min(0)).max()).astype(np.float64) @ Rg.T).astype(np.float32)
part = torch.from_numpy(z13["partial"][0][:8192].copy()).cuda()
gen_s, gt_s = torch.from_numpy(gen_np).cuda(), torch.from_numpy(gt0.copy()).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
img = torch.rand(3, 1024, 1024, device="cuda", generator=g)
cfg = pipeline.default_cfg("cuda", view_num=1024)
job = (part, gen_s, img, gt_s)
for lanes in [int(x) for x in (sys.argv[1:] or ["1", "2", "3", "4"])]:
    dps = [DepthPrompting(cfg) for _ in range(lanes)]
    pipeline.complete_scans([job] * lanes, lanes=lanes, cfg=cfg, dps=dps); torch.cuda.synchronize()
    n = 4 * lanes
    t0 = time.perf_counter(); pipeline.complete_scans([job] * n, lanes=lanes, cfg=cfg, dps=dps); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d lanes: %d scans in %.1f ms = %.1f scans/s" % (lanes, n, dt * 1e3, n / dt), flush=True)
