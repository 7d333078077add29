"""Host-side cost of one completed scan: process CPU time and a cProfile of complete_scan (bench.py's c2 input).   python3 tools/prof_c2_host.py"""
import os, sys, time, cProfile, pstats, io
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
dp = DepthPrompting(cfg)
for _ in range(2): pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp)
torch.cuda.synchronize()
n = 6
w0, c0 = time.perf_counter(), time.process_time()
for _ in range(n): pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp)
torch.cuda.synchronize()
w1, c1 = time.perf_counter(), time.process_time()
print("per scan: wall %.1f ms, process CPU %.1f ms" % ((w1 - w0) / n * 1e3, (c1 - c0) / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(3): pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp, overlap=False)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
