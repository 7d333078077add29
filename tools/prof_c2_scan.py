"""pipeline.complete_scan on the bundled scan (bench.py's c2_pipeline_8192_scans_per_s input) for rocprofv3.   python3 tools/prof_c2_scan.py"""
import os, sys, time
# (under rocprofv3's kernel tracing every launch costs tens of microseconds more, which biases the alignment loop's own
#  timing probe towards its single-launch path: the traced run takes the path the untraced run takes)
os.environ.setdefault("GENPC_POSE_SEEDED", "0")
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
pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp)
torch.cuda.synchronize()
print("one scan: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
