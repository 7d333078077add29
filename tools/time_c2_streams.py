"""One completed scan at a time on the NULL stream, on a stream of its own, and through pipeline.complete_scans(lanes=1); with and
without the alignment loop's side stream.   python3 tools/time_c2_streams.py"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import pipeline, _lib
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


def run(n=4):
    pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


for dual in (1, 0):
    _lib.lib.genpc_pose_dual(dual)
    print("dual %d: null stream %.1f scans/s" % (dual, run()), flush=True)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        print("dual %d: own stream  %.1f scans/s" % (dual, run()), flush=True)
_lib.lib.genpc_pose_dual(-1)
dps = [dp]
pipeline.complete_scans([(part, gen_s, img, gt_s)], lanes=1, cfg=cfg, dps=dps); torch.cuda.synchronize()
t0 = time.perf_counter(); pipeline.complete_scans([(part, gen_s, img, gt_s)] * 4, lanes=1, cfg=cfg, dps=dps); torch.cuda.synchronize()
print("complete_scans(lanes=1): %.1f scans/s" % (4 / (time.perf_counter() - t0)))
