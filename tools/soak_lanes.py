"""Many scans through pipeline.complete_scans, every product checked against a call of its own.   python3 tools/soak_lanes.py [lanes] [scans]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import pipeline
from genpc_amd.DepthPrompting import DepthPrompting
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
count = int(sys.argv[2]) if len(sys.argv) > 2 else 120
z13 = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
cfg = pipeline.default_cfg("cuda", view_num=1024)
g = torch.Generator(device="cuda"); g.manual_seed(1)
img = torch.rand(3, 1024, 1024, device="cuda", generator=g)
jobs = []
for k in range(6):
    gt = z13["gt"][k]
    cc = (gt.max(0) + gt.min(0)) / 2
    gen_np = ((gt - cc) / (gt.max(0) - gt.min(0)).max()).astype(np.float32)
    jobs.append((torch.from_numpy(z13["partial"][k][:8192].copy()).cuda(), torch.from_numpy(gen_np).cuda(), img, torch.from_numpy(gt.copy()).cuda()))
dp = DepthPrompting(cfg)
ref = [pipeline.complete_scan(*j, cfg=cfg, dp=dp, overlap=False) for j in jobs]
torch.cuda.synchronize()
dps = [DepthPrompting(cfg) for _ in range(lanes)]
t0 = time.perf_counter()
outs = pipeline.complete_scans([jobs[i % 6] for i in range(count)], lanes=lanes, cfg=cfg, dps=dps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
bad = 0
for i, o in enumerate(outs):
    r = ref[i % 6]
    same = o["view"] == r["view"] and all(torch.equal(o[k], r[k]) for k in ("visible", "uv", "fused", "pred_metric_points", "gt_metric_points")) \
        and torch.equal(torch.as_tensor(o["metric"]), torch.as_tensor(r["metric"])) and torch.equal(o["reg"]["source"], r["reg"]["source"])
    bad += 0 if same else 1
from genpc_amd import fps as _fps
print("%d scans, %d lanes: %.1f scans/s, %d differ from a call of their own; farthest-point samplings: %d clouds, %d hand-offs timed out, %d sequences failed the device-side check (both drawn again)"
      % (count, lanes, count / dt, bad, _fps.stats["clouds"], _fps.stats["timed_out"], _fps.stats["failed_check"]))
