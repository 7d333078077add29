"""Does what a process did BEFORE its first completed scan change the scan's rate?  (bench.py reads 26.6 where a fresh process reads 28.9)
   python3 tools/c2_after_big.py [big|emd|hpr|fps ...]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import pipeline, _lib
from genpc_amd.DepthPrompting import DepthPrompting
exec(open(os.path.join(ROOT, "tools", "time_c2_streams.py")).read().split("def run(")[0])
def run(n=4, **kw):
    pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp, **kw)
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)
what = sys.argv[1:]
if "dev0" in what:
    cfg = pipeline.default_cfg("cuda:0", view_num=1024); dp = DepthPrompting(cfg)
g = torch.Generator(device="cuda"); g.manual_seed(3)
A = torch.rand(16384, 3, device="cuda", generator=g) - 0.5
if "big" in what:
    from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
    object_pose_optimization(A, (A[:8192] * 0.9).contiguous(), radius=0.02, lr=0.01, iters=200, render_size=224)
if "big8" in what:
    from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
    C8 = torch.rand(8, 16384, 3, device="cuda", generator=g) - 0.5
    object_pose_optimization(C8, (C8[:, :8192] * 0.9).contiguous(), radius=0.02, lr=0.01, iters=200, render_size=224)
if "lanesreg" in what:
    from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
    P8 = (A[:8192] * 0.9).contiguous()
    pipeline.run_in_lanes(lambda li, _: object_pose_optimization(A, P8, radius=0.02, lr=0.01, iters=50, render_size=224), range(6), 6, torch.device("cuda"))
if "lanesonly" in what:
    pipeline.run_in_lanes(lambda li, _: torch.zeros(8, device="cuda").sum(), range(6), 6, torch.device("cuda"))
if "emd" in what:
    from genpc_amd.metric import evaluate_scans
    X = torch.rand(13, 16384, 3, device="cuda", generator=g)
    evaluate_scans(X, X.flip(0).contiguous())
if "fps" in what:
    from genpc_amd.fps import fps_sampling
    fps_sampling(torch.rand(4, 165546, 3, device="cuda", generator=g), 16384)
if "hpr" in what:
    from types import SimpleNamespace
    cfgh = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1, mask_pixel_rate=3, view_num=1024, distance=1.6, downsample_num=10000, removal_radius=10000)
    dph = DepthPrompting(cfgh)
    v = torch.randn(165546, 3, device="cuda", generator=g); v = v / v.norm(dim=1, keepdim=True) * 0.4
    dph.hidden_point_removal(v[:10000].contiguous(), dph.viewpoints, 10000.0)
    dph.hidden_point_removal(v, dph.viewpoints[:2], 10000.0)
torch.cuda.synchronize()
print("%-20s overlap %.1f scans/s, no overlap %.1f" % (" ".join(what) or "fresh", run(), run(overlap=False)), flush=True)
