"""Stage times of one completed scan on the bundled scan (bench.py's c2 input), stages synchronised one by one.
python3 tools/time_c2_scan_stages.py"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import pipeline, reg_xyz
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.ScaleAdapter import ScaleAdapter
from genpc_amd.metric import evaluate_scans
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
sa = ScaleAdapter(cfg)


def t(name, f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); print("%-40s %7.2f ms" % (name, (time.perf_counter() - t0) / reps * 1e3), flush=True); return r


t("complete_scan", lambda: pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp))
t("complete_scan overlap=False", lambda: pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp, overlap=False))
gd = t("stage 1: getDepth", lambda: dp.getDepth(part))
t("  viewpoint_select", lambda: dp.viewpoint_select(part))
t("stage 2a: colorPoint", lambda: sa.colorPoint(gd["uv"], img))
res = t("stage 2b: reg", lambda: reg_xyz.reg(part, gen_s, generative_model=cfg.generative_model, dataset=cfg.dataset, cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True))
fused, (gi,) = t("fuse (+ gt FPS alongside)", lambda: reg_xyz.fuse(res["source"], res["target"], num_points=20000, side_fps=[(gt_s, 16384)]))
print("   fused", tuple(fused.shape))
t("  remove_close_points", lambda: reg_xyz.remove_close_points(res["source"], res["target"]))
t("  remove_noise (k-NN filter)", lambda: reg_xyz.remove_noise_from_point_cloud(fused, std_ratio=2.5))
pred = t("metric: fps_to 16384", lambda: pipeline.fps_to(fused, 16384))
gt = gt_s[gi.long()]
t("metric: evaluate_scans (CD + EMD)", lambda: evaluate_scans(pred[None].contiguous(), gt[None].contiguous()))
