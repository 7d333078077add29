"""Stage times of pipeline.complete_scan at BASELINE config 2's shape.   python3 tools/time_c2.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import pipeline, reg_xyz
from genpc_amd.DepthPrompting import DepthPrompting
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
A = torch.rand(16384, 3, device="cuda", generator=gen) - 0.5
part = ((torch.rand(8192, 3, device="cuda", generator=gen) - 0.5) * 0.9 + 0.01).contiguous()
img = torch.rand(3, 1024, 1024, device="cuda", generator=gen)
cfg = pipeline.default_cfg("cuda", view_num=1024)
dp = DepthPrompting(cfg)
pipeline.complete_scan(part, A, img, A, cfg=cfg, dp=dp)
torch.cuda.synchronize()
def t(f, reps=2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("complete_scan      %.1f ms" % t(lambda: pipeline.complete_scan(part, A, img, A, cfg=cfg, dp=dp)))
print("  getDepth         %.1f ms" % t(lambda: dp.getDepth(part)))
print("    viewpoint_select %.1f ms" % t(lambda: dp.viewpoint_select(part)))
print("  reg              %.1f ms" % t(lambda: reg_xyz.reg(part, A, generative_model=cfg.generative_model, dataset=cfg.dataset, cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True)))
