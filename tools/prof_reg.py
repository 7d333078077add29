"""reg() at BASELINE config 2's shape for rocprofv3 (8192-point partial scan, 16384-point generated shape).   python3 tools/prof_reg.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd import pipeline, reg_xyz
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
A = torch.rand(16384, 3, device="cuda", generator=gen) - 0.5
part = ((torch.rand(8192, 3, device="cuda", generator=gen) - 0.5) * 0.9 + 0.01).contiguous()
cfg = pipeline.default_cfg("cuda", view_num=1024)
print("pose clouds:", reg_xyz.voxel_down_sample(A, 0.02).shape, reg_xyz.voxel_down_sample(part, 0.02).shape)
for _ in range(2):
    reg_xyz.reg(part, A, generative_model=cfg.generative_model, dataset=cfg.dataset, cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True)
torch.cuda.synchronize()
