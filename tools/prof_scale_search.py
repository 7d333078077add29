import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from genpc_amd import reg_xyz
rng = np.random.default_rng(3)
tgt = torch.from_numpy(rng.random((4000, 3), dtype=np.float32) - np.float32(0.5)).cuda()
src = (tgt[:3000] * torch.tensor([0.9, 1.1, 1.0], device="cuda")).contiguous()
for _ in range(2):
    reg_xyz.iterative_scale_search(src, tgt, [(0.8, 1.2)] * 3, 10, cd_inv_weight=0.5)
torch.cuda.synchronize()
print("done")
