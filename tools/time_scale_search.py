"""iterative_scale_search at reg()'s shapes (8192-point partial, 16384-point generated shape, pcn voxel sizes).
   python tools/time_scale_search.py      [GENPC_NN_PATH=grid for the cell-sorted search]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd import reg_xyz
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
A = torch.rand(16384, 3, device="cuda", generator=gen) - 0.5
part = ((torch.rand(8192, 3, device="cuda", generator=gen) - 0.5) * 0.9 + 0.01).contiguous()
if os.environ.get("SHAPE") == "surface":          # a bumpy ellipsoid's surface; the partial scan sees the z > -0.05 side
    d = torch.randn(16384, 3, device="cuda", generator=gen); d = d / d.norm(dim=1, keepdim=True)
    r = 0.35 + 0.06 * torch.sin(7 * d[:, 0]) * torch.cos(5 * d[:, 1])
    A = (d * r[:, None] * torch.tensor([1.0, 0.7, 0.5], device="cuda")).contiguous()
    vis = A[A[:, 2] > -0.05]
    part = (vis[torch.randint(len(vis), (8192,), device="cuda", generator=gen)] * 0.93 + 0.004 * torch.randn(8192, 3, device="cuda", generator=gen)).contiguous()
tgt = reg_xyz.voxel_down_sample(A, 0.04)
print("shapes", part.shape, tgt.shape)
for _ in range(2):
    out = reg_xyz.iterative_scale_search(part, tgt, [(0.8, 1.2)] * 3, 10, cd_inv_weight=0.5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = reg_xyz.iterative_scale_search(part, tgt, [(0.8, 1.2)] * 3, 10, cd_inv_weight=0.5)
torch.cuda.synchronize()
print("scale search: %.2f ms per call, loss %.9g" % ((time.perf_counter() - t0) / 5 * 1e3, out[1]), out[0].diagonal())
