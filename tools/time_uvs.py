import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from types import SimpleNamespace
import torch
from genpc_amd.DepthPrompting import DepthPrompting
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, padding=0.15, rescale=True, point_size=1, mask_pixel_rate=3, view_num=1024, distance=1.6)
dp = DepthPrompting(cfg)
g = torch.Generator(device="cuda"); g.manual_seed(20250101)
pts = (torch.rand(71372, 3, device="cuda", generator=g) - 0.5) * 0.8
for _ in range(3): dp.getUvs(dp.cameras, pts, want_transformed=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(int(os.environ.get("UVS_REPS", "20"))): dp.getUvs(dp.cameras, pts, want_transformed=False)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / int(os.environ.get("UVS_REPS", "20"))
alg = 12 * 71372 + 1024 * 71372 * 12          # the cloud once, uv + depth per (camera, point)
print("get_uvs 1024 x 71372: %.1f us, %.2f TB/s algorithmic (%.0f %% of 8 TB/s, %.0f %% of 6.29)" % (t * 1e6, alg / t / 1e12, alg / t / 8e10, alg / t / 6.29e10))
