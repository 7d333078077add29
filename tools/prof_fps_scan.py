"""rocprofv3 driver: the two samplings of a completed scan's tail on a bundled scan's surface (csrc/fps_grid.hip):
24576 -> 20000 (the fused cloud) and 20000 -> 16384 (the metric's), plus their device-side verification."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd.fps import fps_sampling
z = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
surf = torch.from_numpy(np.concatenate([z["partial"][0][:8192], z["gt"][0]]).astype(np.float32)).cuda()
for _ in range(3):
    a = fps_sampling(surf, 20000)
    b = fps_sampling(surf[:20000].contiguous(), 16384)
torch.cuda.synchronize()
print("done", int(a[5]), int(b[5]))
