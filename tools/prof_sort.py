"""rocprofv3 driver: sorted-mode Chamfer on the 13 bundled scans.   python3 tools/prof_sort.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib, chamfer_3D
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
X, Y = torch.from_numpy(g["partial"]).cuda(), torch.from_numpy(g["gt"]).cuda()
b, n, m = X.shape[0], X.shape[1], Y.shape[1]
d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, m, device="cuda")
i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, m, device="cuda", dtype=torch.int32)
_lib.lib.genpc_nn_tune(3, 4096)
for _ in range(5): chamfer_3D.forward(X, Y, d1, d2, i1, i2)
torch.cuda.synchronize()
print("done")
