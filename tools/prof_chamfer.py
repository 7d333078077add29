"""Minimal driver for rocprofv3: a few Chamfer forward calls at one size.
   rocprofv3 ... -- python3 tools/prof_chamfer.py B N [reps]"""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from genpc_amd import chamfer_3D

b, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
kind = sys.argv[4] if len(sys.argv) > 4 else "uniform"      # uniform | scan (bundled scans, partial vs GT)
rng = np.random.default_rng(20250101)
if kind == "scan":
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
    A = torch.from_numpy(np.ascontiguousarray(np.stack([g["partial"][i % 13][:n] for i in range(b)]))).cuda()
    B = torch.from_numpy(np.ascontiguousarray(np.stack([g["gt"][i % 13][:n] for i in range(b)]))).cuda()
else:
    A = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    B = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).cuda()
d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, n, device="cuda")
i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, n, device="cuda", dtype=torch.int32)
for _ in range(reps):
    chamfer_3D.forward(A, B, d1, d2, i1, i2)
torch.cuda.synchronize()
print("done", int(i1.long().sum().item()))
sys.stdout.flush()

