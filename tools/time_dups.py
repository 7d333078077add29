"""Chamfer forward when one cloud is a pad-repeated subsample (main.py:21-24 repeats points when the scan has fewer
than k): exact duplicates are the NN filter's tie case.   python3 tools/time_dups.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib, chamfer_3D
rng = np.random.default_rng(3)
n = 16384
A = (rng.random((1, n, 3), dtype=np.float32) - 0.5)
for uniq in (16384, 9000, 6000, 5000, 3000):
    base = (rng.random((uniq, 3), dtype=np.float32) - 0.5)
    Bn = base[np.arange(n) % uniq][None].copy()
    X, Y = torch.from_numpy(A).cuda(), torch.from_numpy(Bn).cuda()
    d1 = torch.empty(1, n, device="cuda"); d2 = torch.empty(1, n, device="cuda")
    i1 = torch.empty(1, n, device="cuda", dtype=torch.int32); i2 = torch.empty(1, n, device="cuda", dtype=torch.int32)
    for _ in range(3): chamfer_3D.forward(X, Y, d1, d2, i1, i2)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): chamfer_3D.forward(X, Y, d1, d2, i1, i2)
    e1.record(); e1.synchronize()
    print("unique targets %5d of %d: %.1f us per call" % (uniq, n, e0.elapsed_time(e1) / 10 * 1e3))
