"""k-NN mean distance (the outlier filter's statistic) on the fused cloud of a scan: grid against exhaustive.  GENPC_KNN_GRID=0/1"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import reg_xyz
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scans13_fps16384.npz"))
for name, P in (("scan 20000", torch.from_numpy(np.concatenate([z["gt"][0], z["partial"][0][:3616]])).cuda()),
                ("uniform 20000", torch.rand(20000, 3, device="cuda")), ("scan 163840 (x10 jitter)", (torch.from_numpy(z["gt"][0]).cuda().repeat(10, 1) + 1e-3 * torch.randn(163840, 3, device="cuda")).contiguous())):
    m = reg_xyz.knn_mean_distance(P, 20); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): m = reg_xyz.knn_mean_distance(P, 20)
    torch.cuda.synchronize()
    print("%-28s %8.3f ms  (checksum %.9g)" % (name, (time.perf_counter() - t0) / 5 * 1e3, float(m.double().sum())), flush=True)
