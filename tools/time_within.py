"""remove_close_points at the reference's size (reg_xyz.py:207-212: ~160 k sampled points of the
completed mesh against the partial scan): radius-limited cell search vs the brute-force filter."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import chamfer_3D
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
rng = np.random.default_rng(0)
gt = g["gt"][0]
big = (gt[rng.integers(0, gt.shape[0], 163840)] + rng.standard_normal((163840, 3)).astype(np.float32) * np.float32(0.003))
Q = torch.from_numpy(np.ascontiguousarray(big[None].astype(np.float32))).cuda()
T = torch.from_numpy(np.ascontiguousarray(g["partial"][0:1])).cuda()
d = torch.empty(1, Q.shape[1], device="cuda"); i = torch.empty(1, Q.shape[1], device="cuda", dtype=torch.int32)
d2 = torch.empty_like(d); i2 = torch.empty_like(i)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
a = t(lambda: chamfer_3D.nm_distance(Q, T, d, i))
b = t(lambda: chamfer_3D.nm_distance_within(Q, T, 1e-4, d2, i2))
same = torch.equal((d < 1e-4), (d2 < 1e-4))
print("163840 x 16384: nm_distance (default filter) %.1f us, nm_distance_within(1e-4) %.1f us, same keep mask: %s, removed %d" % (a, b, same, int((d2 < 1e-4).sum())))
