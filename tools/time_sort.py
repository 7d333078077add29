"""Sorted mode (hook 4096) against the default on several workloads.   python3 tools/time_sort.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
import bench
from genpc_amd import _lib, chamfer_3D
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
rng = np.random.default_rng(20250101)
def uni(b, n): return rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)
sc = [bench.synth_scan(k, 32768) for k in range(8)]
cases = [("uniform 1x16384^2", uni(1, 16384), uni(1, 16384)), ("uniform 13x16384^2", uni(13, 16384), uni(13, 16384)),
         ("scans 13x16384 partial vs gt", g["partial"], g["gt"]), ("scans 1x16384", g["partial"][:1], g["gt"][:1]),
         ("uniform 1x32768^2", uni(1, 32768), uni(1, 32768)),
         ("synthetic scans 8x32768 complete vs partial", np.stack([x[0] for x in sc]), np.stack([x[1] for x in sc])),
         ("surface 8x32768 (sphere shells)", None, None)]
for name, A, B in cases:
    if A is None:
        v = rng.normal(size=(8, 32768, 3)); v /= np.linalg.norm(v, axis=2, keepdims=True)
        A = (v * 0.5).astype(np.float32)
        w = rng.normal(size=(8, 32768, 3)); w /= np.linalg.norm(w, axis=2, keepdims=True)
        B = (w * 0.5).astype(np.float32)
    X, Y = torch.from_numpy(np.ascontiguousarray(A)).cuda(), torch.from_numpy(np.ascontiguousarray(B)).cuda()
    b, n, m = X.shape[0], X.shape[1], Y.shape[1]
    d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, m, device="cuda")
    i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, m, device="cuda", dtype=torch.int32)
    row = []
    for hooks in (0, 4096):
        _lib.lib.genpc_nn_tune(3, hooks)
        for _ in range(5): chamfer_3D.forward(X, Y, d1, d2, i1, i2)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): chamfer_3D.forward(X, Y, d1, d2, i1, i2)
        e1.record(); e1.synchronize()
        row.append(e0.elapsed_time(e1) / 30 * 1e3)
        if hooks == 0: ref = (d1.clone(), i1.clone(), d2.clone(), i2.clone())
        else: same = all(torch.equal(x, y) for x, y in zip(ref, (d1, i1, d2, i2)))
    _lib.lib.genpc_nn_tune(3, 0)
    print("%-46s default %8.1f us, sorted %8.1f us  identical: %s" % (name, row[0], row[1], same))
