"""Experiment: how many queries does the cell search's near phase leave unresolved, and what does it cost?
   GENPC_GRID_NEARONLY=1 python tools/grid_near.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib, chamfer_3D
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
rng = np.random.default_rng(1)
cases = {"scans 13x16384 partial/gt": (g["partial"], g["gt"]),
         "scans 1x16384": (g["partial"][:1], g["gt"][:1]),
         "uniform 13x16384": (rng.random((13, 16384, 3), dtype=np.float32), rng.random((13, 16384, 3), dtype=np.float32)),
         "scan 8x8192 vs 16384": (g["partial"][:8, :8192], g["gt"][:8])}
for name, (a, b) in cases.items():
    A, B = torch.from_numpy(np.ascontiguousarray(a)).cuda(), torch.from_numpy(np.ascontiguousarray(b)).cuda()
    bs, n, m = A.shape[0], A.shape[1], B.shape[1]
    d1 = torch.empty(bs, n, device="cuda"); d2 = torch.empty(bs, m, device="cuda")
    i1 = torch.empty(bs, n, device="cuda", dtype=torch.int32); i2 = torch.empty(bs, m, device="cuda", dtype=torch.int32)
    _lib.lib.genpc_nn_tune(4, 512)
    buf = (ctypes.c_ulonglong * 3)()
    _lib.lib.genpc_nn_stats(ctypes.cast(buf, ctypes.c_void_p), 1, None)
    chamfer_3D.forward(A, B, d1, d2, i1, i2)
    _lib.lib.genpc_nn_stats(ctypes.cast(buf, ctypes.c_void_p), 1, None)
    _lib.lib.genpc_nn_tune(4, 0)
    for _ in range(3): chamfer_3D.forward(A, B, d1, d2, i1, i2)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): chamfer_3D.forward(A, B, d1, d2, i1, i2)
    e1.record(); e1.synchronize()
    tg = e0.elapsed_time(e1) / 10 * 1e3
    _lib.lib.genpc_nn_tune(3, 0)
    for _ in range(3): chamfer_3D.forward(A, B, d1, d2, i1, i2)
    e0.record()
    for _ in range(10): chamfer_3D.forward(A, B, d1, d2, i1, i2)
    e1.record(); e1.synchronize()
    tf = e0.elapsed_time(e1) / 10 * 1e3
    print("%-28s queries %7d unresolved %7d (%.1f %%)  near-only grid %.1f us   f16 %.1f us" % (name, buf[0], buf[1], 100.0 * buf[1] / max(1, buf[0]), tg, tf))
