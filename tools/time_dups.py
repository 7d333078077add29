"""Chamfer forward on clouds with exact duplicates (pad-repeated: main.py:21-24; resampled with replacement: SURVEY 8d's
C5 generator), the f16 filter's duplicate pre-pass (csrc/nn_dedupe.hip) off / on / left to the policy, with the share of
queries that took the exhaustive pass.   python3 tools/time_dups.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib, chamfer_3D
L = _lib.lib
rng = np.random.default_rng(3)


def stats():
    buf = (ctypes.c_ulonglong * 3)()
    L.genpc_nn_stats(ctypes.cast(buf, ctypes.c_void_p), 1, None)
    return [int(v) for v in buf]


def run(tag, A, Bn):
    b, n, m = A.shape[0], A.shape[1], Bn.shape[1]
    X, Y = torch.from_numpy(A).cuda(), torch.from_numpy(Bn).cuda()
    d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, m, device="cuda")
    i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, m, device="cuda", dtype=torch.int32)
    row = []
    for name, hooks in (("off", 2048), ("on", 4096), ("policy", 0)):
        L.genpc_nn_tune(-1, hooks | 512)
        for _ in range(3): chamfer_3D.forward(X, Y, d1, d2, i1, i2)
        torch.cuda.synchronize(); stats()
        chamfer_3D.forward(X, Y, d1, d2, i1, i2)
        q, ex, _ = stats()
        L.genpc_nn_tune(-1, hooks)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): chamfer_3D.forward(X, Y, d1, d2, i1, i2)
        e1.record(); e1.synchronize()
        row.append("%s %.1f us (%.2f%% exhaustive)" % (name, e0.elapsed_time(e1) / 20 * 1e3, 100.0 * ex / max(q, 1)))
    L.genpc_nn_tune(-1, 0)
    print("%-38s %s" % (tag, " | ".join(row)), flush=True)


for b, n in ((1, 16384), (1, 4096), (8, 32768)):
    A = (rng.random((b, n, 3), dtype=np.float32) - 0.5)
    for uniq in (n, n // 2, n // 5, n // 13):
        base = (rng.random((b, uniq, 3), dtype=np.float32) - 0.5)
        run("%dx%d pad-repeat %d" % (b, n, uniq), A, np.ascontiguousarray(base[:, np.arange(n) % uniq]))
    base = (rng.random((b, n // 2, 3), dtype=np.float32) - 0.5)
    run("%dx%d resampled from %d" % (b, n, n // 2), A, np.stack([base[e][rng.integers(0, n // 2, n)] for e in range(b)]))
