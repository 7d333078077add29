import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib, chamfer_3D
rng = np.random.default_rng(20250101)
for b, n in ((1, 16384), (1, 32768), (13, 16384), (2, 16384), (4, 16384), (8, 32768), (64, 4096)):
    A = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    B = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, n, device="cuda")
    i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, n, device="cuda", dtype=torch.int32)
    row = []
    for hooks in (2048, 1024):
        _lib.lib.genpc_nn_tune(3, hooks)
        for _ in range(10): chamfer_3D.forward(A, B, d1, d2, i1, i2)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): chamfer_3D.forward(A, B, d1, d2, i1, i2)
        e1.record(); e1.synchronize()
        row.append(e0.elapsed_time(e1) / 100 * 1e3)
    _lib.lib.genpc_nn_tune(3, 0)
    print("%dx%d^2: two launches %.1f us, fused %.1f us" % (b, n, row[0], row[1]))
