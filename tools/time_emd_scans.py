import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib
from genpc_amd.loss_functions import emdModule
L = _lib.lib; em = emdModule()
z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scans13_fps16384.npz"))
def t(X, Y, reps=3):
    em(X, Y, 0.005, 50); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): d, a = em(X, Y, 0.005, 50)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for s in range(13):
    P = torch.from_numpy(z["partial"][s:s+1].copy()).cuda(); G = torch.from_numpy(z["gt"][s:s+1].copy()).cuda()
    row = []
    for g in (0, 1):
        L.genpc_emd_tune(g, -1); row.append(t(P, G))
    L.genpc_emd_tune(-1, -1)
    print("scan %2d %s: tiled %7.3f ms culled %7.3f ms" % (s, z["ids"][s], row[0], row[1]), flush=True)
