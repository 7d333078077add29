"""EMD forward: all rounds in one launch (csrc/emd_auction.hip, tune 2) against a launch per round step with the culled
bid (tune 1), in one process: uniform clouds at the bench shapes, the 13 bundled scans, the Waymo crops.
    python3 tools/time_emd_auction.py [quick]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib
from genpc_amd.loss_functions import emdModule
L = _lib.lib
em = emdModule()


def t(X, Y, reps=5):
    em(X, Y, 0.005, 50); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): d, a = em(X, Y, 0.005, 50)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps, float(torch.sqrt(d).mean()), int(a.long().sum())


def ab(tag, X, Y):
    row = []
    for g in (1, 2):
        L.genpc_emd_tune(g, -1)
        row.append(t(X, Y))
    L.genpc_emd_tune(-1, -1)
    same = row[0][1:] == row[1][1:]
    st = L.genpc_emd_status(1, None)
    print("%-34s per-round %8.3f ms | one launch %8.3f ms | x%.2f | same result %s | status %d" % (tag, row[0][0], row[1][0], row[0][0] / row[1][0], same, st), flush=True)


rng = np.random.default_rng(7)
shapes = ((1, 16384), (2, 16384), (4, 16384), (13, 16384), (1, 2048), (8, 2048), (16, 1024), (1, 8192), (2, 8192), (64, 2048), (1, 512), (4, 4096), (8, 32768))
if "quick" in sys.argv: shapes = ((1, 16384), (13, 16384), (1, 2048))
for b, n in shapes:
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    ab("uniform %dx%d" % (b, n), X, Y)
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
P, G = torch.from_numpy(z["partial"]).cuda(), torch.from_numpy(z["gt"]).cuda()
ab("13 scans partial vs GT (B=13)", P, G)
ab("scan 0 partial vs GT (B=1)", P[:1].contiguous(), G[:1].contiguous())
if "quick" not in sys.argv:
    ab("13 scans GT vs partial (B=13)", G, P)
    w = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "waymo_car59_4096.npz"))
    C = torch.from_numpy(np.repeat(w["complete"][None], 59, 0)).cuda()
    ab("59 waymo: car vs crops 4096", C, torch.from_numpy(w["crops"]).cuda())
