"""rocprofv3 driver: a few EMD forwards at one shape.  python3 tools/prof_emd.py B N [reps] [uniform|scan]"""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from genpc_amd.loss_functions import emdModule

b, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
kind = sys.argv[4] if len(sys.argv) > 4 else "uniform"
rng = np.random.default_rng(7)
if kind == "scan":          # the bundled scans, partial vs ground truth: most points keep bidding for all 50 rounds
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scans13_fps16384.npz"))
    X = torch.from_numpy(z["partial"][:b, :n].copy()).cuda()
    Y = torch.from_numpy(z["gt"][:b, :n].copy()).cuda()
else:
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
em = emdModule()
em(X, Y, 0.005, 50)
torch.cuda.synchronize()      # (the first call's bidder count -- pinned feedback word -- decides the path of the following ones)
for _ in range(reps):
    d, a = em(X, Y, 0.005, 50)
torch.cuda.synchronize()
print("done", int(a.long().sum()))
