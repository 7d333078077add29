"""EMD forward times, one line per shape (A/B through GENPC_LIB / GENPC_EMD_* in the caller's environment).
   python tools/emd_ab.py [BxNxIT ...]"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd.loss_functions import emdModule
em = emdModule()
for spec in sys.argv[1:] or ["1x2048x50", "1x8192x50", "1x16384x50", "2x16384x50", "13x16384x50", "64x2048x50", "8x32768x50"]:
    b, n, it = (int(x) for x in spec.split("x"))
    rng = np.random.default_rng(7)
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    d, a = em(X, Y, 0.005, it)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 8
    e0.record()
    for _ in range(reps):
        d, a = em(X, Y, 0.005, it)
    e1.record(); e1.synchronize()
    print("%-14s %8.3f ms   emd %.9g  asum %d" % (spec, e0.elapsed_time(e1) / reps, float(torch.sqrt(d).mean()), int(a.long().sum())))
