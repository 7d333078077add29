#!/bin/bash
# A/B of the round from which GetMax + Assign run as one launch (run on the GPU box)
for rf in 1000 2 4 8; do
  echo "GENPC_EMD_RESOLVE_FROM=$rf"
  GENPC_EMD_RESOLVE_FROM=$rf python3 - <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from genpc_amd.loss_functions import emdModule
em = emdModule()
for b, n in ((1, 2048), (1, 16384), (13, 16384), (64, 2048), (8, 32768)):
    rng = np.random.default_rng(7)
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    d, a = em(X, Y, 0.005, 50)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): d, a = em(X, Y, 0.005, 50)
    e1.record(); e1.synchronize()
    print("  %dx%d: %.3f ms  emd %.6f asum %d" % (b, n, e0.elapsed_time(e1) / 5, float(torch.sqrt(d).mean()), int(a.long().sum())))
PY
done
