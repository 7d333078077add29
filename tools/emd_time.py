"""Times emdModule forward for a few shapes; A/B the bid pre-filter through
GENPC_EMD_NOFILTER (one subprocess per setting).  python tools/emd_time.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np, torch
from genpc_amd.loss_functions import emdModule
em = emdModule()
out = {}
for spec in sys.argv[1:]:
    b, n, it = (int(x) for x in spec.split("x"))
    rng = np.random.default_rng(7)
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    d, a = em(X, Y, 0.005, it)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record()
    for _ in range(reps): d, a = em(X, Y, 0.005, it)
    e1.record(); e1.synchronize()
    out[spec] = [round(e0.elapsed_time(e1) / reps, 3), float(torch.sqrt(d).mean()), int(a.long().sum())]
print(json.dumps(out))
''' % ROOT

specs = sys.argv[1:] or ["1x2048x50", "1x8192x50", "1x16384x50", "13x16384x50", "64x2048x50", "1x1024x3000"]
for nof in (0, 1):
    env = dict(os.environ)
    if nof:
        env["GENPC_EMD_NOFILTER"] = "1"
    p = subprocess.run([sys.executable, "-c", CHILD] + specs, env=env, capture_output=True, text=True, timeout=600)
    print("filter=%d [ms, emd, asum] %s" % (1 - nof, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-400:]), flush=True)
