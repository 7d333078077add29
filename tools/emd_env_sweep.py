"""One-launch / launch-per-round EMD under a few environment settings (a process per setting).   python3 tools/emd_env_sweep.py"""
import json, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from genpc_amd.loss_functions import emdModule
em = emdModule(); out = {}
rng = np.random.default_rng(7)
for b, n in ((1, 16384), (13, 16384)):
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda(); Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    for _ in range(2): d, a = em(X, Y, 0.005, 50)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): d, a = em(X, Y, 0.005, 50)
    e1.record(); e1.synchronize()
    out["%%dx%%d" %% (b, n)] = round(e0.elapsed_time(e1) / 6, 3)
print(json.dumps(out))
''' % ROOT
settings = [dict(kv.split("=") for kv in arg.split(",") if kv) for arg in sys.argv[1:]] or [{}]
for s in settings:
    env = dict(os.environ); env.update(s)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print("%-60s %s" % (s, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]), flush=True)
