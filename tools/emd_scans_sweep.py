"""EMD on the 13 bundled scans (partial vs ground truth, 16384 points) and a single scan under environment settings.   python3 tools/emd_scans_sweep.py K=V,K=V ..."""
import json, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r'''
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from genpc_amd.loss_functions import emdModule
em = emdModule(); out = {}
z = np.load(%r)
P = torch.from_numpy(z["partial"].astype(np.float32)).cuda(); G = torch.from_numpy(z["gt"].astype(np.float32)).cuda()
for name, X, Y in (("13 scans", P, G), ("scan 0", P[:1].contiguous(), G[:1].contiguous())):
    for _ in range(3): d, a = em(X, Y, 0.005, 50)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): d, a = em(X, Y, 0.005, 50)
    e1.record(); e1.synchronize()
    out[name] = round(e0.elapsed_time(e1) / 4, 3)
print(json.dumps(out))
''' % (ROOT, os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
settings = [dict(kv.split("=") for kv in arg.split(",") if kv) for arg in sys.argv[1:]] or [{}]
for s in settings:
    env = dict(os.environ); env.update(s)
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
    print("%-70s %s" % (s, p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]), flush=True)
