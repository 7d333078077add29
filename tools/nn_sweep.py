"""Experiment driver: times genpc_chamfer_forward for kernel variants selected
through GENPC_NN_* environment variables (one subprocess per variant, since the
library reads them once).  python tools/nn_sweep.py [sizes...]"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

CHILD = r'''
import sys, json, os
sys.path.insert(0, %r)
import numpy as np, torch
from genpc_amd import chamfer_3D
out = {}
for spec in sys.argv[1:]:
    dims = [int(x) for x in spec.split("x")]
    b, n = dims[0], dims[1]
    m = dims[2] if len(dims) > 2 else n
    rng = np.random.default_rng(20250101)
    A = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    B = torch.from_numpy(rng.random((b, m, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, m, device="cuda")
    i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, m, device="cuda", dtype=torch.int32)
    for _ in range(5): chamfer_3D.forward(A, B, d1, d2, i1, i2)
    torch.cuda.synchronize()
    reps = 50
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): chamfer_3D.forward(A, B, d1, d2, i1, i2)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out[spec] = [round(ms * 1e3, 1), round(2.0 * b * n * m / ms / 1e6, 1), int(i1.long().sum().item())]
print(json.dumps(out))
''' % ROOT


def main():
    sizes = sys.argv[1:] or ["1x2048", "1x16384", "1x32768", "13x16384"]
    dbgs = [int(x) for x in os.environ.get("SWEEP_DEBUG", "0").split(",")]
    rs = [int(x) for x in os.environ.get("SWEEP_R", "0,2,4").split(",")]
    ws = [int(x) for x in os.environ.get("SWEEP_WPS", "2,4,8").split(",")]
    # SWEEP_VARIANTS="GENPC_NN_PATH=valu;GENPC_NN_Q=1,GENPC_NN_U=2;..." runs exactly those environments
    if os.environ.get("SWEEP_VARIANTS"):
        for var in os.environ["SWEEP_VARIANTS"].split(";"):
            env = dict(os.environ)
            for kv in var.split(","):
                if kv:
                    k, v = kv.split("=")
                    env[k] = v
            p = subprocess.run([sys.executable, "-c", CHILD] + sizes, env=env, capture_output=True, text=True, timeout=300)
            line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]
            print("%s [us, Gpair/s, idxsum] %s" % (var or "default", line), flush=True)
        return
    for r, wps, dbg in itertools.product(rs, ws, dbgs):
        env = dict(os.environ, GENPC_NN_R=str(r), GENPC_NN_WPS=str(wps), GENPC_NN_DEBUG=str(dbg))
        p = subprocess.run([sys.executable, "-c", CHILD] + sizes, env=env, capture_output=True, text=True, timeout=300)
        line = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]
        print("R=%d blocks/CU=%d dbg=%d [us, Gpair/s, idxsum] %s" % (r, wps, dbg, line), flush=True)


if __name__ == "__main__":
    main()
