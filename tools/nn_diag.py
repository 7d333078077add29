"""Diagnostics for the split-bf16 NN filter: runs chamfer_3D.forward normally and with
GENPC_NN_DEBUG=32 (approximate minimum + candidate count instead of the result) in two
subprocesses and prints the filter's error and candidate statistics.
python tools/nn_diag.py [BxNxM ...]"""
import json, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CHILD = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, %r)
from genpc_amd import chamfer_3D
out = {}
for spec in sys.argv[1:]:
    b, n, m = [int(x) for x in spec.split("x")]
    rng = np.random.default_rng(7)
    A = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    B = torch.from_numpy(rng.random((b, m, 3), dtype=np.float32) - np.float32(0.5)).cuda()
    d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, m, device="cuda")
    i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, m, device="cuda", dtype=torch.int32)
    chamfer_3D.forward(A, B, d1, d2, i1, i2)
    torch.cuda.synchronize()
    np.savez("/tmp/nn_diag_%%s_%%s.npz" %% (spec, sys.argv[0] and __import__("os").environ.get("GENPC_NN_DEBUG", "0")), d1=d1.cpu().numpy(), i1=i1.cpu().numpy())
""" % ROOT
for spec in sys.argv[1:] or ["1x2048x2048", "1x16384x16384"]:
    for dbg in ("0", "32"):
        env = dict(os.environ, GENPC_NN_DEBUG=dbg)
        p = subprocess.run([sys.executable, "-c", CHILD, spec], env=env, capture_output=True, text=True, timeout=300)
        if p.returncode: print(p.stderr[-500:])
    import numpy as np
    r = np.load("/tmp/nn_diag_%s_0.npz" % spec); g = np.load("/tmp/nn_diag_%s_32.npz" % spec)
    err = g["d1"].astype(np.float64) - r["d1"].astype(np.float64)
    cnt = g["i1"] & 0xffff; flg = g["i1"] >> 16
    print(spec, "approx-exact: min %.3e max %.3e mean|.| %.3e; exact d mean %.3e" % (err.min(), err.max(), np.abs(err).mean(), r["d1"].mean()))
    print("   candidates/query: mean %.3f max %d hist %s ; flagged %d of %d" % (cnt.mean(), cnt.max(), np.bincount(cnt.ravel())[:8], int(flg.sum()), flg.size))
