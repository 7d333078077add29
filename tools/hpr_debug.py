"""GPU mask vs the clipping oracle on one cloud / radius (for chasing a mismatch under the GENPC_HPR_NOCULL knobs).
   python3 tools/hpr_debug.py [radius]"""
import os, sys, ctypes
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib
from oracle import oracle
radius = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
rng = np.random.default_rng(0)
out = {}
for n in (1, 2, 3, 50, 700, 3001):
    out["ball%d" % n] = (rng.random((n, 3)) - 0.5).astype(np.float32)
    v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    out["sphere%d" % n] = (v * 0.5).astype(np.float32)
    out["shell%d" % n] = (v * (0.45 + 0.05 * rng.random((n, 1)))).astype(np.float32)
EYES = np.array([[0, 0, 3.0], [2.0, 1.0, -1.5], [-1.1, 0.3, 0.9], [0.2, -1.6, 0.1]])
for name in ("sphere3001", "shell3001", "ball3001"):
    P = out[name]
    Pt = torch.from_numpy(P).cuda(); E = torch.from_numpy(EYES).cuda()
    vis = torch.zeros(4, len(P), device="cuda", dtype=torch.uint8); cnt = torch.zeros(4, device="cuda", dtype=torch.int32)
    second = ctypes.c_int(0)
    rc = _lib.lib.genpc_hpr_visibility(4, len(P), _lib.ptr(Pt), _lib.ptr(E), radius, _lib.ptr(vis), _lib.ptr(cnt), ctypes.addressof(second), None)
    torch.cuda.synchronize()
    g = vis.cpu().numpy().astype(bool)
    o = np.stack([oracle.hpr_visibility(P, e, radius) for e in EYES])
    bad = np.argwhere(g != o)
    print(name, "rc", rc, "mismatches", len(bad), [(int(a), int(b), bool(g[a, b]), bool(o[a, b])) for a, b in bad[:4]], "second", second.value)
