#!/bin/bash
# A/B of the library built with / without packed fp32 instructions on the EMD entry point (run on the GPU box; rebuilds in place there)
# tools/emd_ab_packed.sh
set -u
cd "$GRAFT_REPO_ROOT"
T='
import sys, json, numpy as np, torch
sys.path.insert(0, ".")
from genpc_amd import _lib
from genpc_amd.loss_functions import emdModule
em = emdModule(); L = _lib.lib
z = np.load("tests/golden/scans13_fps16384.npz")
out = {}
def t(name, X, Y, reps=5):
    d, a = em(X, Y, 0.005, 50); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): d, a = em(X, Y, 0.005, 50)
    e1.record(); e1.synchronize()
    out[name] = round(e0.elapsed_time(e1) / reps, 3)
rng = np.random.default_rng(7)
for b, n in ((1, 16384), (13, 16384), (1, 2048)):
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda(); Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    for mode, tag in ((0, "auto"), (1, "per-round")):
        prev = L.genpc_emd_tune(mode, -1) if mode else None
        t("%dx%d %s" % (b, n, tag), X, Y)
        if mode: L.genpc_emd_tune(prev, -1)
P = torch.from_numpy(z["partial"].astype(np.float32)).cuda(); G = torch.from_numpy(z["gt"].astype(np.float32)).cuda()
t("13 scans auto", P, G, 3)
print(json.dumps(out))
'
echo "== built without packed fp32 (shipped)"; python3 -c "$T" 2>&1 | tail -1
GENPC_PACKED_FP32=1 python3 -m genpc_amd.build --force > /dev/null 2>&1
echo "== built WITH packed fp32"; python3 -c "$T" 2>&1 | tail -1
