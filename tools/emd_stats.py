"""What the culled bid does per round (genpc_emd_stats): python3 tools/emd_stats.py B N uniform|scan [rounds...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib, emd
from genpc_amd.loss_functions.emd.emd_module import alloc_state
L = _lib.lib
b, n, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rounds = [int(x) for x in sys.argv[4:]] or [1, 2, 3, 5, 10, 20, 35, 50]
rng = np.random.default_rng(7)
if kind == "scan":
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scans13_fps16384.npz"))
    X = torch.from_numpy(z["partial"][:b, :n].copy()).cuda(); Y = torch.from_numpy(z["gt"][:b, :n].copy()).cuda()
else:
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda(); Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
buf = (ctypes.c_ulonglong * 8)()
prev = np.zeros(8)
L.genpc_emd_tune(1, 1)
for k in rounds:
    st = alloc_state(b, n, n, X.device)
    L.genpc_emd_stats(ctypes.cast(buf, ctypes.c_void_p), 1, None)
    emd.forward(X, Y, st["dist"], st["assignment"], st["price"], st["assignment_inv"], st["bid"], st["bid_increments"],
                st["max_increments"], st["unass_idx"], st["unass_cnt"], st["unass_cnt_sum"], st["cnt_tmp"], st["max_idx"], 0.005, k)
    L.genpc_emd_stats(ctypes.cast(buf, ctypes.c_void_p), 1, None)
    cur = np.array([float(v) for v in buf])
    d = cur - prev          # rounds (prev_k, k]  (the last round of a k-round call is forced: same bids)
    prev = cur
    bd = max(d[0], 1)
    print("rounds <=%3d: bidder-rounds %9d | per bidder: rows %7.1f kept %6.1f objects %8.1f exact %6.2f | ties %d unseeded %d"
          % (k, d[0], d[1] / bd, d[2] / bd, d[3] / bd, d[4] / bd, d[5], d[6]), flush=True)
L.genpc_emd_tune(-1, 0)
