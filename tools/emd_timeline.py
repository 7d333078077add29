"""Phase timeline of the one-launch EMD (GENPC_EMD_TIMELINE=1): per round, workgroup 0's time (us) in: the look at
assignment / count, the bid, barrier 1, settle, barrier 2.
    GENPC_EMD_TIMELINE=1 python3 tools/emd_timeline.py [b n | scan]"""
import os, sys, ctypes
os.environ["GENPC_EMD_TIMELINE"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import _lib
from genpc_amd.loss_functions import emdModule
L = _lib.lib; em = emdModule()
L.genpc_emd_tune(2, -1)
args = sys.argv[1:]
if args and args[0] == "scan":
    z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
    k = int(args[1]) if len(args) > 1 else 1
    X, Y = torch.from_numpy(z["partial"][:k].copy()).cuda(), torch.from_numpy(z["gt"][:k].copy()).cuda()
else:
    b, n = (int(args[0]), int(args[1])) if len(args) >= 2 else (1, 16384)
    rng = np.random.default_rng(7)
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
for _ in range(3): em(X, Y, 0.005, 50)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 1024)()
f = ctypes.CDLL(_lib.LIB_PATH if hasattr(_lib, "LIB_PATH") else os.path.join(os.path.dirname(_lib.__file__), "lib", "libgenpc_hip.so")).genpc_debug_emd_timeline
f.argtypes = [ctypes.c_void_p]
assert f(buf) == 1
t = np.array(buf[:], dtype=np.int64).reshape(64, 16)
print("round      U  nbid(w0) |  look    bid   bar1  settle  bar2  | round total (us) | first pass: seeds proxy sweep merge publish")
tot = np.zeros(5)
for r in range(50):
    s = t[r]
    if s[0] == 0: break
    d = [(s[1] - s[0]) / 100., (s[2] - s[1]) / 100., (s[3] - s[2]) / 100., (s[4] - s[3]) / 100., (s[5] - s[4]) / 100. if s[5] else 0.0]
    tot += d
    f = [(s[8] - s[1]) / 100., (s[9] - s[8]) / 100., (s[10] - s[9]) / 100., (s[11] - s[10]) / 100., (s[12] - s[11]) / 100.] if s[8] else [0] * 5
    print("%5d %6d %6d    | %5.2f %6.2f %6.2f %6.2f %6.2f | %6.2f | %5.2f %5.2f %5.2f %5.2f %5.2f" % (r, s[6], s[7], *d, sum(d), *f))
print("sum                     | %5.1f %6.1f %6.1f %6.1f %6.1f | %6.1f" % (*tot, tot.sum()))
