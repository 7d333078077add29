"""What the filtered NN path does on alignment-loop shaped input (4 starts x 16384 posed points against 8192):
queries answered, queries re-done exhaustively, exact pieces evaluated.   python3 tools/nn_stats_pose.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd import _lib
from genpc_amd import chamfer_3D


def chamfer_forward(X, Y):
    b, n, m = X.shape[0], X.shape[1], Y.shape[1]
    d1 = torch.empty(b, n, device=X.device); d2 = torch.empty(b, m, device=X.device)
    i1 = torch.empty(b, n, device=X.device, dtype=torch.int32); i2 = torch.empty(b, m, device=X.device, dtype=torch.int32)
    chamfer_3D.forward(X, Y, d1, d2, i1, i2)
    return d1, d2, i1, i2


L = _lib.lib
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
for name, mk in (("cube volume", lambda: torch.rand(16384, 3, device="cuda", generator=gen) - 0.5),
                 ("sphere surface", lambda: torch.nn.functional.normalize(torch.randn(16384, 3, device="cuda", generator=gen), dim=1) * 0.5)):
    C = mk()
    P = (C[:8192] * 0.9).contiguous()
    scales = torch.tensor([0.9, 0.92, 0.95, 1.0], device="cuda").view(4, 1, 1)
    X = (C.unsqueeze(0) * scales).contiguous()
    Y = P.unsqueeze(0).repeat(4, 1, 1).contiguous()
    L.genpc_nn_tune(-1, 512)
    out = (ctypes.c_ulonglong * 3)()
    L.genpc_nn_stats(out, 1, None)
    d1, d2, i1, i2 = chamfer_forward(X, Y)
    L.genpc_nn_stats(out, 1, None)
    L.genpc_nn_tune(-1, 0)
    print("%-16s queries %d, exhaustive %d (%.3f %%), exact pieces %d (%.2f per query)" % (name, out[0], out[1], 100.0 * out[1] / out[0], out[2], out[2] / out[0]))
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for _ in range(20): chamfer_forward(X, Y)
    e0.record()
    for _ in range(200): chamfer_forward(X, Y)
    e1.record(); e1.synchronize()
    print("   %.1f us per forward" % (e0.elapsed_time(e1) / 200 * 1e3))
