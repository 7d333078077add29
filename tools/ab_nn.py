"""Step time of the chamfer forward at three shapes; run it twice in ONE gpurun call -- once with GENPC_LIB pointing at
a library built from another commit -- to compare kernels on the same box (boxes of the pool differ by +-15 %).
    GENPC_LIB=$PWD/tools/_lib_head.so python tools/ab_nn.py; python tools/ab_nn.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd import chamfer_3D
def fwd(X, Y):
    b, n, m = X.shape[0], X.shape[1], Y.shape[1]
    d1 = torch.empty(b, n, device=X.device); d2 = torch.empty(b, m, device=X.device)
    i1 = torch.empty(b, n, device=X.device, dtype=torch.int32); i2 = torch.empty(b, m, device=X.device, dtype=torch.int32)
    chamfer_3D.forward(X, Y, d1, d2, i1, i2)
    return d1, d2, i1, i2
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
out = []
for (b, n, m) in ((1, 16384, 16384), (13, 16384, 16384), (4, 16384, 8192)):
    X = torch.rand(b, n, 3, device="cuda", generator=gen) - 0.5
    Y = torch.rand(b, m, 3, device="cuda", generator=gen) - 0.5
    for _ in range(50): fwd(X, Y)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(400): fwd(X, Y)
    e1.record(); e1.synchronize()
    out.append("%dx%dx%d %.2f us" % (b, n, m, e0.elapsed_time(e1) / 400 * 1e3))
print(os.environ.get("GENPC_LIB", "new")[-14:], " | ".join(out))
