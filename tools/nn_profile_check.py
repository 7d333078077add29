import sys; sys.path.insert(0, "/root/repo")
import torch
from genpc_amd import _lib, chamfer_3D
g = torch.Generator(device="cuda"); g.manual_seed(1)
A = torch.rand(1, 16384, 3, device="cuda", generator=g); B = torch.rand(1, 16384, 3, device="cuda", generator=g)
o = [torch.empty(1, 16384, device="cuda"), torch.empty(1, 16384, device="cuda"), torch.empty(1, 16384, device="cuda", dtype=torch.int32), torch.empty(1, 16384, device="cuda", dtype=torch.int32)]
for _ in range(5): chamfer_3D.forward(A, B, *o)
_lib.lib.genpc_nn_profile.restype = __import__("ctypes").c_float
_lib.lib.genpc_nn_profile(1)
k = []
for _ in range(50):
    chamfer_3D.forward(A, B, *o)
    k.append(float(_lib.lib.genpc_nn_profile(1)))
_lib.lib.genpc_nn_profile(0)
k = [x for x in k if x > 0]
print("filter kernel by dispatch events: %.2f us (min %.2f max %.2f, %d samples)" % (1e3 * sum(k) / len(k), 1e3 * min(k), 1e3 * max(k), len(k)))
