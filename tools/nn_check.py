"""Bit-exactness check of genpc_chamfer_forward against the oracle for given shapes.
python tools/nn_check.py BxNxM ..."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import chamfer_3D
from oracle import oracle
for spec in sys.argv[1:]:
    b, n, m = [int(x) for x in spec.split("x")]
    rng = np.random.default_rng(5)
    A = (rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5))
    B = (rng.random((b, m, 3), dtype=np.float32) - np.float32(0.5))
    exp = oracle.chamfer_forward(A, B, 1)
    a, bb = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
    d1 = torch.zeros(b, n, device="cuda"); d2 = torch.zeros(b, m, device="cuda")
    i1 = torch.zeros(b, n, device="cuda", dtype=torch.int32); i2 = torch.zeros(b, m, device="cuda", dtype=torch.int32)
    chamfer_3D.forward(a, bb, d1, d2, i1, i2)
    got = [d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()]
    bad = [int((g != e).sum()) for g, e in zip(got, exp)]
    msg = ""
    if bad[0]:
        w = np.argwhere(got[0] != exp[0])
        msg = " first bad dist1 at %s (of n=%d); bad query ids mod 128: %s" % (w[0], n, sorted(set((w[:, 1] % 128).tolist()))[:20])
    print(spec, "mismatches d1,d2,i1,i2:", bad, msg)
