"""rocprofv3 driver: a few EMD forwards at one shape.  python3 tools/prof_emd.py B N [reps]"""
import os
import sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from genpc_amd.loss_functions import emdModule

b, n = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rng = np.random.default_rng(7)
X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
em = emdModule()
for _ in range(reps):
    d, a = em(X, Y, 0.005, 50)
torch.cuda.synchronize()
print("done", int(a.long().sum()))
