"""EMD forward on fixed buffers (the replay cache's case): ms per call.   python3 tools/time_emd_graph.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import emd as E
from genpc_amd.loss_functions.emd.emd_module import alloc_state
for b, n in ((1, 2048), (1, 16384), (13, 16384), (64, 2048)):
    rng = np.random.default_rng(7)
    X = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    Y = torch.from_numpy(rng.random((b, n, 3), dtype=np.float32)).cuda()
    s = alloc_state(b, n, n, X.device)
    init = {k: v.clone() for k, v in s.items()}
    def call():
        for k in ("assignment", "assignment_inv", "price"):
            s[k].copy_(init[k])
        assert E.forward(X, Y, s["dist"], s["assignment"], s["price"], s["assignment_inv"], s["bid"], s["bid_increments"],
                         s["max_increments"], s["unass_idx"], s["unass_cnt"], s["unass_cnt_sum"], s["cnt_tmp"], s["max_idx"], 0.005, 50) == 1
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); e1.synchronize()
    print("  %dx%d: %.3f ms  (emd %.6f)" % (b, n, e0.elapsed_time(e1) / 20, float(torch.sqrt(s["dist"]).mean())))
