"""Is the alignment loop bound by the host's launch rate?  Time until the call RETURNS (everything enqueued) against time
until the GPU is done.   python3 tools/time_reg_host.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
from genpc_amd import reg_xyz
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
for b, nc, np_ in ((1, 16384, 8192), (8, 16384, 8192), (1, 4096, 4096)):
    C = torch.rand(b, nc, 3, device="cuda", generator=gen) - 0.5
    P = (C[:, :np_] * 0.9).contiguous()
    if b == 1:
        C = reg_xyz.voxel_down_sample(C[0], 0.02)[None].contiguous(); P = reg_xyz.voxel_down_sample(P[0], 0.02)[None].contiguous()
    from genpc_amd import _lib
    import ctypes
    L = _lib.lib
    for kw in ({}, {"cd_only": True}):
        object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=20, render_size=224, **kw); torch.cuda.synchronize()
        # raw C call timing: replicate the wrapper's call
        T = torch.empty(b, 16, device="cuda"); hist = torch.empty(b, 4 * 201, device="cuda"); bp = torch.empty(b, 10, device="cuda")
        col = torch.ones_like(C); pcol = torch.ones_like(P)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = L.genpc_pose_optimize_batch(b, C.shape[1], C.data_ptr(), None if kw else col.data_ptr(), P.shape[1], P.data_ptr(), None if kw else pcol.data_ptr(),
                                         0.01, 200, 4, 0.02, 224, 0.0 if kw else 1.0, T.data_ptr(), hist.data_ptr(), bp.data_ptr(), None)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("b %d %5d x %5d %s: enqueue %.1f ms, done %.1f ms (rc %d)" % (b, C.shape[1], P.shape[1], "cd_only" if kw else "full   ", (t1 - t0) * 1e3, (t2 - t0) * 1e3, rc), flush=True)
