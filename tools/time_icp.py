"""registration_icp at the pipeline's sizes: one candidate and the 11 of the coarse sweep; GENPC_ICP_FUSED=0/1.   python3 tools/time_icp.py"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import reg_xyz
z13 = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
gt = torch.from_numpy(z13["gt"][0].copy()).cuda()
pa = torch.from_numpy(z13["partial"][0].copy()).cuda()
gt_n = reg_xyz.normalize_numpy(gt)[0]
for vs in (0.06, 0.04):
    src = reg_xyz.voxel_down_sample(reg_xyz.normalize_numpy(pa)[0] * 0.9, vs).contiguous()
    tgt = reg_xyz.voxel_down_sample(gt_n, vs).contiguous()
    inits = []
    for sc in np.linspace(1.5, 0.8, 11):
        S = np.eye(4); S[:3, :3] *= sc; inits.append(S)
    for name, init in (("k=1", np.eye(4)), ("k=11", np.stack(inits))):
        out = reg_xyz.registration_icp(src, tgt, 0.075, init); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): out = reg_xyz.registration_icp(src, tgt, 0.075, init)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        T = np.asarray(out[0]).reshape(-1, 4, 4)
        print("voxel %.2f: %d x %d %s: %.3f ms per solve; fitness %s iters %s T[0,:3,3] %s" % (vs, src.shape[0], tgt.shape[0], name, dt * 1e3, np.round(np.atleast_1d(out[1])[:3], 6), np.atleast_1d(out[3])[:3], np.round(T[0, :3, 3], 9)), flush=True)
