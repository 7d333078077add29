"""Golden outcomes of the alignment loop -- OWN fixtures: the loop is bit-reproducible
(tests/test_gpu_determinism.py), so its transforms and loss histories on committed inputs are committed too and the
`-m gpu` tests compare against them at 1e-6 instead of outcome bounds.

    python tests/golden/make_pose_golden.py [out.npz]      # on an MI355X; default gpurun_out/pose_loop_golden.npz

Nothing of the reference is involved: inputs are tests/golden/waymo_car59_4096.npz (make_waymo_c4.py), the bundled-scan
fixture scans13_fps16384.npz and the seeded C5 generator of tests/test_gpu_pipeline.py; outputs are this library's.
Regenerate (and say so in the commit) whenever a kernel of the loop changes its arithmetic or summation order.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization  # noqa: E402
from genpc_amd import reg_xyz  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pose_loop_golden.npz")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    res = {}
    w = np.load(os.path.join(HERE, "waymo_car59_4096.npz"))
    C = torch.from_numpy(w["complete"]).cuda()
    # (a) config 4, the test pair: complete car vs its pad-repeated crop under a known similarity
    T, h, bp = object_pose_optimization(C, torch.from_numpy(w["test_partial"]).cuda(), radius=0.02, lr=0.01, iters=200,
                                        render_size=224, return_history=True)
    res.update(c4_pair_T=T, c4_pair_hist=h, c4_pair_params=bp)
    # (b) config 4, real crops: the complete car against the first 8 Waymo crops, in lock-step, and reg_tensors on crop 0
    P8 = torch.from_numpy(w["crops"][:8]).cuda()
    T8, h8, _ = object_pose_optimization(C[None].expand(8, -1, -1).contiguous(), P8, radius=0.02, lr=0.01, iters=200,
                                         render_size=224, return_history=True)
    res.update(c4_crops8_T=T8, c4_crops8_hist=h8)
    r = reg_xyz.reg_tensors(P8[0], C, generative_model="trellis", dataset="redwood", cd_inv_weight=0.5, diff_init=True,
                            reg_fine_xyz=True)
    res.update(c4_reg0_diff=np.asarray(r["diff_transform"]), c4_reg0_coarse=np.asarray(r["coarse_transformation"]),
               c4_reg0_best_scale=np.float64(r["best_scale"]), c4_reg0_S=np.asarray(r["best_scales_transformation"]),
               c4_reg0_Txyz=np.asarray(r["best_transformation_xyz"]), c4_reg0_target=r["target"].cpu().numpy())
    # (c) config 5's per-rank shape: 8 scans x 32768 in lock-step
    from test_gpu_pipeline import c5_scan, c2_inputs
    scans = [c5_scan(s) for s in range(8)]
    C5 = torch.from_numpy(np.stack([x[0] for x in scans])).cuda()
    P5 = torch.from_numpy(np.stack([x[1] for x in scans])).cuda()
    T5, h5, _ = object_pose_optimization(C5, P5, radius=0.02, lr=0.01, iters=200, render_size=224, return_history=True)
    res.update(c5_T=T5, c5_hist=h5)
    # (d) config 2's registration stage on the bundled scan
    g = np.load(os.path.join(HERE, "scans13_fps16384.npz"))
    partial, gen, img, gt = c2_inputs(lambda name: g)
    r2 = reg_xyz.reg_tensors(torch.from_numpy(partial).cuda(), torch.from_numpy(gen).cuda(), generative_model="trellis",
                             dataset="redwood", cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True)
    res.update(c2_diff=np.asarray(r2["diff_transform"]), c2_coarse=np.asarray(r2["coarse_transformation"]),
               c2_best_scale=np.float64(r2["best_scale"]), c2_S=np.asarray(r2["best_scales_transformation"]),
               c2_Txyz=np.asarray(r2["best_transformation_xyz"]))
    np.savez_compressed(out, **res)
    for k, v in res.items():
        print(k, np.asarray(v).shape)
    print("written", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
