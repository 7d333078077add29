"""Wall time of the stages inside reg_xyz.reg_tensors at config 2's shape (8192-point partial, 16384-point generated shape).
python3 tools/time_reg_stages.py [scan]"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from genpc_amd import reg_xyz
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization

if len(sys.argv) > 1 and sys.argv[1] == "scan":
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "scans13_fps16384.npz"))
    part, A = torch.from_numpy(z["partial"][0, :8192].copy()).cuda(), torch.from_numpy(z["gt"][0].copy()).cuda()
else:
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    A = torch.rand(16384, 3, device="cuda", generator=gen) - 0.5
    part = ((torch.rand(8192, 3, device="cuda", generator=gen) - 0.5) * 0.9 + 0.01).contiguous()


def t(name, f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = f()
    torch.cuda.synchronize(); print("%-34s %7.2f ms" % (name, (time.perf_counter() - t0) / reps * 1e3)); return r


res = t("reg_tensors (all)", lambda: reg_xyz.reg_tensors(part, A, reg_fine_xyz=True))
tv = t("voxel_down_sample target 0.02", lambda: reg_xyz.voxel_down_sample(A, 0.02))
sv = t("voxel_down_sample source 0.02", lambda: reg_xyz.voxel_down_sample(part, 0.02))
print("   sizes", tuple(tv.shape), tuple(sv.shape))
T = t("object_pose_optimization", lambda: object_pose_optimization(tv, sv, radius=0.02, lr=0.01, iters=200, render_size=224))
D = np.linalg.inv(T.astype(np.float64))
src = t("_apply", lambda: reg_xyz._apply(D, part))
tgt = t("normalize_numpy", lambda: reg_xyz.normalize_numpy(A, range=0.5)[0])
s3, t3 = reg_xyz.voxel_down_sample(src, 0.03), reg_xyz.voxel_down_sample(tgt, 0.03)
bs, bl, coarse = t("coarse_scale_sweep", lambda: reg_xyz.coarse_scale_sweep(s3, t3, cd_inv_weight=0.5))
src2 = reg_xyz._apply(coarse, src)
s4 = reg_xyz.voxel_down_sample(src2, 0.03)
t("iterative_scale_search", lambda: reg_xyz.iterative_scale_search(s4, t3, [(0.8, 1.2)] * 3, 10, init_transform=np.eye(4), cd_inv_weight=0.5))
t("fuse (FPS 20000 + outlier filter)", lambda: reg_xyz.fuse(res["source"], res["target"], num_points=20000))
