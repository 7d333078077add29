"""Where a block of mask_splat_kernel spends its time in the alignment loop at config 2's post-voxel size (wall-clock stamps).

    python tools/splat_timeline.py --build     # here: compiles csrc/pose.hip with -DGENPC_SPLAT_TIMELINE into tools/_timeline/
    GENPC_LIB=$PWD/tools/_timeline/libgenpc_hip.so python tools/splat_timeline.py        # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_timeline")
if "--build" in sys.argv:
    from genpc_amd import build as B
    B.build(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    cflags = [f for f in B.FLAGS if f != "-shared"]
    obj = os.path.join(OUT, "pose.o")
    subprocess.check_call([B.HIPCC] + cflags + ["-DGENPC_SPLAT_TIMELINE", "-c", os.path.join(B.CSRC, "pose.hip"), "-o", obj])
    objs = [os.path.join(B.LIBDIR, "obj", os.path.basename(s)[:-4] + ".o") for s in B.sources() if not s.endswith("pose.hip")]
    subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + [obj] + ["-o", os.path.join(OUT, "libgenpc_hip.so")])
    print(os.path.join(OUT, "libgenpc_hip.so"))
    sys.exit(0)
import numpy as np, torch
from genpc_amd import _lib, reg_xyz
from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
z = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
part, A = torch.from_numpy(z["partial"][0, :8192].copy()).cuda(), torch.from_numpy(z["gt"][0].copy()).cuda()
tv, sv = reg_xyz.voxel_down_sample(A, 0.02), reg_xyz.voxel_down_sample(part, 0.02)
_lib.lib.genpc_pose_dual(0)
object_pose_optimization(tv, sv, radius=0.02, lr=0.01, iters=40, render_size=224)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(L, "genpc_splat_timeline_read"):
    sys.exit("not built with -DGENPC_SPLAT_TIMELINE")
buf = (ctypes.c_ulonglong * (4096 * 8))()
assert L.genpc_splat_timeline_read(buf)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
t = t[t[:, 0] > 0]
w0 = t[:, 0].min()
print("%d blocks of the last splat launch; launch span %.2f us (first start -> last end)" % (len(t), (t[:, 6].max() - w0) / 100.0))
cnt = t[:, 7] - 1
print("tiles with points: %d (list lengths: median %d, max %d); empty %d" % ((cnt > 0).sum(), np.median(cnt[cnt > 0]), cnt.max(), (cnt == 0).sum()))
names = ["start -> count known", "-> list filled (ranks, fetch)", "-> strips built", "-> gather done", "-> planes written", "-> tile sums added (end)"]
for sel, label in ((cnt > 0, "tiles with points"), (cnt == 0, "empty tiles")):
    tt = t[sel]
    print(label)
    prev = 0
    for k, nm in zip((1, 2, 3, 4, 5, 6), names):
        ok = (tt[:, k] > 0) & (tt[:, prev] > 0)
        if ok.any():
            v = (tt[ok, k] - tt[ok, prev]) / 100.0
            print("   %-34s median %6.2f us  max %6.2f" % (nm, np.median(v), v.max()))
            prev = k
    v = (tt[:, 6] - tt[:, 0]) / 100.0
    print("   %-34s median %6.2f us  max %6.2f;  start offsets: median %.2f max %.2f us" % ("whole block", np.median(v), v.max(), np.median(tt[:, 0] - w0) / 100.0, (tt[:, 0] - w0).max() / 100.0))
