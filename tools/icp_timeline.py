"""Where the one-workgroup ICP solve spends its time (wall-clock stamps of thread 0, candidate 0).

    python tools/icp_timeline.py --build     # here: compiles csrc/icp.hip with -DGENPC_ICP_TIMELINE into tools/_timeline_icp/
    GENPC_LIB=$PWD/tools/_timeline_icp/libgenpc_hip.so python tools/icp_timeline.py        # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_timeline_icp")
if "--build" in sys.argv:
    from genpc_amd import build as B
    B.build(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    cflags = [f for f in B.FLAGS if f != "-shared"]
    obj = os.path.join(OUT, "icp.o")
    subprocess.check_call([B.HIPCC] + cflags + ["-DGENPC_ICP_TIMELINE", "-c", os.path.join(B.CSRC, "icp.hip"), "-o", obj])
    objs = [os.path.join(B.LIBDIR, "obj", os.path.basename(s)[:-4] + ".o") for s in B.sources() if not s.endswith("icp.hip")]
    subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + [obj] + ["-o", os.path.join(OUT, "libgenpc_hip.so")])
    print(os.path.join(OUT, "libgenpc_hip.so"))
    sys.exit(0)
import numpy as np, torch
from genpc_amd import _lib, reg_xyz
z13 = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
gt = torch.from_numpy(z13["gt"][0].copy()).cuda()
pa = torch.from_numpy(z13["partial"][0].copy()).cuda()
L = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(L, "genpc_icp_timeline_read"):
    sys.exit("not built with -DGENPC_ICP_TIMELINE")
for vs in (0.06, 0.04):
    src = reg_xyz.voxel_down_sample(reg_xyz.normalize_numpy(pa)[0] * 0.9, vs).contiguous()
    tgt = reg_xyz.voxel_down_sample(reg_xyz.normalize_numpy(gt)[0], vs).contiguous()
    for _ in range(2): out = reg_xyz.registration_icp(src, tgt, 0.075)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (2 + 3 * 64))()
    assert L.genpc_icp_timeline_read(buf)
    t = np.frombuffer(buf, dtype=np.uint64).astype(np.int64)
    n = out[3] + 1
    p = t[2:2 + 3 * n].reshape(n, 3)
    print("%d x %d, %d passes: grid %.1f us; per pass search %s us, reduction %s us, update %s us" % (
        src.shape[0], tgt.shape[0], n, (t[1] - t[0]) / 100.0,
        np.round(np.diff(np.concatenate([[t[1]], p[:, 2]]))[:0] if False else (p[:, 0] - np.concatenate([[t[1]], p[:-1, 2]])) / 100.0, 1)[[0, 1, n // 2, n - 1]],
        np.round((p[:, 1] - p[:, 0]) / 100.0, 1)[[0, 1, n // 2, n - 1]], np.round((p[:, 2] - p[:, 1]) / 100.0, 1)[[0, 1, n // 2, n - 1]]))
    print("   totals: search %.1f, reduction %.1f, update %.1f us; whole kernel %.1f us" % (
        (p[:, 0] - np.concatenate([[t[1]], p[:-1, 2]])).sum() / 100.0, (p[:, 1] - p[:, 0]).sum() / 100.0, (p[:, 2] - p[:, 1]).sum() / 100.0, (p[-1, 2] - t[0]) / 100.0))
