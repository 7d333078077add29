"""Where hpr_kernel spends its time (shader-clock ticks per phase, trip counts of the clip loop).

    python tools/hpr_phases.py --build      # here: csrc/hpr.hip with -DGENPC_HPR_PROF into tools/_hprprof/libgenpc_hip.so
    GENPC_LIB=$PWD/tools/_hprprof/libgenpc_hip.so python tools/hpr_phases.py          # on the GPU box
"""
import ctypes, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_hprprof")
if "--build" in sys.argv:
    from genpc_amd import build as B
    B.build(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    cflags = [f for f in B.FLAGS if f != "-shared"]
    obj = os.path.join(OUT, "hpr.o")
    subprocess.check_call([B.HIPCC] + cflags + ["-DGENPC_HPR_PROF", "-c", os.path.join(B.CSRC, "hpr.hip"), "-o", obj])
    objs = [os.path.join(B.LIBDIR, "obj", os.path.basename(s)[:-4] + ".o") for s in B.sources() if not s.endswith("/hpr.hip")]
    subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + [obj, "-o", os.path.join(OUT, "libgenpc_hip.so")])
    print(os.path.join(OUT, "libgenpc_hip.so"))
    sys.exit(0)
import numpy as np, torch
from types import SimpleNamespace
from genpc_amd import _lib
from genpc_amd.DepthPrompting import DepthPrompting
from genpc_amd.fps import fps_sampling
cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                      mask_pixel_rate=3, view_num=1024, distance=1.6, downsample_num=10000, removal_radius=10000)
dp = DepthPrompting(cfg)
L = ctypes.CDLL(_lib.LIB_PATH)
rng = np.random.default_rng(5)
v = rng.normal(size=(165546, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
blob = torch.from_numpy((v * (0.3 + 0.2 * np.abs(np.sin(3 * v[:, :1])))).astype(np.float32)).cuda()
g = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
buf = (ctypes.c_ulonglong * 16)()
for name, pts in (("blob", blob), ("scan partial 0", torch.from_numpy(g["partial"][0]).cuda())):
    sub = pts[fps_sampling(pts, 10000).long()].contiguous()
    dp.hidden_point_removal(sub, dp.viewpoints, 10000.0); torch.cuda.synchronize()
    L.genpc_hpr_prof_read(buf, 1)
    dp.hidden_point_removal(sub, dp.viewpoints, 10000.0); torch.cuda.synchronize()
    L.genpc_hpr_prof_read(buf, 1)
    t = np.array(list(buf), dtype=np.float64)
    w = t[8]
    print("%s: %d waves; ticks per wave: home tiles %.0f, verify %.0f, park %.0f; per wave: marking rounds x lanes %.1f, marked candidates "
          "%.1f (per lane-round %.1f), clip-loop trips %.1f (max over lanes), %.1f summed over lanes"
          % (name, w, t[0] / w, t[1] / w, t[2] / w, t[4] / w, t[7] / w, t[7] / max(t[4], 1), t[5] / w, t[6] / w))
    print("   wave-per-point walk: %d walkers, %.0f ticks each on average, slowest %.0f, %d over 200 k ticks (%d of them end visible); "
          "open polygons (a vertex beyond 1000): %d walkers, %.0f ticks each" % (t[10], t[11] / max(t[10], 1), t[12], t[13], t[9], t[14], t[15] / max(t[14], 1)))
