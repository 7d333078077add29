"""The one-workgroup pruned sampling (csrc/fps_grid.hip) against the multi-workgroup kernel and the oracle, with times and
rounds.   python3 tools/fps_grid_check.py [quick]"""
import ctypes, os, subprocess, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
if "--build" in sys.argv:
    # a private copy of the library whose sampling kernel stamps its phases (-DGENPC_FPS_TIMELINE): tools/_timeline/libgenpc_hip.so;
    # run with GENPC_LIB=$PWD/tools/_timeline/libgenpc_hip.so on the GPU box (the shipped kernel reads no clock: the phase columns are 0)
    from genpc_amd import build as B
    B.build(verbose=False)
    OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_timeline")
    os.makedirs(OUT, exist_ok=True)
    cflags = [f for f in B.FLAGS if f != "-shared"]
    obj = os.path.join(OUT, "fps_grid.o")
    subprocess.check_call([B.HIPCC] + cflags + ["-DGENPC_FPS_TIMELINE", "-c", os.path.join(B.CSRC, "fps_grid.hip"), "-o", obj])
    objs = [os.path.join(B.LIBDIR, "obj", os.path.basename(x)[:-4] + ".o") for x in B.sources() if not x.endswith("fps_grid.hip")]
    subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + [obj] + ["-o", os.path.join(OUT, "libgenpc_hip.so")])
    print(os.path.join(OUT, "libgenpc_hip.so"))
    sys.exit(0)
import numpy as np, torch
from genpc_amd import _lib
from genpc_amd.fps import fps_sampling
from oracle import oracle

L = _lib.lib
z = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "scans13_fps16384.npz"))
rng = np.random.default_rng(3)
cases = []
for s in (0, 5):
    surf = np.concatenate([z["partial"][s][:8192], z["gt"][s]]).astype(np.float32)
    cases.append(("scan%d partial8192+gt16384 -> 20000" % s, surf, 20000))
    cases.append(("scan%d gt16384 -> 16384" % s, z["gt"][s].astype(np.float32), 16384))
    cases.append(("scan%d 20000 -> 16384" % s, surf[:20000].copy(), 16384))
cases.append(("uniform volume 24000 -> 20000", (rng.random((24000, 3), dtype=np.float32) - 0.5), 20000))
cases.append(("uniform volume 24576 -> 4096", (rng.random((24576, 3), dtype=np.float32) - 0.5), 4096))
cases.append(("uniform volume 8192 -> 8192", (rng.random((8192, 3), dtype=np.float32) - 0.5), 8192))
cases.append(("lattice 12^3 x 20000 -> 3000", (rng.integers(0, 12, size=(20000, 3)) / 12.0 - 0.5).astype(np.float32), 3000))
cases.append(("far from origin 9000 -> 9000", (rng.random((9000, 3), dtype=np.float32) * 0.01 + 1000.0).astype(np.float32), 9000))
cases.append(("plane 5000 -> 5000", np.concatenate([rng.random((5000, 2), dtype=np.float32), np.zeros((5000, 1), np.float32)], 1), 5000))
cases.append(("all equal 300 -> 300", np.ones((300, 3), np.float32), 300))
cases.append(("tiny 5 -> 5", rng.random((5, 3), dtype=np.float32), 5))
quick = "quick" in sys.argv
for name, x, k in cases:
    X = torch.from_numpy(x).cuda()
    res = {}
    for tag, bits in (("grid", 0), ("multi-wg", 256)):
        prev = L.genpc_fps_tune(bits)
        try:
            fps_sampling(X, min(k, 8)); torch.cuda.synchronize()
            t0 = time.perf_counter()
            idx = fps_sampling(X, k)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            rounds = (ctypes.c_int * 60)()
            _lib.on_device_of(X, L.genpc_fps_stats, 60, ctypes.addressof(rounds))
            res[tag] = (idx.cpu().numpy(), dt, rounds[0], list(rounds[47:59]))
        finally:
            L.genpc_fps_tune(prev)
    same = np.array_equal(res["grid"][0], res["multi-wg"][0])
    ok_o = ""
    if not quick or x.shape[0] <= 9000:
        want = oracle.fps(x, k, 1)
        ok_o = " oracle %s" % ("ok" if np.array_equal(res["grid"][0], want) else "DIFFERENT (first at %d)" % int(np.argmax(res["grid"][0] != want)))
    print("%-40s grid %7.2f ms (%5d rounds, %.3f us/pick)  multi-wg %7.2f ms (%5d exch)  same %s%s   us: apply %d sweep %d L %d leave %d enum %d rank %d scatter %d; sweeps %d overflows %d items/round %d cands/round %d clock %d MHz"
          % ((name, res["grid"][1] * 1e3, res["grid"][2], res["grid"][1] / k * 1e6, res["multi-wg"][1] * 1e3, res["multi-wg"][2], same, ok_o) + tuple(res["grid"][3])), flush=True)
from genpc_amd import fps as F
print("fps.stats", F.stats)
