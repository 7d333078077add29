"""Where a block of the NN filter kernel spends its time (shader-clock stamps at the phase boundaries).

    python tools/nn_timeline.py --build            # here: compiles csrc/nn_f16.hip with -DGENPC_NN_TIMELINE into
                                                   #       tools/_timeline/libgenpc_hip.so (other objects are the shipped ones)
    GENPC_LIB=$PWD/tools/_timeline/libgenpc_hip.so python tools/nn_timeline.py [B N M]     # on the GPU box
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_timeline")

if "--build" in sys.argv:
    from genpc_amd import build as B
    B.build(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    cflags = [f for f in B.FLAGS if f != "-shared"]
    mine = []
    for name in ("nn_f16", "nn_finish"):
        obj = os.path.join(OUT, name + ".o")
        subprocess.check_call([B.HIPCC] + cflags + ["-DGENPC_NN_TIMELINE", "-c", os.path.join(B.CSRC, name + ".hip"), "-o", obj])
        mine.append(obj)
    objs = [os.path.join(B.LIBDIR, "obj", os.path.basename(s)[:-4] + ".o") for s in B.sources()
            if not s.endswith("nn_f16.hip") and not s.endswith("nn_finish.hip")]
    subprocess.check_call([B.HIPCC, "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + mine + ["-o", os.path.join(OUT, "libgenpc_hip.so")])
    print(os.path.join(OUT, "libgenpc_hip.so"))
    sys.exit(0)

import numpy as np
import torch
from genpc_amd import _lib, chamfer_3D

b, n, m = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1, 16384, 16384)
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
X = torch.rand(b, n, 3, device="cuda", generator=gen) - 0.5
Y = torch.rand(b, m, 3, device="cuda", generator=gen) - 0.5
d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, m, device="cuda")
i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, m, device="cuda", dtype=torch.int32)
for _ in range(20):
    chamfer_3D.forward(X, Y, d1, d2, i1, i2)
torch.cuda.synchronize()
L = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(L, "genpc_nn_timeline_read"):
    sys.exit("this library was not built with -DGENPC_NN_TIMELINE (python tools/nn_timeline.py --build, then GENPC_LIB=...)")
buf = (ctypes.c_ulonglong * (4096 * 8))()
assert L.genpc_nn_timeline_read(buf)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
if os.environ.get("NN_TIMELINE_BLOCKS"):
    # per XCD (block id % 8): start and end of every block relative to the XCD's first start, in block order
    # slots 6, 7: the 100 MHz wall clock (one for the chip) at a block's start and end, in 10 ns units from the first start
    ids = np.nonzero(t[:, 0] > 0)[0]
    w0 = t[ids, 6].min()
    print("wall clock: last end %d (x 10 ns)" % (t[ids, 7].max() - w0))
    for x in range(int(os.environ["NN_TIMELINE_BLOCKS"])):
        sel = ids[ids % 8 == x]
        print("XCD %d: block:start-end (x 10 ns)" % x)
        print(" ".join("%d:%d-%d" % (i, t[i, 6] - w0, t[i, 7] - w0) for i in sel))
t = t[t[:, 0] > 0]
print("%d blocks stamped (shader-clock ticks; per-block differences only -- the counter is per XCD)" % len(t))
names = {1: "scale known (queries + slice read, maximum reduced)", 2: "first LDS tile staged", 3: "second LDS tile staged",
         4: "MFMA loop done", 5: "lists published"}
prev = 0
for k in (1, 2, 3, 4, 5):
    ok = t[:, k] > 0
    if not ok.any():
        continue
    v = (t[:, k] - t[:, prev])[ok]
    print("-> %-52s min %7d median %7d max %7d" % (names[k], v.min(), np.median(v), v.max()))
    prev = k
v = t[:, 5] - t[:, 0]
print("   %-52s min %7d median %7d max %7d" % ("whole block", v.min(), np.median(v), v.max()))
assert L.genpc_nn_timeline_read_finish(buf)
t = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
t = t[t[:, 0] > 0]
print("finish kernel, %d blocks stamped" % len(t))
names = {1: "lists, query, |t|max loaded", 2: "best unit evaluated, threshold", 3: "further candidates listed", 4: "further candidates evaluated", 5: "results written"}
prev = 0
for k in (1, 2, 3, 4, 5):
    v = t[:, k] - t[:, prev]
    print("-> %-52s min %7d median %7d max %7d" % (names[k], v.min(), np.median(v), v.max()))
    prev = k
v = t[:, 5] - t[:, 0]
print("   %-52s min %7d median %7d max %7d" % ("whole block", v.min(), np.median(v), v.max()))
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    chamfer_3D.forward(X, Y, d1, d2, i1, i2)
e1.record(); e1.synchronize()
print("step %.2f us (instrumented build)" % (e0.elapsed_time(e1) / 200 * 1e3))
