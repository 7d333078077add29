"""What the sequences look like that the device-side check of the farthest-point sampling rejects with several scans in flight:
the rejected sequence is kept and compared with the one drawn again.   python3 tools/fps_reject_probe.py [lanes] [scans]"""
import os, sys, time, threading
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import pipeline, fps as F
from genpc_amd.DepthPrompting import DepthPrompting

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 6
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
hook = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # genpc_fps_tune bits: the lanes inherit the caller's setting
seen, lock = [], threading.Lock()
orig = F._fps_multi_direct


def probe(clouds, ks, _attempt=0):
    # the library's own retry lives inside `orig`: run it with the retry disabled to get at the rejected sequence
    import ctypes
    pts = [c.contiguous().float() for c in clouds]
    c = len(pts)
    if c == 0:
        return []
    outs = [torch.empty(int(k), device=pts[0].device, dtype=torch.int32) for k in ks]
    n_arr = (ctypes.c_int * c)(*[int(p.shape[0]) for p in pts]); k_arr = (ctypes.c_int * c)(*[int(k) for k in ks])
    x_arr = (ctypes.c_void_p * c)(*[p.data_ptr() for p in pts]); o_arr = (ctypes.c_void_p * c)(*[o.data_ptr() for o in outs])
    rc = F._lib.on_device_of(pts[0], F._L.genpc_fps_multi, c, ctypes.addressof(n_arr), ctypes.addressof(k_arr), ctypes.addressof(x_arr), ctypes.addressof(o_arr))
    assert rc == 1
    first = torch.stack([o[0] for o in outs]).tolist()
    for j, f in enumerate(first):
        if f != 0:
            bad = outs[j].cpu().numpy().copy()
            good = orig([pts[j]], [ks[j]])[0]
            g = good.cpu().numpy()
            with lock:
                seen.append((int(f), int(pts[j].shape[0]), int(ks[j]), bad, g))
            outs[j] = good
    return outs


F._fps_multi_direct = probe
z13 = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
cfg = pipeline.default_cfg("cuda", view_num=1024)
g = torch.Generator(device="cuda"); g.manual_seed(1)
img = torch.rand(3, 1024, 1024, device="cuda", generator=g)
jobs = []
for k in range(6):
    gt = z13["gt"][k]
    cc = (gt.max(0) + gt.min(0)) / 2
    gen_np = ((gt - cc) / (gt.max(0) - gt.min(0)).max()).astype(np.float32)
    jobs.append((torch.from_numpy(z13["partial"][k][:8192].copy()).cuda(), torch.from_numpy(gen_np).cuda(), img, torch.from_numpy(gt.copy()).cuda()))
dps = [DepthPrompting(cfg) for _ in range(lanes)]
F._lib.lib.genpc_fps_tune(hook)
pipeline.complete_scans([jobs[i % 6] for i in range(count)], lanes=lanes, cfg=cfg, dps=dps)
torch.cuda.synchronize()
print("hook %d: %d scans, %d lanes: %d sequences rejected" % (hook, count, lanes, len(seen)))
same = 0
for f, n, k, bad, good in seen:
    bad = bad.copy(); bad[0] = 0
    d = np.nonzero(bad != good)[0]
    if d.size == 0:
        same += 1
        continue
    j = int(d[0])
    # where the rejected sequence leaves the right one: an extra sample (the rest shifted by one), a missing one, or something else
    shifted_later = bool(d.size and np.array_equal(bad[j + 1:], good[j:-1]))
    shifted_earlier = bool(d.size and np.array_equal(bad[j:-1], good[j + 1:]))
    W = min(64, max(1, (n + 191) // 192))
    i = int(bad[j])
    wg, rest = i % W, i // W
    thread, reg = rest % 192, rest // 192
    print("  code %d n %d k %d: first difference at step %d of %d (%d steps differ); rejected sample %d (workgroup %d of %d, thread %d = wave %d lane %d, register %d), right sample %d; "
          "rest of the rejected sequence = the right one shifted by one: %s; the rejected sample appears in the right sequence at step %s"
          % (f, n, k, j, k, d.size, i, wg, W, thread, thread // 64, thread % 64, reg, int(good[j]), "later" if shifted_later else ("earlier" if shifted_earlier else "no"),
             (np.nonzero(good == i)[0].tolist() or ["never"])[0]))
print("%d of the rejected sequences EQUAL the ones drawn again" % same)
