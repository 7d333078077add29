"""tools/opsel_probe.hip's kernel (a packed fp32 subtract with half selection on known data) beside the library's f16 nearest-
neighbour filter on two other streams -- the neighbour that makes the same form fail inside the farthest-point sampling.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/opsel_probe.hip -o gpurun_out/libopsel.so && python3 tools/opsel_beside_filter.py [seconds]"""
import ctypes, os, sys, threading, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import chamfer_3D
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
lib = ctypes.CDLL(os.path.join(ROOT, "gpurun_out", "libopsel.so"))
z = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
stop = threading.Event()


def filt(k):
    with torch.cuda.stream(torch.cuda.Stream()):
        P = torch.from_numpy(z["partial"][k].copy()).cuda()[None].contiguous(); G = torch.from_numpy(z["gt"][k].copy()).cuda()[None].contiguous()
        d1 = torch.empty(1, 16384, device="cuda"); d2 = torch.empty_like(d1)
        i1 = torch.empty(1, 16384, device="cuda", dtype=torch.int32); i2 = torch.empty_like(i1)
        while not stop.is_set():
            for _ in range(50):
                chamfer_3D.forward(P, G, d1, d2, i1, i2)
            torch.cuda.current_stream().synchronize()


burnlib = ctypes.CDLL(os.path.join(ROOT, "gpurun_out", "libburn.so")) if os.path.exists(os.path.join(ROOT, "gpurun_out", "libburn.so")) else None
scratch = torch.zeros(16, device="cuda")


def burner(bk):
    def run(_):
        with torch.cuda.stream(torch.cuda.Stream()):
            while not stop.is_set():
                for _ in range(20):
                    burnlib.burn(bk, 400, 256, ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                torch.cuda.current_stream().synchronize()
    return run


cases = [(0, "op_sel:[0,1]", 64, "filter"), (0, "op_sel:[0,1]", 256, "filter"), (1, "op_sel_hi:[1,0]", 256, "filter"), (2, "plain pair", 256, "filter")]
if burnlib is not None:
    cases += [(0, "op_sel:[0,1]", 256, "burn %d" % bk) for bk in (1, 2, 4, 8, 3, 5, 7, 15)]
for kind, name, blocks, nb in cases:
    if True:
        stop.clear()
        th = [threading.Thread(target=(filt if nb == "filter" else burner(int(nb.split()[1]))), args=(k,)) for k in (2, 3)]
        for t in th: t.start()
        nhit = torch.zeros(2, device="cuda", dtype=torch.int32); hits = torch.zeros(4096 * 4, device="cuda", dtype=torch.int32)
        done = torch.zeros(1, device="cuda", dtype=torch.int64)
        st = torch.cuda.Stream()
        t0 = time.time()
        while time.time() - t0 < seconds:
            assert lib.opsel_launch(kind, blocks, 0, ctypes.c_void_p(nhit.data_ptr()), ctypes.c_void_p(hits.data_ptr()), ctypes.c_void_p(done.data_ptr()), ctypes.c_void_p(st.cuda_stream))
            st.synchronize()
        stop.set()
        for t in th: t.join()
        n = int(nhit[0].item()) & 0xffffffff
        print("v_pk_add_f32 %-16s %3d blocks beside two streams of %-8s: %.3g wave-iterations, %d wrong results" % (name, blocks, nb, float(done.item()) * 4.0, n), flush=True)
        if n:
            h = hits.cpu().numpy().view(np.uint32).reshape(4096, 4)[:min(n, 4096)]
            lane = h[:, 1] & 63
            print("   first %d records: lanes 0-15 %d, 16-31 %d, 32-47 %d, 48-63 %d; low result wrong %d, high result wrong %d" % (
                len(h), int((lane < 16).sum()), int(((lane >= 16) & (lane < 32)).sum()), int(((lane >= 32) & (lane < 48)).sum()), int((lane >= 48).sum()),
                int(((h[:, 1] >> 6) & 1).sum()), int(((h[:, 1] >> 7) & 1).sum())))
            for r in h[:4]:
                it, ln = int(r[0]), int(r[1] & 63)
                A, B = (it & 1023) + 0.25, (it & 1023) + 4096.5 + ln
                print("   iteration %d lane %d: got (%.2f, %.2f); right (%.2f, %.2f); with the LOW half instead (%.2f, ...)" % (
                    it, ln, float(r[2:3].view(np.float32)[0]), float(r[3:4].view(np.float32)[0]), 1.0 + ln - B, 2.0 + ln - B, 1.0 + ln - A))
