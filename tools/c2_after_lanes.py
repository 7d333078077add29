import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
import numpy as np, torch
from genpc_amd import pipeline, _lib
from genpc_amd.DepthPrompting import DepthPrompting
exec(open(os.path.join(ROOT, "tools", "time_c2_streams.py")).read().split("for dual in")[0])
print("fresh process: %.1f scans/s" % run(), flush=True)
# what bench.py did before this line: lanes of registrations and other streams
pipeline.run_in_lanes(lambda li, _: pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=DepthPrompting(cfg)), range(6), 6, torch.device("cuda"))
torch.cuda.synchronize()
print("after six lanes ran: %.1f scans/s" % run(), flush=True)
print("again: %.1f scans/s" % run(), flush=True)
os.environ["X"] = "1"
for d in (0, 1):
    pipeline._FPS_DEFER = bool(d)
    print("defer %d: %.1f scans/s" % (d, run()), flush=True)
# a minute of the chip running flat out (what bench.py has behind it by then), then the same line
from genpc_amd.loss_functions.Chamfer3D.dist_chamfer_3D import chamfer_3DDist
X = torch.rand(13, 16384, 3, device="cuda"); Y = torch.rand(13, 16384, 3, device="cuda")
cd = chamfer_3DDist()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 25:
    for _ in range(200): cd(X, Y)
    torch.cuda.synchronize()
print("after 25 s of batched Chamfer: %.1f scans/s" % run(), flush=True)
print("again: %.1f scans/s" % run(), flush=True)
time.sleep(5)
print("after 5 s of rest: %.1f scans/s" % run(), flush=True)
