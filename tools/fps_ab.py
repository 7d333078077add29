import os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from genpc_amd.fps import fps_sampling
from genpc_amd import _lib
x = torch.rand(24000, 3, device="cuda")
for tag, leg in (("tags", 0), ("legacy ds_read_b96", 1)):
    _lib.lib.genpc_fps_tune(leg)
    fps_sampling(x, 200); torch.cuda.synchronize()
    t0 = time.perf_counter(); fps_sampling(x, 20000); torch.cuda.synchronize()
    print(tag, "verify", os.environ.get("GENPC_FPS_VERIFY", "1"), "%.2f ms" % ((time.perf_counter() - t0) * 1e3))
