"""Every variant of the sampling's update (genpc_fps_tune bits) ALONE on the GPU: all give the shipped sequence.   python3 tools/fps_hook_alone.py"""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch, numpy as np
from genpc_amd import fps as F, _lib
g = torch.Generator(device="cuda"); g.manual_seed(3)
X = torch.rand(20806, 3, device="cuda", generator=g)
ref = F.fps_sampling(X, 20000).cpu().numpy()
for h in (0, 2, 34, 98, 162, 3):
    _lib.lib.genpc_fps_tune(h)
    before = dict(F.stats)
    try:
        out = F.fps_sampling(X, 20000).cpu().numpy()
        same = bool((out == ref).all())
    except Exception as e:
        same = "error: %s" % str(e)[:60]
    print("hook %3d alone: same as shipped %s, failed checks %d" % (h, same, F.stats["failed_check"] - before["failed_check"]))
_lib.lib.genpc_fps_tune(0)
