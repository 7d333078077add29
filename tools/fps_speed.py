"""Farthest-point sampling times at the shapes the pipeline and the bench use.   python3 tools/fps_speed.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.fps import fps_sampling, fps_sampling_multi
g = torch.Generator(device="cuda"); g.manual_seed(2)
for c, n, k in ((4, 165546, 16384), (1, 165546, 16384), (1, 24576, 20000), (1, 20000, 16384), (1, 16384, 16384), (2, 8192, 4096)):
    X = torch.rand(c, n, 3, device="cuda", generator=g)
    fps_sampling(X, k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fps_sampling(X, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("%d x %6d -> %5d: %7.2f ms  (%.3f us per pick)" % (c, n, k, ms, ms * 1e3 / k))
