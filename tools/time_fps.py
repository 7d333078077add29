"""FPS step latency by cloud size.   python3 tools/time_fps.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from genpc_amd.fps import fps_sampling
for n, k in ((8192, 4096), (16384, 8192), (24000, 20000), (32768, 16384), (165546, 16384)):
    x = torch.rand(n, 3, device="cuda")
    fps_sampling(x, 64); torch.cuda.synchronize()
    t0 = time.perf_counter(); fps_sampling(x, k); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("n %6d k %5d: %7.2f ms, %.2f us per step" % (n, k, dt * 1e3, dt / k * 1e6))
