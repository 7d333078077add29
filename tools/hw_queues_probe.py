"""How many kernels of different streams run at once: n streams, one long single-block spin kernel each (torch.cuda._sleep).
   GPU_MAX_HW_QUEUES=8 python3 tools/hw_queues_probe.py [n_streams] [priority_mix]"""
import sys, time, torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
mix = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.init()
streams = [torch.cuda.Stream(priority=(-1 if (mix and i % 2) else 0)) for i in range(n)]
cyc = 20_000_000      # ~10 ms
torch.cuda._sleep(cyc); torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(cyc); torch.cuda.synchronize(); one = time.perf_counter() - t0
for s in streams:
    with torch.cuda.stream(s): torch.cuda._sleep(1000)
torch.cuda.synchronize()
t0 = time.perf_counter()
for s in streams:
    with torch.cuda.stream(s): torch.cuda._sleep(cyc)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("%d streams (mix %d): one %.2f ms, all %.2f ms -> %.1f at a time" % (n, mix, one * 1e3, tot * 1e3, n * one / tot))
