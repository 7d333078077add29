"""What a per-step fork / join between two streams costs next to the same kernels on one stream (the alignment loop's
nearest-neighbour chain and its splat chain are independent).   python3 tools/ubench_forkjoin.py"""
import time, torch
x = torch.zeros(1 << 20, device="cuda"); y = torch.zeros(1 << 20, device="cuda")
s1 = torch.cuda.current_stream(); s2 = torch.cuda.Stream()
e1 = [torch.cuda.Event() for _ in range(2)]; e2 = [torch.cuda.Event() for _ in range(2)]
def work(t, n):
    for _ in range(n): t.add_(1.0)
def seq(steps):
    for _ in range(steps):
        work(x, 1); work(x, 3); work(y, 3); work(x, 2)
def fj(steps):
    for i in range(steps):
        work(x, 1)
        a = e1[i & 1]; a.record(s1); s2.wait_event(a)
        with torch.cuda.stream(s2): work(y, 3)
        b = e2[i & 1]; b.record(s2)
        work(x, 3)
        s1.wait_event(b)
        work(x, 2)
for name, f in (("one stream", seq), ("fork / join", fj)):
    f(50); torch.cuda.synchronize()
    t0 = time.perf_counter(); f(2000); torch.cuda.synchronize()
    print("%-12s %.1f us per step of 9 small kernels" % (name, (time.perf_counter() - t0) / 2000 * 1e6))
