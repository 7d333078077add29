"""How much of a completed scan's wall time the GPU runs nothing: from a rocprofv3 kernel trace of eight scans one at a time, the
union of kernel intervals over the last six scans, and the idle gaps by what precedes them.
   rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o t -- python3 tools/c2_gpu_idle.py run ; python3 tools/c2_gpu_idle.py <t_kernel_trace.csv>"""
import os, sys, time, csv, collections
if sys.argv[1] == "run":
    ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from genpc_amd import pipeline, _lib
    from genpc_amd.DepthPrompting import DepthPrompting
    exec(open(os.path.join(ROOT, "tools", "time_c2_streams.py")).read().split("def run(")[0])
    for _ in range(2): pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(6): pipeline.complete_scan(part, gen_s, img, gt_s, cfg=cfg, dp=dp)
    torch.cuda.synchronize(); print("6 scans: %.1f ms per scan" % ((time.perf_counter() - t0) / 6 * 1e3))
    sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void genpc::", "").replace("genpc::", "")) for r in rows)
# the last six scans: from the 7th-last emd_auction kernel's end (a scan ends with its metric's auction) to the last one's
au = [e for e in ev if e[2].startswith("emd_auction_kernel")]
w0, w1 = au[-7][1], au[-1][1]
sel = [e for e in ev if e[1] > w0 and e[0] < w1]
cur_end = w0; idle = 0; gaps = []
last_name = "(start)"
for s, e, n in sel:
    if s > cur_end:
        gaps.append((s - cur_end, last_name, n)); idle += s - cur_end
    if e > cur_end: cur_end = e; last_name = n
print("six scans: %.2f ms per scan on the GPU's clock, idle %.2f ms per scan (%.1f %%), %d kernels per scan" % ((w1 - w0) / 6e6, idle / 6e6, 100.0 * idle / (w1 - w0), len(sel) // 6))
by = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    by[(a, b)][0] += 1; by[(a, b)][1] += g
print("idle time by (kernel before -> kernel after), per scan:")
for (a, b), (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:28]:
    print("  %-44s -> %-44s %5.1f x  %7.1f us each  %6.2f ms" % (a[:44], b[:44], c / 6.0, t / c / 1e3, t / 6e6))
