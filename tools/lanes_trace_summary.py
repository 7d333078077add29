"""What the GPU does while several completed scans are in flight: from a rocprofv3 kernel trace of tools/time_c2_lanes.py, over the
last `frac` of the traced time (the timed region): the fraction of the time at least one kernel runs, the mean number of kernels
running, the queues used, and per kernel the calls, the average duration and the sum (to set beside the trace of one scan alone).
   python3 tools/lanes_trace_summary.py <t_kernel_trace.csv> [frac]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void genpc::", ""), r["Queue_Id"]) for r in rows]
ev.sort()
t_end = max(e[1] for e in ev)
t_beg = ev[0][0]
w0 = t_end - int((t_end - t_beg) * frac)
sel = [e for e in ev if e[0] >= w0]
span = t_end - w0
pts = []
for s, e, _, _ in sel:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
busy = 0; area = 0; depth = 0; last = w0; hist = collections.Counter()
for t, d in pts:
    if depth > 0: busy += t - last
    area += depth * (t - last); hist[min(depth, 12)] += t - last
    depth += d; last = t
print("window %.1f ms: %d kernels, busy %.3f, mean kernels running %.2f, queues %d" % (span / 1e6, len(sel), busy / span, area / span, len(set(e[3] for e in sel))))
print("time share by number of kernels running:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, _ in sel:
    agg[n][0] += 1; agg[n][1] += e - s
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-52s calls %6d avg %9.1f us  sum %8.2f ms (%.2f of window)" % (n[:52], c, t / c / 1e3, t / 1e6, t / span))
# gaps between consecutive kernels of one queue (launch / dependency latency as the GPU saw it)
byq = collections.defaultdict(list)
for s, e, n, q in sel: byq[q].append((s, e))
gaps = []
for q, l in byq.items():
    l.sort()
    for a, b in zip(l, l[1:]): gaps.append(max(0, b[0] - a[1]))
gaps.sort()
if gaps:
    print("gap between consecutive kernels of a queue: median %.1f us, mean %.1f us, p90 %.1f us" % (gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3))
