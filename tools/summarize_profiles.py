"""Turns the rocprofv3 CSVs that tools/collect_profiles.sh wrote under gpurun_out/prof_<tag>/ into the
small summaries committed under profiles/.

    python tools/summarize_profiles.py r02

Per family one JSON: for every library kernel its calls and average duration from the kernel trace
and the per-launch averages of the PMC passes.  HBM bytes follow
/opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE and WRITE_SIZE come from separate
passes, are in KiB, and FETCH_SIZE counts wide coalesced reads at half their size on gfx950, so
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     (per launch; WRITE_SIZE uncalibrated).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def short(name):
    n = name.replace("void ", "").replace("genpc::", "")
    return n.split("(")[0]


def trace_stats(d):
    out = {}
    for f in glob.glob(os.path.join(d, "trace", "*kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            if "genpc::" in r["Name"]:
                out[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": round(float(r["AverageNs"]) / 1e3, 2),
                                         "min_us": round(float(r["MinNs"]) / 1e3, 2), "max_us": round(float(r["MaxNs"]) / 1e3, 2),
                                         "total_ms": round(float(r["TotalDurationNs"]) / 1e6, 3)}
    return out


def counters(d, sub):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for f in glob.glob(os.path.join(d, sub, "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "genpc::" not in r["Kernel_Name"]:
                continue
            per[short(r["Kernel_Name"])][r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    out = {}
    for k, disp in per.items():
        rows = list(disp.values())
        keys = set().union(*[set(r) for r in rows])
        out[k] = {c: sum(r.get(c, 0.0) for r in rows) / len(rows) for c in sorted(keys)}
        out[k]["launches"] = len(rows)
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    # (on the GPU box: `summarize_profiles.py <tag> gpurun_out/profiles_<tag>` next to the collection, then the raw CSVs --
    # hundreds of MB of kernel traces -- can be deleted there and only the summaries travel back)
    dst = os.path.join(ROOT, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    ks = os.path.join(src, "bench", "bench_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, tag + "_bench_kernel_stats.csv"))
    for fam in sorted(os.listdir(src)):
        d = os.path.join(src, fam)
        if not os.path.isdir(os.path.join(d, "trace")):
            continue
        tr, fe, wr, sq = trace_stats(d), counters(d, "fetch"), counters(d, "write"), counters(d, "sq")
        kernels = {}
        for k in sorted(set(tr) | set(fe) | set(wr) | set(sq)):
            e = dict(tr.get(k, {}))
            if k in fe:
                e["FETCH_SIZE_KiB"] = round(fe[k].get("FETCH_SIZE", 0.0), 2)
            if k in wr:
                e["WRITE_SIZE_KiB"] = round(wr[k].get("WRITE_SIZE", 0.0), 2)
            if k in fe and k in wr:
                e["hbm_bytes_per_launch"] = (2 * fe[k].get("FETCH_SIZE", 0.0) + wr[k].get("WRITE_SIZE", 0.0)) * 1024
                if e.get("avg_us"):
                    e["hbm_GB_s_at_trace_duration"] = round(e["hbm_bytes_per_launch"] / (e["avg_us"] * 1e-6) / 1e9, 1)
            if k in sq:
                e["sq"] = {c: v for c, v in sq[k].items() if c != "launches"}
                w = sq[k].get("SQ_WAVE_CYCLES")
                if w:
                    e["sq_fractions_of_wave_cycles"] = {c: round(sq[k][c] / w, 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU")
                                                       if c in sq[k]}
            kernels[k] = e
        out = {"family": fam, "tag": tag, "command": "tools/collect_profiles.sh %s (rocprofv3 --kernel-trace --stats / --pmc ... -- python3 <driver>)" % tag,
               "kernels": kernels}
        with open(os.path.join(dst, "%s_%s.json" % (tag, fam)), "w") as fh:
            json.dump(out, fh, indent=1)
        tot = sum(v.get("total_ms", 0.0) for v in kernels.values())
        print("%-28s %2d kernels, %.2f ms traced; top: %s" % (fam, len(kernels), tot, ", ".join(
            "%s %.1fus" % (k, v["avg_us"]) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1].get("total_ms", 0))[:3] if "avg_us" in v)))


if __name__ == "__main__":
    main()
