"""Turns the rocprofv3 CSVs that tools/collect_profiles.sh wrote under
gpurun_out/prof_<tag>/ into the small summaries committed under profiles/.

    python tools/summarize_profiles.py r01

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE
and WRITE_SIZE come from separate passes, are in KiB, and FETCH_SIZE counts wide
coalesced reads at half their size on gfx950, so
    hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024     (per launch).
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def counters(path, kernel_substr):
    per = collections.defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel_substr in r["Kernel_Name"]:
                d = per[r["Dispatch_Id"]]
                d[r["Counter_Name"]] = float(r["Counter_Value"])
                d["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                d["_grid"] = int(r["Grid_Size"])
                d["_lds"] = int(r["LDS_Block_Size"])
                d["_kernel"] = r["Kernel_Name"]
    rows = list(per.values())
    if not rows:
        return {}
    keys = [k for k in rows[-1] if not k.startswith("_")]
    out = {k: sum(r[k] for r in rows) / len(rows) for k in keys}
    out["launches"] = len(rows)
    out["avg_duration_us_under_pmc"] = round(sum(r["_dur_us"] for r in rows) / len(rows), 2)
    out["grid_threads"] = rows[-1]["_grid"]
    out["lds_bytes_per_block"] = rows[-1]["_lds"]
    out["kernel_name"] = rows[-1]["_kernel"]
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    ks = os.path.join(src, "bench", "bench_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(dst, tag + "_bench_kernel_stats.csv"))
    for name, b, n, fetch, write, sq in (("chamfer_B1_16384", 1, 16384, "fetch", "write", "sq"),
                                         ("chamfer_B13_16384", 13, 16384, "fetch13", None, "sq13")):
        # dominant kernel: the f16 MFMA filter; the finish kernel's counters are kept beside it
        d = {"kernel": "nn_f16_kernel", "batch": b, "points": n, "tag": tag,
             "command": "rocprofv3 --pmc <counters> -- python3 tools/prof_chamfer.py %d %d" % (b, n)}
        f = counters(os.path.join(src, fetch, "c_counter_collection.csv"), "nn_f16_kernel") if fetch else {}
        w = counters(os.path.join(src, write, "c_counter_collection.csv"), "nn_f16_kernel") if write else {}
        s = counters(os.path.join(src, sq, "c_counter_collection.csv"), "nn_f16_kernel") if sq else {}
        ff = counters(os.path.join(src, fetch, "c_counter_collection.csv"), "nn_finish_kernel") if fetch else {}
        fw = counters(os.path.join(src, write, "c_counter_collection.csv"), "nn_finish_kernel") if write else {}
        fs = counters(os.path.join(src, sq, "c_counter_collection.csv"), "nn_finish_kernel") if sq else {}
        d["finish_kernel"] = {"FETCH_SIZE_KiB": ff.get("FETCH_SIZE"), "WRITE_SIZE_KiB": fw.get("WRITE_SIZE"), "sq": fs}
        if ff and fw:
            d["finish_kernel"]["hbm_bytes_per_launch"] = (2 * ff["FETCH_SIZE"] + fw["WRITE_SIZE"]) * 1024
        if f:
            d["FETCH_SIZE_KiB"] = f.get("FETCH_SIZE")
        if w:
            d["WRITE_SIZE_KiB"] = w.get("WRITE_SIZE")
        if f and w:
            d["hbm_bytes_per_launch"] = (2 * f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024
        alg = 20.0 * 2 * n * b
        d["algorithmic_bytes_per_launch"] = alg
        if "hbm_bytes_per_launch" in d:
            d["traffic_over_algorithmic"] = round(d["hbm_bytes_per_launch"] / alg, 2)
        d["sq"] = s
        if s:
            d["kernel_name"] = s.get("kernel_name")
        with open(os.path.join(dst, "%s_%s_pmc.json" % (tag, name)), "w") as fh:
            json.dump(d, fh, indent=1)
        print(name, json.dumps(d)[:600])


if __name__ == "__main__":
    main()
