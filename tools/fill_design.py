"""DESIGN.md = tools/DESIGN.template.md with the <...> tokens replaced by the numbers of a bench.py line:
    python tools/fill_design.py gpurun_out/bench_r05.json"""
import json, sys, os
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e = d["extra"]
r = d["roofline"]
tok = {}
tok["NN_FRAC"] = "%.3f" % r["frac"]
tok["NN_KERNEL_US"] = "%.1f" % (r["ms_per_launch"] * 1e3)
tok["NN_STEP_US"] = "%.1f" % (d["ms_per_step"] * 1e3)
tok["NN_TPAIR"] = "%.1f" % (d["value"] / 1e3)
tok["EMD_FRAC"] = "%.3f" % e["emd_fwd_n16384_roofline"]["frac"]
rows = []
for k, v in e["streaming_rooflines_64x32768"].items():
    rows.append("| %s | `%s` | %.0f | %.1f µs | %.0f | %.3f |" % (v["row"], v["kernel"], v["algorithmic_bytes"] / v["points"], v["ms_per_launch"] * 1e3, v["achieved"], v["frac"]))
tok["STREAMING_TABLE"] = "\n".join(rows)
u = e["get_uvs_1024x71372_roofline"]
tok["UVS_US"] = "%.0f" % (u["ms_per_call"] * 1e3)
tok["UVS_FRAC"] = "%.3f" % u["frac"]
tok["UVS_FRAC629"] = "%.2f" % u["frac_of_6.29TBs_achievable"]
tok["REG_LINE"] = ("registration + metric of one scan %.1f scans/s (six such scans in flight %.1f), eight scans in lock-step %.1f (three groups in flight "
                   "%.1f); FPS 4 x 165546 -> 16384 in %.1f ms including its verification; the C2 chain (8192-point partial -> completed scan, metric "
                   "included) %.1f scans/s one at a time, %.1f with six in flight; the C5 rank shape (8 x 32768) %.1f scans/s"
                   % (e["registration_8k_vs_16k_4x201_plus_metric_scans_per_s"], e["registration_8k_vs_16k_6_scans_in_flight_scans_per_s"],
                      e["registration_batch8_8k_vs_16k_4x201_plus_metric_scans_per_s"], e["registration_batch8_3_groups_in_flight_scans_per_s"],
                      e["fps_4x165546_to_16384_ms"], e["c2_pipeline_8192_scans_per_s"], e["c2_pipeline_8192_scans_in_flight_scans_per_s"],
                      e["c5_rank_8x32768_registration_plus_metric_scans_per_s"]))
tok["HPR_1024"] = "%.1f" % e["hpr_1024x10000_R10000_ms"]
tok["HPR_2"] = "%.1f" % e["hpr_2x165546_R10000_ms"]
tok["FPS_VERIFY"] = os.environ.get("FPS_VERIFY_SHARE", "a tenth")
tok["HPR_DECIDE"] = os.environ.get("HPR_DECIDE_MS", "7.9 ms")
tok["BENCH_LINE"] = ("%.0f Gpair/s (%.1f µs per step, spread %.1f–%.1f), `roofline.frac` %.3f, CPU port %.2f Gpair/s on %d threads; EMD 1 × 16384 %.2f ms, "
                     "13 bundled scans %.1f ms, 13 uniform %.2f ms; metric %.0f scans/s (`%s`)"
                     % (d["value"], d["ms_per_step"] * 1e3, d["ms_per_step_spread"]["min"] * 1e3, d["ms_per_step_spread"]["max"] * 1e3, r["frac"],
                        d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], e["emd_fwd_n16384_eps0.005_it50_ms"],
                        e["emd_fwd_13_bundled_scans_n16384_ms"], e["emd_fwd_B13_uniform_n16384_ms"], e["metric_cd_emd_n16384_scans_per_s"],
                        os.path.relpath(sys.argv[1], ROOT) if not sys.argv[1].startswith("profiles") else sys.argv[1]))
st_ = e["streaming_rooflines_64x32768"]
tok["A3_FRAC"] = "%.2f" % st_["a3_chamfer_backward"]["frac"]
tok["A3_US"] = "%.0f" % (st_["a3_chamfer_backward"]["ms_per_launch"] * 1e3)
tok["A15_FRAC"] = "%.2f" % st_["a15_gather_colors"]["frac"]
tok["A14_FRAC"] = "%.2f" % st_["a14_paint_pixels"]["frac"]
tok["C2_ALONE"] = "%.1f" % e["c2_pipeline_8192_scans_per_s"]
tok["C2_FLIGHT"] = "%.1f" % e["c2_pipeline_8192_scans_in_flight_scans_per_s"]
tok["FPS_C2_MS"] = "%.1f" % e.get("fps_scan_24576_to_20000_ms", float("nan"))
tok["FPS_M_MS"] = "%.1f" % e.get("fps_scan_20000_to_16384_ms", float("nan"))
tok["FPS_BIG_MS"] = "%.1f" % e["fps_4x165546_to_16384_ms"]
tok["EMD_1"] = "%.2f" % e["emd_fwd_n16384_eps0.005_it50_ms"]
tok["EMD_13U"] = "%.2f" % e["emd_fwd_B13_uniform_n16384_ms"]
tok["EMD_13S"] = "%.1f" % e["emd_fwd_13_bundled_scans_n16384_ms"]
tok["METRIC"] = "%.0f" % e["metric_cd_emd_n16384_scans_per_s"]
tok["REG8"] = "%.1f" % e["registration_batch8_8k_vs_16k_4x201_plus_metric_scans_per_s"]
tok["REG6"] = "%.1f" % e["registration_8k_vs_16k_6_scans_in_flight_scans_per_s"]
tok["EMD_PPC_NOTE"] = os.environ.get("EMD_PPC_NOTE", "four objects per cell instead of two gives 0.95 against 1.06 ms at 1 x 16384 and is the default from this round on")
# the renderer sensitivity table (tools/renderer_sensitivity.py)
rs = json.load(open(os.path.join(ROOT, "profiles", "r06_renderer_sensitivity.json")))
rows = ["  | input | fall-off | depth in the exponent | CD-L1 partial → aligned | vs default | scale | vs default | winning start |", "  |---|---|---|---|---|---|---|---|"]
for r_ in rs["rows"]:
    rows.append("  | %s | %s | %s | %.5f | %+.2f %% | %.4f | %+.2f %% | %d |" % (r_["case"].split(":")[0], r_["falloff"], r_["depth"], r_["cd_partial_l1"],
                100 * r_["cd_vs_default"], r_["scale"], 100 * r_["scale_vs_default"], r_["best_start"]))
tok["RENDER_TABLE"] = "\n".join(rows)
s = open(os.path.join(ROOT, "tools", "DESIGN.template.md")).read()
for k, v in tok.items():
    s = s.replace("⟨%s⟩" % k, v)
left = [w for w in s.split("⟨")[1:]]
assert not left, [w[:30] for w in left]
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
print("DESIGN.md written: %d bytes" % len(s.encode()))
