"""How much the registration outcome depends on what of Pulsar's renderer is restated FROM MEMORY (VERDICT r5
item 6).  Runs on the CPU, in the oracle only (oracle/genpc_oracle_geom.c: oracle_set_render_variant -- the kernels implement
the defaults): the full alignment loop (4 starts x 201 Adam steps, full objective, patience 300) on the three configurations'
pose-loop inputs under the four combinations of
    fall-off   quadratic a = 1 - r^2 / rho^2 (default)      | linear a = 1 - r / rho
    depth      the sphere's centre in the exponent (default) | the ray-sphere hit
and reports, per run, the Chamfer distance partial -> aligned complete (CD-L1, the quantity the pose initialisation exists to make
small), the recovered scale and the winning start.  (The third choice, a projected ellipse instead of a disc, is not switchable:
at focal 4 and objects within +-0.5 of the axis the ellipse's axes differ by < 1.5 %.)

    python3 tools/renderer_sensitivity.py [iters]        # ~2 minutes on 8 cores; writes profiles/r06_renderer_sensitivity.json
"""
import json, math, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
O.set_blend(1)


def rot(axis, deg):
    a = np.asarray(axis, np.float64); a /= np.linalg.norm(a); t = math.radians(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(t) * K + (1 - math.cos(t)) * K @ K


def c5_scan(seed, n=32768):      # tests/test_gpu_pipeline.py::c5_scan (SURVEY 8d's scan bench)
    rng = np.random.default_rng(1000 + seed)
    u = rng.standard_normal((n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    ell = u * np.array([0.5, 0.3, 0.22])
    box = (rng.random((n, 3)) - 0.5) * np.array([0.3, 0.5, 0.3])
    face = rng.integers(0, 3, n)
    box[np.arange(n), face] = np.sign(box[np.arange(n), face]) * np.array([0.15, 0.25, 0.15])[face]
    pick = rng.random(n) < 0.6
    complete = np.where(pick[:, None], ell, box + np.array([0.1, 0.0, 0.0]))
    complete = (complete - (complete.max(0) + complete.min(0)) / 2) / (complete.max(0) - complete.min(0)).max()
    s = rng.uniform(0.78, 0.9); theta = rng.uniform(-12, 12); t = rng.uniform(-0.04, 0.04, 3)
    R = rot([0, 1, 0], theta); c = complete.mean(0)
    posed = ((complete - c) * s) @ R.T + c + t
    keep = np.nonzero(posed[:, 2] > np.median(posed[:, 2]) - 0.02)[0]
    sel = keep[rng.integers(0, keep.shape[0], n)]
    partial = posed[sel]
    # colours: a smooth function of the rest-frame position, the same for a point of the complete shape and its observation
    col = (0.5 + 0.5 * np.sin(7.0 * complete + np.array([0.3, 1.1, 2.0]))).astype(np.float32)
    return complete.astype(np.float32), partial.astype(np.float32), s, col, col[sel]


cases = []
z13 = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
gt0 = z13["gt"][0]
cc = (gt0.max(0) + gt0.min(0)) / 2
Rg = rot([0.2, 1.0, 0.1], 9.0)
gen = (((gt0 - cc) / (gt0.max(0) - gt0.min(0)).max()).astype(np.float64) @ Rg.T).astype(np.float32)
cases.append(("C2 scan 01184: partial 8192 vs generated shape (both voxel 0.02, as reg() feeds the loop), white as in the pipeline",
              O.voxel_down_sample(gen, 0.02), O.voxel_down_sample(z13["partial"][0][:8192].copy(), 0.02), None, None, None))
w = np.load(os.path.join(ROOT, "tests", "golden", "waymo_car59_4096.npz"))
cases.append(("C4 Waymo car: test crop 4096 vs complete car 4096, white", w["complete"], w["test_partial"], None, None, None))
comp, part, s_true, ccol, pcol = c5_scan(0)
cases.append(("C5 synthetic scan 0 (32768 points, voxel 0.02 for this study), white", O.voxel_down_sample(comp, 0.02), O.voxel_down_sample(part, 0.02), float(s_true), None, None))
cv, cvc = O.voxel_down_sample(comp, 0.02, colors=ccol)
pv, pvc = O.voxel_down_sample(part, 0.02, colors=pcol)
cases.append(("C5 synthetic scan 0, COLOURED (a smooth colour field on the shape: the blend's weights now show in the image)", cv, pv, float(s_true), cvc, pvc))

out = {"iters": iters, "starts": 4, "note": __doc__.split("\n\n")[0].replace("\n", " "), "rows": []}
for name, complete, partial, s_true, ccol_, pcol_ in cases:
    print("%s: %d vs %d points" % (name, len(complete), len(partial)), flush=True)
    base = None
    for fl, dh in ((0, 0), (1, 0), (0, 1), (1, 1)):
        O.set_render_variant(fl, dh)
        t0 = time.time()
        T, hist, bp = O.pose_optimize(complete, partial, lr=0.01, iters=iters, starts=4, radius=0.02, size=224, complete_col=ccol_, partial_col=pcol_)
        O.set_render_variant(0, 0)
        c = complete.astype(np.float64).mean(0)
        aligned = ((complete - c) @ T[:3, :3].T.astype(np.float64) + c + T[:3, 3]).astype(np.float32)
        d1, _, _, _ = O.chamfer_forward(partial[None], aligned[None], 1)
        cd = float(np.sqrt(d1).mean())
        scale = float(np.cbrt(np.linalg.det(T[:3, :3].astype(np.float64))))
        best = int(np.nanargmin(np.nanmin(hist, axis=1)))
        row = {"case": name, "falloff": "linear" if fl else "quadratic", "depth": "ray-sphere hit" if dh else "centre",
               "cd_partial_l1": round(cd, 6), "scale": round(scale, 5), "best_start": best, "final_loss": round(float(np.nanmin(hist[best])), 5),
               "true_scale": s_true, "seconds": round(time.time() - t0, 1)}
        if base is None:
            base = row
        row["cd_vs_default"] = round(cd / base["cd_partial_l1"] - 1.0, 4)
        row["scale_vs_default"] = round(scale / base["scale"] - 1.0, 4)
        out["rows"].append(row)
        print("   fall-off %-9s depth %-14s  CD-L1 partial %.5f (%+.2f %%)  scale %.4f (%+.2f %%)  start %d  loss %.4f  [%.0f s]"
              % (row["falloff"], row["depth"], cd, 100 * row["cd_vs_default"], scale, 100 * row["scale_vs_default"], best, row["final_loss"], row["seconds"]), flush=True)
with open(os.path.join(ROOT, "profiles", "r06_renderer_sensitivity.json"), "w") as f:
    json.dump(out, f, indent=1)
