"""Times and cross-checks the nearest-neighbour kernel families on one process:
    python tools/nn_paths.py [BxNxM ...]          (default: the benchmark shapes)
prints per shape and family: us per call, nominal Gpair/s, and whether the bits equal the VALU family's."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from genpc_amd import _lib, chamfer_3D  # noqa: E402

PATHS = {"valu": 0, "mfma32": 1, "f16": 3, "grid": 4}


def clouds(kind, b, n, m, rng):
    if kind == "scan":
        g = np.load(os.path.join(ROOT, "tests", "golden", "scans13_fps16384.npz"))
        a = np.stack([g["partial"][i % 13][:n] for i in range(b)])
        c = np.stack([g["gt"][i % 13][:m] for i in range(b)])
        return np.ascontiguousarray(a), np.ascontiguousarray(c)
    a = rng.random((b, n, 3), dtype=np.float32) - np.float32(0.5)
    c = rng.random((b, m, 3), dtype=np.float32) - np.float32(0.5)
    if kind == "surface":      # points on a sphere / an ellipsoid
        a /= np.linalg.norm(a, axis=2, keepdims=True)
        c /= np.linalg.norm(c, axis=2, keepdims=True)
        c *= np.float32([1.0, 0.8, 0.6])
    return a, c


def main():
    specs = [s for s in sys.argv[1:] if not s.startswith("--")] or ["1x2048x2048", "1x8192x8192", "1x16384x16384", "1x32768x32768", "13x16384x16384",
                                                                     "8x8192x16384", "64x2048x2048"]
    kinds = ["uniform", "surface", "scan"]
    rng = np.random.default_rng(20250101)
    for spec in specs:
        b, n, m = [int(x) for x in spec.split("x")]
        for kind in kinds:
            if kind == "scan" and max(n, m) > 16384:
                continue
            a, c = clouds(kind, b, n, m, rng)
            A, C = torch.from_numpy(a).cuda(), torch.from_numpy(c).cuda()
            d1 = torch.empty(b, n, device="cuda"); d2 = torch.empty(b, m, device="cuda")
            i1 = torch.empty(b, n, device="cuda", dtype=torch.int32); i2 = torch.empty(b, m, device="cuda", dtype=torch.int32)
            ref = None
            row = []
            for name, pid in PATHS.items():
                if name == "valu" and b * n * m > 4e9:
                    continue
                prev = _lib.lib.genpc_nn_tune(pid, 0)
                for _ in range(3):
                    chamfer_3D.forward(A, C, d1, d2, i1, i2)
                torch.cuda.synchronize()
                reps = 20
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    chamfer_3D.forward(A, C, d1, d2, i1, i2)
                e1.record(); e1.synchronize()
                us = e0.elapsed_time(e1) / reps * 1e3
                _lib.lib.genpc_nn_tune(prev, 0)
                got = [t.clone() for t in (d1, d2, i1, i2)]
                if ref is None:
                    ref = got
                same = all(torch.equal(x.view(torch.int32), y.view(torch.int32)) for x, y in zip(got, ref))
                row.append("%s %.1fus %.0fG %s" % (name, us, 2.0 * b * n * m / us / 1e3, "ok" if same else "MISMATCH"))
            print("%-16s %-8s %s" % (spec, kind, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
