"""Generates the committed fixtures under tests/golden/ (run in the build
container, where /root/reference exists; the GPU box only reads the .npz files).

    python tests/golden/make_golden.py [--scans]

What is stored is DATA: seeded synthetic inputs, deterministic FPS subsamples of
the reference's bundled scans (data/*.ply, data/GT/*.ply -- the reference's only
fixtures), and the outputs of the CPU oracle for them.
The oracle itself is pinned against the survey-time values of BASELINE.md
section 2 in tests/test_oracle_golden.py.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import oracle as O  # noqa: E402

REF = "/root/reference"
SCANS = ["01184", "01373", "05117", "05452", "06127", "06145", "06188", "06830",
         "07089", "07136", "07306", "09639", "09868"]


def gen(seed, s1, s2, shift):
    rng = np.random.default_rng(seed)
    a = rng.random(s1, dtype=np.float32) - np.float32(shift)
    b = rng.random(s2, dtype=np.float32) - np.float32(shift)
    return a, b


def chamfer_case(name, a, b):
    out = dict(xyz1=a, xyz2=b)
    for mode in (0, 1):
        d1, d2, i1, i2 = O.chamfer_forward(a, b, mode)
        out.update({f"dist1_m{mode}": d1, f"dist2_m{mode}": d2, f"idx1_m{mode}": i1, f"idx2_m{mode}": i2,
                    f"cd_l1_m{mode}": O.cd_l1(d1, d2), f"cd_l2_m{mode}": O.cd_l2(d1, d2)})
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, out["cd_l1_m0"], out["cd_l1_m1"])


def emd_case(name, a, b, eps, iters):
    out = dict(xyz1=a, xyz2=b, eps=np.float32(eps), iters=np.int32(iters))
    for mode in (0, 1):
        d, ass, st = O.emd_forward(a, b, eps, iters, mode, return_state=True)
        out.update({f"dist_m{mode}": d, f"assignment_m{mode}": ass, f"price_m{mode}": st["price"],
                    f"assignment_inv_m{mode}": st["assignment_inv"], f"emd_m{mode}": O.emd_loss(d)})
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, out["emd_m0"], out["emd_m1"])


def main():
    # Chamfer: the two survey cases + ragged / tiny / duplicate-point cases
    a, b = gen(1, (1, 2048, 3), (1, 2048, 3), 0.5)
    chamfer_case("chamfer_seed1_b1_2048.npz", a, b)
    a, b = gen(0, (2, 1000, 3), (2, 777, 3), 0.5)
    chamfer_case("chamfer_seed0_b2_1000x777.npz", a, b)
    a, b = gen(7, (3, 5, 3), (3, 3, 3), 0.5)
    chamfer_case("chamfer_seed7_b3_5x3.npz", a, b)
    a, b = gen(11, (2, 300, 3), (2, 260, 3), 0.5)
    b[:, 100:200] = b[:, 0:100]          # duplicated targets: index ties
    a[:, 0:50] = b[:, 100:150]           # exact hits: zero distances
    chamfer_case("chamfer_seed11_dups.npz", a, b)

    # EMD: survey cases + converged + duplicate points
    a, b = gen(0, (2, 1024, 3), (2, 1024, 3), 0.0)
    emd_case("emd_seed0_b2_1024.npz", a, b, 0.005, 50)
    a, b = gen(2, (1, 256, 3), (1, 256, 3), 0.0)
    emd_case("emd_seed2_b1_256_conv.npz", a, b, 0.002, 3000)
    a, b = gen(5, (2, 512, 3), (2, 512, 3), 0.0)
    b[:, 256:512] = b[:, 0:256]          # every object duplicated once: bid ties
    emd_case("emd_seed5_dups.npz", a, b, 0.005, 50)
    a, b = gen(9, (1, 2304, 3), (1, 2304, 3), 0.0)   # crosses the reference's 2048-tile
    b[:, 2048:2304] = b[:, 0:256]
    emd_case("emd_seed9_2304_dups.npz", a, b, 0.005, 20)

    if os.path.isdir(REF):
        p = O.read_ply_xyz(f"{REF}/data/01184.ply").astype(np.float32)
        g = O.read_ply_xyz(f"{REF}/data/GT/01184.ply").astype(np.float32)
        P = p[O.fps(p, 2048)][None]
        G = g[O.fps(g, 2048)][None]
        out = dict(partial=P, gt=G)
        for mode in (0, 1):
            d1, d2, i1, i2 = O.chamfer_forward(P, G, mode)
            d, ass = O.emd_forward(P, G, 0.005, 50, mode)
            out.update({f"cd_l1_m{mode}": O.cd_l1(d1, d2), f"cd_l2_m{mode}": O.cd_l2(d1, d2),
                        f"emd_m{mode}": O.emd_loss(d), f"dist1_m{mode}": d1, f"idx1_m{mode}": i1,
                        f"dist2_m{mode}": d2, f"idx2_m{mode}": i2, f"assignment_m{mode}": ass})
        np.savez_compressed(os.path.join(HERE, "scan01184_fps2048.npz"), **out)
        print("scan01184", out["cd_l1_m0"], out["cd_l2_m0"], out["emd_m0"])

    if "--scans" in sys.argv and os.path.isdir(REF):
        # BASELINE config 3: all 13 bundled scans at 16384 points
        parts, gts = [], []
        for s in SCANS:
            p = O.read_ply_xyz(f"{REF}/data/{s}.ply").astype(np.float32)
            g = O.read_ply_xyz(f"{REF}/data/GT/{s}.ply").astype(np.float32)
            parts.append(p[O.fps(p, 16384)])
            gts.append(g[O.fps(g, 16384)])
            print("fps", s, p.shape, g.shape, flush=True)
        P = np.stack(parts)
        G = np.stack(gts)
        out = dict(ids=np.array(SCANS), partial=P, gt=G)
        for mode in (0, 1):
            d1, d2, i1, i2 = O.chamfer_forward(P, G, mode)
            d, ass = O.emd_forward(P, G, 0.005, 50, mode)
            out[f"cd_l1_m{mode}"] = np.array([O.cd_l1(d1[i], d2[i]) for i in range(len(SCANS))], np.float32)
            out[f"cd_l2_m{mode}"] = np.array([O.cd_l2(d1[i], d2[i]) for i in range(len(SCANS))], np.float32)
            out[f"emd_m{mode}"] = np.sqrt(d).mean(axis=1, dtype=np.float32)
            out[f"idx1_sum_m{mode}"] = i1.astype(np.int64).sum(axis=1)
            out[f"idx2_sum_m{mode}"] = i2.astype(np.int64).sum(axis=1)
            out[f"assignment_sum_m{mode}"] = ass.astype(np.int64).sum(axis=1)
            print("mode", mode, out[f"cd_l1_m{mode}"], out[f"emd_m{mode}"], flush=True)
        np.savez_compressed(os.path.join(HERE, "scans13_fps16384.npz"), **out)


def waymo():
    """BASELINE config 4 inputs: the first 8 Waymo CAR crops resampled to 4096 points
    (FPS when larger, pad-repeat when smaller: duplicated points exercise the tie
    paths on real data), each scored against the next one."""
    import glob
    files = sorted(glob.glob(f"{REF}/data/waymo/CAR/*.ply"))[:8]
    clouds, counts = [], []
    for f in files:
        p = O.read_ply_xyz(f).astype(np.float32)
        counts.append(len(p))
        if len(p) >= 4096:
            clouds.append(p[O.fps(p, 4096)])
        else:
            clouds.append(p[np.arange(4096) % len(p)])
    X = np.stack(clouds)
    Y = np.roll(X, -1, axis=0).copy()
    out = dict(files=np.array([os.path.basename(f) for f in files]), counts=np.array(counts), xyz1=X, xyz2=Y)
    for mode in (0, 1):
        d1, d2, i1, i2 = O.chamfer_forward(X, Y, mode)
        d, ass, st = O.emd_forward(X, Y, 0.005, 50, mode, return_state=True)
        out.update({f"cd_l1_m{mode}": np.array([O.cd_l1(d1[i], d2[i]) for i in range(8)], np.float32),
                    f"idx1_m{mode}": i1, f"idx2_m{mode}": i2, f"dist1_m{mode}": d1,
                    f"assignment_m{mode}": ass, f"emd_dist_m{mode}": d,
                    f"emd_m{mode}": np.sqrt(d).mean(axis=1, dtype=np.float32)})
        print("waymo mode", mode, counts, out[f"cd_l1_m{mode}"], out[f"emd_m{mode}"])
    np.savez_compressed(os.path.join(HERE, "waymo_car8_4096.npz"), **out)


if __name__ == "__main__":
    if "--waymo" in sys.argv:
        waymo()
    else:
        main()
