"""BASELINE config 4 fixture (`configs/config_lidar.yaml`, Waymo partial car scans, 4096 points).

    python tests/golden/make_waymo_c4.py            # in the build container: needs /root/reference/data/waymo/CAR

Writes tests/golden/waymo_car59_4096.npz -- DATA only (coordinates of the reference's bundled scans, subsampled):
  files[59], counts[59]   the CAR crops of data/waymo/CAR in sorted order and their point counts
  crops[59,4096,3]        each crop at 4096 points the way main.py:21-24 does it: deterministic FPS (start 0, the
                          oracle's) when it has more, pad-repeat (`p[arange(4096) % len(p)]`: exact duplicates) when fewer
  complete[4096,3]        a complete car for the crops to be registered against.  The reference generates it with
                          Trellis from the inpainted view (absent here); this one is made of real data instead: the
                          densest crop united with its mirror image across the car's vertical symmetry plane (the crops
                          are axis-aligned, long axis x, box-centred), FPS to 4096, box-normalised like normalize_numpy
  test_T[4,4], test_partial[4096,3]
                          a crop of `complete` under a KNOWN similarity (scale 0.84, 9 degrees about z, a small shift):
                          the half that faces the alignment loop's camera (at +z) kept, pad-repeated to 4096 -- the input of
                          tests/test_gpu_waymo_c4.py, whose golden transform (own deterministic loop) is committed
                          next to it by make_waymo_c4_golden.py on the GPU box.
"""
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import oracle as O  # noqa: E402


def to4096(p):
    p = p.astype(np.float32)
    if len(p) >= 4096:
        return p[O.fps(p, 4096)]
    return p[np.arange(4096) % len(p)]


def main():
    files = sorted(glob.glob(f"{REF}/data/waymo/CAR/*.ply"))
    assert len(files) == 59, len(files)
    raw = [O.read_ply_xyz(f) for f in files]
    counts = np.array([len(p) for p in raw])
    crops = np.stack([to4096(p) for p in raw])
    dense = raw[int(np.argmax(counts))].astype(np.float64)
    ext = dense.max(0) - dense.min(0)
    assert int(np.argmax(ext)) == 0, "densest crop: long axis x expected"
    mirrored = dense * np.array([1.0, -1.0, 1.0])          # y: the lateral axis of a box-centred crop
    both = np.concatenate([dense, mirrored]).astype(np.float32)
    comp = both[O.fps(both, 4096)].astype(np.float64)
    comp = (comp - (comp.max(0) + comp.min(0)) / 2) / (comp.max(0) - comp.min(0)).max()
    comp = comp.astype(np.float32)
    # the test pair: a known similarity the 201-step schedule can reach (scale from 0.75 moves <= 1e-3 per step)
    s, th, t = 0.84, np.deg2rad(9.0), np.array([0.02, -0.015, 0.01])
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])      # about the camera's axis: observable from one side
    c = comp.astype(np.float64).mean(0)
    posed = ((comp.astype(np.float64) - c) * s) @ R.T + c + t
    T = np.eye(4)
    T[:3, :3] = s * R
    T[:3, 3] = t
    near = posed[posed[:, 2] > np.median(posed[:, 2]) - 0.02]      # what the loop's camera at (0, 0, 3) sees, like the C5 generator
    part = near[np.arange(4096) % len(near)].astype(np.float32)
    out = os.path.join(HERE, "waymo_car59_4096.npz")
    np.savez_compressed(out, files=np.array([os.path.basename(f) for f in files]), counts=counts, crops=crops, complete=comp,
                        test_T=T, test_partial=part)
    print(out, os.path.getsize(out), "bytes; counts", counts.min(), counts.max(), "pad-repeated:", int((counts < 4096).sum()))


if __name__ == "__main__":
    main()
