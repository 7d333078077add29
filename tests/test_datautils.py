"""Host-side format helpers (no GPU): PLY round trip in the layout of the
reference's data/*.ply, normalisation and rotation helpers."""
import numpy as np

from genpc_amd.utils import dataUtils as D


def test_ply_roundtrip(tmp_path, oracle):
    rng = np.random.default_rng(0)
    xyz = rng.standard_normal((1234, 3))
    rgb = rng.random((1234, 3))
    p = str(tmp_path / "a.ply")
    D.save_ply_xyzrgb(xyz, rgb, p)
    x2, c2 = D.load_xyz(p)
    np.testing.assert_array_equal(x2, xyz)
    np.testing.assert_allclose(c2, np.rint(rgb * 255) / 255, atol=1e-12)
    np.testing.assert_array_equal(oracle.read_ply_xyz(p), xyz)          # the oracle's reader agrees
    D.save_ply_xyzrgb(xyz, None, p)
    x3, c3 = D.load_xyz(p)
    assert c3 is None
    np.testing.assert_array_equal(x3, xyz)
    header = open(p, "rb").read(200)
    assert header.startswith(b"ply\nformat binary_little_endian 1.0\n") and b"property double x" in header


def test_ascii_ply(tmp_path):
    p = tmp_path / "b.ply"
    p.write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
                 "end_header\n1 2 3\n4 5 6.5\n")
    x, c = D.load_xyz(str(p))
    np.testing.assert_array_equal(x, [[1, 2, 3], [4, 5, 6.5]])
    assert c is None


def test_normalize_and_rotate():
    rng = np.random.default_rng(1)
    xyz = rng.random((500, 3)) * np.array([2.0, 1.0, 0.5]) + 3.0
    n, c, s = D.normalize_numpy(xyz, range=0.5)
    assert abs((n.max(0) - n.min(0)).max() - 1.0) < 1e-12 and np.abs(n.max(0) + n.min(0)).max() < 1e-12
    np.testing.assert_allclose(n * s + c, xyz, atol=1e-12)
    for ax in "xyz":
        R = D.get_rotate_matrix(ax, 37.0)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)
        assert abs(np.linalg.det(R) - 1) < 1e-12
    np.testing.assert_allclose(D.get_rotate_matrix("y", 90) @ np.array([0, 0, 1.0]), [1, 0, 0], atol=1e-12)
