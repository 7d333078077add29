"""Host-side format helpers (no GPU): PLY round trip in the layout of the
reference's data/*.ply, normalisation and rotation helpers."""
import numpy as np

from conftest import write_glb as _write_glb
import pytest

from genpc_amd.utils import dataUtils as D


def test_ply_roundtrip(tmp_path, oracle):
    rng = np.random.default_rng(0)
    xyz = rng.standard_normal((1234, 3))
    rgb = rng.random((1234, 3))
    p = str(tmp_path / "a.ply")
    D.save_ply_xyzrgb(xyz, rgb, p)
    x2, c2 = D.read_ply(p)
    np.testing.assert_array_equal(x2, xyz)
    np.testing.assert_allclose(c2, np.rint(rgb * 255) / 255, atol=1e-12)
    np.testing.assert_array_equal(oracle.read_ply_xyz(p), xyz)          # the oracle's reader agrees
    D.save_ply_xyzrgb(xyz, None, p)
    x3, c3 = D.read_ply(p)
    assert c3 is None
    np.testing.assert_array_equal(x3, xyz)
    header = open(p, "rb").read(200)
    assert header.startswith(b"ply\nformat binary_little_endian 1.0\n") and b"property double x" in header


def test_ascii_ply(tmp_path):
    p = tmp_path / "b.ply"
    p.write_text("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\n"
                 "end_header\n1 2 3\n4 5 6.5\n")
    x, c = D.read_ply(str(p))
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


def test_glb_loader_and_surface_sampling(tmp_path):
    from genpc_amd.utils import mesh_io as M
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], float)
    faces = np.array([[0, 1, 2], [0, 1, 3], [0, 2, 3], [1, 2, 3]])
    cols = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]], float)
    p = str(tmp_path / "t.glb")
    node = {"translation": [1.0, 2.0, 3.0], "scale": [2.0, 2.0, 2.0], "rotation": [0.0, 0.0, 0.7071067811865476, 0.7071067811865476]}
    _write_glb(p, verts, faces, cols, node)
    V, F, C = M.load_glb(p)
    Rz = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])            # 90 degrees about z
    np.testing.assert_allclose(V, (verts * 2) @ Rz.T + [1, 2, 3], atol=1e-6)
    np.testing.assert_array_equal(F, faces)
    np.testing.assert_allclose(C, cols)
    _write_glb(p, verts, faces, None, {"matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 5, 6, 7, 1]}, indices_u16=False)
    V2, F2, C2 = M.load_glb(p)
    np.testing.assert_allclose(V2, verts + [5, 6, 7], atol=1e-6)
    assert C2 is None
    rng = np.random.default_rng(0)
    pts, fi = M.sample_surface(verts, faces, 40000, rng)
    tri = verts[faces[fi]]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    assert np.abs(((pts - tri[:, 0]) * n).sum(1)).max() < 1e-12       # on the face's plane ...
    assert (pts >= -1e-12).all() and (pts.sum(1) <= 1 + 1e-9).all()    # ... and inside the tetrahedron's hull
    area = 0.5 * np.linalg.norm(np.cross(verts[faces][:, 1] - verts[faces][:, 0], verts[faces][:, 2] - verts[faces][:, 0]), axis=1)
    np.testing.assert_allclose(np.bincount(fi, minlength=4) / 40000, area / area.sum(), atol=0.01)
    _write_glb(p, verts, faces, cols, None)
    P, Cc = M.glb2point(p, num_points=5000, rng=np.random.default_rng(1))
    assert P.shape == (5000, 3) and Cc.shape == (5000, 3) and Cc.min() >= 0 and Cc.max() <= 1
    # on face (0,1,2) (z = 0) the colour is the barycentric blend of red / green / blue
    on = np.abs(P[:, 2]) < 1e-12
    np.testing.assert_allclose(Cc[on], np.stack([1 - P[on, 0] - P[on, 1], P[on, 0], P[on, 1]], 1), atol=1e-9)


def _write_textured_glb(path, verts, faces, uv, png_bytes=None, data_uri=False, factor=None):
    """One textured primitive: TEXCOORD_0 + a material whose base-colour texture is an embedded PNG
    (bufferView or data URI), or a material with only a baseColorFactor."""
    import base64
    import json
    import struct
    v = np.asarray(verts, "<f4")
    idx = np.asarray(faces, "<u2").reshape(-1)
    t = np.asarray(uv, "<f4")
    blobs, views = [], []

    def add(b):
        views.append({"buffer": 0, "byteOffset": sum(map(len, blobs)), "byteLength": len(b)})
        blobs.append(b + b"\x00" * (-len(b) % 4))
        return len(views) - 1
    accs = [{"bufferView": add(v.tobytes()), "componentType": 5126, "count": len(v), "type": "VEC3"},
            {"bufferView": add(idx.tobytes()), "componentType": 5123, "count": len(idx), "type": "SCALAR"},
            {"bufferView": add(t.tobytes()), "componentType": 5126, "count": len(t), "type": "VEC2"}]
    g = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
         "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 2}, "indices": 1, "material": 0}]}],
         "accessors": accs}
    if png_bytes is not None:
        img = {"uri": "data:image/png;base64," + base64.b64encode(png_bytes).decode()} if data_uri else \
              {"bufferView": add(png_bytes), "mimeType": "image/png"}
        g.update(images=[img], textures=[{"source": 0}],
                 materials=[{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "baseColorFactor": [0.1, 0.2, 0.3, 1.0]}}])
    else:
        g.update(materials=[{"pbrMetallicRoughness": {"baseColorFactor": list(factor)}}])
    g["bufferViews"] = views
    g["buffers"] = [{"byteLength": sum(map(len, blobs))}]
    js = json.dumps(g).encode()
    js += b" " * (-len(js) % 4)
    binc = b"".join(blobs)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, 12 + 8 + len(js) + 8 + len(binc)))
        f.write(struct.pack("<I4s", len(js), b"JSON") + js)
        f.write(struct.pack("<I4s", len(binc), b"BIN\x00") + binc)


def test_glb_texture_is_baked_to_vertex_colours(tmp_path):
    """TextureVisuals.to_color() of glb2point (utils/dataUtils.py:224-225): nearest texel at
    x = u (W-1), y = v (H-1) of the file's TEXCOORD, GL_REPEAT wrap; the texture wins over the factor."""
    import io
    from PIL import Image
    from genpc_amd.utils import mesh_io as M
    rng = np.random.default_rng(0)
    tex = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)               # H = 5, W = 7
    buf = io.BytesIO()
    Image.fromarray(tex, "RGB").save(buf, format="PNG")
    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], float)
    faces = np.array([[0, 1, 2], [1, 3, 2]])
    uv = np.array([[0.0, 0.0], [1.0, 0.0], [2.0 / 6.0, 1.0], [1.0 + 3.0 / 6.0, 0.5]])     # last one wraps in u
    expect = np.stack([tex[0, 0], tex[0, 6], tex[4, 2], tex[2, (6 + 3) % 7]]) / 255.0
    for data_uri in (False, True):
        p = str(tmp_path / ("tex%d.glb" % data_uri))
        _write_textured_glb(p, verts, faces, uv, buf.getvalue(), data_uri=data_uri)
        V, F, C = M.load_glb(p)
        np.testing.assert_allclose(C, expect, atol=1e-12)
    np.testing.assert_array_equal(M.uv_to_color(np.array([[0.49 / 6, 0.51 / 4], [0.5 / 6, 1.5 / 4]]), np.dstack([tex, tex[:, :, :1]]))[:, :3],
                                  np.stack([tex[1, 0], tex[2, 0]]))             # rounding: 0.49 -> 0, 0.51 -> 1; halves to even
    # sampled colours are barycentric blends of the baked vertex colours
    P, Cc = M.glb2point(str(tmp_path / "tex0.glb"), num_points=2000, rng=np.random.default_rng(1))
    lower = P[:, 0] + P[:, 1] <= 1.0
    b = np.stack([1 - P[lower, 0] - P[lower, 1], P[lower, 0], P[lower, 1]], 1)
    np.testing.assert_allclose(Cc[lower], b @ expect[:3], atol=1e-9)
    # a material with only a factor
    p = str(tmp_path / "factor.glb")
    _write_textured_glb(p, verts, faces, uv, None, factor=[0.25, 0.5, 1.0, 1.0])
    np.testing.assert_allclose(M.load_glb(p)[2], np.tile(np.round(np.array([0.25, 0.5, 1.0]) * 255) / 255, (4, 1)))


def np_voxel_down_sample(xyz, voxel):
    """open3d's published definition in numpy: double index arithmetic, per-voxel mean summed in
    point order (np.add.at is sequential), ascending (i, j, k)."""
    anchor = xyz.min(0).astype(np.float64) - voxel * 0.5
    ijk = np.floor((xyz.astype(np.float64) - anchor) / voxel).astype(np.int64)
    key = (ijk[:, 0] << 42) | (ijk[:, 1] << 21) | ijk[:, 2]
    uniq, inv = np.unique(key, return_inverse=True)
    s = np.zeros((uniq.shape[0], 3), np.float64)
    np.add.at(s, inv, xyz.astype(np.float64))
    cnt = np.bincount(inv, minlength=uniq.shape[0])
    return (s / cnt[:, None]).astype(np.float32)


def test_oracle_voxel_down_sample_matches_numpy(oracle):
    rng = np.random.default_rng(2)
    xyz = (rng.random((5000, 3), dtype=np.float32) - np.float32(0.5)) * np.float32([1.0, 0.6, 0.3])
    xyz[100:140] = xyz[0:40]                      # repeated points
    for voxel in (np.float32(0.02), np.float32(0.03), np.float32(0.2), np.float32(5.0)):
        got = oracle.voxel_down_sample(xyz, voxel)
        exp = np_voxel_down_sample(xyz, np.float64(voxel))
        np.testing.assert_array_equal(got, exp)
    assert oracle.voxel_down_sample(xyz, 5.0).shape == (1, 3)
    with pytest.raises(ValueError):
        bad = xyz.copy()
        bad[3, 1] = np.nan
        oracle.voxel_down_sample(bad, 0.03)
