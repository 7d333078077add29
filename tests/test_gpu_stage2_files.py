"""Stage 2 through the reference's FILE interfaces (main.py's call sites):
a synthetic flag directory -- point_uv.npy, img.png, a coloured GLB -- goes through
ScaleAdapter.colorPoint(flag, xyz, gt, rgb, img_resource), ScaleAdapter.scaleReg(flag) =
reg(cfg, flag, 0.5, True, True) and object_pose_optimization(glb_path, point_path, ...), and the
values agree with the tensor forms."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import write_glb

pytestmark = pytest.mark.gpu


def _ellipsoid_mesh(nu=48, nv=24, radii=(0.45, 0.3, 0.2)):
    u = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = np.linspace(0.05, np.pi - 0.05, nv)
    uu, vv = np.meshgrid(u, v)
    verts = np.stack([radii[0] * np.cos(uu) * np.sin(vv), radii[1] * np.cos(vv), radii[2] * np.sin(uu) * np.sin(vv)], -1).reshape(-1, 3)
    verts[:, 0] += 0.12 * (verts[:, 1] > 0.1)            # a bump: no symmetry for the pose to slide on
    faces = []
    for j in range(nv - 1):
        for i in range(nu):
            a, b = j * nu + i, j * nu + (i + 1) % nu
            faces += [[a, b, a + nu], [b, b + nu, a + nu]]
    cols = 0.2 + 0.8 * (verts - verts.min(0)) / (verts.max(0) - verts.min(0))
    return verts, np.array(faces), cols


def test_flag_directory_round_trip(tmp_path):
    import torch
    from PIL import Image
    from genpc_amd.ScaleAdapter import ScaleAdapter
    from genpc_amd import reg_xyz
    from genpc_amd.optim_registration import diff_obj_pose as POSE
    from genpc_amd.utils import dataUtils as D, mesh_io as M
    flag = "00042"
    base = tmp_path / flag
    os.makedirs(base)
    cfg = SimpleNamespace(output_path=str(tmp_path), device="cuda", generative_model="trellis", dataset="redwood")
    verts, faces, cols = _ellipsoid_mesh()
    glb = str(base / (flag + "_trellis.glb"))
    write_glb(glb, verts, faces, cols, indices_u16=True)
    rng = np.random.default_rng(3)
    # the observed partial cloud: the mesh's front half, at 0.88 scale and slightly moved
    pts, pcols = M.glb2point(glb, num_points=6000, rng=np.random.default_rng(5))
    front = pts[:, 2] > -0.02
    partial = (pts[front] * 0.88 + np.array([0.015, -0.01, 0.01])).astype(np.float32)
    n = len(partial)
    # ---- colorPoint(flag, xyz, gt, rgb, 'depth'): colours from img.png through point_uv.npy
    uv = rng.random((n, 2)).astype(np.float32)
    img = rng.integers(0, 256, (1024, 1024, 3), dtype=np.uint8)
    np.save(base / "point_uv.npy", uv)
    Image.fromarray(img).save(base / "img.png")
    sa = ScaleAdapter(cfg)
    xyz = torch.from_numpy(partial).cuda()
    got = sa.colorPoint(flag, xyz, xyz, None, img_resource="depth")
    pix = np.clip((uv * 1024).astype(np.int64), 0, 1023)                       # (u, v) -> (row = v, col = u), image flipped
    want = img[::-1][pix[:, 1], pix[:, 0]].astype(np.float32) / 255.0
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-7)
    x2, c2 = D.read_ply(str(base / "color_point.ply"))
    np.testing.assert_allclose(x2, partial.astype(np.float64))
    np.testing.assert_allclose(c2, want, atol=0.5 / 255 + 1e-6)
    # give the partial cloud the mesh's own colours for the alignment ('obj' source, :49-51)
    sa.colorPoint(flag, xyz, xyz, torch.from_numpy(pcols[front].astype(np.float32)), img_resource="obj")
    # ---- object_pose_optimization(glb_path, point_path, ...): file form = tensor form on the loaded clouds
    ply = str(base / "color_point.ply")
    pv, pc = POSE.load_point_cloud(ply, torch.device("cuda"), radius=0.02, num_points=8000)
    assert pv.is_cuda and pv.shape == pc.shape and pv.shape[0] < n and 0.0 <= float(pc.min()) and float(pc.max()) <= 1.0
    np.random.seed(0)
    cwd = os.getcwd()
    os.chdir(str(tmp_path))          # the reference drops its side-effect files into the working directory
    try:
        T_file = POSE.object_pose_optimization(glb, ply, radius=0.02, lr=0.01, iters=40, render_size=224, device=torch.device("cuda"))
    finally:
        os.chdir(cwd)
    assert T_file.shape == (4, 4) and T_file[3].tolist() == [0, 0, 0, 1]
    # diff_obj_pose.py:508-509,591: partial.png, partial_mask.png, final_transform.npy
    from PIL import Image
    ref = np.asarray(Image.open(str(tmp_path / "partial.png")))
    msk = np.asarray(Image.open(str(tmp_path / "partial_mask.png")))
    assert ref.shape == (224, 224, 3) and msk.shape == (224, 224) and set(np.unique(msk)) <= {0, 255} and msk.any()
    assert ((ref.max(-1) > 0) >= (msk > 0)).all()      # the mask marks rendered pixels only
    np.testing.assert_array_equal(np.load(str(tmp_path / "final_transform.npy")), T_file)
    s = np.cbrt(np.linalg.det(T_file[:3, :3].astype(np.float64)))
    assert 0.75 < s < 0.95
    # ---- scaleReg(flag) = reg(cfg, flag, 0.5, True, True): writes {flag}_fused.ply with colours
    out = sa.scaleReg(flag, rng=np.random.default_rng(7), cd_only_pose=False)
    fused_path = str(base / (flag + "_fused.ply"))
    assert out["fused_path"] == fused_path and os.path.exists(fused_path)
    fx, fc = D.read_ply(fused_path)
    assert fc is not None and fx.shape == fc.shape and 10000 < len(fx) <= 20000
    np.testing.assert_allclose(fx, out["fused"].double().cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(fc, out["fused_col"].cpu().numpy(), atol=0.5 / 255 + 1e-6)
    # every fused point carries the colour of the input point it came from
    src = np.concatenate([out["source"].cpu().numpy(), out["target"].cpu().numpy()])
    srcc = np.concatenate([out["source_col"].cpu().numpy(), out["target_col"].cpu().numpy()])
    from scipy.spatial import cKDTree
    d, j = cKDTree(src).query(out["fused"].cpu().numpy())
    assert d.max() < 1e-6
    np.testing.assert_allclose(out["fused_col"].cpu().numpy(), srcc[j], atol=1e-6)
    # the aligned complete cloud explains the partial one
    d2, _ = cKDTree(out["target"].cpu().numpy()).query(out["source"].cpu().numpy())
    assert np.mean(d2) < 0.02, np.mean(d2)
    # tensor form of reg gives the same kind of answer without files
    out_t = reg_xyz.reg(xyz, torch.from_numpy(pts.astype(np.float32)).cuda(), cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True)
    assert set(("source", "target", "diff_transform", "coarse_transformation")) <= set(out_t)
