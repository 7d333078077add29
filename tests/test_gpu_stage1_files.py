"""Stage 1 through the reference's FILE interface and on into stage 2 (VERDICT r3 missing #1): main.py:47-60 is
    dp.getImage(xyz, flag, rgb, depth_gen=True, img_gen=True);  sa.scaleAdapter(xyz, flag);  sa.scaleReg(flag)
over one directory ``{output_path}/{flag}/``.  The generator stages (depth inpainting, depth -> image, background
removal, image -> 3-D) are stock modules outside this library: stubs with the reference's call shapes stand in for
them; everything else -- viewpoint selection, projection, splat, masks, colour gather, registration, fusion -- is the
shipped code, chained through the files the reference writes."""
import os
import shutil
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import write_glb
from test_gpu_stage2_files import _ellipsoid_mesh

pytestmark = pytest.mark.gpu


def test_getimage_scaleadapter_scalereg_through_one_flag_directory(tmp_path):
    import torch
    from PIL import Image
    from genpc_amd.DepthPrompting import DepthPrompting
    from genpc_amd.ScaleAdapter import ScaleAdapter
    from genpc_amd.utils import dataUtils as D, mesh_io as M
    flag = "05117"
    cfg = SimpleNamespace(output_path=str(tmp_path), device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True,
                          point_size=1, mask_pixel_rate=3, view_num=64, distance=1.6, downsample_num=2000, removal_radius=10000,
                          generate_res=512, inpainter="flux", generative_model="trellis", dataset="redwood", rembg_model="rembg")
    verts, faces, cols = _ellipsoid_mesh()
    mesh_glb = str(tmp_path / "generated.glb")
    write_glb(mesh_glb, verts, faces, cols, indices_u16=True)
    pts, _ = M.glb2point(mesh_glb, num_points=9000, rng=np.random.default_rng(5))
    front = pts[:, 2] > -0.02
    partial = (pts[front] * 0.88 + np.array([0.015, -0.01, 0.01])).astype(np.float32)      # the observed scan
    xyz = torch.from_numpy(partial).cuda()
    calls = []

    def inpainter(raw_depth, mask):                       # (Flux fill / DDNM / cv2 in the reference)
        calls.append(("inpaint", tuple(raw_depth.shape), tuple(mask.shape)))
        filled = raw_depth.clone()
        filled[mask > 0] = float(raw_depth[raw_depth > 0].mean())
        return filled

    def depth2image(depth, category, size):               # (ControlNet / Flux / Qwen .generate)
        calls.append(("generate", depth.size, category, size))
        return depth.resize((1024, 1024)).convert("RGB")

    def rembg(src, dst):                                   # (RMBG / SAM)
        calls.append(("rembg", os.path.basename(src), os.path.basename(dst)))
        shutil.copy(src, dst)

    def generative(cfg_, flag_, img):                      # (InstantMesh / Trellis)
        calls.append(("generative", flag_, img.size))
        shutil.copy(mesh_glb, f"{cfg_.output_path}/{flag_}/{flag_}_{cfg_.generative_model}.glb")

    dp = DepthPrompting(cfg, inpainter=inpainter, depth2image=depth2image)
    out = dp.getImage(xyz, flag, rgb=None, depth_gen=True, img_gen=True)
    base = tmp_path / flag
    for f in ("raw_depth.png", "mask.png", "depth.png", "point_uv.npy", "viewpoint.npy", "camera.pth", "img.png"):
        assert (base / f).exists(), f
    # the files hold what the stage computed
    np.testing.assert_array_equal(np.load(base / "point_uv.npy"), out["uv"].cpu().numpy())
    np.testing.assert_allclose(np.load(base / "viewpoint.npy"), np.asarray(dp.view))
    raw = np.asarray(Image.open(base / "raw_depth.png"), np.float32)
    np.testing.assert_array_equal(raw, np.floor(out["raw_depth"].permute(1, 2, 0).cpu().numpy() * 255 + 0.5).clip(0, 255))
    m = np.asarray(Image.open(base / "mask.png"), np.float32) / 255
    np.testing.assert_array_equal(m, out["hole_mask1"].permute(1, 2, 0).cpu().numpy())
    cam = torch.load(base / "camera.pth", weights_only=False)
    assert tuple(cam["view"].shape) == (3, 4) and abs(cam["focal"] - dp.focal) < 1e-12
    # select first, project second: the chosen camera's rows equal the rows of projecting through ALL cameras
    uv_all, d_all, _ = dp.getUvs(dp.cameras, xyz, rescale=True, padding=cfg.padding, want_transformed=False)
    if not out["used_opposite"]:
        np.testing.assert_array_equal(out["uv"].cpu().numpy(), uv_all[out["view_index"]].cpu().numpy())
        np.testing.assert_array_equal(out["depth"].cpu().numpy(), d_all[out["view_index"]].cpu().numpy())
    assert 0.2 < float(out["visible"].float().mean()) <= 1.0 and float(out["hole_mask1"].sum()) > 0
    # ---- stage 2 on the same directory, main.py:56-60
    sa = ScaleAdapter(cfg, rembg=rembg, generative=generative)
    sa.scaleAdapter(xyz, flag)
    assert (base / "img_sam.png").exists() and (base / "color_point.ply").exists() and (base / f"{flag}_trellis.glb").exists()
    x2, c2 = D.read_ply(str(base / "color_point.ply"))
    np.testing.assert_allclose(x2, partial.astype(np.float64))
    img = np.asarray(Image.open(base / "img.png").convert("RGB"))
    uv = np.load(base / "point_uv.npy")
    pix = np.clip((uv * 1024).astype(np.int64), 0, 1023)
    want = img[::-1][pix[:, 1], pix[:, 0]].astype(np.float32) / 255.0
    np.testing.assert_allclose(c2, want, atol=0.5 / 255 + 1e-6)
    res = sa.scaleReg(flag, rng=np.random.default_rng(7))
    fx, fc = D.read_ply(str(base / f"{flag}_fused.ply"))
    assert fc is not None and 10000 < len(fx) <= 20000
    from scipy.spatial import cKDTree
    d2, _ = cKDTree(res["target"].cpu().numpy()).query(res["source"].cpu().numpy())
    assert np.mean(d2) < 0.02, np.mean(d2)                 # the aligned generated shape explains the scan
    assert [c[0] for c in calls] == ["inpaint", "generate", "rembg", "generative"]
    assert calls[1][2] == "chair" and calls[1][3] == 512   # getCategory(flag), cfg.generate_res
    # without the stock modules the stage says what is missing instead of guessing
    with pytest.raises(RuntimeError, match="inpainter"):
        DepthPrompting(cfg).getImage(xyz, "00001")
