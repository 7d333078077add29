"""Pins the pose / projection half of the oracle (oracle/genpc_oracle_geom.c).

The reference delegates this arithmetic to torch autograd, torch.optim.Adam and
pytorch3d's rotation_6d_to_matrix (optim_registration/diff_obj_pose.py:408-423,
286-336, 524-547).  torch is importable here, so the oracle's analytic gradient
and its Adam are checked against torch itself on the reference's own expression
of the loss; pytorch3d is absent, so rotation_6d_to_matrix is restated in torch
from its published definition.  Camera conventions come from kaolin (absent):
those tests are properties (orthonormality, look-at geometry, uv range)."""
import math

import numpy as np
import pytest
import torch

from conftest import gen_pair


def t_rot6d(d6):
    a1, a2 = d6[:3], d6[3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = torch.nn.functional.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def t_loss(params, v, center, partial, i1, i2):
    """compute_loss_function's CD term + rot_reg, written like the reference."""
    R = t_rot6d(params[:6])
    scale = torch.exp(params[9:10])[0]
    local = (v - center) * scale
    local = (R @ local.T).T
    pts = local + center + params[6:9]
    d1 = ((pts - partial[i1]) ** 2).sum(-1)          # chamfer_partial_l1(result, ref): dist1
    d2 = ((partial - pts[i2]) ** 2).sum(-1)          # chamfer_partial_l1(ref, result): dist1 of the swapped call
    cd = torch.mean(torch.sqrt(d1)) + 0.5 * torch.mean(torch.sqrt(d2))
    ortho = torch.norm(R @ R.T - torch.eye(3, dtype=v.dtype))
    return cd * 3.0 + 0.001 * ortho, cd, ortho, pts


def make_case(seed, nc=700, npart=500):
    rng = np.random.default_rng(seed)
    v = (rng.random((nc, 3), dtype=np.float32) - 0.5).astype(np.float32)
    partial = (rng.random((npart, 3), dtype=np.float32) - 0.5).astype(np.float32) * 0.8
    params = np.array([0.9, 0.1, -0.3, 0.05, 1.1, 0.2, 0.02, -0.01, 0.03, math.log(0.8)], np.float32)
    return v, partial, params


def test_rot6d_matches_definition(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        d6 = rng.standard_normal(6).astype(np.float32)
        R = oracle.rot6d_to_matrix(d6)
        np.testing.assert_allclose(R, t_rot6d(torch.from_numpy(d6)).numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=3e-6)
        assert abs(np.linalg.det(R.astype(np.float64)) - 1) < 1e-5


def test_pose_transform_matches_torch(oracle):
    v, partial, params = make_case(1)
    c = v.mean(0)
    pts = oracle.pose_transform(v, c, params)
    _, _, _, tp = t_loss(torch.from_numpy(params), torch.from_numpy(v), torch.from_numpy(c), torch.from_numpy(partial),
                         torch.zeros(len(v), dtype=torch.long), torch.zeros(len(partial), dtype=torch.long))
    np.testing.assert_allclose(pts, tp.numpy(), rtol=1e-5, atol=1e-6)


def test_loss_and_gradient_match_torch_autograd(oracle):
    for seed in (2, 3):
        v, partial, params = make_case(seed)
        c = v.mean(0)
        pts = oracle.pose_transform(v, c, params)
        d1, d2, i1, i2 = oracle.chamfer_forward(pts[None], partial[None], 1)
        lo, g = oracle.pose_loss_grad(v, c, params, partial, d1[0], i1[0], d2[0], i2[0])
        P = torch.from_numpy(params).double().requires_grad_(True)
        loss, cd, ortho, _ = t_loss(P, torch.from_numpy(v).double(), torch.from_numpy(c).double(),
                                    torch.from_numpy(partial).double(), torch.from_numpy(i1[0]).long(),
                                    torch.from_numpy(i2[0]).long())
        loss.backward()
        assert abs(float(loss) - lo[0]) < 2e-6 and abs(float(cd) - lo[1]) < 1e-6
        np.testing.assert_allclose(g, P.grad.numpy(), rtol=2e-4, atol=2e-6)


def test_adam_matches_torch(oracle):
    rng = np.random.default_rng(5)
    p0 = rng.standard_normal(10).astype(np.float32)
    params = p0.copy()
    m = np.zeros(10, np.float32)
    v = np.zeros(10, np.float32)
    tp = [torch.nn.Parameter(torch.from_numpy(p0[:6].copy())), torch.nn.Parameter(torch.from_numpy(p0[6:9].copy())),
          torch.nn.Parameter(torch.from_numpy(p0[9:].copy()))]
    lr = 0.01
    opt = torch.optim.Adam([{"params": [tp[0]], "lr": lr}, {"params": [tp[1]], "lr": lr * 0.2},
                            {"params": [tp[2]], "lr": lr * 0.1}])
    for step in range(1, 30):
        g = rng.standard_normal(10).astype(np.float32) * 0.1
        oracle.adam_step(params, g, m, v, step, lr)
        opt.zero_grad()
        tp[0].grad = torch.from_numpy(g[:6].copy())
        tp[1].grad = torch.from_numpy(g[6:9].copy())
        tp[2].grad = torch.from_numpy(g[9:].copy())
        opt.step()
        ref = np.concatenate([t.detach().numpy() for t in tp])
        np.testing.assert_allclose(params, ref, rtol=2e-6, atol=2e-7)


def test_pose_loop_recovers_known_similarity(oracle):
    """A cloud against a rotated/scaled/shifted copy of a subset of itself: the CD-only
    loop must bring the loss down by an order of magnitude and land near the truth."""
    rng = np.random.default_rng(9)
    n = 1200
    u = rng.standard_normal((n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    complete = (u * np.array([0.5, 0.3, 0.2])).astype(np.float32)        # ellipsoid surface ...
    complete[:200] += np.float32([0.15, 0.1, 0.0]) * np.abs(u[:200, :1]).astype(np.float32)   # ... with a bump
    th = math.radians(12.0)
    Rt = np.array([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]])
    c = complete.mean(0)
    full = ((complete - c) * 0.9) @ Rt.T + c + np.array([0.02, -0.01, 0.015])
    partial = full[full[:, 2] > -0.05][:600].astype(np.float32)
    T, hist, bp = oracle.pose_optimize_cd(complete, partial, lr=0.01, iters=200, starts=4)
    assert hist.shape == (4, 201)
    best = int(np.argmin(hist.min(axis=1)))
    assert best == 0                                   # the 0-degree start is the right basin
    assert hist[best, -1] < 0.5 * hist[best, 0]
    s = np.cbrt(np.linalg.det(T[:3, :3].astype(np.float64)))
    assert abs(s - 0.9) < 0.03
    np.testing.assert_allclose(T[:3, :3] / s, Rt, atol=0.03)
    np.testing.assert_allclose(T[:3, 3], [0.02, -0.01, 0.015], atol=0.01)


def test_camera_properties(oracle):
    eyes = oracle.fibonacci_sphere(64, 1.6)
    assert np.allclose(np.linalg.norm(eyes, axis=1), 1.6)
    for eye in eyes[::7]:
        up = oracle.calculate_up_vector(eye, np.zeros(3))
        V = oracle.look_at(eye, np.zeros(3), up).reshape(3, 4)
        np.testing.assert_allclose(V[:, :3] @ V[:, :3].T, np.eye(3), atol=2e-6)
        cam_origin = V[:, :3] @ np.zeros(3) + V[:, 3]
        # the look-at target sits on the -Z axis at the eye distance
        np.testing.assert_allclose(cam_origin, [0, 0, -1.6], atol=1e-5)
        np.testing.assert_allclose(V[:, :3] @ eye + V[:, 3], 0, atol=1e-5)
    # straight down the Y axis: the degenerate branch of calculate_up_vector
    np.testing.assert_array_equal(oracle.calculate_up_vector(np.array([0, 1.6, 0.0]), np.zeros(3)), [0, 0, 1])


def test_get_uvs_properties(oracle):
    a, _ = gen_pair(3, (1, 2000, 3), (1, 1, 3))
    xyz = a[0] * 0.8
    eyes = oracle.fibonacci_sphere(8, 1.6)
    views = np.stack([oracle.look_at(e, np.zeros(3), oracle.calculate_up_vector(e, np.zeros(3))) for e in eyes])
    focal = 1.0 / math.tan(math.radians(49.1) / 2)
    uv, depth, tr, bb = oracle.get_uvs(views, focal, xyz, rescale=True, padding=0.15)
    assert uv.shape == (8, 2000, 2) and depth.shape == (8, 2000)
    assert uv.min() >= 0.15 - 1e-6 and uv.max() <= 0.85 + 1e-6
    # the longer bbox side spans exactly [0.15, 0.85]
    span = uv.max(axis=1) - uv.min(axis=1)
    np.testing.assert_allclose(span.max(axis=1), 0.7, atol=1e-5)
    # depth is monotone in distance from the eye
    for i in range(8):
        dist = np.linalg.norm(xyz - eyes[i], axis=1)
        zc = -(views[i].reshape(3, 4)[2, :3] @ xyz.T + views[i].reshape(3, 4)[2, 3])
        order = np.argsort(zc)
        assert np.all(np.diff(depth[i][order]) >= -1e-6)
        assert dist.min() > 0.5
    uv2, _, _, _ = oracle.get_uvs(views, focal, xyz, rescale=False)
    np.testing.assert_allclose(uv2, (tr[..., :2] + 1) * 0.5, atol=1e-7)


def test_paint_and_gather(oracle):
    res = 16
    pix = np.array([[2, 3], [2, 3], [15, 15], [0, 0]], np.int32)
    col = np.array([[.1, .2, .3], [.4, .5, .6], [.7, .8, .9], [1, 1, 1]], np.float32)
    out, img = oracle.paint_pixels(res, pix, col, 1)
    np.testing.assert_array_equal(img[:, 2, 3], col[1])            # last writer wins
    np.testing.assert_array_equal(out[:, res - 1 - 2, 3], col[1])  # vertical flip
    out2, img2 = oracle.paint_pixels(res, pix, col, 2)             # 3x3 stamp, clipped at the border
    assert (img2[0] != 0).sum() == 9 + 4 + 4
    got = oracle.gather_colors(pix, out)                           # colorPoint reads the flipped image
    np.testing.assert_array_equal(got[1], col[1])
    uv = np.array([[0.2, 0.9], [1.5, -0.2]], np.float32)
    np.testing.assert_array_equal(oracle.uv_to_pixels(uv, 256), [[230, 51], [0, 255]])
