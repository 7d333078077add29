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


def test_oracle_loop_stops_after_300_steps_without_improvement(oracle):
    """diff_obj_pose.py:529-556: patience 300, counted after the optimizer step.  With lr = 0 nothing improves after the
    first iteration: iterations 0 .. 301 run, the rest are NaN in the history, every start alike; with a live learning rate
    and iters <= 300 nothing is cut."""
    v, partial, _ = make_case(3, nc=120, npart=90)
    T, hist, bp = oracle.pose_optimize_cd(v, partial, lr=0.0, iters=330, starts=2)
    assert hist.shape == (2, 331)
    assert np.isfinite(hist[:, :302]).all() and np.isnan(hist[:, 302:]).all()
    assert (hist[:, :302] == hist[:, :1]).all()
    T2, h2, _ = oracle.pose_optimize_cd(v, partial, lr=0.01, iters=60, starts=2)
    assert np.isfinite(h2).all()


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


# ---------------------------------------------------------------------------
# silhouette ("mask") half of the loss
def t_splat(pts, radius, S, col=None):
    """The build's own colour splat, written densely in torch ([P pixels] x [N points]) -> [S,S,3]:
    occupancy O = 1 - prod(1 - a), colour = coverage-weighted mean of the point colours, I = O * colour."""
    zv = 3.0 - pts[:, 2]
    ok = (zv > 1e-4) & (zv < 5.0)
    hs = 0.5 * S
    u = hs * (1.0 + 4.0 * pts[:, 0] / zv)
    v = hs * (1.0 - 4.0 * pts[:, 1] / zv)
    rho = hs * 4.0 * radius / zv
    rr, cc = torch.meshgrid(torch.arange(S, dtype=pts.dtype) + 0.5, torch.arange(S, dtype=pts.dtype) + 0.5, indexing="ij")
    dx = cc.reshape(-1, 1) - u[None]
    dy = rr.reshape(-1, 1) - v[None]
    a = 1.0 - (dx * dx + dy * dy) / (rho * rho)[None]
    a = torch.clamp(a, min=0.0, max=0.999) * ok[None]
    occ = 1.0 - torch.prod(1.0 - a, dim=1)
    if col is None:
        col = torch.ones(pts.shape[0], 3, dtype=pts.dtype)
    den = a.sum(1)
    num = a @ col
    colour = torch.where(den[:, None] > 0, num / torch.clamp(den, min=1e-300)[:, None], torch.zeros_like(num))
    return (occ[:, None] * colour).reshape(S, S, 3)


def t_splat_pulsar(pts, radius, S, col=None, falloff_linear=False, depth_hit=False):
    """Pulsar's published blending function (Lassner & Zollhoefer 2021, eq. 1-2) with the reference's arguments (gamma 1e-2,
    znear 1e-4, zfar 5, bg 0, opacity 1), written densely in torch ([P pixels] x [N points]) -> [S,S,3]; same camera and
    footprint as t_splat.  I = sum a e c / (B + sum a e), e = exp(z / gamma), z = (zfar - Zv) / (zfar - znear), B = exp(eps / gamma)."""
    gamma, znear, zfar, eps = 1e-2, 1e-4, 5.0, 1e-10
    zv = 3.0 - pts[:, 2]
    ok = (zv > 1e-4) & (zv < 5.0)
    hs = 0.5 * S
    u = hs * (1.0 + 4.0 * pts[:, 0] / zv)
    v = hs * (1.0 - 4.0 * pts[:, 1] / zv)
    rho = hs * 4.0 * radius / zv
    rr, cc = torch.meshgrid(torch.arange(S, dtype=pts.dtype) + 0.5, torch.arange(S, dtype=pts.dtype) + 0.5, indexing="ij")
    dx = cc.reshape(-1, 1) - u[None]
    dy = rr.reshape(-1, 1) - v[None]
    sq = (dx * dx + dy * dy) / (rho * rho)[None]
    a = 1.0 - (torch.sqrt(sq) if falloff_linear else sq)
    a = torch.clamp(a, min=0.0, max=0.999) * ok[None]
    if depth_hit:      # the ray-sphere hit (orthographic inside the disc) instead of the sphere's centre
        zh = zv[None] - radius * torch.sqrt(torch.clamp(1.0 - sq, min=0.0))
        ze = (zfar - zh) / (zfar - znear) / gamma
    else:
        ze = ((zfar - zv) / (zfar - znear) / gamma)[None].expand_as(a)
    live = a > 0
    m = torch.where(live, ze, torch.full_like(ze, eps / gamma)).max(dim=1).values.clamp(min=eps / gamma).detach()
    w = torch.where(live, a * torch.exp(torch.where(live, ze, m[:, None]) - m[:, None]), torch.zeros_like(a))
    if col is None:
        col = torch.ones(pts.shape[0], 3, dtype=pts.dtype)
    den = math.exp(eps / gamma) * torch.exp(-m) + w.sum(1)
    return ((w @ col) / den[:, None]).reshape(S, S, 3)


def t_mask_loss(img, ref):
    """compute_loss_function's mask terms as the reference writes them (diff_obj_pose.py:204-217,
    261-278,238-259,304-311) on [S,S,3] images.  (tests/test_reference_vectors.py pins the oracle to the
    reference's code itself; this torch version exists so that autograd can run through splat + loss.)"""
    F = torch.nn.functional
    ref_mean = torch.mean(ref, dim=(0, 1), keepdim=True)
    ref_std = torch.std(ref, dim=(0, 1), keepdim=True) + 1e-6
    res_mean = torch.mean(img, dim=(0, 1), keepdim=True)
    res_std = torch.std(img, dim=(0, 1), keepdim=True) + 1e-6
    norm = torch.clamp((img - res_mean) / res_std * ref_std + ref_mean, 0.0, 1.0)

    def soft(x):
        lum = 0.299 * x[:, :, 0] + 0.587 * x[:, :, 1] + 0.114 * x[:, :, 2]
        return torch.sigmoid((lum - 0.1) / 0.05)
    m, mr = soft(norm), soft(ref)
    loss = F.mse_loss(m, mr) * 30 + F.binary_cross_entropy(m, mr)
    inter = (m.reshape(-1) * mr.reshape(-1)).sum()
    dice = 1 - (2.0 * inter + 1e-6) / (m.sum() + mr.sum() + 1e-6)
    return loss * 1 + dice * 10


def _colours(rng, n, dark=0.0):
    col = (0.25 + 0.75 * rng.random((n, 3))).astype(np.float32)
    k = int(n * dark)
    if k:
        col[:k] *= np.float32(0.08)
    return col


def test_splat_and_mask_loss_match_torch(oracle):
    rng = np.random.default_rng(3)
    S = 40
    pts = ((rng.random((60, 3)) - 0.5) * 0.9).astype(np.float32)
    ref_pts = ((rng.random((400, 3)) - 0.5) * 0.9).astype(np.float32)
    col, ref_col = _colours(rng, 60, 0.3), _colours(rng, 400)
    for c, rc in ((None, None), (col, ref_col)):
        img = oracle.splat_image(pts, 0.025, S, c)
        ref = oracle.splat_image(ref_pts, 0.04, S, rc)
        ti = t_splat(torch.from_numpy(pts).double(), 0.025, S, None if c is None else torch.from_numpy(c).double())
        tr = t_splat(torch.from_numpy(ref_pts).double(), 0.04, S, None if rc is None else torch.from_numpy(rc).double())
        assert img.shape == (S, S, 3)
        np.testing.assert_allclose(img, ti.numpy(), atol=2e-6)
        assert 0.005 < float(img.mean()) < 0.9 and float(ref.max()) > 0.9
        if c is None:
            assert np.array_equal(img[..., 0], img[..., 1]) and np.array_equal(img[..., 0], img[..., 2])
        # the loss on float32 images, as the reference computes it: sigmoid saturates to exactly 1.0f
        # and binary_cross_entropy's -100 clamp decides those pixels (a float64 evaluation differs by
        # tens of percent here -- checked below so that the test input really exercises the clamp)
        l32 = float(t_mask_loss(ti.float(), tr.float()))
        l64 = float(t_mask_loss(ti, tr))
        assert abs(l32 - l64) > 0.01 * l64
        np.testing.assert_allclose(oracle.mask_loss(img, ref), l32, rtol=2e-4)


@pytest.fixture
def pulsar_blend(oracle):
    prev = oracle.set_blend(1)
    yield
    oracle.set_blend(prev)


def test_pulsar_blend_matches_torch(oracle, pulsar_blend):
    """The oracle's restatement of Pulsar's blending function against the dense torch evaluation of the same formulas;
    and what makes it Pulsar's: a near surface hides a far one (the coverage splat averages them)."""
    rng = np.random.default_rng(3)
    S = 40
    pts = ((rng.random((60, 3)) - 0.5) * 0.9).astype(np.float32)
    col = _colours(rng, 60, 0.3)
    for c in (None, col):
        img = oracle.splat_image(pts, 0.03, S, c)
        ti = t_splat_pulsar(torch.from_numpy(pts).double(), 0.03, S, None if c is None else torch.from_numpy(c).double())
        np.testing.assert_allclose(img, ti.numpy(), atol=2e-6)
        assert 0.005 < float(img.mean()) < 0.9
    # two coincident discs, red in front (z = +0.3) of green (z = -0.3): Pulsar's pixel is red, the coverage splat's a mix
    two = np.array([[0.0, 0.0, 0.3], [0.0, 0.0, -0.3]], np.float32)
    tc = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    centre = oracle.splat_image(two, 0.05, S, tc)[S // 2, S // 2]
    assert centre[0] > 0.99 and centre[1] < 1e-4          # (0.6 in depth = e^-12 in weight)
    oracle.set_blend(0)
    mix = oracle.splat_image(two, 0.05, S, tc)[S // 2, S // 2]
    oracle.set_blend(1)
    assert 0.2 < mix[0] < 0.8 and 0.2 < mix[1] < 0.8


@pytest.mark.parametrize("coloured", [False, True])
def test_pulsar_full_loss_gradient_matches_torch_autograd(oracle, pulsar_blend, coloured):
    """As test_full_loss_gradient_matches_torch_autograd with both images drawn by Pulsar's blending function: the
    gradient now also flows through the depth of every point (the softmax weights)."""
    S = 36
    radius = 0.04
    v, partial, params = make_case(5, nc=260, npart=170)
    rng = np.random.default_rng(50)
    vcol = _colours(rng, len(v), 0.33) if coloured else None
    pcol = _colours(rng, len(partial)) if coloured else None
    center = v.astype(np.float64).mean(0).astype(np.float32)
    pts = oracle.pose_transform(v, center, params)
    d1, d2, i1, i2 = oracle.chamfer_forward(pts[None], partial[None], 0)
    ref = oracle.splat_image(partial, radius, S, pcol)
    lo, g = oracle.pose_full_loss_grad(v, center, params, partial, d1[0], i1[0], d2[0], i2[0], radius, S, ref, vert_col=vcol)
    P = torch.tensor(params.astype(np.float64), requires_grad=True)
    tv, tc, tp = (torch.from_numpy(x.astype(np.float64)) for x in (v, center, partial))
    cd_total, cd, ortho, tpts = t_loss(P, tv, tc, tp, torch.from_numpy(i1[0].astype(np.int64)),
                                       torch.from_numpy(i2[0].astype(np.int64)))
    timg = t_splat_pulsar(tpts, 1.1 * radius, S, None if vcol is None else torch.from_numpy(vcol).double())
    ml = t_mask_loss(timg.float(), torch.from_numpy(ref)).double()
    total = cd_total + ml
    total.backward()
    assert abs(lo[3] - float(ml)) < 2e-4 * max(1.0, abs(float(ml)))
    tg = P.grad.numpy()
    P2 = torch.tensor(params.astype(np.float64), requires_grad=True)
    t_loss(P2, tv, tc, tp, torch.from_numpy(i1[0].astype(np.int64)), torch.from_numpy(i2[0].astype(np.int64)))[0].backward()
    if coloured:       # (a white cloud's Pulsar image is flat inside the silhouette: little gradient beside the rim's)
        assert np.abs(tg - P2.grad.numpy()).max() > 0.02 * np.abs(tg).max()
    np.testing.assert_allclose(g, tg, rtol=2e-3, atol=2e-3 * np.abs(tg).max())
    # and it is not the coverage splat's gradient
    oracle.set_blend(0)
    ref0 = oracle.splat_image(partial, radius, S, pcol)
    lo0, g0 = oracle.pose_full_loss_grad(v, center, params, partial, d1[0], i1[0], d2[0], i2[0], radius, S, ref0, vert_col=vcol)
    oracle.set_blend(1)
    assert abs(lo0[3] - lo[3]) > 1e-3


@pytest.mark.parametrize("variant", [(1, 0), (0, 1), (1, 1)])
def test_renderer_variants_match_torch(oracle, pulsar_blend, variant):
    """The switchable from-memory choices of the renderer (oracle only: linear fall-off, ray-sphere hit depth; DESIGN.md
    section 2, tools/renderer_sensitivity.py): image and full-objective gradient against the dense torch evaluation."""
    fl, dh = variant
    S, radius = 36, 0.04
    v, partial, params = make_case(5, nc=260, npart=170)
    rng = np.random.default_rng(51)
    vcol, pcol = _colours(rng, len(v), 0.33), _colours(rng, len(partial))
    center = v.astype(np.float64).mean(0).astype(np.float32)
    oracle.set_render_variant(fl, dh)
    try:
        pts = oracle.pose_transform(v, center, params)
        d1, d2, i1, i2 = oracle.chamfer_forward(pts[None], partial[None], 0)
        ref = oracle.splat_image(partial, radius, S, pcol)
        tref = t_splat_pulsar(torch.from_numpy(partial).double(), radius, S, torch.from_numpy(pcol).double(), bool(fl), bool(dh))
        np.testing.assert_allclose(ref, tref.numpy(), atol=3e-6)
        lo, g = oracle.pose_full_loss_grad(v, center, params, partial, d1[0], i1[0], d2[0], i2[0], radius, S, ref, vert_col=vcol)
    finally:
        oracle.set_render_variant(0, 0)
    assert np.abs(ref - oracle.splat_image(partial, radius, S, pcol)).max() > 1e-3       # (not the default renderer's image)
    P = torch.tensor(params.astype(np.float64), requires_grad=True)
    tv, tc, tp = (torch.from_numpy(x.astype(np.float64)) for x in (v, center, partial))
    cd_total, cd, ortho, tpts = t_loss(P, tv, tc, tp, torch.from_numpy(i1[0].astype(np.int64)), torch.from_numpy(i2[0].astype(np.int64)))
    timg = t_splat_pulsar(tpts, 1.1 * radius, S, torch.from_numpy(vcol).double(), bool(fl), bool(dh))
    ml = t_mask_loss(timg.float(), torch.from_numpy(ref)).double()
    (cd_total + ml).backward()
    assert abs(lo[3] - float(ml)) < 2e-4 * max(1.0, abs(float(ml)))
    tg = P.grad.numpy()
    np.testing.assert_allclose(g, tg, rtol=3e-3, atol=3e-3 * np.abs(tg).max())


@pytest.mark.parametrize("coloured", [False, True])
def test_full_loss_gradient_matches_torch_autograd(oracle, coloured):
    """mask_loss + 3 cd + 1e-3 ortho: the oracle's analytic gradient against autograd through the
    dense torch splat, the reference's normalisation / soft-mask / MSE+BCE+Dice code and the CD
    term.  The pose, the splat and the CD term run in float64; the image is cast to float32 before
    the reference's loss code, as the reference's images are (saturated soft masks carry no
    gradient there).  coloured: both clouds carry colours, a third of the complete cloud dark."""
    S = 36
    radius = 0.04
    v, partial, params = make_case(5, nc=260, npart=170)
    rng = np.random.default_rng(50)
    vcol = _colours(rng, len(v), 0.33) if coloured else None
    pcol = _colours(rng, len(partial)) if coloured else None
    center = v.astype(np.float64).mean(0).astype(np.float32)
    pts = oracle.pose_transform(v, center, params)
    d1, d2, i1, i2 = oracle.chamfer_forward(pts[None], partial[None], 0)
    ref = oracle.splat_image(partial, radius, S, pcol)
    lo, g = oracle.pose_full_loss_grad(v, center, params, partial, d1[0], i1[0], d2[0], i2[0], radius, S, ref, vert_col=vcol)
    P = torch.tensor(params.astype(np.float64), requires_grad=True)
    tv, tc, tp = (torch.from_numpy(x.astype(np.float64)) for x in (v, center, partial))
    cd_total, cd, ortho, tpts = t_loss(P, tv, tc, tp, torch.from_numpy(i1[0].astype(np.int64)),
                                       torch.from_numpy(i2[0].astype(np.int64)))
    timg = t_splat(tpts, 1.1 * radius, S, None if vcol is None else torch.from_numpy(vcol).double())
    ml = t_mask_loss(timg.float(), torch.from_numpy(ref)).double()
    total = cd_total + ml
    total.backward()
    assert abs(lo[3] - float(ml)) < 2e-4 * max(1.0, abs(float(ml)))
    assert abs(lo[0] - float(total)) < 2e-4 * max(1.0, abs(float(total)))
    tg = P.grad.numpy()
    # the mask term must matter in this case, otherwise the check proves nothing
    P2 = torch.tensor(params.astype(np.float64), requires_grad=True)
    t_loss(P2, tv, tc, tp, torch.from_numpy(i1[0].astype(np.int64)), torch.from_numpy(i2[0].astype(np.int64)))[0].backward()
    assert np.abs(tg - P2.grad.numpy()).max() > 0.05 * np.abs(tg).max()
    np.testing.assert_allclose(g, tg, rtol=2e-3, atol=2e-3 * np.abs(tg).max())
    if coloured:
        # and the colours must matter: the same geometry drawn white gives another loss and gradient
        lo_w, g_w = oracle.pose_full_loss_grad(v, center, params, partial, d1[0], i1[0], d2[0], i2[0], radius, S,
                                               oracle.splat_image(partial, radius, S), vert_col=None)
        assert abs(lo_w[3] - lo[3]) > 1e-2 and np.abs(g_w - g).max() > 1e-2 * np.abs(g).max()


def test_full_objective_loop_recovers_pose(oracle):
    """The multi-start loop with the mask term on a small asymmetric shape: finite, decreasing
    history; the mask term changes the trajectory; the scale stays in a sane range (the silhouette of
    a PARTIAL view pulls the scale below the Chamfer-only optimum: a property of the reference's
    objective, not of this restatement); colours change the trajectory."""
    rng = np.random.default_rng(4)
    n = 500
    complete = (rng.random((n, 3), dtype=np.float32) - np.float32(0.5)) * np.float32([0.9, 0.5, 0.3])
    complete[: n // 4, 0] += np.float32(0.25)
    c = complete.mean(0)
    full = ((complete - c) * 0.9) + c + np.array([0.02, -0.01, 0.015], np.float32)
    sel = full[:, 2] > -0.05
    partial = full[sel][: n // 2].astype(np.float32)
    T, hist, bp = oracle.pose_optimize(complete, partial, lr=0.01, iters=120, starts=2, radius=0.03, size=64)
    T0, hist0, _ = oracle.pose_optimize_cd(complete, partial, lr=0.01, iters=120, starts=2)
    assert np.isfinite(hist).all() and hist[0, -1] < hist[0, 0]
    assert np.abs(hist[0, :50] - hist0[0, :50]).max() > 1e-3
    s = np.cbrt(np.linalg.det(T[:3, :3].astype(np.float64)))
    assert 0.6 < s < 1.1
    ccol = _colours(rng, n, 0.4)
    pcol = ccol[sel][: n // 2]
    Tc, histc, _ = oracle.pose_optimize(complete, partial, lr=0.01, iters=40, starts=1, radius=0.03, size=64,
                                        complete_col=ccol, partial_col=pcol)
    assert np.isfinite(histc).all() and np.abs(histc[0, :40] - hist[0, :40]).max() > 1e-3
