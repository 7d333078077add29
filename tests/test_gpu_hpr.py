"""f3: exact hidden-point removal on the GPU (genpc_hpr_visibility, csrc/hpr.hip) against
  * oracle/genpc_oracle_hpr.c -- the same normal-cone clipping in C, same arithmetic and candidate order:
    the masks must be IDENTICAL, and
  * oracle/hpr.py -- the reference's operator through qhull itself (scipy), the library open3d calls:
    identical on the clouds below (random balls / shells, real scans, a lattice full of cospherical ties).
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hp():
    import torch
    from genpc_amd import _lib
    from oracle import oracle, hpr

    def run(points, eyes, radius):
        P = torch.from_numpy(np.ascontiguousarray(points, np.float32)).cuda()
        E = torch.from_numpy(np.ascontiguousarray(eyes, np.float64).reshape(-1, 3)).cuda()
        c, n = E.shape[0], P.shape[0]
        vis = torch.zeros(c, n, device="cuda", dtype=torch.uint8)
        cnt = torch.full((c,), -7, device="cuda", dtype=torch.int32)
        second = ctypes.c_int(-1)
        rc = _lib.lib.genpc_hpr_visibility(c, n, _lib.ptr(P), _lib.ptr(E), float(radius), _lib.ptr(vis), _lib.ptr(cnt),
                                           ctypes.addressof(second), None)
        torch.cuda.synchronize()
        assert rc == 1, _lib.last_error()
        return vis.cpu().numpy().astype(bool), cnt.cpu().numpy(), second.value

    def qhull(points, eyes, radius):
        out = np.zeros((len(eyes), len(points)), bool)
        for k, e in enumerate(eyes):
            out[k, hpr.hidden_point_removal(points, e, radius)] = True
        return out

    def clip(points, eyes, radius):
        return np.stack([oracle.hpr_visibility(points, e, radius) for e in eyes])
    return dict(run=run, qhull=qhull, clip=clip, lib=_lib)


EYES = np.array([[0, 0, 3.0], [2.0, 1.0, -1.5], [-1.1, 0.3, 0.9], [0.2, -1.6, 0.1]])


def clouds():
    rng = np.random.default_rng(0)
    out = {}
    for n in (1, 2, 3, 50, 700, 3001):
        out["ball%d" % n] = (rng.random((n, 3)) - 0.5).astype(np.float32)
        v = rng.normal(size=(n, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        out["sphere%d" % n] = (v * 0.5).astype(np.float32)
        out["shell%d" % n] = (v * (0.45 + 0.05 * rng.random((n, 1)))).astype(np.float32)
    return out


@pytest.mark.parametrize("radius", [0.3, 1.6, 3.0, 100.0, 800.0, 10000.0])      # (below ~1.5 the flipped points lie BEHIND the eye)
def test_hpr_equals_qhull_and_the_clipping_oracle(hp, radius):
    for name, P in clouds().items():
        vis, cnt, second = hp["run"](P, EYES, radius)
        np.testing.assert_array_equal(vis, hp["clip"](P, EYES, radius), err_msg=name)
        if len(P) >= 4:          # qhull needs a full-dimensional input
            ref = hp["qhull"](P, EYES, radius)
            if radius >= 3.0:
                np.testing.assert_array_equal(vis, ref, err_msg=name)
            else:
                # Flipped points behind the eye, and "sphere" clouds lie exactly on a sphere: a point can sit ON a
                # facet of the others' hull to 5e-17 (measured on sphere3001, radius 0.3: two such points in 12004),
                # where qhull reports a vertex and the strict test here does not.  Nothing to pin there.
                assert (vis != ref).sum() <= 2, (name, int((vis != ref).sum()))
        np.testing.assert_array_equal(cnt, vis.sum(1))


def test_hpr_real_scans(hp, golden):
    """16384-point scans, the reference's radius and a geometric one, four viewpoints each."""
    g = golden("scans13_fps16384.npz")
    eyes = np.array([[0, 0, 2.0], [1.2, 1.0, -1.0], [-1.6, 0.0, 0.0], [0.0, 1.6, 0.2]])
    for k, name in ((0, "partial"), (5, "gt"), (9, "partial")):
        P = g[name][k]
        for radius in (100.0, 10000.0):
            vis, cnt, second = hp["run"](P, eyes, radius)
            np.testing.assert_array_equal(vis, hp["qhull"](P, eyes, radius), err_msg="%s %d %g" % (name, k, radius))
            np.testing.assert_array_equal(vis[:2], hp["clip"](P, eyes[:2], radius))
            np.testing.assert_array_equal(cnt, vis.sum(1))


def test_hpr_lattice_takes_the_second_pass(hp):
    """A regular lattice: dozens of cospherical neighbours per point, polygons of > 24 vertices ->
    the second (wave-per-point) pass; still the oracle's mask, and qhull's."""
    gr = np.stack(np.meshgrid(*[np.arange(12)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) / 12 - 0.5
    for radius in (100.0, 10000.0):
        vis, cnt, second = hp["run"](gr, EYES[:2], radius)
        assert second > 0 or radius < 1000.0          # (at radius 100 this lattice's polygons stay under 24 vertices)
        np.testing.assert_array_equal(vis, hp["clip"](gr, EYES[:2], radius))
        np.testing.assert_array_equal(vis, hp["qhull"](gr, EYES[:2], radius))
        np.testing.assert_array_equal(cnt, vis.sum(1))


def test_hpr_polygons_over_128_vertices(hp):
    """A point on the viewing axis inside a ring of 300 points at the same distance from the eye and the same angle
    from the axis: its normal-cone polygon is a 300-gon -- beyond the 16 vertices of the first pass and beyond the 128
    of the wave-per-point pass's first tier, i.e. the large-polygon launch.  Masks equal the clipping oracle's and qhull's."""
    m = 300
    th = 2.0 * np.pi * (np.arange(m) + 0.25) / m
    ang = 0.2
    ring = np.stack([np.sin(ang) * np.cos(th), np.sin(ang) * np.sin(th), np.full(m, np.cos(ang))], 1)
    rng = np.random.default_rng(11)
    far = rng.normal(size=(200, 3))
    far = far / np.linalg.norm(far, axis=1, keepdims=True) * 1.5            # a shell behind: hidden or not, it cuts nothing near the axis
    P = np.concatenate([[[0.0, 0.0, 1.0]], ring, far]).astype(np.float32)
    eye = np.zeros((1, 3))
    for radius in (100.0, 10000.0):
        vis, cnt, second = hp["run"](P, eye, radius)
        assert second >= 1 or radius > 100.0      # (at the large radius the point's own direction already separates it)
        np.testing.assert_array_equal(vis, hp["clip"](P, eye, radius))
        np.testing.assert_array_equal(vis, hp["qhull"](P, eye, radius))
        assert vis[0, 0]


def test_hpr_polygons_over_1024_vertices(hp):
    """The same construction with a ring of 2500 points: a 2500-gon, beyond the 1024 vertices the LDS tiers hold --
    round 3 returned an error there; the polygon now moves to global memory (n + 8 vertices per buffer: it cannot
    overflow).  Masks equal the clipping oracle's (its own cap is 4096) and qhull's."""
    m = 2500
    th = 2.0 * np.pi * (np.arange(m) + 0.25) / m
    ang = 0.2
    ring = np.stack([np.sin(ang) * np.cos(th), np.sin(ang) * np.sin(th), np.full(m, np.cos(ang))], 1)
    rng = np.random.default_rng(12)
    far = rng.normal(size=(300, 3))
    far = far / np.linalg.norm(far, axis=1, keepdims=True) * 1.5
    P = np.concatenate([[[0.0, 0.0, 1.0]], ring, far]).astype(np.float32)
    eye = np.zeros((1, 3))
    vis, cnt, second = hp["run"](P, eye, 100.0)
    assert second >= 1
    np.testing.assert_array_equal(vis, hp["clip"](P, eye, 100.0))
    np.testing.assert_array_equal(vis, hp["qhull"](P, eye, 100.0))
    np.testing.assert_array_equal(cnt, vis.sum(1))
    assert vis[0, 0]


def test_hpr_edge_cases(hp):
    import torch
    lib = hp["lib"].lib
    rng = np.random.default_rng(3)
    P = (rng.random((500, 3)) - 0.5).astype(np.float32)
    # exact duplicates (incl. -0 against +0): the lowest-index copy stands for the group, later copies are hidden and cut
    # nothing; qhull reports one copy per group too (an arbitrary one): its counts and visible locations are these
    D = np.concatenate([P[:40], P, P[:40], P[100:130]])
    D[5, 1] = -0.0
    D[45, 1] = 0.0
    D[545, 1] = 0.0
    vis, cnt, _ = hp["run"](D, EYES[:2], 100.0)
    uniq = np.concatenate([D[:40], D[80:540]])
    base, bcnt, _ = hp["run"](uniq, EYES[:2], 100.0)
    np.testing.assert_array_equal(vis[:, :40], base[:, :40])
    assert not vis[:, 40:80].any() and not vis[:, 540:].any()
    np.testing.assert_array_equal(vis[:, 80:540], base[:, 40:])
    np.testing.assert_array_equal(cnt, bcnt)
    np.testing.assert_array_equal(vis, hp["clip"](D, EYES[:2], 100.0))          # the oracle applies the same rule
    q = hp["qhull"](D, EYES[:2], 100.0)
    np.testing.assert_array_equal(q.sum(1), cnt)
    for v in range(2):
        locs = lambda m: {tuple(x) for x in (D[m] + np.float32(0.0)).tolist()}
        assert locs(q[v].astype(bool)) == locs(vis[v].astype(bool))
    # a NaN point is hidden and hides nothing; the eye on a point
    N = P.copy()
    N[7] = np.nan
    vis, _, _ = hp["run"](N, EYES[:1], 100.0)
    ref, _, _ = hp["run"](np.delete(P, 7, 0), EYES[:1], 100.0)
    assert not vis[0, 7]
    np.testing.assert_array_equal(np.delete(vis[0], 7), ref[0])
    vis, _, _ = hp["run"](P, P[3:4].astype(np.float64), 100.0)
    np.testing.assert_array_equal(vis, hp["clip"](P, P[3:4].astype(np.float64), 100.0))
    # empty inputs, bad arguments
    vis, cnt, _ = hp["run"](np.zeros((0, 3), np.float32), EYES, 10.0)
    assert vis.shape == (4, 0) and (cnt == 0).all()
    assert lib.genpc_hpr_visibility(0, 5, None, None, 1.0, None, None, None, None) == 1
    assert lib.genpc_hpr_visibility(1, 5, None, None, 1.0, None, None, None, None) == -1
    assert lib.genpc_hpr_visibility(1, 5, None, None, 0.0, None, None, None, None) == -1
    assert lib.genpc_hpr_visibility(-1, 5, None, None, 1.0, None, None, None, None) == -1
    torch.cuda.synchronize()


def test_hpr_viewpoint_select_sizes(hp):
    """viewpoint_select's shape: 64 viewpoints x 10000 FPS-ordered points at the reference's radius;
    the mask equals the clipping oracle on a sample of views, the counts equal qhull's everywhere."""
    import torch
    from genpc_amd.DepthPrompting import fibonacci_sphere
    from genpc_amd.fps import fps_sampling
    from oracle import hpr
    rng = np.random.default_rng(5)
    v = rng.normal(size=(40000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    cloud = (v * (0.3 + 0.2 * np.abs(np.sin(3 * v[:, :1])))).astype(np.float32)
    pts = torch.from_numpy(cloud).cuda()
    sub = pts[fps_sampling(pts, 10000).long()].cpu().numpy()
    eyes = np.asarray(fibonacci_sphere(64, 1.6), np.float64)
    vis, cnt, second = hp["run"](sub, eyes, 10000.0)
    np.testing.assert_array_equal(cnt, hpr.visible_counts(sub, eyes, 10000.0))
    np.testing.assert_array_equal(vis[[0, 31, 63]], hp["clip"](sub, eyes[[0, 31, 63]], 10000.0))


def test_hpr_differential_fuzz(hp):
    """120 random configurations -- sizes 1 .. 4000, radii 0.2 .. 1e5, eyes outside, near and inside the cloud,
    balls / spheres / shells / planes / lines / clusters / duplicated points / a lattice, three eyes each: the GPU
    mask equals the clipping oracle's in EVERY case (the two run the same arithmetic; every shortcut the kernel
    takes -- tile culling, early accept, verify phase, hand-offs -- must leave no trace)."""
    rng = np.random.default_rng(2024)
    kinds = ["ball", "sphere", "shell", "plane", "line", "clusters", "dups", "lattice"]
    total = second_total = 0
    for case in range(120):
        kind = kinds[case % len(kinds)]
        n = int(rng.integers(1, 4001)) if case % 5 else int(rng.integers(1, 140))
        scale = 10.0 ** rng.uniform(-2, 2)
        if kind == "ball":
            P = rng.random((n, 3)) - 0.5
        elif kind in ("sphere", "shell"):
            v = rng.normal(size=(n, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            P = v * (0.5 if kind == "sphere" else (0.4 + 0.1 * rng.random((n, 1))))
        elif kind == "plane":
            P = np.concatenate([rng.random((n, 2)) - 0.5, 1e-3 * rng.normal(size=(n, 1))], 1)
        elif kind == "line":
            P = np.outer(rng.random(n) - 0.5, [1.0, 0.3, -0.2]) + 1e-4 * rng.normal(size=(n, 3))
        elif kind == "clusters":
            c = rng.random((8, 3)) - 0.5
            P = c[rng.integers(0, 8, n)] + 0.01 * rng.normal(size=(n, 3))
        elif kind == "dups":
            base = rng.random((max(1, n // 3), 3)) - 0.5
            P = base[rng.integers(0, len(base), n)]
        else:
            g = int(max(2, round(n ** (1 / 3))))
            P = np.stack(np.meshgrid(*[np.arange(g)] * 3, indexing="ij"), -1).reshape(-1, 3) / g - 0.5
        P = (P * scale + rng.normal(size=3) * scale * 0.1).astype(np.float32)
        ext = float(np.abs(P - P.mean(0)).max()) + 1e-6
        eyes = np.stack([P.mean(0) + rng.normal(size=3) * ext * f for f in (4.0, 1.2, 0.3)]).astype(np.float64)
        radius = ext * 10.0 ** rng.uniform(-0.7, 5)
        vis, cnt, second = hp["run"](P, eyes, radius)
        exp = hp["clip"](P, eyes, radius)
        assert np.array_equal(vis, exp), (case, kind, len(P), radius, int((vis != exp).sum()))
        np.testing.assert_array_equal(cnt, vis.sum(1))
        total += vis.size
        second_total += second
    assert total > 100000


def test_best_view_counts_agree_with_full_pass(golden):
    """genpc_hpr_best_view_counts (viewpoint_select's pass: views that can no longer see the most points are dropped
    after the first polygon kernel): argmax and the exact views' counts and masks equal the full pass; the pruned
    views' counts are lower bounds below the maximum.  Three scans x 1024 viewpoints, and a tie (two identical
    viewpoints: the first must win)."""
    import ctypes
    import torch
    from types import SimpleNamespace
    from genpc_amd import _lib
    from genpc_amd.DepthPrompting import DepthPrompting
    cfg = SimpleNamespace(device="cuda", fovy=49.1, res=256, cam_res=256, padding=0.15, rescale=True, point_size=1,
                          mask_pixel_rate=3, view_num=1024, distance=1.6, downsample_num=10000, removal_radius=10000)
    dp = DepthPrompting(cfg)
    g = golden("scans13_fps16384.npz")
    for k, arr in ((0, g["partial"]), (5, g["gt"]), (9, g["partial"])):
        pts = torch.from_numpy(np.ascontiguousarray(arr[k][:6000])).cuda()
        vis, cnt, _ = dp.hidden_point_removal(pts, dp.viewpoints, 10000.0)
        eyes = torch.as_tensor(np.asarray(dp.viewpoints, np.float64)).cuda()
        c, n = eyes.shape[0], pts.shape[0]
        v2 = torch.zeros(c, n, device="cuda", dtype=torch.uint8)
        c2 = torch.empty(c, device="cuda", dtype=torch.int32)
        ex = torch.empty(c, device="cuda", dtype=torch.uint8)
        second = ctypes.c_int(0)
        rc = _lib.on_device_of(pts, _lib.lib.genpc_hpr_best_view_counts, c, n, _lib.ptr(pts), _lib.ptr(eyes), 10000.0,
                               _lib.ptr(v2), _lib.ptr(c2), _lib.ptr(ex), ctypes.addressof(second))
        assert rc == 1, _lib.last_error()
        exact = ex.bool()
        assert int(torch.argmax(c2)) == int(torch.argmax(cnt))
        assert bool(exact[int(torch.argmax(cnt))]) and 0 < int(exact.sum()) < c          # something was pruned
        assert torch.equal(c2[exact], cnt[exact]) and torch.equal(v2[exact].bool(), vis[exact])
        assert bool((c2[~exact] <= cnt[~exact]).all()) and bool((c2[~exact] < cnt.max()).all())
        assert dp.viewpoint_select(pts) == int(torch.argmax(cnt))
        print("scan %d: %d of %d views exact" % (k, int(exact.sum()), c))
    # ties: every viewpoint twice -> the first copy wins, both copies stay exact
    eyes2 = torch.cat([eyes[:512], eyes[:512]])
    vis_t, cnt_t, _ = dp.hidden_point_removal(pts, eyes2.cpu().numpy(), 10000.0)
    _, cnt_b, _ = dp.hidden_point_removal(pts, eyes2.cpu().numpy(), 10000.0, best_only=True)
    assert int(torch.argmax(cnt_b)) == int(torch.argmax(cnt_t)) < 512
    best = int(torch.argmax(cnt_t))
    assert int(cnt_b[best]) == int(cnt_b[best + 512]) == int(cnt_t[best])
