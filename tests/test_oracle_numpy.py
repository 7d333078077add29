"""Cross-checks the C oracle against independent restatements: plain numpy fp32
(no FMA: equals oracle mode 0 bit for bit), float64 brute force, analytic
gradients, a pure-Python auction, and scipy's exact assignment as a bound."""
import numpy as np
import pytest

from conftest import gen_pair


def np_chamfer(a, b):
    """fp32, dx = target - query, (dx*dx + dy*dy) + dz*dz, first arg-min."""
    d = b[:, None, :, :] - a[:, :, None, :]           # [B,N,M,3]
    sq = d * d
    D = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
    return D.min(axis=2), D.argmin(axis=2).astype(np.int32)


@pytest.mark.parametrize("shape", [((2, 257, 3), (2, 129, 3)), ((1, 3, 3), (1, 700, 3)), ((3, 64, 3), (3, 1, 3))])
def test_chamfer_matches_numpy_fp32(oracle, shape):
    a, b = gen_pair(3, *shape)
    d1, d2, i1, i2 = oracle.chamfer_forward(a, b, 0)
    nd1, ni1 = np_chamfer(a, b)
    nd2, ni2 = np_chamfer(b, a)
    np.testing.assert_array_equal(d1, nd1)
    np.testing.assert_array_equal(i1, ni1)
    np.testing.assert_array_equal(d2, nd2)
    np.testing.assert_array_equal(i2, ni2)


@pytest.mark.parametrize("mode", [0, 1])
def test_chamfer_close_to_float64(oracle, mode):
    a, b = gen_pair(4, (1, 500, 3), (1, 400, 3))
    d1, _, i1, _ = oracle.chamfer_forward(a, b, mode)
    D = ((b.astype(np.float64)[:, None] - a.astype(np.float64)[:, :, None]) ** 2).sum(-1)
    np.testing.assert_allclose(d1, D.min(axis=2), rtol=3e-7)
    # the fp32 arg-min may differ from the fp64 one only at near-ties
    picked = np.take_along_axis(D, i1[..., None].astype(np.int64), axis=2)[..., 0]
    np.testing.assert_allclose(picked, D.min(axis=2), rtol=1e-6)


def test_chamfer_duplicate_targets_lowest_index(oracle):
    a, b = gen_pair(5, (1, 50, 3), (1, 40, 3))
    b[0, 20:40] = b[0, 0:20]
    _, _, i1, _ = oracle.chamfer_forward(b[:, :20].copy(), b, 0)
    np.testing.assert_array_equal(i1[0], np.arange(20))
    _, _, i1, _ = oracle.chamfer_forward(a, b, 1)
    assert i1.max() < 20


def test_chamfer_backward_analytic(oracle):
    a, b = gen_pair(6, (2, 120, 3), (2, 90, 3))
    d1, d2, i1, i2 = oracle.chamfer_forward(a, b, 0)
    rng = np.random.default_rng(0)
    g1 = rng.random(d1.shape, dtype=np.float32)
    g2 = rng.random(d2.shape, dtype=np.float32)
    gx1, gx2 = oracle.chamfer_backward(a, b, g1, g2, i1, i2)
    A, Bm = a.astype(np.float64), b.astype(np.float64)
    e1 = np.zeros_like(A)
    e2 = np.zeros_like(Bm)
    for bi in range(a.shape[0]):
        for j in range(a.shape[1]):
            v = 2 * g1[bi, j] * (A[bi, j] - Bm[bi, i1[bi, j]])
            e1[bi, j] += v
            e2[bi, i1[bi, j]] -= v
        for k in range(b.shape[1]):
            v = 2 * g2[bi, k] * (Bm[bi, k] - A[bi, i2[bi, k]])
            e2[bi, k] += v
            e1[bi, i2[bi, k]] -= v
    np.testing.assert_allclose(gx1, e1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(gx2, e2, rtol=1e-5, atol=1e-6)


def py_auction(x1, x2, eps, iters):
    """Straight restatement of SURVEY.md appendix B for one batch element
    (plain first-index arg-max; valid when bid values do not tie exactly)."""
    n = x1.shape[0]
    ass = np.full(n, -1, np.int64)
    inv = np.full(n, -1, np.int64)
    price = np.zeros(n, np.float32)
    max_inc = np.zeros(n, np.float32)
    max_idx = np.zeros(n, np.int64)
    bid = np.zeros(n, np.int64)
    inc = np.zeros(n, np.float32)
    for it in range(iters):
        last = it == iters - 1
        U = np.nonzero(ass == -1)[0]
        for j in U:
            d = x2 - x1[j]
            sq = d * d
            s = (sq[:, 0] + sq[:, 1]) + sq[:, 2]
            v = ((3.0 - np.sqrt(s).astype(np.float64)) - price.astype(np.float64)).astype(np.float32)
            order = np.argsort(-v, kind="stable")
            best, better = v[order[0]], v[order[1]]
            bid[j] = order[0]
            inc[j] = np.float32(np.float32(best - better) + np.float32(eps))
            max_inc[bid[j]] = max(max_inc[bid[j]], inc[j])
        for j in U:
            if abs(float(inc[j]) - float(max_inc[bid[j]])) <= 1e-6:
                max_idx[bid[j]] = j
        for j in U:
            b = bid[j]
            if last or max_idx[b] == j:
                prev = inv[b]
                if not last and prev != -1:
                    ass[prev] = -1
                inv[b] = j
                ass[j] = b
                price[b] = np.float32(price[b] + inc[j])
                max_inc[b] = np.float32(-1e9)
    d = x1 - x2[ass]
    sq = d * d
    return (sq[:, 0] + sq[:, 1]) + sq[:, 2], ass.astype(np.int32)


def test_emd_matches_python_auction(oracle):
    a, b = gen_pair(8, (1, 256, 3), (1, 256, 3), 0.0)
    d, ass = oracle.emd_forward(a, b, 0.005, 12, 0)
    pd, pass_ = py_auction(a[0], b[0], 0.005, 12)
    np.testing.assert_array_equal(ass[0], pass_)
    np.testing.assert_array_equal(d[0], pd)


def test_emd_converged_close_to_exact_assignment(oracle, golden):
    from scipy.optimize import linear_sum_assignment
    g = golden("emd_seed2_b1_256_conv.npz")
    a, b = g["xyz1"], g["xyz2"]
    d, ass = oracle.emd_forward(a, b, float(g["eps"]), int(g["iters"]), 0)
    assert len(np.unique(ass[0])) == 256                      # converged: a bijection
    C = np.sqrt(((a[0][:, None].astype(np.float64) - b[0][None].astype(np.float64)) ** 2).sum(-1))
    r, c = linear_sum_assignment(C)
    exact = C[r, c].mean()
    got = float(np.sqrt(d).mean())
    assert exact <= got + 1e-7 and got - exact < 0.002 * 3     # eps-optimality of the auction


def test_emd_dist_consistent_with_assignment(oracle):
    a, b = gen_pair(10, (2, 512, 3), (2, 512, 3), 0.0)
    for mode in (0, 1):
        d, ass = oracle.emd_forward(a, b, 0.005, 30, mode)
        pick = np.take_along_axis(b, ass[..., None].astype(np.int64), axis=1)
        e = (a.astype(np.float64) - pick) ** 2
        np.testing.assert_allclose(d, e.sum(-1), rtol=1e-6, atol=1e-12)
        assert (ass >= 0).all() and (ass < 512).all()


def test_emd_backward(oracle):
    a, b = gen_pair(12, (2, 256, 3), (2, 256, 3), 0.0)
    d, ass = oracle.emd_forward(a, b, 0.005, 10)
    g = np.random.default_rng(1).random(d.shape, dtype=np.float32)
    gx = oracle.emd_backward(a, b, g, ass)
    pick = np.take_along_axis(b, ass[..., None].astype(np.int64), axis=1)
    np.testing.assert_allclose(gx, 2 * g[..., None] * (a - pick), rtol=1e-6, atol=1e-7)


def test_fps_and_ply_reader(oracle, tmp_path):
    pts = np.random.default_rng(3).random((500, 3))
    p = tmp_path / "t.ply"
    with open(p, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\ncomment x\nelement vertex 500\n"
                b"property double x\nproperty double y\nproperty double z\nend_header\n")
        f.write(pts.astype("<f8").tobytes())
    np.testing.assert_array_equal(oracle.read_ply_xyz(str(p)), pts)
    x = pts.astype(np.float32)
    idx = oracle.fps(x, 32)
    assert idx[0] == 0 and len(set(idx.tolist())) == 32
    # each pick is the farthest point from the set picked so far
    d = np.full(500, np.inf, np.float32)
    for s in range(31):
        dd = ((x - x[idx[s]]) ** 2)
        d = np.minimum(d, (dd[:, 0] + dd[:, 1]) + dd[:, 2])
        assert idx[s + 1] == int(np.argmax(d))


def py_nm_distance_literal(q, t):
    """Loop-for-loop reading of NmDistanceKernel's control flow for one batch element
    (chamfer3D.cu:15-129), fp32 without contraction: tiles of 512, the 4-way unrolled
    body with its 'k==0 ||' only on the first slot (:36), the scalar tail (:113-124) and
    the 'k2==0 || result>best' merge (:126).  Slow; small cases only."""
    f = np.float32
    n, m = q.shape[0], t.shape[0]
    res = np.zeros(n, np.float32)
    res_i = np.zeros(n, np.int32)

    def dist(j, k):
        x2, y2, z2 = f(t[k, 0] - q[j, 0]), f(t[k, 1] - q[j, 1]), f(t[k, 2] - q[j, 2])
        return f(f(f(x2 * x2) + f(y2 * y2)) + f(z2 * z2))

    with np.errstate(invalid="ignore", over="ignore"):
        for k2 in range(0, m, 512):
            end_k = min(m, k2 + 512) - k2
            end_ka = end_k - (end_k & 3)
            for j in range(n):
                best_i, best = 0, f(0)
                for k in range(0, end_ka, 4):
                    for s in range(4):
                        d = dist(j, k2 + k + s)
                        if (s == 0 and k == 0) or d < best:
                            best, best_i = d, k + k2 + s
                for k in range(end_ka, end_k):
                    d = dist(j, k2 + k)
                    if k == 0 or d < best:
                        best, best_i = d, k + k2
                if k2 == 0 or res[j] > best:
                    res[j], res_i[j] = best, best_i
    return res, res_i


def test_chamfer_non_finite_follows_the_tile_structure(oracle):
    """NaN / inf inputs: the oracle must reproduce what the reference's tiled scan does --
    a NaN distance at the first target of a 512-tile discards that tile (tile 0: the result
    is that NaN at index 0), a NaN elsewhere drops one target (chamfer3D.cu:30-36,126)."""
    a, b = gen_pair(8, (1, 40, 3), (1, 1300, 3))
    a[0, 13] = np.nan                  # query with only NaN distances -> (NaN, 0)
    a[0, 14, 2] = np.inf               # query at infinity: every distance +inf -> (inf, 0)
    b[0, 5, 1] = np.nan                # mid-tile: one target dropped
    b[0, 512] = np.nan                 # tile 1 head: targets 512..1023 invisible
    b[0, 1024, 0] = np.inf             # tile 2 head at +inf: tile lives on (inf is not NaN)
    b[0, 1100, 0] = np.inf
    a[0, 15, 0] = np.inf               # inf - inf = NaN against targets 1024 and 1100: tile 2 dead for this query
    for blk in (b, b[:, :1024].copy(), b[:, :513].copy()):
        d1, _, i1, _ = oracle.chamfer_forward(a, blk, 0)
        e, ei = py_nm_distance_literal(a[0], blk[0])
        np.testing.assert_array_equal(i1[0], ei)
        assert np.array_equal(d1[0], e, equal_nan=True)
    d1, _, i1, _ = oracle.chamfer_forward(a, b, 0)
    assert np.isnan(d1[0, 13]) and i1[0, 13] == 0 and np.isinf(d1[0, 14]) and i1[0, 14] == 0
    live = np.ones(1300, bool)
    live[512:1024] = False
    live[5] = False
    assert not ((i1[0] >= 512) & (i1[0] < 1024)).any()
    # the nearest target of tile 1 would have won for some query if the tile were visible
    a2 = b[:, 600:640].copy() + np.float32(1e-4)
    a2[np.isnan(a2)] = 0
    d1, _, i1, _ = oracle.chamfer_forward(a2, b, 0)
    assert not ((i1[0] >= 512) & (i1[0] < 1024)).any()
    b0 = b.copy()
    b0[0, 0, 2] = np.nan               # tile 0 head: every query ends with (NaN, 0)
    d1, _, i1, _ = oracle.chamfer_forward(a, b0, 1)
    assert np.isnan(d1).all() and (i1 == 0).all()


def test_hpr_oracle_on_a_sphere_and_a_shell():
    """Katz' operator through qhull.  With a moderate radius it is geometric visibility: ~40 % of a
    sampled sphere, 95 % of them on the camera's side, an inner sphere invisible.  With the
    reference's radius (configs/config.yaml: removal_radius 10000, 10000 points, camera at 1.6) the
    flipped cloud is so flat that a point is only removed when another one sits at almost the same
    direction: most of the cloud counts as visible from every side -- the regime viewpoint_select
    really runs in."""
    from oracle import hpr
    rng = np.random.default_rng(0)
    u = rng.standard_normal((3000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    sphere = u * 0.4
    cam = np.array([0.0, 0.0, 1.6])
    vis = hpr.hidden_point_removal(sphere, cam, 100)
    assert 0.3 < len(vis) / 3000.0 < 0.55
    facing = sphere[:, 2] > 0.4 * 0.4 / 1.6              # beyond the tangent cone: geometrically hidden
    assert facing[vis].mean() > 0.9
    both = np.vstack([sphere, sphere * 0.5])
    assert (hpr.hidden_point_removal(both, cam, 100) >= 3000).mean() < 0.02
    big = hpr.hidden_point_removal(sphere, cam, 10000)
    assert len(big) / 3000.0 > 0.7 and facing[big].mean() < 0.7
    counts = hpr.visible_counts(sphere, [cam, -cam, [1.6, 0, 0]], 10000)
    assert counts.shape == (3,) and counts.min() > 1000


def test_hpr_clipping_oracle_equals_qhull(oracle, golden):
    """oracle/genpc_oracle_hpr.c (the hull-vertex test as a normal-cone polygon per point -- the form the
    GPU computes) against the reference's operator through qhull (oracle/hpr.py): IDENTICAL masks on
    random balls / spheres / shells, at the reference's radii and a tiny one, on two real scans, and on a
    lattice (cospherical ties, polygons of 100+ vertices)."""
    from oracle import hpr
    rng = np.random.default_rng(0)

    def both(P, eye, radius):
        a = oracle.hpr_visibility(P, eye, radius)
        b = np.zeros(len(P), bool)
        b[hpr.hidden_point_removal(P, eye, radius)] = True
        np.testing.assert_array_equal(a, b)
        return a
    for n in (50, 500, 3000):
        v = rng.normal(size=(n, 3))
        v /= np.linalg.norm(v, axis=1, keepdims=True)
        for P in ((rng.random((n, 3)) - 0.5), v * 0.5, v * (0.45 + 0.05 * rng.random((n, 1)))):
            for eye in ([0, 0, 3.0], [2.0, 1.0, -1.5]):
                for radius in (3.0, 100.0, 800.0, 10000.0):
                    both(P.astype(np.float32), eye, radius)
    g = golden("scans13_fps16384.npz")
    vis = both(g["partial"][0][:6000], [1.2, 1.0, -1.0], 10000.0)
    assert 0.3 < vis.mean() < 1.0
    both(g["gt"][5][:6000], [0, 0, 2.0], 100.0)
    gr = np.stack(np.meshgrid(*[np.arange(10)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32) / 10 - 0.5
    a, mv = oracle.hpr_visibility(gr, [0, 0, 3.0], 10000.0, True)
    b = np.zeros(len(gr), bool)
    b[hpr.hidden_point_removal(gr, [0, 0, 3.0], 10000.0)] = True
    np.testing.assert_array_equal(a, b)
    assert mv > 24          # larger than the GPU kernel's LDS polygons: the case its second pass exists for
    # exact duplicates: the lowest-index copy stands for the group (qhull reports one copy too -- an arbitrary one, so
    # the visible LOCATIONS and the counts are what agrees); a NaN point is hidden and hides nothing
    P = (rng.random((300, 3)) - 0.5).astype(np.float32)
    base = oracle.hpr_visibility(P, [0, 0, 3.0], 100.0)
    D = np.concatenate([P[:20], P, P[:20], P[40:60]])          # groups of three and of two, lowest copies in front
    D[5, 2] = -0.0
    D[25, 2] = 0.0                                             # -0 and +0 are the same coordinate
    d = oracle.hpr_visibility(D, [0, 0, 3.0], 100.0)
    base_d = oracle.hpr_visibility(np.concatenate([D[:20], D[40:320]]), [0, 0, 3.0], 100.0)      # the unique points, in order
    np.testing.assert_array_equal(d[:20], base_d[:20])
    assert not d[20:40].any() and not d[320:].any()            # later copies are hidden
    np.testing.assert_array_equal(d[40:320], base_d[20:])
    q = np.zeros(len(D), bool)
    q[hpr.hidden_point_removal(D, [0, 0, 3.0], 100.0)] = True
    assert int(q.sum()) == int(d.sum())                        # qhull's count
    locs = lambda m: {tuple(x) for x in (D[m] + np.float32(0.0)).tolist()}
    assert locs(q) == locs(d)                                  # and its visible locations


def test_hpr_clipping_oracle_equals_qhull_fuzz(oracle):
    """60 random generic configurations (balls, shells, clusters; 5 .. 3000 points; scale 1e-2 .. 1e2; the eye far
    from, near and inside the cloud; radius 0.2 .. 1e5 of the extent): the clipping oracle's mask equals qhull's
    point for point (79 000 points)."""
    from oracle import hpr
    rng = np.random.default_rng(77)
    total = 0
    for case in range(60):
        n = int(rng.integers(5, 3000))
        kind = case % 3
        if kind == 0:
            P = rng.random((n, 3)) - 0.5
        elif kind == 1:
            v = rng.normal(size=(n, 3))
            v /= np.linalg.norm(v, axis=1, keepdims=True)
            P = v * (0.4 + 0.1 * rng.random((n, 1)))
        else:
            c = rng.random((6, 3)) - 0.5
            P = c[rng.integers(0, 6, n)] + 0.05 * rng.normal(size=(n, 3))
        scale = 10.0 ** rng.uniform(-2, 2)
        P = (P * scale).astype(np.float32)
        ext = float(np.abs(P - P.mean(0)).max())
        eye = (P.mean(0) + rng.normal(size=3) * ext * float(rng.choice([4.0, 1.2, 0.3]))).astype(np.float64)
        radius = ext * 10.0 ** rng.uniform(-0.7, 5)
        a = oracle.hpr_visibility(P, eye, radius)
        b = np.zeros(n, bool)
        b[hpr.hidden_point_removal(P, eye, radius)] = True
        np.testing.assert_array_equal(a, b, err_msg="case %d" % case)
        total += n
    assert total > 70000
