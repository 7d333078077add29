"""a17: batched ICP and scale search on the GPU against the oracle's ICP
(oracle_icp: same fp32 NN, double sums) -- transforms within 1e-6 (sums are reduced
in another order), fitness / iteration counts equal -- and by outcome (known rigid
motions and anisotropic scales recovered)."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rg():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from genpc_amd import reg_xyz
    return dict(torch=torch, R=reg_xyz)


def rot(axis, deg):
    a = np.asarray(axis, float)
    a /= np.linalg.norm(a)
    th = math.radians(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(th) * K + (1 - math.cos(th)) * K @ K


def shape(seed, n):
    rng = np.random.default_rng(seed)
    u = rng.standard_normal((n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    return (u * np.array([0.5, 0.3, 0.2]) + 0.05 * np.abs(u[:, :1])).astype(np.float32)


def test_icp_matches_oracle(rg, oracle):
    torch = rg["torch"]
    target = shape(5, 3000)
    R0, t0 = rot([0.2, 1, 0.1], 6.0), np.array([0.02, -0.015, 0.01])
    src = ((target[::2].astype(np.float64) - t0) @ R0).astype(np.float32)
    for md in (0.075, 0.01):
        T, fit, rmse, its = rg["R"].registration_icp(torch.from_numpy(src).cuda(), torch.from_numpy(target).cuda(), md)
        oT, ofit, ormse, oits = oracle.icp(src, target, md)
        assert fit == ofit and its == oits
        np.testing.assert_allclose(T, oT, atol=1e-6)
        assert abs(rmse - ormse) < 1e-7
    T, fit, rmse, its = rg["R"].registration_icp(torch.from_numpy(src).cuda(), torch.from_numpy(target).cuda(), 0.075)
    np.testing.assert_allclose(T[:3, :3], R0, atol=5e-3)
    np.testing.assert_allclose(T[:3, 3], t0, atol=2e-3)


def test_icp_batch_equals_singles(rg, oracle):
    torch = rg["torch"]
    target = shape(6, 2500)
    src = (target[1::2].astype(np.float64) @ rot([0, 1, 0], 4.0).T * 1.02 + 0.01).astype(np.float32)
    inits = []
    for sc in np.linspace(1.5, 0.8, 11):
        S = np.eye(4)
        S[:3, :3] *= sc
        inits.append(S)
    S_, T_ = torch.from_numpy(src).cuda(), torch.from_numpy(target).cuda()
    Tb, fb, rb, ib = rg["R"].registration_icp(S_, T_, 0.075, np.stack(inits))
    for k in (0, 5, 10):
        oT, ofit, ormse, oits = oracle.icp(src, target, 0.075, init=inits[k])
        np.testing.assert_allclose(Tb[k], oT, atol=1e-6)
        assert fb[k] == ofit and ib[k] == oits


def test_coarse_sweep_and_scale_search(rg, oracle):
    torch = rg["torch"]
    target = shape(7, 3000)
    true_scales = np.array([1.1111111, 0.9333333, 0.8444444])      # on the 10-step grid of [0.8, 1.2]
    src = (target[::2].astype(np.float64) / true_scales).astype(np.float32)
    S_, T_ = torch.from_numpy(src).cuda(), torch.from_numpy(target).cuda()
    Sm, loss, Tb = rg["R"].iterative_scale_search(S_, T_, [(0.8, 1.2)] * 3, 10, cd_inv_weight=0.5)
    np.testing.assert_allclose(np.diag(Sm)[:3], true_scales, atol=1e-6)
    np.testing.assert_allclose(Tb, np.eye(4), atol=2e-3)
    # scores of a 3x3x3 grid against the oracle's Chamfer on each scaled source
    xs = np.linspace(0.8, 1.2, 3)
    cand = np.array([[x, y, z] for z in xs for x in xs for y in xs])
    exp = []
    for c in cand:
        sc = (src.astype(np.float64) * c).astype(np.float32)
        d1, d2, _, _ = oracle.chamfer_forward(sc[None], target[None], 1)
        exp.append(float(oracle.cd_partial_l1(d1)) + 0.5 * float(oracle.cd_partial_l1(d2)))
    Sm3, loss3, _ = rg["R"].iterative_scale_search(S_, T_, [(0.8, 1.2)] * 3, 3, cd_inv_weight=0.5)
    best = int(np.argmin(exp))
    np.testing.assert_allclose(np.diag(Sm3)[:3], cand[best])
    assert abs(loss3 - exp[best]) < 1e-6
    # coarse sweep: the source is the target shrunk by 1/1.29 -> scale 1.29 wins
    src2 = (target[::2].astype(np.float64) / 1.29).astype(np.float32)
    bs, bl, Tc = rg["R"].coarse_scale_sweep(torch.from_numpy(src2).cuda(), T_, cd_inv_weight=0.5)
    assert abs(bs - 1.29) < 1e-9
    s = np.cbrt(np.linalg.det(Tc[:3, :3]))
    assert abs(s - 1.29) < 1e-6


def test_voxel_down_sample_and_normalize(rg):
    torch = rg["torch"]
    xyz = torch.from_numpy(shape(3, 5000)).cuda()
    v = rg["R"].voxel_down_sample(xyz, 0.03)
    assert 100 < v.shape[0] < 5000
    # every output is the mean of the inputs in its voxel
    origin = xyz.min(0).values - 0.015
    kin = torch.floor((xyz - origin) / 0.03).long()
    kout = torch.floor((v - origin) / 0.03).long()
    assert len({tuple(k) for k in kout.tolist()}) == v.shape[0]
    k0 = kout[7]
    sel = (kin == k0).all(1)
    assert torch.allclose(xyz[sel].double().mean(0).float(), v[7], atol=1e-6)
    nrm, c, s = rg["R"].normalize_numpy(xyz, range=0.5)
    ext = nrm.max(0).values - nrm.min(0).values
    assert abs(float(ext.max()) - 1.0) < 1e-6 and float((nrm.max(0).values + nrm.min(0).values).abs().max()) < 1e-6


def test_reg_end_to_end(rg):
    """Completed-scan unit: a complete cloud in its generator frame, and a partial
    observation of the same object in a camera frame (anisotropically scaled, rotated,
    shifted).  After reg() the aligned complete cloud explains the partial one."""
    torch = rg["torch"]
    from genpc_amd.utils.loss_util import Completionloss
    complete = shape(21, 40000)
    c = (complete.max(0) + complete.min(0)) / 2
    complete = ((complete - c) / (complete.max(0) - complete.min(0)).max()).astype(np.float32)   # already "normalised"
    R0 = rot([0.1, 1.0, 0.05], 8.0)
    obs = (complete.astype(np.float64) * np.array([0.95, 1.1, 0.9]) * 0.8) @ R0.T + np.array([0.03, -0.02, 0.04])
    partial = obs[obs[:, 2] > obs[:, 2].mean() - 0.05][::3].astype(np.float32)
    res = rg["R"].reg(torch.from_numpy(partial).cuda(), torch.from_numpy(complete).cuda(), generative_model="trellis",
                      dataset="redwood", cd_inv_weight=0.5, diff_init=True, reg_fine_xyz=True)
    assert torch.allclose(res["source"], torch.from_numpy(partial).cuda(), atol=1e-5)   # source returns to its frame
    cl = Completionloss("cd_l1")
    before = cl.chamfer_partial_l1(torch.from_numpy(partial).cuda()[None], torch.from_numpy(complete).cuda()[None]).item()
    after = cl.chamfer_partial_l1(res["source"][None].contiguous(), res["target"][None].contiguous()).item()
    assert after < 0.012 and after < 0.35 * before, (before, after)


def test_fusion_tail(rg, oracle):
    torch = rg["torch"]
    tgt = shape(31, 30000)
    src = tgt[tgt[:, 2] > 0.1][:3000] + np.float32(1e-4)  # the partial cloud covers one cap of the target
    S_, T_ = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    filt, keep = rg["R"].remove_close_points(S_, T_, 1e-4)
    d1, _, _, _ = oracle.chamfer_forward(tgt[None], src[None], 1)
    np.testing.assert_array_equal(keep.cpu().numpy(), ~(d1[0] < np.float32(1e-4)))
    assert 0 < filt.shape[0] < tgt.shape[0]
    fused = rg["R"].fuse(S_, T_, num_points=20000, std_ratio=None)
    assert fused.shape == (20000, 3)
    allp = np.concatenate([src, tgt[keep.cpu().numpy()]])
    np.testing.assert_array_equal(fused.cpu().numpy(), allp[oracle.fps(allp, 20000, 1)])


@pytest.mark.parametrize("mode", [0, 1])
def test_statistical_outlier_filter(rg, oracle, mode):
    torch = rg["torch"]
    from genpc_amd import _lib
    rng = np.random.default_rng(8)
    pts = np.concatenate([shape(41, 6000), (rng.random((60, 3), dtype=np.float32) - 0.5) * 3]).astype(np.float32)
    pts[100:110] = pts[0:10]                           # duplicates: several zero distances
    prev = _lib.lib.genpc_set_arith(mode)
    try:
        m = rg["R"].knn_mean_distance(torch.from_numpy(pts).cuda(), 20).cpu().numpy()
        filt, keep = rg["R"].remove_noise_from_point_cloud(torch.from_numpy(pts).cuda(), 20, 1.5)
    finally:
        _lib.lib.genpc_set_arith(prev)
    np.testing.assert_array_equal(m, oracle.knn_mean_distance(pts, 20, mode))
    np.testing.assert_array_equal(keep.cpu().numpy(), oracle.statistical_outlier_mask(pts, 20, 1.5, mode))
    assert keep[:6000].float().mean() > 0.95 and keep[6000:].float().mean() < 0.2      # the far-away noise goes
    with pytest.raises(ValueError):
        rg["R"].knn_mean_distance(torch.from_numpy(pts).cuda(), 7)
    tiny = torch.from_numpy(pts[:5]).cuda()            # fewer points than k: mean over what exists
    mt = rg["R"].knn_mean_distance(tiny, 8).cpu().numpy()
    np.testing.assert_array_equal(mt, oracle.knn_mean_distance(pts[:5], 8, 1))


def test_knn_mean_distance_fuzz(rg, oracle):
    """12 random clouds (2 .. 20000 points; uniform, clustered, repeated points; scales 1e-2 .. 1e2; k in
    {8, 16, 20, 32}; both arithmetic modes): the k-NN mean distances are the oracle's, bit for bit."""
    torch = rg["torch"]
    from genpc_amd import _lib
    rng = np.random.default_rng(515)
    for case in range(12):
        n = int(rng.integers(2, 20001)) if case % 3 else int(rng.integers(2, 60))
        kind = case % 3
        if kind == 0:
            P = rng.random((n, 3)) - 0.5
        elif kind == 1:
            c = rng.random((5, 3)) - 0.5
            P = c[rng.integers(0, 5, n)] + 0.02 * rng.normal(size=(n, 3))
        else:
            base = rng.random((max(1, n // 3), 3)) - 0.5
            P = base[rng.integers(0, len(base), n)]
        P = (P * 10.0 ** rng.uniform(-2, 2)).astype(np.float32)
        k = int(rng.choice([8, 16, 20, 32]))
        mode = case & 1
        prev = _lib.lib.genpc_set_arith(mode)
        try:
            m = rg["R"].knn_mean_distance(torch.from_numpy(P).cuda(), k).cpu().numpy()
        finally:
            _lib.lib.genpc_set_arith(prev)
        np.testing.assert_array_equal(m, oracle.knn_mean_distance(P, k, mode), err_msg="case %d n %d k %d" % (case, n, k))


def test_icp_one_workgroup_fuzz(rg, oracle):
    """The one-workgroup solve (csrc/icp.hip icp_fused_kernel: target grid in LDS, 27-cell search) against the oracle's
    exhaustive search on inputs that stress the grid: coordinates far from the origin, a correspondence distance far below
    and far above the cloud's extent (cell-count cap, a single cell), sizes that are no multiple of a wave, duplicated
    targets and a lattice (exact distance ties: the lower index must win as in the exhaustive search), a source outside the
    target's bounding box, no inlier at all."""
    torch = rg["torch"]
    rng = np.random.default_rng(23)
    base = shape(8, 2600)

    def lattice(m):
        g = np.stack(np.meshgrid(*([np.arange(m)] * 3), indexing="ij"), -1).reshape(-1, 3)
        return (g * np.float32(0.05) - 0.3).astype(np.float32)

    cases = []
    for off in (0.0, 100.0):
        tgt = (base + np.float32(off)).astype(np.float32)
        src = ((base[::3].astype(np.float64) @ rot([0.1, 1, 0.3], 5.0).T) * 1.01 + 0.012 + off).astype(np.float32)
        cases.append(("offset %g" % off, src, tgt, 0.075))
    cases.append(("tiny max_dist", (base[::2] + np.float32(2e-4)).astype(np.float32), base, 1e-3))
    cases.append(("huge max_dist", (base[::5] * np.float32(1.1)).astype(np.float32), base, 10.0))
    cases.append(("ragged sizes", base[:37] + np.float32(0.01), base[:50], 0.075))
    dup = np.concatenate([base[:1500], base[:700], base[200:400]])
    cases.append(("duplicated targets", (base[::2] + np.float32([0.01, -0.005, 0.0])).astype(np.float32), dup, 0.075))
    lat = lattice(13)
    cases.append(("lattice ties", (lat[::2] + np.float32(0.025)).astype(np.float32), lat, 0.075))          # midway between lattice points
    cases.append(("source outside", (base[::4] + np.float32([1.5, 0, 0])).astype(np.float32), base, 0.075))
    cases.append(("partly outside", (base[::4] * np.float32(1.6)).astype(np.float32), base, 0.05))
    for name, src, tgt, md in cases:
        T, fit, rmse, its = rg["R"].registration_icp(torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), md)
        oT, ofit, ormse, oits = oracle.icp(src, tgt, md)
        assert fit == ofit and its == oits, (name, fit, ofit, its, oits)
        np.testing.assert_allclose(T, oT, atol=1e-6 * max(1.0, float(np.abs(tgt).max())), err_msg=name)
        assert abs(rmse - ormse) < 1e-7, name
    # a batch of random initial transforms in one call
    tgt = base
    src = ((base[1::2].astype(np.float64) @ rot([0, 1, 0.2], 3.0).T) + 0.01).astype(np.float32)
    inits = []
    for k in range(7):
        M = np.eye(4)
        M[:3, :3] = rot(rng.standard_normal(3), float(rng.uniform(0, 8))) * float(rng.uniform(0.9, 1.1))
        M[:3, 3] = rng.uniform(-0.03, 0.03, 3)
        inits.append(M)
    Tb, fb, rb, ib = rg["R"].registration_icp(torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda(), 0.075, np.stack(inits))
    for k in range(7):
        oT, ofit, ormse, oits = oracle.icp(src, tgt, 0.075, init=inits[k])
        assert fb[k] == ofit and ib[k] == oits, k
        np.testing.assert_allclose(Tb[k], oT, atol=1e-6)
