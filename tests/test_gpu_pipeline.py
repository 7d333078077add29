"""BASELINE config 2 (the geometric stages of one completed scan, chained, 8192-point partial
scan) and config 5's per-rank shape (8 scans x 32768 points in lock-step through the full
registration objective).  Every stage of the chain is checked against the oracle ON THE STAGE'S
ACTUAL INPUT (the previous stage's GPU output); the end of the chain by outcome."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from genpc_amd import pipeline, reg_xyz
    from genpc_amd.DepthPrompting import DepthPrompting
    cfg = pipeline.default_cfg("cuda", view_num=256)
    return dict(torch=torch, P=pipeline, R=reg_xyz, cfg=cfg, dp=DepthPrompting(cfg))


def rot(axis, deg):
    a = np.asarray(axis, np.float64)
    a /= np.linalg.norm(a)
    t = np.deg2rad(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * K @ K


def c2_inputs(golden, scan=0):
    """One bundled scan: the 16384-point ground truth is the 'generated' shape after an unknown
    similarity transform (what the image-to-3D model returns: its own frame and scale); the
    partial scan is the observation at 8192 points."""
    g = golden("scans13_fps16384.npz")
    gt = g["gt"][scan]
    partial = g["partial"][scan][:8192].copy()
    c = (gt.max(0) + gt.min(0)) / 2
    gen = (((gt - c) / (gt.max(0) - gt.min(0)).max()).astype(np.float64) @ rot([0.2, 1.0, 0.1], 9.0).T).astype(np.float32)
    rng = np.random.default_rng(7)
    img = rng.random((3, 1024, 1024), dtype=np.float32)
    return partial, gen, img, gt


def test_config2_chain_8192(env, oracle, golden):
    torch = env["torch"]
    partial, gen, img, gt = c2_inputs(golden)
    Pt, Gt, It, GTt = (torch.from_numpy(x).cuda() for x in (partial, gen, img, gt))
    out = env["P"].complete_scan(Pt, Gt, It, GTt, cfg=env["cfg"], dp=env["dp"])
    cfg = env["cfg"]
    # ---- stage 1: the chosen viewpoint's visible set is Katz' (qhull), then projection / pixels / splat of the
    # visible points against the oracle, on the chosen camera ----
    from oracle import hpr
    dp = env["dp"]
    eye = np.asarray(dp.view, np.float64)
    np.testing.assert_allclose(np.abs(eye), np.abs(np.asarray(dp.viewpoints[out["view"]], np.float64)))      # the view or its opposite
    vis = np.zeros(len(partial), bool)
    vis[hpr.hidden_point_removal(partial, eye, cfg.removal_radius)] = True
    np.testing.assert_array_equal(out["visible"].cpu().numpy(), vis)
    assert 0.2 < vis.mean() <= 1.0
    cam = dp.cam.reshape(1, 12).cpu().numpy()
    ouv, odepth, _, _ = oracle.get_uvs(cam, env["dp"].focal, partial, rescale=True, padding=cfg.padding)
    np.testing.assert_array_equal(out["uv"].cpu().numpy(), ouv[0])
    np.testing.assert_array_equal(out["depth"].cpu().numpy(), odepth[0])
    opix = oracle.uv_to_pixels(ouv[0], cfg.res, cfg.res - 1)
    np.testing.assert_array_equal(out["pixels"].cpu().numpy(), opix)
    d = odepth[0][vis]
    grey = (np.float32(0.1) + np.float32(0.8) * (np.float32(1) - (d - d.min()) / (d.max() - d.min()))).astype(np.float32)
    osparse, _ = oracle.paint_pixels(cfg.res, opix[vis], np.repeat(grey[:, None], 3, 1), cfg.point_size)
    np.testing.assert_allclose(out["sparse_depth"].cpu().numpy(), osparse, atol=1e-6)
    assert float(out["hole_mask1"].sum()) > 0 and out["sparse_img"].shape == (3, cfg.res, cfg.res)
    # ---- stage 2a: colours of the partial points from the generated image ----
    opix1024 = oracle.uv_to_pixels(ouv[0], 1024, 1023)
    np.testing.assert_array_equal(out["point_colors"].cpu().numpy(), oracle.gather_colors(opix1024, img))
    # ---- stage 2b: registration by outcome: the aligned generated shape explains the scan ----
    res = out["reg"]
    src, tgt = res["source"].cpu().numpy(), res["target"].cpu().numpy()
    np.testing.assert_allclose(src, partial, atol=1e-5)            # the scan returns to its own frame
    d1, _, _, _ = oracle.chamfer_forward(src[None], tgt[None], 1)
    cd_partial = float(np.sqrt(d1).mean())
    assert cd_partial < 0.025, cd_partial
    # the loop is bit-reproducible (tests/test_gpu_determinism.py): its transforms against the committed golden ones
    # (tests/golden/pose_loop_golden.npz, this library's own outcome on this input; make_pose_golden.py)
    gg = golden("pose_loop_golden.npz")
    np.testing.assert_allclose(np.asarray(res["diff_transform"]), gg["c2_diff"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.asarray(res["coarse_transformation"]), gg["c2_coarse"], rtol=0, atol=1e-6)
    assert float(res["best_scale"]) == float(gg["c2_best_scale"])
    np.testing.assert_allclose(np.asarray(res["best_scales_transformation"]), gg["c2_S"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.asarray(res["best_transformation_xyz"]), gg["c2_Txyz"], rtol=0, atol=1e-6)
    # ---- stage 2c: fusion tail against the oracle on the registered clouds ----
    dd, _, _, _ = oracle.chamfer_forward(tgt[None], src[None], 1)
    keep = ~(dd[0] < np.float32(1e-4))
    allp = np.concatenate([src, tgt[keep]])
    fused_no_filter = env["R"].fuse(res["source"], res["target"], num_points=20000, std_ratio=None).cpu().numpy()
    sel = oracle.fps(allp, 20000, 1) if allp.shape[0] > 20000 else np.arange(allp.shape[0])
    np.testing.assert_array_equal(fused_no_filter, allp[sel])
    omask = oracle.statistical_outlier_mask(fused_no_filter, 20, 2.5, 1)
    np.testing.assert_array_equal(out["fused"].cpu().numpy(), fused_no_filter[omask])
    # ---- metric on the chain's own output ----
    pred, gtm = out["pred_metric_points"].cpu().numpy(), out["gt_metric_points"].cpu().numpy()
    e1, e2, _, _ = oracle.chamfer_forward(pred[None], gtm[None], 1)
    ed, _ = oracle.emd_forward(pred[None], gtm[None], 0.005, 50, 1)
    m = out["metric"].cpu().numpy()
    np.testing.assert_allclose(m[0], oracle.cd_l1(e1, e2), rtol=3e-7)
    np.testing.assert_allclose(m[1], oracle.cd_l2(e1, e2), rtol=3e-7)
    np.testing.assert_allclose(m[2], oracle.emd_loss(ed), rtol=3e-7)
    # the completed scan is far closer to the ground truth than the partial scan was
    p1, p2, _, _ = oracle.chamfer_forward(env["P"].fps_to(Pt, 16384).cpu().numpy()[None], gtm[None], 1)
    assert m[0] < 0.6 * float(oracle.cd_l1(p1, p2)), (m[0], float(oracle.cd_l1(p1, p2)))


def test_stage1_under_the_alignment_loop_changes_nothing(env, golden):
    """complete_scan runs stage 1 on a second stream / host thread under reg() (pipeline.py); one after the other gives
    the same bits for every product (three runs of each: a race would show as a difference somewhere)."""
    torch = env["torch"]
    partial, gen, img, gt = c2_inputs(golden)
    Pt, Gt, It, GTt = (torch.from_numpy(x).cuda() for x in (partial, gen, img, gt))
    ref = env["P"].complete_scan(Pt, Gt, It, GTt, cfg=env["cfg"], dp=env["dp"], overlap=False)
    keys = ("visible", "uv", "depth", "pixels", "sparse_img", "sparse_depth", "point_colors", "fused", "pred_metric_points")
    for rep in range(3):
        for ov in (True, False):
            out = env["P"].complete_scan(Pt, Gt, It, GTt, cfg=env["cfg"], dp=env["dp"], overlap=ov)
            assert out["view"] == ref["view"]
            for k in keys:
                assert torch.equal(out[k], ref[k]), (k, ov, rep)
            assert torch.equal(out["reg"]["transformation"], ref["reg"]["transformation"]) if "transformation" in out["reg"] else True
            assert torch.equal(out["reg"]["source"], ref["reg"]["source"]) and torch.equal(out["reg"]["target"], ref["reg"]["target"])
            ma, mb = out["metric"], ref["metric"]
            if isinstance(ma, dict):
                assert all(torch.equal(torch.as_tensor(ma[k]), torch.as_tensor(mb[k])) for k in mb)
            else:
                assert torch.equal(torch.as_tensor(ma), torch.as_tensor(mb))


def test_side_streams_and_the_deferred_sampling_check_change_nothing(env, golden):
    """pipeline.prepare_streams (the single-scan path's side streams made and used up front) is idempotent, and a scan whose
    samplings are checked beside the tail (genpc_fps_defer, the default) gives the bits of one whose check is in line."""
    torch = env["torch"]
    P = env["P"]
    partial, gen, img, gt = c2_inputs(golden)
    Pt, Gt, It, GTt = (torch.from_numpy(x).cuda() for x in (partial, gen, img, gt))
    P.prepare_streams("cuda")
    P.prepare_streams(torch.device("cuda", torch.cuda.current_device()))
    keys = ("visible", "uv", "depth", "pixels", "point_colors", "fused", "pred_metric_points", "gt_metric_points")
    was = P._FPS_DEFER
    try:
        P._FPS_DEFER = False
        ref = P.complete_scan(Pt, Gt, It, GTt, cfg=env["cfg"], dp=env["dp"])
        P._FPS_DEFER = True
        for _ in range(2):
            out = P.complete_scan(Pt, Gt, It, GTt, cfg=env["cfg"], dp=env["dp"])
            for k in keys:
                assert torch.equal(out[k], ref[k]), k
            assert torch.equal(torch.as_tensor(out["metric"]), torch.as_tensor(ref["metric"]))
    finally:
        P._FPS_DEFER = was
    from genpc_amd import _lib
    assert _lib.on_device_of(Pt, _lib.lib.genpc_fps_deferred_check) == 0


def test_scans_in_flight_side_by_side_change_nothing(env, golden):
    """pipeline.complete_scans: several scans at a time on one GPU (a host thread and a stream pair per lane); every scan's
    products are the bits of a call of complete_scan on its own."""
    torch = env["torch"]
    partial, gen, img, gt = c2_inputs(golden)
    Pt, Gt, It, GTt = (torch.from_numpy(x).cuda() for x in (partial, gen, img, gt))
    # a second, different scan: the same shape seen mirrored (different view, different registration)
    P2 = (Pt * torch.tensor([1.0, -1.0, 1.0], device="cuda")).contiguous()
    G2 = (Gt * torch.tensor([1.0, -1.0, 1.0], device="cuda")).contiguous()
    GT2 = (GTt * torch.tensor([1.0, -1.0, 1.0], device="cuda")).contiguous()
    jobs = [(Pt, Gt, It, GTt), (P2, G2, It, GT2)] * 3
    ref = [env["P"].complete_scan(*j, cfg=env["cfg"], dp=env["dp"], overlap=False) for j in jobs[:2]]
    keys = ("visible", "uv", "depth", "point_colors", "fused", "pred_metric_points")
    for lanes in (2, 3):
        outs = env["P"].complete_scans(jobs, lanes=lanes, cfg=env["cfg"])
        assert len(outs) == len(jobs)
        for k, out in enumerate(outs):
            r = ref[k % 2]
            assert out["view"] == r["view"]
            for key in keys:
                assert torch.equal(out[key], r[key]), (key, k, lanes)
            assert torch.equal(out["reg"]["source"], r["reg"]["source"]) and torch.equal(out["reg"]["target"], r["reg"]["target"])
            ma, mb = out["metric"], r["metric"]
            if isinstance(ma, dict):
                assert all(torch.equal(torch.as_tensor(ma[q]), torch.as_tensor(mb[q])) for q in mb)
            else:
                assert torch.equal(torch.as_tensor(ma), torch.as_tensor(mb))


def test_lanes_inherit_the_callers_thread_modes(env):
    """the library's per-call modes are thread-local and run_in_lanes / complete_scan's stage-1 thread are new
    threads: a caller in strict arithmetic got FMA bits from its lanes.  The caller's modes (and torch's grad mode) now
    travel with the work."""
    import torch
    from genpc_amd import _lib, pipeline
    L = _lib.lib
    dev = torch.device("cuda")

    def probe(li, _):
        st = _lib.thread_state()
        return (L.genpc_get_arith(), st[0][1], st[0][3], st[0][5], st[0][6], torch.is_grad_enabled())

    base = pipeline.run_in_lanes(probe, range(3), 3, dev)
    prev_a = L.genpc_set_arith_thread(0)
    prev_n = L.genpc_nn_tune(1, -1)
    prev_e = L.genpc_emd_tune(1, -1)
    prev_p = L.genpc_pose_tune(0)
    prev_f = L.genpc_fps_tune(1)
    try:
        with torch.no_grad():
            got = pipeline.run_in_lanes(probe, range(3), 3, dev)
    finally:
        L.genpc_set_arith_thread(prev_a); L.genpc_emd_tune(prev_e, -1); L.genpc_pose_tune(prev_p); L.genpc_fps_tune(prev_f)
        _lib.apply_thread_state((tuple(_lib.thread_state()[0][:1]) + (-1,) + tuple(_lib.thread_state()[0][2:]), torch.is_grad_enabled()))
    assert all(g == (0, 1, 1, 0, 3, False) for g in got), got          # (genpc_fps_tune(1) = bits 1 | 2: the pre-fix form)
    assert all(b[0] == L.genpc_get_arith() and b[5] for b in base), base
    # and a lane's strict-mode sampling is the caller's strict-mode sampling
    x = torch.rand(1, 5000, 3, device=dev)
    from genpc_amd.fps import fps_sampling
    prev_a = L.genpc_set_arith_thread(0)
    try:
        want = fps_sampling(x, 1000)
        lanes = pipeline.run_in_lanes(lambda li, _: fps_sampling(x, 1000), range(2), 2, dev)
    finally:
        L.genpc_set_arith_thread(prev_a)
    assert all(torch.equal(want, r) for r in lanes)


def test_voxel_down_sample_vs_oracle(env, oracle, golden):
    torch = env["torch"]
    g = golden("scans13_fps16384.npz")
    for xyz in (g["gt"][2], g["partial"][5][:5000], np.repeat(g["gt"][1][:300], 7, axis=0)):
        for voxel in (0.02, 0.03, 0.04, 0.5):
            got = env["R"].voxel_down_sample(torch.from_numpy(np.ascontiguousarray(xyz)).cuda(), voxel).cpu().numpy()
            np.testing.assert_array_equal(got, oracle.voxel_down_sample(xyz, voxel))
    with pytest.raises(ValueError):
        bad = g["gt"][0].copy()
        bad[5, 2] = np.inf
        env["R"].voxel_down_sample(torch.from_numpy(bad).cuda(), 0.03)


def c5_scan(seed, n=32768):
    """SURVEY 8d scan bench: a complete shape (ellipsoid + box union surface, max extent 1) and a
    partial observation of it (the half facing +z), under a known similarity transform."""
    rng = np.random.default_rng(1000 + seed)
    u = rng.standard_normal((n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    ell = u * np.array([0.5, 0.3, 0.22])
    box = (rng.random((n, 3)) - 0.5) * np.array([0.3, 0.5, 0.3])
    face = rng.integers(0, 3, n)
    box[np.arange(n), face] = np.sign(box[np.arange(n), face]) * np.array([0.15, 0.25, 0.15])[face]
    pick = rng.random(n) < 0.6
    complete = np.where(pick[:, None], ell, box + np.array([0.1, 0.0, 0.0]))
    complete = (complete - (complete.max(0) + complete.min(0)) / 2) / (complete.max(0) - complete.min(0)).max()
    # the loop starts at scale 0.75 and Adam moves log-scale by at most lr * 0.1 = 1e-3 per step
    # (diff_obj_pose.py:367,524-528): 201 steps reach at most 0.75 e^0.2 = 0.916
    s = rng.uniform(0.78, 0.9)
    theta = rng.uniform(-12, 12)
    t = rng.uniform(-0.04, 0.04, 3)
    R = rot([0, 1, 0], theta)
    c = complete.mean(0)
    posed = ((complete - c) * s) @ R.T + c + t
    front = posed[posed[:, 2] > np.median(posed[:, 2]) - 0.02]
    partial = front[rng.integers(0, front.shape[0], n)]
    return complete.astype(np.float32), partial.astype(np.float32), s, R, t


def test_config5_rank_shape_8x32768(env, golden):
    """8 scans x 32768 points through object_pose_optimization (full objective) in lock-step: transforms and
    loss histories equal the committed golden ones (the loop is bit-reproducible; tests/golden/pose_loop_golden.npz),
    and what they mean: every scan's scale / rotation / in-plane translation is recovered and the posed shape explains
    the observation (translation ALONG the fixed camera's axis trades against scale in the
    silhouette term and against the unobserved back half in the one-sided Chamfer term: it is
    checked through the distance it leaves, not as a number).  The partial clouds are SURVEY 8d's plain
    resampling: a third of their points are exact triplicates or more (csrc/nn_dedupe.hip)."""
    torch = env["torch"]
    from genpc_amd.optim_registration.diff_obj_pose import object_pose_optimization
    scans = [c5_scan(s) for s in range(8)]
    C = torch.from_numpy(np.stack([x[0] for x in scans])).cuda()
    P = torch.from_numpy(np.stack([x[1] for x in scans])).cuda()
    T, h, _ = object_pose_optimization(C, P, radius=0.02, lr=0.01, iters=200, render_size=224, return_history=True)
    assert T.shape == (8, 4, 4)
    gg = golden("pose_loop_golden.npz")
    np.testing.assert_allclose(T, gg["c5_T"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(h, gg["c5_hist"], rtol=1e-6, atol=1e-6)
    from genpc_amd.utils.loss_util import Completionloss
    cl = Completionloss("cd_l1")
    for i, (_, _, s, R, t) in enumerate(scans):
        sc = np.cbrt(np.linalg.det(T[i][:3, :3].astype(np.float64)))
        # (what the golden transforms mean, with room: 201 steps of the reference's schedule do not converge)
        assert abs(sc - s) < 0.08, (i, sc, s)
        np.testing.assert_allclose(T[i][:3, :3] / sc, R, atol=0.15)
        np.testing.assert_allclose(T[i][:2, 3], t[:2], atol=0.04)
        assert abs(T[i][2, 3] - t[2]) < 0.15
        Tt = torch.from_numpy(T[i]).cuda()
        c = C[i].mean(0)
        aligned = (C[i] - c) @ Tt[:3, :3].T + c + Tt[:3, 3]
        d = cl.chamfer_partial_l1(P[i][None].contiguous(), aligned[None].contiguous()).item()
        assert d < 0.05, (i, d)          # 5 % of the object's extent after 201 steps (reg() refines from here)


def test_voxel_down_sample_fuzz(env, oracle):
    """30 random clouds (1 .. 60000 points, scales 1e-3 .. 1e3, offsets, clusters with many points per voxel,
    exact duplicates) x random voxel sizes: the GPU grid equals the oracle's, bit for bit and in the same order."""
    torch = env["torch"]
    rng = np.random.default_rng(31)
    for case in range(30):
        n = int(rng.integers(1, 60001)) if case % 4 else int(rng.integers(1, 50))
        scale = 10.0 ** rng.uniform(-3, 3)
        kind = case % 3
        if kind == 0:
            P = rng.random((n, 3)) - 0.5
        elif kind == 1:
            c = rng.random((5, 3)) - 0.5
            P = c[rng.integers(0, 5, n)] + 0.02 * rng.normal(size=(n, 3))
        else:
            base = rng.random((max(1, n // 4), 3)) - 0.5
            P = base[rng.integers(0, len(base), n)]
        P = (P * scale + rng.normal(size=3) * scale * rng.choice([0.0, 3.0])).astype(np.float32)
        ext = float(np.ptp(P, axis=0).max()) + 1e-6 * scale
        voxel = ext * 10.0 ** rng.uniform(-2.2, 0.3)
        got = env["R"].voxel_down_sample(torch.from_numpy(P).cuda(), voxel).cpu().numpy()
        np.testing.assert_array_equal(got, oracle.voxel_down_sample(P, voxel), err_msg="case %d n %d voxel %g" % (case, n, voxel))
        if case % 2 == 0:         # coloured clouds (load_xyz / glb2point): the colours are averaged per voxel like the points
            col = rng.random((n, 3)).astype(np.float32)
            gx, gc = env["R"].voxel_down_sample(torch.from_numpy(P).cuda(), voxel, colors=torch.from_numpy(col).cuda())
            ox, oc = oracle.voxel_down_sample(P, voxel, colors=col)
            np.testing.assert_array_equal(gx.cpu().numpy(), ox, err_msg="case %d" % case)
            np.testing.assert_array_equal(gc.cpu().numpy(), oc, err_msg="case %d colours" % case)
    # the voxel size travels as a double (open3d's 0.03, not 0.03f): a point a hair inside a cell face of the
    # double grid stays there
    P = np.array([[0.0, 0.0, 0.0], [0.045 * (1 - 2e-8), 0.0, 0.0], [0.1, 0.1, 0.1]], np.float32)
    got = env["R"].voxel_down_sample(torch.from_numpy(P).cuda(), 0.03).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.voxel_down_sample(P, 0.03))
