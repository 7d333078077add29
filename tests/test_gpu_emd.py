"""Parity of the HIP auction-EMD path against the CPU oracle: assignment and
dist BIT-EXACT in both arithmetic modes (the round structure, the double-precision
bid value, the 1e-6 window and the forced last round are all preserved), prices
bit-exact except after a forced last round with several bidders on one object
(accumulation order there is a race in the reference too: 1e-6 relative)."""
import numpy as np
import pytest

from conftest import gen_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from genpc_amd import _lib, emd
    from genpc_amd.loss_functions import emdModule
    from genpc_amd.loss_functions.emd.emd_module import alloc_state
    from genpc_amd.utils.loss_util import Completionloss
    return dict(torch=torch, lib=_lib, emd=emd, mod=emdModule(), alloc=alloc_state, CL=Completionloss)


@pytest.fixture(autouse=True, params=[2, 1, 0], ids=["one_launch", "culled_bid", "tiled_bid"])
def bid_kernel(request, gp):
    """Every test of this file runs with all three implementations: all rounds in one launch whose threads own the
    points (csrc/emd_auction.hip, the default whenever the launch fits the chip), a launch per round step with the
    cell-sorted culled bid (csrc/emd_grid.hip) and with the tiled bid over all objects (emd_bid_kernel)."""
    prev = gp["lib"].lib.genpc_emd_tune(request.param, -1)
    yield request.param
    gp["lib"].lib.genpc_emd_tune(prev, -1)
    if request.param == 2:
        # no one-launch call of the test was abandoned (its dist would be NaN)
        assert gp["lib"].lib.genpc_emd_status(1, None) == 0


def run_hip(gp, a, b, eps, iters, mode):
    torch = gp["torch"]
    prev = gp["lib"].lib.genpc_set_arith(mode)
    try:
        A, B = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        s = gp["alloc"](a.shape[0], a.shape[1], b.shape[1], A.device)
        rc = gp["emd"].forward(A, B, s["dist"], s["assignment"], s["price"], s["assignment_inv"], s["bid"],
                               s["bid_increments"], s["max_increments"], s["unass_idx"], s["unass_cnt"],
                               s["unass_cnt_sum"], s["cnt_tmp"], s["max_idx"], eps, iters)
        torch.cuda.synchronize()
        assert rc == 1
    finally:
        gp["lib"].lib.genpc_set_arith(prev)
    return {k: v.cpu().numpy() for k, v in s.items()}


@pytest.mark.parametrize("name", ["emd_seed0_b2_1024.npz", "emd_seed5_dups.npz", "emd_seed9_2304_dups.npz",
                                  "emd_seed2_b1_256_conv.npz"])
@pytest.mark.parametrize("mode", [0, 1])
def test_golden_fixtures(gp, golden, name, mode):
    g = golden(name)
    s = run_hip(gp, g["xyz1"], g["xyz2"], float(g["eps"]), int(g["iters"]), mode)
    np.testing.assert_array_equal(s["assignment"], g[f"assignment_m{mode}"])
    np.testing.assert_array_equal(s["dist"], g[f"dist_m{mode}"])
    np.testing.assert_array_equal(s["assignment_inv"], g[f"assignment_inv_m{mode}"])
    np.testing.assert_allclose(s["price"], g[f"price_m{mode}"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("shape,iters", [((1, 256), 1), ((1, 256), 2), ((3, 768), 17), ((2, 2048), 50),
                                         ((1, 4096), 50), ((16, 512), 25)])
@pytest.mark.parametrize("mode", [0, 1])
def test_vs_oracle(gp, oracle, shape, iters, mode):
    b_, n = shape
    a, b = gen_pair(40 + n, (b_, n, 3), (b_, n, 3), 0.0)
    s = run_hip(gp, a, b, 0.005, iters, mode)
    d, ass, st = oracle.emd_forward(a, b, 0.005, iters, mode, return_state=True)
    np.testing.assert_array_equal(s["assignment"], ass)
    np.testing.assert_array_equal(s["dist"], d)
    np.testing.assert_array_equal(s["assignment_inv"], st["assignment_inv"])
    np.testing.assert_array_equal(s["bid"], st["bid"])
    np.testing.assert_array_equal(s["bid_increments"], st["bid_increments"])


def test_settle_rounds_beyond_the_stamp_period(gp, oracle):
    """emd_settle_kernel (GetMax + Assign in one launch from the bidder chains) is taken for every round of a single
    cloud of more than 4096 points; a record carries 8 bits of the round, so the chain heads are cleared every 255 rounds:
    300 rounds at eps 0.001 (many bidders stay: chains of several bidders in late rounds too), every state array
    against the oracle; and a negative eps, which keeps the reference's two launches."""
    a, b = gen_pair(77, (1, 4352, 3), (1, 4352, 3), 0.0)
    for eps, iters in ((0.001, 300), (-0.0005, 6)):
        s = run_hip(gp, a, b, eps, iters, 1)
        d, ass, st = oracle.emd_forward(a, b, eps, iters, 1, return_state=True)
        np.testing.assert_array_equal(s["assignment"], ass)
        np.testing.assert_array_equal(s["dist"], d)
        np.testing.assert_array_equal(s["assignment_inv"], st["assignment_inv"])
        np.testing.assert_array_equal(s["bid"], st["bid"])
        np.testing.assert_array_equal(s["bid_increments"], st["bid_increments"])


def test_survey_scalars(gp, golden):
    """EMD 0.07579704/0.07364403 (seed 0) and 0.065699235
    (scan 01184), strict arithmetic, through Completionloss.emd_loss."""
    torch = gp["torch"]
    prev = gp["lib"].lib.genpc_set_arith(0)
    try:
        a, b = gen_pair(0, (2, 1024, 3), (2, 1024, 3), 0.0)
        d, ass = gp["mod"](torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), 0.005, 50)
        per = torch.sqrt(d).mean(1).cpu().numpy()
        assert abs(per[0] - 0.07579704) < 2e-8 and abs(per[1] - 0.07364403) < 2e-8
        assert [int(x.unique().numel()) for x in ass] == [979, 975]
        g = golden("scan01184_fps2048.npz")
        v = gp["CL"]("emd").get_loss(torch.from_numpy(g["partial"]).cuda(), torch.from_numpy(g["gt"]).cuda()).item()
        assert abs(v - 0.065699235) < 2e-8
    finally:
        gp["lib"].lib.genpc_set_arith(prev)


def test_input_checks(gp):
    torch = gp["torch"]
    z = lambda *s: torch.zeros(*s).cuda()
    with pytest.raises(AssertionError):
        gp["mod"](z(1, 256, 3), z(1, 512, 3), 0.005, 2)
    with pytest.raises(AssertionError):
        gp["mod"](z(1, 100, 3), z(1, 100, 3), 0.005, 2)
    # the C ABI itself reports -1 like emd_cuda.cu:236-249
    s = gp["alloc"](1, 100, 100, "cuda")
    rc = gp["emd"].forward(z(1, 100, 3), z(1, 100, 3), s["dist"], s["assignment"], s["price"], s["assignment_inv"],
                           s["bid"], s["bid_increments"], s["max_increments"], s["unass_idx"], s["unass_cnt"],
                           s["unass_cnt_sum"], s["cnt_tmp"], s["max_idx"], 0.005, 2)
    assert rc == -1


def test_backward_and_determinism(gp, oracle):
    torch = gp["torch"]
    a, b = gen_pair(77, (2, 1024, 3), (2, 1024, 3), 0.0)
    A = torch.from_numpy(a).cuda().requires_grad_(True)
    B = torch.from_numpy(b).cuda()
    d, ass = gp["mod"](A, B, 0.005, 30)
    g = np.random.default_rng(3).random(d.shape, dtype=np.float32)
    (d * torch.from_numpy(g).cuda()).sum().backward()
    e = oracle.emd_backward(a, b, g, ass.cpu().numpy())
    np.testing.assert_array_equal(A.grad.cpu().numpy(), e)
    d2, ass2 = gp["mod"](A.detach(), B, 0.005, 30)
    assert torch.equal(ass, ass2) and torch.equal(d.detach(), d2)


def test_full_size_properties(gp):
    """16384 points (BASELINE config 3 size), 50 rounds: dist is exactly the squared
    distance to the assigned point, indices in range, run-to-run identical, and the
    value is an upper-bounded lower estimate: >= the NN (Chamfer) lower bound."""
    torch = gp["torch"]
    from genpc_amd.loss_functions import chamfer_3DDist
    a, b = gen_pair(16384, (1, 16384, 3), (1, 16384, 3), 0.0)
    A, B = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d, ass = gp["mod"](A, B, 0.005, 50)
    d2, ass2 = gp["mod"](A, B, 0.005, 50)
    assert torch.equal(ass, ass2) and torch.equal(d, d2)
    assert int(ass.min()) >= 0 and int(ass.max()) < 16384
    pick = B[0][ass[0].long()]
    dd = A[0] - pick
    t = dd[:, 1] * dd[:, 1]
    t = torch.addcmul(t, dd[:, 0], dd[:, 0])       # not bit-exact by construction: tolerance
    np.testing.assert_allclose(d[0].cpu().numpy(), (dd * dd).sum(-1).cpu().numpy(), rtol=1e-5, atol=1e-9)
    nn1, _, _, _ = chamfer_3DDist()(A, B)
    assert bool((d >= nn1).all())


def test_extreme_ties_and_limits(gp, oracle):
    """Every object at the same place (all bids tie exactly: the reference's
    thread-major order decides every index), eps = 0, a single round, and the batch
    limit of emd_cuda.cu:241-244."""
    torch = gp["torch"]
    a, _ = gen_pair(3, (2, 512, 3), (2, 512, 3), 0.0)
    b = np.zeros_like(a) + np.float32(0.5)
    for iters in (1, 7):
        s = run_hip(gp, a, b, 0.005, iters, 1)
        d, ass = oracle.emd_forward(a, b, 0.005, iters, 1)
        np.testing.assert_array_equal(s["assignment"], ass)
        np.testing.assert_array_equal(s["dist"], d)
    a2, b2 = gen_pair(4, (1, 256, 3), (1, 256, 3), 0.0)
    s = run_hip(gp, a2, b2, 0.0, 5, 1)
    d, ass = oracle.emd_forward(a2, b2, 0.0, 5, 1)
    np.testing.assert_array_equal(s["assignment"], ass)
    z = lambda *sh: torch.zeros(*sh).cuda()
    st = gp["alloc"](513, 256, 256, "cuda")
    rc = gp["emd"].forward(z(513, 256, 3), z(513, 256, 3), st["dist"], st["assignment"], st["price"],
                           st["assignment_inv"], st["bid"], st["bid_increments"], st["max_increments"], st["unass_idx"],
                           st["unass_cnt"], st["unass_cnt_sum"], st["cnt_tmp"], st["max_idx"], 0.005, 2)
    assert rc == -1
    a3, b3 = gen_pair(5, (512, 256, 3), (512, 256, 3), 0.0)       # B = 512 is allowed
    s = run_hip(gp, a3, b3, 0.005, 3, 1)
    d, ass = oracle.emd_forward(a3, b3, 0.005, 3, 1)
    np.testing.assert_array_equal(s["assignment"], ass)


@pytest.mark.parametrize("scale,offset", [(100.0, 0.0), (30.0, 250.0), (1000.0, -500.0), (1e-3, 0.0)])
def test_unnormalised_clouds(gp, oracle, scale, offset):
    """Clouds far outside the unit cube (bid values of magnitude 1e2..1e3, where a fixed 2e-6
    slack in the bid pre-filter would be below one ulp): bids, increments, assignment and
    distances still bit-exact against the oracle."""
    a, b = gen_pair(77, (2, 1024, 3), (2, 1024, 3), 0.0)
    a = (a * np.float32(scale) + np.float32(offset)).astype(np.float32)
    b = (b * np.float32(scale) + np.float32(offset)).astype(np.float32)
    for iters in (3, 50):
        s = run_hip(gp, a, b, 0.005, iters, 1)
        d, ass, st = oracle.emd_forward(a, b, 0.005, iters, 1, return_state=True)
        np.testing.assert_array_equal(s["assignment"], ass)
        np.testing.assert_array_equal(s["dist"], d)
        np.testing.assert_array_equal(s["bid"], st["bid"])
        np.testing.assert_array_equal(s["bid_increments"], st["bid_increments"])


def test_emd_repeated_calls_on_the_same_buffers(gp, oracle):
    """Five calls
    on the SAME tensors with the inputs overwritten in place and the state re-initialised each time --
    every call's assignment and distances equal the oracle's for the data of that call."""
    torch = gp["torch"]
    b, n = 2, 1024
    x = torch.empty(b, n, 3, device="cuda")
    y = torch.empty(b, n, 3, device="cuda")
    s = gp["alloc"](b, n, n, x.device)
    init = {k: v.clone() for k, v in s.items()}
    for call in range(5):
        rng = np.random.default_rng(100 + call)
        xn = rng.random((b, n, 3), dtype=np.float32)
        yn = rng.random((b, n, 3), dtype=np.float32)
        x.copy_(torch.from_numpy(xn))
        y.copy_(torch.from_numpy(yn))
        for k in s:
            s[k].copy_(init[k])
        rc = gp["emd"].forward(x, y, s["dist"], s["assignment"], s["price"], s["assignment_inv"], s["bid"],
                               s["bid_increments"], s["max_increments"], s["unass_idx"], s["unass_cnt"],
                               s["unass_cnt_sum"], s["cnt_tmp"], s["max_idx"], 0.005, 30)
        assert rc == 1
        od, oa = oracle.emd_forward(xn, yn, 0.005, 30)[:2]
        np.testing.assert_array_equal(s["assignment"].cpu().numpy(), oa, err_msg="call %d" % call)
        np.testing.assert_array_equal(s["dist"].cpu().numpy(), od, err_msg="call %d" % call)


def test_emd_fuzz(gp, oracle):
    """16 random cases: B 1 .. 6, n 256 .. 3072, iters 1 .. 40, eps 1e-3 .. 2e-2, uniform / clustered / lattice /
    duplicated clouds at scales 0.1 .. 3, both arithmetic modes: assignment, assignment_inv and dist bit-exact."""
    rng = np.random.default_rng(4242)
    for case in range(16):
        b = int(rng.integers(1, 7))
        n = 256 * int(rng.integers(1, 13))
        iters = int(rng.integers(1, 41))
        eps = float(10.0 ** rng.uniform(-3, -1.7))
        kind = case % 4

        def cloud():
            if kind == 0:
                x = rng.random((b, n, 3))
            elif kind == 1:
                c = rng.random((b, 6, 3))
                x = np.stack([c[i][rng.integers(0, 6, n)] for i in range(b)]) + 0.03 * rng.normal(size=(b, n, 3))
            elif kind == 2:
                x = rng.integers(0, 9, size=(b, n, 3)) / 9.0
            else:
                base = rng.random((b, n // 4, 3))
                x = np.stack([base[i][rng.integers(0, n // 4, n)] for i in range(b)])
            return (x * rng.uniform(0.1, 3.0)).astype(np.float32)
        x, y = cloud(), cloud()
        mode = case & 1
        s = run_hip(gp, x, y, eps, iters, mode)
        d, ass, st = oracle.emd_forward(x, y, eps, iters, mode, return_state=True)
        np.testing.assert_array_equal(s["assignment"], ass, err_msg="case %d" % case)
        np.testing.assert_array_equal(s["dist"], d, err_msg="case %d" % case)
        np.testing.assert_array_equal(s["assignment_inv"], st["assignment_inv"], err_msg="case %d" % case)


def test_emd_elementwise_at_16384(gp, oracle):
    """BASELINE config 3's size, element by element: assignment, assignment_inv, bids and dist
    of one 16384-point pair after the reference's 50 rounds at eps 0.005 equal the oracle's, bit for bit."""
    rng = np.random.default_rng(163)
    x = rng.random((1, 16384, 3), dtype=np.float32)
    y = rng.random((1, 16384, 3), dtype=np.float32)
    got = run_hip(gp, x, y, 0.005, 50, 1)
    od, oass = oracle.emd_forward(x, y, 0.005, 50, 1)
    np.testing.assert_array_equal(got["assignment"], oass)
    np.testing.assert_array_equal(got["dist"], od)
    inv = np.full(16384, -1, np.int64)
    inv[oass[0]] = np.arange(16384)               # (the forced last round makes it many-to-one: compare where unique)
    uniq, cnt = np.unique(oass[0], return_counts=True)
    np.testing.assert_array_equal(got["assignment_inv"][0][uniq[cnt == 1]], inv[uniq[cnt == 1]])


def test_emd_partial_scan_vs_ground_truth_elementwise(gp, oracle, golden):
    """A bundled scan (8192-point subsample) against its ground truth: bidders on a surface, half of the objects far
    from every bidder -- prices climb and the culled bid's search boxes grow round after round (the opposite regime
    of uniform clouds); and a cloud squeezed into a plane and a line (grids with one cell on an axis)."""
    g = golden("scans13_fps16384.npz")
    for scan in (0, 7):
        x, y = g["partial"][scan][None, :8192].copy(), g["gt"][scan][None, :8192].copy()
        got = run_hip(gp, x, y, 0.005, 50, 1)
        od, oass, st = oracle.emd_forward(x, y, 0.005, 50, 1, return_state=True)
        np.testing.assert_array_equal(got["assignment"], oass)
        np.testing.assert_array_equal(got["dist"], od)
        np.testing.assert_array_equal(got["bid"], st["bid"])
        np.testing.assert_array_equal(got["bid_increments"], st["bid_increments"])
    rng = np.random.default_rng(9)
    for flat in ((1, 1, 0), (1, 0, 0), (0, 0, 0)):
        x = (rng.random((2, 1024, 3), dtype=np.float32) * np.array(flat, np.float32)).astype(np.float32)
        y = (rng.random((2, 1024, 3), dtype=np.float32) * np.array(flat, np.float32)).astype(np.float32)
        got = run_hip(gp, x, y, 0.005, 20, 1)
        od, oass = oracle.emd_forward(x, y, 0.005, 20, 1)
        np.testing.assert_array_equal(got["assignment"], oass, err_msg=str(flat))
        np.testing.assert_array_equal(got["dist"], od, err_msg=str(flat))


def test_emd_non_finite_clouds(gp, oracle):
    """A NaN / inf object: the culled bid searches such a cloud without culling (its grid has no meaning
    there); the results stay the oracle's."""
    rng = np.random.default_rng(12)
    x = rng.random((2, 1024, 3), dtype=np.float32)
    y = rng.random((2, 1024, 3), dtype=np.float32)
    y[0, 5, 1] = np.inf
    y[1, 100, 0] = np.nan           # (a non-finite BIDDER bids on object -1 in the reference: an out-of-bounds write, not tested)
    got = run_hip(gp, x, y, 0.005, 12, 1)
    od, oass = oracle.emd_forward(x, y, 0.005, 12, 1)
    np.testing.assert_array_equal(got["assignment"], oass)
    np.testing.assert_array_equal(got["dist"].view(np.uint32), od.view(np.uint32))


@pytest.mark.parametrize("iters", [1, 2, 9, 50])
def test_emd_crowded_objects(gp, oracle, iters):
    """Hundreds of bidders on a handful of objects, round after round (a partial scan whose ground truth is mis-framed:
    bundled scan 06830): the settle kernel handles such an object without walking its chain of bidders -- window test,
    atomicMax, ticket, the last ticket assigns -- also in a forced last round; every state array against the oracle."""
    rng = np.random.default_rng(50 + iters)
    b, n = 2, 2048
    x = (rng.random((b, n, 3), dtype=np.float32) * np.float32(0.05)).astype(np.float32)          # bidders: a small cluster
    y = (rng.random((b, n, 3), dtype=np.float32) + np.float32(2.0)).astype(np.float32)           # objects: far away ...
    y[:, :6] = rng.random((b, 6, 3), dtype=np.float32) * np.float32(0.05) + np.float32(0.1)      # ... but six next to the cluster
    y[1, 6:40] = y[1, 5] + rng.random((34, 3), dtype=np.float32) * np.float32(1e-4)              # and a knot of near-equal ones
    got = run_hip(gp, x, y, 0.005, iters, 1)
    d, ass, st = oracle.emd_forward(x, y, 0.005, iters, 1, return_state=True)
    np.testing.assert_array_equal(got["assignment"], ass)
    np.testing.assert_array_equal(got["dist"], d)
    np.testing.assert_array_equal(got["assignment_inv"], st["assignment_inv"])
    np.testing.assert_array_equal(got["bid"], st["bid"])
    np.testing.assert_array_equal(got["bid_increments"], st["bid_increments"])
    np.testing.assert_allclose(got["price"], st["price"], rtol=1e-5, atol=2e-6)
