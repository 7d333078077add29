"""Parity of the HIP Chamfer path (through the C ABI, via the reference-shaped
Python API) against the CPU oracle.  Bar: distances and indices BIT-EXACT in both
arithmetic modes; gradients within 1e-6 relative (atomic accumulation order is
unspecified in the reference as well); reductions within 2 ulp of the oracle's
fp32 means."""
import numpy as np
import pytest

from conftest import gen_pair

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from genpc_amd import _lib
    from genpc_amd.loss_functions import chamfer_3DDist
    from genpc_amd.utils.loss_util import Completionloss
    return dict(torch=torch, lib=_lib, cd=chamfer_3DDist(), CL=Completionloss)


def run_hip(gp, a, b, mode):
    torch = gp["torch"]
    prev = gp["lib"].lib.genpc_set_arith(mode)
    try:
        d1, d2, i1, i2 = gp["cd"](torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
        torch.cuda.synchronize()
    finally:
        gp["lib"].lib.genpc_set_arith(prev)
    return d1.cpu().numpy(), d2.cpu().numpy(), i1.cpu().numpy(), i2.cpu().numpy()


def assert_same(got, exp):
    for g, e, nme in zip(got, exp, ("dist1", "dist2", "idx1", "idx2")):
        np.testing.assert_array_equal(g, e, err_msg=nme)


@pytest.mark.parametrize("name", ["chamfer_seed1_b1_2048.npz", "chamfer_seed0_b2_1000x777.npz",
                                  "chamfer_seed7_b3_5x3.npz", "chamfer_seed11_dups.npz"])
@pytest.mark.parametrize("mode", [0, 1])
def test_golden_fixtures(gp, golden, name, mode):
    g = golden(name)
    got = run_hip(gp, g["xyz1"], g["xyz2"], mode)
    assert_same(got, [g[f"dist1_m{mode}"], g[f"dist2_m{mode}"], g[f"idx1_m{mode}"], g[f"idx2_m{mode}"]])


@pytest.mark.parametrize("shape", [
    ((1, 1, 3), (1, 1, 3)), ((1, 63, 3), (1, 65, 3)), ((2, 257, 3), (2, 31, 3)), ((3, 1000, 3), (3, 4097, 3)),
    ((1, 8192, 3), (1, 8192, 3)), ((5, 300, 3), (5, 5000, 3)), ((64, 512, 3), (64, 640, 3))])
@pytest.mark.parametrize("mode", [0, 1])
def test_vs_oracle_shapes(gp, oracle, shape, mode):
    a, b = gen_pair(21, *shape)
    assert_same(run_hip(gp, a, b, mode), oracle.chamfer_forward(a, b, mode))


def test_survey_scalars_through_completionloss(gp, oracle, golden):
    """BASELINE.md section 2 values, reproduced on the GPU in strict mode."""
    torch = gp["torch"]
    prev = gp["lib"].lib.genpc_set_arith(0)
    try:
        a, b = gen_pair(1, (1, 2048, 3), (1, 2048, 3))
        A, B = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        l1 = gp["CL"]("cd_l1").get_loss(A, B).item()
        l2 = gp["CL"]("cd_l2").get_loss(A, B).item()
        assert abs(l1 - 0.04512813) < 1e-8 and abs(l2 - 0.0046723103) < 1e-9
        g = golden("scan01184_fps2048.npz")
        P, G = torch.from_numpy(g["partial"]).cuda(), torch.from_numpy(g["gt"]).cuda()
        assert abs(gp["CL"]("cd_l1").get_loss(P, G).item() - 0.040335327) < 1e-8
        assert abs(gp["CL"]("cd_l2").get_loss(P, G).item() - 0.0099346815) < 1e-9
        cl = gp["CL"]("cd_l1")
        d1, _, _, _ = oracle.chamfer_forward(g["partial"], g["gt"], 0)
        assert abs(cl.chamfer_partial_l1(P, G).item() - float(oracle.cd_partial_l1(d1))) < 1e-8
        assert abs(gp["CL"]("cd_l2").chamfer_partial_l2(P, G).item() - float(oracle.cd_partial_l2(d1))) < 1e-9
    finally:
        gp["lib"].lib.genpc_set_arith(prev)


def test_duplicates_and_exact_hits(gp, oracle):
    a, b = gen_pair(5, (2, 700, 3), (2, 900, 3))
    b[:, 450:900] = b[:, 0:450]
    a[:, :100] = b[:, 200:300]
    got = run_hip(gp, a, b, 1)
    assert_same(got, oracle.chamfer_forward(a, b, 1))
    assert (got[0][:, :100] == 0).all() and (got[2] < 450).all()


def test_empty_clouds_leave_zeros(gp):
    torch = gp["torch"]
    d1, d2, i1, i2 = gp["cd"](torch.zeros(2, 0, 3).cuda(), torch.rand(2, 7, 3).cuda())
    assert d1.shape == (2, 0) and d2.shape == (2, 7) and float(d2.abs().sum()) == 0 and int(i2.abs().sum()) == 0


def test_noncontiguous_input_and_stream(gp, oracle):
    torch = gp["torch"]
    a, b = gen_pair(9, (2, 500, 3), (2, 300, 3))
    A = torch.from_numpy(np.ascontiguousarray(a.transpose(0, 2, 1))).cuda().transpose(1, 2)   # non-contiguous view
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d1, d2, i1, i2 = gp["cd"](A, torch.from_numpy(b).cuda())
    s.synchronize()
    assert_same([t.cpu().numpy() for t in (d1, d2, i1, i2)], oracle.chamfer_forward(a, b, 1))


def test_backward_vs_oracle(gp, oracle):
    torch = gp["torch"]
    a, b = gen_pair(31, (3, 900, 3), (3, 1100, 3))
    A = torch.from_numpy(a).cuda().requires_grad_(True)
    B = torch.from_numpy(b).cuda().requires_grad_(True)
    d1, d2, i1, i2 = gp["cd"](A, B)
    rng = np.random.default_rng(2)
    g1 = rng.random(d1.shape, dtype=np.float32)
    g2 = rng.random(d2.shape, dtype=np.float32)
    (d1 * torch.from_numpy(g1).cuda()).sum().add((d2 * torch.from_numpy(g2).cuda()).sum()).backward()
    e1, e2 = oracle.chamfer_backward(a, b, g1, g2, i1.cpu().numpy(), i2.cpu().numpy())
    np.testing.assert_allclose(A.grad.cpu().numpy(), e1, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(B.grad.cpu().numpy(), e2, rtol=1e-5, atol=1e-6)


def test_backward_large_call_forms(gp, oracle):
    """From 262144 points per call the backward is two launches: own rows by plain read-modify-write, then the scattered
    halves through LDS tiles that own their output rows (csrc/chamfer.hip: chamfer_grad_scatter_tiled_kernel) -- against the
    oracle, ragged sizes (a last tile of 904 rows, a cloud smaller than a tile), indices that pile up on few targets."""
    torch = gp["torch"]
    rng = np.random.default_rng(77)
    bsz, n, m = 24, 9096, 3000            # 24 x (9096 + 3000) = 290 k points
    a = (rng.random((bsz, n, 3), dtype=np.float32) - np.float32(0.5))
    b = (rng.random((bsz, m, 3), dtype=np.float32) - np.float32(0.5))
    i1 = rng.integers(0, m, (bsz, n)).astype(np.int32)
    i2 = rng.integers(0, 40, (bsz, m)).astype(np.int32)           # crowded: 3000 queries on 40 targets
    g1 = rng.random((bsz, n), dtype=np.float32)
    g2 = rng.random((bsz, m), dtype=np.float32)
    from genpc_amd import _lib
    L, p = _lib.lib, _lib.ptr
    A, B, G1, G2, I1, I2 = (torch.from_numpy(x).cuda() for x in (a, b, g1, g2, i1, i2))
    ga, gb = torch.zeros_like(A), torch.zeros_like(B)
    assert _lib.on_device_of(A, L.genpc_chamfer_backward, bsz, n, p(A), m, p(B), p(G1), p(I1), p(G2), p(I2), p(ga), p(gb)) == 1
    e1, e2 = oracle.chamfer_backward(a, b, g1, g2, i1, i2)
    np.testing.assert_allclose(ga.cpu().numpy(), e1, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(gb.cpu().numpy(), e2, rtol=2e-5, atol=2e-4)      # (75 terms per crowded row, summed in another order)
    # it accumulates into what the caller hands in, like the reference (a second call doubles the result)
    assert _lib.on_device_of(A, L.genpc_chamfer_backward, bsz, n, p(A), m, p(B), p(G1), p(I1), p(G2), p(I2), p(ga), p(gb)) == 1
    np.testing.assert_allclose(ga.cpu().numpy(), 2 * e1, rtol=1e-5, atol=4e-6)


def test_full_size_properties(gp):
    """BASELINE sizes (16384 and 32768 points): size-independent properties.
    (i) self-distance is exactly zero with idx = identity; (ii) a permutation of the
    targets permutes the indices and leaves distances bit-identical; (iii) splitting
    the targets in two and taking the element-wise min reproduces the joint result."""
    torch = gp["torch"]
    for n in (16384, 32768):
        a, b = gen_pair(n, (1, n, 3), (1, n, 3))
        A, B = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        d1, d2, i1, i2 = gp["cd"](A, A.clone())
        assert float(d1.abs().max()) == 0 and bool((i1[0] == torch.arange(n, device="cuda", dtype=torch.int32)).all())
        d1, d2, i1, i2 = gp["cd"](A, B)
        perm = torch.randperm(n, device="cuda")
        p1, p2, j1, j2 = gp["cd"](A, B[:, perm])
        assert torch.equal(p1, d1) and torch.equal(perm[j1[0].long()].int(), i1[0])
        h = n // 2 + 37
        l1, _, li, _ = gp["cd"](A, B[:, :h].contiguous())
        r1, _, ri, _ = gp["cd"](A, B[:, h:].contiguous())
        take_r = r1 < l1
        assert torch.equal(torch.where(take_r, r1, l1), d1)
        assert torch.equal(torch.where(take_r, ri + h, li), i1)


def test_more_query_blocks_than_the_resident_counter_area(gp, oracle):
    """300 batch elements x 8192 queries: > 16384 (batch, query block) units, i.e. more arrival
    counters than the 64 KiB that stay zero between calls (the scale search's 1000 candidates
    on dense clouds hit this); then a small call again."""
    a, b = gen_pair(5, (300, 8192, 3), (300, 64, 3))
    assert_same(run_hip(gp, a, b, 1), oracle.chamfer_forward(a, b, 1))
    a2, b2 = gen_pair(6, (2, 3000, 3), (2, 2500, 3))
    assert_same(run_hip(gp, a2, b2, 1), oracle.chamfer_forward(a2, b2, 1))


def test_degenerate_and_extreme_shapes(gp, oracle):
    """All points identical (every distance ties: index 0 must win everywhere), one
    query against a long cloud and vice versa, sizes around the tiling constants."""
    torch = gp["torch"]
    z = np.zeros((2, 300, 3), np.float32) + np.float32(0.25)
    d1, d2, i1, i2 = run_hip(gp, z, z[:, :77].copy(), 1)
    assert (d1 == 0).all() and (d2 == 0).all() and (i1 == 0).all() and (i2 == 0).all()
    for n, m in ((1, 70001), (70001, 1), (255, 2049), (2048, 2047), (513, 33)):
        a, b = gen_pair(n * 7 + m, (1, n, 3), (1, m, 3))
        assert_same(run_hip(gp, a, b, 1), oracle.chamfer_forward(a, b, 1))
    a, b = gen_pair(99, (1, 400, 3), (1, 500, 3))
    a *= np.float32(1e4)                       # large coordinates: same arithmetic, no overflow
    b *= np.float32(1e4)
    assert_same(run_hip(gp, a, b, 0), oracle.chamfer_forward(a, b, 0))


PATHS = {"valu": 0, "mfma32": 1, "f16": 3, "grid": 4}


def run_path(gp, a, b, mode, path, hooks=0):
    """Runs one nearest-neighbour kernel family (genpc_nn_tune) and restores the default."""
    lib = gp["lib"].lib
    prev = lib.genpc_nn_tune(path, hooks)
    try:
        return run_hip(gp, a, b, mode)
    finally:
        lib.genpc_nn_tune(prev, 0)


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("mode", [0, 1])
def test_every_kernel_family_is_bit_exact(gp, oracle, path, mode):
    """The VALU brute force, the two MFMA filters (fp32, two-piece f16) and the cell search
    must all return the oracle's bits: ragged sizes, several slices, unequal clouds."""
    for shape in (((2, 777, 3), (2, 4097, 3)), ((1, 5000, 3), (1, 130, 3)), ((3, 64, 3), (3, 64, 3))):
        a, b = gen_pair(77, *shape)
        assert_same(run_path(gp, a, b, mode, PATHS[path]), oracle.chamfer_forward(a, b, mode))


@pytest.mark.parametrize("path", ["mfma32", "f16"])
@pytest.mark.parametrize("hooks", [8, 16])
def test_filter_fallback_paths(gp, oracle, path, hooks):
    """Test hooks of the filtered paths: 8 sends every query through the exhaustive
    pass, 16 makes the finish step evaluate every listed tile -- same bits every way."""
    a, b = gen_pair(3, (2, 600, 3), (2, 1500, 3))
    assert_same(run_path(gp, a, b, 1, PATHS[path], hooks), oracle.chamfer_forward(a, b, 1))


@pytest.mark.parametrize("path", ["mfma32", "f16"])
def test_filter_adversarial_inputs(gp, oracle, path):
    """Inputs chosen against the filters' error bounds and the f16 path's scaling:
    exact ties on an integer grid (many tiles within the bound -> exhaustive pass, first
    index must win), clouds far from the origin and from each other, tiny and huge
    coordinate scales, a cloud that is a single repeated point plus one outlier."""
    rng = np.random.default_rng(12)
    grid = rng.integers(0, 6, size=(1, 3000, 3)).astype(np.float32)          # heavy exact ties
    assert_same(run_path(gp, grid[:, :1400].copy(), grid[:, 1400:].copy(), 1, PATHS[path]),
                oracle.chamfer_forward(grid[:, :1400].copy(), grid[:, 1400:].copy(), 1))
    a, b = gen_pair(41, (1, 900, 3), (1, 1300, 3))
    for scale, off_a, off_b in ((1.0, 1000.0, 1000.0), (1.0, 0.0, 300.0), (1e-18, 0.0, 0.0), (1e12, 0.0, 0.0),
                                (1e-3, 5.0, 5.0)):
        aa = (a * np.float32(scale) + np.float32(off_a)).astype(np.float32)
        bb = (b * np.float32(scale) + np.float32(off_b)).astype(np.float32)
        assert_same(run_path(gp, aa, bb, 1, PATHS[path]), oracle.chamfer_forward(aa, bb, 1))
    c = np.zeros((1, 500, 3), np.float32) + np.float32(0.3)
    c[0, 317] = (7.0, -2.0, 1.0)
    assert_same(run_path(gp, a, c, 1, PATHS[path]), oracle.chamfer_forward(a, c, 1))
    assert_same(run_path(gp, c, a, 1, PATHS[path]), oracle.chamfer_forward(c, a, 1))


@pytest.mark.parametrize("path", list(PATHS))
def test_non_finite_points(gp, oracle, path):
    """NaN / inf coordinates: the reference's strict '<' never selects a NaN distance and keeps
    target 0 when every distance is NaN; every kernel family must agree with the oracle
    (indices exactly; distances equal or both NaN)."""
    a, b = gen_pair(8, (1, 700, 3), (1, 900, 3))
    a[0, 13] = np.nan                    # a query that sees only NaN distances
    b[0, 5, 1] = np.nan                  # a target nobody can select
    b[0, 77] = np.inf                    # an infinitely far target
    got = run_path(gp, a, b, 1, PATHS[path])
    exp = oracle.chamfer_forward(a, b, 1)
    for g, e in zip(got[2:], exp[2:]):
        np.testing.assert_array_equal(g, e)
    for g, e in zip(got[:2], exp[:2]):
        assert np.array_equal(g, e, equal_nan=True)
