"""Element-wise parity of the DEFAULT nearest-neighbour dispatch at the benchmark sizes, a
randomized differential run of the f16-MFMA filter path against the brute-force VALU path
(no filter, no error bound: a different algorithm), and the reference's tile semantics for
non-finite input.  Bar everywhere: distances and indices bit-exact."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import ROOT, gen_pair

pytestmark = pytest.mark.gpu

PATHS = {"valu": 0, "mfma32": 1, "f16": 3, "grid": 4}
HOOK_COUNT = 512


@pytest.fixture(scope="module")
def gp():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from genpc_amd import _lib
    from genpc_amd.loss_functions import chamfer_3DDist
    return dict(torch=torch, lib=_lib.lib, cd=chamfer_3DDist())


def run_hip(gp, a, b, mode=1, path=None, hooks=0):
    """Default dispatch (path None) or one kernel family; returns numpy (d1, d2, i1, i2)."""
    torch, lib = gp["torch"], gp["lib"]
    prev_mode = lib.genpc_set_arith(mode)
    prev_path = lib.genpc_nn_tune(path if path is not None else -1, hooks) if (path is not None or hooks) else None
    try:
        out = gp["cd"](torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
        torch.cuda.synchronize()
    finally:
        lib.genpc_set_arith(prev_mode)
        if prev_path is not None:
            lib.genpc_nn_tune(prev_path, 0)
    return [t.cpu().numpy() for t in out]


def read_stats(gp, reset=True):
    buf = (ctypes.c_ulonglong * 3)()
    assert gp["lib"].genpc_nn_stats(ctypes.cast(buf, ctypes.c_void_p), 1 if reset else 0, None) == 1
    return [int(v) for v in buf]


def assert_bits(got, exp, msg=""):
    for g, e, nme in zip(got, exp, ("dist1", "dist2", "idx1", "idx2")):
        g, e = np.ascontiguousarray(g), np.ascontiguousarray(e)
        np.testing.assert_array_equal(g.view(np.uint32), e.view(np.uint32), err_msg="%s %s" % (nme, msg))


@pytest.mark.parametrize("mode", [0, 1])
def test_bench_input_elementwise(gp, oracle, mode):
    """The exact input bench.py times (default_rng(20250101), B=1, 16384 x 16384) through the
    default dispatch (f16-MFMA filter + finish), every distance and index against the oracle."""
    a, b = gen_pair(20250101, (1, 16384, 3), (1, 16384, 3))
    assert_bits(run_hip(gp, a, b, mode), oracle.chamfer_forward(a, b, mode))


@pytest.mark.parametrize("shape", [((1, 32768, 3), (1, 32768, 3)), ((2, 32768, 3), (2, 16384, 3))])
def test_config5_size_elementwise(gp, oracle, shape):
    """C5 cloud size (32768 points), element-wise, default dispatch."""
    a, b = gen_pair(32768, *shape)
    assert_bits(run_hip(gp, a, b, 1), oracle.chamfer_forward(a, b, 1))


@pytest.mark.parametrize("shape", [((1, 12001, 3), (1, 13003, 3)), ((4, 15403, 3), (4, 7855, 3)), ((2, 5000, 3), (2, 20011, 3)),
                                   ((1, 16384, 3), (1, 3001, 3)), ((3, 9000, 3), (3, 9000, 3))])
def test_single_round_filter_forms_elementwise(gp, oracle, shape):
    """Shapes whose launches are a single round with slices over 1024 targets: the filter's 2048-target LDS tiles
    (slice resident or not), the 8-wave blocks whose last block is partly empty (query counts that are no multiple of
    1024), the alignment loop's four starts x (15403 vs 7855), and the finish kernel's best-unit speculation on all of them."""
    a, b = gen_pair(sum(shape[0]) + sum(shape[1]), *shape)
    assert_bits(run_hip(gp, a, b, 1), oracle.chamfer_forward(a, b, 1), str(shape))


def test_thirteen_real_scans_elementwise(gp, oracle, golden):
    """C3 input (13 bundled scans at 16384 points) in one batched call, element-wise."""
    g = golden("scans13_fps16384.npz")
    assert_bits(run_hip(gp, g["partial"], g["gt"], 1), oracle.chamfer_forward(g["partial"], g["gt"], 1))


# ---------------------------------------------------------------------------
# differential fuzz
def _cloud(rng, kind, n, scans):
    f = np.float32
    if kind == "uniform":
        p = rng.random((n, 3), dtype=f) - f(0.5)
    elif kind == "clustered":
        k = int(rng.integers(1, 9))
        c = rng.random((k, 3), dtype=f) - f(0.5)
        p = c[rng.integers(0, k, n)] + rng.standard_normal((n, 3)).astype(f) * f(10.0 ** rng.uniform(-5, -1))
    elif kind == "planar":
        p = rng.random((n, 3), dtype=f) - f(0.5)
        p[:, int(rng.integers(0, 3))] = f(rng.uniform(-0.5, 0.5)) if rng.random() < 0.5 else p[:, 0] * f(1e-4)
    elif kind == "line":
        t = rng.random((n, 1), dtype=f)
        p = t * (rng.random((1, 3), dtype=f) - f(0.5)) + f(0.1)
    elif kind == "grid":
        g = int(rng.integers(2, 40))
        p = rng.integers(0, g, size=(n, 3)).astype(f) / f(g)          # many exact ties
    elif kind == "dups":
        base = rng.random((max(1, n // int(rng.integers(2, 6))), 3), dtype=f) - f(0.5)
        p = base[rng.integers(0, base.shape[0], n)]
    elif kind == "scan":
        s = scans[int(rng.integers(0, scans.shape[0]))]
        p = s[rng.integers(0, s.shape[0], n)] if n > s.shape[0] else s[rng.permutation(s.shape[0])[:n]]
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(p, dtype=f)


KINDS = ["uniform", "clustered", "planar", "line", "grid", "dups", "scan"]


def _fuzz_case(rng, scans_p, scans_g):
    ka, kb = KINDS[int(rng.integers(0, len(KINDS)))], KINDS[int(rng.integers(0, len(KINDS)))]
    if rng.random() < 0.5:
        kb = ka
    big = rng.random() < 0.25
    hi = np.log(40000.0) if big else np.log(6000.0)
    n, m = int(np.exp(rng.uniform(0, hi))), int(np.exp(rng.uniform(0, hi)))
    bsz = 1 if max(n, m) > 8000 else int(rng.integers(1, 4))
    scale = np.float32(10.0 ** rng.uniform(-6, 6)) if rng.random() < 0.7 else np.float32(1.0)
    off = (rng.random(3, dtype=np.float32) - np.float32(0.5)) * scale * np.float32(10.0 ** rng.uniform(-1, 3)) \
        if rng.random() < 0.5 else np.zeros(3, np.float32)
    sep = (rng.random(3, dtype=np.float32) - np.float32(0.5)) * scale * np.float32(rng.choice([0.0, 0.0, 0.3, 3.0, 100.0]))
    a = np.stack([_cloud(rng, ka, n, scans_p) for _ in range(bsz)]) * scale + off
    b = np.stack([_cloud(rng, kb, m, scans_g) for _ in range(bsz)]) * scale + off + sep
    return ka, kb, np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)


def test_differential_fuzz_f16_vs_valu(gp, oracle, golden):
    """>= 300 random cases: the f16-MFMA filter path (forced, also below the size where the
    planner would pick it) must return the VALU brute force's bits; every tenth small case is
    also checked against the oracle, and the default dispatch against both.  Counters (hook 512)
    record how often the filter's proof failed and the exhaustive pass ran."""
    g = golden("scans13_fps16384.npz")
    rng = np.random.default_rng(777)
    per_kind = {}
    ncases = 320
    for case in range(ncases):
        ka, kb, a, b = _fuzz_case(rng, g["partial"], g["gt"])
        mode = case & 1
        ref = run_hip(gp, a, b, mode, PATHS["valu"])
        read_stats(gp)
        got = run_hip(gp, a, b, mode, PATHS["f16"], HOOK_COUNT)
        st = read_stats(gp)
        msg = "case %d %s/%s %s x %s mode %d" % (case, ka, kb, a.shape, b.shape, mode)
        assert_bits(got, ref, msg)
        assert st[0] == a.shape[0] * (a.shape[1] + b.shape[1]), (msg, st)
        acc = per_kind.setdefault("%s/%s" % (ka, kb), [0, 0, 0, 0])
        for i in range(3):
            acc[i] += st[i]
        acc[3] += 1
        assert_bits(run_hip(gp, a, b, mode), ref, "default dispatch, " + msg)
        assert_bits(run_hip(gp, a, b, mode, PATHS["grid"]), ref, "cell-sorted search, " + msg)
        if case % 10 == 0 and a.shape[0] * a.shape[1] * b.shape[1] <= 6e7:
            assert_bits(ref, oracle.chamfer_forward(a, b, mode), "valu vs oracle, " + msg)
    tot = [sum(v[i] for v in per_kind.values()) for i in range(3)]
    # Generators without exact ties essentially never need the exhaustive pass (the others do by
    # construction: grids, repeated points, scans sampled with replacement, clusters so tight that
    # fp32 quantisation repeats points).
    clean = [k for k in per_kind if k == "uniform/uniform"]
    cq = sum(per_kind[k][0] for k in clean)
    cx = sum(per_kind[k][1] for k in clean)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    ok = cq > 0 and cx <= 1e-3 * cq
    with open(os.path.join(out, "nn_fuzz_stats.json"), "w") as f:
        json.dump({"cases": ncases, "queries": tot[0], "exhaustive_queries": tot[1], "exact_pieces": tot[2],
                   "tie_free_generators": {"queries": cq, "exhaustive_queries": cx},
                   "per_generator_pair": {k: dict(queries=v[0], exhaustive=v[1], pieces=v[2], cases=v[3])
                                          for k, v in sorted(per_kind.items())}}, f, indent=1)
    assert ok, (cq, cx)


def test_filter_counters_on_the_bench_input(gp):
    """How often the proof fails on the benchmark input: no exhaustive pass at all, about one
    exact 16-target piece set per query."""
    a, b = gen_pair(20250101, (1, 16384, 3), (1, 16384, 3))
    read_stats(gp)
    run_hip(gp, a, b, 1, PATHS["f16"], HOOK_COUNT)
    q, ex, pieces = read_stats(gp)
    assert q == 32768 and ex == 0 and pieces < 6 * q, (q, ex, pieces)


# ---------------------------------------------------------------------------
@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("mode", [0, 1])
def test_non_finite_tile_semantics(gp, oracle, path, mode):
    """chamfer3D.cu:30-36,126: a NaN distance at the first target of a 512-tile drops the
    tile (tile 0: the result is NaN at index 0); elsewhere it drops one target; inf - inf is
    a NaN too.  Every kernel family against the oracle's tiled restatement: indices exactly,
    distances equal or both NaN."""
    a, b = gen_pair(8, (2, 700, 3), (2, 2300, 3))
    a[0, 13] = np.nan
    a[0, 14, 2] = np.inf
    a[0, 15, 0] = np.inf
    b[0, 5, 1] = np.nan
    b[0, 512] = np.nan                 # tile 1 head
    b[0, 1024, 0] = np.inf             # tile 2 head at infinity: alive, except for query 15 (inf - inf)
    b[0, 1100, 0] = np.inf
    b[1, 2047, 2] = np.nan             # last target of tile 3: one target dropped
    b[1, 2048, 2] = np.nan             # tile 4 head (ragged tile of 252)
    # queries sitting on targets of the dead tiles: they must not find them
    a[0, 100:140] = b[0, 600:640]
    a[1, 100:140] = b[1, 2100:2140]
    exp = oracle.chamfer_forward(a, b, mode)
    assert not ((exp[2][0] >= 512) & (exp[2][0] < 1024)).any() and not (exp[2][1] >= 2048).any()
    got = run_hip(gp, a, b, mode, PATHS[path])
    for gg, e in zip(got[2:], exp[2:]):
        np.testing.assert_array_equal(gg, e)
    for gg, e in zip(got[:2], exp[:2]):
        assert np.array_equal(gg, e, equal_nan=True)
    # tile 0 head NaN: every query of that batch element ends with (NaN, 0)
    b0 = b.copy()
    b0[1, 0, 0] = np.nan
    exp = oracle.chamfer_forward(a, b0, mode)
    assert np.isnan(exp[0][1]).all() and (exp[2][1] == 0).all()
    got = run_hip(gp, a, b0, mode, PATHS[path])
    for gg, e in zip(got[2:], exp[2:]):
        np.testing.assert_array_equal(gg, e)
    for gg, e in zip(got[:2], exp[:2]):
        assert np.array_equal(gg, e, equal_nan=True)


def test_non_finite_default_dispatch_large(gp, oracle):
    """Same semantics through the default dispatch at a size that takes the two-launch path."""
    a, b = gen_pair(18, (1, 4000, 3), (1, 5000, 3))
    b[0, 1536, 1] = np.nan
    b[0, 4608] = np.inf
    a[0, 7] = np.nan
    a[0, 2000:2040] = b[0, 1600:1640]
    exp = oracle.chamfer_forward(a, b, 1)
    got = run_hip(gp, a, b, 1)
    for gg, e in zip(got[2:], exp[2:]):
        np.testing.assert_array_equal(gg, e)
    for gg, e in zip(got[:2], exp[:2]):
        assert np.array_equal(gg, e, equal_nan=True)


@pytest.mark.parametrize("mode", [0, 1])
def test_radius_limited_search(gp, oracle, golden, mode):
    """genpc_nm_distance_within: inside the limit the reference's bits, outside (+inf, -1) --
    on a real scan against its ground truth (many queries far from every target), on random
    clouds, with limits below, around and far above the cell size, 0 and +inf."""
    import torch
    from genpc_amd import chamfer_3D
    g = golden("scans13_fps16384.npz")
    cases = [(g["gt"][3:4], g["partial"][3:4, :6000].copy())]
    a, b = gen_pair(91, (2, 5000, 3), (2, 3000, 3))
    cases.append((a, b))
    lib = gp["lib"]
    prev = lib.genpc_set_arith(mode)
    try:
        for q, t in cases:
            ed, _, ei, _ = oracle.chamfer_forward(q, t, mode)
            Q, T = torch.from_numpy(np.ascontiguousarray(q)).cuda(), torch.from_numpy(np.ascontiguousarray(t)).cuda()
            for r2 in (0.0, 1e-6, 1e-4, 3e-3, 0.05, 10.0, float("inf"), float(np.median(ed))):
                d = torch.empty(q.shape[0], q.shape[1], device="cuda")
                i = torch.empty(q.shape[0], q.shape[1], device="cuda", dtype=torch.int32)
                assert chamfer_3D.nm_distance_within(Q, T, r2, d, i) == 1
                inside = ed <= np.float32(r2)
                np.testing.assert_array_equal(i.cpu().numpy(), np.where(inside, ei, -1))
                np.testing.assert_array_equal(d.cpu().numpy(), np.where(inside, ed, np.float32(np.inf)))
    finally:
        lib.genpc_set_arith(prev)
    assert lib.genpc_nm_distance_within(1, 4, None, 4, None, -1.0, None, None, None) == -1
