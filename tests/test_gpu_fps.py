"""GPU farthest point sampling vs the oracle: index sequences bit-exact in both
arithmetic modes, one and several workgroups per cloud, several clouds per launch."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fg():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from genpc_amd import _lib
    from genpc_amd.fps import fps_sampling, fps_subsample
    return dict(torch=torch, lib=_lib, fps=fps_sampling, sub=fps_subsample)


@pytest.mark.parametrize("n,k,c", [(1000, 100, 1), (5000, 2048, 3), (40000, 4096, 2), (165546, 2000, 1)])
@pytest.mark.parametrize("mode", [0, 1])
def test_fps_matches_oracle(fg, oracle, n, k, c, mode):
    torch = fg["torch"]
    rng = np.random.default_rng(n + c)
    x = (rng.random((c, n, 3), dtype=np.float32) - 0.5).astype(np.float32)
    x[:, 10:20] = x[:, 0:10]                      # duplicates: arg-max ties
    prev = fg["lib"].lib.genpc_set_arith(mode)
    try:
        idx = fg["fps"](torch.from_numpy(x).cuda(), k).cpu().numpy()
    finally:
        fg["lib"].lib.genpc_set_arith(prev)
    for i in range(c):
        np.testing.assert_array_equal(idx[i], oracle.fps(x[i], k, mode))
    assert idx.shape == (c, k) and (idx[:, 0] == 0).all()


def test_fps_reproduces_the_scan_fixture(fg, golden, oracle):
    """The committed 2048-point fixture of scan 01184 was subsampled by the oracle;
    subsampling the fixture itself again (k = N) must return a permutation, and a
    full-size run is timed for the record."""
    torch = fg["torch"]
    g = golden("scan01184_fps2048.npz")
    P = torch.from_numpy(g["partial"][0]).cuda()
    idx = fg["fps"](P, 2048).cpu().numpy()
    assert sorted(idx.tolist()) == list(range(2048))
    np.testing.assert_array_equal(idx, oracle.fps(g["partial"][0], 2048, 1))
    sub = fg["sub"](P[None], 256)
    assert sub.shape == (1, 256, 3)
    big = torch.rand(4, 165546, 3, device="cuda")
    fg["fps"](big, 16)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fg["fps"](big, 16384)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("FPS 4 x 165546 -> 16384: %.1f ms" % (dt * 1e3))
    assert out.shape == (4, 16384) and int(out.max()) < 165546


def test_fps_one_workgroup_kernel_in_the_pipelines_regime(fg, oracle, golden):
    """csrc/fps_grid.hip (clouds of up to 24576 points: one workgroup, spatially pruned updates, several samples per round) where
    the pipeline uses it -- most of the cloud is sampled (24576 -> 20000, 20000 -> 16384 on a bundled scan's surface) -- and on
    the inputs that stress its rules: exact ties by the thousand (a lattice: the tie path), coordinates whose spacing is a few
    ulps (far from the origin), a plane (one layer of cells), one repeated point (the running maximum is 0 after the first
    sample), k == n.  Every sequence equals the oracle's AND the multi-workgroup kernel's (genpc_fps_tune bit 256)."""
    torch = fg["torch"]
    L = fg["lib"].lib
    g = golden("scans13_fps16384.npz")
    rng = np.random.default_rng(3)
    surf = np.concatenate([g["partial"][0][:8192], g["gt"][0]]).astype(np.float32)
    cases = [("scan surface 24576 -> 20000", surf, 20000, 1), ("scan surface 20000 -> 16384", surf[:20000].copy(), 16384, 0),
             ("uniform 8192 -> 8192", (rng.random((8192, 3), dtype=np.float32) - 0.5), 8192, 1),
             ("lattice 12^3 x 20000 -> 3000", (rng.integers(0, 12, size=(20000, 3)) / 12.0 - 0.5).astype(np.float32), 3000, 1),
             ("far from origin 9000 -> 9000", (rng.random((9000, 3), dtype=np.float32) * 0.01 + 1000.0).astype(np.float32), 9000, 0),
             ("plane 5000 -> 5000", np.concatenate([rng.random((5000, 2), dtype=np.float32), np.zeros((5000, 1), np.float32)], 1), 5000, 1),
             ("one point 300 times", np.ones((300, 3), np.float32), 300, 1), ("five points", rng.random((5, 3), dtype=np.float32), 5, 0)]
    for name, x, k, mode in cases:
        X = torch.from_numpy(x).cuda()
        prev_a = L.genpc_set_arith(mode)
        try:
            got = fg["fps"](X, k).cpu().numpy()
            prev = L.genpc_fps_tune(256)
            try:
                old = fg["fps"](X, k).cpu().numpy()
            finally:
                L.genpc_fps_tune(prev)
        finally:
            L.genpc_set_arith(prev_a)
        np.testing.assert_array_equal(got, old, err_msg=name + " (against the multi-workgroup kernel)")
        np.testing.assert_array_equal(got, oracle.fps(x, k, mode), err_msg=name + " (against the oracle)")


def test_fps_check_beside_the_callers_stream(fg):
    """genpc_fps_defer: the indices leave without waiting for the device-side check, which runs on copies on a side stream;
    genpc_fps_deferred_check collects the verdict.  Same indices as the in-line form; a check that fails (hook 2 zeroes one
    recorded minimum of the check's copy) is counted once and the count is then clear."""
    torch = fg["torch"]
    from genpc_amd import _lib
    L = _lib.lib
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    pts = torch.rand(20000, 3, device="cuda", generator=g)
    ref = fg["fps"](pts, 6000)
    prev = L.genpc_fps_defer(1)
    try:
        for _ in range(6):          # (more calls than the ring of check buffers holds)
            got = fg["fps"](pts, 6000)
            assert torch.equal(got, ref)
        assert _lib.on_device_of(pts, L.genpc_fps_deferred_check) == 0
        L.genpc_fps_defer(2)
        got = fg["fps"](pts, 6000)
        assert torch.equal(got, ref)          # (the sampling itself is untouched)
        assert _lib.on_device_of(pts, L.genpc_fps_deferred_check) == 1
        assert _lib.on_device_of(pts, L.genpc_fps_deferred_check) == 0
        # clouds of the many-workgroup kernel keep the in-line check
        big = torch.rand(40000, 3, device="cuda", generator=g)
        L.genpc_fps_defer(1)
        assert int(fg["fps"](big, 512)[0]) == 0
        assert _lib.on_device_of(pts, L.genpc_fps_deferred_check) == 0
    finally:
        L.genpc_fps_defer(prev)


def test_fps_bad_args(fg):
    torch = fg["torch"]
    with pytest.raises(ValueError):
        fg["fps"](torch.rand(10, 3).cuda(), 11)


def test_fps_fuzz(fg, oracle):
    """24 random cases: 1 .. 90000 points (one to eleven workgroups per cloud), k from 1 to n, 1 .. 5 clouds,
    lattices and repeated points (arg-max ties: the first index wins), scales 1e-3 .. 1e3, both arithmetic modes."""
    torch = fg["torch"]
    rng = np.random.default_rng(808)
    for case in range(24):
        n = int(rng.integers(1, 90001)) if case % 3 else int(rng.integers(1, 300))
        k = int(rng.integers(1, min(n, 3000) + 1))
        c = int(rng.integers(1, 6)) if n < 30000 else 1
        kind = case % 3
        if kind == 0:
            x = rng.random((c, n, 3)) - 0.5
        elif kind == 1:
            g = rng.integers(0, 12, size=(c, n, 3)).astype(np.float64) / 12           # lattice: many exact ties
            x = g - 0.5
        else:
            base = rng.random((c, max(1, n // 5), 3)) - 0.5
            x = np.stack([base[i][rng.integers(0, base.shape[1], n)] for i in range(c)])
        x = (x * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
        mode = case & 1
        prev = fg["lib"].lib.genpc_set_arith(mode)
        try:
            idx = fg["fps"](torch.from_numpy(x).cuda(), k).cpu().numpy()
        finally:
            fg["lib"].lib.genpc_set_arith(prev)
        for i in range(c):
            np.testing.assert_array_equal(idx[i], oracle.fps(x[i], k, mode), err_msg="case %d n %d k %d" % (case, n, k))


def test_fps_multi_ragged(fg, oracle):
    """genpc_fps_multi: clouds of different sizes and sample counts in one launch (1 .. 64 workgroups per
    cloud, k == n included) give the oracle's sequences; same result as one call per cloud."""
    torch = fg["torch"]
    from genpc_amd.fps import fps_sampling_multi
    rng = np.random.default_rng(77)
    shapes = [(172000, 1500), (16384, 16384), (19000, 3000), (300, 300), (2049, 100), (1, 1)]
    clouds = [(rng.random((n, 3), dtype=np.float32) - 0.5).astype(np.float32) for n, _ in shapes]
    clouds[2][100:200] = clouds[2][0:100]                    # duplicates: ties
    got = fps_sampling_multi([torch.from_numpy(c).cuda() for c in clouds], [k for _, k in shapes])
    for (n, k), c, g in zip(shapes, clouds, got):
        np.testing.assert_array_equal(g.cpu().numpy(), oracle.fps(c, k, 1), err_msg="n %d k %d" % (n, k))
        np.testing.assert_array_equal(fg["fps"](torch.from_numpy(c).cuda(), k).cpu().numpy(), g.cpu().numpy())
    assert sorted(got[1].cpu().numpy().tolist()) == list(range(16384))      # k == n: a permutation


def test_fps_combiner_gives_each_thread_its_own_sequences(fg):
    """fps.FpsCombiner: samplings of several host threads (a stream each) leave in shared launches; every thread gets the
    sequences a call of its own gives."""
    import threading
    torch = fg["torch"]
    from genpc_amd.fps import fps_sampling_multi, FpsCombiner
    rng = np.random.default_rng(5)
    jobs = [[(rng.random((n, 3), dtype=np.float32) - 0.5) for n in sizes] for sizes in ((9000, 300), (20000,), (4096, 4096, 77), (12000,), (700, 15000))]
    ks = [[min(len(c), 2500) for c in cl] for cl in jobs]
    dev_jobs = [[torch.from_numpy(c).cuda() for c in cl] for cl in jobs]
    want = [[o.cpu().numpy() for o in fps_sampling_multi(cl, k)] for cl, k in zip(dev_jobs, ks)]
    torch.cuda.synchronize()
    got, errs = [None] * len(jobs), []

    def run(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                for _ in range(3):                       # (several rounds: requests queue up behind launches in flight)
                    outs = fps_sampling_multi(dev_jobs[i], ks[i])
                    single = fg["fps"](dev_jobs[i][0], ks[i][0])
                torch.cuda.current_stream().synchronize()
                got[i] = [o.cpu().numpy() for o in outs] + [single.cpu().numpy()]
        except BaseException as e:
            errs.append(e)

    with FpsCombiner.installed("cuda") as comb:
        th = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert comb.clouds == 3 * sum(len(j) + 1 for j in jobs) and 1 <= comb.launches <= 3 * 2 * len(jobs)
    assert FpsCombiner._current is None
    for w, g in zip(want, got):
        for a, b in zip(w, g[:-1]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(w[0], g[-1])


def test_fps_step_time(fg):
    """Recorded, not asserted tightly: microseconds per sequential step (round 2: 2.8 at 4 x 165546)."""
    torch = fg["torch"]
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    for c, n, k in ((4, 165546, 16384), (1, 172000, 20000), (1, 16384, 16384), (1, 8192, 8192)):
        x = torch.rand(c, n, 3, device="cuda", generator=g)
        fg["fps"](x, 64)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fg["fps"](x, k)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) * 1e6 / k
        import ctypes
        rounds = (ctypes.c_int * 1)()
        fg["lib"].on_device_of(x, fg["lib"].lib.genpc_fps_stats, 1, ctypes.addressof(rounds))
        print("fps %d x %d -> %d: %.3f us/step, %.1f samples per exchange" % (c, n, k, us, k / max(1, rounds[0])))
        assert us < 5.0


def test_build_on_the_gpu_box_and_device_sized_launches():
    """The library is compiled WHERE IT RUNS (genpc_amd.build(force=True) into a scratch directory: hipcc, gfx950), and
    the launch planners size their grids from the device's CU count, not from a constant."""
    import ctypes
    import subprocess
    import sys
    import tempfile
    import torch
    from conftest import ROOT
    with tempfile.TemporaryDirectory() as tmp:
        code = ("import sys; sys.path.insert(0, %r); from genpc_amd import build as B; import os; "
                "B.LIBDIR = %r; B.LIB = os.path.join(B.LIBDIR, 'libgenpc_hip.so'); print(B.build(force=True, verbose=False))" % (ROOT, tmp))
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1500)
        assert p.returncode == 0, p.stderr[-2000:]
        so = p.stdout.strip().splitlines()[-1]
        lib = ctypes.CDLL(so)
        from genpc_amd import _lib
        assert lib.genpc_abi_version() == _lib.ABI_VERSION
    assert torch.cuda.get_device_properties(0).multi_processor_count >= 1
