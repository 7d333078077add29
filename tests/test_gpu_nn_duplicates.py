"""Clouds with exact duplicates -- resampled with replacement (SURVEY 8d's C5 generator), pad-repeated to a fixed
size (the Waymo crops of C4, main.py:21-24) -- through the f16 filter with its duplicate pre-pass (csrc/nn_dedupe.hip).
The mask against numpy; distances and indices bit-exact against the oracle with the pre-pass forced on, forced off
and left to the policy; the exhaustive pass's share with and without it."""
import ctypes

import numpy as np
import pytest

from test_gpu_chamfer_parity import assert_bits, read_stats, run_hip

pytestmark = pytest.mark.gpu

HOOK_COUNT, HOOK_OFF, HOOK_ON = 512, 2048, 4096


@pytest.fixture(scope="module")
def gp():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from genpc_amd import _lib
    from genpc_amd.loss_functions import chamfer_3DDist
    return dict(torch=torch, lib=_lib.lib, cd=chamfer_3DDist())


def later_copies(x):
    """numpy: point k is a later copy iff an earlier point of the same cloud has the same BITS."""
    out = np.zeros(x.shape[:2], bool)
    for e in range(x.shape[0]):
        keys = np.ascontiguousarray(x[e]).view(np.uint32).reshape(-1, 3)
        _, first, inv = np.unique(keys, axis=0, return_index=True, return_inverse=True)
        out[e] = first[inv.reshape(-1)] != np.arange(x.shape[1])
    return out


def gpu_mask(gp, x):
    torch = gp["torch"]
    b, n, _ = x.shape
    X = torch.from_numpy(x).cuda()
    nw = (n + 31) // 32
    M = torch.full((b, nw), -1, dtype=torch.int32, device="cuda")
    assert gp["lib"].genpc_nn_duplicate_mask(b, n, ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(M.data_ptr()), None) == 1
    torch.cuda.synchronize()
    bits = np.unpackbits(M.cpu().numpy().view(np.uint8), axis=1, bitorder="little")
    return bits[:, :n].astype(bool), bits[:, n:]


def resampled(rng, n, uniq, b=1):
    base = rng.random((b, uniq, 3), dtype=np.float32) - np.float32(0.5)
    return np.stack([base[e][rng.integers(0, uniq, n)] for e in range(b)])


def pad_repeated(rng, n, uniq, b=1):
    base = rng.random((b, uniq, 3), dtype=np.float32) - np.float32(0.5)
    return np.ascontiguousarray(base[:, np.arange(n) % uniq])


@pytest.mark.parametrize("b,n,uniq", [(1, 4096, 300), (3, 5000, 1700), (2, 33, 5), (1, 64, 64), (1, 1, 1), (4, 16384, 9000),
                                      (1, 100000, 1), (2, 8191, 8191)])
def test_duplicate_mask_matches_numpy(gp, b, n, uniq):
    rng = np.random.default_rng(n + uniq)
    for x in (resampled(rng, n, uniq, b), pad_repeated(rng, n, uniq, b)):
        if n > 8:
            x[0, 1, 0] = np.float32(0.0)
            x[0, 3] = x[0, 1]
            x[0, 3, 0] = np.float32(-0.0)           # -0 / +0: the bit compare keeps them apart (numpy's does too)
        got, tail = gpu_mask(gp, x)
        np.testing.assert_array_equal(got, later_copies(x))
        assert not tail.any(), "bits past the cloud's end must be clear"
    # second call on the same table (older generation entries count as free)
    y = pad_repeated(rng, n, max(1, uniq // 2), b)
    np.testing.assert_array_equal(gpu_mask(gp, y)[0], later_copies(y))


CASES = [("pad300_4096", lambda r: (r.random((1, 4096, 3), dtype=np.float32) - np.float32(0.5), pad_repeated(r, 4096, 300))),
         ("resampled_16384", lambda r: (r.random((1, 16384, 3), dtype=np.float32) - np.float32(0.5), resampled(r, 16384, 8000))),
         ("both_resampled_b4", lambda r: (resampled(r, 15403, 5000, 4), resampled(r, 7855, 4000, 4))),
         ("pad_both_8x8192", lambda r: (pad_repeated(r, 8192, 1000, 8), pad_repeated(r, 8192, 2500, 8))),
         ("ragged_b2", lambda r: (resampled(r, 12001, 3000, 2), pad_repeated(r, 13003, 700, 2)))]


@pytest.mark.parametrize("name,make", CASES)
def test_duplicates_bit_exact_in_every_policy(gp, oracle, name, make):
    """Forced on, forced off (everything flagged goes through the exhaustive pass) and the default policy: the same
    bits as the oracle, and with the pre-pass on (almost) no query is left to the exhaustive pass."""
    a, b = make(np.random.default_rng(len(name)))
    exp = oracle.chamfer_forward(a, b, 1)
    read_stats(gp)
    assert_bits(run_hip(gp, a, b, 1, path=3, hooks=HOOK_COUNT | HOOK_OFF), exp, name + " off")
    q_off, ex_off, _ = read_stats(gp)
    assert_bits(run_hip(gp, a, b, 1, path=3, hooks=HOOK_COUNT | HOOK_ON), exp, name + " on")
    q_on, ex_on, _ = read_stats(gp)
    assert q_on == q_off == a.shape[0] * (a.shape[1] + b.shape[1])
    assert ex_on * 200 <= q_on, (name, ex_on, q_on)
    assert ex_off >= ex_on
    assert_bits(run_hip(gp, a, b, 1), exp, name + " default")


def test_policy_switches_on_after_a_flagged_call_and_off_again(gp, oracle):
    """Single-round launches run the pre-pass only while the input asks for it: the call after one that sent many
    queries to the exhaustive pass is de-duplicated, and the call after a pre-pass that found nothing is not."""
    torch = gp["torch"]
    rng = np.random.default_rng(5)
    a = rng.random((1, 16384, 3), dtype=np.float32) - np.float32(0.5)
    dup = pad_repeated(rng, 16384, 1260)        # 13 copies of every point, several per candidate list: every query of a is flagged
    clean = rng.random((1, 16384, 3), dtype=np.float32) - np.float32(0.5)
    exp = oracle.chamfer_forward(a, dup, 1)
    run_hip(gp, a, clean, 1, path=3, hooks=HOOK_COUNT)      # whatever state earlier tests left: two clean calls end "off"
    run_hip(gp, a, clean, 1, path=3, hooks=HOOK_COUNT)
    read_stats(gp)
    seen = []
    for _ in range(4):
        assert_bits(run_hip(gp, a, dup, 1, path=3, hooks=HOOK_COUNT), exp)
        torch.cuda.synchronize()
        seen.append(read_stats(gp)[1])
    assert seen[0] > 8000 and seen[-1] * 200 <= 32768, seen
    for _ in range(3):
        run_hip(gp, a, clean, 1, path=3, hooks=HOOK_COUNT)
    read_stats(gp)
    assert_bits(run_hip(gp, a, dup, 1, path=3, hooks=HOOK_COUNT), exp)
    assert read_stats(gp)[1] > 8000, "the switch should have gone off on clean input"
