"""csrc/fastdiv.h -- the shared-reciprocal fp32 division of get_uvs' write pass -- against the compiler's correctly
rounded division, bit for bit, wherever the callers' range test admits it.

The claim is structural (fastdiv.h: the compiler's own sequence minus the exponent rescue and the special-case fix-up,
both identities inside the range), so the check is differential: uniformly random bit patterns over the admitted
exponent range, operands that sit on rounding boundaries (quotients of neighbouring floats, denominators with all-ones
and single-bit mantissas, numerators a few ulp around exact multiples), and the operand shapes get_uvs produces.
Outside the range nothing is claimed; the range flag itself is checked against its definition."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LO, HI = 2.0 ** -50, 2.0 ** 50


def _probe(torch, lib, num, den):
    n = num.shape[0]
    a = torch.from_numpy(np.ascontiguousarray(num, np.float32)).cuda()
    d = torch.from_numpy(np.ascontiguousarray(den, np.float32)).cuda()
    fast, fast2, ieee = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    ok = torch.empty(n, dtype=torch.uint8, device="cuda")
    rc = lib.on_device_of(a, lib.lib.genpc_fastdiv_probe, n, lib.ptr(a), lib.ptr(d), lib.ptr(fast), lib.ptr(fast2), lib.ptr(ieee),
                          lib.ptr(ok))
    assert rc == 1, lib.last_error()
    return fast.cpu().numpy(), fast2.cpu().numpy(), ieee.cpu().numpy(), ok.cpu().numpy().astype(bool)


def _check(torch, lib, num, den):
    fast, fast2, ieee, ok = _probe(torch, lib, num, den)
    want_ok = (np.abs(num) >= LO) & (np.abs(num) <= HI) & (np.abs(den) >= LO) & (np.abs(den) <= HI)
    np.testing.assert_array_equal(ok, want_ok)
    np.testing.assert_array_equal(fast.view(np.uint32)[ok], ieee.view(np.uint32)[ok])
    np.testing.assert_array_equal(fast2.view(np.uint32)[ok], ieee.view(np.uint32)[ok])
    # and the compiler's division is the correctly rounded one (float64 quotient of two floats rounds to the same float
    # unless it sits within 2^-29 of a float32 rounding boundary; those are left to the bit comparison above)
    q = num[ok].astype(np.float64) / den[ok].astype(np.float64)
    q32 = q.astype(np.float32)
    near_tie = np.abs((q - q32.astype(np.float64))) > 0.49999 * np.spacing(np.abs(q32)).astype(np.float64)
    np.testing.assert_array_equal(ieee[ok][~near_tie], q32[~near_tie])
    return int(ok.sum())


def _random_bits(rng, n, e_lo, e_hi):
    """uniform sign, exponent in [e_lo, e_hi], uniform 23-bit mantissa"""
    e = rng.integers(e_lo + 127, e_hi + 128, n, dtype=np.uint32)
    m = rng.integers(0, 1 << 23, n, dtype=np.uint32)
    s = rng.integers(0, 2, n, dtype=np.uint32)
    return ((s << 31) | (e << 23) | m).view(np.float32)


def test_random_operands_over_the_whole_range():
    import torch
    from genpc_amd import _lib as lib
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(4):
        n = 1 << 24
        checked += _check(torch, lib, _random_bits(rng, n, -50, 49), _random_bits(rng, n, -50, 49))
    assert checked > 60_000_000


def test_rounding_boundaries():
    import torch
    from genpc_amd import _lib as lib
    rng = np.random.default_rng(1)
    n = 1 << 22
    den = _random_bits(rng, n, -20, 20)
    # special mantissas: all ones, a single bit, zero
    den.view(np.uint32)[: n // 8] |= np.uint32(0x7fffff)
    den.view(np.uint32)[n // 8: n // 4] &= np.uint32(0xff800001)
    den.view(np.uint32)[n // 4: 3 * n // 8] &= np.uint32(0xff800000)
    # numerators a few ulp around q * den for a float q: the true quotient then sits next to a float or a tie
    q = _random_bits(rng, n, -20, 20)
    prod = (q.astype(np.float64) * den.astype(np.float64)).astype(np.float32)
    for k in (-2, -1, 0, 1, 2):
        num = (prod.view(np.int32) + np.int32(k)).view(np.float32)
        _check(torch, lib, num, den)
    # midpoints: numerators of the form (q + ulp/2) * den rounded
    half = ((q.astype(np.float64) + 0.5 * np.spacing(q).astype(np.float64)) * den.astype(np.float64)).astype(np.float32)
    for k in (-1, 0, 1):
        _check(torch, lib, (half.view(np.int32) + np.int32(k)).view(np.float32), den)
    # neighbouring floats and equal operands
    _check(torch, lib, (den.view(np.int32) + np.int32(1)).view(np.float32), den)
    _check(torch, lib, den.copy(), den)


def test_operands_of_get_uvs():
    """numerators focal * xc in a few units, denominators the camera distance (~1.6) or a box extent (~0.5)"""
    import torch
    from genpc_amd import _lib as lib
    rng = np.random.default_rng(2)
    n = 1 << 23
    num = ((rng.random(n, dtype=np.float32) - np.float32(0.5)) * np.float32(4.0)).astype(np.float32)
    den = (np.float32(1.6) + (rng.random(n, dtype=np.float32) - np.float32(0.5))).astype(np.float32)
    assert _check(torch, lib, num, den) > 0.99 * n
    den2 = (np.float32(0.3) + rng.random(n, dtype=np.float32)).astype(np.float32)
    _check(torch, lib, num, den2)


def test_out_of_range_is_flagged():
    import torch
    from genpc_amd import _lib as lib
    num = np.array([0.0, -0.0, 1.0, 1.0, np.inf, 1e-30, 1.0, 2.0 ** 51, 2.0 ** -51, 2.0 ** 50, 2.0 ** -50], np.float32)
    den = np.array([1.0, 1.0, 0.0, np.inf, 1.0, 1.0, 1e-30, 1.0, 1.0, 2.0 ** -50, 2.0 ** 50], np.float32)
    fast, fast2, ieee, ok = _probe(torch, lib, num, den)
    np.testing.assert_array_equal(ok, [0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1])
    np.testing.assert_array_equal(fast[ok], ieee[ok])


def test_list_codes_are_lower_bounds():
    """csrc/nn.h list_enc / list_dec: the filter hands a2 and a3 to the finish step as 16-bit codes of lower bounds.
    decode(encode(base, v)) <= v for every v >= base (incl. equal, adjacent floats, opposite signs, heavy cancellation,
    infinities), and close: within 0.4 % of (v - base) plus a few ulps of the larger magnitude."""
    import torch
    from genpc_amd import _lib as lib
    rng = np.random.default_rng(5)
    n = 1 << 24
    base = _random_bits(rng, n, -30, 20)
    kinds = rng.integers(0, 6, n)
    gap = np.abs(_random_bits(rng, n, -40, 20))
    v = base + gap                                                       # generic
    rel = np.abs(base) * (2.0 ** rng.uniform(-24, -1, n)).astype(np.float32)
    v = np.where(kinds == 1, base + rel.astype(np.float32), v)          # gaps of a few ulps .. half the magnitude
    v = np.where(kinds == 2, base, v)                                    # equal
    v = np.where(kinds == 3, np.nextafter(base, np.float32(np.inf)), v)  # adjacent floats
    v = np.where(kinds == 4, np.float32(np.inf), v)
    v = np.where(kinds == 5, np.abs(base) * np.float32(3.0), v)          # across zero when base < 0
    v = np.maximum(v.astype(np.float32), base)
    b = torch.from_numpy(base).cuda()
    vv = torch.from_numpy(v).cuda()
    out = torch.empty_like(b)
    rc = lib.on_device_of(b, lib.lib.genpc_list_code_probe, n, lib.ptr(b), lib.ptr(vv), lib.ptr(out))
    assert rc == 1, lib.last_error()
    got = out.cpu().numpy()
    assert (got <= v).all()
    assert (got >= base).all()
    fin = np.isfinite(v)
    assert np.isinf(got[~fin]).all()
    slack = (v[fin].astype(np.float64) - got[fin].astype(np.float64))
    allowed = 0.0040 * (v[fin].astype(np.float64) - base[fin].astype(np.float64)) + 6.0 * 2.0 ** -21 * np.maximum(np.abs(v[fin]), np.abs(base[fin])).astype(np.float64)
    assert (slack <= allowed).all(), float((slack - allowed).max())
