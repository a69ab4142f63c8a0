"""div3 (csrc/pt_device.hip.h): the three quotients v / t of a vector by one divisor through ONE correctly rounded
reciprocal and one residual step each, against the compiler's IEEE division -- bit for bit.
The suite checks every numerator significand against a sample of divisor significands (the edges of [1, 2), the divisors
whose reciprocals sit next to a rounding boundary, random ones) and the guard / fallback ranges;
`python tools/div3_exhaustive.py` walks ALL 2^23 divisor significands (2^46 pairs, about a minute of GPU time; its log is
profiles/r02_div3_exhaustive.txt)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(P, s, first, count, mode):
    out = (C.c_uint * 9)()
    P.lib.ptrt_debug_div3_check.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_uint)]
    assert P.lib.ptrt_debug_div3_check(s.ctx, first, count, mode, out) == 0
    assert out[0] == 0, f"mode {mode}, divisors {first:#x}+{count}: {out[0]} mismatches, first {{a, t}} bits: " \
                        f"{[hex(v) for v in list(out)[1:9]]}"


def test_shared_reciprocal_division_is_ieee_for_sampled_divisors(P):
    s = P.Scene(16, 16)
    rng = np.random.default_rng(7)
    starts = [0, 1 << 22, (1 << 23) - 2048, 0x3504f3 - 1024, 0x2aaaaa - 1024, 0x555555 - 1024]  # 1, 1.5, ->2, sqrt2, 4/3, 5/3
    starts += [int(v) for v in rng.integers(0, (1 << 23) - 2048, 10)]
    for first in starts:
        _check(P, s, first, 2048, 0)
    s.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_out_of_range_operands_take_the_ieee_division(P, mode):
    """Subnormal, tiny, huge, infinite and NaN components, signed zeros; divisors scaled out of the fast range."""
    s = P.Scene(16, 16)
    for first in (0, 0x3fffff, (1 << 23) - 512):
        _check(P, s, first, 512, mode)
    s.close()


def test_the_checker_rejects_an_unrefined_reciprocal(P):
    """Negative control: the same residual step on the RAW v_rcp_f32 (1 ulp) instead of the correctly rounded reciprocal is
    NOT the IEEE quotient for about one divisor significand in fifty (47,045 reported pairs over all 2^46) -- the checker sees it."""
    s = P.Scene(16, 16)
    out = (C.c_uint * 9)()
    P.lib.ptrt_debug_div3_check.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_uint)]
    assert P.lib.ptrt_debug_div3_check(s.ctx, 0x700000, 16384, 3, out) == 0
    assert out[0] > 0
    s.close()
