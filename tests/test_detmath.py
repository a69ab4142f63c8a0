"""Deterministic elementary functions of the oracle against float64 references (ulp bounds).
The GPU copy is compared bit for bit in test_detmath_gpu.py."""
import numpy as np
import pytest


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_sincos_on_sampling_range(O):
    x = np.linspace(0, 2 * np.pi, 200001, dtype=np.float32)
    s, c = O.detmath(0, x), O.detmath(1, x)
    # absolute error bound (near zeros of sin/cos the ulp of the result is tiny)
    assert np.max(np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64)))) < 1.5e-7
    assert np.max(np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64)))) < 1.5e-7
    assert np.max(np.abs(s.astype(np.float64) ** 2 + c.astype(np.float64) ** 2 - 1.0)) < 4e-7


def test_cos_large_arguments(O):
    """iridescence phase reaches tens of radians (pbr_utils.cuh:118-120)."""
    x = np.random.RandomState(1).uniform(-400, 400, 100000).astype(np.float32)
    c = O.detmath(1, x)
    assert np.max(np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64)))) < 3e-7


def test_exp_log_pow(O):
    rs = np.random.RandomState(2)
    x = rs.uniform(-87, 88, 200000).astype(np.float32)
    assert ulp_err(O.detmath(2, x), np.exp(x.astype(np.float64))).max() <= 2.0
    y = np.exp(rs.uniform(-80, 80, 200000)).astype(np.float32)
    lg = O.detmath(3, y)
    ref = np.log(y.astype(np.float64))
    near1 = np.abs(ref) < 0.1
    assert ulp_err(lg[~near1], ref[~near1]).max() <= 2.0
    assert np.max(np.abs(lg[near1] - ref[near1])) < 2e-8 + 2e-7 * np.max(np.abs(ref[near1]))
    z = rs.uniform(0.0031308, 1.0, 100000).astype(np.float32)
    p = O.detmath(4, z, np.full_like(z, np.float32(1.0 / 2.4)))
    assert ulp_err(p, z.astype(np.float64) ** (1.0 / 2.4)).max() <= 6.0  # exp(y*log x): only feeds the 8-bit sRGB output


def test_special_values(O):
    f = np.float32
    assert O.detmath(2, np.array([-200.0, 0.0, 100.0], f)).tolist() == [0.0, 1.0, np.inf]
    lg = O.detmath(3, np.array([1.0, 0.0, -1.0, np.inf], f))
    assert lg[0] == 0.0 and lg[1] == -np.inf and np.isnan(lg[2]) and lg[3] == np.inf
    den = O.detmath(3, np.array([1e-40], f))
    assert abs(den[0] - np.log(np.float64(f(1e-40)))) < 1e-4
    tiny = O.detmath(2, np.array([-100.0], f))  # denormal result, two-step scaling
    assert 0 < tiny[0] < 1e-38
