"""pvt_math.h (the bit-reproducible functions shared by the HIP kernel and the
oracle's portable mode) against the host libm: <= 1 ulp apart on the tracer's
domains, exact on the special values the loop relies on."""
import numpy as np
import pytest

from oracle import oracle as O

RNG = np.random.default_rng(0)
N = 200_000
CASES = {
    "log": (1.0 - RNG.random(N), np.log),
    "sin": (RNG.random(N) * 2 * np.pi, np.sin),
    "cos": (RNG.random(N) * 2 * np.pi, np.cos),
    "asin": (RNG.random(N) * 2 - 1, np.arcsin),
    "acos": (RNG.random(N) * 2 - 1, np.arccos),
}


@pytest.mark.parametrize("fn", sorted(CASES))
def test_portable_within_one_ulp_of_libm(fn):
    x, ref = CASES[fn]
    y = O.math(fn, x, math_mode=O.MATH_PORTABLE)
    r = ref(x)
    ulps = np.abs(y - r) / np.spacing(np.abs(r))
    assert not np.isnan(y).any()
    assert ulps.max() <= 1.0, (fn, ulps.max())
    # and the libm mode of the oracle really is the C library (numpy's array loops may
    # use their own SIMD kernels, so compare with the `math` module, which calls libm)
    import math

    sub = x[:5000]
    want = np.array([getattr(math, fn)(float(v)) for v in sub])
    assert np.array_equal(O.math(fn, sub, math_mode=O.MATH_LIBM), want)


def test_special_values():
    P = O.MATH_PORTABLE
    assert O.math("log", [1.0], P)[0] == 0.0
    assert O.math("log", [2.0 ** -53], P)[0] == np.log(2.0 ** -53)
    assert O.math("sin", [0.0], P)[0] == 0.0 and O.math("cos", [0.0], P)[0] == 1.0
    assert O.math("acos", [1.0, -1.0, 0.0], P).tolist() == [0.0, np.pi, np.pi / 2]
    assert O.math("asin", [1.0, -1.0, 0.0], P).tolist() == [np.pi / 2, -np.pi / 2, 0.0]
    assert np.isnan(O.math("acos", [1.0000001], P)[0])
    assert np.isneginf(O.math("log", [0.0], P)[0])


def test_sincos_large_arguments_still_accurate():
    x = np.linspace(-3.0e5, 3.0e5, 20001)
    for fn, ref in (("sin", np.sin), ("cos", np.cos)):
        y = O.math(fn, x, O.MATH_PORTABLE)
        assert np.max(np.abs(y - ref(x))) < 2e-16 * 2


def test_absorber_depth_known_answer():
    """Reference tests/test_refractored_tracer.py:168-194: with numpy seed 0 the second
    uniform is 0.7151893663724195 and a 10 cm^-1 absorber stops the ray 0.1255930762965882 cm
    into the box (-0.5 + depth = -0.3744069237034118)."""
    u = 0.7151893663724195
    for mode in (O.MATH_LIBM, O.MATH_PORTABLE):
        depth = -O.math("log", [1.0 - u], mode)[0] / 10.0
        assert depth == pytest.approx(0.1255930762965882, rel=1e-15)
        assert -0.5 + depth == pytest.approx(-0.3744069237034118, rel=1e-14)
