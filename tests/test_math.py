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


# ---- a second, independent accuracy referee -----------------------------------------------------------
# The GPU and its CPU referee both evaluate pvt_math.h, so a flaw in that header would be common-mode; the
# libm comparison above is one outside check.  These are two more, neither of which shares code with libm's
# double-precision paths: glibc's 80-bit long-double functions at 10^6 points per function, and mpmath
# (arbitrary precision, 40 digits) at 2*10^4.  The error is measured against the exact value, in units of
# the last place of the double result: < 1 ulp everywhere on the tracer's domains.
BIG = 1_000_000
RNG2 = np.random.default_rng(20260928)
DOMAINS = {
    "log": lambda n: 1.0 - RNG2.random(n),              # log(1 - u), u in [0, 1)  (_kernel.pyx:765)
    "sin": lambda n: RNG2.random(n) * 2 * np.pi,        # azimuths and polar angles
    "cos": lambda n: RNG2.random(n) * 2 * np.pi,
    "asin": lambda n: RNG2.random(n) * 2 - 1,
    "acos": lambda n: np.concatenate([RNG2.random(n // 2) * 2 - 1, 1.0 - RNG2.random(n - n // 2) ** 4]),  # incl. near 1 (grazing exits)
}
DOMAINS.update({
    # the composed functions: turn fractions in [0, 1); cosines / sines in [-1, 1] incl. the ends
    "sin2pi": lambda n: RNG2.random(n), "cos2pi": lambda n: RNG2.random(n),
    "sqrt1m2": lambda n: np.concatenate([RNG2.random(n // 2) * 2 - 1, 1.0 - RNG2.random(n - n // 2) ** 4]),
})
_TWO_PI_LONG = 2 * np.arccos(np.longdouble(-1))
LONG = {"log": np.log, "sin": np.sin, "cos": np.cos, "asin": np.arcsin, "acos": np.arccos,
        "sin2pi": lambda g: np.sin(_TWO_PI_LONG * g), "cos2pi": lambda g: np.cos(_TWO_PI_LONG * g),
        "sqrt1m2": lambda c: np.sqrt((1 - c) * (1 + c))}
ULP_BOUND = {"sqrt1m2": 1.3}   # documented in pvt_math.h; everything else < 1


def _ulp_error(y, exact_longdouble):
    err = np.abs(y.astype(np.longdouble) - exact_longdouble)
    scale = np.spacing(np.abs(y)).astype(np.longdouble)
    return (err / scale).astype(np.float64)


@pytest.mark.parametrize("fn", sorted(DOMAINS))
def test_portable_within_one_ulp_of_long_double_evaluation_at_a_million_points(fn):
    assert np.finfo(np.longdouble).nmant >= 63, "needs x87 extended precision"
    x = DOMAINS[fn](BIG)
    y = O.math(fn, x, math_mode=O.MATH_PORTABLE)
    exact = LONG[fn](x.astype(np.longdouble))
    keep = np.abs(exact) > 1e-300          # (sin/cos exactly at a zero crossing: the ulp of ~0 is meaningless)
    if fn in ("sin2pi", "cos2pi"):
        # near a zero crossing the 64-bit mantissa of the long-double 2*pi*g is itself the limit (its rounding is
        # 0.002 / |result| ulp of the double result): stay clear of them here; mpmath below covers them
        keep &= np.abs(exact) > 0.05
    ulps = _ulp_error(y[keep], exact[keep])
    assert ulps.max() < ULP_BOUND.get(fn, 1.0), (fn, float(ulps.max()), float(x[keep][np.argmax(ulps)]))
    assert np.mean(ulps <= 0.5 + 1e-9) > 0.7            # mostly correctly rounded


@pytest.mark.parametrize("fn", sorted(DOMAINS))
def test_portable_within_one_ulp_of_mpmath(fn):
    mpmath = pytest.importorskip("mpmath")
    mpmath.mp.dps = 40
    x = DOMAINS[fn](20_000)
    if fn in ("sin2pi", "cos2pi"):   # and the zero crossings of both
        x = np.concatenate([x, 0.25 + RNG2.normal(size=2000) * 1e-5, 0.5 + RNG2.normal(size=2000) * 1e-7,
                            0.75 + RNG2.normal(size=2000) * 1e-3, RNG2.random(1000) * 1e-6])
    y = O.math(fn, x, math_mode=O.MATH_PORTABLE)
    f = {"sin2pi": lambda g: mpmath.sin(2 * mpmath.pi * g), "cos2pi": lambda g: mpmath.cos(2 * mpmath.pi * g),
         "sqrt1m2": lambda c: mpmath.sqrt((1 - c) * (1 + c))}.get(fn) or getattr(mpmath, fn)
    worst = 0.0
    for xv, yv in zip(x.tolist(), y.tolist()):
        exact = f(mpmath.mpf(xv))
        if abs(exact) < 1e-300:
            continue
        ulp = mpmath.mpf(float(np.spacing(abs(yv))))
        worst = max(worst, float(abs(mpmath.mpf(yv) - exact) / ulp))
    assert worst < ULP_BOUND.get(fn, 1.0), (fn, worst)
