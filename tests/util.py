"""Helpers shared by the test modules."""
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INT_KEYS = ("counts", "rec_distinct", "rec_crossings", "rec_bins", "kind", "hit", "container",
            "adjacent", "component", "source")
F64_KEYS = ("position", "direction", "normal", "wavelength", "travelled", "duration")


def load_golden(name):
    return np.load(os.path.join(GOLD, name))


def assert_same_tables(compiled, gold):
    """The fixture was made with these exact tables; any drift invalidates it."""
    for key, value in compiled.tables().items():
        if f"tab_{key}" not in gold.files:
            # table added after the fixture was made (extension): must be in its neutral state
            assert not np.any(np.asarray(value) > 0), f"{key} is active but absent from the fixture"
            continue
        want = gold[f"tab_{key}"]
        assert np.array_equal(np.asarray(value), want), f"flattener table {key} drifted from fixture"


def assert_bundles_identical(got, want, sums_rtol=None, what=""):
    """Every key equal bit for bit; rec_sums optionally to a tolerance (its
    summation order is not defined on a parallel machine)."""
    for key in want:
        a, b = np.asarray(got[key]), np.asarray(want[key])
        assert a.shape == b.shape, (what, key, a.shape, b.shape)
        if key == "rec_sums" and sums_rtol is not None:
            assert np.allclose(a, b, rtol=sums_rtol, atol=0.0), (what, key)
        else:   # NaNs (ill-posed fuzz scenes) must sit at the same places
            same = np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)
            assert same, (what, key, int(np.sum(a != b)))


def rows_of_ray(data, j, max_events):
    n = int(data["counts"][j])
    return slice(j * max_events, j * max_events + n)


def three_sigma(p_hat_a, p_hat_b, n_a, n_b):
    """|pa - pb| <= 3 sqrt(p(1-p)(1/na + 1/nb)) with the pooled p."""
    p = (p_hat_a * n_a + p_hat_b * n_b) / (n_a + n_b)
    return 3.0 * np.sqrt(max(p * (1 - p), 1e-12) * (1.0 / n_a + 1.0 / n_b))
