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


def describe_lsc_scene(scene):
    """What an `LSC(...)` builder made, as a flat {key: array} dict: per node (level order) its name, box size, refractive
    index, pose, components (class, name, coefficient, emission spectrum, quantum yield, phase function) and light (name,
    delegates by name and arguments).  Works on the product's scene and on the one the REFERENCE's LSC class builds
    (tests/golden/make_golden.py: make_lsc_scenes) -- both keep these facts under the same attribute names."""
    import functools

    def delegate(d):
        if d is None:
            return "None"
        if isinstance(d, functools.partial):
            return d.func.__name__ + repr(tuple(float(a) for a in d.args))
        if hasattr(d, "__dict__") and not callable(getattr(d, "__name__", None)):
            args = {k: v for k, v in vars(d).items() if isinstance(v, (int, float))}
            return type(d).__name__ + repr(sorted(args.items()))
        return getattr(d, "__name__", type(d).__name__)

    out = {}
    nodes = list(scene.root.levelorder())
    out["names"] = np.array([n.name for n in nodes])
    for i, n in enumerate(nodes):
        out[f"{i}_pose"] = np.asarray(n.pose, dtype=float)
        out[f"{i}_parent"] = np.array(-1 if n.parent is None else nodes.index(n.parent))
        if n.geometry is not None:
            m = n.geometry.material
            out[f"{i}_size"] = np.asarray(n.geometry.size, dtype=float)
            out[f"{i}_n"] = np.array(float(m.refractive_index))
            out[f"{i}_surface"] = np.array(type(m.surface.delegate).__name__)
            out[f"{i}_components"] = np.array([type(c).__name__ + ":" + c.name for c in m.components] or [""])
            for k, c in enumerate(m.components):
                out[f"{i}_c{k}_coefficient"] = np.asarray(c._coefficient, dtype=float)
                out[f"{i}_c{k}_qy"] = np.array(float(c.quantum_yield))
                out[f"{i}_c{k}_phase"] = np.array(delegate(c.phase_function))
                if hasattr(c, "_emission"):
                    out[f"{i}_c{k}_emission"] = np.asarray(c._emission, dtype=float)
        if getattr(n, "light", None) is not None:
            out[f"{i}_light"] = np.array([n.light.name, delegate(n.light.wavelength), delegate(n.light.position), delegate(n.light.direction)])
    return out
