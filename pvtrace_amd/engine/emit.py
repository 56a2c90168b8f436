"""Light sampling for the engine: host (numpy) and device (HIP) emitters.

The reference emits every bundle in Python before its timed region
(pvtrace/engine/api.py:230-245, emit.py:92-134) and only vectorises delegates
it recognises by *class*; ``functools.partial(cone, theta)`` — what LSC() and
the examples use — drops to one Python call per ray (emit.py:116-124), 70x
slower than the trace itself.  Here every built-in delegate, in any spelling,
is lowered to a small `EmitterTables` record that either

* the host sampler below draws with numpy (vectorised; `seed=None` uses the
  global ``np.random`` state exactly like the reference), or
* the HIP emission kernel draws on the GPU from per-ray RNG streams, so a
  10^8-photon run never materialises a 5.6 GB host array.

A delegate that cannot be lowered (a lambda, a user function) keeps working in
host mode: THAT delegate is called once per ray, the light's other delegates are
still sampled vectorised and no Ray object is built per photon.

**Vectorised user delegates.**  A user's own delegate is called once per BUNDLE instead
of once per ray when it says it can be: an object with a ``sample(n)`` method, or a
callable carrying ``vectorized = True`` (then called as ``delegate(n)``), returning
``n`` wavelengths (shape ``(n,)``) or ``n`` positions / directions (shape ``(n, 3)``)
in the light's frame.  Called without arguments it must still give one sample (that is
what `Light.emit` and the reference's per-ray path do, pvtrace/light/light.py:226-233),
so the same object works everywhere; `vectorized_delegate` wraps a per-bundle function
into such an object.
"""
import functools
import threading

import collections.abc

import numpy as np

from pvtrace_amd import light as Lm
from pvtrace_amd import material as Mm
from pvtrace_amd.engine.compiler import UnsupportedSceneError

WL_CONSTANT, WL_SPECTRUM, WL_SPECTRUM_HIST = 0, 1, 2
POS_POINT, POS_RECT, POS_CIRCLE, POS_CUBE = 0, 1, 2, 3
DIR_Z, DIR_CONE, DIR_ISOTROPIC, DIR_LAMBERTIAN, DIR_HG = 0, 1, 2, 3, 4


def _partial_of(delegate, func, nargs):
    return (
        isinstance(delegate, functools.partial)
        and delegate.func is func
        and not delegate.keywords
        and len(delegate.args) == nargs
    )


def classify_wavelength(d):
    if d is Lm.default_wavelength or isinstance(d, Lm.DefaultWavelength):
        return (WL_CONSTANT, 555.0, None)
    if isinstance(d, Lm.ConstantWavelengthMask):
        return (WL_CONSTANT, float(d.nanometers), None)
    dist = None
    if isinstance(d, Lm.SpectrumWavelengthMask):
        dist = d.distribution
    elif getattr(d, "__func__", None) is Lm.SpectrumWavelengthMask.__call__:
        dist = d.__self__.distribution          # the bound `mask.__call__`
    if dist is not None:
        if dist._x is None:     # a constant "distribution"
            return (WL_CONSTANT, float(dist._y), None)
        return (WL_SPECTRUM_HIST if dist.hist else WL_SPECTRUM, 0.0, dist)
    return None


def classify_position(d):
    if d is Lm.default_position or isinstance(d, Lm.DefaultPosition):
        return (POS_POINT, (0.0, 0.0, 0.0))
    if isinstance(d, Lm.RectangularMask):
        return (POS_RECT, (d.x, d.y, 0.0))
    if _partial_of(d, Lm.rectangular_mask, 2):
        return (POS_RECT, (float(d.args[0]), float(d.args[1]), 0.0))
    if isinstance(d, Lm.CircularMask):
        return (POS_CIRCLE, (float(d.radius), 0.0, 0.0))
    if _partial_of(d, Lm.circular_mask, 1):
        return (POS_CIRCLE, (float(d.args[0]), 0.0, 0.0))
    if isinstance(d, Lm.CubeMask):
        return (POS_CUBE, (float(d.x), float(d.y), float(d.z)))
    if _partial_of(d, Lm.cube_mask, 3):
        return (POS_CUBE, tuple(float(v) for v in d.args))
    return None


def classify_direction(d):
    if d is Lm.default_direction or isinstance(d, Lm.DefaultDirection):
        return (DIR_Z, 0.0)
    if isinstance(d, Mm.Cone):
        return (DIR_CONE, d.theta_max)
    if _partial_of(d, Mm.cone, 1):
        return (DIR_CONE, float(d.args[0]))
    if d is Mm.isotropic:
        return (DIR_ISOTROPIC, 0.0)
    if d is Mm.lambertian:
        return (DIR_LAMBERTIAN, 0.0)
    g = None
    if isinstance(d, Mm.HenyeyGreenstein):
        g = d.g
    elif _partial_of(d, Mm.henyey_greenstein, 1):
        g = float(d.args[0])
    if g is not None:
        return (DIR_ISOTROPIC, 0.0) if abs(g) < 1e-12 else (DIR_HG, g)
    return None


def vectorized_delegate(sample_n):
    """Wrap `sample_n(n) -> array of n samples` into a light delegate: `delegate()` gives one sample (any Light, the
    per-ray paths), `delegate.sample(n)` the whole bundle at once (what `emit_bundle` uses)."""

    class _Vectorized:
        vectorized = True

        def __call__(self, n=None):
            if n is None:
                one = np.asarray(sample_n(1))
                return float(one[0]) if one.ndim == 1 else tuple(one[0].tolist())
            return sample_n(int(n))

        def sample(self, n):
            return sample_n(int(n))

    return _Vectorized()


def _bundle_sampler(delegate):
    """`f(n)` when the user's delegate can produce a whole bundle at once (see the module docstring), else None."""
    if callable(getattr(delegate, "sample", None)) and not isinstance(delegate, type):
        return delegate.sample
    if getattr(delegate, "vectorized", False):
        return delegate
    return None


def _user_samples(call, count, width):
    """`count` samples of a user delegate, (count,) for width 1 else (count, width): one call for the bundle when the
    delegate offers it, else one call per ray (the reference's behaviour, pvtrace/engine/emit.py:116-124)."""
    many = _bundle_sampler(call)
    if many is not None:
        out = np.asarray(many(count), dtype=np.float64)
        want = (count,) if width == 1 else (count, width)
        if out.shape != want:
            raise ValueError(f"vectorised light delegate returned shape {out.shape}, expected {want}")
        return np.ascontiguousarray(out)
    if width == 1:
        return np.fromiter((call() for _ in range(count)), dtype=np.float64, count=count)
    return np.array([tuple(np.asarray(call()).tolist()) for _ in range(count)], dtype=np.float64).reshape(count, width)


class EmitterTables:
    """SoA description of the scene's lights (one row per light, level order)."""

    def __init__(self, scene, strict=True):
        nodes = scene.light_nodes
        if not nodes:
            raise UnsupportedSceneError("Scene has no lights.")
        n = len(nodes)
        self.n_lights = n
        self.names = [node.light.name for node in nodes]
        self.nodes = nodes
        self.wl_type = np.zeros(n, dtype=np.int32)
        self.wl_value = np.zeros(n, dtype=np.float64)
        self.wl_spec_start = np.zeros(n, dtype=np.int32)
        self.wl_spec_n = np.zeros(n, dtype=np.int32)
        self.pos_type = np.zeros(n, dtype=np.int32)
        self.pos_param = np.zeros((n, 3), dtype=np.float64)
        self.dir_type = np.zeros(n, dtype=np.int32)
        self.dir_param = np.zeros(n, dtype=np.float64)
        self.light_to_world = np.zeros((n, 4, 4), dtype=np.float64)
        self.builtin = np.ones(n, dtype=bool)
        # delegates the tables cannot express, per light: {"wavelength" | "position" | "direction": callable};
        # the host sampler calls those once per ray and still samples the light's other delegates vectorised
        self.custom = [dict() for _ in range(n)]
        sx, sc = [], []
        for i, node in enumerate(nodes):
            light = node.light
            self.light_to_world[i] = np.asarray(node.transformation_to(scene.root), dtype=np.float64)
            w = classify_wavelength(light.wavelength)
            p = classify_position(light.position)
            d = classify_direction(light.direction)
            if w is None or p is None or d is None:
                if strict:
                    raise UnsupportedSceneError(
                        f"Light {light.name!r} uses a custom delegate; device-side "
                        "emission needs built-in wavelength/position/direction delegates."
                    )
                self.builtin[i] = False
                if w is None:
                    self.custom[i]["wavelength"] = light.wavelength
                    w = (WL_CONSTANT, 0.0, None)
                if p is None:
                    self.custom[i]["position"] = light.position
                    p = (POS_POINT, (0.0, 0.0, 0.0))
                if d is None:
                    self.custom[i]["direction"] = light.direction
                    d = (DIR_Z, 0.0)
            self.wl_type[i], self.wl_value[i], dist = w
            if dist is not None:
                self.wl_spec_start[i] = len(sx)
                sx.extend(np.asarray(dist._x, dtype=np.float64).tolist())
                sc.extend(np.asarray(dist._cdf, dtype=np.float64).tolist())
                self.wl_spec_n[i] = len(sx) - self.wl_spec_start[i]
            self.pos_type[i], self.pos_param[i] = p
            self.dir_type[i], self.dir_param[i] = d
        self.spec_x = np.array(sx, dtype=np.float64)
        self.spec_cdf = np.array(sc, dtype=np.float64)


_POOL = None
_IN_WORKER = threading.local()


def _chunked(fn, n, min_rows=200_000):
    """Run `fn(lo, hi)` over row ranges covering [0, n): on a few worker threads when the bundle is large
    (numpy's elementwise kernels release the GIL).  The draws themselves stay sequential, so the result does
    not depend on the split."""
    if n < 2 * min_rows or getattr(_IN_WORKER, "active", False):   # (never wait on the pool from one of its workers)
        fn(0, n)
        return
    global _POOL
    import concurrent.futures
    import os

    workers = max(1, min(8, (os.cpu_count() or 1), n // min_rows))
    if _POOL is None:
        _POOL = concurrent.futures.ThreadPoolExecutor(max_workers=8, thread_name_prefix="pvt-emit")
    edges = np.linspace(0, n, workers + 1).astype(np.int64)
    list(_POOL.map(lambda k: fn(int(edges[k]), int(edges[k + 1])), range(workers)))


def _sphere(theta, phi):
    out = np.empty((theta.shape[0], 3))

    def rows(lo, hi):
        st = np.sin(theta[lo:hi])
        np.multiply(st, np.cos(phi[lo:hi]), out=out[lo:hi, 0])
        np.multiply(st, np.sin(phi[lo:hi]), out=out[lo:hi, 1])
        np.cos(theta[lo:hi], out=out[lo:hi, 2])

    _chunked(rows, theta.shape[0])
    return out


def _map(fn, x):
    """Elementwise `fn` over a 1-D array, chunked like `_sphere`."""
    out = np.empty_like(x)

    def rows(lo, hi):
        out[lo:hi] = fn(x[lo:hi])

    _chunked(rows, x.shape[0])
    return out


def _to_world(local, matrix, translate):
    """(n,3) local rows -> world rows; skips the matrix product for axis-aligned lights."""
    rot = matrix[:3, :3]
    if np.array_equal(rot, np.eye(3)):
        out = local
    elif np.array_equal(rot, np.diag(np.diag(rot))):
        out = local * np.diag(rot)        # axis flips / scalings: x*r + 0*y + 0*z, the matrix product's own bits up to a zero's sign
    else:
        out = local @ rot.T
    if translate and np.any(matrix[:3, 3] != 0.0):
        out = out + matrix[:3, 3]
    return out


def _sample_light(tab, i, n, uniform):
    """Vectorised local-frame samples of light row i; `uniform(n)` -> U[0,1)."""
    if tab.wl_type[i] == WL_SPECTRUM:
        s, m = tab.wl_spec_start[i], tab.wl_spec_n[i]
        wl = np.interp(uniform(n), tab.spec_cdf[s:s + m], tab.spec_x[s:s + m])
    elif tab.wl_type[i] == WL_SPECTRUM_HIST:    # Distribution.sample, hist branch: x[searchsorted(cdf, p)]
        s, m = tab.wl_spec_start[i], tab.wl_spec_n[i]
        wl = tab.spec_x[s:s + m][np.minimum(np.searchsorted(tab.spec_cdf[s:s + m], uniform(n)), m - 1)]
    else:
        wl = np.full(n, tab.wl_value[i])
    pp = tab.pos_param[i]
    kind = tab.pos_type[i]
    if kind == POS_RECT:
        pos = np.column_stack((-pp[0] + 2 * pp[0] * uniform(n), -pp[1] + 2 * pp[1] * uniform(n),
                               np.zeros(n)))
    elif kind == POS_CIRCLE:
        ang = 2 * np.pi * uniform(n)
        rad = np.sqrt(uniform(n)) * pp[0]
        pos = np.column_stack((rad * np.cos(ang), rad * np.sin(ang), np.zeros(n)))
    elif kind == POS_CUBE:
        pos = np.column_stack([-pp[a] + 2 * pp[a] * uniform(n) for a in range(3)])
    else:
        pos = np.zeros((n, 3))
    kind, prm = tab.dir_type[i], tab.dir_param[i]
    if kind == DIR_CONE:
        theta = _map(lambda u: np.arcsin(np.sqrt(u) * np.sin(prm)), uniform(n))
        direc = _sphere(theta, 2 * np.pi * uniform(n))
    elif kind == DIR_ISOTROPIC:
        phi = 2 * np.pi * uniform(n)
        direc = _sphere(_map(lambda u: np.arccos(2 * u - 1), uniform(n)), phi)
    elif kind == DIR_LAMBERTIAN:
        theta = _map(lambda u: np.arcsin(np.sqrt(u)), uniform(n))
        direc = _sphere(theta, 2 * np.pi * uniform(n))
    elif kind == DIR_HG:
        s = 2 * uniform(n) - 1
        mu = (1 + prm * prm - ((1 - prm * prm) / (1 + prm * s)) ** 2) / (2 * prm)
        direc = _sphere(np.arccos(mu), 2 * np.pi * uniform(n))
    else:
        direc = np.tile((0.0, 0.0, 1.0), (n, 1))
    return pos, direc, wl


def emit_bundle(scene, num_rays, seed=None):
    """Emit `num_rays` on the host as world-frame arrays.

    Returns (positions (n,3), directions (n,3), wavelengths (n), sources list);
    ray i comes from light ``i % n_lights`` like `Scene.emit`.  With `seed`
    None the global numpy generator is used (reference behaviour); otherwise
    a private ``np.random.default_rng(seed)``.
    """
    tab = EmitterTables(scene, strict=False)
    if seed is None:
        uniform = lambda n: np.random.uniform(0, 1, n)
    else:
        rng = np.random.default_rng(seed)
        uniform = lambda n: rng.random(n)
    positions = np.zeros((num_rays, 3))
    directions = np.zeros((num_rays, 3))
    wavelengths = np.zeros(num_rays)
    sources = None if all(tab.builtin) else np.empty(num_rays, dtype=object)
    for i, node in enumerate(tab.nodes):
        rows = slice(i, num_rays, tab.n_lights)
        count = len(range(i, num_rays, tab.n_lights))
        if count == 0:
            continue
        pos, direc, wl = _sample_light(tab, i, count, uniform)
        if not tab.builtin[i]:
            # A delegate the tables cannot express (a lambda, a user function) is called once per ray, as the
            # reference does for the whole light (pvtrace/engine/emit.py:116-124, via Light.emit) -- but only
            # THAT delegate: the light's recognised delegates were sampled vectorised above, no Ray object is
            # built per photon and the change of frame below is one matrix product for the whole bundle.
            custom = tab.custom[i]
            # (a delegate that offers it is asked for the whole bundle at once: `sample(n)` / `vectorized = True`)
            if "wavelength" in custom:
                wl = _user_samples(custom["wavelength"], count, 1)
            if "position" in custom:
                pos = _user_samples(custom["position"], count, 3)
            if "direction" in custom:
                direc = _user_samples(custom["direction"], count, 3)
        m = tab.light_to_world[i]
        positions[rows] = _to_world(pos, m, translate=True)
        directions[rows] = _to_world(direc, m, translate=False)
        wavelengths[rows] = wl
        if sources is not None:
            sources[rows] = tab.names[i]
    if sources is None:
        return positions, directions, wavelengths, RoundRobinSources(tab.names, num_rays)
    return positions, directions, wavelengths, sources.tolist()


def emit_bundles(scene, counts, seeds):
    """`emit_bundle(scene, counts[k], seeds[k])` for every k, on the emission worker threads (numpy's elementwise
    kernels release the GIL; each bundle has its own generator, so the results do not depend on the schedule).
    Scenes with a per-ray Python delegate are sampled one bundle after the other (user code is not assumed
    thread-safe)."""
    global _POOL
    jobs = list(zip(counts, seeds))
    if len(jobs) < 2 or not all(EmitterTables(scene, strict=False).builtin):
        return [emit_bundle(scene, c, seed=sd) for c, sd in jobs]
    import concurrent.futures

    if _POOL is None:
        _POOL = concurrent.futures.ThreadPoolExecutor(max_workers=8, thread_name_prefix="pvt-emit")
    def one(job):
        _IN_WORKER.active = True
        try:
            return emit_bundle(scene, job[0], seed=job[1])
        finally:
            _IN_WORKER.active = False

    return list(_POOL.map(one, jobs))


class RoundRobinSources(collections.abc.Sequence):
    """The `sources` list of a bundle -- ray i comes from light ``(offset + i) % n_lights``
    (reference emit.py:112-116 builds the list eagerly; 10^6 Python strings cost more than
    tracing 10^6 photons here, so this is the same sequence, computed on demand).  Compares
    equal to the equivalent list."""

    def __init__(self, names, length, offset=0):
        self.names = list(names)
        self.length = int(length)
        self.offset = int(offset)

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        if isinstance(index, slice):
            start, stop, step = index.indices(self.length)
            if step == 1:
                return RoundRobinSources(self.names, max(stop - start, 0), self.offset + start)
            return [self[i] for i in range(start, stop, step)]
        if index < 0:
            index += self.length
        if not 0 <= index < self.length:
            raise IndexError("ray index out of range")
        return self.names[(self.offset + index) % len(self.names)]

    def __eq__(self, other):
        if isinstance(other, (list, tuple, collections.abc.Sequence)) and not isinstance(other, str):
            return len(other) == self.length and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def tolist(self):
        return list(self)

    def __repr__(self):
        return f"RoundRobinSources({self.names!r}, {self.length})"


class ChainedSources(collections.abc.Sequence):
    """Several `sources` sequences one after the other (the bundles of a group, the shards of a split bundle),
    without materialising 10^6 Python strings: an index is looked up in the part that holds it, a slice that lies
    inside one part is that part's slice.  Compares equal to the equivalent list."""

    def __init__(self, parts):
        self.parts = [p for p in parts if len(p)]
        self.starts = np.concatenate(([0], np.cumsum([len(p) for p in self.parts]))).astype(np.int64)

    def __len__(self):
        return int(self.starts[-1])

    def __getitem__(self, index):
        if isinstance(index, slice):
            start, stop, step = index.indices(len(self))
            if step == 1 and stop > start:
                k = int(np.searchsorted(self.starts, start, side="right")) - 1
                if stop <= self.starts[k + 1]:
                    return self.parts[k][start - int(self.starts[k]):stop - int(self.starts[k])]
                pieces, at = [], start
                while at < stop:
                    k = int(np.searchsorted(self.starts, at, side="right")) - 1
                    upto = min(stop, int(self.starts[k + 1]))
                    pieces.append(self.parts[k][at - int(self.starts[k]):upto - int(self.starts[k])])
                    at = upto
                return ChainedSources(pieces)
            return [self[i] for i in range(start, stop, step)]
        if index < 0:
            index += len(self)
        if not 0 <= index < len(self):
            raise IndexError("ray index out of range")
        k = int(np.searchsorted(self.starts, index, side="right")) - 1
        return self.parts[k][index - int(self.starts[k])]

    def __eq__(self, other):
        if isinstance(other, (list, tuple, RoundRobinSources, ChainedSources)):
            return len(other) == len(self) and all(a == b for a, b in zip(self, other))
        return NotImplemented

    def tolist(self):
        return [x for p in self.parts for x in p]

    def __repr__(self):
        return f"ChainedSources({len(self.parts)} parts, {len(self)})"


def sources_for(scene, num_rays, offset=0):
    """Light name of each ray index (round-robin), without sampling anything."""
    return RoundRobinSources([node.light.name for node in scene.light_nodes], num_rays, offset)
