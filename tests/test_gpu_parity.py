"""Parity of the HIP engine, through the C ABI (host-buffer entry pvt_trace_bundle),
against the CPU referee and the committed reference fixtures.  Bar: integer/index
columns and tallies bit-exact; floating-point columns bit-exact too (the engine and
the oracle's portable mode share one arithmetic); recorder moment sums to 1e-12
relative (parallel summation order is not defined)."""
import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd.engine import _kernel, compile_scene, native
from pvtrace_amd.engine.emit import EmitterTables, emit_bundle
from tests import scenes
from tests.util import assert_bundles_identical, assert_same_tables, load_golden

pytestmark = pytest.mark.gpu

MODES = [(1, 64, 1000, 0), (0, 128, 50, 1), (7, 16, 1000, 2)]


def gpu_and_oracle(scene, n, mode, seed=42, emit_seed=123, **kw):
    record_every, max_events, maxsteps, emit_method = mode
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=emit_seed)
    gpu = _kernel.trace_bundle(compiled, pos, dirs, wl, seed, maxsteps, max_events, emit_method, 1,
                               record_every, **kw)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, seed, maxsteps, max_events, emit_method, 1,
                         record_every, math_mode=O.MATH_PORTABLE, **kw)
    return gpu, cpu


def _division_operands(rng, n, divisor):
    """Random operands over many binades plus the hard cases of a division: numerators whose
    quotient by `divisor(x)` lands on, or half an ulp away from, a representable number."""
    x = np.concatenate((rng.random(n // 2) * 10.0 ** rng.uniform(-30, 30, n // 2), rng.normal(size=n // 2) * 1e-9))
    q = rng.random(n // 2) * 10.0 ** rng.uniform(-12, 12, n // 2)
    d = divisor(q) if callable(divisor) else np.full_like(q, divisor)
    exact = (q.astype(np.longdouble) * d.astype(np.longdouble)).astype(np.float64)
    half = ((q.astype(np.longdouble) + np.spacing(q).astype(np.longdouble) / 2) * d.astype(np.longdouble)).astype(np.float64)
    return np.concatenate((x, exact, half, [0.0, 1.0, 2.99792458e10, 1.5, 400.0]))


@pytest.mark.parametrize("fn", ["log", "sin", "cos", "asin", "acos", "sqrt", "rcp",
                                "sincos_product", "uniform2", "ratio", "div_c", "div_n", "div_hist", "div_any",
                                "sin2pi", "cos2pi", "sqrt1m2", "rcp_normal", "div_normal", "sqrt_normal"])
def test_device_arithmetic_is_bit_identical_to_host(fn):
    """The premise of everything below: IEEE divide/sqrt, u64->f64 and pvt_math.h give the
    same bits on gfx950 (hipcc, -ffp-contract=off) as on the host (gcc)."""
    rng = np.random.default_rng(5)
    n = 400_000
    x = {
        "log": 1.0 - rng.random(n), "sin": rng.random(n) * 2 * np.pi, "cos": rng.random(n) * 2 * np.pi,
        "asin": rng.random(n) * 2 - 1, "acos": rng.random(n) * 2 - 1, "sqrt": rng.random(n) * 1e6,
        "rcp": rng.normal(size=n) * 1e3, "sincos_product": rng.random(n) * 2 * np.pi,
        "uniform2": np.floor(rng.random(n) * 2 ** 52), "ratio": rng.random(n) * 100,
        # the kernel's known-divisor division (5 multiply-adds) must equal the host's IEEE division
        "div_c": _division_operands(rng, n, 2.99792458e10), "div_n": _division_operands(rng, n, 1.5),
        "div_hist": _division_operands(rng, n, 400.0),
        "div_any": _division_operands(rng, n, lambda q: q * 0.7310585786300049 + 0.25),
        "sin2pi": rng.random(n), "cos2pi": rng.random(n), "sqrt1m2": rng.random(n) * 2 - 1, "div_normal": rng.random(n),
        # squares and their neighbours (ties and near-ties of the final rounding), the whole range, 0 and +inf
        "sqrt_normal": np.concatenate((
            rng.random(n), rng.random(n) * 10.0 ** rng.uniform(-200, 200, n), 1.0 - rng.random(n) ** 2,
            np.nextafter((rng.random(20000) * 3) ** 2, 0.0), np.nextafter((rng.random(20000) * 3) ** 2, 9.0),
            (rng.integers(1, 2 ** 26, 20000).astype(np.float64) * 2.0 ** -26) ** 2, [0.0, np.inf, 4.0, 2.0 ** -700, -1.0])),
        # the slab test's 1/d without operand scaling and fix-up: every normal operand with a normal reciprocal,
        # with the awkward mantissas (all ones, one above a power of two, powers of two)
        "rcp_normal": np.concatenate((
            rng.normal(size=n) * 10.0 ** rng.uniform(-299, 299, n), rng.random(n) * 2 - 1,
            np.ldexp(np.nextafter(2.0, 0.0), rng.integers(-996, 1000, 2000)), np.ldexp(np.nextafter(1.0, 2.0), rng.integers(-996, 1000, 2000)),
            np.ldexp(1.0, np.arange(-996, 1000)), -np.ldexp(np.nextafter(2.0, 0.0), rng.integers(-996, 1000, 2000)))),
    }[fn]
    if fn == "div_normal":    # x / (0.73 x + 0.25): quotients from 1e-280 to 1.37, both signs of x below the pole
        x = np.concatenate((rng.random(n) * 100, rng.random(n), 10.0 ** rng.uniform(-280, 2, n), -rng.random(n) * 0.3, [0.0]))
    if not fn.startswith("div_"):   # (a subnormal quotient is outside div_known's stated domain)
        x = np.concatenate((x, [1.0, 0.5, 1e-300, 0.9999999999999999]))
    if fn == "rcp_normal":
        x = x[(np.abs(x) >= 1e-300) & (np.abs(x) <= 1e300)]
    if fn == "sqrt_normal":
        x = x[(x <= 0.0) | (x >= 1e-200)]
    dev = native.selftest_math(O.MATH_FN[fn], x)
    host = O.math(fn, x, math_mode=O.MATH_PORTABLE)
    assert np.array_equal(dev, host, equal_nan=True), int(np.sum(dev != host))


@pytest.mark.parametrize("name", sorted(scenes.ALL_SCENES))
@pytest.mark.parametrize("mode", MODES)
def test_every_scene_is_bit_identical_to_the_oracle(name, mode):
    gpu, cpu = gpu_and_oracle(scenes.ALL_SCENES[name](), 3000, mode)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=name)
    if mode[0] == 1:
        assert cpu["counts"].min() >= 2 or name == "trapped_light"


@pytest.mark.parametrize("name", sorted(scenes.REFERENCE_SCENES))
def test_gpu_against_reference_kernel_fixtures(name):
    """Expected outputs came from the REFERENCE's compiled kernel.  The GPU differs from it
    only through <=1-ulp differences of log/sin/cos/asin/acos (and the direct evaluation of
    sin(acos c), cos(acos c), pvt_math.h), so: identical event sequences for (all but a couple of)
    rays, values equal to 1e-8, and Snell refraction directions bit-identical until a ray's first
    absorption (north_star)."""
    g = load_golden(f"trace_{name}.npz")
    compiled = compile_scene(scenes.REFERENCE_SCENES[name]())
    assert_same_tables(compiled, g)
    m = int(g["par_max_events"])
    gpu = _kernel.trace_bundle(compiled, g["in_pos"], g["in_dir"], g["in_wl"], int(g["par_seed"]),
                               int(g["par_maxsteps"]), m, int(g["par_emit_method"]), 1,
                               int(g["par_record_every"]))
    ref = {k[4:]: g[k] for k in g.files if k.startswith("ref_")}
    n = ref["counts"].shape[0]
    same = snell = 0
    for j in range(n):
        rows = slice(j * m, j * m + int(ref["counts"][j]))
        if ref["counts"][j] == gpu["counts"][j] and np.array_equal(ref["kind"][rows], gpu["kind"][rows]):
            same += 1
            for key in ("hit", "container", "adjacent", "component", "source"):
                assert np.array_equal(gpu[key][rows], ref[key][rows]), (name, j, key)
            assert np.allclose(gpu["position"][rows], ref["position"][rows], rtol=0, atol=1e-8)
            assert np.allclose(gpu["direction"][rows], ref["direction"][rows], rtol=0, atol=1e-8)
            assert np.allclose(gpu["wavelength"][rows], ref["wavelength"][rows], rtol=1e-10)
            assert np.allclose(gpu["duration"][rows], ref["duration"][rows], rtol=1e-9, atol=1e-22)
        for row in range(rows.start, rows.stop):
            if ref["kind"][row] == 3 or gpu["kind"][row] != ref["kind"][row]:
                break   # first absorption, or (counted by `same`) the sequence parted ways at a draw
            if ref["kind"][row] == 2:
                assert np.array_equal(gpu["direction"][row], ref["direction"][row]), (name, j, row)
                snell += 1
    assert same >= n - 2, (name, same, n)
    if name != "trapped_light":
        assert snell > 100


@pytest.mark.parametrize("name", ["fresnel_box", "lsc_equivalent", "nested_cylinders", "touching_boxes"])
def test_tally_mode_rays_that_start_outside_or_on_the_root(name):
    """In tally mode without an `exit` recorder the kernel visits the root last and only for the lanes that
    need its exact distance, deciding the others by a lower bound that assumes the photon is INSIDE the
    root.  Rays that start outside the root, on its surface, or a hair inside it must take the exact path
    and still match the referee (integer tallies exactly)."""
    scene = scenes.ALL_SCENES[name]()
    compiled = compile_scene(scene)
    assert compiled.rec_node.shape[0] > 0 or name in ("fresnel_box", "touching_boxes")
    rng = np.random.default_rng(9)
    n = 4000
    root_size = compiled.geom_params[compiled.root_id]
    reach = float(root_size[0]) if compiled.geom_type[compiled.root_id] == 1 else 0.5 * float(max(root_size[:3]))
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    scale = np.concatenate([np.full(n // 4, 1.2), np.full(n // 4, 1.0), np.full(n // 4, 1.0 - 1e-13), rng.random(n - 3 * (n // 4))])
    pos = u * (reach * scale)[:, None]
    dirs = -u + 0.3 * rng.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    wl = np.full(n, 555.0)
    gpu = _kernel.trace_bundle(compiled, pos, dirs, wl, 5, 200, 16, 0, 1, 0)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, 5, 200, 16, 0, 1, 0, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=name)
    # and with histories (the exact path for every lane) the event logs agree as well
    gpu = _kernel.trace_bundle(compiled, pos, dirs, wl, 5, 200, 16, 0, 1, 1)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, 5, 200, 16, 0, 1, 1, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=name + " histories")


@pytest.mark.parametrize("world,index", [("box", 1.0), ("sphere", 1.0), ("box", 1.5), ("sphere", 1.05)])
def test_fused_exit_clears_grazing_departures_exactly(world, index):
    """One unrotated box in an empty, unobserved world: tally launches end a photon when it is sent away from the
    box's surface on the outside (its exit step could not be seen by anybody).  The kernel must PROVE that the
    reference's next slab test would ignore the box; photons leaving at grazing angles, whose rounding error on
    the face's axis the reference turns into a genuine re-entry (a second crossing of the face), have to take the
    step.  A third of the rays start just inside a face and head out of it at 1e-9..1e-2 rad, a third start just
    outside and graze it (near-total Fresnel reflection at index 1.5), the rest are random."""
    from pvtrace_amd import Box, Material, Node, Scene, Sphere
    geometry = Box((300.0, 300.0, 300.0), material=Material(refractive_index=1.0)) if world == "box" else \
        Sphere(radius=200.0, material=Material(refractive_index=1.0))
    root = Node(name="world", geometry=geometry)
    slab = Node(name="slab", geometry=Box((2.0, 3.0, 1.0), material=Material(refractive_index=index)), parent=root,
                recorders=scenes.face_recorders(hist=False))
    centre = np.array([0.3, -0.2, 0.5])
    slab.location = tuple(centre)
    compiled = compile_scene(Scene(root))
    rng = np.random.default_rng(77)
    n = 600_000
    half = np.array([1.0, 1.5, 0.5])
    local = (rng.random((n, 3)) * 2 - 1) * half * 0.999
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    k = 2 * n // 3
    axis = rng.integers(0, 3, k)
    sign = rng.choice([-1.0, 1.0], k)
    gap = 10.0 ** rng.uniform(-12, -3, k)
    slope = 10.0 ** rng.uniform(-9, -2, k)
    inside = np.arange(k) < k // 2
    rows = np.arange(k)
    local[rows, axis] = sign * (half[axis] + np.where(inside, -gap, gap))
    d[rows, axis] = np.where(inside, sign, -sign) * slope      # out of the face from inside, into it from outside
    d /= np.linalg.norm(d, axis=1)[:, None]
    pos = local + centre
    wl = np.full(n, 555.0)
    gpu = _kernel.trace_bundle(compiled, pos, d, wl, 3, 1000, 16, 0, 1, 0)
    cpu = O.trace_bundle(compiled, pos, d, wl, 3, 1000, 16, 0, 1, 0, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=world)
    assert int(gpu["rec_crossings"].sum()) > n // 2
    # small step budgets: a photon whose NEXT step would be the kill is not ended early either
    for maxsteps in (1, 2, 3):
        gpu = _kernel.trace_bundle(compiled, pos[::10], d[::10], wl[::10], 3, maxsteps, 16, 0, 1, 0)
        cpu = O.trace_bundle(compiled, pos[::10], d[::10], wl[::10], 3, maxsteps, 16, 0, 1, 0, math_mode=O.MATH_PORTABLE)
        assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"{world} maxsteps={maxsteps}")


@pytest.mark.parametrize("index", [1.5, 1.05, 2.417])
def test_critical_angle_is_decided_like_the_reference_to_the_last_bit(index):
    """The reference decides total internal reflection by `acos(cosine) > asin(n2/n1)`; the kernel compares the
    cosine with a threshold the host derives from the same pvt_acos / pvt_asin and proves over the neighbouring
    doubles.  Rays that meet the top face of a cube at every double within 3000 ulps of that threshold (and a
    coarse sweep of the rest) must take the same branch, draw the same numbers and leave the same rows."""
    from pvtrace_amd import Box, Material, Node, Scene, Sphere
    root = Node(name="world", geometry=Sphere(radius=20.0, material=Material(refractive_index=1.0)))
    Node(name="cube", geometry=Box((1.0, 1.0, 1.0), material=Material(refractive_index=index)), parent=root,
         recorders=scenes.face_recorders(hist=False))
    compiled = compile_scene(Scene(root))
    centre = float(np.sqrt(1.0 - 1.0 / index ** 2))        # cosine of the critical angle, to an ulp or so
    near = centre + np.arange(-3000, 3001) * np.spacing(centre)
    cz = np.concatenate([near, np.linspace(1e-3, 1.0, 2000)])
    d = np.column_stack([np.sqrt(np.maximum(0.0, 1.0 - cz * cz)), np.zeros_like(cz), cz])
    pos = np.tile([0.0, 0.0, 0.0], (len(cz), 1)) - d * 1e-3     # the top face is met first
    wl = np.full(len(cz), 555.0)
    for rec_every in (1, 0):
        gpu = _kernel.trace_bundle(compiled, pos, d, wl, 11, 40, 48, 0, 1, rec_every)
        cpu = O.trace_bundle(compiled, pos, d, wl, 11, 40, 48, 0, 1, rec_every, math_mode=O.MATH_PORTABLE)
        assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"n={index} rec_every={rec_every}")
    kinds = gpu["rec_crossings"]
    assert kinds.sum() > 0


def test_ragged_and_empty_bundles():
    scene = scenes.bench_slab(recorders=True)
    for n in (1, 2, 63, 64, 65, 127, 129, 257, 1000):
        gpu, cpu = gpu_and_oracle(scene, n, (1, 32, 1000, 0), seed=n)
        assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"n={n}")
        gpu, cpu = gpu_and_oracle(scene, n, (3, 32, 1000, 0), seed=n)
        assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"n={n} every 3")
    compiled = compile_scene(scene)
    empty = _kernel.trace_bundle(compiled, np.zeros((0, 3)), np.zeros((0, 3)), np.zeros(0), 1, 10, 8, 0, 1, 1)
    assert empty["counts"].shape == (0,) and empty["kind"].shape == (0,)
    assert not empty["rec_distinct"].any()


def test_streamed_bundles_equal_one_call():
    """reference api.py:249-264: bundles with seed offsets union to a single call."""
    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    n = 20000
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=9)
    whole = _kernel.trace_bundle(compiled, pos, dirs, wl, 1234, 1000, 128, 0, 1, 0)
    acc = None
    for start in range(0, n, 6000):
        stop = min(n, start + 6000)
        part = _kernel.trace_bundle(compiled, pos[start:stop], dirs[start:stop], wl[start:stop],
                                    1234, 1000, 128, 0, 1, 0, ray_offset=start)
        same_as_seed_shift = _kernel.trace_bundle(compiled, pos[start:stop], dirs[start:stop],
                                                  wl[start:stop], 1234 + start, 1000, 128, 0, 1, 0)
        for key in ("rec_distinct", "rec_crossings", "rec_bins"):
            assert np.array_equal(part[key], same_as_seed_shift[key])
        acc = part if acc is None else {k: acc[k] + part[k] for k in acc}
    for key in ("rec_distinct", "rec_crossings", "rec_bins"):
        assert np.array_equal(acc[key], whole[key]), key
    assert np.allclose(acc["rec_sums"], whole["rec_sums"], rtol=1e-11)


@pytest.mark.parametrize("name", ["kitchen_sink", "lsc_equivalent", "coated_slab", "lambertian_sheet", "hist_lamp"])
def test_device_emission_matches_oracle_emitter(name):
    scene = scenes.ALL_SCENES[name]()
    compiled = compile_scene(scene)
    tab = EmitterTables(scene)
    n = 5000
    pos, dirs, wl = O.emit(tab, n, emit_seed=77, ray_offset=1000)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, 5, 1000, 48, 0, 1, 1, ray_offset=1000,
                         math_mode=O.MATH_PORTABLE)
    gpu = _kernel.trace_bundle(compiled, None, None, n, 5, 1000, 48, 0, 1, 1, ray_offset=1000,
                               emitter=tab, emit_seed=77)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=name)
    # GENERATE rows carry the emitted ray itself
    assert np.array_equal(gpu["position"][::48], pos) and np.array_equal(gpu["wavelength"][::48], wl)


def _scene_with_many_recorders(n_rec):
    from pvtrace_amd.engine import Histogram, Recorder
    scene = scenes.bench_slab(recorders=False)
    slab = scene.root.children[0]
    events = ["entering", "escaping", "reflected", "lost"]
    slab.recorders = [Recorder(f"r{i}", event=events[i % 4],
                               histograms=[Histogram("wavelength", 300, 1000, 7)] if i % 5 == 0 else [])
                      for i in range(n_rec)]
    return scene


def test_more_than_64_recorders_uses_the_wide_seen_mask():
    gpu, cpu = gpu_and_oracle(_scene_with_many_recorders(150), 4000, (0, 16, 1000, 0))
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12)
    assert cpu["rec_distinct"][64:].sum() > 0
    gpu, cpu = gpu_and_oracle(_scene_with_many_recorders(256), 2000, (4, 16, 1000, 0))
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12)


def test_histograms_too_large_for_lds_fall_back_to_global_atomics():
    from pvtrace_amd.engine import Heatmap, Recorder
    scene = scenes.bench_slab(recorders=False)
    slab = scene.root.children[0]
    slab.recorders = [Recorder("fine-map", event="entering",
                               histograms=[Heatmap("x", "y", (-2.5, 2.5, 300), (-2.5, 2.5, 300))]),
                      Recorder("lost", event="lost")]
    gpu, cpu = gpu_and_oracle(scene, 6000, (0, 16, 1000, 0))
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12)
    assert cpu["rec_bins"].sum() == cpu["rec_distinct"][0] > 0


@pytest.mark.parametrize("tables", ["as the library places them", "heads", "global"])
def test_spectra_too_large_for_lds_are_read_from_hbm(tables, monkeypatch):
    """Three placements of the scene tables (kernel template TAB_LDS): everything in LDS; records and class tables in LDS
    with the spectra and their guide tables in global memory (what the library picks here); everything in global memory
    (what is left when not even the records fit) -- the last two also forced on the headline-sized scene."""
    from pvtrace_amd import Absorber, Box, Light, Luminophore, Material, Node, Scene, Sphere
    if tables != "as the library places them":
        monkeypatch.setenv("PVT_TABLES", tables)
        gpu, cpu = gpu_and_oracle(scenes.lsc_equivalent(), 2000, (1, 48, 1000, 0))
        assert_bundles_identical(gpu, cpu, sums_rtol=1e-12)
    from pvtrace_amd.material import gaussian
    x = np.linspace(300.0, 1000.0, 6000)        # 4 pooled tables x 6000 doubles = 192 KB > LDS
    world = Node(name="world", geometry=Sphere(10.0, material=Material(1.0)))
    Node(name="slab", parent=world, geometry=Box((5.0, 5.0, 1.0), material=Material(1.5, components=[
        Luminophore(np.column_stack((x, 5.0 * gaussian(x, 1.0, 480.0, 40.0))),
                    emission=np.column_stack((x, gaussian(x, 1.0, 600.0, 40.0))), quantum_yield=0.9),
        Absorber(0.3)])))
    light = Node(name="light", parent=world, light=Light())
    light.location = (0.0, 0.0, -3.0)
    gpu, cpu = gpu_and_oracle(Scene(world), 3000, (1, 48, 1000, 0))
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12)


def test_node_limit_raises_value_error_like_the_reference():
    from pvtrace_amd import Box, Light, Material, Node, Scene
    world = Node(name="w", geometry=Box((1000.0, 10.0, 10.0), material=Material(1.0)))
    for i in range(128):
        Node(name=f"b{i}", parent=world, location=(-400.0 + 6.0 * i, 0.0, 0.0),
             geometry=Box((1.0, 1.0, 1.0), material=Material(1.5)))
    Node(name="l", parent=world, light=Light())
    compiled = compile_scene(Scene(world))
    with pytest.raises(ValueError):
        _kernel.trace_bundle(compiled, np.zeros((4, 3)), np.tile((0.0, 0.0, 1.0), (4, 1)),
                             np.full(4, 555.0), 1, 10, 8, 0, 1, 0)


def test_128_nodes_trace_correctly():
    from pvtrace_amd import Box, Light, Material, Node, Scene
    from pvtrace_amd.material import Cone
    world = Node(name="w", geometry=Box((1000.0, 10.0, 10.0), material=Material(1.0)))
    for i in range(127):
        Node(name=f"b{i}", parent=world, location=(-400.0 + 6.0 * i, 0.0, 0.0),
             geometry=Box((1.0, 1.0, 1.0), material=Material(1.2 + 0.003 * i)))
    light = Node(name="l", parent=world, light=Light(direction=Cone(0.001)))
    light.location = (-450.0, 0.0, 0.0)
    light.look_at((1.0, 0.0, 0.0))
    gpu, cpu = gpu_and_oracle(Scene(world), 600, (1, 600, 2000, 0))
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12)
    assert cpu["counts"].max() > 200


def _mesh_ball_scene(subdivisions):
    from pvtrace_amd import Light, Material, Mesh, Node, Scatterer, Scene, Sphere
    from pvtrace_amd.engine import Histogram, Recorder
    from pvtrace_amd.material import isotropic

    world = Node(name="world", geometry=Sphere(10.0, material=Material(refractive_index=1.0)))
    ball = Node(name="ball", parent=world, geometry=Mesh.icosphere(subdivisions, 1.0, material=Material(
        refractive_index=1.5, components=[Scatterer(1.2, phase_function=isotropic, name="haze")])))
    ball.location = (0.0, 0.0, 2.0)
    ball.rotate(0.3, (0.2, 1.0, 0.1))
    ball.recorders = [Recorder("in", event="entering"), Recorder("out", event="escaping",
                                                                 histograms=[Histogram("angle", 0, 1.6, 16)])]
    Node(name="lamp", parent=world, light=Light(name="lamp"))
    return Scene(world), ball


def test_bvh_walk_finds_exactly_the_crossings_of_the_brute_force_oracle():
    """5120 faces (a 12-level BVH in HBM) against the oracle's loop over every face; rays are
    spread over the ball, and a block of them is aimed EXACTLY at mesh vertices and edge
    midpoints (zero edge functions: the half-plane tie rule decides)."""
    scene, ball = _mesh_ball_scene(4)
    compiled = compile_scene(scene)
    n = 4000
    rng = np.random.default_rng(12)
    pos = np.zeros((n, 3)); pos[:, :2] = rng.uniform(-1.1, 1.1, (n, 2))
    dirs = np.tile([0.0, 0.0, 1.0], (n, 1))
    l2w = np.asarray(ball.transformation_to(scene.root))
    verts = ball.geometry.vertices @ l2w[:3, :3].T + l2w[:3, 3]
    edges = 0.5 * (verts[ball.geometry.faces[:, 0]] + verts[ball.geometry.faces[:, 1]])
    aim = np.concatenate((verts[:600], edges[:600]))
    d = aim - pos[:1200]
    dirs[:1200] = d / np.linalg.norm(d, axis=1)[:, None]
    wl = np.full(n, 555.0)
    gpu = _kernel.trace_bundle(compiled, pos, dirs, wl, 5, 1000, 48, 0, 1, 1)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, 5, 1000, 48, 0, 8, 1, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what="icosphere-4")
    assert cpu["rec_distinct"][0] > 0.6 * n


@pytest.mark.parametrize("top_bytes", ["0", "96", "224", "4064", "default"])
def test_bvh_walk_is_the_same_whatever_part_of_the_tree_sits_in_lds(top_bytes, monkeypatch):
    """The walk reads the top levels of a tree from a copy in LDS and the rest from global memory; cursors and skip links
    name either (pvt_bvh.h: stage_top).  No copy at all (0), the root and its children (3 records = 96 bytes), three levels,
    seven levels, and what the library picks: always the brute-force oracle's crossings, in a scene of TWO meshes of
    different size (the copy is shared out between the trees) plus an analytic node."""
    import pvtrace_amd as pv
    if top_bytes != "default":
        monkeypatch.setenv("PVT_MESH_TOP_BYTES", top_bytes)
    scene, ball = _mesh_ball_scene(3)
    small = pv.Node(name="small", parent=scene.root,
                    geometry=pv.Mesh.icosphere(1, 0.3, material=pv.Material(refractive_index=1.3)))
    small.location = (2.0, 0.3, 0.1)
    compiled = compile_scene(scene)
    n = 3000
    rng = np.random.default_rng(31)
    pos = rng.uniform(-2.6, 2.6, (n, 3)); pos[:, 2] = rng.uniform(-1.5, 1.5, n)
    target = np.where(rng.random((n, 1)) < 0.5, np.array([[0.0, 0.0, 2.0]]), np.array([[2.0, 0.3, 0.1]])) + rng.normal(0, 0.25, (n, 3))
    dirs = target - pos
    dirs /= np.linalg.norm(dirs, axis=1)[:, None]
    wl = np.full(n, 555.0)
    gpu = _kernel.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 0, 1, 1)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 0, 8, 1, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"two meshes, copy of {top_bytes} bytes")


def test_every_kernel_variant_at_once():
    """The template axes together: mesh geometry (MESH), 6000-point spectra that do not fit LDS
    (TAB_LDS=false), 150 recorders (wide seen mask), sampled event log (RECORD) and device-side
    emission (EMIT) — against the oracle's emitter + tracer on the same seeds."""
    from pvtrace_amd import Absorber, Light, Luminophore, Material, Mesh, Node, Scene, Sphere
    from pvtrace_amd.engine import Histogram, Recorder
    from pvtrace_amd.light import ConstantWavelengthMask
    from pvtrace_amd.material import gaussian

    x = np.linspace(300.0, 1000.0, 6000)
    world = Node(name="world", geometry=Sphere(10.0, material=Material(1.0)))
    gem = Node(name="gem", parent=world, geometry=Mesh.icosphere(2, 1.2, material=Material(1.5, components=[
        Luminophore(np.column_stack((x, 3.0 * gaussian(x, 1.0, 480.0, 40.0))),
                    emission=np.column_stack((x, gaussian(x, 1.0, 600.0, 40.0))), quantum_yield=0.9, name="dye"),
        Absorber(0.2, name="host")])))
    gem.location = (0.0, 0.0, 2.5)
    gem.rotate(0.5, (1.0, 0.2, 0.0))
    events = ["entering", "escaping", "reflected", "lost"]
    gem.recorders = [Recorder(f"r{i}", event=events[i % 4],
                              histograms=[Histogram("wavelength", 300, 1000, 9)] if i % 7 == 0 else [])
                     for i in range(150)]
    Node(name="lamp", parent=world, light=Light(wavelength=ConstantWavelengthMask(470.0), name="lamp"))
    scene = Scene(world)
    compiled = compile_scene(scene)
    emitter = EmitterTables(scene)
    n, seed, emit_seed, record_every, max_events = 5000, 21, 8, 3, 40
    gpu = _kernel.trace_bundle(compiled, None, None, n, seed, 1000, max_events, 0, 1, record_every,
                               emitter=emitter, emit_seed=emit_seed)
    pos, dirs, wl = O.emit(emitter, n, emit_seed=emit_seed)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 4, record_every,
                         math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what="all-variants")
    assert cpu["rec_distinct"][64:].sum() > 0 and cpu["counts"].max() > 4


@pytest.mark.parametrize("extensions", [False, True])
@pytest.mark.parametrize("seed", range(30))
def test_random_scenes_gpu_equals_oracle(seed, extensions):
    """Differential fuzz (tests/fuzz.py): random scenes -- with `extensions` also meshes, coatings,
    histogram-sampled spectra and source-filtered recorders -- traced on the GPU and by the oracle,
    alternating host rays and device emission: every output array identical."""
    from tests.fuzz import random_scene

    scene = random_scene(1000 * int(extensions) + seed, extensions=extensions)
    compiled = compile_scene(scene)
    mode = [(1, 48, 300, 0), (3, 16, 40, 1), (0, 8, 300, 2)][seed % 3]
    record_every, max_events, maxsteps, emit_method = mode
    n = 1500
    try:
        emitter = EmitterTables(scene) if seed % 2 else None
    except Exception:
        emitter = None
    if emitter is not None:
        pos, dirs, wl = O.emit(emitter, n, emit_seed=seed)
        gpu = _kernel.trace_bundle(compiled, None, None, n, 9 + seed, maxsteps, max_events, emit_method, 1,
                                   record_every, emitter=emitter, emit_seed=seed)
    else:
        pos, dirs, wl, _ = emit_bundle(scene, n, seed=seed)
        gpu = _kernel.trace_bundle(compiled, pos, dirs, wl, 9 + seed, maxsteps, max_events, emit_method, 1,
                                   record_every)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, 9 + seed, maxsteps, max_events, emit_method, 4,
                         record_every, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"fuzz scene {seed} ext={extensions}")
