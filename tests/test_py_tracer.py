"""The pure-Python per-ray tracer (oracle/py_tracer.py, a restatement of the reference's
pvtrace/algorithm/photon_tracer.py over the host scene API) against the table-driven path.

This is the reference's own validation scheme for its engine (tests/test_engine.py:95-167,
:321-350): per-ray event-count means of a few hundred Python-traced rays must agree with the
kernel's within Monte-Carlo error (Welch, 5 sigma).  On CPU the kernel is played by the C referee;
the GPU test does the same against the HIP engine.  BASELINE configs[0] is the first case."""
import math

import numpy as np
import pytest

from oracle import oracle as O
from oracle import py_tracer as P
from pvtrace_amd.engine import compile_scene, tally_histories
from pvtrace_amd.engine.api import EngineResult
from pvtrace_amd.engine.emit import emit_bundle
from pvtrace_amd.light import Event
from tests import scenes

KINDS = (Event.GENERATE, Event.ABSORB, Event.EMIT, Event.SCATTER, Event.REFLECT, Event.TRANSMIT,
         Event.NONRADIATIVE, Event.REACT, Event.EXIT, Event.KILL)


def python_counts(scene, n, seed):
    """(n, kinds) per-ray event counts from the Python tracer, and the histories."""
    np.random.seed(seed)
    histories = [list(P.step_forward(scene, ray)) for ray in scene.emit(n)]
    table = np.zeros((n, len(KINDS)))
    for j, history in enumerate(histories):
        for _, event, _ in history:
            table[j, KINDS.index(event)] += 1
    return table, histories


def table_counts(data, max_events):
    """Same table from a `data` dict of the kernel / referee (all rays recorded)."""
    n = len(data["counts"])
    kinds = data["kind"].reshape(n, max_events)
    valid = np.arange(max_events)[None, :] < data["counts"][:, None]
    return np.stack([((kinds == k.value) & valid).sum(axis=1) for k in KINDS], axis=1).astype(float)


def assert_means_close(a, b, nsigma=5.0):
    """Welch comparison of per-ray event-count means (reference tests/test_engine.py:126-136)."""
    for k, kind in enumerate(KINDS):
        se = math.sqrt(a[:, k].var(ddof=1) / len(a) + b[:, k].var(ddof=1) / len(b))
        assert abs(a[:, k].mean() - b[:, k].mean()) <= nsigma * se + 1e-9, (
            kind.name, a[:, k].mean(), b[:, k].mean(), nsigma * se)


def referee_counts(scene, n, seed, max_events=256):
    compiled = compile_scene(scene)
    pos, dirs, wl, src = emit_bundle(scene, n, seed=seed)
    data = O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, max_events, 0, 8, 1)
    assert data["counts"].max() < max_events - 1
    return table_counts(data, max_events), EngineResult(compiled, data, src, max_events, 1, 0.0)


def test_config1_hello_world_one_thousand_rays_in_pure_python():
    """BASELINE configs[0]: glass sphere in the world sphere, 1 000 rays, Python per-ray path on
    the CPU.  Expected per-ray means from SURVEY.md §8(d): GENERATE 1, TRANSMIT ~1.91,
    REFLECT ~0.087, EXIT 1."""
    py, histories = python_counts(scenes.hello_world(), 1000, seed=1)
    mean = dict(zip(KINDS, py.mean(axis=0)))
    assert mean[Event.GENERATE] == 1.0 and mean[Event.EXIT] == 1.0 and mean[Event.KILL] == 0.0
    assert mean[Event.TRANSMIT] == pytest.approx(1.91, abs=0.03)
    assert mean[Event.REFLECT] == pytest.approx(0.087, abs=0.03)
    ref, _ = referee_counts(scenes.hello_world(), 20000, seed=7, max_events=64)
    assert_means_close(py, ref)
    # every history is a connected polyline that ends on the world sphere
    for history in histories[:50]:
        assert history[-1][1] == Event.EXIT
        assert np.isclose(np.linalg.norm(history[-1][0].position), 10.0)


@pytest.mark.parametrize("name,n_python", [("fresnel_box", 800), ("bench_slab", 400), ("lsc_equivalent", 400),
                                           ("nested_cylinders", 600), ("mesh_gem", 600), ("l_prism", 500)])
def test_python_tracer_and_table_driven_path_agree(name, n_python):
    """reference tests/test_engine.py:139-167 (Fresnel scene, dye slab) plus the headline LSC and the
    nested rotated cylinders, and the faceted gem (triangle meshes inside a mesh world, HG
    scatterer in a rotated frame) and the non-convex L-prism lit from inside (crossing-parity containment):
    two implementations that share no code below the scene objects."""
    scene = scenes.bench_slab(recorders=False) if name == "bench_slab" else scenes.ALL_SCENES[name]()
    py, _ = python_counts(scene, n_python, seed=42)
    ref, _ = referee_counts(scene, 20000, seed=7)
    assert_means_close(py, ref)


def test_python_tracer_tallies_match_statistically():
    """reference tests/test_engine.py:321-350: recorders tallied from Python histories vs the
    kernel-side accumulators, within 5 sigma."""
    scene = scenes.lsc_equivalent()
    n_python, n_table = 400, 20000
    _, histories = python_counts(scene, n_python, seed=5)
    python_tallies = tally_histories(scene, histories)
    _, result = referee_counts(scene, n_table, seed=13)
    for name in ("top", "bottom", "lost", "entering", "reflected"):
        a, b = python_tallies[name].rays, result.recorders[name].rays
        p = (a + b) / (n_python + n_table)
        se = math.sqrt(max(p * (1 - p), 1e-12) * (1 / n_python + 1 / n_table))
        assert abs(a / n_python - b / n_table) <= 5 * se, (name, a / n_python, b / n_table)


def test_coatings_python_delegates_against_the_coating_tables():
    """Config 5 (Coatings notebook scene + scatterer): the host `CoatedSurfaceDelegate` is CALLED by
    the Python tracer, the flattener lowers the same coatings to tables for the kernel.  The
    reference engine cannot express this scene, so this statistical agreement is its check."""
    scene = scenes.coated_slab()
    py, _ = python_counts(scene, 800, seed=3)
    ref, _ = referee_counts(scene, 20000, seed=9)
    assert_means_close(py, ref)


# -- the REFERENCE's Python tracer calling the REFERENCE's LSC delegates (tests/golden/lsc_tracer.npz) ------------
def lsc_product_cases():
    from pvtrace_amd import light as product_light, material as product_material
    from pvtrace_amd.data import lumogen_f_red_305
    from pvtrace_amd.device.lsc import LSC

    return scenes.lsc_tracer_cases(LSC, product_material.cone, product_light.rectangular_mask, lumogen_f_red_305)


def reference_lsc_counts(g, name):
    """(n, kinds) in this file's KINDS order from the fixture's by-event-value table."""
    by_value = g[f"{name}/counts"].astype(float)
    return np.stack([by_value[:, kind.value] for kind in KINDS], axis=1)


def facet_classes(last, where, size):
    """What the reference's `LSC.simulate` keeps of a ray (device/lsc.py:349-359) put into classes: (last event, facet of
    the stored exit position) -- facet 0 for a position inside the slab, 1..6 for -x +x -y +y -z +z."""
    half = 0.5 * np.asarray(size, dtype=float)
    facet = np.zeros(len(last), dtype=int)
    for axis in range(3):
        facet[np.isclose(where[:, axis], -half[axis], atol=1e-6) & (facet == 0)] = 1 + 2 * axis
        facet[np.isclose(where[:, axis], half[axis], atol=1e-6) & (facet == 0)] = 2 + 2 * axis
    return last.astype(int) * 8 + facet


def data_last_and_where(data, max_events):
    counts = data["counts"].astype(int)
    base = np.arange(len(counts)) * max_events
    last = data["kind"][base + counts - 1]
    row = np.where(last == Event.EXIT.value, base + counts - 2, base + counts - 1)
    return last, data["position"][row]


def assert_fractions_close(a, b, nsigma=5.0):
    """Two-sample binomial comparison of every class's share."""
    for cls in np.union1d(np.unique(a), np.unique(b)):
        pa, pb = (a == cls).mean(), (b == cls).mean()
        p = ((a == cls).sum() + (b == cls).sum()) / (len(a) + len(b))
        se = math.sqrt(max(p * (1 - p), 1e-12) * (1 / len(a) + 1 / len(b)))
        assert abs(pa - pb) <= nsigma * se + 1e-9, (int(cls) // 8, int(cls) % 8, pa, pb, nsigma * se)


LSC_CASE_SIZES = {"default": (5.0, 5.0, 1.0), "cells": (5.0, 5.0, 1.0), "custom": (8.0, 4.0, 0.5)}


@pytest.mark.parametrize("name", ["default", "cells", "custom"])
def test_references_tracer_with_its_lsc_delegates_against_the_coating_tables(name):
    """SURVEY §8(c)(5).  tests/golden/lsc_tracer.npz: 3 000 rays per device traced by the REFERENCE's
    `photon_tracer.follow` through the scene the REFERENCE's `LSC` class builds, its `OptionalMirrorAndSolarCell` /
    `AirGapMirror` delegates called at every hit (solar cells on the edges, back-surface mirror, lambertian air-gap
    mirror).  The product builds the same device, lowers the delegates to coating tables, and the table-driven path
    must give the same per-ray event-count means (Welch, 5 sigma) and the same shares of (last event, exit facet)."""
    from tests.util import load_golden

    g = load_golden("lsc_tracer.npz")
    device = lsc_product_cases()[name]
    mine, result = referee_counts(device.scene, 20000, seed=21)
    assert_means_close(reference_lsc_counts(g, name), mine)
    last, where = data_last_and_where(result.data, 256)
    assert_fractions_close(facet_classes(g[f"{name}/last"], g[f"{name}/where"], LSC_CASE_SIZES[name]),
                           facet_classes(last, where, LSC_CASE_SIZES[name]))


@pytest.mark.gpu
def test_references_tracer_with_its_lsc_delegates_against_the_gpu_engine():
    from pvtrace_amd import engine
    from tests.util import load_golden

    g = load_golden("lsc_tracer.npz")
    for name, device in lsc_product_cases().items():
        result = engine.simulate(device.scene, 50000, seed=23, max_events=512)
        # (a photon in 10^5 is trapped between the mirror and the faces for hundreds of events: the few histories that
        # fill their rows end in KILL and would only blur the comparison by their own share)
        assert (result.data["counts"] >= 511).sum() <= 2
        assert_means_close(reference_lsc_counts(g, name), table_counts(result.data, 512))
        last, where = data_last_and_where(result.data, 512)
        assert_fractions_close(facet_classes(g[f"{name}/last"], g[f"{name}/where"], LSC_CASE_SIZES[name]),
                               facet_classes(last, where, LSC_CASE_SIZES[name]))


# -- BASELINE configs[4] under the REFERENCE's tracer and the Coatings notebook's own delegate (tests/golden/cfg5_tracer.npz) ----
def test_config5_references_tracer_with_the_notebooks_mirror_against_the_coating_tables():
    """SURVEY §8(c)(5) / §8(d) cfg5.  tests/golden/cfg5_tracer.npz: 20 000 rays traced by the REFERENCE's
    `photon_tracer.follow` through the Coatings notebook's scene (cell 5) + `Scatterer(1.0)`, its `PartialTopSurfaceMirror`
    (cell 3, exec'd from the notebook where it lies by tests/golden/make_golden.py:make_cfg5_tracer) called at every hit.
    The product's cfg5 (benchmarks/configs.py) states the same mirror as a `Coating` with a region, lowered to tables:
    same per-ray event-count means (Welch, 5 sigma) and the same shares of (last event, exit facet) -- here on the C
    referee, below on the GPU."""
    from benchmarks.configs import cfg5_coated_slab
    from tests.util import load_golden

    g = load_golden("cfg5_tracer.npz")
    mine, result = referee_counts(cfg5_coated_slab(), 40000, seed=31)
    assert_means_close(reference_lsc_counts(g, "cfg5"), mine)
    last, where = data_last_and_where(result.data, 256)
    assert_fractions_close(facet_classes(g["cfg5/last"], g["cfg5/where"], (10.0, 10.0, 1.0)),
                           facet_classes(last, where, (10.0, 10.0, 1.0)))
    # the mirror is where the notebook puts it: a quarter of the lamp's 5 x 5 cm footprint lies on it, and those rays are
    # turned back outside the glass without another event (GENERATE, REFLECT, EXIT) -- in the reference's run and in ours
    ref_counts = g["cfg5/counts"].astype(int)
    ref_bounced = (ref_counts.sum(axis=1) == 3) & (ref_counts[:, Event.REFLECT.value] == 1) & (g["cfg5/where"][:, 0] > 0) & (g["cfg5/where"][:, 1] > 0)
    assert abs(ref_bounced.mean() - 0.25) < 5 * math.sqrt(0.25 * 0.75 / len(ref_counts))
    my_bounced = (mine.sum(axis=1) == 3) & (mine[:, KINDS.index(Event.REFLECT)] == 1) & (where[:, 0] > 0) & (where[:, 1] > 0)
    assert abs(my_bounced.mean() - 0.25) < 5 * math.sqrt(0.25 * 0.75 / len(mine))


@pytest.mark.gpu
def test_config5_references_tracer_with_the_notebooks_mirror_against_the_gpu_engine():
    from benchmarks.configs import cfg5_coated_slab
    from pvtrace_amd import engine
    from tests.util import load_golden

    g = load_golden("cfg5_tracer.npz")
    result = engine.simulate(cfg5_coated_slab(), 100000, seed=33, max_events=256)
    assert (result.data["counts"] >= 255).sum() == 0
    assert_means_close(reference_lsc_counts(g, "cfg5"), table_counts(result.data, 256))
    last, where = data_last_and_where(result.data, 256)
    assert_fractions_close(facet_classes(g["cfg5/last"], g["cfg5/where"], (10.0, 10.0, 1.0)),
                           facet_classes(last, where, (10.0, 10.0, 1.0)))


@pytest.mark.gpu
def test_python_tracer_against_the_gpu_engine():
    from pvtrace_amd import engine

    for name, n_python in (("hello_world", 1000), ("lsc_equivalent", 400)):
        scene = scenes.REFERENCE_SCENES[name]()
        py, _ = python_counts(scene, n_python, seed=11)
        result = engine.simulate(scene, 20000, seed=7, max_events=256)
        assert result.data["counts"].max() < 255
        assert_means_close(py, table_counts(result.data, 256))


# -- the reference's seeded known answers for its Python tracer ------------------------------
# (tests/test_refractored_tracer.py:116-377: numpy seed 0, ray from (0,0,-1) along +z; the
# expected positions/events below are that file's constants, so they pin oracle/py_tracer.py --
# draw order included -- to the reference's photon_tracer.follow)

def _embedded(n1=1.5, components=None):
    from pvtrace_amd import Box, Material, Node, Scene, Sphere

    world = Node(name="world (air)", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    box = Node(name="box", parent=world,
               geometry=Box((1.0, 1.0, 1.0), material=Material(refractive_index=n1, components=components)))
    return Scene(world), world, box


def _touching():
    from pvtrace_amd import Box, Material, Node, Scene, Sphere

    world = Node(name="world (air)", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    boxes = []
    for k in range(3):
        b = Node(name=f"box {k + 1}", parent=world, geometry=Box((1.0, 1.0, 1.0), material=Material(refractive_index=1.5)))
        b.translate((0.0, 0.0, float(k)))
        boxes.append(b)
    return Scene(world), world, boxes


def _follow_seed0(scene):
    from pvtrace_amd.light import Ray

    np.random.seed(0)
    path = P.follow(scene, Ray(position=(0.0, 0.0, -1.0), direction=(0.0, 0.0, 1.0), wavelength=555.0))
    return [r.position for r, _ in path], [e for _, e in path]


@pytest.mark.parametrize("case", ["glass", "mirror-like", "absorber", "reactor", "touching"])
def test_reference_known_answers_of_the_python_tracer(case):
    from pvtrace_amd import Absorber, Reactor

    if case == "glass":            # :116-139
        scene = _embedded()[0]
        want_p = [(0, 0, -1.0), (0, 0, -0.5), (0, 0, 0.5), (0, 0, 10.0)]
        want_e = [Event.GENERATE, Event.TRANSMIT, Event.TRANSMIT, Event.EXIT]
    elif case == "mirror-like":    # :142-165, n = 100: the first draw (0.5488) is below R = 0.96
        scene = _embedded(n1=100.0)[0]
        want_p = [(0, 0, -1.0), (0, 0, -0.5), (0, 0, -10.0)]
        want_e = [Event.GENERATE, Event.REFLECT, Event.EXIT]
    elif case == "absorber":       # :168-195, depth = -ln(1 - 0.7151893...) / 10
        scene = _embedded(components=[Absorber(coefficient=10.0)])[0]
        want_p = [(0, 0, -1.0), (0, 0, -0.5), (0, 0, -0.3744069237034118)]
        want_e = [Event.GENERATE, Event.TRANSMIT, Event.ABSORB]
    elif case == "reactor":        # :198-224
        scene = _embedded(components=[Reactor(coefficient=10.0)])[0]
        want_p = [(0, 0, -1.0), (0, 0, -0.5), (0, 0, -0.3744069237034118), (0, 0, -0.3744069237034118)]
        want_e = [Event.GENERATE, Event.TRANSMIT, Event.ABSORB, Event.REACT]
    else:                          # :262-293, three touching glass cubes
        scene = _touching()[0]
        want_p = [(0, 0, -1.0), (0, 0, -0.5), (0, 0, 0.5), (0, 0, 1.5), (0, 0, 2.5), (0, 0, 10.0)]
        want_e = [Event.GENERATE] + [Event.TRANSMIT] * 4 + [Event.EXIT]
    positions, events = _follow_seed0(scene)
    assert events[:len(want_e)] == want_e
    for got, want in zip(positions, want_p):
        assert np.allclose(got, want, atol=2.3e-13), (got, want)


def test_find_container_known_answers():
    """tests/test_refractored_tracer.py:247-260, :296-377."""
    scene, world, boxes = _touching()
    assert tuple(x.hit for x in scene.intersections((0, 0, -1.0), (0, 0, 1.0))) == (
        boxes[0], boxes[0], boxes[1], boxes[1], boxes[2], boxes[2], world)
    for z, want in ((-1.0, world), (-0.4, boxes[0]), (0.6, boxes[1]), (1.6, boxes[2]), (2.6, world)):
        assert P.find_container(scene.intersections((0.0, 0.0, z), (0.0, 0.0, 1.0))) is want
    scene, world, box = _embedded()
    for z, want in ((-1.0, world), (-0.4, box), (0.6, world)):
        assert P.find_container(scene.intersections((0.0, 0.0, z), (0.0, 0.0, 1.0))) is want


def test_py_tracer_reproduces_the_references_python_tracer_history_by_history():
    """tests/golden/py_tracer.npz: 300 rays through the REFERENCE's `photon_tracer.follow`
    (pvtrace/algorithm/photon_tracer.py:26-328) -- on its own Sphere / Cylinder, Material, Luminophore / Absorber /
    Scatterer, Fresnel surface delegate, Distribution and Ray, hung on the product's Node / Scene (the reference's scene
    graph needs anytree, which is absent; no stand-in for it is written) -- each under its own numpy seed.  The
    restatement must produce the SAME history from the same seed: every event, in order, with its position, direction and
    wavelength (the order of the random draws is part of what is pinned).  SURVEY §8(a), last row."""
    from tests.util import load_golden
    from pvtrace_amd.light import Ray

    g = load_golden("py_tracer.npz")
    dirs, wls, seeds = scenes.py_tracer_pin_rays()
    assert np.array_equal(dirs, g["directions"]) and np.array_equal(wls, g["wavelengths"]) and np.array_equal(seeds, g["seeds"])
    scene = scenes.py_tracer_pin_scene()
    at = 0
    seen = set()
    for d, w, sd, n in zip(dirs, wls, seeds, g["counts"]):
        np.random.seed(int(sd))
        hist = P.follow(scene, Ray(position=(0.0, 0.0, 0.0), direction=tuple(d), wavelength=float(w)))
        assert len(hist) == n, (sd, len(hist), n)
        for k, item in enumerate(hist):
            ray, event = item[0], item[1]
            assert event.value == g["kind"][at + k], (sd, k)
            assert np.allclose(ray.position, g["position"][at + k], rtol=0, atol=1e-9), (sd, k)
            assert np.allclose(ray.direction, g["direction"][at + k], rtol=0, atol=1e-9), (sd, k)
            assert abs(ray.wavelength - g["wavelength"][at + k]) < 1e-9, (sd, k)
            seen.add(event)
        at += n
    assert at == len(g["kind"])
    assert {Event.REFLECT, Event.TRANSMIT, Event.ABSORB, Event.EMIT, Event.SCATTER, Event.NONRADIATIVE, Event.EXIT} <= seen


# -- the PRODUCT's per-ray path on the scene objects (pvtrace_amd/algorithm/photon_tracer.py, backend="host") ----------------
def test_config1_hello_world_through_the_products_own_per_ray_path_without_a_gpu():
    """BASELINE configs[0] as it is written: hello_world, 1 000 rays, the package's Python-level per-ray path, no GPU.
    `photon_tracer.follow` steps the ray through the scene objects (their per-interaction methods, numpy's generator) when
    no GPU is visible; `backend="host"` asks for that path outright, so the test means the same on the GPU box.  SURVEY §8(d):
    per-ray event means GENERATE 1, TRANSMIT ~1.91, REFLECT ~0.087, EXIT 1; and Welch 5 sigma against the C referee."""
    from pvtrace_amd import photon_tracer

    scene = scenes.hello_world()
    np.random.seed(1)
    table = np.zeros((1000, len(KINDS)))
    for j, ray in enumerate(scene.emit(1000)):
        history = photon_tracer.follow(scene, ray, backend="host")
        assert history[0][1] == Event.GENERATE and history[-1][1] == Event.EXIT
        assert np.isclose(np.linalg.norm(history[-1][0].position), 10.0)
        for _, event in history:
            table[j, KINDS.index(event)] += 1
    mean = dict(zip(KINDS, table.mean(axis=0)))
    assert mean[Event.GENERATE] == 1.0 and mean[Event.EXIT] == 1.0 and mean[Event.KILL] == 0.0
    assert mean[Event.TRANSMIT] == pytest.approx(1.91, abs=0.03)
    assert mean[Event.REFLECT] == pytest.approx(0.087, abs=0.03)
    ref, _ = referee_counts(scenes.hello_world(), 20000, seed=7, max_events=64)
    assert_means_close(table, ref)


def test_products_per_ray_path_reproduces_the_references_python_tracer_history_by_history():
    """The same 300 reference histories as above (tests/golden/py_tracer.npz), asked of the PRODUCT's host path: it calls
    the host classes' per-interaction methods, which keep the reference's draw order, so equal numpy seeds must give equal
    histories -- every event, position, direction and wavelength."""
    from pvtrace_amd import photon_tracer
    from pvtrace_amd.light import Ray
    from tests.util import load_golden

    g = load_golden("py_tracer.npz")
    dirs, wls, seeds = scenes.py_tracer_pin_rays()
    scene = scenes.py_tracer_pin_scene()
    at = 0
    for d, w, sd, n in zip(dirs, wls, seeds, g["counts"]):
        np.random.seed(int(sd))
        hist = photon_tracer.follow(scene, Ray(position=(0.0, 0.0, 0.0), direction=tuple(d), wavelength=float(w)), backend="host")
        assert len(hist) == n, (sd, len(hist), n)
        for k, (ray, event) in enumerate(hist):
            assert event.value == g["kind"][at + k], (sd, k)
            assert np.allclose(ray.position, g["position"][at + k], rtol=0, atol=1e-9), (sd, k)
            assert np.allclose(ray.direction, g["direction"][at + k], rtol=0, atol=1e-9), (sd, k)
            assert abs(ray.wavelength - g["wavelength"][at + k]) < 1e-9, (sd, k)
        at += n
    assert at == len(g["kind"])


def test_products_per_ray_path_honours_maxsteps_and_maxpathlength():
    """reference photon_tracer.py:162-172: KILL at the start of the first step entered beyond either limit."""
    from pvtrace_amd import photon_tracer

    scene = scenes.trapped_light()
    np.random.seed(3)
    ray = next(iter(scene.emit(1)))
    capped = photon_tracer.follow(scene, ray, maxsteps=5, backend="host")
    assert capped[-1][1] == Event.KILL and len(capped) <= 2 * 5 + 2
    np.random.seed(3)
    short = photon_tracer.follow(scene, ray, maxpathlength=0.5, backend="host")
    assert short[-1][1] == Event.KILL and short[-1][0].travelled > 0.5
    with pytest.raises(ValueError):
        photon_tracer.follow(scene, ray, backend="cpu")
