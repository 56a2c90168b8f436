"""Unit-level golden vectors made by tests/golden/make_golden.py from the REFERENCE's
own Python modules (spectra, Distribution, Fresnel helpers, Sphere/Cylinder,
Transformable), checked against (a) the host API of pvtrace_amd and (b) the oracle's
C functions — the same functions whose GPU twins the -m gpu tests compare bit-for-bit."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd import Box, Cylinder, Node, Sphere
from pvtrace_amd.data import fluro_red, lumogen_f_red_305
from pvtrace_amd.geometry import Transformable
from pvtrace_amd.material import (
    Distribution, fresnel_reflectivity, fresnel_refraction, specular_reflection,
)

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def test_spectra_bit_exact():
    g = load("spectra.npz")
    assert np.array_equal(lumogen_f_red_305.absorption(g["x"]), g["lumogen_abs"])
    assert np.array_equal(lumogen_f_red_305.emission(g["x"]), g["lumogen_ems"])
    assert np.array_equal(lumogen_f_red_305.absorption(g["xw"]), g["lumogen_abs_w"])
    assert np.array_equal(lumogen_f_red_305.emission(g["xw"]), g["lumogen_ems_w"])
    assert np.array_equal(fluro_red.absorption(g["xw"]), g["fluro_abs_w"])
    assert np.array_equal(fluro_red.emission(g["xw"]), g["fluro_ems_w"])


def test_distribution_cdf_sample_lookup():
    g = load("spectra.npz")
    d = Distribution(g["x"], g["lumogen_ems"])
    assert np.array_equal(d._cdf, g["lumogen_ems_cdf"])
    assert np.array_equal(d.sample(g["sample_p"]), g["sample_x"])
    assert np.array_equal(d.lookup(g["lookup_x"]), g["lookup_p"])
    assert np.array_equal(d(g["lookup_x"]), g["call_y"])
    # the kernel's clamped interpolation reproduces np.interp on the same tables
    xs = g["x"].astype(float)
    for p, want in zip(g["sample_p"], g["sample_x"]):
        assert O.interp(p, d._cdf, xs) == pytest.approx(want, rel=1e-15, abs=1e-12)
    for x, want in zip(g["lookup_x"], g["lookup_p"]):
        assert O.interp(x, xs, d._cdf) == pytest.approx(want, rel=1e-14, abs=1e-16)


def test_distribution_reference_endpoint_cases():
    """reference tests/test_distibution.py:9-21 (sample(0)/sample(1) hit the range ends)."""
    x = np.linspace(400, 1010, 2000)
    y = np.exp(-((x - 700.0) / 60.0) ** 2)
    d = Distribution(x, y)
    assert np.isclose(d.sample(0), x.min()) and np.isclose(d.sample(1), x.max())
    assert np.isclose(d.lookup(x.min()), 0.0) and np.isclose(d.lookup(x.max()), 1.0)
    with pytest.raises(ValueError):
        d.sample(1.5)
    with pytest.raises(ValueError):
        d.lookup(1200.0)


def test_fresnel_reflectivity_golden():
    g = load("optics.npz")
    for (n1, n2), row in zip(g["pairs"], g["reflectivity"]):
        mine = np.array([fresnel_reflectivity(a, n1, n2) for a in g["angles"]])
        assert np.allclose(mine, row, rtol=1e-13, atol=1e-16)
        for mode in (O.MATH_LIBM, O.MATH_PORTABLE):
            orc = np.array([O.fresnel_reflectivity(a, n1, n2, mode) for a in g["angles"]])
            assert np.allclose(orc, row, rtol=1e-12, atol=1e-15)
    # reference tests/test_frensel_reflection.py:8-10
    assert np.isclose(O.fresnel_reflectivity(0.0, 1.0, 1.5), 0.04)
    assert O.fresnel_reflectivity(1.0, 1.5, 1.0) == 1.0  # beyond the critical angle


def test_reflection_refraction_vectors_golden():
    g = load("optics.npz")
    for d, n, want in zip(g["d"], g["n"], g["specular"]):
        assert np.allclose(specular_reflection(d, n), want, rtol=0, atol=1e-15)
        assert np.allclose(O.specular_reflect(d, n), want, rtol=0, atol=1e-15)
    for d, nf, want in zip(g["d"], g["nf"], g["refract_up"]):
        assert np.allclose(fresnel_refraction(d, nf, 1.0, 1.5), want, rtol=0, atol=1e-15)
        assert np.allclose(O.fresnel_refract(d, nf, 1.0, 1.5), want, rtol=0, atol=1e-15)
    ok = g["down_mask"]
    for d, nf, want in zip(g["d"][ok], g["nf"][ok], g["refract_down"]):
        assert np.allclose(O.fresnel_refract(d, nf, 1.5, 1.0), want, rtol=0, atol=2e-15)
    # reference tests/test_frensel_refraction.py: normal incidence is not bent
    assert np.allclose(O.fresnel_refract((0, 0, -1.0), (0, 0, -1.0), 1.0, 1.5), (0, 0, -1.0))
    # reference tests/test_frensel_reflection.py: the normal's sign does not matter
    assert np.allclose(O.specular_reflect((0, 0, -1.0), (0, 0, 1.0)), (0, 0, 1.0))
    assert np.allclose(O.specular_reflect((0, 0, -1.0), (0, 0, -1.0)), (0, 0, 1.0))


@pytest.mark.parametrize("shape", ["sphere", "cyl"])
def test_intersections_and_normals_golden(shape):
    g = load("geometry.npz")
    if shape == "sphere":
        geom, gtype, prm = Sphere(float(g["sphere_radius"])), 1, [float(g["sphere_radius"])]
    else:
        geom = Cylinder(float(g["cyl_length"]), float(g["cyl_radius"]))
        gtype, prm = 2, [float(g["cyl_length"]), float(g["cyl_radius"])]
    counts, points = g[f"{shape}_count"], g[f"{shape}_points"]
    k = 0
    for i, (o, d) in enumerate(zip(g["origin"], g["direction"])):
        ts = np.sort(O.intersect(gtype, prm, o, d))
        assert len(ts) == counts[i], i
        host = geom.intersections(o, d)
        assert len(host) == counts[i]
        for j, t in enumerate(ts):
            assert np.allclose(o + t * d, points[i, j], rtol=0, atol=1e-9)
            assert np.allclose(host[j], points[i, j], rtol=0, atol=1e-9)
        if counts[i] > 0:
            want = g[f"{shape}_normals"][k]
            k += 1
            assert np.allclose(O.normal(gtype, prm, points[i, 0]), want, atol=1e-9)
            assert np.allclose(geom.normal(points[i, 0]), want, atol=1e-9)


def test_reference_geometry_known_answers():
    """Numbers from the reference's tests/test_sphere.py, test_box.py, test_geometry_utils.py."""
    assert Sphere(1).intersections((-2.0, 0.0, 0.0), (1.0, 0.0, 0.0)) == ((-1.0, 0.0, 0.0), (1.0, 0.0, 0.0))
    assert Box((1, 1, 1)).intersections((-2.0, 0.0, 0.0), (1.0, 0.0, 0.0)) == ((-0.5, 0.0, 0.0), (0.5, 0.0, 0.0))
    for p, n in [((0.5, 0, 0), (1, 0, 0)), ((0, 0.5, 0), (0, 1, 0)), ((0, 0, 0.5), (0, 0, 1)),
                 ((-0.5, 0, 0), (-1, 0, 0)), ((0, -0.5, 0), (0, -1, 0)), ((0, 0, -0.5), (0, 0, -1))]:
        assert np.allclose(Box((1, 1, 1)).normal(p), n)
        assert np.allclose(O.normal(0, [1, 1, 1], p), n)
    assert Sphere(1).contains((0, 0, 0)) and not Sphere(1).contains((0, 0, 1.0))
    assert Box((1, 1, 1)).contains((0, 0, 0)) and not Box((1, 1, 1)).contains((0, 0, 0.5))
    assert Box((1, 1, 1)).is_on_surface((0.5, 0, 0)) and not Box((1, 1, 1)).is_on_surface((0.501, 0, 0))
    # ray / z-cylinder cases (tests/test_geometry_utils.py:63-101), length 1 radius 1
    cyl = Cylinder(1.0, 1.0)
    unit = lambda v: np.asarray(v, float) / np.linalg.norm(v)
    got = cyl.intersections((0.2, 0.2, -1), unit((0, 0, 1.0)))
    assert np.allclose(got, ((0.2, 0.2, -0.5), (0.2, 0.2, 0.5)))
    got = cyl.intersections((-2, 0.2, 0.0), unit((1.0, 0.2, -0.2)))
    assert np.allclose(got, ((-0.9082895433880116, 0.41834209132239775, -0.2183420913223977), (0.5, 0.7, -0.5)))
    got = cyl.intersections((-2, 0.2, 0.0), unit((1.0, 0.2, 0.2)))
    assert np.allclose(got, ((-0.9082895433880116, 0.41834209132239775, 0.2183420913223977), (0.5, 0.7, 0.5)))
    ts = O.intersect(2, [1.0, 1.0], (-2, 0.2, 0.0), unit((1.0, 0.2, 0.2)))
    assert np.allclose(np.sort(ts), [np.linalg.norm(np.subtract(p, (-2, 0.2, 0.0))) for p in got])


def test_transformable_and_node_poses_golden():
    g = load("transforms.npz")
    a = Node(name="A"); a.translate((0, 0, 2)); a.rotate(np.pi * 0.2, (0, 1, 0))
    b = Node(name="B", parent=a); b.rotate(np.pi / 2, (1, 0, 0))
    world = Node(name="W"); a.parent = world
    assert np.array_equal(a.pose, g["A"]) and np.array_equal(b.pose, g["B"])
    assert np.allclose(b.transformation_to(world), g["B_in_world"], rtol=0, atol=1e-15)
    light = Transformable(); light.location = (0.0, 0.0, 5.0); light.rotate(np.radians(180), (1, 0, 0))
    assert np.array_equal(light.pose, g["light"])
    slab = Transformable(); slab.translate((0.5, -0.3, 1.0)); slab.rotate(0.35, (1.0, 0.4, 0.2))
    assert np.array_equal(slab.pose, g["slab"])
    loc = Transformable(location=(1.0, 2.0, 3.0)); loc.rotate(1.1, (0.0, 1.0, 0.3)); loc.translate((0.5, 0.5, -2.0))
    assert np.array_equal(loc.pose, g["located"])
    # round trips (reference tests/test_node.py:38-63 style)
    p = (0.3, -0.2, 0.9)
    assert np.allclose(world.point_to_node(b.point_to_node(p, world), b), p)
    assert np.allclose(np.linalg.norm(b.vector_to_node((0, 0, 1), world)), 1.0)


def test_histogram_sampled_distribution_golden():
    """hist=True spectra (extension over the reference ENGINE, which rejects them): the step-table
    rule of the reference's Python Distribution, pinned by vectors from that class."""
    g = load("spectra_hist.npz")
    d = Distribution(g["x"], g["y"], hist=True)
    assert np.array_equal(d._cdf, g["cdf"])
    assert np.array_equal(d(g["query_x"]), g["value"])
    assert np.array_equal(d.lookup(g["query_x"]), g["lookup"])
    assert np.array_equal(d.sample(g["query_p"]), g["sample"])
    for q, v, l in zip(g["query_x"], g["value"], g["lookup"]):
        assert O.step_lookup(q, g["x"], g["y"]) == v
        assert O.step_lookup(q, g["x"], g["cdf"]) == l
    for p, s in zip(g["query_p"], g["sample"]):
        assert O.step_lookup(p, g["cdf"], g["x"]) == s, p


# ---- round 5: more of the path pinned by the reference's own modules (no stand-ins) ----------------------
PHASE_CASES = ["isotropic", "hg", "hg_back", "cone", "lambertian"]


@pytest.mark.parametrize("case", PHASE_CASES)
def test_phase_functions_against_the_reference_functions(case):
    """tests/golden/phase.npz: the reference's material/utils.py phase functions, fed the first two draws of our per-ray
    stream.  The oracle's sample_phase (what the trace loop calls) must give the same direction from the same stream --
    libm mode to rounding, portable mode (the GPU's arithmetic) to a few ulp -- and so must the product's own Python
    functions.  `lambertian` is the extension tag PVT_PHASE_LAMBERTIAN: no reference KERNEL counterpart, but this
    reference function."""
    import pvtrace_amd.material as M

    g = load("phase.npz")
    tag, param = int(g[f"{case}_tag"]), float(g[f"{case}_param"])
    mine = {"isotropic": M.isotropic, "hg": lambda: M.henyey_greenstein(param), "hg_back": lambda: M.henyey_greenstein(param),
            "cone": lambda: M.cone(param), "lambertian": M.lambertian}[case]
    real = np.random.uniform
    for seed, draws, want in zip(g["seeds"], g["draws"], g[f"{case}_dir"]):
        assert np.array_equal(O.uniforms(int(seed), 2), draws)
        assert np.allclose(O.phase(tag, param, int(seed), O.MATH_LIBM), want, rtol=0, atol=2e-15), (case, seed)
        assert np.allclose(O.phase(tag, param, int(seed), O.MATH_PORTABLE), want, rtol=0, atol=1e-14), (case, seed)
        feed = list(draws)

        def fake(low=0.0, high=1.0, size=None):
            return feed.pop(0) if size is None else np.array([feed.pop(0) for _ in range(int(size))])

        np.random.uniform = fake
        try:
            got = np.asarray(mine(), dtype=float)
        finally:
            np.random.uniform = real
        assert np.allclose(got, want, rtol=0, atol=2e-15), (case, seed)
    if case == "lambertian":
        assert np.all(g["lambertian_dir"][:, 2] >= 0.0)   # never below the surface (material/utils.py:180)


def test_lambertian_phase_is_the_cone_of_half_angle_pi_over_two():
    """What the host packer relies on when it lowers PVT_PHASE_LAMBERTIAN to the kernel's cone branch (pvt_trace.hip):
    sin(pi/2) is the double 1.0 in the kernel's arithmetic, so the two tags sample identical bits."""
    assert O.math("sin", np.array([1.5707963267948966]), math_mode=O.MATH_PORTABLE)[0] == 1.0
    for seed in range(1, 60):
        for mode in (O.MATH_PORTABLE, O.MATH_LIBM):
            assert np.array_equal(O.phase(3, 0.0, seed, mode), O.phase(2, 1.5707963267948966, seed, mode))


@pytest.mark.parametrize("shape", ["sphere", "cyl"])
def test_surface_branch_against_the_reference_fresnel_delegate(shape):
    """tests/golden/surface.npz: the reference's FresnelSurfaceDelegate (material/surface.py:102-177) on the reference's
    own Sphere / Cylinder.  The oracle's surface branch -- normal, flipped normal, clamped cosine, angle, reflectivity,
    specular and Snell directions, the functions trace_one runs in that order -- answers the same."""
    g = load("surface.npz")
    gtype, prm = int(g[f"{shape}_type"]), list(g[f"{shape}_params"])
    tir = 0
    for p, d, (n1, n2), nrm, r, refl, trans in zip(g[f"{shape}_points"], g[f"{shape}_directions"], g[f"{shape}_indices"],
                                                   g[f"{shape}_normals"], g[f"{shape}_reflectivity"],
                                                   g[f"{shape}_reflected"], g[f"{shape}_transmitted"]):
        for mode in (O.MATH_LIBM, O.MATH_PORTABLE):
            o_n, o_r, o_refl, o_trans = O.surface(gtype, prm, p, d, n1, n2, mode)
            assert np.allclose(o_n, nrm, rtol=0, atol=1e-9)       # (the point is on the surface to ~1e-15)
            assert o_r == pytest.approx(r, rel=1e-9, abs=1e-12)
            assert np.allclose(o_refl, refl, rtol=0, atol=1e-9)
            if r < 1.0:
                assert np.allclose(o_trans, trans, rtol=0, atol=1e-9)
        tir += r == 1.0
    assert 0 < tir < len(g[f"{shape}_points"])   # both sides of the critical angle are in the sample


def test_recorder_vocabulary_is_the_references():
    """tests/golden/recorder_ids.json: PROPERTIES / EVENTS of the reference's engine/recorder.py:33-55, the messages
    of what its constructors refuse, its defaults -- against the product's module and the ids of the C ABI."""
    import json
    import re

    from pvtrace_amd.engine import recorder as R

    with open(os.path.join(GOLD, "recorder_ids.json")) as fp:
        doc = json.load(fp)
    assert R.PROPERTIES == doc["PROPERTIES"] and R.EVENTS == doc["EVENTS"]
    header = open(os.path.join(os.path.dirname(GOLD), "..", "include", "pvtrace_hip.h")).read()
    abi = {k.lower(): int(v) for k, v in re.findall(r"PVT_REC_(\w+) = (\d+)", header)}
    assert abi == doc["EVENTS"]
    refused = {
        "histogram_unknown_property": lambda: R.Histogram("colour", 0, 1, 4),
        "histogram_empty_range": lambda: R.Histogram("x", 1, 1, 4),
        "histogram_no_bins": lambda: R.Histogram("x", 0, 1, 0),
        "recorder_unknown_event": lambda: R.Recorder("r", event="vanished"),
        "recorder_bad_histogram": lambda: R.Recorder("r", histograms=[3]),
    }
    for key, fn in refused.items():
        with pytest.raises(ValueError) as err:
            fn()
        assert str(err.value) == doc["refused"][key], key
    r = R.Recorder("r")
    assert (r.event, r.atol, r.facet) == (doc["defaults"]["event"], doc["defaults"]["atol"], doc["defaults"]["facet"])


def test_host_objects_make_the_references_decisions_under_the_same_numpy_seeds():
    """tests/golden/object_methods.npz: the reference's `Material.penetration_depth / is_absorbed / component`
    (material/material.py:22-63), its components' `is_radiative / nonradiative_absorb / emit` (component.py:168-196, :236-239,
    :381-440: HG and cone phase functions, kT / redshift / full re-emission, radiative and non-radiative lifetimes) and its
    `Surface.is_reflected / reflect / transmit` (surface.py:224-272) on a glass ball, called in a fixed order under numpy
    seeds.  The product's classes run the same script (tests/scenes.py::object_method_script): same draws in the same order,
    so the same numbers, bit for bit -- what code that steps rays itself through these objects relies on."""
    import types

    import pvtrace_amd as P
    from pvtrace_amd.data import lumogen_f_red_305
    from tests import scenes
    from tests.util import load_golden

    g = load_golden("object_methods.npz")
    c = types.SimpleNamespace(Material=P.Material, Absorber=P.Absorber, Scatterer=P.Scatterer, Reactor=P.Reactor,
                              Luminophore=P.Luminophore, Surface=P.Surface, NullSurfaceDelegate=P.NullSurfaceDelegate,
                              Sphere=P.Sphere, Ray=P.Ray, henyey_greenstein=P.henyey_greenstein, cone=P.cone,
                              lumogen=lumogen_f_red_305)
    mine = scenes.object_method_script(c)
    assert set(mine) == set(g.files)
    for key in g.files:
        assert mine[key].shape == g[key].shape and np.array_equal(mine[key], g[key]), key
    # the fixture has teeth: both outcomes of every decision occur
    assert 0 < g["surface_reflected"].mean() < 1 and 0 < g["dye_radiative"].mean() < 1 and g["host_radiative"].sum() == 0
    assert len(np.unique(g["component"])) == 4 and np.isinf(g["clear_depth"][0])
    assert (g["host_ended_duration"] > 0).all() and np.array_equal(g["react_ended_duration"], g["react_ended_duration"])
