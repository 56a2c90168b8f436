"""Host side: scene graph, flattener, recorders, emitters, LSC builder, and the
pure-Python tally — the parts of the reference API the hot path sits behind
(constructor surface of SURVEY.md §8(b))."""
import functools
import os

import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd import (
    LSC, Absorber, Box, Coating, CoatedSurfaceDelegate, Cylinder, Light, Luminophore, Material,
    Node, Reactor, Scatterer, Scene, Sphere, Surface, SurfaceDelegate, cone, engine, isotropic,
)
from pvtrace_amd.engine import (
    Heatmap, Histogram, Recorder, UnsupportedSceneError, compile_scene, tally_histories,
)
from pvtrace_amd.engine import compiler as C
from pvtrace_amd.engine.api import EngineResult
from pvtrace_amd.engine.emit import EmitterTables, emit_bundle, sources_for
from pvtrace_amd.light import Event, SpectrumWavelengthMask
from tests import scenes


def test_node_tree_and_preorder():
    w = Node(name="w"); a = Node(name="a", parent=w); b = Node(name="b", parent=w)
    c = Node(name="c", parent=a)
    assert [n.name for n in w.preorder()] == ["w", "a", "c", "b"]
    assert [n.name for n in w.levelorder()] == ["w", "a", "b", "c"]
    assert c.root is w and c.path == (w, a, c) and w.leaves == (c, b)
    c.parent = b
    assert [n.name for n in w.preorder()] == ["w", "a", "b", "c"]
    with pytest.raises(Exception):
        w.parent = c


def test_compile_headline_scene_tables():
    c = compile_scene(scenes.lsc_equivalent())
    assert c.node_names == ["World", "LSC"] and c.root_id == 0
    assert c.geom_type.tolist() == [C.GEOM_BOX, C.GEOM_BOX]
    assert c.geom_params[:, :3].tolist() == [[500.0, 500.0, 100.0], [5.0, 5.0, 1.0]]
    assert c.refractive_index.tolist() == [1.0, 1.5]
    assert c.comp_start.tolist() == [0, 0] and c.comp_count.tolist() == [0, 2]
    assert c.comp_type.tolist() == [C.COMP_LUMINOPHORE, C.COMP_ABSORBER]
    assert c.comp_abs_n.tolist() == [400, 1] and c.comp_ems_n.tolist() == [400, 0]
    assert c.abs_x.shape == (401,) and c.abs_x[-1] == 0.0 and c.abs_y[-1] == 0.1  # constant -> (0, c)
    assert c.ems_cdf[0] == 0.0 and c.ems_cdf[399] == 1.0
    assert c.component_names == ["Lumogen F Red 305", "Background"]
    assert len(c.recorder_names) == 10 and c.total_bins == 6 * 80
    assert np.array_equal(c.local_to_world[1], np.eye(4))
    assert c.rec_event.tolist() == [1] * 6 + [3, 0, 2, 5]
    for dtype_i in ("geom_type", "surface_type", "comp_type", "rec_node", "hist_offset"):
        assert getattr(c, dtype_i).dtype == np.int32


def test_compile_rejects_what_the_reference_rejects():
    class Custom(SurfaceDelegate):
        def reflectivity(self, *a): return 0.5
        def reflected_direction(self, *a): return (0, 0, 1)
        def transmitted_direction(self, *a): return (0, 0, 1)

    def scene_with(material=None, geometry=None, recorders=None, light=True):
        w = Node(name="w", geometry=Sphere(10.0, material=Material(1.0)))
        g = geometry if geometry is not None else Box((1, 1, 1), material=material)
        Node(name="n", parent=w, geometry=g, recorders=recorders)
        if light:
            Node(name="l", parent=w, light=Light())
        return Scene(w)

    with pytest.raises(UnsupportedSceneError):   # custom Python surface delegate
        compile_scene(scene_with(Material(1.5, surface=Surface(delegate=Custom()))))
    with pytest.raises(UnsupportedSceneError):   # custom phase function
        compile_scene(scene_with(Material(1.5, components=[Scatterer(1.0, phase_function=lambda: (0, 0, 1))])))
    # histogram-sampled spectra: the reference compiler raises (compiler.py:313-317); here they are
    # lowered with a per-component flag (extension)
    x = np.linspace(400, 800, 10)
    ch = compile_scene(scene_with(Material(1.5, components=[Absorber(np.column_stack((x, x * 0 + 1)), hist=True)])))
    assert ch.comp_abs_hist.tolist() == [1] and ch.comp_ems_hist.tolist() == [0]
    with pytest.raises(UnsupportedSceneError):   # geometry without material
        compile_scene(scene_with(None))
    with pytest.raises(UnsupportedSceneError):   # facet on a volume event
        compile_scene(scene_with(Material(1.5), recorders=[Recorder("x", event="lost", facet=(0, 0, 1))]))
    with pytest.raises(UnsupportedSceneError):   # duplicate names
        compile_scene(scene_with(Material(1.5), recorders=[Recorder("x"), Recorder("x")]))
    with pytest.raises(UnsupportedSceneError):   # non-rigid pose
        s = scene_with(Material(1.5)); n = s.root.children[0]
        pose = np.eye(4); pose[0, 0] = 2.0; n.pose = pose
        compile_scene(s)
    with pytest.raises(UnsupportedSceneError):   # root without geometry
        compile_scene(Scene(Node(name="empty")))

    class Blob:  # unsupported geometry type (the reference rejects meshes, tests/test_engine.py:418-436)
        material = Material(1.5)
    with pytest.raises(UnsupportedSceneError):
        compile_scene(scene_with(geometry=Blob()))
    with pytest.raises(ValueError):
        Recorder("r", event="nonsense")
    with pytest.raises(ValueError):
        Histogram("wavelength", 5, 5, 10)
    with pytest.raises(ValueError):
        Histogram("colour", 0, 1, 10)


def test_simulate_validates_before_touching_the_gpu():
    with pytest.raises(ValueError):
        engine.simulate(scenes.fresnel_box(), 10, emit_method="planck")
    class Custom(SurfaceDelegate):
        def reflectivity(self, *a): return 0.5
        def reflected_direction(self, *a): return (0, 0, 1)
        def transmitted_direction(self, *a): return (0, 0, 1)
    w = Node(name="w", geometry=Sphere(10.0, material=Material(1.0)))
    Node(name="n", parent=w, geometry=Box((1, 1, 1), material=Material(1.5, surface=Surface(delegate=Custom()))))
    Node(name="l", parent=w, light=Light())
    with pytest.raises(UnsupportedSceneError):
        engine.simulate(Scene(w), 10)


def test_phase_functions_lowered_in_every_spelling():
    from pvtrace_amd.material import Cone, HenyeyGreenstein, henyey_greenstein
    for phase, want in [(None, (0, 0.0)), (isotropic, (0, 0.0)), (HenyeyGreenstein(0.3), (1, 0.3)),
                        (Cone(0.2), (2, 0.2)), (functools.partial(cone, 0.25), (2, 0.25)),
                        (functools.partial(henyey_greenstein, -0.4), (1, -0.4))]:
        w = Node(name="w", geometry=Sphere(5.0, material=Material(1.0, components=[Scatterer(0.5, phase_function=phase)])))
        Node(name="l", parent=w, light=Light())
        c = compile_scene(Scene(w))
        assert (int(c.comp_phase_type[0]), float(c.comp_phase_param[0])) == want


def test_emit_bundle_matches_reference_distributions():
    scene = scenes.lsc_equivalent()
    pos, dirs, wl, src = emit_bundle(scene, 20000, seed=4)
    assert pos.shape == (20000, 3) and np.all(pos == (0.0, 0.0, 5.0)) and np.all(wl == 555.0)
    assert set(src) == {"Light"}
    assert np.allclose(np.linalg.norm(dirs, axis=1), 1.0)
    # flipped about x: the cone points down; polar angle <= 20 degrees
    cosang = -dirs[:, 2]
    assert cosang.min() >= np.cos(np.radians(20)) - 1e-12
    # pdf ~ cos(theta) sin(theta): E[sin^2 theta] = sin^2(theta_max)/2
    assert abs(np.mean(1 - cosang ** 2) - np.sin(np.radians(20)) ** 2 / 2) < 2e-3
    # round-robin over two lights, level order
    ks = scenes.kitchen_sink()
    _, _, wl2, src2 = emit_bundle(ks, 10, seed=1)
    assert src2 == ["lamp", "glow"] * 5 and sources_for(ks, 10) == src2
    assert np.all(wl2[1::2] == 480.0)
    # global numpy state like the reference when seed is None
    np.random.seed(3); a = emit_bundle(ks, 50)[0]
    np.random.seed(3); b = emit_bundle(ks, 50)[0]
    assert np.array_equal(a, b)


def test_large_bundles_emitted_on_worker_threads_equal_the_single_threaded_arrays(monkeypatch):
    """Bundles of 4e5 rays and more evaluate their trigonometry in row chunks on a few threads; the draws stay
    sequential, so the arrays must not depend on the split."""
    from pvtrace_amd.engine import emit as E
    from tests import scenes as S

    for scene in (S.lsc_equivalent(), S.hello_world(), S.lambertian_sheet()):
        threaded = E.emit_bundle(scene, 450_000, seed=11)
        with monkeypatch.context() as m:
            m.setattr(E, "_chunked", lambda fn, n, min_rows=0: fn(0, n))
            single = E.emit_bundle(scene, 450_000, seed=11)
        for a, b in zip(threaded[:3], single[:3]):
            assert np.array_equal(a, b)


def test_emit_falls_back_for_custom_delegates_and_device_mode_refuses():
    w = Node(name="w", geometry=Sphere(5.0, material=Material(1.0)))
    Node(name="l", parent=w, light=Light(direction=lambda: (0.0, 1.0, 0.0), name="odd"))
    scene = Scene(w)
    pos, dirs, wl, src = emit_bundle(scene, 7, seed=1)
    assert np.all(dirs == (0.0, 1.0, 0.0)) and src == ["odd"] * 7
    with pytest.raises(UnsupportedSceneError):
        EmitterTables(scene, strict=True)


def test_oracle_emitter_statistics_and_reproducibility():
    ks = scenes.kitchen_sink()
    tab = EmitterTables(ks)
    p1, d1, w1 = O.emit(tab, 4000, emit_seed=9)
    p2, d2, w2 = O.emit(tab, 1000, emit_seed=9, ray_offset=3000)
    assert np.array_equal(p1[3000:], p2) and np.array_equal(d1[3000:], d2) and np.array_equal(w1[3000:], w2)
    assert np.allclose(np.linalg.norm(d1, axis=1), 1.0)
    lamp = slice(0, None, 2)
    assert np.all(np.hypot(p1[lamp, 0], p1[lamp, 1]) <= 1.5 + 1e-12) and np.all(p1[lamp, 2] == 8.0)
    assert 350 <= w1[lamp].min() and w1[lamp].max() <= 900 and abs(w1[lamp].mean() - 500) < 5
    assert np.all(w1[1::2] == 480.0)


def test_lsc_builder_lowers_its_delegates_to_coatings():
    lsc = LSC((5.0, 5.0, 1.0))
    c = compile_scene(lsc.scene)
    assert c.node_names == ["World", "LSC"] and c.n_coatings == 0      # plain Fresnel
    ref = compile_scene(scenes.lsc_equivalent(recorders=False))
    for key in ("geom_params", "local_to_world", "refractive_index", "abs_x", "abs_y", "ems_x", "ems_cdf",
                "comp_type", "comp_qy"):
        assert np.array_equal(getattr(c, key), getattr(ref, key)), key
    lsc2 = LSC((5.0, 5.0, 1.0))
    lsc2.add_solar_cell({"left", "right"}); lsc2.add_back_surface_mirror(); lsc2.add_air_gap_mirror(lambertian=True)
    c2 = compile_scene(lsc2.scene)
    assert c2.node_names == ["World", "LSC", "Air Gap Mirror"]
    assert c2.coat_count.tolist() == [0, 3, 6]
    assert c2.coat_reflectivity[:3].tolist() == [1.0, 0.0, 0.0] and c2.coat_transmit_mode[:3].tolist() == [0, 1, 1]
    assert set(c2.coat_reflect_mode[3:].tolist()) == {1} and np.all(c2.coat_reflectivity[3:] == 1.0)
    with pytest.raises(ValueError):
        lsc2.add_solar_cell({"top"})


def test_python_tally_reproduces_oracle_tallies_exactly():
    """reference tests/test_engine.py:286-318, with the oracle standing in for the engine."""
    scene = scenes.bench_slab(recorders=True)
    compiled = compile_scene(scene)
    pos, dirs, wl, src = emit_bundle(scene, 1500, seed=17)
    data = O.trace_bundle(compiled, pos, dirs, wl, 17, 1000, 256, 0, 1, 1)
    result = EngineResult(compiled, data, src, 256, 1, 0.0)
    python_side = tally_histories(scene, result.histories())
    for name, rec in result.recorders.items():
        assert python_side[name].rays == rec.rays and python_side[name].crossings == rec.crossings, name
        for i in range(len(rec.spec.histograms)):
            assert np.array_equal(python_side[name]._bins[i], rec._bins[i]), (name, i)
        assert np.allclose(python_side[name]._moments, rec._moments, rtol=1e-9), name
    assert result.recorders["entering"].mean("wavelength") == pytest.approx(555.0)
    counts = result.event_counts()
    assert counts[Event.GENERATE] == 1500 and counts[Event.EXIT] + counts[Event.NONRADIATIVE] + counts[Event.KILL] == 1500
    edges, values = result.recorders["entering"].histogram(0)
    assert edges.size == 51 and values.sum() == result.recorders["entering"].rays
    xe, ye, heat = result.recorders["top"].histogram(0)
    assert heat.shape == (20, 20) and heat.sum() == result.recorders["top"].rays


def test_coating_semantics_on_the_oracle():
    """Extension (no reference-engine counterpart): mirror quadrant reflects with p=1,
    everything else is Fresnel; index-matched cell faces transmit undeviated."""
    scene = scenes.coated_slab(scatter=0.0)
    compiled = compile_scene(scene)
    n = 4000
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=8)
    data = O.trace_bundle(compiled, pos, dirs, wl, 5, 1000, 32, 0, 1, 1)
    first = np.arange(n) * 32 + 1          # the event after GENERATE: hits the slab's top face
    on_mirror = (pos[:, 0] > 0) & (pos[:, 1] > 0)
    assert np.all(data["kind"][first][on_mirror] == 1)                    # REFLECT, always
    frac_reflected = np.mean(data["kind"][first][~on_mirror] == 1)
    assert abs(frac_reflected - 0.04) < 0.012                             # Fresnel at normal incidence
    lsc = LSC((5.0, 5.0, 1.0)); lsc.add_solar_cell({"left", "right", "near", "far"})
    c = compile_scene(lsc.scene)
    p, d, w, _ = emit_bundle(lsc.scene, 3000, seed=2)
    out = O.trace_bundle(c, p, d, w, 3, 1000, 64, 0, 1, 1)
    rows = (out["kind"] == 2) & (out["hit"] == 1) & (np.abs(out["normal"][:, 2]) < 0.5) & (out["container"] == 1)
    idx = np.flatnonzero(rows)
    assert idx.size > 100
    assert np.array_equal(out["direction"][idx], out["direction"][idx - 1])   # straight through the cell faces


def test_source_filter_on_the_oracle_and_python_tally():
    """Recorder(source=...) (extension): lights / components / a named component."""
    from pvtrace_amd.engine.recorder import SOURCE_COMPONENT, SOURCE_COMPONENTS, SOURCE_LIGHTS

    scene = scenes.bench_slab(recorders=False)
    slab = scene.root.children[0]
    slab.recorders = [Recorder("any", event="escaping"), Recorder("solar", event="escaping", source="lights"),
                      Recorder("lum", event="escaping", source="components"),
                      Recorder("dye", event="escaping", source="dye"),
                      Recorder("bg", event="escaping", source="background")]
    compiled = compile_scene(scene)
    assert compiled.rec_source_mode.tolist() == [0, SOURCE_LIGHTS, SOURCE_COMPONENTS, SOURCE_COMPONENT, SOURCE_COMPONENT]
    assert compiled.rec_source_id.tolist() == [-1, -1, -1, 0, 1]
    pos, dirs, wl, src = emit_bundle(scene, 3000, seed=2)
    data = O.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 256, 0, 1, 1)
    result = EngineResult(compiled, data, src, 256, 1, 0.0)
    recs = result.recorders
    assert recs["any"].crossings == recs["solar"].crossings + recs["lum"].crossings
    assert recs["lum"].rays == recs["dye"].rays > 0 and recs["bg"].rays == 0 and recs["solar"].rays > 0
    python_side = tally_histories(scene, result.histories())
    for name, rec in recs.items():
        assert python_side[name].rays == rec.rays and python_side[name].crossings == rec.crossings, name
    slab.recorders = [Recorder("bad", event="escaping", source="nobody")]
    with pytest.raises(UnsupportedSceneError):
        compile_scene(scene)


def test_auto_recorders_match_the_reference_shorthand():
    """reference tests/test_engine.py:376-415 (`record: true`): 6 faces + volume loss for a box."""
    from pvtrace_amd.engine import auto_recorders, instrument

    scene = scenes.bench_slab(recorders=False)
    slab = scene.root.children[0]
    recs = {r.name: r for r in auto_recorders(slab)}
    assert len(recs) == 7 and recs["slab-top"].facet == (0.0, 0.0, 1.0) and "slab-lost" in recs
    assert len(recs["slab-top"].histograms) == 3
    assert recs["slab-top"].histograms[2].a.bins == 50 and recs["slab-east"].histograms[2].b.bins == 10
    ball = Node(name="ball", geometry=Sphere(1.0, material=Material(1.5)))
    assert {r.name for r in auto_recorders(ball)} == {"ball-lost", "ball-escaping"}
    instrument(slab, explicit=[Recorder("slab-top", event="escaping", facet=(0, 0, 1))])
    names = [r.name for r in slab.recorders]
    assert len(names) == 7 and len(set(names)) == 7
    assert len([r for r in slab.recorders if r.name == "slab-top"][0].histograms) == 0   # explicit wins
    compile_scene(scene)


def test_host_side_scene_intersections():
    """Scene.intersections / Node.intersections (reference scene.py:153-195, node.py:137-174):
    every forward crossing, nearest first, in the root frame -- nested_cylinders along +z crosses
    A, the protruding child B twice, A again and the world (examples/nested_cylinders.py)."""
    scene = scenes.nested_cylinders()
    found = scene.intersections((0.0, 0.0, -1.0), (0.0, 0.0, 1.0))
    assert [x.hit.name for x in found] == ["A", "B", "B", "A", "World"]
    assert np.allclose([x.distance for x in found], [2.149349, 2.6, 3.4, 3.850651, 11.0], atol=1e-6)
    assert all(x.coordsys is scene.root for x in found) and np.allclose(found[-1].point, (0, 0, 10))
    a = [n for n in scene.root.children if n.name == "A"][0]
    local = a.intersections((0.0, 0.0, -5.0), (0.0, 0.0, 1.0))     # a ray along A's own axis
    assert [x.hit.name for x in local][:2] == ["A", "A"] and local[0].coordsys is a
    assert local[0].to(scene.root) == local[0].to(scene.root)


def test_node_frames_known_answers_of_the_reference():
    """reference tests/test_node.py:47-74 (frame conversions up and down the tree), :108-125
    (intersections through a translated child), :141-152 (look_at)."""
    from pvtrace_amd.geometry import rotation_matrix

    a = Node(name="a"); b = Node(name="b", parent=a); c = Node(name="c", parent=b); d = Node(name="d", parent=a)
    b.translate((1, 1, 1)); c.translate((0, 1, 1)); d.translate((-1, -1, -1))
    theta = 0.5 * np.pi
    b.rotate(theta, (0, 0, 1)); c.rotate(theta, (1, 0, 0)); d.rotate(theta, (0, 1, 0))
    assert np.allclose(d.point_to_node((0, 0, 0), a), (-1, -1, -1))
    assert np.allclose(d.point_to_node((1, 1, 1), a), (0, 0, -2))
    assert np.allclose(d.vector_to_node((1, 0, 0), a), (0, 0, -1))
    assert np.allclose(d.vector_to_node((0, 1, 0), a), (0, 1, 0))
    assert np.allclose(d.vector_to_node((0, 0, 1), a), (1, 0, 0))
    assert np.allclose(c.point_to_node((0, 0, 0), d), (-3, 2, 1))
    assert np.allclose(c.point_to_node((1, 1, 1), d), (-4, 3, 2))
    assert np.allclose(c.vector_to_node((1, 0, 0), d), (0, 1, 0))
    assert np.allclose(c.vector_to_node((0, 1, 0), d), (-1, 0, 0))
    assert np.allclose(c.vector_to_node((0, 0, 1), d), (0, 0, 1))

    top = Node(name="A"); ball = Node(name="B", parent=top, geometry=Sphere(radius=1.0))
    ball.translate((1.0, 0.0, 0.0))
    found = top.intersections((-2.0, 0.0, 0.0), (1.0, 0.0, 0.0))
    assert np.allclose([x.point for x in found], ((-1, 0, 0), (1, 0, 0)))          # in B's frame
    assert np.allclose([x.to(top).point for x in found], ((0, 0, 0), (2, 0, 0)))   # shifted 1 along x in A

    n = Node(name="n"); n.look_at([1, 0, 0])
    assert np.allclose(n.pose, rotation_matrix(np.pi / 2, [0, 1, 0]))
    n = Node(name="n"); n.look_at([0, 0, -1])
    assert np.allclose(n.pose, rotation_matrix(np.pi, [0, 1, 0]))


@pytest.mark.parametrize("seed", range(24))
def test_python_tally_reproduces_kernel_tallies_on_random_scenes(seed):
    """The host-side recorder semantics (engine/tally.py) against the kernel-side accumulators on
    random scenes (tests/fuzz.py): rays, crossings and every histogram bin -- facets, source
    filters, heatmaps in the recorder node's frame included."""
    from tests.fuzz import random_scene

    scene = random_scene(400 + seed, extensions=bool(seed % 2))
    compiled = compile_scene(scene)
    pos, dirs, wl, src = emit_bundle(scene, 200, seed=seed)
    data = O.trace_bundle(compiled, pos, dirs, wl, 5 + seed, 300, 64, 0, 1, 1)
    if data["counts"].max() >= 63 or np.isnan(data["direction"]).any() or not compiled.recorder_names:
        pytest.skip("truncated or NaN histories cannot be re-tallied")
    result = EngineResult(compiled, data, src, 64, 1, 0.0)
    python_side = tally_histories(scene, result.histories())
    for name, rec in result.recorders.items():
        assert python_side[name].rays == rec.rays and python_side[name].crossings == rec.crossings, name
        for i in range(len(rec.spec.histograms)):
            assert np.array_equal(python_side[name]._bins[i], rec._bins[i]), (name, i)


def test_histogram_sampled_light_spectrum_is_lowered_to_the_tables():
    """`SpectrumWavelengthMask(Distribution(..., hist=True))` used to drop the light to the per-ray Python path;
    it is a table now (PVT_WL_SPECTRUM_HIST): x[searchsorted(cdf, u)], the Distribution's own hist branch."""
    from pvtrace_amd.engine import emit as E
    from pvtrace_amd.material import Distribution

    x = np.array([400.0, 410.0, 425.0, 430.0, 455.0, 500.0, 520.0, 600.0])
    y = np.array([0.0, 0.0, 1.0, 3.0, 2.0, 0.0, 0.0, 4.0])
    dist = Distribution(x, y, hist=True)
    world = Node(name="w", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    Node(name="lamp", parent=world, light=Light(wavelength=SpectrumWavelengthMask(dist), name="lamp"))
    scene = Scene(world)
    tab = EmitterTables(scene, strict=True)          # no custom delegate any more
    assert tab.wl_type.tolist() == [E.WL_SPECTRUM_HIST] and tab.wl_spec_n.tolist() == [8]
    # host sampler: the same uniforms through Distribution.sample
    rng = np.random.default_rng(7)
    u = rng.random(5000)
    pos, dirs, wl, _ = emit_bundle(scene, 5000, seed=7)
    assert np.array_equal(wl, np.asarray(dist.sample(u)))
    assert set(np.unique(wl)) <= set(x.tolist()) and 400.0 not in wl and 600.0 in wl
    # the referee's device-style emitter draws from the same table
    p2, d2, w2 = O.emit(tab, 20000, emit_seed=3)
    assert set(np.unique(w2)) <= set(x.tolist())
    want = np.diff(np.concatenate(([0.0], dist._cdf)))
    got = np.array([(w2 == v).mean() for v in x])
    assert np.abs(got - want).max() < 0.012


def test_only_the_unrecognised_delegate_is_called_per_ray():
    calls = {"wl": 0, "pos": 0}

    def my_wavelength():
        calls["wl"] += 1
        return 500.0 + (calls["wl"] % 7)

    def my_position():
        calls["pos"] += 1
        return (0.25, -0.5, 0.0)

    world = Node(name="w", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    a = Node(name="a", parent=world, light=Light(wavelength=my_wavelength, direction=functools.partial(cone, 0.3), name="a"))
    a.location = (0.0, 0.0, 1.0)
    b = Node(name="b", parent=world, light=Light(position=my_position, name="b"))
    b.rotate(np.pi, (1, 0, 0))
    scene = Scene(world)
    with pytest.raises(UnsupportedSceneError):
        EmitterTables(scene, strict=True)
    n = 4001
    pos, dirs, wl, src = emit_bundle(scene, n, seed=5)
    assert calls == {"wl": 2001, "pos": 2000}                       # one call per ray of THAT light, nothing else
    assert list(src[:4]) == ["a", "b", "a", "b"]
    assert np.array_equal(wl[0::2], 500.0 + (np.arange(1, 2002) % 7)) and np.all(wl[1::2] == 555.0)
    assert np.all(pos[0::2] == (0.0, 0.0, 1.0)) and np.allclose(pos[1::2], (0.25, 0.5, 0.0))   # b is flipped about x
    cos_t = dirs[0::2, 2]
    assert cos_t.min() >= np.cos(0.3) - 1e-12 and np.allclose(np.linalg.norm(dirs, axis=1), 1.0)
    assert np.allclose(dirs[1::2], (0.0, 0.0, -1.0))


def test_bundles_of_a_group_sampled_side_by_side_equal_one_after_the_other():
    from pvtrace_amd.engine.emit import emit_bundles
    from tests import scenes

    scene = scenes.lsc_equivalent()
    counts, seeds = [3000, 3000, 3000, 1234], [11, 3011, 6011, 9011]
    together = emit_bundles(scene, counts, seeds)
    for (count, seed), got in zip(zip(counts, seeds), together):
        want = emit_bundle(scene, count, seed=seed)
        assert all(np.array_equal(got[k], want[k]) for k in range(3)) and list(got[3][:5]) == list(want[3][:5])


def test_chained_sources_behave_like_the_concatenated_list():
    from pvtrace_amd.engine.emit import ChainedSources, RoundRobinSources

    a, b, c = RoundRobinSources(["x", "y", "z"], 7), ["q", "r"], RoundRobinSources(["x", "y", "z"], 5, offset=2)
    chained, flat = ChainedSources([a, b, [], c]), list(a) + b + list(c)
    assert len(chained) == 14 and list(chained) == flat and chained == flat and chained.tolist() == flat
    assert chained[-1] == flat[-1] and chained[7] == "q"
    with pytest.raises(IndexError):
        chained[14]
    for lo, hi in ((2, 6), (5, 12), (7, 9), (0, 14), (3, 3), (9, 14)):
        assert list(chained[lo:hi]) == flat[lo:hi]
    assert chained[::3] == flat[::3]
    assert isinstance(chained[2:6], RoundRobinSources)      # a slice inside one part is that part's own slice


def test_lazy_log_columns_behave_like_the_dict_they_stand_for():
    import pickle

    from pvtrace_amd.engine.api import LazyLogColumns

    built = []

    def column(name):
        def build():
            built.append(name)
            return np.arange(3) + len(name)
        return build

    def fresh():
        return LazyLogColumns({"counts": np.array([1, 2])}, {"kind": column("kind"), "hit": column("hit")},
                              np.array([0, 1, 3]), {"kind": np.array([1, 2, 3]), "hit": np.array([0, 0, 1])})

    d = fresh()
    assert list(d) == ["counts", "kind", "hit"] and "kind" in d and len(d) == 3 and built == []
    assert d["kind"].tolist() == [4, 5, 6] and built == ["kind"]
    d["kind"]
    assert built == ["kind"]                                   # built once
    plain = dict(d)
    assert type(plain) is dict and built == ["kind", "hit"] and plain["hit"].tolist() == [3, 4, 5]
    assert {**fresh()}["kind"] is not None and fresh().get("hit").tolist() == [3, 4, 5] and fresh().get("nope", 5) == 5
    assert all(v is not None for v in fresh().values()) and dict(fresh().items())["hit"].tolist() == [3, 4, 5]
    thawed = pickle.loads(pickle.dumps(fresh()))
    assert type(thawed) is dict and thawed["kind"].tolist() == [4, 5, 6]
    e = fresh()
    e["kind"] = 7
    assert e["kind"] == 7 and e.pop("hit").tolist() == [3, 4, 5] and "hit" not in e


def test_a_vectorised_user_delegate_is_called_once_per_bundle():
    """The reference calls a delegate it does not recognise once per ray (pvtrace/engine/emit.py:116-124: 22 k rays/s).
    A user's delegate that offers the whole bundle -- `sample(n)`, or `vectorized = True` -- is asked once."""
    import time

    from pvtrace_amd.engine.emit import vectorized_delegate

    calls = {"wl": 0, "pos": 0}
    gen = np.random.default_rng(3)

    class Lamp:   # wavelengths of a two-line lamp, by its own method
        def __call__(self):
            return float(self.sample(1)[0])

        def sample(self, n):
            calls["wl"] += 1
            return np.where(gen.random(n) < 0.25, 436.0, 546.0)

    def ring(n):   # positions on a ring of radius 0.5 in the light's xy plane
        calls["pos"] += 1
        phi = 2.0 * np.pi * gen.random(n)
        return np.column_stack((0.5 * np.cos(phi), 0.5 * np.sin(phi), np.zeros(n)))

    world = Node(name="w", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    a = Node(name="a", parent=world, light=Light(wavelength=Lamp(), position=vectorized_delegate(ring),
                                                 direction=functools.partial(cone, 0.2), name="a"))
    a.location = (0.0, 0.0, 2.0)
    scene = Scene(world)
    with pytest.raises(UnsupportedSceneError):
        EmitterTables(scene, strict=True)       # still not a device emitter: user code stays on the host
    n = 200_000
    tic = time.perf_counter()
    pos, dirs, wl, src = emit_bundle(scene, n, seed=5)
    elapsed = time.perf_counter() - tic
    assert calls == {"wl": 1, "pos": 1}
    assert set(np.unique(wl)) == {436.0, 546.0} and abs(np.mean(wl == 436.0) - 0.25) < 0.01
    assert np.allclose(np.hypot(pos[:, 0], pos[:, 1]), 0.5) and np.all(pos[:, 2] == 2.0)
    assert n / elapsed > 1e6, n / elapsed      # (per ray this is ~2e4/s in the reference and ~1e5/s here)
    # one sample at a time still works (Light.emit, the per-ray paths)
    one = next(iter(scene.emit(1)))
    assert one.wavelength in (436.0, 546.0) and abs(np.hypot(one.position[0], one.position[1]) - 0.5) < 1e-12
    # a delegate that promises a bundle and returns the wrong shape is refused, not broadcast
    bad = Node(name="b", parent=world, light=Light(wavelength=vectorized_delegate(lambda k: np.zeros((k, 2))), name="b"))
    with pytest.raises(ValueError, match="shape"):
        emit_bundle(Scene(world), 10, seed=1)


def test_host_emitter_is_the_references_bit_for_bit_under_a_numpy_seed():
    """tests/golden/emit.npz: the REFERENCE's `emit_bundle` (engine/emit.py:92-134) on five posed lights with every
    built-in wavelength / position / direction delegate, under numpy seeds.  With `seed=None` the product's host emitter
    draws from numpy's global generator like the reference -- same draws, same order, same arithmetic: identical arrays
    and the same round-robin of sources."""
    from tests import scenes
    from tests.util import load_golden

    g = load_golden("emit.npz")
    scene = scenes.emit_pin_scene()
    for n in (1, 7, 1003):
        np.random.seed(int(g[f"n{n}_seed"]))
        pos, dirs, wl, sources = emit_bundle(scene, n, seed=None)
        assert np.array_equal(pos, g[f"n{n}_position"]) and np.array_equal(dirs, g[f"n{n}_direction"])
        assert np.array_equal(wl, g[f"n{n}_wavelength"])
        assert list(sources) == g[f"n{n}_sources"].tolist()


def test_lsc_builder_makes_the_scene_the_references_lsc_class_makes():
    """tests/golden/lsc_scenes.npz: what the REFERENCE's `LSC` class builds (device/lsc.py:95-219) for the default
    device -- BASELINE configs[1] -- and for one configured through every public `add_*` method that works there
    (`add_scatterer` raises NameError in the reference), described node by node: names, tree, box sizes, refractive
    indices, poses, surface delegate classes, components with coefficient and emission arrays, lights with their
    delegates.  The product's builder must describe the same, array for array, bit for bit."""
    from pvtrace_amd import light as product_light, material as product_material
    from pvtrace_amd.data import lumogen_f_red_305
    from tests import scenes
    from tests.util import describe_lsc_scene, load_golden

    g = load_golden("lsc_scenes.npz")
    cases = scenes.lsc_builder_cases(LSC, product_material.cone, product_light.rectangular_mask, lumogen_f_red_305)
    for name, device in cases.items():
        mine = describe_lsc_scene(device.scene)
        theirs = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(name + "/")}
        assert set(mine) == set(theirs), (name, sorted(set(mine) ^ set(theirs)))
        for key, want in theirs.items():
            assert mine[key].shape == want.shape and np.array_equal(mine[key], want), (name, key)
    assert g["default/names"].tolist() == ["World", "LSC", "Light"]
    assert g["custom/names"].tolist() == ["World", "LSC", "Air Gap Mirror", "Lamp"]


def test_engine_result_answers_like_the_references_on_the_same_kernel_output():
    """tests/golden/engine_result.npz: the reference's whole host pipeline around its kernel -- its `compile_scene`, its
    `emit_bundle` under a numpy seed, its compiled `_kernel.trace_bundle`, its `EngineResult` / `RecorderResult`
    (engine/api.py:26-194) -- on the kitchen-sink scene (900 rays, every third with a history).  The product's result
    object, given the SAME kernel output and the product's tables of the product's twin scene, must answer the same:
    recorder rays / crossings / mean / std / error of the four properties, histogram edges and counts (1-D and 2-D), event
    counts, and every history -- Ray fields, source names (light or component), metadata keys and node names."""
    import importlib.util

    from pvtrace_amd.engine.api import EngineResult
    from tests import scenes
    from tests.util import GOLD, load_golden

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)           # (only its describe_engine_result is used: nothing of the reference is touched)

    g = load_golden("engine_result.npz")
    data = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith("data/")}
    n, max_events, record_every = (int(v) for v in g["par"])
    compiled = compile_scene(scenes.kitchen_sink())
    result = EngineResult(compiled, data, g["sources"].tolist(), max_events, record_every, 0.0)
    mine = mg.describe_engine_result(result)
    theirs = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith("ref/")}
    assert set(mine) == set(theirs), sorted(set(mine) ^ set(theirs))
    for key, want in theirs.items():
        got = mine[key]
        assert got.shape == want.shape, key
        if want.dtype.kind == "f":
            assert np.array_equal(got, want, equal_nan=True), key
        else:
            assert np.array_equal(got, want), key
    assert theirs["history_lengths"].sum() > 1000 and len(theirs["recorder_names"]) >= 8
    # ... and the pure-Python tally of those histories (reference engine/tally.py:86-150 on the reference's scene) against
    # the product's tally_histories on the product's twin: counts and histograms exactly, moments to rounding
    mine = mg.describe_recorders(tally_histories(scenes.kitchen_sink(), list(result.histories())))
    theirs = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith("tally/")}
    assert set(mine) == set(theirs)
    for key, want in theirs.items():
        got = mine[key]
        assert got.shape == want.shape, key
        if want.dtype.kind == "f":
            assert np.allclose(got, want, rtol=1e-12, atol=0, equal_nan=True), key
        else:
            assert np.array_equal(got, want), key


def test_photon_tracer_entry_points_shape_a_history_like_the_references(monkeypatch):
    """`pvtrace_amd.photon_tracer.step_forward` / `follow` (reference algorithm/photon_tracer.py:112-328) around a history the
    engine would return -- no GPU here, the history is injected: first metadata None, `follow` drops metadata, and the
    path-length rule (:162-172) ends the photon with KILL, the ray as it stood, at the first step entered beyond the limit."""
    from pvtrace_amd import photon_tracer
    from pvtrace_amd.light import Event, Ray

    def ray(t):
        return Ray(position=(0.0, 0.0, t), direction=(0.0, 0.0, 1.0), wavelength=555.0, travelled=t, source="Light")

    injected = [(ray(0.0), Event.GENERATE, {}), (ray(1.0), Event.TRANSMIT, {"hit": "slab", "container": "world", "adjacent": "slab"}),
                (ray(1.5), Event.ABSORB, {"component": "dye", "container": "slab"}), (ray(1.5), Event.EMIT, {"component": "dye", "container": "slab"}),
                (ray(3.0), Event.REFLECT, {"hit": "slab", "container": "slab", "adjacent": "world"}),
                (ray(4.0), Event.TRANSMIT, {"hit": "slab", "container": "slab", "adjacent": "world"}),
                (ray(9.0), Event.EXIT, {"hit": "world", "container": "world", "adjacent": None})]
    monkeypatch.setattr(photon_tracer, "_history", lambda *a, **k: list(injected))
    import functools

    for name in ("step_forward", "follow"):   # (the engine's form of the entry points: its history is what is injected)
        monkeypatch.setattr(photon_tracer, name, functools.partial(getattr(photon_tracer, name), backend="gpu"))
    steps = list(photon_tracer.step_forward(None, ray(0.0)))
    assert steps[0] == (ray(0.0), Event.GENERATE, None) and [e for _, e, _ in steps] == [e for _, e, _ in injected]
    assert photon_tracer.follow(None, ray(0.0)) == [(r, e) for r, e, _ in injected]
    cut = photon_tracer.follow(None, ray(0.0), maxpathlength=1.2)
    # travelled 1.5 at ABSORB does not end a step; the EMIT row does, and the next step is entered at 1.5 > 1.2
    assert [e for _, e in cut] == [Event.GENERATE, Event.TRANSMIT, Event.ABSORB, Event.EMIT, Event.KILL] and cut[-1][0] == ray(1.5)
    killed = list(photon_tracer.step_forward(None, ray(0.0), maxpathlength=3.5))[-1]
    assert killed[1] == Event.KILL and killed[2] == {"maxpathlength": 4.0, "container": "world"}
    # a limit only the closing row exceeds changes nothing (the reference checks at the START of a step)
    assert photon_tracer.follow(None, ray(0.0), maxpathlength=8.0) == [(r, e) for r, e, _ in injected]


def test_scene_simulate_refuses_a_seed_with_several_workers_and_classifies_end_rays():
    """reference tests/test_scene.py:145-157 (a seed with several workers raises ValueError -- decided before anything is
    traced, so no GPU is needed to see it) and scene/scene.py:32-58 (`is_end_ray`)."""
    from pvtrace_amd.light import Event
    from pvtrace_amd.scene import is_end_ray

    scene = scenes.fresnel_box()
    with pytest.raises(ValueError, match="Seed must be None"):
        scene.simulate(64, workers=4, seed=1)
    assert not is_end_ray(Event.ABSORB, {}) and not is_end_ray(Event.EMIT, {}) and not is_end_ray(Event.SCATTER, {})
    assert all(is_end_ray(e, None) for e in (Event.GENERATE, Event.NONRADIATIVE, Event.REACT, Event.KILL, Event.EXIT))
    into = {"hit": "slab", "container": "world", "adjacent": "slab"}
    out_of = {"hit": "slab", "container": "slab", "adjacent": "world"}
    assert is_end_ray(Event.TRANSMIT, into) and is_end_ray(Event.REFLECT, into) and is_end_ray(Event.TRANSMIT, out_of)
    assert not is_end_ray(Event.REFLECT, out_of)   # total internal reflection inside a node: not an end ray


def test_the_references_module_paths_import_and_the_package_can_answer_to_its_name():
    """`pvtrace_amd.compat`: every module path of the reference on the traced path and around it resolves to the module
    here that holds its names, under `pvtrace_amd.` always and under `pvtrace.` after `compat.install()` -- in a fresh
    interpreter, so that this process keeps no `pvtrace` of ours."""
    import subprocess
    import sys

    from pvtrace_amd.material.surface import FresnelSurfaceDelegate, SurfaceDelegate   # noqa: F401
    from pvtrace_amd.geometry.utils import angle_between, flip                          # noqa: F401
    from pvtrace_amd.light.light import Light, rectangular_mask                         # noqa: F401
    from pvtrace_amd.scene.node import Node as by_path
    from pvtrace_amd import Node
    assert by_path is Node

    script = r"""
import sys
import pvtrace_amd.compat
pvtrace_amd.compat.install()
from pvtrace import *
from pvtrace.material.utils import cone, isotropic, lambertian, henyey_greenstein
from pvtrace.material.surface import SurfaceDelegate, FresnelSurfaceDelegate, NullSurfaceDelegate, Surface
from pvtrace.material.component import Luminophore, Absorber, Scatterer, Reactor
from pvtrace.material.distribution import Distribution
from pvtrace.material.material import Material
from pvtrace.geometry.utils import flip, angle_between, EPS_ZERO, norm
from pvtrace.geometry.sphere import Sphere
from pvtrace.geometry.box import Box
from pvtrace.geometry.cylinder import Cylinder
from pvtrace.geometry.mesh import Mesh
from pvtrace.light.light import Light, rectangular_mask, CircularMask
from pvtrace.light.ray import Ray
from pvtrace.light.event import Event
from pvtrace.scene.node import Node
from pvtrace.scene.scene import Scene
from pvtrace.algorithm import photon_tracer
from pvtrace.data import lumogen_f_red_305, fluro_red
from pvtrace.device.lsc import LSC
from pvtrace.engine import simulate, simulate_stream, compile_scene, Recorder, Histogram, Heatmap, UnsupportedSceneError, is_available
from pvtrace.engine.api import EngineResult
from pvtrace.common.errors import AppError
from pvtrace.cli.parse import parse
import pvtrace
assert pvtrace is sys.modules["pvtrace_amd"] and pvtrace.engine.compile_scene is compile_scene
world = Node(name="world", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
Node(name="ball", parent=world, geometry=Sphere(radius=1.0, material=Material(refractive_index=1.5)))
Node(name="light", parent=world, light=Light(direction=lambda: cone(0.3)))
assert compile_scene(Scene(world)).node_names == ["world", "ball"]
spec = {"version": "1.0", "nodes": {"world": {"sphere": {"radius": 10.0, "material": {"refractive-index": 1.0}}},
                                    "slab": {"record": True, "box": {"size": [5, 5, 1], "material": {"refractive-index": 1.5}}},
                                    "laser": {"location": [0, 0, 3], "direction": [0, 0, -1], "light": {"wavelength": 555}}}}
names = {r.name for n in parse(spec).root.preorder() for r in n.recorders}     # (reference tests/test_engine.py:374-415)
assert len(names) == 7 and {"slab-top", "slab-lost"} <= names
try:
    from pvtrace.scene.renderer import MeshcatRenderer
except ImportError:
    print("ok")
"""
    done = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert done.returncode == 0 and done.stdout.strip() == "ok", done.stderr[-2000:]
    # a different package already imported under the name is not shadowed unless asked
    script = "import types, sys; sys.modules['pvtrace'] = types.ModuleType('pvtrace')\nimport pvtrace_amd.compat as c\ntry:\n    c.install()\nexcept ImportError:\n    c.install(force=True); import pvtrace; print(pvtrace.__name__)"
    done = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert done.returncode == 0 and done.stdout.strip() == "pvtrace_amd", done.stderr[-2000:]


def test_intersection_objects_known_answers():               # reference tests/test_intersection.py:14-23
    from pvtrace_amd import Node
    from pvtrace_amd.geometry.intersection import Intersection     # (the reference's module path: pvtrace_amd.compat)

    inter1 = Intersection(coordsys=Node, hit=Node, point=(0.0, 0.0, 0.0), distance=0.0)
    inter2 = Intersection(coordsys=Node, hit=Node, point=(0.0, 0.0, 0.0), distance=0.0)
    assert type(inter1) == Intersection and inter1 == inter2
    assert inter1 != Intersection(coordsys=Node, hit=Node, point=(0.0, 0.0, 1.0), distance=1.0)
