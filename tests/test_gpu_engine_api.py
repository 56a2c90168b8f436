"""engine.simulate() on the GPU (device-resident entry + torch plumbing): the
behaviours the reference's own tests/test_engine.py pins for its engine."""
import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd import LSC, engine
from pvtrace_amd.engine import Heatmap, Histogram, Recorder, compile_scene, tally_histories
from pvtrace_amd.engine.emit import EmitterTables, emit_bundle
from pvtrace_amd.light import Event
from tests import scenes
from tests.util import assert_bundles_identical

pytestmark = pytest.mark.gpu


def test_engine_is_available_and_native_library_is_loaded():
    import os
    assert engine.is_available()
    maps = open(f"/proc/{os.getpid()}/maps").read()
    engine.simulate(scenes.fresnel_box(), 100, seed=1, record_every=0)
    maps = open(f"/proc/{os.getpid()}/maps").read()
    assert "libpvtrace_hip.so" in maps


def test_simulate_equals_oracle_on_the_same_emitted_rays():
    scene = scenes.bench_slab(recorders=True)
    result = engine.simulate(scene, 4000, seed=11, emit_seed=3, max_events=96, emission="host")
    pos, dirs, wl, src = emit_bundle(scene, 4000, seed=3)
    cpu = O.trace_bundle(result.compiled, pos, dirs, wl, 11, 1000, 96, 0, 1, 1, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(result.data, cpu, sums_rtol=1e-12)
    assert result.sources == src and result.num_rays == 4000 and result.num_recorded == 4000
    assert result.elapsed > 0 and result.kernel_ms > 0


def test_engine_is_deterministic_for_seed_and_ignores_workers():
    scene = scenes.bench_slab()
    a = engine.simulate(scene, 500, seed=123, workers=4, emit_seed=1)
    b = engine.simulate(scene, 500, seed=123, workers=2, emit_seed=1)
    for key in a.data:
        assert np.array_equal(a.data[key], b.data[key]), key


def test_histories_look_like_python_histories():
    result = engine.simulate(scenes.fresnel_box(), 10, seed=5)
    histories = list(result.histories())
    assert len(histories) == 10
    for history in histories:
        ray, event, meta = history[0]
        assert event == Event.GENERATE and ray.travelled == 0.0 and ray.source == "Light"
        ray, event, meta = history[-1]
        assert event in (Event.EXIT, Event.NONRADIATIVE, Event.KILL, Event.REACT)
        if event == Event.EXIT:
            assert meta["hit"] == "world"
    assert result.event_counts()[Event.EXIT] == 10


def test_recorders_match_event_log_and_python_tally():
    scene = scenes.bench_slab(recorders=True)
    result = engine.simulate(scene, 4000, seed=21, max_events=256, emit_seed=2)
    recs = result.recorders
    n_enter = n_top = n_lost = n_exit = 0
    for history in result.histories():
        entered = top = lost = exited = False
        for ray, event, meta in history:
            if event == Event.TRANSMIT and meta["hit"] == "slab":
                if meta["adjacent"] == "slab":
                    entered = True
                elif meta["container"] == "slab":
                    nx, ny, nz = meta["normal"]
                    top |= abs(nx) <= 1e-6 and abs(ny) <= 1e-6 and abs(nz - 1.0) <= 1e-6
            elif event == Event.NONRADIATIVE and meta["container"] == "slab":
                lost = True
            elif event == Event.EXIT:
                exited = True
        n_enter += entered; n_top += top; n_lost += lost; n_exit += exited
    assert recs["entering"].rays == n_enter > 0 and recs["top"].rays == n_top > 0
    assert recs["lost"].rays == n_lost > 0 and recs["exit"].rays == n_exit > 0
    assert recs["entering"].mean("wavelength") == pytest.approx(555.0)
    assert recs["entering"].crossings >= recs["entering"].rays
    python_side = tally_histories(scene, result.histories())
    for name, rec in recs.items():
        assert python_side[name].rays == rec.rays and python_side[name].crossings == rec.crossings
        for i in range(len(rec.spec.histograms)):
            assert np.array_equal(python_side[name]._bins[i], rec._bins[i]), (name, i)
        assert np.allclose(python_side[name]._moments, rec._moments, rtol=1e-9)


def test_recorders_independent_of_history_sampling():
    scene = scenes.bench_slab(recorders=True)
    full = engine.simulate(scene, 3000, seed=9, record_every=1, max_events=256, emit_seed=4)
    sampled = engine.simulate(scene, 3000, seed=9, record_every=100, emit_seed=4)
    none = engine.simulate(scene, 3000, seed=9, record_every=0, emit_seed=4)
    assert full.recorders["entering"].rays == sampled.recorders["entering"].rays == none.recorders["entering"].rays
    assert full.recorders["entering"].crossings == none.recorders["entering"].crossings
    assert (full.num_recorded, sampled.num_recorded, none.num_recorded) == (3000, 30, 0)
    assert len(list(sampled.histories())) == 30 and len(list(none.histories())) == 0
    assert np.array_equal(sampled.recorded_indices, np.arange(0, 3000, 100))
    # the sampled histories are exactly those rays' histories in the full log
    m = 256
    for j, i in enumerate(sampled.recorded_indices):
        k = int(sampled.data["counts"][j])
        assert k == full.data["counts"][i]
        assert np.array_equal(sampled.data["kind"][j * 128:j * 128 + k], full.data["kind"][i * m:i * m + k])


def test_simulate_stream_accumulates_to_a_single_call():
    scene = scenes.lsc_equivalent()
    total = None
    for result, traced in engine.simulate_stream(scene, 30000, bundle=8000, seed=77, record_every=0,
                                                 emission="device", emit_seed=5):
        part = {k: result.data[k].copy() for k in ("rec_distinct", "rec_crossings", "rec_bins")}
        total = part if total is None else {k: total[k] + part[k] for k in part}
    assert traced == 30000
    whole = engine.simulate(scene, 30000, seed=77, record_every=0, emission="device", emit_seed=5)
    for key in total:
        assert np.array_equal(total[key], whole.data[key]), key


def test_device_emission_mode_equals_oracle():
    scene = scenes.kitchen_sink()
    result = engine.simulate(scene, 6000, seed=3, emission="device", emit_seed=12, max_events=64)
    pos, dirs, wl = O.emit(EmitterTables(scene), 6000, emit_seed=12)
    cpu = O.trace_bundle(result.compiled, pos, dirs, wl, 3, 1000, 64, 0, 1, 1, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(result.data, cpu, sums_rtol=1e-12)
    assert result.sources[:4] == ["lamp", "glow", "lamp", "glow"]


def test_lsc_high_level_api_runs_on_the_engine():
    """LSC().simulate/counts/summary (reference lsc.py:338-620) on recorder tallies."""
    n = 200000
    lsc = LSC((5.0, 5.0, 1.0))
    lsc.simulate(n, seed=4, emit_seed=6)
    c = lsc.counts()
    s = lsc.summary()
    assert abs(c["Solar In"]["top"] / n - 0.960) < 0.003 and c["Solar In"]["bottom"] == 0
    assert abs(c["Solar Out"]["top"] / n - 0.040) < 0.003            # Fresnel reflection off the top
    lum_out = sum(c["Luminescent Out"][f] for f in ("left", "right", "near", "far", "top", "bottom"))
    assert abs(lum_out / n - 0.62) < 0.01 and c["Luminescent In"]["top"] == 0
    assert s["Optical Efficiency"] == 0.0 and s["Incident"] == sum(c["Solar In"])   # no cells attached
    assert abs(s["Non-radiative Loss (fraction):"] - 0.340 / 0.960) < 0.01
    # spectrum(facets, kind, source, events): the reference's signature (lsc.py:505-566) and return form -- one wavelength
    # per selected ray (known to its 5 nm bin) -- from tallies, the exact histogram riding along as .edges / .counts
    lum = lsc.spectrum(source=lsc.component_names(), events={"transmit"})
    edges = lum.edges
    assert len(lum) == lum.counts.sum() == lum_out and edges[np.argmax(lum.counts)] > 580 and 580 < np.median(lum) < 700
    assert np.array_equal(np.histogram(lum, bins=edges)[0], lum.counts)     # what `plt.hist(lsc.spectrum(...), bins=...)` draws
    recs = lsc._result.recorders
    assert len(lsc.spectrum()) == n                      # kind="last", every source: one row per photon (lost, out, or reflected)
    first = lsc.spectrum(kind="first")
    assert len(first) == n and first.counts[np.searchsorted(edges, 555.0, side="right") - 1] == n    # the lamp's 555 nm, in or reflected
    assert abs(float(first[0]) - 555.0) <= 2.5
    assert len(lsc.spectrum(facets={"top"}, kind="first", source="Light", events={"transmit"})) == c["Solar In"]["top"]
    lost = lsc.spectrum(events={"absorb"})
    assert len(lost) == recs["lost"].rays
    lost_lum = lsc.spectrum(events={"absorb"}, source="Lumogen F Red 305")
    assert len(lost_lum) == recs["lost-lum"].rays and 0 < len(lost_lum) < len(lost)
    edge_lum = lsc.spectrum(facets={"left", "right"}, source={"Lumogen F Red 305"})
    assert len(edge_lum) == c["Luminescent Out"]["left"] + c["Luminescent Out"]["right"]
    assert len(lsc.spectrum(source="Background")) == 0               # an absorber emits nothing
    assert len(lsc.spectrum(kind=None)) == 2 * n                     # both rows of every photon
    for bad in (dict(kind="middle"), dict(source="Sun"), dict(facets="top"), dict(events={"teleport"}), dict(events="absorb")):
        with pytest.raises(ValueError):
            lsc.spectrum(**bad)
    cells = LSC((5.0, 5.0, 1.0)); cells.add_solar_cell({"left", "right", "near", "far"}); cells.add_back_surface_mirror()
    cells.simulate(n, seed=4, emit_seed=6)
    t, ct = cells.summary(), cells.counts()
    assert ct["Luminescent Out"]["bottom"] == 0 and ct["Solar Out"]["bottom"] == 0   # perfect back mirror
    assert 0.2 < t["Optical Efficiency"] < 0.6 and 0.5 < t["Waveguide Efficiency"] < 0.9
    assert abs(t["Waveguide Efficiency (Thermodynamic Prediction)"] - 2.25 / (1.25 + 2.25)) < 1e-12
    mirror = LSC((5.0, 5.0, 1.0)); mirror.add_air_gap_mirror(lambertian=True)
    r = mirror.simulate(50000, seed=4, emit_seed=6)
    assert r.compiled.node_names == ["World", "LSC", "Air Gap Mirror"]


def test_source_filtered_recorders_and_auto_instrumentation():
    from pvtrace_amd.engine import auto_recorders, instrument, recorders_from_spec

    scene = scenes.bench_slab(recorders=False)
    slab = scene.root.children[0]
    instrument(slab)                                  # the `record: true` shorthand
    recorders_from_spec({
        "solar-out": {"node": "slab", "event": "escaping", "source": "lights"},
        "lum-out": {"node": "slab", "event": "escaping", "source": "components",
                    "histograms": {"wavelength": [300, 1000, 70], "position": ["x", "y", [-2.5, 2.5, 10], [-2.5, 2.5, 10]]}},
        "dye-out": {"node": "slab", "event": "escaping", "source": "dye"},
        "all-out": {"node": "slab", "event": "escaping"},
    }, {"slab": slab})
    assert {r.name for r in auto_recorders(slab)} == {"slab-lost", "slab-top", "slab-bottom", "slab-east",
                                                      "slab-west", "slab-north", "slab-south"}
    result = engine.simulate(scene, 20000, seed=3, emit_seed=1, max_events=200)
    recs = result.recorders
    assert recs["lum-out"].rays == recs["dye-out"].rays > 0 and recs["solar-out"].rays > 0
    assert recs["all-out"].crossings == recs["lum-out"].crossings + recs["solar-out"].crossings
    assert recs["lum-out"].mean("wavelength") > 570 > 556 > recs["solar-out"].mean("wavelength") - 1e-9
    faces = sum(recs[f"slab-{f}"].crossings for f in ("top", "bottom", "east", "west", "north", "south"))
    assert faces == recs["all-out"].crossings
    # the pure-Python tally (with the same source semantics) reproduces every recorder exactly
    python_side = tally_histories(scene, result.histories())
    for name, rec in recs.items():
        assert python_side[name].rays == rec.rays and python_side[name].crossings == rec.crossings, name
    # and the oracle agrees bit for bit
    pos, dirs, wl, _ = emit_bundle(scene, 20000, seed=1)
    cpu = O.trace_bundle(result.compiled, pos, dirs, wl, 3, 1000, 200, 0, 1, 1, math_mode=O.MATH_PORTABLE)
    assert_bundles_identical(result.data, cpu, sums_rtol=1e-12)


def test_sharded_simulate_over_rccl_world_size_one():
    """The N>1 code path (index-range shard + RCCL all-reduce of int64/f64 tallies) on the one
    GPU we have: a single-rank nccl group must reproduce simulate() exactly."""
    import torch.distributed as dist

    from pvtrace_amd.engine.distributed import simulate_sharded

    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
    try:
        scene = scenes.lsc_equivalent()
        sharded = simulate_sharded(scene, 300000, seed=8, emit_seed=2, record_every=0, device=0)
        whole = engine.simulate(scene, 300000, seed=8, record_every=0, emission="device", emit_seed=2)
        for key in ("rec_distinct", "rec_crossings", "rec_bins"):
            assert np.array_equal(sharded.data[key], whole.data[key]), key
        assert sharded.shard == (0, 300000)
        # the pipelined job: tallies summed over the ranks once at the end, or per bundle
        from pvtrace_amd.engine import BundlePipeline, compile_scene, native
        from pvtrace_amd.engine.emit import EmitterTables

        dscene = native.DeviceScene(compile_scene(scene), device=0, emitter=EmitterTables(scene))
        try:
            for mode in ("end", "bundle"):
                pipe = BundlePipeline(dscene, depth=3, distributed=True, reduce=mode)
                for k in range(5):
                    pipe.submit(None, 60000, seed=8, ray_offset=60000 * k, emit_seed=2, emit_method=0)
                totals = pipe.totals_host()
                for key in ("rec_distinct", "rec_crossings", "rec_bins"):
                    assert np.array_equal(totals[key], whole.data[key]), (mode, key)
                if mode == "end":
                    with pytest.raises(RuntimeError):
                        pipe.submit(None, 10, seed=1)
        finally:
            dscene.close()
    finally:
        dist.destroy_process_group()


def test_unsupported_scene_and_bad_arguments_raise():
    from pvtrace_amd import Box, Light, Material, Node, Scene, Sphere, Surface, SurfaceDelegate

    class Custom(SurfaceDelegate):
        def reflectivity(self, *a): return 0.5
        def reflected_direction(self, *a): return (0, 0, 1)
        def transmitted_direction(self, *a): return (0, 0, 1)
    w = Node(name="w", geometry=Sphere(10.0, material=Material(1.0)))
    Node(name="n", parent=w, geometry=Box((1, 1, 1), material=Material(1.5, surface=Surface(delegate=Custom()))))
    Node(name="l", parent=w, light=Light())
    with pytest.raises(engine.UnsupportedSceneError):
        engine.simulate(Scene(w), 10)
    with pytest.raises(ValueError):
        engine.simulate(scenes.fresnel_box(), 10, emit_method="nope")
    w2 = Node(name="w", geometry=Sphere(5.0, material=Material(1.0)))
    Node(name="l", parent=w2, light=Light(direction=lambda: (0.0, 1.0, 0.0)))
    with pytest.raises(engine.UnsupportedSceneError):
        engine.simulate(Scene(w2), 10, emission="device")
    r = engine.simulate(Scene(w2), 10, seed=1)    # host fallback emitter still works
    assert r.event_counts()[Event.EXIT] == 10


def test_pipelined_bundles_equal_serial_bundles():
    """BundlePipeline keeps two bundles in flight on two HIP streams (their launches
    overlap and share the workgroup-consolidation machinery); tallies must equal the
    serial schedule and a single big call exactly."""
    import torch

    from pvtrace_amd.engine import BundlePipeline, native, trace_stream

    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    n, bundles = 200_000, 7
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=21)
    dscene = native.DeviceScene(compiled, device=0)
    rays = tuple(torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (pos, dirs, wl))
    results = {}
    for depth in (1, 2, 3):
        pipe = BundlePipeline(dscene, depth=depth)
        pipe.wait_for_inputs()
        for k in range(bundles):
            pipe.submit(rays, n, seed=1000 + k * n, maxsteps=1000, emit_method=0)
        results[depth] = pipe.totals_host()
        assert len(pipe.kernel_ms()) == bundles
    for key in ("rec_distinct", "rec_crossings", "rec_bins"):
        assert np.array_equal(results[1][key], results[2][key]), key
        assert np.array_equal(results[1][key], results[3][key]), key
    assert np.allclose(results[1]["rec_sums"], results[2]["rec_sums"], rtol=1e-11)
    # oracle check of one bundle's worth
    cpu = O.trace_bundle(compiled, pos, dirs, wl, 1000, 1000, 128, 0, 8, 0, math_mode=O.MATH_PORTABLE)
    pipe = BundlePipeline(dscene, depth=2)
    pipe.wait_for_inputs()
    pipe.submit(rays, n, seed=1000)
    one = pipe.totals_host()
    for key in ("rec_distinct", "rec_crossings", "rec_bins"):
        assert np.array_equal(one[key], cpu[key]), key
    dscene.close()
    # streamed job with device emission == one simulate call
    c2, data, _ = trace_stream(scene, 500_000, bundle=120_000, seed=5, emit_seed=9, depth=2)
    whole = engine.simulate(scene, 500_000, seed=5, record_every=0, emission="device", emit_seed=9)
    for key in ("rec_distinct", "rec_crossings", "rec_bins"):
        assert np.array_equal(data[key], whole.data[key]), key


def test_empty_and_tiny_jobs_through_the_public_api():
    scene = scenes.lsc_equivalent()
    for emission in ("device", "host"):
        none = engine.simulate(scene, 0, seed=1, emission=emission)
        assert none.num_rays == 0 and none.num_recorded == 0 and list(none.histories()) == []
        assert none.data["rec_distinct"].sum() == 0 and none.data["kind"].shape == (0,)
        one = engine.simulate(scene, 1, seed=1, emission=emission, record_every=0)
        assert one.num_rays == 1 and one.data["counts"].shape == (0,) and one.data["rec_distinct"].sum() >= 1
        assert list(engine.simulate_stream(scene, 0, seed=1, emission=emission)) == []
        sizes = [r.num_rays for r, _ in engine.simulate_stream(scene, 1001, bundle=250, seed=1,
                                                               emission=emission, record_every=7)]
        assert sizes == [250, 250, 250, 250, 1]
    assert engine.simulate(scene, 5, seed=2, max_events=2).data["counts"].tolist() == [2] * 5
    with pytest.raises(ValueError):
        engine.simulate(scene, 5, seed=2, max_events=1)
    with pytest.raises(ValueError):
        engine.simulate(scene, -3, seed=2)


def test_the_binding_printed_in_INTEGRATION_md_works_as_documented():
    """INTEGRATION.md shows the ctypes module a pvtrace maintainer would drop in as
    `pvtrace/engine/_kernel_hip.py`.  Execute exactly that text (library path aside) against the
    flattener's tables and compare with the package's own binding."""
    import os
    import re

    from pvtrace_amd.engine import native

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C, numpy as np\n.*?)```", text, flags=re.S).group(1)
    native.load_library()                                   # HIP runtime of torch first (native.py)
    code = code.replace('C.CDLL("libpvtrace_hip.so")', f"C.CDLL({native.LIB_PATH!r})")
    module = {}
    exec(compile(code, "INTEGRATION.md", "exec"), module)
    scene = scenes.bench_slab(recorders=True)
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, 3000, seed=4)
    for record_every in (1, 0):
        theirs = module["trace_bundle"](compiled, pos, dirs, wl, 9, 1000, 48, 0, 1, record_every)
        ours = _kernel_module().trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 0, 1, record_every)
        assert_bundles_identical(theirs, ours, sums_rtol=1e-12, what="INTEGRATION.md stub")


def test_the_INTEGRATION_md_stub_splits_a_bundle_over_a_device_list():
    """Same stub, `devices=(0, 0)`: two shards (here on the one GPU of the test box) give the
    single-device arrays — event log rows in their global slots, integer tallies exact."""
    import os
    import re

    from pvtrace_amd.engine import native

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes as C, numpy as np\n.*?)```", text, flags=re.S).group(1)
    native.load_library()
    code = code.replace('C.CDLL("libpvtrace_hip.so")', f"C.CDLL({native.LIB_PATH!r})")
    module = {}
    exec(compile(code, "INTEGRATION.md", "exec"), module)
    scene = scenes.bench_slab(recorders=True)
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, 3001, seed=4)
    for record_every in (1, 7, 0):
        one = module["trace_bundle"](compiled, pos, dirs, wl, 9, 1000, 48, 0, 1, record_every)
        for devices in ((0, 0), (0, 0, 0)):
            many = module["trace_bundle"](compiled, pos, dirs, wl, 9, 1000, 48, 0, 1, record_every, devices=devices)
            assert_bundles_identical(many, one, sums_rtol=1e-12, what=f"devices={devices} record_every={record_every}")


def test_device_list_entry_sums_the_tallies_with_rccl_when_asked(monkeypatch):
    """pvt_trace_bundle_multi sums the shards' tallies ON THE DEVICES (ncclCommInitAll over the device list +
    ncclReduce) when every entry is a different GPU.  The one-GPU test box can only run that path as a communicator
    of ONE rank (PVT_MULTI_REDUCE=rccl forces it): RCCL really loads, builds the communicator, reduces the four
    buffers and the result is the single-device result.  Two entries on the same GPU fall back to the host sum."""
    from pvtrace_amd.engine import _kernel, native

    scene = scenes.bench_slab(recorders=True)
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, 3001, seed=4)
    lib = native.load_library()
    one = _kernel.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 0, 1, 7)
    monkeypatch.setenv("PVT_MULTI_REDUCE", "rccl")
    forced = _kernel.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 0, 1, 7, devices=[0])
    assert lib.pvt_last_multi_reduce() == 2                     # summed on the device by RCCL
    assert_bundles_identical(forced, one, sums_rtol=1e-12, what="rccl, one rank")
    monkeypatch.delenv("PVT_MULTI_REDUCE")
    two = _kernel.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 0, 1, 7, devices=[0, 0])
    assert lib.pvt_last_multi_reduce() == 1                     # same GPU twice: host sum
    assert_bundles_identical(two, one, sums_rtol=1e-12, what="host sum")


def test_rccl_reduce_of_a_ten_thousand_bin_scene_and_the_reduce_overrides(monkeypatch):
    """The device-side sum of pvt_trace_bundle_multi at the buffer sizes of a real scene (a 100 x 100 heat map: 10 000
    bins next to the recorder counts) -- as a one-rank communicator, the only kind this box has -- twice in a row (the
    communicator of a device list is built once and kept), then the same job under PVT_MULTI_REDUCE=host and over a
    device list that names the GPU three times (host sum): every variant is the single-device result."""
    from pvtrace_amd.engine import Heatmap, Recorder, _kernel, native

    scene = scenes.bench_slab(recorders=False)
    slab = scene.root.children[0]
    slab.recorders = [Recorder("map", event="entering", histograms=[Heatmap("x", "y", (-2.5, 2.5, 100), (-2.5, 2.5, 100))]),
                      Recorder("lost", event="lost"), Recorder("out", event="escaping")]
    compiled = compile_scene(scene)
    assert int(compiled.total_bins) == 10_000
    pos, dirs, wl, _ = emit_bundle(scene, 20_000, seed=4)
    lib = native.load_library()
    one = _kernel.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 1, 1, 0)
    assert one["rec_bins"].sum() == one["rec_distinct"][0] > 10_000
    monkeypatch.setenv("PVT_MULTI_REDUCE", "rccl")
    for attempt in range(2):
        forced = _kernel.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 1, 1, 0, devices=[0])
        assert lib.pvt_last_multi_reduce() == 2
        assert_bundles_identical(forced, one, sums_rtol=1e-12, what=f"rccl, one rank, attempt {attempt}")
    monkeypatch.setenv("PVT_MULTI_REDUCE", "host")
    for devices in ([0], [0, 0, 0]):
        host = _kernel.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 1, 1, 0, devices=devices)
        assert_bundles_identical(host, one, sums_rtol=1e-12, what=f"host sum, devices={devices}")
    assert lib.pvt_last_multi_reduce() == 1
    monkeypatch.delenv("PVT_MULTI_REDUCE")
    thrice = _kernel.trace_bundle(compiled, pos, dirs, wl, 9, 1000, 48, 1, 1, 0, devices=[0, 0, 0])
    assert lib.pvt_last_multi_reduce() == 1            # a repeated id can never be an RCCL communicator: host sum
    assert_bundles_identical(thrice, one, sums_rtol=1e-12, what="repeated device id")


@pytest.mark.parametrize("emission", ["host", "device"])
def test_simulate_over_a_device_list_equals_one_device(emission):
    scene = scenes.bench_slab(recorders=True)
    kw = dict(seed=17, emit_seed=5, max_events=64, emission=emission)
    for record_every in (1, 5, 0):
        if emission == "host":
            np.random.seed(3)
        one = engine.simulate(scene, 2503, record_every=record_every, **kw)
        if emission == "host":
            np.random.seed(3)
        two = engine.simulate(scene, 2503, record_every=record_every, devices=[0, 0], **kw)
        assert_bundles_identical(two.data, one.data, sums_rtol=1e-12, what=f"{emission} {record_every}")
        assert two.sources == one.sources and two.num_recorded == one.num_recorded
    with pytest.raises(ValueError):
        engine.simulate(scene, 10, seed=1, device=0, devices=[0])


def test_simulate_stream_over_a_device_list_yields_the_same_bundles_in_order():
    scene = scenes.bench_slab(recorders=True)
    kw = dict(bundle=700, seed=23, emit_seed=9, record_every=0)
    one = list(engine.simulate_stream(scene, 3000, **kw))
    two = list(engine.simulate_stream(scene, 3000, devices=[0, 0], **kw))
    assert [t for _, t in one] == [t for _, t in two] == [700, 1400, 2100, 2800, 3000]
    for (a, _), (b, _) in zip(one, two):
        assert_bundles_identical(b.data, a.data, sums_rtol=1e-12, what="stream over devices")


@pytest.mark.parametrize("emission", ["device", "host"])
def test_grouped_stream_bundles_equal_bundles_traced_one_by_one(emission):
    """In tally mode `simulate_stream` traces a group of bundles with one launch (one tally set per
    bundle, PvtTraceParams.tally_bundle).  Every yielded bundle must be what tracing that bundle alone
    gives (reference contract api.py:249-264: bundle b = rays [b*m, (b+1)*m) with seeds seed + traced + i)."""
    scene = scenes.bench_slab(recorders=True)
    m, total, seed, emit_seed = 600, 2000, 41, 13
    if emission == "host":
        np.random.seed(5)
    got = list(engine.simulate_stream(scene, total, bundle=m, seed=seed, emit_seed=emit_seed, record_every=0,
                                      emission=emission))
    assert [t for _, t in got] == [600, 1200, 1800, 2000]
    traced = 0
    for result, upto in got:
        n = upto - traced
        if emission == "device":
            alone = engine.simulate(scene, n, seed=seed, emit_seed=emit_seed, ray_offset=traced, record_every=0,
                                    emission="device")
        else:
            alone = engine.simulate(scene, n, seed=seed + traced, emit_seed=emit_seed + traced, record_every=0,
                                    emission="host")
        assert_bundles_identical(result.data, alone.data, sums_rtol=1e-12, what=f"bundle at {traced}")
        assert list(result.sources) == list(alone.sources) and result.num_rays == n
        traced = upto
    # and the C entry with host buffers: sets with strides
    from pvtrace_amd.engine import _kernel, native as N

    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, total, seed=2)
    whole = [_kernel.trace_bundle(compiled, pos[a:a + m], dirs[a:a + m], wl[a:a + m], seed, 1000, 16, 0, 1, 0,
                                  ray_offset=a) for a in range(0, total, m)]
    sets = _kernel.trace_bundle_sets(compiled, pos, dirs, wl, seed, 1000, 0, m)
    for j, want in enumerate(whole):
        for key in ("rec_distinct", "rec_crossings", "rec_bins"):
            assert np.array_equal(sets[j][key], want[key]), (j, key)
        assert np.allclose(sets[j]["rec_sums"], want["rec_sums"], rtol=1e-12, atol=0)


def test_packed_event_log_holds_exactly_the_written_rows():
    """`packed_log=True` (opt-in): the same histories without the reference's dense padding."""
    scene = scenes.bench_slab(recorders=True)
    dense = engine.simulate(scene, 3000, seed=51, emit_seed=4, max_events=64)
    packed = engine.simulate(scene, 3000, seed=51, emit_seed=4, max_events=64, packed_log=True)
    assert packed.packed and not dense.packed
    assert np.array_equal(packed.data["counts"], dense.data["counts"])
    assert packed.data["row_start"][-1] == dense.data["counts"].sum() == len(packed.data["kind"])
    for key in ("rec_distinct", "rec_crossings", "rec_bins"):
        assert np.array_equal(packed.data[key], dense.data[key])
    for j in (0, 1, 17, 2999):
        a, b = dense.rows_of(j), packed.rows_of(j)
        for name in ("kind", "hit", "container", "adjacent", "component", "source", "position", "direction",
                     "normal", "wavelength", "travelled", "duration"):
            assert np.array_equal(dense.data[name][a], packed.data[name][b]), (j, name)
    assert packed.event_counts() == dense.event_counts()
    for h_dense, h_packed in zip(dense.histories(), packed.histories()):
        assert [(e, m) for _, e, m in h_dense] == [(e, m) for _, e, m in h_packed]
        assert [r.position for r, _, _ in h_dense] == [r.position for r, _, _ in h_packed]


def test_the_engine_runs_from_worker_threads_like_the_studio_consumer():
    """The reference's streaming consumer drives the engine from an executor thread
    (studio/server.py:225-230); two threads may also trace two scenes on one GPU at once.  Results are
    those of the same calls made serially from the main thread."""
    import concurrent.futures

    slab, box = scenes.bench_slab(recorders=True), scenes.fresnel_box()

    def stream(scene, seed):
        return [r.data for r, _ in engine.simulate_stream(scene, 2400, bundle=800, seed=seed, emit_seed=seed + 1,
                                                          record_every=3, max_events=48)]

    serial = [stream(slab, 31), stream(box, 37)]
    with concurrent.futures.ThreadPoolExecutor(max_workers=1) as pool:     # off the main thread
        off_main = pool.submit(stream, slab, 31).result()
    with concurrent.futures.ThreadPoolExecutor(max_workers=2) as pool:     # two scenes at once
        futures = [pool.submit(stream, slab, 31), pool.submit(stream, box, 37)]
        both = [f.result() for f in futures]
    for got, want in ((off_main, serial[0]), (both[0], serial[0]), (both[1], serial[1])):
        assert len(got) == len(want) == 3
        for a, b in zip(got, want):
            assert_bundles_identical(a, b, sums_rtol=1e-12, what="threaded stream")


def _kernel_module():
    from pvtrace_amd.engine import _kernel

    return _kernel


def test_resident_scenes_are_reused_and_an_edited_scene_is_a_new_one():
    """`simulate` keeps the last few scenes resident on the GPU (digest of every flat table).  Same scene again: the
    same device scene, same results; a scene edited in between: a different digest, hence different tables on the
    device and the results of the EDITED scene; two sessions at once never share one resident scene."""
    from pvtrace_amd.engine import api

    engine.release_resident_scenes()
    scene = scenes.bench_slab(recorders=True)
    kw = dict(seed=3, emit_seed=4, record_every=0, emission="device")
    a = engine.simulate(scene, 20_000, **kw)
    assert len(api._RESIDENT) == 1
    handle = api._RESIDENT[0][1].handle.value
    b = engine.simulate(scene, 20_000, **kw)
    assert len(api._RESIDENT) == 1 and api._RESIDENT[0][1].handle.value == handle       # reused
    assert_bundles_identical(b.data, a.data, sums_rtol=1e-12, what="reused scene")
    slab = [n for n in scene.root.children if n.geometry is not None][0]
    slab.geometry.material.refractive_index = 1.7                                         # edit the scene
    c = engine.simulate(scene, 20_000, **kw)
    fresh = engine.simulate(scenes.bench_slab(recorders=True), 20_000, **kw)
    assert len(api._RESIDENT) == 2
    assert not np.array_equal(c.data["rec_distinct"], a.data["rec_distinct"])            # n = 1.7 reflects more
    assert_bundles_identical(fresh.data, a.data, sums_rtol=1e-12, what="original scene again")
    with engine.Session(scene, device=0) as s1, engine.Session(scene, device=0) as s2:
        assert s1.dscene is not s2.dscene
        r1, r2 = s1.run(5000, 1, record_every=0, emit_seed=2), s2.run(5000, 1, record_every=0, emit_seed=2)
        assert_bundles_identical(r1.data, r2.data, sums_rtol=1e-12, what="two sessions")
    for _ in range(6):     # the cache is bounded
        other = scenes.bench_slab(recorders=True)
        [n for n in other.root.children if n.geometry is not None][0].geometry.material.refractive_index = 1.1 + 0.1 * _
        engine.simulate(other, 100, **kw)
    assert len(api._RESIDENT) <= api._RESIDENT_MAX
    engine.release_resident_scenes()
    assert api._RESIDENT == []


@pytest.mark.gpu
def test_dense_log_columns_are_built_on_demand_and_equal_the_packed_rows():
    from pvtrace_amd.engine.api import EngineResult, LazyLogColumns
    from pvtrace_amd.engine.native import EVENT_LOG_COLUMNS

    scene = scenes.lsc_equivalent()
    result = engine.simulate(scene, 3000, seed=3, emit_seed=8)
    assert isinstance(result.data, LazyLogColumns) and len(result.data._builders) == len(EVENT_LOG_COLUMNS)
    counts, histories = result.event_counts(), list(result.histories())
    assert len(result.data._builders) == len(EVENT_LOG_COLUMNS)          # neither needed a dense column
    position = result.data["position"]
    assert position.shape == (3000 * result.max_events, 3) and len(result.data._builders) == len(EVENT_LOG_COLUMNS) - 1
    dense = EngineResult(result.compiled, dict(result.data), result.sources, result.max_events, result.record_every,
                         result.elapsed)
    assert type(dense.data) is dict and dense.event_counts() == counts
    assert [[(repr(r), e, m) for r, e, m in h] for h in dense.histories()] == \
           [[(repr(r), e, m) for r, e, m in h] for h in histories]
    packed = engine.simulate(scene, 3000, seed=3, emit_seed=8, packed_log=True)
    for name, _, _ in EVENT_LOG_COLUMNS:
        assert np.array_equal(packed.data[name], result.data.packed_rows[name])
    assert np.array_equal(packed.data["row_start"], result.data.row_start)
    # the dense columns are the reference's: pre-filled, the written rows at j * max_events + k
    j = int(np.argmax(result.data["counts"]))
    rows = result.rows_of(j)
    assert np.array_equal(dense.data["kind"][rows], packed.data["kind"][packed.rows_of(j)])
    assert dense.data["hit"][rows.stop] == -1 or rows.stop % result.max_events == 0


def test_simulate_is_a_drop_in_for_the_references_simulate_under_the_same_seeds():
    """tests/golden/engine_result.npz holds what the REFERENCE's own pipeline -- its compile_scene, its emit_bundle under
    numpy seed 77, its compiled kernel with seed 31, `emit_method="full"`, every third ray recorded -- returned for the
    kitchen-sink scene.  `engine.simulate` with host emission draws the same rays from the same global generator (bit for
    bit: tests/test_scene_api.py) and traces them with the reference's per-ray seeds; what differs is <= 1 ulp in a few
    transcendental calls, so: the GENERATE rows are identical, all but a handful of recorded rays have the
    reference's event sequence with the reference's ids, and the integer tallies differ by a few photons at most."""
    from tests.util import load_golden

    g = load_golden("engine_result.npz")
    ref = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith("data/")}
    n, max_events, record_every = (int(v) for v in g["par"])
    np.random.seed(77)
    result = engine.simulate(scenes.kitchen_sink(), n, seed=31, max_events=max_events, emit_method="full",
                             record_every=record_every, emission="host")
    assert result.sources == g["sources"].tolist()
    got = result.data
    first = np.arange(len(ref["counts"])) * max_events
    for key in ("position", "direction", "wavelength"):
        assert np.array_equal(np.asarray(got[key])[first], ref[key][first]), key        # the emitted rays themselves
    same = 0
    for j in range(len(ref["counts"])):
        rows = slice(j * max_events, j * max_events + int(ref["counts"][j]))
        if got["counts"][j] == ref["counts"][j] and np.array_equal(np.asarray(got["kind"])[rows], ref["kind"][rows]):
            same += 1
            for key in ("hit", "container", "adjacent", "component", "source"):
                assert np.array_equal(np.asarray(got[key])[rows], ref[key][rows]), (j, key)
            assert np.allclose(np.asarray(got["position"])[rows], ref["position"][rows], rtol=0, atol=1e-8)
    assert same >= len(ref["counts"]) - 3, (same, len(ref["counts"]))
    assert np.abs(np.asarray(got["rec_distinct"]) - ref["rec_distinct"]).max() <= 5
    assert np.abs(np.asarray(got["rec_crossings"]) - ref["rec_crossings"]).max() <= 8


def test_photon_tracer_follow_traces_ray_by_ray_like_the_references_loop():
    """reference `for ray in scene.emit(n): history = photon_tracer.follow(scene, ray)` (algorithm/photon_tracer.py:276-328,
    the loop of its `LSC.simulate`, device/lsc.py:349-350): the same loop here, each ray a bundle of one on the engine.  The
    history of a ray is what `engine.simulate` gives for the same ray and stream (the oracle's, bit for bit), the path-length
    limit ends it where the reference's rule does, and a reused Session keeps the scene resident."""
    from pvtrace_amd import photon_tracer
    from pvtrace_amd.engine.api import Session

    scene = scenes.lsc_equivalent()
    np.random.seed(8)
    rays = list(scene.emit(12))
    compiled = compile_scene(scene)
    with Session(scene, emission="host") as session:
        for k, ray in enumerate(rays):
            history = photon_tracer.follow(scene, ray, seed=500 + k, session=session)
            assert history[0][1] == Event.GENERATE and history[0][0].position == tuple(ray.position)
            assert history[-1][1] in (Event.EXIT, Event.NONRADIATIVE, Event.KILL, Event.REACT, Event.ABSORB)
            pos, d, wl = (np.asarray([ray.position]), np.asarray([ray.direction]), np.asarray([ray.wavelength]))
            cpu = O.trace_bundle(compiled, pos, d, wl, 500 + k, 1000, 2008, 0, 1, 1, math_mode=O.MATH_PORTABLE)
            n = int(cpu["counts"][0])
            assert len(history) == n
            assert [e.value for _, e in history] == cpu["kind"][:n].tolist()
            assert np.array_equal(np.array([r.position for r, _ in history]), cpu["position"][:n])
            assert np.array_equal(np.array([r.wavelength for r, _ in history]), cpu["wavelength"][:n])
            # metadata in the generator's rows, None on the first as in the reference
            steps = list(photon_tracer.step_forward(scene, ray, seed=500 + k, session=session))
            assert steps[0][2] is None and all(set(m) >= {"container"} for _, e, m in steps[1:] if e != Event.KILL)
    # without a session and without a seed: numpy's generator names the stream, so a seeded sequence repeats
    np.random.seed(3)
    a = photon_tracer.follow(scene, rays[0])
    np.random.seed(3)
    b = photon_tracer.follow(scene, rays[0])
    assert a == b
    # path-length limit (:162-172): KILL at the start of the first step entered beyond it, with the ray as it stood
    closes = (Event.GENERATE, Event.REFLECT, Event.TRANSMIT, Event.EMIT, Event.SCATTER)
    k, long = max(((kk, photon_tracer.follow(scene, r, seed=900 + kk)) for kk, r in enumerate(rays)), key=lambda t: len(t[1]))
    assert len(long) >= 6
    mid = next(i for i in range(len(long) // 2, len(long) - 1) if long[i][1] in closes)
    limit = long[mid][0].travelled * (1 - 1e-9)
    first = next(i for i in range(len(long) - 1) if long[i][1] in closes and long[i][0].travelled > limit)
    cut = photon_tracer.follow(scene, rays[k], maxpathlength=limit, seed=900 + k)
    assert len(cut) == first + 2 and cut[:-1] == long[:first + 1]
    assert cut[-1][1] == Event.KILL and cut[-1][0] == cut[-2][0] and cut[-1][0].travelled > limit
    assert photon_tracer.follow(scene, rays[k], maxpathlength=1e9, seed=900 + k) == long
    # a ray that arrives with a past keeps it
    import dataclasses
    aged = dataclasses.replace(rays[0], travelled=2.5, duration=1e-9)
    h0, h1 = photon_tracer.follow(scene, rays[0], seed=7), photon_tracer.follow(scene, aged, seed=7)
    assert [e for _, e in h0] == [e for _, e in h1] and h1[-1][0].travelled == h0[-1][0].travelled + 2.5


# -- Scene.simulate: the reference's tests of it (tests/test_scene.py:20-182), one by one ---------------------------------
def reference_scene_of_test_scene():
    import functools
    from pvtrace_amd import Box, Light, Material, Node, Scene, Sphere, cone

    world = Node(name="world (air)", geometry=Sphere(radius=50.0, material=Material(refractive_index=1.0)))
    Node(name="sphere (glass)", geometry=Box((10.0, 10.0, 1.0), material=Material(refractive_index=1.5)), parent=world)
    light = Node(name="Light (555nm)", light=Light(direction=functools.partial(cone, np.pi / 16)), parent=world)
    light.rotate(np.radians(60), (1.0, 0.0, 0.0))
    return Scene(world)


def test_scene_simulate_keeps_the_references_contract():
    import os
    import queue as queue_mod

    RAYS = max(16, os.cpu_count())
    scene = reference_scene_of_test_scene()
    for workers in (1, None):                                   # :129-137
        results = scene.simulate(RAYS, workers=workers)
        assert isinstance(results, list) and len(results) == RAYS, "Missing simulation results"
    for history in results:                                     # each a list of (Ray, Event), GENERATE first
        assert history[0][1] == Event.GENERATE and history[-1][1] in (Event.EXIT, Event.KILL)
        assert all(len(item) == 2 for item in history)
    r1, r2 = scene.simulate(RAYS, workers=1, seed=1), scene.simulate(RAYS, workers=1, seed=1)   # :139-143
    assert r1 == r2, "Simulation should be identical"
    with pytest.raises(ValueError):                             # :145-157
        scene.simulate(RAYS, workers=os.cpu_count() if os.cpu_count() > 1 else 2, seed=1)
    assert scene.simulate(RAYS, workers=1, seed=1) != scene.simulate(RAYS, workers=1, seed=2)     # :159-163
    assert scene.simulate(RAYS, workers=os.cpu_count()) != scene.simulate(RAYS, workers=os.cpu_count())   # :165-169
    assert scene.simulate(RAYS) != scene.simulate(RAYS)         # :171-175
    assert scene.simulate(1) != scene.simulate(1)               # :177-181
    # the queue form (:60-89): every event, or only the "end rays", as (pid, index, Ray, Event, metadata); returns the pid
    q = queue_mod.Queue()
    pid = scene.simulate(RAYS, workers=1, seed=5, queue=q)
    items = []
    while not q.empty():
        items.append(q.get())
    direct = scene.simulate(RAYS, workers=1, seed=5)
    assert pid == os.getpid() and len(items) == sum(len(h) for h in direct)
    assert sorted({i[1] for i in items}) == list(range(RAYS)) and items[0][3] == Event.GENERATE and items[0][4] is None
    assert [(i[2], i[3]) for i in items if i[1] == 3] == direct[3]
    q = queue_mod.Queue()
    scene.simulate(RAYS, workers=1, seed=5, queue=q, end_rays=True)
    kept = []
    while not q.empty():
        kept.append(q.get())
    from pvtrace_amd.scene import is_end_ray
    assert 0 < len(kept) <= len(items) and all(is_end_ray(i[3], i[4]) for i in kept)
    assert len(kept) == sum(1 for i in items if is_end_ray(i[3], i[4]))


def test_a_script_written_for_the_reference_runs_unchanged_after_compat_install():
    """After `pvtrace_amd.compat.install()` the package answers to `pvtrace` with the reference's module layout: the body of
    this script is written in the reference's vocabulary only (its README quick start, examples/hello_box.py and the LSC
    notebook's calls: `from pvtrace import *`, `scene.simulate`, `photon_tracer.follow`, `LSC(...).simulate / report`,
    `pvtrace.engine.simulate` with recorders) and runs on the engine, in a fresh interpreter."""
    import os
    import subprocess
    import sys

    script = r"""
import pvtrace_amd.compat; pvtrace_amd.compat.install()
# ---- from here on: a user's script for the reference --------------------------------------------------------
import functools
import numpy as np
from pvtrace import *
from pvtrace.material.utils import cone
from pvtrace.engine import simulate, Recorder, Histogram

world = Node(name="world (air)", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
sphere = Node(name="sphere (glass)", geometry=Sphere(radius=1.0, material=Material(refractive_index=1.5)), parent=world)
sphere.location = (0, 0, 2)
light = Node(name="Light (555nm)", light=Light(direction=functools.partial(cone, np.pi / 8)), parent=world)
scene = Scene(world)
np.random.seed(0)
results = scene.simulate(40, workers=1)
assert len(results) == 40 and all(h[-1][1] == Event.EXIT for h in results)
for ray in scene.emit(10):
    steps = photon_tracer.follow(scene, ray, maxsteps=10)
    assert steps[0][1] == Event.GENERATE
sphere.recorders = [Recorder("entering", event="entering", histograms=[Histogram("angle", 0.0, np.pi / 2, 9)])]
result = simulate(scene, 200000, seed=1, record_every=0)
assert 0.4 < result.recorders["entering"].rays / 200000 < 1.0
# (the README's own spelling of the two imports, and its recorder on the world)
from pvtrace.algorithm import photon_tracer as tracer_again
import pvtrace.engine as engine
assert tracer_again is photon_tracer
world.recorders = [Recorder("escaped", event="exit", histograms=[Histogram("wavelength", 400, 800, 40)])]
result = engine.simulate(scene, 100000)
escaped = result.recorders["escaped"]
assert escaped.rays == 100000 and round(escaped.mean("wavelength"), 1) == 555.0 and result.elapsed > 0

lsc = LSC((5.0, 5.0, 1.0))
lsc.add_solar_cell({"left", "right"})
lsc.simulate(20000)
summary = lsc.summary()
assert 0.0 < float(summary["Optical Efficiency"]) < 1.0
print("ok")
"""
    done = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(__file__)))
    assert done.returncode == 0 and done.stdout.strip().endswith("ok"), (done.stdout[-500:], done.stderr[-2000:])
