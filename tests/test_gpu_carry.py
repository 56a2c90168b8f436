"""Photons carried from launch to launch of a stream of bundles (PVT_FLAG_CARRY_OUT): a launch hands the photons
still alive when it runs out of new rays to the next launch on its HIP stream.  Which launch finishes a photon
must not change its history: totals over the job equal the totals of launches that each finish their own."""
import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd.engine import BundlePipeline, compile_scene, native
from pvtrace_amd.engine.emit import EmitterTables, emit_bundle
from tests import scenes

pytestmark = pytest.mark.gpu
INT_KEYS = ("rec_distinct", "rec_crossings", "rec_bins")


def _host(t, compiled):
    nrec, pad = int(compiled.rec_node.shape[0]), max(int(compiled.rec_node.shape[0]), 1)
    ints = t["_ints"].cpu().numpy()
    return {"rec_distinct": ints[:nrec], "rec_crossings": ints[pad:pad + nrec],
            "rec_bins": ints[2 * pad:2 * pad + int(compiled.total_bins)],
            "rec_sums": t["_sums"].cpu().numpy()[: nrec * 8].reshape(nrec, 4, 2)}


@pytest.mark.parametrize("name", ["lsc_equivalent", "nested_cylinders", "kitchen_sink", "coated_slab"])
def test_parked_photons_are_finished_by_the_next_launch_bit_for_bit(name):
    import torch

    scene = scenes.ALL_SCENES[name]()
    compiled = compile_scene(scene)
    n, seed = 30_011, 77
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=4)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, 16, 0, 4, 0, math_mode=O.MATH_PORTABLE)
    dscene = native.DeviceScene(compiled, device=0)
    try:
        dev = torch.device("cuda", 0)
        rays = tuple(torch.from_numpy(a).to(dev) for a in (pos, dirs, wl))
        tallies = dscene.new_tallies()
        edges = [0, 9000, 9000 + 64, 21_000, n]           # four launches, the second a single chunk
        for a, b in zip(edges[:-1], edges[1:]):
            part = tuple(t[a:b] for t in rays)
            dscene.trace(part, b - a, seed, tallies, ray_offset=a, carry_out=True)
        torch.cuda.synchronize()
        assert dscene.carry_pending()
        partial = _host(tallies, compiled)
        # something was left for later (every scene here has histories of several steps) ...
        assert partial["rec_crossings"].sum() < cpu["rec_crossings"].sum()
        # ... a history launch cannot take it over ...
        log = dscene.new_event_log(64, 1, 16)
        with pytest.raises(ValueError, match="parked"):
            dscene.trace(tuple(t[:64] for t in rays), 64, seed, dscene.new_tallies(), log=log, record_every=1, max_events=16)
        # ... a launch without new rays finishes it (array-input scene: no emitter, no rays)
        dscene.trace(None, 0, 0, tallies)
        torch.cuda.synchronize()
        assert not dscene.carry_pending()
        got = _host(tallies, compiled)
        for key in INT_KEYS:
            assert np.array_equal(got[key], cpu[key]), (name, key)
        assert np.allclose(got["rec_sums"], cpu["rec_sums"], rtol=1e-11)
        # nothing waiting: a launch of zero rays is a no-op
        dscene.trace(None, 0, 0, tallies)
        torch.cuda.synchronize()
        assert np.array_equal(_host(tallies, compiled)["rec_distinct"], cpu["rec_distinct"])
    finally:
        dscene.close()


def test_pipeline_totals_do_not_depend_on_carrying():
    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    dscene = native.DeviceScene(compiled, device=0, emitter=EmitterTables(scene))
    try:
        results = {}
        for carry in (False, True):
            pipe = BundlePipeline(dscene, depth=3, carry=carry)
            at = 0
            for k, m in enumerate((50_000, 70_001, 64, 30_000, 1, 120_000, 65_000)):
                pipe.submit(None, m, seed=5, ray_offset=at, emit_seed=6, tail=(k == 6))
                at += m
            first = pipe.totals_host()
            # a second job on the same pipeline must start clean: photons parked by warm-up style launches
            # (no tail launch at all here) are finished by reset_totals, not tallied into what follows
            pipe.reset_totals()
            pipe.submit(None, 40_000, seed=5, ray_offset=at, emit_seed=6)
            pipe.reset_totals()
            for k in range(4):
                pipe.submit(None, 25_000, seed=9, ray_offset=k * 25_000, emit_seed=6)
            second = pipe.totals_host()
            results[carry] = (first, second)
        for a, b in zip(results[False], results[True]):
            for key in INT_KEYS:
                assert np.array_equal(a[key], b[key]), key
            assert np.allclose(a["rec_sums"], b["rec_sums"], rtol=1e-11)
        n_first = 50_000 + 70_001 + 64 + 30_000 + 1 + 120_000 + 65_000
        names = list(compiled.recorder_names)
        first, second = results[True]
        assert first["rec_distinct"][names.index("entering")] + first["rec_distinct"][names.index("reflected")] == n_first
        assert second["rec_distinct"][names.index("entering")] + second["rec_distinct"][names.index("reflected")] == 100_000
    finally:
        dscene.close()


def test_a_narrower_launch_still_resumes_every_parked_photon():
    """a launch of 4 workgroups per CU parks up to 64 photons per wave; its successor with 1 workgroup per
    CU and a handful of new rays has a quarter of the waves.  Every parked photon must still be resumed (the library
    widens the resuming launch's grid to the width of the one that parked)."""
    import torch

    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    n, seed = 600_000, 3
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=12)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, 16, 0, 8, 0, math_mode=O.MATH_PORTABLE)
    dscene = native.DeviceScene(compiled, device=0)
    try:
        dev = torch.device("cuda", 0)
        rays = tuple(torch.from_numpy(a).to(dev) for a in (pos, dirs, wl))
        tallies = dscene.new_tallies()
        cut = n - 100
        dscene.trace(tuple(t[:cut] for t in rays), cut, seed, tallies, carry_out=True, workgroups_per_cu=4)
        wide = dscene.launch_info()["grid"]
        dscene.trace(tuple(t[cut:] for t in rays), n - cut, seed, tallies, ray_offset=cut, carry_out=False, workgroups_per_cu=1)
        assert dscene.launch_info()["grid"] >= wide
        torch.cuda.synchronize()
        got = _host(tallies, compiled)
        for key in INT_KEYS:
            assert np.array_equal(got[key], cpu[key]), key
    finally:
        dscene.close()


def test_parked_photons_keep_their_rules_or_are_refused():
    """The launch that resumes parked photons must trace by the rules of the launch that parked them (maxsteps,
    emit_method): anything else is refused, in the C ABI and in BundlePipeline.submit; an abandoned job's photons
    can be dropped (pvt_scene_carry_discard) and are dropped when a new pipeline takes over the stream."""
    import torch

    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    dscene = native.DeviceScene(compiled, device=0, emitter=EmitterTables(scene))
    try:
        tallies = dscene.new_tallies()
        dscene.trace(None, 50_000, 1, tallies, emit_seed=2, carry_out=True, maxsteps=1000, emit_method=0)
        assert dscene.carry_pending()
        with pytest.raises(ValueError, match="maxsteps"):
            dscene.trace(None, 50_000, 1, tallies, emit_seed=2, ray_offset=50_000, maxsteps=500, emit_method=0)
        with pytest.raises(ValueError, match="emit_method"):
            dscene.trace(None, 0, 0, tallies, maxsteps=1000, emit_method=2)
        assert dscene.carry_pending()                 # a refused launch changes nothing
        dscene.carry_discard()
        assert not dscene.carry_pending()
        dscene.trace(None, 0, 0, tallies, maxsteps=500, emit_method=2)      # nothing waiting: any rules, a no-op
        torch.cuda.synchronize()

        # the pipeline says the same in Python, before anything is enqueued
        pipe = BundlePipeline(dscene, depth=2)
        pipe.submit(None, 40_000, seed=5, emit_seed=6, maxsteps=1000)
        pipe.submit(None, 40_000, seed=5, ray_offset=40_000, emit_seed=6, maxsteps=1000)
        with pytest.raises(ValueError, match="parts of one job"):
            pipe.submit(None, 40_000, seed=5, ray_offset=80_000, emit_seed=6, maxsteps=200)
        streams = [s.cuda_stream for s in pipe.streams]
        assert any(dscene.carry_pending(s) for s in streams)
        # ... and an abandoned pipeline does not leak its photons into the next job on the scene
        del pipe
        assert not any(dscene.carry_pending(s) for s in streams)
        fresh = BundlePipeline(dscene, depth=2)
        for k in range(4):
            fresh.submit(None, 25_000, seed=9, ray_offset=k * 25_000, emit_seed=6)
        got = fresh.totals_host()
        names = list(compiled.recorder_names)
        assert got["rec_distinct"][names.index("entering")] + got["rec_distinct"][names.index("reflected")] == 100_000
    finally:
        dscene.close()


def test_host_buffer_entry_ignores_the_carry_flag():
    """pvt_trace_bundle creates a scene, traces once and destroys it: nothing could ever resume parked photons, so the
    flag is masked there and the tallies are complete."""
    from pvtrace_amd.engine import _kernel

    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, 20_000, seed=3)
    plain = _kernel.trace_bundle(compiled, pos, dirs, wl, 11, 1000, 16, 0, 1, 0)
    flagged = _kernel.trace_bundle(compiled, pos, dirs, wl, 11, 1000, 16, 0, 1, 0, flags=native.FLAG_CARRY_OUT)
    for key in INT_KEYS:
        assert np.array_equal(plain[key], flagged[key]), key


def test_a_stream_that_carries_a_live_job_is_not_handed_to_another_pipeline():
    """torch reuses stream handles (a pool per device) and a resident scene outlives pipelines: a SECOND live pipeline
    that were given a handle an older one still uses would drop that job's parked photons.  The scene keeps
    the ownership: a live owner refuses the claim, a dead or closed one hands the handle over with nothing parked,
    and a scene with live pipelines is not trimmed under them."""
    scene = scenes.lsc_equivalent()
    compiled = compile_scene(scene)
    dscene = native.DeviceScene(compiled, device=0, emitter=EmitterTables(scene))
    try:
        first = BundlePipeline(dscene, depth=2)
        first.submit(None, 40_000, seed=5, emit_seed=6)
        first.submit(None, 40_000, seed=5, ray_offset=40_000, emit_seed=6)
        handles = [s.cuda_stream for s in first.streams]
        assert any(dscene.carry_pending(h) for h in handles)

        class Other:   # (stands for a second pipeline that torch happened to give the same stream)
            pass

        other = Other()
        with pytest.raises(RuntimeError, match="another live BundlePipeline"):
            dscene.claim_stream(other, handles[0])
        with pytest.raises(RuntimeError, match="still hold streams"):
            dscene.trim()
        assert any(dscene.carry_pending(h) for h in handles)      # nothing was dropped: the first job is intact ...
        first.submit(None, 20_000, seed=5, ray_offset=80_000, emit_seed=6, closing=True)
        first.submit(None, 0, seed=5, ray_offset=100_000, emit_seed=6, closing=True)
        got = first.totals_host()
        names = list(compiled.recorder_names)
        assert got["rec_distinct"][names.index("entering")] + got["rec_distinct"][names.index("reflected")] == 100_000
        # ... and once it is closed its handles are free again, with nothing parked on them
        first.submit(None, 30_000, seed=7, emit_seed=6)
        first.close()
        assert not dscene.stream_owners()
        dscene.claim_stream(other, handles[0])
        assert not dscene.carry_pending(handles[0])
        dscene.release_stream(other, handles[0])
        dscene.trim()
    finally:
        dscene.close()
