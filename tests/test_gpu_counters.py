"""The kernel's always-on step counters (pvt_scene_counters; KArgs::counters): trips of the photon loop counted on the
device must add up to the reference's loop count (`count`, _kernel.pyx:655) that the referee reports for the same
photons -- an integer identity over a whole bundle, in every kernel variant -- and cfg2's steps per photon is the
number SURVEY.md §8(a) quotes (6.92 loop steps per photon)."""
import numpy as np
import pytest

from oracle import oracle as O
from pvtrace_amd.engine import compile_scene, native
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes

pytestmark = pytest.mark.gpu


def _trace(dscene, rays, n, seed, **kw):
    import torch

    dscene.trace(rays, n, seed, dscene.new_tallies(), **kw)
    torch.cuda.synchronize()


@pytest.mark.parametrize("name", ["lsc_equivalent", "nested_cylinders", "coated_slab", "kitchen_sink", "mesh_gem", "tiles6"])
def test_device_step_count_equals_the_referees_loop_count(name):
    import torch

    scene = (scenes.ALL_SCENES.get(name) or getattr(scenes, name))()
    compiled = compile_scene(scene)
    n, seed = 40_003, 31
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=8)
    O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, 16, 0, 4, 0, math_mode=O.MATH_PORTABLE)
    want = O.last_steps()
    dscene = native.DeviceScene(compiled, device=0)
    try:
        rays = tuple(torch.from_numpy(a).to(torch.device("cuda", 0)) for a in (pos, dirs, wl))
        # one launch that finishes its photons
        assert dscene.counters(reset=True)["steps"] == 0
        _trace(dscene, rays, n, seed)
        c = dscene.counters(reset=True)
        assert c["steps"] == want, (name, c, want)
        assert c["lane_steps"] <= 64 * c["wave_iterations"] and 0.0 < c["lane_utilisation"] <= 1.0
        assert c["waves"] == 4 * dscene.launch_info()["grid"]
        # the same photons as a stream of launches that carry their survivors along, then the closing flush
        edges = [0, 9_000, 9_064, 25_000, n]
        for a, b in zip(edges[:-1], edges[1:]):
            dscene.trace(tuple(t[a:b] for t in rays), b - a, seed, dscene.new_tallies(), ray_offset=a, carry_out=True)
        _trace(dscene, None, 0, 0)
        assert dscene.counters(reset=True)["steps"] == want, name
        # history launches count the same trips (no lazy root, no fused exit there: every trip is run)
        log = dscene.new_event_log(n, 1, 16)
        dscene.trace(rays, n, seed, dscene.new_tallies(), log=log, record_every=1, max_events=16)
        torch.cuda.synchronize()
        O.trace_bundle(compiled, pos, dirs, wl, seed, 1000, 16, 0, 4, 1, math_mode=O.MATH_PORTABLE)
        c = dscene.counters(reset=True)
        assert c["steps"] == O.last_steps() and c["fused_exits"] == 0, name
    finally:
        dscene.close()


def test_cfg2_steps_per_photon_is_the_surveys():
    """BASELINE configs[1] as the bench builds it: SURVEY.md §8(a) measured 6.92 loop steps per photon on the reference
    kernel; the device's counters give the referee's count exactly and that number to within 1 %."""
    import torch

    from benchmarks.configs import cfg2_lsc

    scene = cfg2_lsc()
    compiled = compile_scene(scene)
    n = 200_000
    pos, dirs, wl, _ = emit_bundle(scene, n, seed=1)
    O.trace_bundle(compiled, pos, dirs, wl, 12345, 1000, 128, 0, 8, 0, math_mode=O.MATH_PORTABLE)
    want = O.last_steps()
    dscene = native.DeviceScene(compiled, device=0)
    try:
        rays = tuple(torch.from_numpy(a).to(torch.device("cuda", 0)) for a in (pos, dirs, wl))
        dscene.counters(reset=True)
        _trace(dscene, rays, n, 12345)
        c = dscene.counters()
        assert c["steps"] == want
        assert abs(c["steps"] / n - 6.92) < 0.0692, c["steps"] / n
        assert c["fused_exits"] > 0.5 * n   # most photons leave the slab for good: their empty last step is not run
    finally:
        dscene.close()
