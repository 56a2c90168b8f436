"""Scenes of many nodes on the GPU: the per-lane walk of the node grid (kernel GRID variants) against the CPU
referee, which -- like the reference, pvtrace/engine/_kernel.pyx:666-714 -- intersects EVERY node in every step
and sorts the crossings.  A node the walk skips must never have mattered: every output array identical, on the
tiled LSC arrays of the bench (2 ... 122 nodes), on arrays whose tiles SHARE faces (crossings of two nodes at the
same distance, ordered by node index in the reference), on random scenes of 8 ... 120 boxes, spheres and cylinders
(shared faces, nesting, overlaps, rotations), with and without event logs, with host rays and device emission --
and a negative control: the same walk with its margin turned against it is caught by these very tests."""
import os

import numpy as np
import pytest

from benchmarks.configs import tiles_lsc
from oracle import oracle as O
from pvtrace_amd.engine import _kernel, compile_scene, native
from pvtrace_amd.engine.emit import EmitterTables, emit_bundle
from tests.fuzz import random_many_scene
from tests.util import assert_bundles_identical

pytestmark = pytest.mark.gpu

MODES = [(1, 64, 1000, 0), (0, 128, 60, 1), (5, 24, 1000, 2)]   # record_every, max_events, maxsteps, emit_method


def both(scene, n, mode, seed=31, emit_seed=5, device_emission=False):
    record_every, max_events, maxsteps, emit_method = mode
    compiled = compile_scene(scene)
    if device_emission:
        emitter = EmitterTables(scene)
        pos, dirs, wl = O.emit(emitter, n, emit_seed=emit_seed)
        gpu = _kernel.trace_bundle(compiled, None, None, n, seed, maxsteps, max_events, emit_method, 1, record_every,
                                   emitter=emitter, emit_seed=emit_seed)
    else:
        pos, dirs, wl, _ = emit_bundle(scene, n, seed=emit_seed)
        gpu = _kernel.trace_bundle(compiled, pos, dirs, wl, seed, maxsteps, max_events, emit_method, 1, record_every)
    cpu = O.trace_bundle(compiled, pos, dirs, wl, seed, maxsteps, max_events, emit_method, 4, record_every,
                         math_mode=O.MATH_PORTABLE)
    return compiled, gpu, cpu


@pytest.mark.parametrize("k", [3, 6, 11])
@pytest.mark.parametrize("mode", MODES)
def test_tile_arrays_are_bit_identical_to_the_oracle(k, mode):
    compiled, gpu, cpu = both(tiles_lsc(k, recorders="all" if k <= 6 else "centre"), 4000, mode)
    assert native.node_grid_plan(compiled) is not None          # served by the grid walk
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"tiles{k}")
    if mode[0] == 1:
        assert cpu["counts"].mean() > 6      # photons do wander through the array


def touching_tiles(k=5):
    """tiles_lsc with the pitch equal to the tile: neighbours share a face, so a photon leaving one tile enters
    the next at the very same distance -- the reference picks `hit` / `adjacent` by node index there."""
    from benchmarks import configs

    pitch = configs.TILE_PITCH
    configs.TILE_PITCH = 5.0
    try:
        return tiles_lsc(k, recorders="all")
    finally:
        configs.TILE_PITCH = pitch


@pytest.mark.parametrize("mode", MODES)
def test_tiles_that_share_faces(mode):
    compiled, gpu, cpu = both(touching_tiles(), 4000, mode, device_emission=mode[0] == 0)
    assert native.node_grid_plan(compiled) is not None
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what="touching tiles")
    if mode[0] == 1:
        # the case exists: events whose hit and adjacent nodes are two different tiles (never the world)
        kinds, hit, adj = cpu["kind"].ravel(), cpu["hit"].ravel(), cpu["adjacent"].ravel()
        assert np.sum((hit > 0) & (adj > 0) & (hit != adj)) > 100


@pytest.mark.parametrize("seed", range(40))
def test_random_scenes_of_many_nodes(seed):
    scene = random_many_scene(seed)
    mode = [(1, 48, 300, 0), (3, 16, 40, 1), (0, 8, 300, 2)][seed % 3]
    try:
        EmitterTables(scene)
        device_emission = seed % 2 == 1
    except Exception:
        device_emission = False
    compiled, gpu, cpu = both(scene, 2500, mode, seed=9 + seed, emit_seed=seed, device_emission=device_emission)
    assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"many-node scene {seed}")


def test_most_random_scenes_do_get_a_grid():
    planned = [native.node_grid_plan(compile_scene(random_many_scene(seed))) for seed in range(40)]
    assert sum(p is not None for p in planned) >= 35
    assert any(p is not None and p["odd"] for p in planned) and any(p is not None and not p["odd"] for p in planned)


@pytest.mark.parametrize("what", ["PVT_GRID_DEV_GUARD", "PVT_GRID_DEV_SHRINK"])
def test_negative_control_a_walk_that_skips_nodes_is_caught(what):
    """The walk's early exit and the filing of the nodes rest on one margin (pvt_trace.hip: plan_node_grid).
    Developer switches read at scene creation turn it against the walk -- the exit taken as soon as any two
    crossings are known, or the nodes' boxes filed a centimetre too small -- and the comparisons above must then
    FAIL: the tests have the power to see a skipped node."""
    caught = 0
    for seed in (2, 3, 6, 11):
        scene = random_many_scene(seed)
        os.environ[what] = "1"
        try:
            compiled, gpu, cpu = both(scene, 4000, (1, 48, 300, 0), seed=9 + seed, emit_seed=seed)
        finally:
            del os.environ[what]
        if native.node_grid_plan(compiled) is None:
            continue
        caught += any(not np.array_equal(gpu[key], cpu[key]) for key in ("kind", "hit", "adjacent", "container", "counts"))
        compiled, gpu, cpu = both(scene, 4000, (1, 48, 300, 0), seed=9 + seed, emit_seed=seed)
        assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"scene {seed} without the switch")
    assert caught >= 2, (what, caught)
