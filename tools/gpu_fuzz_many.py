"""Developer tool (GPU box): extended differential fuzz of the node-grid path -- random scenes of 8 ... 120 nodes
(tests/fuzz.py: random_many_scene), GPU vs oracle, beyond the seeds the test suite runs.
Usage: python tools/gpu_fuzz_many.py FIRST_SEED COUNT"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from pvtrace_amd.engine import _kernel, compile_scene, native
from pvtrace_amd.engine.emit import EmitterTables, emit_bundle
from tests.fuzz import random_many_scene
from tests.util import assert_bundles_identical

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = grids = odd = 0
for seed in range(first, first + count):
    scene = random_many_scene(seed)
    c = compile_scene(scene)
    plan = native.node_grid_plan(c)
    grids += plan is not None
    odd += plan is not None and plan["odd"]
    record_every, max_events, maxsteps, emit_method = [(1, 48, 300, 0), (3, 16, 40, 1), (0, 8, 300, 2)][seed % 3]
    n = 1500
    emitter = None
    if seed % 2:
        try:
            emitter = EmitterTables(scene)
        except Exception:
            emitter = None
    if emitter is not None:
        pos, dirs, wl = O.emit(emitter, n, emit_seed=seed)
        gpu = _kernel.trace_bundle(c, None, None, n, 9 + seed, maxsteps, max_events, emit_method, 1, record_every,
                                   emitter=emitter, emit_seed=seed)
    else:
        pos, dirs, wl, _ = emit_bundle(scene, n, seed=seed)
        gpu = _kernel.trace_bundle(c, pos, dirs, wl, 9 + seed, maxsteps, max_events, emit_method, 1, record_every)
    cpu = O.trace_bundle(c, pos, dirs, wl, 9 + seed, maxsteps, max_events, emit_method, 8, record_every,
                         math_mode=O.MATH_PORTABLE)
    try:
        assert_bundles_identical(gpu, cpu, sums_rtol=1e-12, what=f"many-node seed {seed}")
    except AssertionError as e:
        bad += 1
        print("MISMATCH", str(e)[:200], flush=True)
print(f"{count} scenes of 8-120 nodes ({grids} through the node grid, {odd} of them with cylinders), {bad} mismatches")
