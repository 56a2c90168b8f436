import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from pvtrace_amd import engine
from pvtrace_amd.engine import compile_scene
from pvtrace_amd.engine.emit import EmitterTables
from tests import scenes
scene = scenes.bench_slab(recorders=True)
compiled = compile_scene(scene)
m, total, seed, emit_seed = 600, 2000, 41, 13
pos, dirs, wl = O.emit(EmitterTables(scene), total, emit_seed=emit_seed)
got = list(engine.simulate_stream(scene, total, bundle=m, seed=seed, emit_seed=emit_seed, record_every=0, emission="device"))
traced = 0
for result, upto in got:
    n = upto - traced
    cpu = O.trace_bundle(compiled, pos[traced:upto], dirs[traced:upto], wl[traced:upto], seed + traced, 1000, 16, 0, 1, 0, math_mode=O.MATH_PORTABLE)
    for rep in range(3):
        alone = engine.simulate(scene, n, seed=seed, emit_seed=emit_seed, ray_offset=traced, record_every=0, emission="device")
        print(traced, "set", result.data["rec_distinct"], "alone", alone.data["rec_distinct"], "cpu", cpu["rec_distinct"], flush=True)
    traced = upto
