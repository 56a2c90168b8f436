"""Developer tool (build container only: needs /root/reference): extended differential fuzz of the reference's
compiled kernel against the oracle's libm mode, beyond the 40 seeds of tests/test_oracle_vs_reference.py.
usage: python tools/cpu_fuzz_reference.py FIRST_SEED COUNT"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle as O
from pvtrace_amd.engine import compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests.fuzz import random_scene
from tests.util import assert_bundles_identical

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + count):
    scene = random_scene(seed)
    compiled = compile_scene(scene)
    pos, dirs, wl, _ = emit_bundle(scene, 600, seed=seed)
    record_every, max_events, maxsteps, emit_method = [(1, 48, 300, 0), (3, 16, 40, 1), (0, 8, 300, 2)][seed % 3]
    ref = O.reference_trace_bundle(compiled, pos, dirs, wl, 77 + seed, maxsteps, max_events, emit_method, 1, record_every)
    mine = O.trace_bundle(compiled, pos, dirs, wl, 77 + seed, maxsteps, max_events, emit_method, 1, record_every,
                          math_mode=O.MATH_LIBM)
    try:
        assert_bundles_identical(mine, ref, what=f"fuzz scene {seed}")
    except AssertionError as e:
        bad += 1
        print("MISMATCH", str(e)[:200], flush=True)
print(f"{count} scenes, reference kernel vs oracle (libm arithmetic): {bad} mismatching")
