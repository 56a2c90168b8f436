"""Developer tool (GPU box): kernel time per test scene (tally mode and histories), best of 5.  PVT_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes

names = sys.argv[1:] or ["lsc_equivalent", "nested_cylinders", "hello_world", "coated_slab", "kitchen_sink", "bench_slab", "fresnel_box"]
for name in names:
    sc = scenes.ALL_SCENES[name](); c = compile_scene(sc)
    out = [name.ljust(18)]
    for n, rec_every, maxev in ((2_000_000, 0, 16), (200_000, 1, 64)):
        pos, d, wl, _ = emit_bundle(sc, n, seed=5)
        ts = []
        for rep in range(5):
            t = {}
            _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, maxev, 0, 1, rec_every, timing=t)
            ts.append(t["kernel_ms"])
        out.append(f"n={n} rec_every={rec_every}: best {min(ts):.3f} median {sorted(ts)[2]:.3f} ms")
    print("  ".join(out), flush=True)
