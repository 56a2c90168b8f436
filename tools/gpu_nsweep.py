"""Developer tool (GPU box): kernel time against photons per launch (4 096 ... 4 10^6) of the headline scene, tally mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes
sc = scenes.lsc_equivalent(); c = compile_scene(sc)
pos, d, wl, _ = emit_bundle(sc, 4_000_000, seed=5)
for n in (4096, 16384, 65536, 262144, 524288, 1_000_000, 2_000_000, 4_000_000):
    best = 1e9
    for rep in range(4):
        t = {}; _kernel.trace_bundle(c, pos[:n], d[:n], wl[:n], 1 + rep, 1000, 128, 0, 1, 0, timing=t); best = min(best, t["kernel_ms"])
    print(f"n={n:>8d} {best:7.3f} ms {n/best/1e3:8.1f} M/s", flush=True)
