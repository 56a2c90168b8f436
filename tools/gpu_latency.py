"""Developer tool (GPU box): kernel time of one launch of the headline scene from 64 to 10^6 photons (launch floor and ramp)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes
sc = scenes.lsc_equivalent(); c = compile_scene(sc)
pos, d, wl, _ = emit_bundle(sc, 1_000_000, seed=5)
out = []
for n in (64, 4096, 262144, 1_000_000):
    best = 1e9
    for rep in range(4):
        t = {}; _kernel.trace_bundle(c, pos[:n], d[:n], wl[:n], 1 + rep, 1000, 128, 0, 1, 0, timing=t); best = min(best, t["kernel_ms"])
    out.append(f"n={n}: {best:.3f} ms")
print(" | ".join(out), flush=True)
