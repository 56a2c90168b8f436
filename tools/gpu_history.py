"""Developer tool (GPU box): kernel time of history-keeping launches (record_every=1) on the headline scene and hello_world."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes
for name, n, maxev in (("lsc_equivalent", 200_000, 128), ("hello_world", 1_000_000, 128), ("bench_slab", 200_000, 256)):
    sc = scenes.ALL_SCENES[name]() if name in scenes.ALL_SCENES else getattr(scenes, name)()
    c = compile_scene(sc)
    pos, d, wl, _ = emit_bundle(sc, n, seed=5)
    best = 1e9
    for rep in range(4):
        t = {}
        _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, maxev, 0, 1, 1, timing=t)
        best = min(best, t["kernel_ms"])
    print(f"{name}: {n} rays with full histories (max_events={maxev}): kernel {best:.3f} ms = {n / best / 1e3:.1f} M rays/s", flush=True)
