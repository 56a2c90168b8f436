"""Developer tool (GPU box): history-mode (record_every=1) kernel time vs bundle size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes
for name, maxev in (("lsc_equivalent", 64), ("hello_world", 16)):
    sc = scenes.ALL_SCENES[name](); c = compile_scene(sc)
    for n in (200_000, 1_000_000, 3_000_000):
        pos, d, wl, _ = emit_bundle(sc, n, seed=5)
        ts = []
        for rep in range(3):
            t = {}
            out = _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, maxev, 0, 1, 1, timing=t)
            ts.append(t["kernel_ms"])
        ev = int(out["counts"].sum())
        print(f"{name} n={n} max_events={maxev}: best {min(ts):.3f} ms  {n/min(ts)/1e3:.0f} M rays/s  {ev/min(ts)/1e3:.0f} M events/s  ({ev/n:.1f} events/ray)", flush=True)
