"""Developer tool (GPU box): kernel-time probe of the trace kernel on the headline scene
and a few others; env PVT_BLOCKS_PER_CU is honoured by the library."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import EmitterTables, emit_bundle
from tests import scenes

def probe(name, scene, n, reps=5, device_emit=False):
    c = compile_scene(scene)
    tab = EmitterTables(scene) if device_emit else None
    if not device_emit:
        pos, d, wl, _ = emit_bundle(scene, n, seed=5)
    best = 1e9
    for rep in range(reps):
        t = {}
        if device_emit:
            _kernel.trace_bundle(c, None, None, n, 1 + rep, 1000, 128, 0, 1, 0, emitter=tab, emit_seed=3, timing=t)
        else:
            _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, 128, 0, 1, 0, timing=t)
        best = min(best, t["kernel_ms"])
    print(f"{name:18s} n={n:>9d} emit={'dev ' if device_emit else 'host'} best {best:8.3f} ms  {n / best / 1e3:9.1f} M photons/s", flush=True)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "lsc"
    if which == "lsc":
        for n in (1_000_000, 4_000_000):
            probe("lsc_equivalent", scenes.lsc_equivalent(), n)
    elif which == "one":
        probe("lsc_equivalent", scenes.lsc_equivalent(), 4_000_000, reps=4)
    elif which == "mesh":
        import pvtrace_amd as pv
        probe("mesh_lsc (12 tri)", scenes.mesh_lsc(), 1_000_000)
        probe("mesh_gem (400 tri)", scenes.mesh_gem(), 1_000_000)
        for sub in (3, 5, 7):   # 1280, 20480, 327680 faces
            sc = scenes.hello_world()
            ball = [n for n in sc.root.children if n.geometry is not None][0]
            ball.geometry = pv.Mesh.icosphere(sub, 1.0, material=ball.geometry.material)
            probe(f"hello_world ico{sub}", sc, 1_000_000)
        probe("hello_world", scenes.hello_world(), 1_000_000)
    else:
        probe("lsc_equivalent", scenes.lsc_equivalent(), 1_000_000)
        probe("lsc_equivalent", scenes.lsc_equivalent(), 10_000_000, reps=3, device_emit=True)
        probe("hello_world", scenes.hello_world(), 1_000_000)
        probe("nested_cylinders", scenes.nested_cylinders(), 10_000_000, reps=3, device_emit=True)
        probe("coated_slab", scenes.coated_slab(), 10_000_000, reps=3, device_emit=True)
        probe("bench_slab", scenes.bench_slab(recorders=True), 2_000_000)
        probe("kitchen_sink", scenes.kitchen_sink(), 2_000_000)
