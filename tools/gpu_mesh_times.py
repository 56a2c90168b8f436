"""Developer tool (GPU box): mesh scenes, tally mode and histories, best kernel time of 5.  PVT_LIB selects the build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pvtrace_amd as pv
from pvtrace_amd.engine import _kernel, compile_scene
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes

def ico(sub):
    sc = scenes.hello_world()
    ball = [n for n in sc.root.children if n.geometry is not None][0]
    ball.geometry = pv.Mesh.icosphere(sub, 1.0, material=ball.geometry.material)
    return sc

for name, sc in (("mesh_lsc", scenes.mesh_lsc()), ("mesh_gem", scenes.mesh_gem()), ("l_prism", scenes.l_prism()),
                 ("ico3", ico(3)), ("ico5", ico(5)), ("ico7", ico(7))):
    c = compile_scene(sc)
    out = [name.ljust(10)]
    for n, rec_every, maxev in ((1_000_000, 0, 16), (200_000, 1, 64)):
        pos, d, wl, _ = emit_bundle(sc, n, seed=5)
        ts = []
        for rep in range(5):
            t = {}
            _kernel.trace_bundle(c, pos, d, wl, 1 + rep, 1000, maxev, 0, 1, rec_every, timing=t)
            ts.append(t["kernel_ms"])
        out.append(f"n={n} rec_every={rec_every}: {min(ts):.3f} ms")
    print("  ".join(out), flush=True)
