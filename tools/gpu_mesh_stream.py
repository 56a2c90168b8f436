"""Developer tool (GPU box): mesh scenes in the bench's pipelined stream (photons carried between launches, three
bundles in flight) -- the sustained rate, which a single launch's time hides behind its longest history.
PVT_LIB selects the build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import pvtrace_amd as pv
from pvtrace_amd.engine import BundlePipeline, compile_scene, native
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes

def ico(sub):
    sc = scenes.hello_world()
    ball = [n for n in sc.root.children if n.geometry is not None][0]
    ball.geometry = pv.Mesh.icosphere(sub, 1.0, material=ball.geometry.material)
    return sc

dev = torch.device("cuda", 0)
n = 1_000_000
names = sys.argv[1:] or ["hello_world", "mesh_lsc", "mesh_gem", "l_prism", "ico3", "ico5", "ico7"]
for name in names:
    scene = ico(int(name[3:])) if name.startswith("ico") else getattr(scenes, name)()
    c = compile_scene(scene)
    sets = []
    for b in range(4):
        p, d, w, _ = emit_bundle(scene, n, seed=10 + b)
        sets.append(tuple(torch.from_numpy(a).to(dev) for a in (p, d, w)))
    ds = native.DeviceScene(c, device=0)
    pipe = BundlePipeline(ds, depth=3)
    def run(steps):
        for k in range(steps):
            pipe.submit(sets[k % 4], n, seed=1 + k * n, timed=False, closing=k >= steps - 3, tail=k == steps - 1)
        pipe.reduce_totals(); pipe.synchronize()
    steps = int(os.environ.get("PVT_STREAM_STEPS", "300"))
    run(max(3, steps // 10))
    torch.cuda.synchronize(); t = time.perf_counter(); run(steps); dt = time.perf_counter() - t
    print(f"{name:12s} {steps * n / dt:.4e} photons/s   {dt / steps * 1e3:.4f} ms per 10^6", flush=True)
    ds.close()
