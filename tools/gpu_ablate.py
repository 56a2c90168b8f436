"""Developer tool (GPU box): what the parts of a step cost in the pipelined steady state -- the bench's stream on
variants of the headline scene: without recorders (no tallies), with the dye's quantum yield at zero (no re-emission),
without the dye (no absorption at all: Fresnel optics only)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from pvtrace_amd.engine import BundlePipeline, compile_scene, native
from pvtrace_amd.engine.emit import emit_bundle
from tests import scenes

def variants():
    yield "headline", scenes.lsc_equivalent()
    yield "no recorders", scenes.lsc_equivalent(recorders=False)
    s = scenes.lsc_equivalent()
    slab = [n for n in s.root.children if n.geometry is not None][0]
    slab.geometry.material.components[0].quantum_yield = 0.0
    yield "quantum yield 0 (no re-emission)", s
    s = scenes.lsc_equivalent()
    slab = [n for n in s.root.children if n.geometry is not None][0]
    slab.geometry.material.components = []
    yield "no components (Fresnel only)", s

dev = torch.device("cuda", 0)
n = 1_000_000
for name, scene in variants():
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
    run(300)
    torch.cuda.synchronize(); t = time.perf_counter(); run(3000); dt = time.perf_counter() - t
    print(f"{name:36s} {3000 * n / dt:.4e} photons/s   {dt / 3000 * 1e3:.4f} ms per 10^6", flush=True)
    ds.close()
