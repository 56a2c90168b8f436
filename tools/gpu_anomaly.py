"""Developer tool (GPU box): why one nested_cylinders launch took 4.18 ms.
The kernel time of a single launch is bulk + the longest history x the lone-wave step time; in this scene a photon
can be trapped by total internal reflection inside a loss-free cylinder until `maxsteps` (1000) ends it.  20 launches
per emission mode: kernel ms next to the number of photons the launch killed at maxsteps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from pvtrace_amd import engine
from pvtrace_amd.engine import Recorder
from tests import scenes

scene = scenes.nested_cylinders()
for node in [scene.root] + list(scene.root.children) + [c for n in scene.root.children for c in n.children]:
    if node.geometry is not None:
        node.recorders = list(node.recorders) + [Recorder(f"{node.name}-killed", event="killed")]
engine.simulate(scene, 1000, seed=1, record_every=0)
for emission in ("host", "device"):
    rows = []
    for rep in range(20):
        r = engine.simulate(scene, 1_000_000, seed=100 + rep, emit_seed=200 + rep, emission=emission, record_every=1000)
        killed = sum(rec.rays for name, rec in r.recorders.items() if name.endswith("-killed"))
        longest = int(r.data["counts"].max())
        rows.append((r.kernel_ms, killed, longest))
    print(f"emission={emission}:")
    for ms, killed, longest in rows:
        print(f"   kernel {ms:6.3f} ms   photons killed at maxsteps: {killed}   (longest sampled history: {longest} events)")
    ms = np.array([r[0] for r in rows]); k = np.array([r[1] for r in rows])
    print(f"   launches with a killed photon: {int((k > 0).sum())}/20, their mean kernel time {ms[k > 0].mean() if (k > 0).any() else float('nan'):.3f} ms; "
          f"without: {ms[k == 0].mean() if (k == 0).any() else float('nan'):.3f} ms")
