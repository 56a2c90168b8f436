"""Triangle meshes on the engine: a faceted glass ball (icosphere) with a hazy interior, lit by a
narrow beam; the same `Mesh` accepts any object with `.vertices` / `.faces` (a trimesh.Trimesh) or
an STL file via `Mesh.from_file`.

    python examples/mesh_gem.py               # needs an MI355X
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pvtrace_amd import Light, Material, Mesh, Node, Scatterer, Scene, Sphere, engine   # noqa: E402
from pvtrace_amd.engine import Heatmap, Recorder                                        # noqa: E402
from pvtrace_amd.light import CircularMask                                              # noqa: E402
from pvtrace_amd.material import Cone, HenyeyGreenstein                                 # noqa: E402

world = Node(name="world", geometry=Sphere(10.0, material=Material(refractive_index=1.0)))
gem = Node(name="gem", parent=world, geometry=Mesh.icosphere(4, 1.0, material=Material(
    refractive_index=1.5, components=[Scatterer(0.5, phase_function=HenyeyGreenstein(0.6), name="haze")])))
gem.location = (0.0, 0.0, 2.0)
gem.recorders = [Recorder("in", event="entering"),
                 Recorder("out", event="escaping", histograms=[Heatmap("x", "y", (-1, 1, 8), (-1, 1, 8))])]
world.recorders = [Recorder("exit", event="exit")]
Node(name="beam", parent=world, light=Light(position=CircularMask(0.3), direction=Cone(0.05), name="beam"))

result = engine.simulate(Scene(world), 1_000_000, seed=1, record_every=0)
print(f"{len(gem.geometry.faces)} faces, {result.num_rays} photons in {result.elapsed * 1e3:.2f} ms")
for name, rec in result.recorders.items():
    print(f"  {name:5s} {rec.rays:8d} rays, {rec.crossings:8d} crossings")
xe, ye, heat = result.recorders["out"].histogram(0)
print("where the light leaves the gem (local x, y):")
print(np.array2string(heat, max_line_width=120))
