"""The reference's examples/hello_world.py scene (glass sphere in an air sphere, 22.5-degree cone
light) on the MI355X engine: same scene-building calls, `engine.simulate` instead of the Python
tracer + renderer.

    python examples/hello_world.py            # needs an MI355X
"""
import functools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pvtrace_amd import Light, Material, Node, Scene, Sphere, cone, engine   # noqa: E402
from pvtrace_amd.engine import Histogram, Recorder                          # noqa: E402

world = Node(name="world (air)", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
sphere = Node(name="sphere (glass)", parent=world,
              geometry=Sphere(radius=1.0, material=Material(refractive_index=1.5)))
sphere.location = (0, 0, 2)
Node(name="Light (555nm)", parent=world, light=Light(direction=functools.partial(cone, np.pi / 8)))
sphere.recorders = [Recorder("entered the glass", event="entering",
                             histograms=[Histogram("angle", 0.0, np.pi / 2, 9)]),
                    Recorder("reflected off it", event="reflected")]
scene = Scene(world)

result = engine.simulate(scene, 1_000_000, seed=1, record_every=100_000)
print(f"{result.num_rays} rays in {result.elapsed * 1e3:.2f} ms ({result.num_rays / result.elapsed / 1e6:.0f} M rays/s)")
for name, rec in result.recorders.items():
    print(f"  {name:20s} {rec.rays:8d} rays, mean angle of incidence {np.degrees(rec.mean('angle')):.1f} deg")
print("first recorded history:")
for ray, event, meta in next(iter(result.histories())):
    print(f"  {event.name:9s} at ({ray.position[0]:+.3f}, {ray.position[1]:+.3f}, {ray.position[2]:+.3f})")
