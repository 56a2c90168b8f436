"""The reference's examples/hello_box.py, call for call, on the MI355X engine: a glass box in an air sphere, `scene.simulate`
for a list of ray histories and the ray-by-ray loop with `photon_tracer.follow` -- the two forms the reference calls
equivalent.  Both are traced on the GPU here (a ray is a bundle of one); for more than a few thousand rays use
`engine.simulate(scene, n)` and its recorders instead of Python lists of histories.

    python examples/hello_box.py            # needs an MI355X
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pvtrace_amd import Box, Event, Light, Material, Node, Scene, Sphere, photon_tracer   # noqa: E402
from pvtrace_amd.engine.api import Session                                               # noqa: E402

world = Node(name="world (air)", geometry=Sphere(radius=50.0, material=Material(refractive_index=1.0)))
box = Node(name="box (glass)", geometry=Box((10.0, 10.0, 1.0), material=Material(refractive_index=1.5)), parent=world)
light = Node(name="Light (555nm)", light=Light(), parent=world)
light.rotate(np.radians(60), (1.0, 0.0, 0.0))
scene = Scene(world)

num_rays = 100
start_t = time.time()
results = scene.simulate(num_rays, workers=1)
print(f"Took {time.time() - start_t:.2f}s to trace {num_rays} rays.")
print(f"Got {len(results)} ray histories")

start_t = time.time()
results = []
with Session(scene, emission="host") as session:   # (optional: keeps the scene resident between the calls)
    for ray in scene.emit(100):
        results.append(photon_tracer.follow(scene, ray, session=session))
print(f"Took {time.time() - start_t:.2f}s to trace 100 rays.")
print(f"Got {len(results)} ray histories")
reflected = sum(1 for history in results if history[1][1] == Event.REFLECT)
print(f"{reflected} of them were reflected at their first hit (the light sits INSIDE the glass, as in the reference's example: "
      f"60 degrees is beyond the critical angle of 41.8, so all of them are)")
