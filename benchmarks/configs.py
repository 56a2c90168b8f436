"""The BASELINE.json configurations as scenes, built with the product's own API (nothing from tests/).

* cfg2 / cfg3: ``LSC((5, 5, 1))`` -- the library's LSC builder (reference pvtrace/device/lsc.py:95-219:
  500x500x100 cm world, 5x5x1 cm n = 1.5 slab with Lumogen F Red 305 at 10 cm^-1 peak, qy 1, plus 0.1 cm^-1
  background, point light at (0, 0, 5) facing down, 20-degree cone, 555 nm) carrying the tally set of
  SURVEY.md §8(d) (`engine.instrument.face_recorders`).
* cfg4: reference examples/nested_cylinders.py:21-64 (two glass cylinders, the child rotated and protruding from
  its parent, in a 10 cm air sphere; 30-degree cone from z = -1).
* cfg5: reference examples/006 Coatings.ipynb cell 5 (10x10x1 cm slab, perfect mirror on the x>0, y>0 quadrant of
  the top face, 5x5 rectangular source above it) plus an isotropic 1 cm^-1 scatterer (qy 1) in the slab.
* tiles<k>: the scene-size family: a k x k array of cfg2's slab (5x5x1 cm, Lumogen F Red 305 +
  background, plain Fresnel surfaces, so the reference kernel runs it too) on a 5.5 cm pitch in cfg2's world,
  lit from above by one rectangular source that covers the array, 20-degree cone, 555 nm: k*k + 1 nodes.  The
  reference intersects every node in every step (pvtrace/engine/_kernel.pyx:666-680).
* mesh<s>: the triangle-mesh extension on the reference's hello_world scene, the ball an icosphere of 20 * 4^s faces.
"""
import functools

import numpy as np

from pvtrace_amd import (
    LSC, Absorber, Box, Coating, CoatedSurfaceDelegate, Cylinder, Light, Luminophore, Material, Node, Scatterer,
    Scene, Sphere, Surface, cone, rectangular_mask,
)
from pvtrace_amd.data import lumogen_f_red_305
from pvtrace_amd.engine import Heatmap, Histogram, Recorder
from pvtrace_amd.engine.instrument import face_recorders


def cfg2_lsc():
    lsc = LSC((5.0, 5.0, 1.0))
    scene = lsc.scene
    slab = next(n for n in scene.root.children if n.name == "LSC")
    slab.recorders = face_recorders()
    return scene


def cfg4_nested_cylinders():
    world = Node(name="World", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    a = Node(name="A", parent=world,
             geometry=Cylinder(length=2, radius=0.5, material=Material(refractive_index=1.5)))
    a.translate((0, 0, 2))
    a.rotate(np.pi * 0.2, (0, 1, 0))
    b = Node(name="B", parent=a,
             geometry=Cylinder(length=2.0, radius=0.4, material=Material(refractive_index=1.5)))
    b.rotate(np.pi / 2, (1, 0, 0))
    light = Node(name="Light (555nm)", parent=world,
                 light=Light(direction=functools.partial(cone, np.radians(30)), name="Light (555nm)"))
    light.translate((0, 0, -1))
    a.recorders = [Recorder("A-escaping", event="escaping"), Recorder("A-entering", event="entering")]
    b.recorders = [Recorder("B-escaping", event="escaping"), Recorder("B-entering", event="entering")]
    world.recorders = [Recorder("exit", event="exit", histograms=[Histogram("angle", 0.0, np.pi / 2, 18)])]
    return Scene(world)


def cfg5_coated_slab(scatter=1.0):
    world = Node(name="world (air)", geometry=Box((15.0, 15.0, 15.0), material=Material(refractive_index=1.0)))
    mirror = Coating((0, 0, 1), reflectivity=1.0, region=((0.0, None), (0.0, None), None))
    slab = Node(
        name="box (glass)", parent=world,
        geometry=Box((10.0, 10.0, 1.0),
                     material=Material(refractive_index=1.5, components=[Scatterer(float(scatter), name="Scatterer")],
                                       surface=Surface(delegate=CoatedSurfaceDelegate([mirror])))))
    slab.recorders = face_recorders(wavelength=None) + [
        Recorder("top-reflect-map", event="reflected", facet=(0, 0, 1),
                 histograms=[Heatmap("x", "y", (-5, 5, 20), (-5, 5, 20))])]
    light = Node(name="Light", parent=world,
                 light=Light(position=functools.partial(rectangular_mask, 5, 5), name="Light"))
    light.location = (0, 0, 2)
    light.rotate(np.radians(180), (1, 0, 0))
    return Scene(world)


TILE_PITCH = 5.5


def tiles_lsc(k, recorders="centre"):
    """k x k slabs of the headline LSC in one world (k*k + 1 nodes).  `recorders`: "centre" = the headline's ten
    face recorders on the middle tile; "all" = additionally `escaping` and `lost` on every tile (2 k^2 + 10)."""
    x = np.arange(400, 800)
    world = Node(name="World", geometry=Box((500.0, 500.0, 100.0), material=Material(refractive_index=1.0)))
    middle = (k // 2) * k + k // 2
    for i in range(k * k):
        row, col = divmod(i, k)
        tile = Node(
            name=f"tile-{row}-{col}", parent=world,
            geometry=Box((5.0, 5.0, 1.0), material=Material(
                refractive_index=1.5,
                components=[
                    Luminophore(coefficient=np.column_stack((x, lumogen_f_red_305.absorption(x) * 10.0)),
                                emission=np.column_stack((x, lumogen_f_red_305.emission(x))),
                                quantum_yield=1.0, name=f"Lumogen F Red 305 ({row},{col})"),
                    Absorber(0.1, name=f"Background ({row},{col})"),
                ])))
        tile.location = ((col - 0.5 * (k - 1)) * TILE_PITCH, (row - 0.5 * (k - 1)) * TILE_PITCH, 0.0)
        recs = face_recorders() if i == middle else []
        if recorders == "all":
            recs = recs + [Recorder(f"escaping-{row}-{col}", event="escaping"), Recorder(f"lost-{row}-{col}", event="lost")]
        tile.recorders = recs
    half = 0.5 * k * TILE_PITCH
    light = Node(name="Light", parent=world,
                 light=Light(position=functools.partial(rectangular_mask, half, half),
                             direction=functools.partial(cone, np.radians(20)), name="Light"))
    light.location = (0.0, 0.0, 5.0)
    light.rotate(np.radians(180), (1, 0, 0))
    return Scene(world)


def hello_world():
    """BASELINE configs[0] and the scene of the reference's one published engine figure (README.md:163-170,
    examples/hello_world.py:8-32): a glass ball of radius 1 at (0, 0, 2) in a 10 cm air sphere, pi/8 cone from the origin
    at 555 nm; no recorders -- the README's call keeps every event of every ray instead."""
    world = Node(name="world", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    ball = Node(name="ball-lens", parent=world, geometry=Sphere(radius=1.0, material=Material(refractive_index=1.5)))
    ball.location = (0, 0, 2)
    Node(name="green-laser", parent=world, light=Light(direction=functools.partial(cone, np.pi / 8), name="green-laser"))
    return Scene(world)


def mesh_ball(subdivisions):
    """The reference's hello_world (examples/hello_world.py:8-32: a glass ball of radius 1 at (0, 0, 2) in a 10 cm air
    sphere, pi/8 cone from the origin) with the ball as a TRIANGLE MESH -- an icosphere of 20 * 4^subdivisions faces --
    and whole-surface recorders on it.  Meshes are an extension (the reference engine rejects them); this is the scene
    the mesh walk's numbers are quoted on."""
    from pvtrace_amd import Mesh
    world = Node(name="world", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    ball = Node(name="ball-lens", parent=world,
                geometry=Mesh.icosphere(subdivisions, 1.0, material=Material(refractive_index=1.5)))
    ball.location = (0, 0, 2)
    ball.recorders = [Recorder("ball-entering", event="entering"), Recorder("ball-escaping", event="escaping")]
    Node(name="green-laser", parent=world, light=Light(direction=functools.partial(cone, np.pi / 8), name="green-laser"))
    return Scene(world)


def _mesh_config(subdivisions):
    return dict(build=functools.partial(mesh_ball, subdivisions), emit_method="kT",
                workload=f"triangle-mesh extension: hello_world's glass ball as an icosphere of {20 * 4 ** subdivisions} faces "
                         f"(BVH walk), pi/8 cone @555 nm, 2 recorders, record_every=0")


MESH_SIZES = (3, 5, 7)


def fine_spectra_lsc(points=8001):
    """cfg2's slab with its two spectra tabulated on a fine grid (8001 points: 400-800 nm in 0.05 nm steps -- what a
    measured spectrum straight from a spectrometer file looks like): 260 KB of tables, far beyond a workgroup's LDS."""
    x = np.linspace(400.0, 800.0, points)
    world = Node(name="World", geometry=Box((500.0, 500.0, 100.0), material=Material(refractive_index=1.0)))
    slab = Node(name="LSC", parent=world, geometry=Box((5.0, 5.0, 1.0), material=Material(
        refractive_index=1.5,
        components=[Luminophore(coefficient=np.column_stack((x, lumogen_f_red_305.absorption(x) * 10.0)),
                                emission=np.column_stack((x, lumogen_f_red_305.emission(x))),
                                quantum_yield=1.0, name="Lumogen F Red 305"),
                    Absorber(0.1, name="Background")])))
    slab.recorders = face_recorders()
    light = Node(name="Light", parent=world, light=Light(direction=functools.partial(cone, np.radians(20)), name="Light"))
    light.location = (0.0, 0.0, 5.0)
    light.rotate(np.radians(180), (1, 0, 0))
    return Scene(world)


def _tiles_config(k):
    return dict(build=functools.partial(tiles_lsc, k), emit_method="kT",
                workload=f"scene-size family: {k}x{k} array of cfg2's slab on a 5.5 cm pitch ({k * k + 1} nodes), "
                         f"rectangular source over the array, 20-degree cone @555 nm, 10 recorders on the middle "
                         f"tile, record_every=0")


TILE_SIZES = (1, 2, 3, 4, 6, 8, 11)

CONFIGS = {
    "cfg2": dict(build=cfg2_lsc, emit_method="kT",
                 workload="BASELINE configs[1]: 5x5x1 cm LSC((5,5,1)), Lumogen F Red 305 (10 cm^-1 peak, qy 1) + "
                          "0.1 cm^-1 background, 20-degree cone @555 nm, 10 recorders, record_every=0, "
                          "emit_method=kT, maxsteps=1000"),
    "cfg4": dict(build=cfg4_nested_cylinders, emit_method="kT",
                 workload="BASELINE configs[3]: nested_cylinders (two rotated glass cylinders in a 10 cm air sphere, "
                          "30-degree cone), 5 recorders incl. world exit, record_every=0"),
    "cfg5": dict(build=cfg5_coated_slab, emit_method="kT",
                 workload="BASELINE configs[4]: 10x10x1 cm slab, mirror coating on a quadrant of the top face + "
                          "1 cm^-1 isotropic scatterer, 5x5 cm rectangular source, 11 recorders, record_every=0"),
}
CONFIGS.update({f"tiles{k}": _tiles_config(k) for k in TILE_SIZES})
CONFIGS.update({f"mesh{k}": _mesh_config(k) for k in MESH_SIZES})
CONFIGS["fine"] = dict(build=fine_spectra_lsc, emit_method="kT",
                       workload="cfg2's slab with 8001-point spectra (tables beyond LDS), 20-degree cone @555 nm, 10 recorders")
