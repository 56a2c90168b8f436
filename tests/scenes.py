"""Scene library shared by the tests, the golden-fixture generator and bench.py.

Each builder returns a fresh `pvtrace_amd.Scene`.  The first five are the
BASELINE.json configs; the rest exercise branches those do
not reach (every phase function, scatterer/reactor, lifetimes, Null surfaces,
rotated children, touching boxes, heatmap recorders).
"""
import functools

import numpy as np

from pvtrace_amd import (
    Absorber, Box, Coating, CoatedSurfaceDelegate, Cylinder, Light, Luminophore, Material, Mesh, Node,
    NullSurfaceDelegate, Reactor, Scatterer, Scene, Sphere, Surface, cone, isotropic, lambertian,
    rectangular_mask,
)
from pvtrace_amd.data import lumogen_f_red_305
from pvtrace_amd.engine import Heatmap, Histogram, Recorder
from pvtrace_amd.light import (
    CircularMask, ConstantWavelengthMask, CubeMask, SpectrumWavelengthMask,
)
from pvtrace_amd.material import Cone, Distribution, HenyeyGreenstein, gaussian

FACES = {
    "top": (0, 0, 1), "bottom": (0, 0, -1), "right": (1, 0, 0), "left": (-1, 0, 0),
    "far": (0, 1, 0), "near": (0, -1, 0),
}


def face_recorders(prefix="", hist=True):
    """escaping x 6 facets (+80-bin wavelength histogram), lost, entering, reflected, killed."""
    recs = []
    for label, normal in FACES.items():
        hs = [Histogram("wavelength", 400, 800, 80)] if hist else []
        recs.append(Recorder(f"{prefix}{label}", event="escaping", facet=normal, histograms=hs))
    recs += [
        Recorder(f"{prefix}lost", event="lost"),
        Recorder(f"{prefix}entering", event="entering"),
        Recorder(f"{prefix}reflected", event="reflected"),
        Recorder(f"{prefix}killed", event="killed"),
    ]
    return recs


# -- config 1 -----------------------------------------------------------------
def hello_world():
    """examples/hello_world.py:8-32: glass ball in an air sphere, pi/8 cone."""
    world = Node(name="world", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    ball = Node(name="ball-lens", geometry=Sphere(radius=1.0, material=Material(refractive_index=1.5)),
                parent=world)
    ball.location = (0, 0, 2)
    Node(name="green-laser",
         light=Light(direction=functools.partial(cone, np.pi / 8), name="green-laser"),
         parent=world)
    return Scene(world)


# -- config 2 / 3 ---------------------------------------------------------------
def lsc_equivalent(recorders=True, size=(5.0, 5.0, 1.0)):
    """The plain-Fresnel equivalent of LSC((5,5,1)) (device/lsc.py:115-219):
    world box 100x, slab n=1.5 with Lumogen F Red (10 cm^-1 peak, qy 1) + 0.1 cm^-1
    background, point light at (0,0,5) flipped, 20-degree cone, 555 nm."""
    l, w, d = size
    x = np.arange(400, 800)
    world = Node(name="World", geometry=Box((l * 100, w * 100, d * 100),
                                            material=Material(refractive_index=1.0)))
    Node(
        name="LSC",
        geometry=Box(
            (l, w, d),
            material=Material(
                refractive_index=1.5,
                components=[
                    Luminophore(
                        coefficient=np.column_stack((x, lumogen_f_red_305.absorption(x) * 10.0)),
                        emission=np.column_stack((x, lumogen_f_red_305.emission(x))),
                        quantum_yield=1.0, name="Lumogen F Red 305"),
                    Absorber(0.1, name="Background"),
                ],
            ),
        ),
        parent=world,
        recorders=face_recorders() if recorders else None,
    )
    light = Node(name="Light", parent=world,
                 light=Light(direction=functools.partial(cone, np.radians(20)), name="Light"))
    light.location = (0.0, 0.0, d * 5)
    light.rotate(np.radians(180), (1, 0, 0))
    return Scene(world)


# -- config 4 -------------------------------------------------------------------
def nested_cylinders(recorders=True):
    """examples/nested_cylinders.py:21-64: two glass cylinders, the child rotated
    and protruding from its parent, 30-degree cone from z=-1."""
    world = Node(name="World", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    a = Node(name="A", geometry=Cylinder(length=2, radius=0.5, material=Material(refractive_index=1.5)),
             parent=world)
    a.translate((0, 0, 2))
    a.rotate(np.pi * 0.2, (0, 1, 0))
    b = Node(name="B", geometry=Cylinder(length=2.0, radius=0.4, material=Material(refractive_index=1.5)),
             parent=a)
    b.rotate(np.pi / 2, (1, 0, 0))
    light = Node(name="Light (555nm)", parent=world,
                 light=Light(direction=functools.partial(cone, np.radians(30)), name="Light (555nm)"))
    light.translate((0, 0, -1))
    if recorders:
        a.recorders = [Recorder("A-escaping", event="escaping"), Recorder("A-entering", event="entering")]
        b.recorders = [Recorder("B-escaping", event="escaping"), Recorder("B-entering", event="entering")]
        world.recorders = [Recorder("exit", event="exit",
                                    histograms=[Histogram("angle", 0.0, np.pi / 2, 18)])]
    return Scene(world)


# -- config 5 -------------------------------------------------------------------
def coated_slab(recorders=True, scatter=1.0):
    """examples/006 Coatings.ipynb cell 5: 10x10x1 slab with a perfect mirror on the
    x>0, y>0 quadrant of the top face, rectangular 5x5 source above it, plus an
    isotropic Scatterer of `scatter` cm^-1 (qy 1) in the slab (BASELINE config 5)."""
    world = Node(name="world (air)", geometry=Box((15.0, 15.0, 15.0), material=Material(refractive_index=1.0)))
    mirror = Coating((0, 0, 1), reflectivity=1.0, region=((0.0, None), (0.0, None), None))
    comps = [Scatterer(float(scatter), name="Scatterer")] if scatter else []
    slab = Node(
        name="box (glass)",
        geometry=Box((10.0, 10.0, 1.0),
                     material=Material(refractive_index=1.5, components=comps,
                                       surface=Surface(delegate=CoatedSurfaceDelegate([mirror])))),
        parent=world,
    )
    if recorders:
        slab.recorders = face_recorders(hist=False) + [
            Recorder("top-reflect-map", event="reflected", facet=(0, 0, 1),
                     histograms=[Heatmap("x", "y", (-5, 5, 20), (-5, 5, 20))])]
    light = Node(name="Light", parent=world,
                 light=Light(position=functools.partial(rectangular_mask, 5, 5), name="Light"))
    light.location = (0, 0, 2)
    light.rotate(np.radians(180), (1, 0, 0))
    return Scene(world)


# -- reference test / benchmark scenes ---------------------------------------------
def fresnel_box():
    """tests/test_engine.py:36-54: glass cube at z=2, pi/16 cone."""
    world = Node(name="world", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    box = Node(name="box", geometry=Box((1.0, 1.0, 1.0), material=Material(refractive_index=1.5)),
               parent=world)
    box.location = (0.0, 0.0, 2.0)
    Node(name="light", light=Light(direction=functools.partial(cone, np.pi / 16)), parent=world)
    return Scene(world)


def bench_slab(recorders=False):
    """benchmarks/benchmark_engine.py:26-55 == tests/test_engine.py:57-93:
    Gaussian dye qy 0.9 + 0.3 cm^-1 background, collimated light from below."""
    x = np.linspace(300.0, 1000.0, 200)
    absorption = np.column_stack((x, 5.0 * gaussian(x, 1.0, 480.0, 40.0)))
    emission = np.column_stack((x, gaussian(x, 1.0, 600.0, 40.0)))
    world = Node(name="world", geometry=Sphere(radius=10.0, material=Material(refractive_index=1.0)))
    slab = Node(
        name="slab",
        geometry=Box((5.0, 5.0, 1.0), material=Material(
            refractive_index=1.5,
            components=[
                Luminophore(coefficient=absorption, emission=emission, quantum_yield=0.9, name="dye"),
                Absorber(coefficient=0.3, name="background"),
            ])),
        parent=world,
    )
    light = Node(name="light", light=Light(), parent=world)
    light.location = (0.0, 0.0, -3.0)
    if recorders:
        slab.recorders = [
            Recorder("entering", event="entering", histograms=[Histogram("wavelength", 400, 900, 50)]),
            Recorder("top", event="escaping", facet=(0, 0, 1),
                     histograms=[Heatmap("x", "y", (-2.5, 2.5, 20), (-2.5, 2.5, 20))]),
            Recorder("lost", event="lost"),
        ]
        world.recorders = [Recorder("exit", event="exit",
                                    histograms=[Histogram("angle", 0.0, np.pi / 2, 18)])]
    return Scene(world)


# -- branch coverage --------------------------------------------------------------
def kitchen_sink():
    """Every component type, phase function, lifetime and surface tag at once:
    a rotated box with HG scatterer + lossy luminophore with lifetimes, a sphere
    with a Reactor and cone-phase scatterer, a Null-surface cylinder probe, two
    lights (spectrum + circular mask, isotropic + cube mask) and 2-D recorders."""
    x = np.linspace(350.0, 900.0, 111)
    world = Node(name="world", geometry=Box((40.0, 40.0, 40.0), material=Material(
        refractive_index=1.0, components=[Absorber(0.002, name="air-haze")])))
    slab = Node(
        name="slab",
        geometry=Box((6.0, 4.0, 1.5), material=Material(
            refractive_index=1.49,
            components=[
                Luminophore(np.column_stack((x, 2.5 * gaussian(x, 1.0, 520.0, 45.0))),
                            emission=np.column_stack((x, gaussian(x, 1.0, 640.0, 35.0))),
                            quantum_yield=0.85, tau_rad=6e-9, tau_nr=2e-9,
                            phase_function=None, name="dye"),
                Scatterer(0.4, phase_function=HenyeyGreenstein(0.7), quantum_yield=0.95, name="hg"),
                Absorber(np.column_stack((x, 0.05 + 0.0005 * (x - 350.0))), tau_nr=1e-9, name="host"),
            ])),
        parent=world,
    )
    slab.translate((0.5, -0.3, 1.0))
    slab.rotate(0.35, (1.0, 0.4, 0.2))
    ball = Node(
        name="ball",
        geometry=Sphere(1.2, material=Material(
            refractive_index=1.33,
            components=[Reactor(0.6, name="reactor"),
                        Scatterer(0.8, phase_function=Cone(0.5), name="cone-scatter")])),
        parent=world,
    )
    ball.location = (-4.0, 1.0, -2.0)
    probe = Node(
        name="probe",
        geometry=Cylinder(3.0, 0.8, material=Material(
            refractive_index=1.0, surface=Surface(delegate=NullSurfaceDelegate()))),
        parent=world,
    )
    probe.location = (3.0, 3.0, -3.0)
    probe.rotate(1.1, (0.0, 1.0, 0.3))
    inner = Node(
        name="inner",
        geometry=Cylinder(1.0, 0.3, material=Material(
            refractive_index=1.7, components=[Absorber(1.5, name="core")])),
        parent=slab,
    )
    inner.rotate(np.pi / 3, (0, 1, 0))
    spectrum = Distribution(x, gaussian(x, 1.0, 500.0, 60.0))
    l1 = Node(name="lamp", parent=world, light=Light(
        wavelength=SpectrumWavelengthMask(spectrum), position=CircularMask(1.5),
        direction=Cone(0.3), name="lamp"))
    l1.location = (0.0, 0.0, 8.0)
    l1.rotate(np.pi, (1, 0, 0))
    l2 = Node(name="glow", parent=world, light=Light(
        wavelength=ConstantWavelengthMask(480.0), position=CubeMask(0.5, 0.5, 0.5),
        direction=isotropic, name="glow"))
    l2.location = (-3.0, -2.0, 0.5)
    slab.recorders = [
        Recorder("slab-in", event="entering", histograms=[
            Histogram("wavelength", 350, 900, 55), Heatmap("x", "y", (-3, 3, 12), (-2, 2, 8))]),
        Recorder("slab-out-top", event="escaping", facet=tuple(
            np.asarray(slab.transformation_to(world))[:3, :3] @ np.array([0.0, 0.0, 1.0])),
            histograms=[Histogram("angle", 0, np.pi / 2, 9), Histogram("duration", 0, 5e-8, 25)]),
        Recorder("slab-lost", event="lost", histograms=[Histogram("pathlength", 0, 40, 20),
                                                        Histogram("z", -0.75, 0.75, 6)]),
        Recorder("slab-reflected", event="reflected"),
    ]
    ball.recorders = [Recorder("ball-reacted", event="reacted"),
                      Recorder("ball-entering", event="entering")]
    probe.recorders = [Recorder("probe-cross", event="entering"),
                       Recorder("probe-leave", event="escaping")]
    inner.recorders = [Recorder("core-lost", event="lost")]
    world.recorders = [Recorder("world-exit", event="exit",
                                histograms=[Heatmap("angle", "wavelength", (0, np.pi / 2, 6), (350, 900, 11))]),
                       Recorder("world-lost", event="lost"), Recorder("world-killed", event="killed")]
    return Scene(world)


def touching_boxes():
    """Two glass cubes sharing a face (reference tests/test_refractored_tracer.py:253-299)
    inside a box world; a slightly divergent beam runs through both."""
    world = Node(name="world", geometry=Box((10.0, 10.0, 10.0), material=Material(refractive_index=1.0)))
    a = Node(name="a", geometry=Box((1.0, 1.0, 1.0), material=Material(refractive_index=1.5)), parent=world)
    a.location = (0.0, 0.0, 0.5)
    b = Node(name="b", geometry=Box((1.0, 1.0, 1.0), material=Material(refractive_index=1.5)), parent=world)
    b.location = (0.0, 0.0, 1.5)
    light = Node(name="light", light=Light(direction=Cone(0.1)), parent=world)
    light.location = (0.0, 0.0, -1.0)
    for node in (a, b):
        node.recorders = [Recorder(f"{node.name}-in", event="entering"),
                          Recorder(f"{node.name}-out", event="escaping")]
    return Scene(world)


def trapped_light():
    """Emitter INSIDE a lossless high-index sphere: total internal reflection forever,
    so photons die by the maxsteps / event-budget KILL paths."""
    world = Node(name="world", geometry=Sphere(10.0, material=Material(refractive_index=1.0)))
    Node(name="orb", geometry=Sphere(1.0, material=Material(refractive_index=2.4)), parent=world,
         recorders=[Recorder("orb-killed", event="killed"), Recorder("orb-out", event="escaping")])
    light = Node(name="light", light=Light(direction=isotropic), parent=world)
    light.location = (0.0, 0.0, 0.9)
    world.recorders = [Recorder("exit", event="exit")]
    return Scene(world)


def lambertian_sheet():
    """A Lambertian perfect mirror sheet (AirGapMirror-style coating on every face)
    lit by a lambertian source from above."""
    world = Node(name="world", geometry=Box((20.0, 20.0, 20.0), material=Material(refractive_index=1.0)))
    coats = [Coating(n, reflectivity=1.0, reflection="lambertian") for n in FACES.values()]
    sheet = Node(name="sheet", parent=world, geometry=Box(
        (4.0, 4.0, 0.25), material=Material(refractive_index=1.0,
                                            surface=Surface(delegate=CoatedSurfaceDelegate(coats)))))
    sheet.rotate(0.4, (0.0, 1.0, 0.0))
    sheet.recorders = [Recorder("sheet-reflected", event="reflected",
                                histograms=[Histogram("angle", 0, np.pi / 2, 9)])]
    light = Node(name="light", parent=world, light=Light(direction=lambertian, name="light"))
    light.location = (0.0, 0.0, 3.0)
    light.rotate(np.pi, (1, 0, 0))
    world.recorders = [Recorder("exit", event="exit", histograms=[Histogram("z", -10, 10, 4)])]
    return Scene(world)


def hist_slab():
    """Histogram-sampled spectra (hist=True; the reference engine rejects these, its Python
    Distribution defines them): a dye with step-function absorption and emission tables."""
    xa = np.array([380.0, 420.0, 470.0, 500.0, 530.0, 560.0, 600.0, 650.0])
    ya = np.array([0.0, 0.5, 2.0, 4.0, 3.0, 1.0, 0.2, 0.0])
    xe = np.array([520.0, 560.0, 590.0, 610.0, 640.0, 680.0, 720.0, 760.0])
    ye = np.array([0.0, 1.0, 3.0, 4.0, 2.0, 1.0, 0.0, 0.0])
    world = Node(name="world", geometry=Box((30.0, 30.0, 30.0), material=Material(refractive_index=1.0)))
    slab = Node(name="slab", parent=world, geometry=Box((6.0, 6.0, 0.8), material=Material(
        refractive_index=1.5, components=[
            Luminophore(np.column_stack((xa, ya)), emission=np.column_stack((xe, ye)), hist=True,
                        quantum_yield=0.9, name="step-dye"),
            Absorber(np.column_stack((xa, 0.05 + 0.0 * ya)), hist=True, name="step-host")])))
    slab.recorders = face_recorders(hist=False) + [
        Recorder("out-spectrum", event="escaping", histograms=[Histogram("wavelength", 380, 780, 40)])]
    spectrum = Distribution(np.linspace(400.0, 620.0, 23), gaussian(np.linspace(400.0, 620.0, 23), 1.0, 500.0, 50.0))
    light = Node(name="lamp", parent=world, light=Light(wavelength=SpectrumWavelengthMask(spectrum),
                                                        direction=Cone(0.4), name="lamp"))
    light.location = (0.0, 0.0, 4.0)
    light.rotate(np.pi, (1, 0, 0))
    return Scene(world)


def hist_lamp():
    """hist_slab lit by two lamps, one of them with a HISTOGRAM-sampled spectrum (PVT_WL_SPECTRUM_HIST)."""
    scene = hist_slab()
    world = scene.root
    lines = Distribution(np.array([405.0, 436.0, 492.0, 546.0, 578.0, 615.0]), np.array([1.0, 3.0, 0.0, 4.0, 2.0, 0.5]),
                         hist=True)
    lamp = Node(name="line-lamp", parent=world,
                light=Light(wavelength=SpectrumWavelengthMask(lines), position=CircularMask(1.5),
                            direction=Cone(0.2), name="line-lamp"))
    lamp.location = (0.5, -0.5, 5.0)
    lamp.rotate(np.pi, (1, 0, 0))
    return Scene(world)


def mesh_lsc():
    """lsc_equivalent with the slab given as a 12-triangle mesh: must behave like the analytic
    box (same events; positions to ~1e-13 cm)."""
    scene = lsc_equivalent()
    slab = [n for n in scene.root.children if n.name == "LSC"][0]
    slab.geometry = Mesh.box((5.0, 5.0, 1.0), material=slab.geometry.material)
    return scene


def mesh_gem():
    """Faceted glass ball (320-face icosphere, rotated and off-centre) holding a scatterer and an
    analytic sphere of denser glass, in a mesh world (80-face icosphere): mesh root EXIT,
    mesh <-> analytic nesting, facet recorder on one triangle, heatmaps in the mesh frame."""
    world = Node(name="world", geometry=Mesh.icosphere(1, 10.0, material=Material(refractive_index=1.0)))
    gem = Node(name="gem", parent=world, geometry=Mesh.icosphere(2, 1.0, material=Material(
        refractive_index=1.5,
        components=[Scatterer(0.7, phase_function=HenyeyGreenstein(0.3), name="haze"),
                    Absorber(0.05, name="tint")])))
    gem.location = (0.2, -0.1, 2.0)
    gem.rotate(0.4, (0.3, 1.0, 0.2))
    core = Node(name="core", parent=gem, geometry=Sphere(0.35, material=Material(
        refractive_index=1.9, components=[Absorber(0.8, name="core-abs")])))
    core.location = (0.1, 0.0, -0.2)
    light = Node(name="lamp", parent=world, light=Light(
        position=CircularMask(0.6), direction=Cone(0.25), name="lamp"))
    light.location = (0.0, 0.0, -4.0)
    world_normals = gem.geometry.face_normals @ np.asarray(gem.transformation_to(world))[:3, :3].T
    gem.recorders = [
        Recorder("gem-in", event="entering", histograms=[Heatmap("x", "y", (-1, 1, 8), (-1, 1, 8))]),
        Recorder("gem-out", event="escaping", histograms=[Histogram("angle", 0, np.pi / 2, 12)]),
        Recorder("gem-facet", event="entering", facet=tuple(world_normals[int(np.argmin(world_normals[:, 2]))])),
        Recorder("gem-lost", event="lost"), Recorder("gem-reflected", event="reflected"),
    ]
    core.recorders = [Recorder("core-in", event="entering"), Recorder("core-lost", event="lost")]
    world.recorders = [Recorder("exit", event="exit", histograms=[Histogram("angle", 0, np.pi / 2, 9)])]
    return Scene(world)


def l_prism_mesh(material=None):
    """A NON-CONVEX closed mesh: the L-shaped hexagon (0,0)-(2,0)-(2,1)-(1,1)-(1,2)-(0,2) extruded over
    z in [0, 1] (24 triangles; the extra vertex (0,1) avoids a T-junction)."""
    poly = [(0, 0), (2, 0), (2, 1), (1, 1), (1, 2), (0, 2), (0, 1)]
    tris = [(0, 1, 2), (0, 2, 3), (0, 3, 6), (6, 3, 4), (6, 4, 5)]
    n = len(poly)
    verts = [(x, y, 0.0) for x, y in poly] + [(x, y, 1.0) for x, y in poly]
    faces = [(a, c, b) for a, b, c in tris] + [(a + n, b + n, c + n) for a, b, c in tris]
    for i in range(n):
        j = (i + 1) % n
        faces += [(i, j, j + n), (i, j + n, i + n)]
    return Mesh((np.array(verts, dtype=np.float64), np.array(faces, dtype=np.int32)), material=material, recenter=False)


def l_prism(recorders=True):
    """Glass L-prism (non-convex mesh) with a dye, lit from inside one arm across the notch: a ray that
    starts inside the mesh crosses its surface three times before reaching the world."""
    world = Node(name="world", geometry=Sphere(10.0, material=Material(refractive_index=1.0)))
    x = np.linspace(400.0, 700.0, 31)
    dye = Luminophore(np.column_stack((x, 2.0 * gaussian(x, 1.0, 480.0, 50.0))),
                      emission=np.column_stack((x, gaussian(x, 1.0, 600.0, 30.0))), quantum_yield=0.95, name="dye")
    prism = Node(name="L", parent=world, geometry=l_prism_mesh(Material(refractive_index=1.5, components=[dye, Absorber(0.05, name="bg")])))
    prism.rotate(0.3, (0.2, 1.0, 0.1))
    inside = Node(name="inside-lamp", parent=prism, light=Light(direction=Cone(0.6), wavelength=lambda: 470.0, name="inside-lamp"))
    inside.location = (1.6, 0.5, 0.5)       # in the x-arm ...
    inside.look_at((-1.0, 2.0, 0.0))        # ... aimed across the notch at the y-arm
    outside = Node(name="outside-lamp", parent=world, light=Light(position=CircularMask(1.5), direction=Cone(0.3), name="outside-lamp"))
    outside.location = (1.0, 1.0, -4.0)
    if recorders:
        prism.recorders = [Recorder("in", event="entering"), Recorder("out", event="escaping", histograms=[Histogram("wavelength", 400, 700, 30)]),
                           Recorder("lost", event="lost"), Recorder("refl", event="reflected")]
        world.recorders = [Recorder("exit", event="exit")]
    return Scene(world)


def lambertian_fog():
    """A slab whose two scatterers use the Lambertian and a cone phase function (extension: the reference engine rejects
    `lambertian` as a custom phase function, compiler.py:300-310; its scene-spec parser builds it, cli/parse.py:166-167)."""
    world = Node(name="world", geometry=Box((30.0, 30.0, 30.0), material=Material(refractive_index=1.0)))
    fog = Node(name="fog", parent=world, geometry=Box((6.0, 6.0, 2.0), material=Material(refractive_index=1.3, components=[
        Scatterer(coefficient=0.8, quantum_yield=0.97, phase_function=lambertian, name="lambert"),
        Scatterer(coefficient=0.3, quantum_yield=0.9, phase_function=Cone(0.6), name="cone"),
        Absorber(coefficient=0.05, name="loss"),
    ])))
    fog.rotate(0.3, (1.0, 0.0, 0.0))
    fog.recorders = face_recorders("fog-", hist=False)
    light = Node(name="light", parent=world, light=Light(direction=functools.partial(cone, 0.3), name="light"))
    light.location = (0.0, 0.0, 4.0)
    light.rotate(np.pi, (1, 0, 0))
    world.recorders = [Recorder("exit", event="exit", histograms=[Histogram("angle", 0, np.pi / 2, 9)])]
    return Scene(world)


def reference_spec_scene(name="tests/data/pvtrace-scene-spec.yml"):
    """One of the reference's own scene-spec files (parsed dicts: tests/golden/spec_dicts.json, made by
    tests/golden/make_spec_fixtures.py), built by the product's front-end.  The default is the reference's parser fixture:
    a sphere and a cylinder holding isotropic / Lambertian / cone / Henyey-Greenstein scatterers, an STL cube, eight
    lights with every mask."""
    import json
    import os

    from pvtrace_amd import spec

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(gold, "spec_dicts.json")) as fp:
        return spec.load(json.load(fp)[name], base=os.path.join(gold, "spec_data"))


def py_tracer_pin_scene(classes=None):
    """The scene tests/golden/py_tracer.npz was made on: a dyed, scattering glass ball and a tilted glass rod in an air
    sphere.  `classes`: a namespace with Sphere / Cylinder / Material / Luminophore / Absorber / Scatterer and the
    `lumogen` data module -- the product's by default; tests/golden/make_golden.py passes the REFERENCE's own classes
    (hung on the product's Node / Scene: the reference's scene graph needs anytree), so that the reference's
    photon_tracer.follow runs on its own geometry, materials, components and surface delegates."""
    import types

    if classes is None:
        classes = types.SimpleNamespace(Sphere=Sphere, Cylinder=Cylinder, Material=Material, Luminophore=Luminophore,
                                        Absorber=Absorber, Scatterer=Scatterer, lumogen=lumogen_f_red_305)
    c = classes
    x = np.arange(400.0, 800.0)
    comps = [c.Luminophore(np.column_stack((x, c.lumogen.absorption(x) * 3.0)),
                           emission=np.column_stack((x, c.lumogen.emission(x))), quantum_yield=0.9),
             c.Absorber(0.3), c.Scatterer(0.5, quantum_yield=0.95)]
    world = Node(name="world", geometry=c.Sphere(10.0, material=c.Material(refractive_index=1.0)))
    ball = Node(name="ball", parent=world, geometry=c.Sphere(1.0, material=c.Material(refractive_index=1.5, components=comps)))
    ball.translate((0.0, 0.0, 2.0))
    rod = Node(name="rod", parent=world, geometry=c.Cylinder(2.0, 0.5, material=c.Material(refractive_index=1.4)))
    rod.translate((0.3, 0.0, -2.0))
    rod.rotate(0.7, (0.0, 1.0, 0.2))
    return Scene(world)


def emit_pin_scene(light_module=None, phase_module=None, distribution_class=None):
    """Five posed lights with every built-in mask and direction delegate: the scene tests/golden/emit.npz was made on.
    The modules the delegates come from: the product's by default; make_golden.py passes the reference's
    (pvtrace.light.light, pvtrace.material.utils, Distribution)."""
    from pvtrace_amd import light as product_light, material as product_material

    L = light_module or product_light
    U = phase_module or product_material
    D = distribution_class or Distribution
    world = Node(name="w", geometry=Sphere(20.0, material=Material(refractive_index=1.0)))
    x = np.linspace(400, 700, 31)
    y = np.exp(-((x - 550) / 40.0) ** 2)
    specs = [dict(direction=U.Cone(0.4)), dict(direction=U.isotropic, position=L.RectangularMask(1.0, 2.0)),
             dict(direction=U.lambertian, position=L.CircularMask(1.5), wavelength=L.ConstantWavelengthMask(610.0)),
             dict(direction=U.HenyeyGreenstein(0.7), position=L.CubeMask(1, 2, 3), wavelength=L.SpectrumWavelengthMask(D(x, y))),
             dict()]
    for k, kw in enumerate(specs):
        n = Node(name=f"L{k}", parent=world, light=L.Light(name=f"L{k}", **kw))
        n.translate((0.3 * k, -0.2 * k, 1.0 + k))
        n.rotate(0.3 + 0.2 * k, (1.0, 0.2 * k, 0.1))
    return Scene(world)


def lsc_builder_cases(LSC, cone, rectangular_mask, lumogen):
    """The LSC objects tests/golden/lsc_scenes.npz describes, built through `LSC`'s public methods: the default device
    (BASELINE configs[1]) and one with everything set by hand.  Called with the product's classes by the test and with
    the reference's by tests/golden/make_golden.py."""
    import functools

    x = np.arange(450.0, 750.0, 2.0)
    custom = LSC((8.0, 4.0, 0.5), wavelength_range=x, n0=1.0, n1=1.6)
    custom.add_luminophore("Dye", np.column_stack((x, lumogen.absorption(x) * 4.0)), np.column_stack((x, lumogen.emission(x))), 0.85)
    custom.add_absorber("Host", 0.02)   # (no add_scatterer: the reference's raises NameError -- `Scatterer` is not imported in device/lsc.py:250-258)
    custom.add_light("Lamp", (0.5, -0.5, 3.0), rotation=(np.radians(180), (1, 0, 0)), direction=functools.partial(cone, np.radians(10)),
                     position=functools.partial(rectangular_mask, 2.0, 1.0))
    custom.add_solar_cell({"left", "right"})
    custom.add_back_surface_mirror()
    custom.add_air_gap_mirror(lambertian=True)
    return {"default": LSC((5.0, 5.0, 1.0)), "custom": custom}


def lsc_tracer_cases(LSC, cone, rectangular_mask, lumogen):
    """The LSC objects of tests/golden/lsc_tracer.npz (per-ray event counts of the reference's Python tracer calling the
    reference's LSC surface delegates): the two above and the default device with solar cells on all four edges and a
    back-surface mirror -- SURVEY §8(c)(5)'s case."""
    cases = lsc_builder_cases(LSC, cone, rectangular_mask, lumogen)
    cells = LSC((5.0, 5.0, 1.0))
    cells.add_solar_cell({"left", "right", "near", "far"})
    cells.add_back_surface_mirror()
    cases["cells"] = cells
    return cases


def py_tracer_pin_rays(n=300):
    """(directions, wavelengths, numpy seeds) of the rays of that fixture, all from the origin."""
    rng = np.random.default_rng(5)
    dirs, wls = [], []
    for k in range(n):
        d = rng.normal(size=3)
        if k % 2:   # every other ray straight at the ball (+z) or at the rod (-z)
            d = np.array([0.05 * rng.normal(), 0.05 * rng.normal(), 1.0 if k % 4 == 1 else -1.0])
        dirs.append(d / np.linalg.norm(d))
        wls.append(float(rng.uniform(450, 650)))
    return np.array(dirs), np.array(wls), 1000 + np.arange(n)


REFERENCE_SCENES = {   # expressible in the reference engine (no coatings)
    "hello_world": hello_world,
    "lsc_equivalent": lsc_equivalent,
    "nested_cylinders": nested_cylinders,
    "fresnel_box": fresnel_box,
    "bench_slab": lambda: bench_slab(recorders=True),
    "kitchen_sink": kitchen_sink,
    "touching_boxes": touching_boxes,
    "trapped_light": trapped_light,
}
EXTENSION_SCENES = {   # need an extension: coatings, hist spectra, meshes
    "coated_slab": coated_slab,
    "l_prism": l_prism,
    "lambertian_sheet": lambertian_sheet,
    "hist_slab": hist_slab,
    "hist_lamp": hist_lamp,
    "mesh_lsc": mesh_lsc,
    "mesh_gem": mesh_gem,
    "lambertian_fog": lambertian_fog,
    "reference_spec": reference_spec_scene,
}
ALL_SCENES = dict(REFERENCE_SCENES, **EXTENSION_SCENES)


def hello_world_recorded():
    """hello_world with whole-surface recorders on the ball and an exit recorder on the world (the example has
    none; the 10^6-photon reference tallies need something to count)."""
    scene = hello_world()
    world = scene.root
    ball = [n for n in world.children if n.geometry is not None][0]
    ball.recorders = [Recorder("ball-entering", event="entering"), Recorder("ball-escaping", event="escaping"),
                      Recorder("ball-reflected", event="reflected",
                               histograms=[Histogram("angle", 0.0, np.pi / 2, 18)])]
    world.recorders = [Recorder("exit", event="exit", histograms=[Histogram("angle", 0.0, np.pi / 2, 18)])]
    return scene


def tiles6():
    """The 6 x 6 tile array of the scene-size family (benchmarks/configs.py: tiles_lsc; 37 nodes, plain Fresnel
    surfaces) with `escaping` and `lost` recorders on every tile and the headline's face recorders on a middle one:
    82 recorders.  Large enough for the kernel's node grid, and the reference kernel can run it."""
    from benchmarks.configs import tiles_lsc
    return tiles_lsc(6, recorders="all")


TALLY_SCENES = {   # reference-kernel tallies at 10^6 photons (tests/golden/tallies_<name>_1e6.npz)
    "nested_cylinders": nested_cylinders,
    "hello_world_recorded": hello_world_recorded,
    "bench_slab_recorded": lambda: bench_slab(recorders=True),
    "tiles6": tiles6,
}


def bose_fluro_red_sample(depth=0.26):
    """The FULLSPECTRUM validation sample (reference examples/Validation.ipynb cell 6 and
    tests/test_3D_flux_comparison.py:11-64): 4.8 x 1.8 x `depth` cm plate, Fluro Red dye
    (peak 11.387815 cm^-1, qy 0.95) + 0.02 cm^-1 host absorption, n = 1.5, uniform
    top-surface illumination with the fitted lamp spectrum.  Returns an `LSC` object
    (built-in light delegates, so it also runs with device-side emission)."""
    from pvtrace_amd import LSC
    from pvtrace_amd.data import fluro_red
    from pvtrace_amd.light import RectangularMask

    x = np.arange(400, 801, dtype=float)
    size = (l, w, d) = (4.8, 1.8, depth)
    lsc = LSC(size, wavelength_range=x, n1=1.5)
    lsc.add_luminophore("Fluro Red", np.column_stack((x, fluro_red.absorption(x) * 11.387815)),
                        np.column_stack((x, fluro_red.emission(x))), quantum_yield=0.95)
    lsc.add_absorber("PMMA", 0.02)

    def g(v, a, p, wd):
        return a * np.exp(-(((p - v) / wd) ** 2))

    lamp = (g(x, 0.53025700136646192, 512.91400020614333, 93.491838802960473)
            + g(x, 0.63578999789955015, 577.63100003089369, 66.031706473985736))
    lsc.add_light("Oriel Lamp + Filter", (0.0, 0.0, 0.5 * d + 0.01),
                  rotation=(np.radians(180), (1, 0, 0)),
                  wavelength=SpectrumWavelengthMask(Distribution(x, lamp)),
                  position=RectangularMask(l / 2, w / 2))
    return lsc


def object_method_script(c):
    """The per-interaction methods of the host objects (reference material/material.py:22-63, component.py:168-196, :381-440,
    surface.py:224-272) called in a fixed order under numpy seeds -> {name: array}.  `c`: a namespace with Material, Absorber,
    Scatterer, Reactor, Luminophore, Surface, NullSurfaceDelegate, Sphere, Ray, henyey_greenstein, cone and the lumogen
    spectra module -- the product's classes in the test, the reference's in tests/golden/make_golden.py."""
    import functools
    import types

    x = np.arange(400.0, 801.0, 1.0)
    dye = c.Luminophore(np.column_stack((x, c.lumogen.absorption(x) * 8.0)), np.column_stack((x, c.lumogen.emission(x))),
                        quantum_yield=0.9, tau_rad=6e-9, tau_nr=2e-9, phase_function=functools.partial(c.henyey_greenstein, 0.4), name="dye")
    host = c.Absorber(0.3, tau_nr=1e-9, name="host")
    haze = c.Scatterer(np.column_stack((x, 0.2 + 0.001 * (x - 400.0))), quantum_yield=0.8, phase_function=functools.partial(c.cone, 0.5), name="haze")
    react = c.Reactor(0.05, name="react")
    material = c.Material(1.5, components=[dye, host, haze, react])
    clear = c.Material(1.0)
    out = {}
    np.random.seed(2024)
    wls = np.random.uniform(450.0, 700.0, 40)
    rays = [c.Ray(position=(0.1 * k, 0.0, 0.0), direction=(0.0, 0.0, 1.0), wavelength=float(w), duration=1e-10 * k) for k, w in enumerate(wls)]
    np.random.seed(7)
    out["depth"] = np.array([material.penetration_depth(float(w)) for w in wls])
    out["clear_depth"] = np.array([clear.penetration_depth(555.0)])
    absorbed = [material.is_absorbed(r, 0.8) for r in rays]
    out["absorbed"] = np.array([float(a) for a, _ in absorbed])
    out["absorbed_at"] = np.array([d for _, d in absorbed])
    names = ["dye", "host", "haze", "react"]
    out["component"] = np.array([names.index(material.component(float(w)).name) for w in wls])
    for comp in (dye, host, haze, react):
        out[f"{comp.name}_radiative"] = np.array([float(comp.is_radiative(r)) for r in rays])
        out[f"{comp.name}_ended_duration"] = np.array([comp.nonradiative_absorb(r).duration for r in rays])
    for method in ("kT", "redshift", "full"):
        new = [dye.emit(r, method=method) for r in rays if r.wavelength > 460.0]
        out[f"dye_emit_{method}"] = np.array([list(n.direction) + [n.wavelength, n.duration] for n in new])
        assert all(n.source == "dye" for n in new)
    new = [haze.emit(r) for r in rays]
    out["haze_emit"] = np.array([list(n.direction) + [n.wavelength, n.duration] for n in new])
    assert all(n.source == "haze" for n in new)
    # surfaces: a glass ball in air, rays meeting it from outside and from inside
    ball = c.Sphere(radius=1.0, material=material)
    inside = types.SimpleNamespace(geometry=types.SimpleNamespace(material=types.SimpleNamespace(refractive_index=1.5)))
    outside = types.SimpleNamespace(geometry=types.SimpleNamespace(material=types.SimpleNamespace(refractive_index=1.0)))
    rng = np.random.default_rng(3)
    decisions, turned = [], []
    for k in range(60):
        p = rng.normal(size=3); p /= np.linalg.norm(p)
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        leaving = k % 2 == 0
        if (np.dot(d, p) > 0) != leaving:
            d = -d
        ray = c.Ray(position=tuple(p.tolist()), direction=tuple(d.tolist()), wavelength=555.0)
        where = (ball, inside, outside) if leaving else (ball, outside, inside)
        surface = material.surface
        hit = surface.is_reflected(ray, *where)
        decisions.append(float(hit))
        turned.append(list((surface.reflect(ray, *where) if hit else surface.transmit(ray, *where)).direction))
    out["surface_reflected"] = np.array(decisions)
    out["surface_direction"] = np.array(turned)
    null = c.Surface(delegate=c.NullSurfaceDelegate())
    ray = c.Ray(position=(0.0, 0.0, 1.0), direction=(0.0, 0.6, 0.8), wavelength=555.0)
    out["null"] = np.array([float(null.is_reflected(ray, ball, inside, outside))] + list(null.transmit(ray, ball, inside, outside).direction))
    return out
