"""Random scene generator for differential tests: every shape, component, phase function,
surface tag, light mask and recorder kind in random combinations, with random (also overlapping)
placement and nesting.  Overlaps are deliberate -- two implementations of the same rules must agree
on ill-posed scenes too.  `extensions=True` adds what the reference engine cannot express
(meshes, coatings, histogram-sampled spectra, source-filtered recorders)."""
import numpy as np

from pvtrace_amd import (
    Absorber, Box, Coating, CoatedSurfaceDelegate, Cylinder, Light, Luminophore, Material, Mesh, Node,
    NullSurfaceDelegate, Reactor, Scatterer, Scene, Sphere, Surface, isotropic, lambertian,
)
from pvtrace_amd.engine import Heatmap, Histogram, Recorder
from pvtrace_amd.light import CircularMask, ConstantWavelengthMask, CubeMask, RectangularMask, SpectrumWavelengthMask
from pvtrace_amd.material import Cone, Distribution, HenyeyGreenstein, gaussian

PROPS = ("wavelength", "angle", "duration", "pathlength", "x", "y", "z")
RANGES = {"wavelength": (350.0, 850.0), "angle": (0.0, 1.6), "duration": (0.0, 2e-9), "pathlength": (0.0, 40.0),
          "x": (-2.0, 2.0), "y": (-2.0, 2.0), "z": (-2.0, 2.0)}


def _spectrum(rng, even):
    n = int(rng.integers(2, 40)) if rng.random() > 0.04 else int(rng.integers(3000, 7000))   # some too big for LDS
    x = np.linspace(360.0, 840.0, n) if even else np.sort(rng.uniform(360.0, 840.0, n))
    x[0], x[-1] = 360.0, 840.0
    y = rng.uniform(0.0, 1.0) * gaussian(x, 1.0, rng.uniform(420, 760), rng.uniform(15, 120)) + rng.uniform(0, 0.05)
    return np.column_stack((x, y))


def _phase(rng):
    k = rng.integers(0, 4)
    return [None, isotropic, HenyeyGreenstein(float(rng.uniform(-0.9, 0.9))), Cone(float(rng.uniform(0.05, 1.4)))][k]


def _components(rng, extensions):
    out = []
    for k in range(int(rng.integers(0, 4))):
        kind = rng.integers(0, 4)
        hist = bool(extensions and rng.random() < 0.2)
        coeff = float(rng.uniform(0.05, 3.0)) if rng.random() < 0.4 else rng.uniform(0.3, 4.0) * _spectrum(rng, rng.random() < 0.5)
        if isinstance(coeff, np.ndarray):
            coeff[:, 1] = np.maximum(coeff[:, 1], 1e-3)
        tau = [None, float(rng.uniform(1e-10, 5e-9))][int(rng.integers(0, 2))]
        name = f"c{rng.integers(1 << 30)}"
        if kind == 0:
            out.append(Absorber(coeff, tau_nr=tau, name=name, hist=hist and isinstance(coeff, np.ndarray)))
        elif kind == 1:
            out.append(Scatterer(coeff, quantum_yield=float(rng.uniform(0.3, 1.0)), phase_function=_phase(rng),
                                 name=name, hist=hist and isinstance(coeff, np.ndarray)))
        elif kind == 2:
            out.append(Luminophore(coeff, emission=_spectrum(rng, rng.random() < 0.5), quantum_yield=float(rng.uniform(0.3, 1.0)),
                                   tau_rad=tau, tau_nr=tau, phase_function=_phase(rng), name=name,
                                   hist=hist and isinstance(coeff, np.ndarray)))
        else:
            out.append(Reactor(coeff, name=name))
    return out


def _geometry(rng, material, extensions):
    k = rng.integers(0, 4 if extensions else 3)
    if k == 0:
        return Box(tuple(rng.uniform(0.8, 5.0, 3)), material=material)
    if k == 1:
        return Sphere(float(rng.uniform(0.5, 2.8)), material=material)
    if k == 2:
        return Cylinder(float(rng.uniform(0.8, 5.0)), float(rng.uniform(0.4, 2.0)), material=material)
    if rng.random() < 0.5:
        return Mesh.icosphere(int(rng.integers(0, 3)), float(rng.uniform(0.6, 2.6)), material=material)
    return Mesh.box(tuple(rng.uniform(0.8, 4.0, 3)), material=material)


def _recorders(rng, node, is_root, component_names, extensions):
    recs = []
    for k in range(int(rng.integers(0, 4))):
        event = "exit" if is_root and rng.random() < 0.5 else str(rng.choice(["entering", "escaping", "reflected", "lost", "reacted", "killed"]))
        hists = []
        for _ in range(int(rng.integers(0, 3))):
            if rng.random() < 0.3:
                a, b = rng.choice(PROPS, 2, replace=False)
                hists.append(Heatmap(str(a), str(b), (*RANGES[str(a)], int(rng.integers(1, 9))), (*RANGES[str(b)], int(rng.integers(1, 9)))))
            else:
                a = str(rng.choice(PROPS))
                hists.append(Histogram(a, *RANGES[a], int(rng.integers(1, 40))))
        facet = None
        if event in ("entering", "escaping", "reflected") and rng.random() < 0.4:
            axis = np.zeros(3); axis[rng.integers(0, 3)] = rng.choice([-1.0, 1.0])
            facet = tuple(np.asarray(node.transformation_to(node.root))[:3, :3] @ axis)
        source = None
        if extensions and rng.random() < 0.3:
            source = str(rng.choice(["lights", "components"] + list(component_names))) if component_names or True else None
        recs.append(Recorder(f"{node.name}-r{k}", event=event, facet=facet, histograms=hists,
                             **({"source": source} if source else {})))
    return recs


def random_scene(seed, extensions=False):
    rng = np.random.default_rng(seed)
    world_material = Material(1.0, components=_components(rng, extensions) if rng.random() < 0.3 else [])
    world = Node(name="world", geometry=(Sphere(12.0, material=world_material) if rng.random() < 0.7
                                         else Box((20.0, 20.0, 20.0), material=world_material)))
    nodes = [world]
    for k in range(int(rng.integers(1, 7))):
        surface = None
        pick = rng.random()
        if pick < 0.15:
            surface = Surface(delegate=NullSurfaceDelegate())
        elif extensions and pick < 0.4:
            facet = np.zeros(3); facet[rng.integers(0, 3)] = rng.choice([-1.0, 1.0])
            surface = Surface(delegate=CoatedSurfaceDelegate([Coating(
                facet=tuple(facet), reflectivity=[None, 0.0, 1.0, float(rng.uniform(0, 1))][int(rng.integers(0, 4))],
                reflection=str(rng.choice(["specular", "lambertian"])),
                transmission=str(rng.choice(["fresnel", "matched"])))]))
        material = Material(float(rng.uniform(1.0, 2.2)), surface=surface, components=_components(rng, extensions))
        parent = world if rng.random() < 0.6 else nodes[int(rng.integers(0, len(nodes)))]
        node = Node(name=f"n{k}", parent=parent, geometry=_geometry(rng, material, extensions))
        node.translate(tuple(rng.uniform(-1.5, 1.5, 3)))
        if rng.random() < 0.6:
            node.rotate(float(rng.uniform(0, np.pi)), tuple(rng.normal(size=3)))
        nodes.append(node)
    names = [c.name for n in nodes for c in n.geometry.material.components]
    for node in nodes:
        node.recorders = _recorders(rng, node, node is world, names, extensions)
    if rng.random() < 0.08:   # sometimes more than 64 recorders: the wide per-ray "seen" mask
        target = nodes[int(rng.integers(0, len(nodes)))]
        target.recorders = list(target.recorders) + [
            Recorder(f"extra{k}", event=str(rng.choice(["entering", "escaping", "reflected", "lost"])),
                     histograms=[Histogram("wavelength", 350.0, 850.0, 5)] if k % 9 == 0 else [])
            for k in range(int(rng.integers(65, 120)))]
    taken = set()
    for node in nodes:      # recorder names must be unique
        node.recorders = [r for r in node.recorders if not (r.name in taken or taken.add(r.name))]
    for k in range(int(rng.integers(1, 3))):
        wl = ConstantWavelengthMask(float(rng.uniform(400, 800))) if rng.random() < 0.6 else SpectrumWavelengthMask(
            Distribution(*_spectrum(rng, True).T))
        pos = [None, RectangularMask(0.8, 0.5), CircularMask(0.7), CubeMask(0.3, 0.3, 0.3)][int(rng.integers(0, 4))]
        direc = [None, None, Cone(float(rng.uniform(0.05, 0.6))), isotropic, lambertian, Cone(float(rng.uniform(0.05, 1.2))),
                 HenyeyGreenstein(float(rng.uniform(-0.8, 0.8)))][int(rng.integers(0, 7))]
        light = Node(name=f"light{k}", parent=world, light=Light(wavelength=wl, position=pos, direction=direc, name=f"light{k}"))
        light.translate(tuple(rng.uniform(-3.0, 3.0, 3)))
        light.look_at(tuple(-np.asarray(light.location) + rng.normal(scale=0.3, size=3)))
    return Scene(world)


def random_many_scene(seed, n_nodes=None):
    """Scenes of MANY nodes (8 ... 120; the kernel's node-grid path): small shapes from a pool of shared materials
    scattered over a box region, a good part of them on a lattice whose pitch EQUALS their size (faces shared with
    the neighbours: crossings of two nodes at the very same distance, which the reference orders by node index),
    some rotated by one of a few shared rotations or a random one, some nested inside earlier ones, some overlapping.
    The reference engine can express all of it (no extensions)."""
    rng = np.random.default_rng(50_000 + seed)
    n = int(n_nodes or rng.integers(8, 121))
    world_material = Material(1.0, components=[Absorber(float(rng.uniform(0.001, 0.01)), name="haze")] if rng.random() < 0.2 else [])
    world = Node(name="world", geometry=(Sphere(30.0, material=world_material) if rng.random() < 0.4
                                         else Box((44.0, 40.0, 36.0), material=world_material)))
    pool = []
    for k in range(int(rng.integers(1, 5))):
        comps = _components(rng, False)
        for j, c in enumerate(comps):
            c.name = f"m{k}c{j}"
        surface = Surface(delegate=NullSurfaceDelegate()) if rng.random() < 0.1 else None
        pool.append(Material(float(rng.choice([1.0, 1.33, 1.5, 1.5, 1.7, float(rng.uniform(1.0, 2.2))])), surface=surface, components=comps))
    kinds = rng.choice(3, p=[[0.7, 0.2, 0.1], [0.34, 0.33, 0.33], [1.0, 0.0, 0.0]][int(rng.integers(0, 3))], size=n)
    rotations = [(float(rng.uniform(0, np.pi)), tuple(rng.normal(size=3))) for _ in range(3)]
    lattice = float(rng.choice([1.0, 1.5, 2.0]))
    flat = rng.random() < 0.4          # a tile array: one layer
    span = max(2, int(np.ceil(n ** (0.5 if flat else 1 / 3.0))) + 1)
    nodes, used = [world], set()
    for k in range(n):
        material = pool[int(rng.integers(0, len(pool)))]
        # (materials are shared objects: the flattener repeats their tables per node, the packer stores them once)
        nested = k > 2 and rng.random() < 0.15
        on_lattice = not nested and rng.random() < 0.6
        if kinds[k] == 0:
            size = (lattice,) * 3 if on_lattice else tuple(rng.uniform(0.3, 2.5, 3))
            if nested:
                size = tuple(rng.uniform(0.1, 0.4, 3))
            geometry = Box(size, material=material)
        elif kinds[k] == 1:
            geometry = Sphere(float(0.5 * lattice if on_lattice else rng.uniform(0.1 if nested else 0.3, 0.35 if nested else 1.4)), material=material)
        else:
            geometry = Cylinder(float(lattice if on_lattice else rng.uniform(0.2 if nested else 0.5, 0.4 if nested else 2.5)),
                                float(0.5 * lattice if on_lattice else rng.uniform(0.1, 0.3 if nested else 1.0)), material=material)
        parent = nodes[int(rng.integers(1, len(nodes)))] if nested else world
        node = Node(name=f"n{k}", parent=parent, geometry=geometry)
        if nested:
            node.translate(tuple(rng.uniform(-0.2, 0.2, 3)))
        elif on_lattice:
            for _ in range(50):
                cell = (int(rng.integers(0, span)), int(rng.integers(0, span)), 0 if flat else int(rng.integers(0, span)))
                if cell not in used:
                    break
            used.add(cell)
            node.translate(tuple((np.array(cell) - 0.5 * span) * lattice))
        else:
            node.translate(tuple(rng.uniform(-0.6 * span * lattice, 0.6 * span * lattice, 3) * (1.0, 1.0, 0.2 if flat else 1.0)))
        if not on_lattice and rng.random() < 0.35:
            angle, axis = rotations[int(rng.integers(0, 3))] if rng.random() < 0.6 else (float(rng.uniform(0, np.pi)), tuple(rng.normal(size=3)))
            node.rotate(angle, axis)
        nodes.append(node)
    for node in nodes:
        if rng.random() < (1.0 if node is world else 12.0 / n):
            node.recorders = _recorders(rng, node, node is world, [], False)
    taken = set()
    for node in nodes:
        node.recorders = [r for r in (node.recorders or []) if not (r.name in taken or taken.add(r.name))]
    for k in range(int(rng.integers(1, 3))):
        pos = [None, RectangularMask(0.5 * span * lattice, 0.5 * span * lattice), CubeMask(1.0, 1.0, 1.0)][int(rng.integers(0, 3))]
        direc = [isotropic, Cone(float(rng.uniform(0.2, 1.2))), lambertian, None][int(rng.integers(0, 4))]
        light = Node(name=f"light{k}", parent=world, light=Light(wavelength=ConstantWavelengthMask(float(rng.uniform(420, 700))),
                                                                 position=pos, direction=direc, name=f"light{k}"))
        light.translate(tuple(rng.uniform(-1.0, 1.0, 3) * (1.0, 1.0, 0.0) + (0.0, 0.0, float(rng.choice([-1, 1])) * (0.8 if flat else 0.6 * span) * lattice + 0.3)))
        light.look_at(tuple(-np.asarray(light.location) + rng.normal(scale=0.5, size=3)))
    return Scene(world)
