"""TEST INFRASTRUCTURE -- a pure-Python, one-ray-at-a-time photon tracer over the host scene API.

It restates the reference's *Python* path, `pvtrace/algorithm/photon_tracer.py` (`find_container`
:26-57, `next_hit` :60-109, `step_forward` :112-273, `follow` :276-328) on `pvtrace_amd`'s own
`Scene` / `Node` / geometry / material objects: the scene graph is walked directly, surface
delegates and phase functions are CALLED, spectra are sampled through `Distribution`.  Nothing is
flattened and no table of the engine is involved, so it is an independent check of the flattener
plus kernel (statistics only: it draws from numpy's global generator, like the reference).

Used by tests only (BASELINE configs[0]: hello_world, 1 000 rays, "pure-Python photon_tracer on CPU,
plumbing, no GPU").  It is ~10^3 rays/s; never imported by `pvtrace_amd`.

Differences from the kernel that show up in statistics are the ones SURVEY.md lists for the
reference's two tracers (spectra raise instead of clamping outside their range; re-emission
directions are drawn in the container's frame).
"""
import collections
import math

import numpy as np

from pvtrace_amd.light import Event
from pvtrace_amd.material import Luminophore, Reactor, Scatterer

EPS_ZERO = 2.220446049250313e-13
KB_EV = 1.380649e-23 / 1.60217662e-19


def _holders(crossings):
    """Crossings of the nodes the ray starts inside of, nearest first: crossed exactly once
    (photon_tracer.py:26-57) -- or, for a triangle mesh, which may be non-convex, an odd number of
    times (extension: Mesh.contains semantics, geometry/mesh.py:29-32)."""
    from pvtrace_amd.geometry import Mesh

    count = collections.Counter(id(x.hit) for x in crossings)
    firsts, seen = [], set()
    for x in sorted(crossings, key=lambda c: c.distance):
        if id(x.hit) in seen:
            continue
        seen.add(id(x.hit))
        n = count[id(x.hit)]
        if (n % 2 == 1) if isinstance(x.hit.geometry, Mesh) else n == 1:
            firsts.append(x)
    return firsts


def find_container(crossings):
    """The node the ray is inside (photon_tracer.py:26-57)."""
    if len(crossings) == 1:
        return crossings[0].hit
    return _holders(crossings)[0].hit


def next_hit(scene, ray):
    """(hit node, (container, adjacent), point, distance) of the next interface (:60-109)."""
    crossings = [x for x in scene.intersections(ray.position, ray.direction) if x.distance > EPS_ZERO]
    if not crossings:
        return None
    first = crossings[0]
    if len(crossings) == 1:
        return first.hit, (first.hit, None), first.point, first.distance
    container = find_container(crossings)
    adjacent = crossings[1].hit if container is first.hit else first.hit
    from pvtrace_amd.geometry import Mesh

    if container is first.hit and isinstance(first.hit.geometry, Mesh):
        holders = _holders(crossings)
        if len(holders) > 1:      # beyond the surface of a mesh: the next node that holds the ray
            adjacent = holders[1].hit
    return first.hit, (container, adjacent), first.point, first.distance


def _pick_component(material, wavelength):
    """Component responsible for an absorption, proportional to its coefficient
    (material/material.py:49-63)."""
    weights = np.array([c.coefficient(wavelength) for c in material.components], dtype=np.float64)
    running = np.cumsum(weights)
    index = int(np.searchsorted(running, np.random.uniform() * running[-1], side="left"))
    return material.components[min(index, len(running) - 1)]


def _emit(component, ray, method):
    """New direction (component's phase function) and, for a luminophore, new wavelength
    (material/component.py:381-440)."""
    direction = tuple(float(v) for v in component.phase_function())
    wavelength = ray.wavelength
    if isinstance(component, Luminophore):
        dist = component._ems_dist
        if method == "full":
            p1 = 0.0
        else:
            nm = ray.wavelength
            if method == "kT":
                nm = 1240.0 / (1240.0 / nm + 1.5 * KB_EV * 300.0)
            lo, hi = float(dist._x[0]), float(dist._x[-1])
            p1 = float(dist.lookup(min(max(nm, lo), hi)))
        gamma = p1 + (1.0 - p1) * np.random.uniform()
        wavelength = float(dist.sample(gamma))
    return direction, wavelength


def step_forward(scene, ray, maxsteps=1000, emit_method="kT"):
    """Generator of (ray, event, metadata), as the reference's (:112-273)."""
    from dataclasses import replace

    root = scene.root
    count = 0
    yield ray, Event.GENERATE, None
    while True:
        count += 1
        info = next_hit(scene, ray)
        if info is None:
            return
        hit, (container, adjacent), point, distance = info
        if count > maxsteps:
            yield ray, Event.KILL, {"container": container.name}
            return
        material = container.geometry.material
        n_container = material.refractive_index
        if hit is root:
            yield ray.propagate(distance, n_container), Event.EXIT, {
                "hit": hit.name, "container": container.name,
                "adjacent": None if adjacent is None else adjacent.name}
            return
        alpha = material.total_attenutation_coefficient(ray.wavelength) if material.components else 0.0
        depth = -math.log(1.0 - np.random.uniform()) / alpha if alpha > 0.0 else math.inf
        if depth < distance:
            ray = ray.propagate(depth, n_container)
            component = _pick_component(material, ray.wavelength)
            yield ray, Event.ABSORB, {"component": component.name, "container": container.name}
            radiative = isinstance(component, Scatterer) and np.random.uniform() < component.quantum_yield
            if not radiative:
                kind = Event.REACT if isinstance(component, Reactor) else Event.NONRADIATIVE
                yield ray, kind, {"component": component.name, "container": container.name}
                return
            local = ray.representation(root, container)
            direction, wavelength = _emit(component, local, emit_method)
            ray = replace(local, direction=direction, wavelength=wavelength,
                          source=component.name).representation(container, root)
            kind = Event.EMIT if isinstance(component, Luminophore) else Event.SCATTER
            yield ray, kind, {"component": component.name, "container": container.name}
            continue
        # surface interaction, in the frame of the node that was hit (:218-273)
        ray = ray.propagate(distance, n_container)
        surface = hit.geometry.material.surface
        local = ray.representation(root, hit)
        normal = hit.vector_to_node(hit.geometry.normal(local.position), root)
        meta = {"hit": hit.name, "container": container.name,
                "adjacent": None if adjacent is None else adjacent.name, "normal": normal}
        delegate = surface.delegate
        reflectivity = delegate.reflectivity(surface, local, hit.geometry, container, adjacent)
        if reflectivity > 0.0 and np.random.uniform() < reflectivity:
            new_dir = delegate.reflected_direction(surface, local, hit.geometry, container, adjacent)
            kind = Event.REFLECT
        else:
            new_dir = delegate.transmitted_direction(surface, local, hit.geometry, container, adjacent)
            kind = Event.TRANSMIT
        ray = replace(local, direction=tuple(float(v) for v in new_dir)).representation(hit, root)
        yield ray, kind, meta


def follow(scene, ray, maxsteps=1000, emit_method="kT"):
    """[(ray, event)] of one photon (:276-328)."""
    return [(r, e) for r, e, _ in step_forward(scene, ray, maxsteps=maxsteps, emit_method=emit_method)]


def event_means(scene, num_rays, seed=0, **kwargs):
    """Mean number of each event kind per ray over `num_rays` emitted rays, and all histories."""
    np.random.seed(seed)
    totals = collections.Counter()
    histories = []
    for ray in scene.emit(num_rays):
        history = list(step_forward(scene, ray, **kwargs))
        histories.append(history)
        totals.update(event for _, event, _ in history)
    return {event: totals[event] / num_rays for event in totals}, histories
