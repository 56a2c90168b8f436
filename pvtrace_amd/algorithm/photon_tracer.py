"""The reference's per-ray entry points (`pvtrace/algorithm/photon_tracer.py:112-328`: `step_forward`, `follow`), for callers
that trace ray by ray -- `for ray in scene.emit(n): history = photon_tracer.follow(scene, ray)`, the loop of the reference's
`LSC.simulate` (`device/lsc.py:349-350`) and of its examples.

With a GPU visible the ray is traced by the HIP engine as a bundle of one with its full history, and the
history is handed back in the reference's form, `[(Ray, Event), ...]` (`follow`) or `(Ray, Event, metadata)` one by one
(`step_forward`).  Same scene semantics as `engine.simulate` (the reference's two tracers differ in documented corners,
SURVEY.md section 5: spectra clamp outside their range, re-emission is sampled per `emit_method` by the kernel's rule).
Random decisions come from the kernel's per-ray stream: `seed` names it; without one it is drawn from numpy's global
generator, as the reference's tracer draws everything from it -- so `numpy.random.seed` makes a sequence of calls
reproducible here too, though not the same photons as the reference's.  For more than a handful of rays call
`engine.simulate(scene, n)` (one launch for all of them) and read `.histories()`.

BASELINE configs[0] ("hello_world, 1 000 rays, pure-Python photon_tracer on CPU -- plumbing, no GPU"): without a GPU -- or
with `backend="host"` -- the same two entry points step the ray through the scene OBJECTS, as the reference's Python tracer
does: `Scene.intersections` for the next interface, then the per-interaction methods the host classes carry
(`Material.is_absorbed / component`, `Component.is_radiative / emit / nonradiative_absorb`, `Surface.is_reflected / reflect /
transmit`; each held to the reference's under numpy seeds, tests/golden/object_methods.npz), every decision drawn from
numpy's global generator.  That path serves per-ray callers only and says so on stderr the first time it is taken because no
GPU was found; `engine.simulate` / `_kernel.trace_bundle` have no CPU path and raise `EngineUnavailableError`."""
import numpy as np

from pvtrace_amd.light import Event

# a step writes at most two rows (ABSORB + what follows); GENERATE and the closing row come on top
_ROWS_PER_STEP = 2


def _history(scene, ray, maxsteps, emit_method, seed, session):
    from pvtrace_amd.engine.api import Session

    if seed is None:
        seed = np.random.randint(0, 2 ** 31 - 1)
    max_events = _ROWS_PER_STEP * int(maxsteps) + 8
    rays = (np.asarray([ray.position], dtype=np.float64), np.asarray([ray.direction], dtype=np.float64),
            np.asarray([ray.wavelength], dtype=np.float64), [ray.source])
    own = session is None
    if own:
        session = Session(scene, emission="host")
    try:
        pending = session.submit(1, int(seed), maxsteps=int(maxsteps), max_events=max_events, emit_method=emit_method,
                                 record_every=1, host_rays=rays)
        result = session.collect(pending)
    finally:
        if own:
            session.close()
    history = next(iter(result.histories()))
    # the incoming ray's clocks are where the history starts (the kernel starts every ray at zero)
    if ray.travelled or ray.duration:
        import dataclasses

        history = [(dataclasses.replace(r, travelled=r.travelled + ray.travelled, duration=r.duration + ray.duration), e, m)
                   for r, e, m in history]
    return history


def _interface_ahead(scene, ray):
    """The next interface on the ray's line -> (node hit, node the ray is in, node beyond the interface, distance) or None
    (reference `next_hit` / `find_container`, photon_tracer.py:26-109).  A node the ray will cross exactly once more
    holds the ray (a solid is left once; one that is entered is also left); the nearest such crossing names the container.
    Triangle meshes (an extension) may be non-convex: an odd number of crossings holds the ray."""
    from pvtrace_amd.geometry import Mesh

    ahead = [x for x in scene.intersections(ray.position, ray.direction) if x.distance > _EPS]
    if not ahead:
        return None
    nearest = ahead[0]
    if len(ahead) == 1:
        return nearest.hit, nearest.hit, None, nearest.distance
    times = {}
    for x in ahead:
        times[id(x.hit)] = times.get(id(x.hit), 0) + 1
    holding = []   # nodes that hold the ray, innermost first
    for x in ahead:
        n = times[id(x.hit)]
        inside = (n % 2 == 1) if isinstance(x.hit.geometry, Mesh) else n == 1
        if inside and all(x.hit is not h for h in holding):
            holding.append(x.hit)
    container = holding[0]
    if container is not nearest.hit:
        beyond = nearest.hit                      # entering the node that is hit
    elif isinstance(container.geometry, Mesh) and len(holding) > 1:
        beyond = holding[1]                       # leaving a mesh: the next node that holds the ray
    else:
        beyond = ahead[1].hit                     # leaving the container: whatever comes next on the line
    return nearest.hit, container, beyond, nearest.distance


def _steps_on_host_objects(scene, ray, maxsteps, maxpathlength, emit_method):
    """The photon loop on the scene objects (reference `step_forward`, photon_tracer.py:112-273), one decision at a time by
    the objects' own methods.  Order of a step, as there: count it; find the interface; KILL when over `maxsteps` or
    `maxpathlength`; EXIT when the interface is the world's; else ask the container's material whether the photon is
    absorbed on the way (ABSORB, then EMIT / SCATTER and on, or NONRADIATIVE / REACT and out) or reaches the surface,
    where the hit node's surface reflects or transmits it in that node's frame."""
    from pvtrace_amd.material import Luminophore, Reactor, Scatterer

    root = scene.root
    yield ray, Event.GENERATE, None
    taken = 0
    while True:
        taken += 1
        found = _interface_ahead(scene, ray)
        if found is None:
            return
        hit, container, beyond, distance = found
        if taken > maxsteps or ray.travelled > maxpathlength:
            yield ray, Event.KILL, {"maxsteps": taken, "maxpathlength": ray.travelled, "container": container.name}
            return
        medium = container.geometry.material
        index = medium.refractive_index
        names = {"hit": hit.name, "container": container.name, "adjacent": None if beyond is None else beyond.name}
        if hit is root:
            yield ray.propagate(distance, index), Event.EXIT, names
            return
        absorbed, depth = medium.is_absorbed(ray, distance)
        if absorbed:
            ray = ray.propagate(depth, index)
            taker = medium.component(ray.wavelength)
            who = {"component": taker.name, "container": container.name}
            yield ray, Event.ABSORB, dict(who)
            if not taker.is_radiative(ray):
                yield taker.nonradiative_absorb(ray), (Event.REACT if isinstance(taker, Reactor) else Event.NONRADIATIVE), who
                return
            if not isinstance(taker, (Luminophore, Scatterer)):
                raise ValueError("Unknown component")
            ray = taker.emit(ray.representation(root, container), method=emit_method).representation(container, root)
            yield ray, (Event.EMIT if isinstance(taker, Luminophore) else Event.SCATTER), dict(who, emit_method=emit_method)
            continue
        ray = ray.propagate(distance, index)
        skin = hit.geometry.material.surface
        local = ray.representation(root, hit)   # the surface's questions are asked in the frame of the node that is hit
        names["normal"] = hit.vector_to_node(hit.geometry.normal(local.position), root)
        if skin.is_reflected(local, hit.geometry, container, beyond):
            ray, what = skin.reflect(local, hit.geometry, container, beyond).representation(hit, root), Event.REFLECT
        else:
            ray, what = skin.transmit(local, hit.geometry, container, beyond).representation(hit, root), Event.TRANSMIT
        yield ray, what, names


_EPS = 2.220446049250313e-13     # the reference's EPS_ZERO (common.py): crossings nearer than this are the surface just left
_HOST_PATH_ANNOUNCED = []


def _use_host_objects(backend):
    if backend not in ("auto", "gpu", "host"):
        raise ValueError("backend must be 'auto', 'gpu' or 'host'")
    if backend == "host":
        return True
    if backend == "gpu":
        return False
    from pvtrace_amd.engine import native

    if native.is_available():
        return False
    if not _HOST_PATH_ANNOUNCED:
        import sys

        _HOST_PATH_ANNOUNCED.append(True)
        print("[pvtrace_amd] no MI355X visible: photon_tracer.follow / step_forward step this ray through the scene objects "
              "on the host (the reference's per-ray Python path, ~10^3 rays/s); engine.simulate has no CPU path.", file=sys.stderr)
    return True


def step_forward(scene, ray, maxsteps=1000, maxpathlength=np.inf, emit_method="kT", *, seed=None, session=None, backend="auto"):
    """Generates `(Ray, Event, metadata)` along one photon's path, as the reference's generator does (:112-273).
    `metadata` holds the node / component names the reference's carries (`hit`, `container`, `adjacent`, `component`,
    `normal` where there is one); the GENERATE row's is None.  `maxpathlength`: the reference ends a photon (KILL) at the
    start of the first step it enters having travelled further than this (:162-172); so does this, on the finished history.
    `session`: an `engine.Session` of the scene to reuse between calls (the scene then stays resident on the GPU).
    `backend`: "gpu", "host" (the scene objects' own per-interaction methods, numpy's global generator; `seed` / `session`
    do not apply) or "auto" = the GPU when one is visible, else the host objects (see the module docstring)."""
    if _use_host_objects(backend):
        yield from _steps_on_host_objects(scene, ray, maxsteps, maxpathlength, emit_method)
        return
    history = _history(scene, ray, maxsteps, emit_method, seed, session)
    closes_step = {Event.GENERATE, Event.REFLECT, Event.TRANSMIT, Event.EMIT, Event.SCATTER}
    for k, (r, event, metadata) in enumerate(history):
        yield (r, event, None if event == Event.GENERATE else metadata)
        if (event in closes_step and r.travelled > maxpathlength and k + 1 < len(history)):
            container = (history[k + 1][2] or {}).get("container")
            yield (r, Event.KILL, {"maxpathlength": r.travelled, "container": container})
            return


def follow(scene, ray, maxsteps=1000, maxpathlength=np.inf, emit_method="kT", *, seed=None, session=None, backend="auto"):
    """One photon's full path: `[(Ray, Event), ...]` (reference :276-328; metadata dropped, as there)."""
    return [(r, event) for r, event, _ in step_forward(scene, ray, maxsteps=maxsteps, maxpathlength=maxpathlength,
                                                       emit_method=emit_method, seed=seed, session=session, backend=backend)]
