"""The reference's per-ray entry points (`pvtrace/algorithm/photon_tracer.py:112-328`: `step_forward`, `follow`), for callers
that trace ray by ray -- `for ray in scene.emit(n): history = photon_tracer.follow(scene, ray)`, the loop of the reference's
`LSC.simulate` (`device/lsc.py:349-350`) and of its examples.

There is no Python tracer here: the ray is traced by the HIP engine as a bundle of one with its full history, and the
history is handed back in the reference's form, `[(Ray, Event), ...]` (`follow`) or `(Ray, Event, metadata)` one by one
(`step_forward`).  Same scene semantics as `engine.simulate` (the reference's two tracers differ in documented corners,
SURVEY.md section 5: spectra clamp outside their range, re-emission is sampled per `emit_method` by the kernel's rule).
Random decisions come from the kernel's per-ray stream: `seed` names it; without one it is drawn from numpy's global
generator, as the reference's tracer draws everything from it -- so `numpy.random.seed` makes a sequence of calls
reproducible here too, though not the same photons as the reference's.  For more than a handful of rays call
`engine.simulate(scene, n)` (one launch for all of them) and read `.histories()`."""
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


def step_forward(scene, ray, maxsteps=1000, maxpathlength=np.inf, emit_method="kT", *, seed=None, session=None):
    """Generates `(Ray, Event, metadata)` along one photon's path, as the reference's generator does (:112-273).
    `metadata` holds the node / component names the reference's carries (`hit`, `container`, `adjacent`, `component`,
    `normal` where there is one); the GENERATE row's is None.  `maxpathlength`: the reference ends a photon (KILL) at the
    start of the first step it enters having travelled further than this (:162-172); so does this, on the finished history.
    `session`: an `engine.Session` of the scene to reuse between calls (the scene then stays resident on the GPU)."""
    history = _history(scene, ray, maxsteps, emit_method, seed, session)
    closes_step = {Event.GENERATE, Event.REFLECT, Event.TRANSMIT, Event.EMIT, Event.SCATTER}
    for k, (r, event, metadata) in enumerate(history):
        yield (r, event, None if event == Event.GENERATE else metadata)
        if (event in closes_step and r.travelled > maxpathlength and k + 1 < len(history)):
            container = (history[k + 1][2] or {}).get("container")
            yield (r, Event.KILL, {"maxpathlength": r.travelled, "container": container})
            return


def follow(scene, ray, maxsteps=1000, maxpathlength=np.inf, emit_method="kT", *, seed=None, session=None):
    """One photon's full path: `[(Ray, Event), ...]` (reference :276-328; metadata dropped, as there)."""
    return [(r, event) for r, event, _ in step_forward(scene, ray, maxsteps=maxsteps, maxpathlength=maxpathlength,
                                                       emit_method=emit_method, seed=seed, session=session)]
