"""`engine.simulate()` — the drop-in entry point, running on MI355X.

Same call signature, result object and `data` dictionary as the reference's
pvtrace/engine/api.py:197-264 (`simulate`, `simulate_stream`, `EngineResult`,
`RecorderResult`, `is_available`).  What differs is everything underneath: the
scene is flattened to SoA tables, uploaded once to HBM, and the whole per-photon
loop runs in the HIP kernel of csrc/pvt_trace.hip; tallies come back as a few
KB.  Keyword-only extras: `device` (GPU index), `emission` ("host" = numpy
sampling like the reference, "device" = sampled on the GPU from per-ray
streams) and `emit_seed`.  `workers` is accepted for compatibility and ignored
(there are no CPU tracing threads; there is no CPU path at all).
"""
import collections
import os
import time

import numpy as np

from pvtrace_amd.engine import native
from pvtrace_amd.engine.compiler import EMIT_METHODS, compile_scene
from pvtrace_amd.engine.recorder import Heatmap
from pvtrace_amd.light import Event, Ray

# moment accumulators kept for every recorder, in kernel order
MOMENT_PROPERTIES = ("wavelength", "angle", "duration", "pathlength")


def is_available() -> bool:
    """True when the HIP engine is built and a GPU is visible."""
    return native.is_available()


class RecorderResult:
    """Tallied statistics of one recorder (reference api.py:26-78).

    `rays`: distinct rays; `crossings`: every matching interaction; moments
    and histograms are per distinct ray."""

    def __init__(self, spec, rays, crossings, moments, bins):
        self.spec = spec
        self.rays = int(rays)
        self.crossings = int(crossings)
        self._moments = moments  # (4, 2): sum, sum of squares
        self._bins = bins        # one array per histogram of the spec

    def _row(self, prop):
        return self._moments[MOMENT_PROPERTIES.index(prop)]

    def mean(self, prop):
        row = self._row(prop)
        return float("nan") if self.rays == 0 else row[0] / self.rays

    def std(self, prop):
        row = self._row(prop)
        if self.rays == 0:
            return float("nan")
        mean = row[0] / self.rays
        return float(np.sqrt(max(row[1] / self.rays - mean * mean, 0.0)))

    def error(self, prop):
        if self.rays == 0:
            return float("nan")
        return self.std(prop) / np.sqrt(self.rays)

    def histogram(self, index=0):
        """(edges, counts), or (edges_a, edges_b, counts[a, b]) for a Heatmap."""
        spec = self.spec.histograms[index]
        values = self._bins[index]
        if isinstance(spec, Heatmap):
            ea = np.linspace(spec.a.start, spec.a.stop, spec.a.bins + 1)
            eb = np.linspace(spec.b.start, spec.b.stop, spec.b.bins + 1)
            return ea, eb, values.reshape(spec.a.bins, spec.b.bins)
        return np.linspace(spec.start, spec.stop, spec.bins + 1), values

    def __repr__(self):
        return (f"RecorderResult({self.spec.name!r}, rays={self.rays}, "
                f"crossings={self.crossings})")


class EngineResult:
    """Outcome of one bundle (reference api.py:81-194).

    `data` has the reference's keys/dtypes: counts, rec_distinct,
    rec_crossings, rec_sums (R,4,2), rec_bins, and the event-log columns kind,
    hit, container, adjacent, component, source, position, direction, normal,
    wavelength, travelled, duration (row of event k of recorded ray j =
    j*max_events + k)."""

    def __init__(self, compiled, data, sources, max_events, record_every, elapsed,
                 kernel_ms=None):
        self.compiled = compiled
        self.data = data
        self.sources = sources
        self.max_events = max_events
        self.record_every = record_every
        self.elapsed = elapsed
        self.kernel_ms = kernel_ms

    @property
    def num_rays(self):
        return len(self.sources)

    @property
    def num_recorded(self):
        return len(self.data["counts"])

    @property
    def recorded_indices(self):
        if self.record_every <= 0:
            return np.zeros(0, dtype=np.int64)
        return np.arange(0, self.num_rays, self.record_every, dtype=np.int64)

    @property
    def recorders(self):
        c = self.compiled
        out = {}
        for r, spec in enumerate(c.recorder_specs):
            first = c.rec_hist_start[r]
            bins = []
            for h in range(len(spec.histograms)):
                lo = c.hist_offset[first + h]
                size = c.hist_na[first + h] * c.hist_nb[first + h]
                bins.append(self.data["rec_bins"][lo:lo + size])
            out[spec.name] = RecorderResult(
                spec, self.data["rec_distinct"][r], self.data["rec_crossings"][r],
                self.data["rec_sums"][r], bins)
        return out

    def event_counts(self):
        """Counter of logged events (recorded rays only)."""
        counts = self.data["counts"]
        if len(counts) == 0:
            return collections.Counter()
        kinds = self.data["kind"].reshape(self.num_recorded, self.max_events)
        valid = np.arange(self.max_events)[None, :] < counts[:, None]
        values, tallies = np.unique(kinds[valid], return_counts=True)
        return collections.Counter({Event(int(v)): int(t) for v, t in zip(values, tallies)})

    def _node(self, index):
        return self.compiled.node_names[index] if index >= 0 else None

    def _component(self, index):
        return self.compiled.component_names[index] if index >= 0 else None

    def histories(self):
        """One list of (Ray, Event, metadata) per recorded ray."""
        d = self.data
        which = self.recorded_indices
        for j in range(self.num_recorded):
            first = j * self.max_events
            steps = []
            for row in range(first, first + int(d["counts"][j])):
                sid = int(d["source"][row])
                source = self.sources[int(which[j])] if sid < 0 else self._component(sid)
                ray = Ray(
                    position=tuple(d["position"][row].tolist()),
                    direction=tuple(d["direction"][row].tolist()),
                    wavelength=float(d["wavelength"][row]),
                    travelled=float(d["travelled"][row]),
                    duration=float(d["duration"][row]),
                    source=source,
                )
                event = Event(int(d["kind"][row]))
                meta = {
                    "hit": self._node(int(d["hit"][row])),
                    "container": self._node(int(d["container"][row])),
                    "adjacent": self._node(int(d["adjacent"][row])),
                    "component": self._component(int(d["component"][row])),
                }
                if event in (Event.REFLECT, Event.TRANSMIT):
                    meta["normal"] = tuple(d["normal"][row].tolist())
                steps.append((ray, event, meta))
            yield steps


def _default_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


def download(compiled, tallies, log, n_rays, record_every, max_events):
    """Device buffers -> the reference's `data` dict (host numpy)."""
    nrec = int(compiled.rec_node.shape[0])
    n_recorded = native.num_recorded(n_rays, record_every)
    rows = n_recorded * max_events
    data = {
        "counts": (log["counts"][:n_recorded].cpu().numpy() if log is not None
                   else np.zeros(0, dtype=np.int32)),
        "rec_distinct": tallies["rec_distinct"][:nrec].cpu().numpy(),
        "rec_crossings": tallies["rec_crossings"][:nrec].cpu().numpy(),
        "rec_sums": tallies["rec_sums"][: nrec * 8].cpu().numpy().reshape(nrec, 4, 2),
        "rec_bins": tallies["rec_bins"][: int(compiled.total_bins)].cpu().numpy(),
    }
    for name, dtype, width in native.EVENT_LOG_COLUMNS:
        if log is not None and rows > 0:
            col = log[name][: rows * width].cpu().numpy()
        else:
            col = np.zeros(0, dtype=dtype)
        data[name] = col.reshape(rows, 3) if width == 3 else col
    return data


def simulate(
    scene,
    num_rays,
    seed=None,
    workers=None,
    maxsteps=1000,
    max_events=128,
    emit_method="kT",
    record_every=1,
    *,
    device=None,
    emission="host",
    emit_seed=None,
    ray_offset=0,
):
    """Trace `num_rays` through `scene` on the GPU.

    Recorders attached to nodes tally every ray; full event histories are kept
    for every `record_every`-th ray (all when 1, none when 0).  Raises
    `UnsupportedSceneError` for scenes that cannot be flattened, `ValueError`
    for a bad `emit_method`, `EngineUnavailableError` without a GPU.
    """
    if emit_method not in EMIT_METHODS:
        raise ValueError(f"emit_method must be one of {sorted(EMIT_METHODS)}")
    if emission not in ("host", "device"):
        raise ValueError("emission must be 'host' or 'device'")
    compiled = compile_scene(scene)
    if seed is None:
        seed = np.random.randint(0, 2 ** 31 - 1)
    if device is None:
        device = _default_device()

    import torch

    from pvtrace_amd.engine import emit as emit_mod

    emitter = None
    if emission == "device":
        emitter = emit_mod.EmitterTables(scene, strict=True)
        if emit_seed is None:
            emit_seed = np.random.randint(0, 2 ** 31 - 1)
        sources = emit_mod.sources_for(scene, num_rays)
        rays = None
    dscene = native.DeviceScene(compiled, device=device, emitter=emitter)
    try:
        with torch.cuda.device(device):
            if emission == "host":
                pos, direc, wl, sources = emit_mod.emit_bundle(scene, num_rays, seed=emit_seed)
                dev = torch.device("cuda", device)
                rays = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                             for a in (pos, direc, wl))
            tallies = dscene.new_tallies()
            log = (dscene.new_event_log(num_rays, record_every, max_events)
                   if record_every > 0 else None)
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(device)
            tic = time.perf_counter()
            start.record()
            dscene.trace(rays, num_rays, int(seed), tallies, log=log, ray_offset=ray_offset,
                         emit_seed=int(emit_seed or 0), record_every=int(record_every),
                         maxsteps=int(maxsteps), max_events=int(max_events),
                         emit_method=EMIT_METHODS[emit_method])
            stop.record()
            torch.cuda.synchronize(device)
            elapsed = time.perf_counter() - tic
            kernel_ms = start.elapsed_time(stop)
            data = download(compiled, tallies, log, num_rays, record_every, max_events)
    finally:
        dscene.close()
    return EngineResult(compiled, data, sources, max_events, record_every, elapsed,
                        kernel_ms=kernel_ms)


def simulate_stream(scene, num_rays, bundle=50000, seed=None, **kwargs):
    """Trace in bundles, yielding (EngineResult, rays_traced_so_far).

    Bundle b uses per-ray seeds ``seed + traced + i`` (reference api.py:249-264),
    so the union of the streamed results equals one `simulate` call; sum the
    `rec_*` arrays to accumulate."""
    if seed is None:
        seed = np.random.randint(0, 2 ** 31 - 1)
    emit_seed = kwargs.pop("emit_seed", None)
    traced = 0
    while traced < num_rays:
        n = min(bundle, num_rays - traced)
        extra = {}
        if kwargs.get("emission") == "device":
            extra = {"emit_seed": emit_seed, "ray_offset": traced}
            result = simulate(scene, n, seed=int(seed), **kwargs, **extra)
        else:
            bundle_emit_seed = None if emit_seed is None else int(emit_seed) + traced
            result = simulate(scene, n, seed=int(seed) + traced, emit_seed=bundle_emit_seed,
                              **kwargs)
        traced += n
        yield result, traced
