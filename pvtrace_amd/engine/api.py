"""`engine.simulate()` — the drop-in entry point, running on MI355X.

Same call signature, result object and `data` dictionary as the reference's
pvtrace/engine/api.py:197-264 (`simulate`, `simulate_stream`, `EngineResult`,
`RecorderResult`, `is_available`).  What differs is everything underneath: the
scene is flattened to SoA tables, uploaded once to HBM, and the whole per-photon
loop runs in the HIP kernel of csrc/pvt_trace_kernel.h; tallies come back as a few
KB.  Keyword-only extras: `device` (GPU index), `emission` ("host" = numpy
sampling like the reference, "device" = sampled on the GPU from per-ray
streams, "auto" = device when the lights allow it) and `emit_seed`.  `workers` is accepted for compatibility and ignored
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


class LazyLogColumns(dict):
    """The reference's `data` dict whose DENSE event-log columns (`recorded * max_events` rows each, pre-filled
    like the reference's) are built when they are first looked at.  What came off the GPU are the rows that were
    written (`packed_rows`: the columns, `row_start`: first packed row of each recorded ray); scattering them into
    twelve dense host arrays is most of the time of a default `simulate()` call and 15 GB of page faults for
    10^6 rays x 128 events -- spent only on the columns somebody indexes.  Everything else behaves like the dict
    the reference returns (iteration, `items()`, equality, pickling materialise what they need)."""

    def __init__(self, eager, builders, row_start, packed_rows):
        super().__init__(eager)
        self._builders = dict(builders)
        for name in self._builders:
            super().__setitem__(name, None)
        self.row_start, self.packed_rows = row_start, packed_rows

    def _build(self, name):
        build = self._builders.pop(name, None)
        if build is not None:
            super().__setitem__(name, build())

    def _build_all(self):
        for name in list(self._builders):
            self._build(name)

    def __getitem__(self, name):
        self._build(name)
        return super().__getitem__(name)

    def __setitem__(self, name, value):
        self._builders.pop(name, None)
        super().__setitem__(name, value)

    def __delitem__(self, name):
        self._builders.pop(name, None)
        super().__delitem__(name)

    def __iter__(self):   # (also keeps dict(self) / {**self} off the C fast path that would copy the placeholders)
        return iter(list(super().keys()))

    def get(self, name, default=None):
        return self[name] if name in self else default

    def pop(self, name, *default):
        self._build(name)
        return super().pop(name, *default)

    def items(self):
        self._build_all()
        return super().items()

    def values(self):
        self._build_all()
        return super().values()

    def copy(self):
        self._build_all()
        return dict(super().items())

    def __eq__(self, other):
        self._build_all()
        return dict.__eq__(self, other)

    __hash__ = None

    def __reduce__(self):
        return (dict, (self.copy(),))


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

    @property
    def packed(self):
        """True when the event log holds only the rows that were written (`simulate(packed_log=True)`):
        the rows of recorded ray j are ``row_start[j] : row_start[j+1]`` instead of the reference's
        ``j*max_events : j*max_events + counts[j]``."""
        return "row_start" in self.data

    def rows_of(self, j):
        """Slice of the event-log columns holding the history of recorded ray j (either layout)."""
        if self.packed:
            return slice(int(self.data["row_start"][j]), int(self.data["row_start"][j + 1]))
        first = j * self.max_events
        return slice(first, first + int(self.data["counts"][j]))

    def event_counts(self):
        """Counter of logged events (recorded rays only)."""
        counts = self.data["counts"]
        if len(counts) == 0:
            return collections.Counter()
        if self.packed or isinstance(self.data, LazyLogColumns):   # the written rows are at hand: no mask over the dense log
            kinds = self.data["kind"] if self.packed else self.data.packed_rows["kind"]
            values, tallies = np.unique(kinds, return_counts=True)
            return collections.Counter({Event(int(v)): int(t) for v, t in zip(values, tallies)})
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
        lazy = isinstance(d, LazyLogColumns)
        if lazy:   # walk the written rows themselves instead of building twelve dense columns for it
            starts, d = d.row_start, d.packed_rows
        for j in range(self.num_recorded):
            rows = slice(int(starts[j]), int(starts[j + 1])) if lazy else self.rows_of(j)
            steps = []
            for row in range(rows.start, rows.stop):
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


GROUP_PHOTONS = 1_000_000   # photons one launch of simulate_stream traces in tally mode (bundles grouped)
_WORKERS_WARNED = False


def _warn_workers(workers):
    """`workers` meant CPU threads in the reference (api.py:201); nothing here runs on CPU threads."""
    global _WORKERS_WARNED
    if workers not in (None, 0, 1) and not _WORKERS_WARNED:
        import warnings

        _WORKERS_WARNED = True
        warnings.warn("pvtrace_amd.engine: `workers` is accepted for compatibility and ignored "
                      "(the trace runs on the GPU; results never depended on it).", stacklevel=3)


def _default_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


_PINNED_FROM = 1 << 20   # bytes: smaller results take the plain pageable road,
_PINNED_UP_TO = 1 << 30  # and so do arrays beyond a gigabyte each (page-locking that much of the host is not this call's to decide)


def to_host(tensor):
    """A device tensor -> a host numpy array.  Results of a megabyte and more land in PINNED host memory from torch's
    caching host allocator: a fresh pageable array is committed page by page while the copy runs (5-8 GB/s measured on
    the GPU box for 56-512 MB, `profiles/r06_host_io.txt`) whereas a pinned block moves at the link's rate (50 GB/s)
    and, once the result that owns it has been dropped, is handed out again by the allocator without being faulted in
    a second time.  The array keeps its block alive (numpy `base`); nothing here is shared between results."""
    nbytes = tensor.numel() * tensor.element_size()
    if nbytes < _PINNED_FROM or nbytes > _PINNED_UP_TO or os.environ.get("PVT_NO_PINNED_RESULTS"):
        return tensor.cpu().numpy()
    import torch

    try:
        host = torch.empty(tensor.shape, dtype=tensor.dtype, pin_memory=True)
    except RuntimeError:   # (no page-locked memory to be had: the pageable road still works)
        return tensor.cpu().numpy()
    host.copy_(tensor)
    return host.numpy()


def download(compiled, tallies, log, n_rays, record_every, max_events, packed=False):
    """Device buffers -> the reference's `data` dict (host numpy).  `packed`: keep only the written
    rows of the event log (plus `row_start`, see `EngineResult.packed`) instead of rebuilding the
    reference's dense `rows = recorded * max_events` arrays."""
    nrec = int(compiled.rec_node.shape[0])
    n_recorded = native.num_recorded(n_rays, record_every)
    rows = n_recorded * max_events
    data = {"counts": (to_host(log["counts"][:n_recorded]) if log is not None
                       else np.zeros(0, dtype=np.int32))}
    if "_ints" in tallies:   # DeviceScene.new_tallies(): distinct | crossings | bins share one buffer
        ints = tallies["_ints"].cpu().numpy()
        pad = max(nrec, 1)
        data["rec_distinct"] = ints[:nrec]
        data["rec_crossings"] = ints[pad:pad + nrec]
        data["rec_bins"] = ints[2 * pad: 2 * pad + int(compiled.total_bins)]
    else:
        data["rec_distinct"] = tallies["rec_distinct"][:nrec].cpu().numpy()
        data["rec_crossings"] = tallies["rec_crossings"][:nrec].cpu().numpy()
        data["rec_bins"] = tallies["rec_bins"][: int(compiled.total_bins)].cpu().numpy()
    data["rec_sums"] = tallies["rec_sums"][: nrec * 8].cpu().numpy().reshape(nrec, 4, 2)
    if log is None or rows == 0:
        for name, dtype, width in native.EVENT_LOG_COLUMNS:
            col = np.zeros(0, dtype=dtype)
            data[name] = col.reshape(0, 3) if width == 3 else col
        return data
    # The log is `max_events` rows per recorded ray but a ray writes only `counts[j]` of them (10 of 128 on the
    # LSC): the written RECORDS (one 128-byte row per event, PvtEventRecords) are gathered on the GPU with one
    # index_select, moved, decoded into the reference's columns and -- unless `packed` -- scattered into host
    # arrays pre-filled like the reference's (`np.zeros` / `np.full(-1)`, _kernel.pyx:1035-1047).  Untouched pages
    # of the zero-filled columns are never committed, as in the reference.
    import torch

    counts = log["counts"][:n_recorded]
    used = int(counts.sum().item())
    if packed:
        starts = np.zeros(n_recorded + 1, dtype=np.int64)
        np.cumsum(data["counts"], out=starts[1:])
        data["row_start"] = starts
    # rows written = event k < counts[j] of recorded ray j, i.e. row j*max_events + k: built from the counts
    # (one entry per written row), not by scanning a mask over every row of the log
    counts64 = counts.to(torch.int64)
    first = torch.cumsum(counts64, 0) - counts64                     # first packed position of each ray
    ray = torch.repeat_interleave(torch.arange(n_recorded, device=counts.device), counts64, output_size=used)
    index = ray * max_events + (torch.arange(used, device=counts.device) - first[ray])
    # decoded into columns on the GPU (slices of the gathered records), so that what crosses PCIe and lands in
    # host memory is already the reference's contiguous arrays
    picked = log["rows"][:rows].index_select(0, index)
    if used * 128 <= (8 << 20):
        # a sampled log (a few thousand rows): one transfer, decoded on the host -- twelve small device kernels and
        # twelve small transfers cost more than the rows themselves
        written = native.decode_records(picked.cpu().numpy())
        picked = None
    i32, f64 = (picked.view(torch.int32), picked.view(torch.float64)) if picked is not None else (None, None)
    if picked is not None:
        columns = {"kind": i32[:, 5].to(torch.uint8), "hit": i32[:, 0], "container": i32[:, 1], "adjacent": i32[:, 2],
                   "component": i32[:, 3], "source": i32[:, 4], "position": f64[:, 3:6], "direction": f64[:, 6:9],
                   "normal": f64[:, 9:12], "wavelength": f64[:, 12], "travelled": f64[:, 13], "duration": f64[:, 14]}
        written = {name: to_host(col.contiguous()) for name, col in columns.items()}
        del picked, i32, f64, columns
    if packed:
        data.update(written)
        return data
    index_host = to_host(index)

    def dense(name, dtype, width):
        def build():
            fill = -1 if name in native._ID_COLUMNS else 0
            shape = (rows, 3) if width == 3 else (rows,)
            # zero columns stay uncommitted (calloc) where no row was written; the -1 fill is the reference's cost too
            host = np.zeros(shape, dtype=dtype) if fill == 0 else np.full(shape, fill, dtype=dtype)
            if used:
                host[index_host] = written[name]
            return host
        return build

    starts = np.zeros(n_recorded + 1, dtype=np.int64)
    np.cumsum(data["counts"], out=starts[1:])
    return LazyLogColumns(data, {name: dense(name, dtype, width) for name, dtype, width in native.EVENT_LOG_COLUMNS},
                          starts, written)


# Scenes recently resident on a GPU, kept for the next `simulate` call on the same scene: flattening is 0.1 ms, but
# creating the device scene (a dozen allocations, the packed tables, for meshes the BVH) and destroying it again
# (hipFree synchronises the device) cost 0.4 ms per call -- as much as tracing 10^6 photons of the headline scene.
# The key is a digest of every flat table (and of the emitter tables), so a scene edited between two calls is a
# different scene; a resident scene is handed to ONE session at a time.  PVT_NO_SCENE_CACHE=1 switches it off.
_RESIDENT = []        # [(key, DeviceScene)], most recently released last
_RESIDENT_MAX = 4
_RESIDENT_LOCK = __import__("threading").Lock()


def _scene_key(compiled, emitter, device):
    import hashlib

    h = hashlib.blake2b(digest_size=16)
    h.update(repr(int(device)).encode())
    for name in compiled.TABLE_FIELDS:
        a = np.ascontiguousarray(getattr(compiled, name))
        h.update(a.dtype.str.encode()); h.update(repr(a.shape).encode()); h.update(a.data if a.size else b"")
    h.update(repr((int(compiled.root_id), int(compiled.total_bins))).encode())
    if emitter is not None:
        for name in ("wl_type", "wl_value", "wl_spec_start", "wl_spec_n", "pos_type", "pos_param", "dir_type",
                     "dir_param", "light_to_world", "spec_x", "spec_cdf"):
            a = np.ascontiguousarray(getattr(emitter, name))
            h.update(repr(a.shape).encode()); h.update(a.data if a.size else b"")
    return h.digest()


def _acquire_scene(compiled, emitter, device):
    key = None
    if not os.environ.get("PVT_NO_SCENE_CACHE"):
        key = _scene_key(compiled, emitter, device)
        with _RESIDENT_LOCK:
            for k, (have, dscene) in enumerate(_RESIDENT):
                if have == key:
                    del _RESIDENT[k]
                    dscene.compiled = compiled
                    return key, dscene
    return key, native.DeviceScene(compiled, device=device, emitter=emitter)


def _release_scene(key, dscene):
    if key is None or dscene.handle is None:
        dscene.close()
        return
    evicted = None
    dscene.trim()   # a scene put aside keeps its tables only: no staging buffers, nobody's parked photons
    with _RESIDENT_LOCK:
        _RESIDENT.append((key, dscene))
        if len(_RESIDENT) > _RESIDENT_MAX:
            evicted = _RESIDENT.pop(0)[1]
    if evicted is not None:
        evicted.close()


def release_resident_scenes():
    """Free every scene kept on a GPU for reuse, and the device block the host-buffer entry keeps between calls
    (also run at interpreter exit)."""
    with _RESIDENT_LOCK:
        held = [d for _, d in _RESIDENT]
        del _RESIDENT[:]
    for d in held:
        d.close()
    if native._lib is not None:
        native._lib.pvt_release_cached_memory()
    # the pinned blocks dropped results gave back to torch's host allocator (`to_host`); blocks of results still alive stay
    torch = __import__("sys").modules.get("torch")
    empty = getattr(getattr(torch, "_C", None), "_host_emptyCache", None) if torch is not None else None
    if empty is not None:
        try:
            empty()
        except Exception:   # noqa: BLE001 -- a private entry of torch: its absence or refusal is not this call's failure
            pass


__import__("atexit").register(release_resident_scenes)

_SIDE_STREAMS = {}   # device index -> [torch.cuda.Stream, torch.cuda.Stream]
_SIDE_STREAMS_LOCK = __import__("threading").Lock()


class Session:
    """A scene compiled and resident on one GPU, ready to trace any number of bundles.

    `simulate` opens one per call; `simulate_stream` keeps one for the whole stream, so the
    flattening, the table upload and the buffer allocations happen once, not once per bundle
    (the reference re-flattens per `simulate` call, api.py:222, which costs it nothing next to
    its trace time; here a 50 000-ray bundle traces in well under a millisecond)."""

    def __init__(self, scene, device=None, emission="auto"):
        if emission not in ("auto", "host", "device"):
            raise ValueError("emission must be 'auto', 'host' or 'device'")
        from pvtrace_amd.engine import emit as emit_mod
        from pvtrace_amd.engine.compiler import UnsupportedSceneError

        self.scene = scene
        self.compiled = compile_scene(scene)
        self.device = _default_device() if device is None else device
        self.emitter = None
        if emission in ("auto", "device"):
            try:
                self.emitter = emit_mod.EmitterTables(scene, strict=True)
            except UnsupportedSceneError:
                if emission == "device":
                    raise
        self.emission = "device" if self.emitter is not None else "host"
        self._key, self.dscene = _acquire_scene(self.compiled, self.emitter, self.device)
        self._slots = []
        self._submitted = 0

    def close(self):
        if self.dscene is not None:
            import torch

            if self._slots:   # nothing of this session may still be running on a scene the next one will be handed
                for slot in self._slots:
                    slot["stream"].synchronize()
            _release_scene(self._key, self.dscene)
            self.dscene = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def submit(self, num_rays, seed, maxsteps=1000, max_events=128, emit_method="kT", record_every=1,
               emit_seed=None, ray_offset=0, workgroups_per_cu=0, host_rays=None, tally_bundle=0,
               packed_log=False):
        """Enqueue one bundle on one of two HIP streams and return a handle for `collect`.
        Two bundles may be in flight: the next one is traced while the caller consumes the
        previous result.  `host_rays`: (positions, directions, wavelengths, sources) already emitted
        on the host (a shard of a bundle emitted once for several GPUs).  `tally_bundle` = m > 0
        (tally mode): the rays are consecutive bundles of m rays traced by ONE launch, each tallied on
        its own (PvtTraceParams.tally_bundle); `collect_bundles` returns one result per bundle."""
        import torch

        from pvtrace_amd.engine import emit as emit_mod

        if emit_method not in EMIT_METHODS:
            raise ValueError(f"emit_method must be one of {sorted(EMIT_METHODS)}")
        if tally_bundle and int(record_every) != 0:
            raise ValueError("tally_bundle needs record_every == 0 (one tally set per bundle, no event log)")
        device, dscene = self.device, self.dscene
        with torch.cuda.device(device):
            if host_rays is not None:
                pos, direc, wl, sources = host_rays
                dev = torch.device("cuda", device)
                rays = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                             for a in (pos, direc, wl))
            elif self.emission == "device":
                if emit_seed is None:
                    emit_seed = np.random.randint(0, 2 ** 31 - 1)
                sources = emit_mod.sources_for(self.scene, num_rays, offset=ray_offset)
                rays = None
            else:
                pos, direc, wl, sources = emit_mod.emit_bundle(self.scene, num_rays, seed=emit_seed)
                dev = torch.device("cuda", device)
                rays = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                             for a in (pos, direc, wl))
            if not self._slots:
                # the two side streams are shared by every Session on this device: torch's caching
                # allocator pools memory per stream, so fresh streams per call would re-malloc
                # (and later free) every buffer, multi-GB event logs included
                with _SIDE_STREAMS_LOCK:
                    streams = _SIDE_STREAMS.setdefault(device, [])
                    while len(streams) < 2:
                        streams.append(torch.cuda.Stream(device=device))
                self._slots = [{"stream": st, "tallies": dscene.new_tallies()} for st in streams[:2]]
            slot = self._slots[self._submitted % 2]
            self._submitted += 1
            sets = -(-int(num_rays) // int(tally_bundle)) if tally_bundle else 1
            if slot["tallies"].get("sets", 1) < sets:
                slot["tallies"] = dscene.new_tallies(sets=sets)
            stream, tallies = slot["stream"], slot["tallies"]
            stream.wait_stream(torch.cuda.current_stream(device))   # ray upload, earlier downloads
            with torch.cuda.stream(stream):
                tallies["_ints"].zero_()
                tallies["_sums"].zero_()
                log = (dscene.new_event_log(num_rays, record_every, max_events)
                       if record_every > 0 else None)
                start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record(stream)
                tic = time.perf_counter()
                dscene.trace(rays, num_rays, int(seed), tallies, log=log, ray_offset=ray_offset,
                             emit_seed=int(emit_seed or 0), record_every=int(record_every),
                             maxsteps=int(maxsteps), max_events=int(max_events),
                             emit_method=EMIT_METHODS[emit_method], stream=stream.cuda_stream,
                             workgroups_per_cu=workgroups_per_cu, tally_bundle=int(tally_bundle),
                             log_prefill=False)   # `download` reads written rows only (or repairs the rest)
                stop.record(stream)
        return {"stream": stream, "tallies": tallies, "log": log, "events": (start, stop), "tic": tic, "launch_no": self._submitted,
                "rays": rays, "sources": sources, "num_rays": num_rays, "record_every": record_every,
                "max_events": max_events, "tally_bundle": int(tally_bundle), "packed_log": bool(packed_log)}

    def collect(self, pending, wall_clock=False):
        """Wait for a submitted bundle and bring its results to the host -> `EngineResult`.
        `elapsed` is the trace alone, like the reference's (api.py:232-245): HIP-event time, or
        host wall time around launch + completion when `wall_clock`."""
        import torch

        if pending.get("tally_bundle"):
            raise ValueError("this launch was submitted with tally_bundle: use collect_bundles() (one result per bundle)")
        with torch.cuda.device(self.device):
            pending["stream"].synchronize()
            wall = time.perf_counter() - pending["tic"]
            # The launch as the GPU saw it (the kernel's own 100 MHz stamps) when this bundle is still the last launch on
            # its stream; HIP events otherwise.  The events bracket host code -- ctypes, the library's planning -- and a
            # host thread descheduled between them reads as tens of milliseconds of "kernel" (profiles/r06_e2e_outlier.txt).
            kernel_ms = pending["events"][0].elapsed_time(pending["events"][1])
            if self._submitted - pending.get("launch_no", -2) < 2 and pending["num_rays"] > 0:   # (two streams, used in turn)
                kernel_ms = min(kernel_ms, self.dscene.launch_span_ms(pending["stream"].cuda_stream) or kernel_ms)
            with torch.cuda.stream(pending["stream"]):
                data = download(self.compiled, pending["tallies"], pending["log"], pending["num_rays"],
                                pending["record_every"], pending["max_events"], packed=pending.get("packed_log", False))
        return EngineResult(self.compiled, data, pending["sources"], pending["max_events"],
                            pending["record_every"], wall if wall_clock else kernel_ms * 1e-3,
                            kernel_ms=kernel_ms)

    def collect_bundles(self, pending):
        """Wait for a launch submitted with `tally_bundle` -> one `EngineResult` per bundle, in order
        (one download for all of them)."""
        import torch

        m = pending["tally_bundle"]
        n = pending["num_rays"]
        sets = -(-n // m)
        c = self.compiled
        nrec = int(c.rec_node.shape[0])
        pad, nbins = max(nrec, 1), int(c.total_bins)
        t = pending["tallies"]
        with torch.cuda.device(self.device):
            pending["stream"].synchronize()
            kernel_ms = pending["events"][0].elapsed_time(pending["events"][1])
            with torch.cuda.stream(pending["stream"]):
                ints = t["_ints"][: sets * t["stride_i64"]].cpu().numpy().reshape(sets, t["stride_i64"])
                sums = t["_sums"][: sets * t["stride_f64"]].cpu().numpy().reshape(sets, t["stride_f64"])
        empty = {"counts": np.zeros(0, dtype=np.int32)}
        for name, dtype, width in native.EVENT_LOG_COLUMNS:
            col = np.zeros(0, dtype=dtype)
            empty[name] = col.reshape(0, 3) if width == 3 else col
        results = []
        sources = pending["sources"]
        for j in range(sets):
            lo, hi = j * m, min((j + 1) * m, n)
            data = dict(empty)
            data["rec_distinct"] = ints[j, :nrec]
            data["rec_crossings"] = ints[j, pad:pad + nrec]
            data["rec_bins"] = ints[j, 2 * pad: 2 * pad + nbins]
            data["rec_sums"] = sums[j, : nrec * 8].reshape(nrec, 4, 2)
            results.append(EngineResult(c, data, sources[lo:hi], pending["max_events"], 0,
                                        kernel_ms * 1e-3 * (hi - lo) / n, kernel_ms=kernel_ms * (hi - lo) / n))
        return results

    def run(self, num_rays, seed, **kwargs):
        """Trace one bundle -> `EngineResult` (submit + collect)."""
        import torch

        if kwargs.get("tally_bundle"):
            raise ValueError("run() returns ONE result; submit(..., tally_bundle=m) + collect_bundles() for a group")
        torch.cuda.synchronize(self.device)
        return self.collect(self.submit(num_rays, seed, **kwargs), wall_clock=True)


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
    emission="auto",
    emit_seed=None,
    ray_offset=0,
    devices=None,
    packed_log=False,
):
    """Trace `num_rays` through `scene` on the GPU.

    `packed_log`: return the event log PACKED -- only the rows that were written, with `data["row_start"]`
    giving each recorded ray's slice (`EngineResult.rows_of`, `.histories()` and `.event_counts()` work on
    either layout) -- instead of the reference's dense `recorded * max_events` rows, of which a ray fills a
    dozen: with the defaults (`record_every=1`, `max_events=128`) the dense arrays are 117 bytes x 128 per ray
    and rebuilding them on the host costs far more than the trace.

    `devices`: a list of GPU ids (repeats allowed) — the bundle is split over them by contiguous
    index range inside this one call (one host thread and one resident scene per entry); ray i keeps
    the stream `seed + ray_offset + i`, so the result does not depend on the list.

    Recorders attached to nodes tally every ray; full event histories are kept
    for every `record_every`-th ray (all when 1, none when 0).  Raises
    `UnsupportedSceneError` for scenes that cannot be flattened, `ValueError`
    for a bad `emit_method`, `EngineUnavailableError` without a GPU.

    `emission`: "device" samples the lights on the GPU from per-ray streams (seeded by
    `emit_seed`), "host" samples them with numpy like the reference's `emit_bundle`
    (emit.py:92-134) and uploads the rays; "auto" (default) uses the device whenever every
    light is built from the library's masks and falls back to the host for custom delegates.
    """
    if emit_method not in EMIT_METHODS:
        raise ValueError(f"emit_method must be one of {sorted(EMIT_METHODS)}")
    if seed is None:
        seed = np.random.randint(0, 2 ** 31 - 1)
    _warn_workers(workers)
    if devices is not None:
        if device is not None:
            raise ValueError("give `device` or `devices`, not both")
        if packed_log:
            raise ValueError("packed_log is per device; use `device`")
        return _simulate_on_devices(scene, num_rays, seed, list(devices), maxsteps, max_events, emit_method,
                                    record_every, emission, emit_seed, ray_offset)
    with Session(scene, device=device, emission=emission) as session:
        return session.run(num_rays, seed, maxsteps=maxsteps, max_events=max_events,
                           emit_method=emit_method, record_every=record_every,
                           emit_seed=emit_seed, ray_offset=ray_offset, packed_log=packed_log)


def merge_shards(results):
    """EngineResults of consecutive index-range shards of ONE bundle (boundaries on multiples of
    `record_every`) -> the EngineResult of the whole bundle: tallies add, event logs and sources
    concatenate in shard order."""
    first = results[0]
    data = {}
    for key in ("rec_distinct", "rec_crossings", "rec_sums", "rec_bins"):
        total = np.array(first.data[key], copy=True)
        for r in results[1:]:
            total += r.data[key]
        data[key] = total
    if all(isinstance(r.data, LazyLogColumns) for r in results):
        # the shards' dense columns stay unbuilt: the merged ones are built (shard by shard, then joined) on demand
        data["counts"] = np.concatenate([r.data["counts"] for r in results])
        starts = np.zeros(len(data["counts"]) + 1, dtype=np.int64)
        np.cumsum(data["counts"], out=starts[1:])
        packed_rows = {name: np.concatenate([r.data.packed_rows[name] for r in results])
                       for name in first.data.packed_rows}
        builders = {name: (lambda name=name: np.concatenate([r.data[name] for r in results]))
                    for name, _, _ in native.EVENT_LOG_COLUMNS}
        data = LazyLogColumns(data, builders, starts, packed_rows)
    else:
        for key in first.data:
            if key not in data:
                data[key] = np.concatenate([r.data[key] for r in results])
    from pvtrace_amd.engine.emit import ChainedSources

    sources = ChainedSources([r.sources for r in results])
    kernel_ms = [r.kernel_ms for r in results if r.kernel_ms is not None]
    return EngineResult(first.compiled, data, sources, first.max_events, first.record_every,
                        max(r.elapsed for r in results), kernel_ms=max(kernel_ms) if kernel_ms else None)


def _simulate_on_devices(scene, num_rays, seed, devices, maxsteps, max_events, emit_method, record_every,
                         emission, emit_seed, ray_offset):
    """One bundle over several GPUs of this process (reference: one bundle over OpenMP threads,
    _kernel.pyx:1074-1095; either way the outcome is independent of the split)."""
    import concurrent.futures

    from pvtrace_amd.engine import emit as emit_mod
    from pvtrace_amd.engine.distributed import shard_range

    if not devices:
        raise ValueError("`devices` is empty")
    sessions = []
    try:
        for d in devices:   # one at a time: a failure on the k-th device must not leak the first k-1 resident scenes
            sessions.append(Session(scene, device=d, emission=emission))
        host = None
        if sessions[0].emission == "host":   # the lights are sampled once, in the reference's global order
            host = emit_mod.emit_bundle(scene, num_rays, seed=emit_seed)
        elif emit_seed is None:
            emit_seed = np.random.randint(0, 2 ** 31 - 1)
        spans = [shard_range(num_rays, g, len(devices), align=record_every) for g in range(len(devices))]

        def run(g):
            lo, hi = spans[g]
            if hi <= lo:
                return None
            kw = dict(maxsteps=maxsteps, max_events=max_events, emit_method=emit_method,
                      record_every=record_every, ray_offset=ray_offset + lo)
            if host is not None:
                pos, direc, wl, sources = host
                kw["host_rays"] = (pos[lo:hi], direc[lo:hi], wl[lo:hi], sources[lo:hi])
                return sessions[g].run(hi - lo, int(seed), **kw)
            return sessions[g].run(hi - lo, int(seed), emit_seed=emit_seed, **kw)

        with concurrent.futures.ThreadPoolExecutor(max_workers=len(devices)) as pool:
            results = [r for r in pool.map(run, range(len(devices))) if r is not None]
    finally:
        for session in sessions:
            session.close()
    if not results:   # num_rays == 0
        with Session(scene, device=devices[0], emission=emission) as session:
            return session.run(0, int(seed), maxsteps=maxsteps, max_events=max_events, emit_method=emit_method,
                               record_every=record_every, emit_seed=emit_seed, ray_offset=ray_offset)
    return merge_shards(results)


def simulate_stream(scene, num_rays, bundle=50000, seed=None, **kwargs):
    """Trace in bundles, yielding (EngineResult, rays_traced_so_far).

    Bundle b uses per-ray seeds ``seed + traced + i`` (reference api.py:249-264),
    so the union of the streamed results equals one `simulate` call; sum the
    `rec_*` arrays to accumulate.  The scene is flattened and uploaded once for the
    whole stream."""
    if seed is None:
        seed = np.random.randint(0, 2 ** 31 - 1)
    emit_seed = kwargs.pop("emit_seed", None)
    _warn_workers(kwargs.pop("workers", None))
    base_offset = int(kwargs.pop("ray_offset", 0))   # global index of the stream's first ray (sharded jobs)
    devices = kwargs.pop("devices", None)
    device = kwargs.pop("device", None)
    if devices is not None and device is not None:
        raise ValueError("give `device` or `devices`, not both")
    emission = kwargs.pop("emission", "auto")
    # one resident scene per GPU; bundle b goes to GPU b mod N, two bundles in flight per GPU, results are
    # yielded in bundle order
    sessions = []
    state = {"emit_seed": emit_seed}

    # Tally mode: a launch serves a GROUP of consecutive bundles (one tally set each), about a million
    # photons' worth -- a 50 000-photon bundle alone leaves most of an MI355X idle and pays the launch and
    # the drain of a kernel for very little.  The bundles of a group are yielded one by one, as ever.
    per_group = 1
    if int(kwargs.get("record_every", 1)) == 0 and bundle > 0:
        per_group = max(1, min(1024, GROUP_PHOTONS // int(bundle)))

    def submit(index, traced):
        session = sessions[index % len(sessions)]
        n = min(bundle * per_group, num_rays - traced)
        group = {"tally_bundle": bundle} if per_group > 1 else {}
        if session.emission == "device":
            # one emission stream for the whole job: ray i of the job is the same photon
            # whatever the bundle size
            return session, session.submit(n, int(seed), emit_seed=state["emit_seed"], ray_offset=base_offset + traced,
                                           workgroups_per_cu=3, **group, **kwargs), n     # two launches in flight
        if per_group > 1:   # the lights are sampled bundle by bundle, in the reference's order, then traced together
            from pvtrace_amd.engine import emit as emit_mod

            starts = range(0, n, bundle)
            if state["emit_seed"] is None:   # the global numpy generator: drawn in order
                parts = [emit_mod.emit_bundle(scene, min(bundle, n - at), seed=None) for at in starts]
            else:   # every bundle has its own generator: the bundles of the group are sampled side by side
                parts = emit_mod.emit_bundles(scene, [min(bundle, n - at) for at in starts],
                                              [int(state["emit_seed"]) + traced + at for at in starts])
            host = tuple(np.concatenate([p[k] for p in parts]) for k in range(3)) + (emit_mod.ChainedSources([p[3] for p in parts]),)
            return session, session.submit(n, int(seed) + traced, ray_offset=base_offset, workgroups_per_cu=3,
                                           host_rays=host, **group, **kwargs), n
        bundle_emit_seed = None if state["emit_seed"] is None else int(state["emit_seed"]) + traced
        return session, session.submit(n, int(seed) + traced, emit_seed=bundle_emit_seed, ray_offset=base_offset,
                                       workgroups_per_cu=3, **kwargs), n

    in_flight = collections.deque()
    submitted, index, traced = 0, 0, 0
    try:
        for d in (devices if devices is not None else [device]):   # inside the try: a failing k-th Session must not
            sessions.append(Session(scene, device=d, emission=emission))   # leak the k-1 resident scenes before it
        if state["emit_seed"] is None:
            # one draw from the global generator seeds the whole stream's emission (reproducible under
            # np.random.seed): device emission needs a seed anyway, and on the host the bundles of a group can then
            # be sampled side by side from their own generators instead of one after the other from the global one
            state["emit_seed"] = np.random.randint(0, 2 ** 31 - 1)
        window = 2 * len(sessions)
        while traced < num_rays or in_flight:
            # launches k+1 .. are traced while the consumer works on the bundles of launch k
            while submitted < num_rays and len(in_flight) < window:
                item = submit(index, submitted)
                in_flight.append(item)
                submitted += item[2]
                index += 1
            if not in_flight:
                break
            session, handle, n = in_flight.popleft()
            if handle["tally_bundle"]:
                for result in session.collect_bundles(handle):
                    traced += result.num_rays
                    yield result, traced
            else:
                result = session.collect(handle)
                traced += n
                yield result, traced
    finally:
        for session in sessions:
            session.close()
