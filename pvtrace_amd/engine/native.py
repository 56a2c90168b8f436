"""ctypes binding to libpvtrace_hip.so (the C ABI in include/pvtrace_hip.h).

This is the only place the Python host touches the native engine.  There is NO
CPU fallback: if the shared library is missing, or no GPU is visible, tracing
raises `EngineUnavailableError` — it never silently routes somewhere else.

`torch` is used purely as plumbing: device buffers are torch tensors (so they
can be all-reduced with torch.distributed over RCCL) and the kernel is enqueued
on torch's current HIP stream.  Pointers and sizes are all the library sees.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get(  # PVT_LIB: developer override (ablation builds); never a CPU path
    "PVT_LIB", os.path.join(os.path.dirname(_HERE), "csrc", "libpvtrace_hip.so"))

_p_i32 = C.POINTER(C.c_int32)
_p_f64 = C.POINTER(C.c_double)
_p_i64 = C.POINTER(C.c_int64)
_p_u8 = C.POINTER(C.c_uint8)


class EngineUnavailableError(RuntimeError):
    """The HIP engine cannot run here (library not built, or no MI355X visible)."""


class PvtSceneTables(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("root_id", C.c_int32), ("n_components", C.c_int32),
        ("n_abs", C.c_int32), ("n_ems", C.c_int32), ("n_recorders", C.c_int32),
        ("n_hists", C.c_int32), ("total_bins", C.c_int32), ("n_coatings", C.c_int32),
        ("reserved0", C.c_int32),
        ("geom_type", _p_i32), ("geom_params", _p_f64), ("local_to_world", _p_f64),
        ("world_to_local", _p_f64), ("refractive_index", _p_f64), ("surface_type", _p_i32),
        ("comp_start", _p_i32), ("comp_count", _p_i32), ("coat_start", _p_i32),
        ("coat_count", _p_i32),
        ("comp_type", _p_i32), ("comp_qy", _p_f64), ("comp_tau_rad", _p_f64),
        ("comp_tau_nr", _p_f64), ("comp_phase_type", _p_i32), ("comp_phase_param", _p_f64),
        ("comp_abs_start", _p_i32), ("comp_abs_n", _p_i32), ("comp_ems_start", _p_i32),
        ("comp_ems_n", _p_i32),
        ("abs_x", _p_f64), ("abs_y", _p_f64), ("ems_x", _p_f64), ("ems_cdf", _p_f64),
        ("rec_node", _p_i32), ("rec_event", _p_i32), ("rec_has_facet", _p_i32),
        ("rec_facet", _p_f64), ("rec_atol", _p_f64), ("rec_hist_start", _p_i32),
        ("rec_hist_n", _p_i32),
        ("hist_prop_a", _p_i32), ("hist_prop_b", _p_i32), ("hist_na", _p_i32),
        ("hist_nb", _p_i32), ("hist_lo_a", _p_f64), ("hist_hi_a", _p_f64),
        ("hist_lo_b", _p_f64), ("hist_hi_b", _p_f64), ("hist_offset", _p_i32),
        ("coat_facet", _p_f64), ("coat_lo", _p_f64), ("coat_hi", _p_f64),
        ("coat_reflectivity", _p_f64), ("coat_reflect_mode", _p_i32),
        ("coat_transmit_mode", _p_i32),
        ("rec_source_mode", _p_i32), ("rec_source_id", _p_i32),
        ("comp_abs_hist", _p_i32), ("comp_ems_hist", _p_i32),
        ("n_mesh_vertices", C.c_int32), ("n_mesh_faces", C.c_int32),
        ("mesh_face_start", _p_i32), ("mesh_face_count", _p_i32), ("mesh_vertices", _p_f64),
        ("mesh_faces", _p_i32), ("mesh_normals", _p_f64),
    ]


class PvtEmitterTables(C.Structure):
    _fields_ = [
        ("n_lights", C.c_int32), ("n_spec", C.c_int32),
        ("wl_type", _p_i32), ("wl_value", _p_f64), ("wl_spec_start", _p_i32),
        ("wl_spec_n", _p_i32), ("pos_type", _p_i32), ("pos_param", _p_f64),
        ("dir_type", _p_i32), ("dir_param", _p_f64), ("light_to_world", _p_f64),
        ("spec_x", _p_f64), ("spec_cdf", _p_f64),
    ]


class PvtTraceParams(C.Structure):
    _fields_ = [
        ("n_rays", C.c_int64), ("seed", C.c_uint64), ("ray_offset", C.c_uint64),
        ("emit_seed", C.c_uint64), ("record_every", C.c_int64), ("maxsteps", C.c_int32),
        ("max_events", C.c_int32), ("emit_method", C.c_int32), ("workgroups_per_cu", C.c_int32),
        ("tally_bundle", C.c_int64), ("tally_stride_i64", C.c_int64), ("tally_stride_f64", C.c_int64),
        ("flags", C.c_int64),
    ]


class PvtRays(C.Structure):
    _fields_ = [("position", _p_f64), ("direction", _p_f64), ("wavelength", _p_f64)]


class PvtTallies(C.Structure):
    _fields_ = [
        ("rec_distinct", _p_i64), ("rec_crossings", _p_i64), ("rec_sums", _p_f64),
        ("rec_bins", _p_i64),
    ]


class PvtEventLog(C.Structure):
    _fields_ = [
        ("counts", _p_i32), ("kind", _p_u8), ("hit", _p_i32), ("container", _p_i32),
        ("adjacent", _p_i32), ("component", _p_i32), ("source", _p_i32),
        ("position", _p_f64), ("direction", _p_f64), ("normal", _p_f64),
        ("wavelength", _p_f64), ("travelled", _p_f64), ("duration", _p_f64),
    ]


class PvtEventRecords(C.Structure):
    _fields_ = [("counts", _p_i32), ("rows", C.POINTER(C.c_uint64))]


_CTYPE_OF = {
    np.dtype(np.int32): C.c_int32, np.dtype(np.float64): C.c_double,
    np.dtype(np.int64): C.c_int64, np.dtype(np.uint8): C.c_uint8,
}


def np_ptr(array):
    """ctypes pointer to a C-contiguous numpy array's data."""
    return array.ctypes.data_as(C.POINTER(_CTYPE_OF[array.dtype]))


def addr_ptr(address, ctype):
    """ctypes pointer from a raw (device) address."""
    return C.cast(C.c_void_p(int(address)), C.POINTER(ctype))


_TABLE_POINTER_FIELDS = [
    name for name, ctype in PvtSceneTables._fields_ if ctype in (_p_i32, _p_f64)
]


def scene_tables_struct(compiled):
    """Build a PvtSceneTables over the arrays of a CompiledScene.

    Returns (struct, keepalive): `keepalive` holds the contiguous arrays the
    pointers refer to and must outlive every use of `struct`."""
    keep = {}
    st = PvtSceneTables()
    st.n_nodes = int(compiled.geom_type.shape[0])
    st.root_id = int(compiled.root_id)
    st.n_components = int(compiled.comp_type.shape[0])
    st.n_abs = int(compiled.abs_x.shape[0])
    st.n_ems = int(compiled.ems_x.shape[0])
    st.n_recorders = int(compiled.rec_node.shape[0])
    st.n_hists = int(compiled.hist_prop_a.shape[0])
    st.total_bins = int(compiled.total_bins)
    st.n_coatings = int(getattr(compiled, "n_coatings", 0))
    st.n_mesh_vertices = int(getattr(compiled, "n_mesh_vertices", 0))
    st.n_mesh_faces = int(getattr(compiled, "n_mesh_faces", 0))
    for name in _TABLE_POINTER_FIELDS:
        want = np.int32 if dict(PvtSceneTables._fields_)[name] is _p_i32 else np.float64
        value = getattr(compiled, name, None)
        if value is None:  # tables from an engine without the coating extension
            value = np.zeros(max(st.n_nodes, 1) * 3, dtype=want)
        arr = np.ascontiguousarray(value, dtype=want)
        if arr.size == 0:
            arr = np.zeros(1, dtype=want)  # never hand out NULL for an empty table
        keep[name] = arr
        setattr(st, name, np_ptr(arr))
    return st, keep


def emitter_tables_struct(emitter):
    """PvtEmitterTables over an `emit.EmitterTables` object -> (struct, keepalive)."""
    keep = {}
    st = PvtEmitterTables()
    st.n_lights = int(emitter.n_lights)
    st.n_spec = int(emitter.spec_x.shape[0])
    fields = dict(PvtEmitterTables._fields_)
    for name, ctype in fields.items():
        if ctype not in (_p_i32, _p_f64):
            continue
        want = np.int32 if ctype is _p_i32 else np.float64
        arr = np.ascontiguousarray(getattr(emitter, name), dtype=want)
        if arr.size == 0:
            arr = np.zeros(1, dtype=want)
        keep[name] = arr
        setattr(st, name, np_ptr(arr))
    return st, keep


def trace_params(n_rays, seed, ray_offset, emit_seed, record_every, maxsteps, max_events,
                 emit_method, workgroups_per_cu=0, tally_bundle=0, tally_stride_i64=0, tally_stride_f64=0, flags=0):
    mask = (1 << 64) - 1
    return PvtTraceParams(
        int(n_rays), int(seed) & mask, int(ray_offset) & mask, int(emit_seed) & mask,
        int(record_every), int(maxsteps), int(max_events), int(emit_method), int(workgroups_per_cu),
        int(tally_bundle), int(tally_stride_i64), int(tally_stride_f64), int(flags),
    )


def num_recorded(n_rays, record_every):
    return (n_rays + record_every - 1) // record_every if record_every > 0 else 0


EVENT_LOG_COLUMNS = (
    # name, dtype, per-row width
    ("kind", np.uint8, 1), ("hit", np.int32, 1), ("container", np.int32, 1),
    ("adjacent", np.int32, 1), ("component", np.int32, 1), ("source", np.int32, 1),
    ("position", np.float64, 3), ("direction", np.float64, 3), ("normal", np.float64, 3),
    ("wavelength", np.float64, 1), ("travelled", np.float64, 1), ("duration", np.float64, 1),
)


RECORD_WORDS = 16   # uint64 words of one event record (include/pvtrace_hip.h PvtEventRecords)
_ID_COLUMNS = ("hit", "container", "adjacent", "component", "source")


def decode_records(rows):
    """(m, 16) int64 event records (host numpy) -> dict of the reference's columns for those m rows
    (layout: include/pvtrace_hip.h PvtEventRecords)."""
    rows = np.ascontiguousarray(rows).reshape(-1, RECORD_WORDS)
    i32 = rows.view(np.int32).reshape(-1, 2 * RECORD_WORDS)
    f64 = rows.view(np.float64).reshape(-1, RECORD_WORDS)
    return {
        "kind": i32[:, 5].astype(np.uint8), "hit": i32[:, 0].copy(), "container": i32[:, 1].copy(),
        "adjacent": i32[:, 2].copy(), "component": i32[:, 3].copy(), "source": i32[:, 4].copy(),
        "position": f64[:, 3:6].copy(), "direction": f64[:, 6:9].copy(), "normal": f64[:, 9:12].copy(),
        "wavelength": f64[:, 12].copy(), "travelled": f64[:, 13].copy(), "duration": f64[:, 14].copy(),
    }


def declare_signatures(lib, names):
    """Set argtypes/restype of the C-ABI entry points present in `lib`."""
    vp = C.c_void_p
    sigs = {
        "pvt_abi_version": ([], C.c_int),
        "pvt_last_error": ([], C.c_char_p),
        "pvt_device_count": ([], C.c_int),
        "pvt_scene_create": ([C.POINTER(PvtSceneTables), C.c_int, C.POINTER(vp)], C.c_int),
        "pvt_scene_set_emitter": ([vp, C.POINTER(PvtEmitterTables)], C.c_int),
        "pvt_scene_destroy": ([vp], None),
        "pvt_trace_device": (
            [vp, C.POINTER(PvtRays), C.POINTER(PvtTraceParams), C.POINTER(PvtTallies),
             C.POINTER(PvtEventLog), vp], C.c_int),
        "pvt_trace_device_records": (
            [vp, C.POINTER(PvtRays), C.POINTER(PvtTraceParams), C.POINTER(PvtTallies),
             C.POINTER(PvtEventRecords), vp], C.c_int),
        "pvt_scene_carry_pending": ([vp, vp], C.c_int),
        "pvt_scene_carry_discard": ([vp, vp], C.c_int),
        "pvt_scene_trim": ([vp], C.c_int),
        "pvt_last_multi_reduce": ([], C.c_int),
        "pvt_unpack_records_device": (
            [C.POINTER(PvtEventRecords), C.c_int64, C.c_int32, C.POINTER(PvtEventLog), C.c_int, vp], C.c_int),
        "pvt_trace_bundle": (
            [C.POINTER(PvtSceneTables), C.POINTER(PvtEmitterTables), C.POINTER(PvtRays),
             C.POINTER(PvtTraceParams), C.POINTER(PvtTallies), C.POINTER(PvtEventLog), C.c_int,
             C.POINTER(C.c_double)], C.c_int),
        "pvt_trace_bundle_multi": (
            [C.POINTER(PvtSceneTables), C.POINTER(PvtEmitterTables), C.POINTER(PvtRays),
             C.POINTER(PvtTraceParams), C.POINTER(PvtTallies), C.POINTER(PvtEventLog), C.POINTER(C.c_int),
             C.c_int, C.POINTER(C.c_double)], C.c_int),
        "pvt_release_cached_memory": ([], None),
        "pvt_shard_range": ([C.c_int64, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)], C.c_int),
        "pvt_emit_device": ([vp, C.POINTER(PvtTraceParams), vp, vp, vp, vp], C.c_int),
        "pvt_selftest_math": ([C.c_int, vp, vp, C.c_int64, C.c_int], C.c_int),
        "pvt_mesh_bvh_check": (
            [C.POINTER(PvtSceneTables), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
             C.POINTER(C.c_int32)], C.c_int),
        "pvt_scene_launch_info": (
            [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)], C.c_int),
        "pvt_scene_counters": ([vp, C.POINTER(C.c_uint64), C.c_int], C.c_int),
        "pvt_scene_clock": ([vp, C.POINTER(C.c_uint64)], C.c_int),
        "pvt_scene_launch_span": ([vp, vp, C.POINTER(C.c_uint64)], C.c_int),
        "pvt_node_grid_plan": (
            [C.POINTER(PvtSceneTables), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double),
             C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.c_int64], C.c_int),
    }
    for name in names:
        argtypes, restype = sigs[name]
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    return sigs


ABI_SYMBOLS = (
    "pvt_abi_version", "pvt_last_error", "pvt_device_count", "pvt_scene_create",
    "pvt_scene_set_emitter", "pvt_scene_destroy", "pvt_trace_device", "pvt_trace_bundle",
    "pvt_emit_device", "pvt_selftest_math", "pvt_scene_launch_info", "pvt_mesh_bvh_check",
    "pvt_trace_bundle_multi", "pvt_shard_range", "pvt_trace_device_records", "pvt_unpack_records_device",
    "pvt_scene_carry_pending", "pvt_last_multi_reduce", "pvt_node_grid_plan", "pvt_scene_carry_discard", "pvt_scene_trim",
    "pvt_scene_counters", "pvt_scene_clock", "pvt_scene_launch_span", "pvt_release_cached_memory",
)

_lib = None
ABI_VERSION = 13  # include/pvtrace_hip.h PVT_ABI_VERSION
FLAG_NO_LOG_PREFILL = 1   # PvtTraceParams.flags
FLAG_CARRY_OUT = 2        # park the photons still alive at the end of the launch for the next launch on the stream


def library_built():
    return os.path.exists(LIB_PATH)


def _preload_torch_hip_runtime():
    """One process must not hold two HIP/HSA runtimes.  PyTorch wheels bundle their own
    libamdhip64.so.7 (+ libhsa-runtime64) under torch/lib; our library's NEEDED entry has
    the same SONAME and a RUNPATH to /opt/rocm.  If ours loaded first, the system runtime
    would be mapped and torch's bundled HSA would later fail with "No HIP GPUs are
    available".  So when torch is installed, map ITS runtime first (without importing
    torch); the dynamic loader then resolves our NEEDED entry to that same copy, whatever
    the import order.  Without torch the RUNPATH copy is used."""
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    try:
        return C.CDLL(path, mode=C.RTLD_GLOBAL)
    except OSError:
        return None


def load_library():
    """dlopen the engine (needs libamdhip64 from /opt/rocm, present on CPU boxes too)."""
    global _lib
    if _lib is None:
        if not library_built():
            raise EngineUnavailableError(
                f"{LIB_PATH} is not built; run `python -c 'import __graft_entry__ as g; g.build()'`"
                " (hipcc --offload-arch=gfx950)."
            )
        _preload_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        declare_signatures(lib, ABI_SYMBOLS)
        if lib.pvt_abi_version() != ABI_VERSION:   # a stale .so would misread the table struct
            raise EngineUnavailableError(
                f"{LIB_PATH} implements ABI v{lib.pvt_abi_version()}, this package needs v{ABI_VERSION}; "
                "rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)."
            )
        _lib = lib
    return _lib


def check(code, what):
    if code == 0:
        return
    lib = load_library()
    detail = lib.pvt_last_error()
    detail = detail.decode() if detail else ""
    if code == -2:
        raise ValueError("Engine supports at most 128 geometry nodes.")
    if code == -1:
        raise ValueError(f"{what}: invalid argument ({detail})")
    raise EngineUnavailableError(f"{what} failed with code {code}: {detail}")


def device_count():
    return int(load_library().pvt_device_count())


def is_available():
    """True iff the HIP library is built AND at least one GPU is visible."""
    if not library_built():
        return False
    try:
        return device_count() > 0
    except OSError:
        return False


def shard_range(n_rays, shard, n_shards, align=1):
    """The library's own shard arithmetic (pvt_shard_range; pure host code, no GPU needed)."""
    lib = load_library()
    a, b = C.c_int64(), C.c_int64()
    check(lib.pvt_shard_range(int(n_rays), int(shard), int(n_shards), int(align), C.byref(a), C.byref(b)),
          "pvt_shard_range")
    return a.value, b.value


def selftest_math(fn, x, device=0):
    """y = f(x) evaluated on the GPU (see pvt_selftest_math in the header)."""
    lib = load_library()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    check(lib.pvt_selftest_math(int(fn), x.ctypes.data, y.ctypes.data, x.size, int(device)),
          "pvt_selftest_math")
    return y


def mesh_bvh_check(compiled, node):
    """Build the BVH of mesh node `node` on the host and verify its invariants (no GPU needed)
    -> (bvh nodes, leaves, depth)."""
    lib = load_library()
    st, keep = scene_tables_struct(compiled)
    out = [C.c_int32(0) for _ in range(3)]
    check(lib.pvt_mesh_bvh_check(C.byref(st), int(node), *(C.byref(v) for v in out)), "pvt_mesh_bvh_check")
    del keep
    return tuple(int(v.value) for v in out)


def node_grid_plan(compiled):
    """The node grid the library files the nodes of a many-node scene under (host only, no GPU needed;
    include/pvtrace_hip.h: pvt_node_grid_plan) -> dict(dims, lo, cell, guard, odd, masks[cells, 2] uint64),
    or None when the scene is served by the plain node loop."""
    import numpy as np

    lib = load_library()
    st, keep = scene_tables_struct(compiled)
    dims = (C.c_int32 * 3)()
    lo, cell = (C.c_double * 3)(), (C.c_double * 3)()
    guard, odd = C.c_double(0.0), C.c_int32(0)
    check(lib.pvt_node_grid_plan(C.byref(st), dims, lo, cell, C.byref(guard), C.byref(odd), None, 0), "pvt_node_grid_plan")
    if dims[0] == 0:
        del keep
        return None
    cells = int(dims[0]) * int(dims[1]) * int(dims[2])
    masks = np.zeros((cells, 2), dtype=np.uint64)
    check(lib.pvt_node_grid_plan(C.byref(st), dims, lo, cell, C.byref(guard), C.byref(odd),
                                 masks.ctypes.data_as(C.POINTER(C.c_uint64)), masks.size), "pvt_node_grid_plan")
    del keep
    return {"dims": tuple(int(v) for v in dims), "lo": np.array(list(lo)), "cell": np.array(list(cell)),
            "guard": float(guard.value), "odd": bool(odd.value), "masks": masks}


class DeviceScene:
    """Scene tables resident in HBM on one GPU (uploaded once, reused per bundle)."""

    def __init__(self, compiled, device=0, emitter=None):
        self.lib = load_library()
        if device_count() <= 0:
            raise EngineUnavailableError("No HIP device visible; the engine has no CPU path.")
        self.compiled = compiled
        self.device = int(device)
        st, keep = scene_tables_struct(compiled)
        handle = C.c_void_p()
        check(self.lib.pvt_scene_create(C.byref(st), self.device, C.byref(handle)),
              "pvt_scene_create")
        self.handle = handle
        self.has_emitter = False
        # HIP stream handle -> weak reference to the BundlePipeline whose job lives on it (parked photons belong to a
        # job, and torch hands the same stream handles out again: see claim_stream)
        self._stream_owners = {}
        self._owners_lock = threading.Lock()
        if emitter is not None:
            self.set_emitter(emitter)

    def set_emitter(self, emitter):
        st, keep = emitter_tables_struct(emitter)
        check(self.lib.pvt_scene_set_emitter(self.handle, C.byref(st)), "pvt_scene_set_emitter")
        self.has_emitter = True

    def close(self):
        if getattr(self, "handle", None):
            self.lib.pvt_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counters(self, reset=False):
        """The scene's always-on step counters since creation / the last reset (pvt_scene_counters): wave-iterations,
        lane-steps, fused exits, waves retired, and what follows from them.  `lane_steps + fused_exits` is the
        reference's loop count (_kernel.pyx:655) summed over the photons traced.  Synchronises the device first."""
        import torch

        torch.cuda.synchronize(self.device)
        out = (C.c_uint64 * 4)()
        check(self.lib.pvt_scene_counters(self.handle, out, 1 if reset else 0), "pvt_scene_counters")
        it, ls, fused, waves = (int(v) for v in out)
        return {"wave_iterations": it, "lane_steps": ls, "fused_exits": fused, "waves": waves,
                "steps": ls + fused, "lane_utilisation": ls / (64.0 * it) if it else 0.0}

    def clock(self):
        """The clocks the scene's launches ran at since its creation / the last `counters(reset=True)`, read on the GPU
        (pvt_scene_clock): shader cycles and 100 MHz ticks summed over the workgroups' lives, and their ratio in MHz.
        Synchronises the device first."""
        import torch

        torch.cuda.synchronize(self.device)
        out = (C.c_uint64 * 2)()
        check(self.lib.pvt_scene_clock(self.handle, out), "pvt_scene_clock")
        cycles, ticks = int(out[0]), int(out[1])
        return {"shader_cycles": cycles, "ticks_100mhz": ticks, "shader_clock_mhz": 100.0 * cycles / ticks if ticks else 0.0}

    def launch_span_ms(self, stream=None):
        """GPU-side duration of the last launch on `stream` (a raw handle; default: torch's current stream), from the
        kernel's own 100 MHz stamps (pvt_scene_launch_span) -- unlike a pair of HIP events around `trace`, it cannot contain
        time the host spent between recording them.  Synchronises the stream."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        out = (C.c_uint64 * 2)()
        check(self.lib.pvt_scene_launch_span(self.handle, C.c_void_p(stream), out), "pvt_scene_launch_span")
        return (int(out[1]) - int(out[0])) * 1e-5

    def launch_info(self):
        g, b, l = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.lib.pvt_scene_launch_info(self.handle, C.byref(g), C.byref(b), C.byref(l)),
              "pvt_scene_launch_info")
        return {"grid": g.value, "block": b.value, "lds_bytes": l.value}

    # -- device buffers (torch tensors) ----------------------------------
    def new_tallies(self, sets=1):
        """Zeroed recorder accumulators on the GPU.  The three integer tables are
        views of ONE int64 buffer (`_ints`: distinct | crossings | bins) and the
        moment sums are `_sums`, so a whole tally set is zeroed by two memsets
        and all-reduced by two collectives.  `sets` > 1: that many consecutive sets
        (the bundles of a stream traced by one launch, PvtTraceParams.tally_bundle)."""
        import torch

        dev = torch.device("cuda", self.device)
        c = self.compiled
        nrec = max(int(c.rec_node.shape[0]), 1)
        nbins = max(int(c.total_bins), 1)
        ints = torch.zeros(sets * (2 * nrec + nbins), dtype=torch.int64, device=dev)
        sums = torch.zeros(sets * nrec * 8, dtype=torch.float64, device=dev)
        return {
            "rec_distinct": ints[:nrec],
            "rec_crossings": ints[nrec:2 * nrec],
            "rec_sums": sums,
            "rec_bins": ints[2 * nrec:],
            "_ints": ints,
            "_sums": sums,
            "sets": sets, "stride_i64": 2 * nrec + nbins, "stride_f64": nrec * 8,
        }

    def new_event_log(self, n_rays, record_every, max_events):
        """Event-RECORD buffers for one bundle (PvtEventRecords): `counts` (recorded rays) and `rows`
        (recorded rays x max_events, 16) int64 -- one 128-byte row per event, uninitialised: only the
        rows k < counts[j] of recorded ray j are ever written (`decode_records`)."""
        import torch

        dev = torch.device("cuda", self.device)
        nrec = num_recorded(n_rays, record_every)
        rows = nrec * max_events
        return {"counts": torch.empty(max(nrec, 1), dtype=torch.int32, device=dev),
                "rows": torch.empty((max(rows, 1), RECORD_WORDS), dtype=torch.int64, device=dev)}

    def new_event_columns(self, n_rays, record_every, max_events):
        """The reference's thirteen column arrays on the device (PvtEventLog), for `unpack_records`."""
        import torch

        dev = torch.device("cuda", self.device)
        nrec = num_recorded(n_rays, record_every)
        rows = nrec * max_events
        log = {"counts": torch.empty(max(nrec, 1), dtype=torch.int32, device=dev)}
        tmap = {np.uint8: torch.uint8, np.int32: torch.int32, np.float64: torch.float64}
        for name, dtype, width in EVENT_LOG_COLUMNS:
            log[name] = torch.empty(max(rows, 1) * width, dtype=tmap[dtype], device=dev)
        return log

    @staticmethod
    def _columns_struct(log):
        el = PvtEventLog()
        el.counts = addr_ptr(log["counts"].data_ptr(), C.c_int32)
        for name, dtype, _ in EVENT_LOG_COLUMNS:
            setattr(el, name, addr_ptr(log[name].data_ptr(), _CTYPE_OF[np.dtype(dtype)]))
        return el

    def unpack_records(self, records, columns, n_recorded, max_events, prefill=True, stream=None):
        """Records -> column arrays on the device (pvt_unpack_records_device)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        rec = PvtEventRecords(addr_ptr(records["counts"].data_ptr(), C.c_int32),
                              addr_ptr(records["rows"].data_ptr(), C.c_uint64))
        el = self._columns_struct(columns)
        check(self.lib.pvt_unpack_records_device(C.byref(rec), int(n_recorded), int(max_events), C.byref(el),
                                                 1 if prefill else 0, C.c_void_p(stream)), "pvt_unpack_records_device")

    def trace(self, rays, n_rays, seed, tallies, log=None, ray_offset=0, emit_seed=0,
              record_every=0, maxsteps=1000, max_events=128, emit_method=0, stream=None,
              workgroups_per_cu=0, tally_bundle=0, log_prefill=True, carry_out=False):
        """Enqueue one bundle (`workgroups_per_cu`: see PvtTraceParams; 0 = the library default).
        `rays` is None (device emission) or a tuple of
        three float64 CUDA tensors (positions (n,3), directions (n,3), wavelengths (n)).
        `carry_out` (tally launches of a stream of bundles): PVT_FLAG_CARRY_OUT -- the photons still alive when the
        launch runs out of new rays are parked for the NEXT launch on this HIP stream instead of being traced to
        completion; finish a job with a launch without it (`n_rays` may be 0), see `carry_pending`."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        if tally_bundle:
            need = -(-int(n_rays) // int(tally_bundle))
            if record_every > 0:
                raise ValueError("tally_bundle needs record_every == 0")
            if tallies.get("sets", 1) < need:
                raise ValueError(f"{need} tally sets needed, the buffers hold {tallies.get('sets', 1)}")
            if need > 1 and not ("stride_i64" in tallies and "stride_f64" in tallies):
                raise ValueError("tally_bundle needs the buffers of new_tallies(sets=...) (stride_i64 / stride_f64)")
        params = trace_params(n_rays, seed, ray_offset, emit_seed, record_every, maxsteps,
                              max_events, emit_method, workgroups_per_cu, tally_bundle,
                              tallies.get("stride_i64", 0) if tally_bundle else 0,
                              tallies.get("stride_f64", 0) if tally_bundle else 0,
                              (0 if log_prefill else FLAG_NO_LOG_PREFILL) | (FLAG_CARRY_OUT if carry_out else 0))
        tl = PvtTallies(
            addr_ptr(tallies["rec_distinct"].data_ptr(), C.c_int64),
            addr_ptr(tallies["rec_crossings"].data_ptr(), C.c_int64),
            addr_ptr(tallies["rec_sums"].data_ptr(), C.c_double),
            addr_ptr(tallies["rec_bins"].data_ptr(), C.c_int64),
        )
        rays_ref = None
        if rays is not None:
            pos, direc, wl = rays
            for t in (pos, direc, wl):
                if t.dtype != torch.float64 or not t.is_contiguous() or t.device.index != self.device:
                    raise ValueError("rays must be contiguous float64 tensors on the scene's GPU")
            rays_ref = C.byref(PvtRays(addr_ptr(pos.data_ptr(), C.c_double),
                                       addr_ptr(direc.data_ptr(), C.c_double),
                                       addr_ptr(wl.data_ptr(), C.c_double)))
        elif not self.has_emitter and n_rays > 0:
            raise ValueError("device emission requested but the scene has no emitter tables")
        if record_every > 0 and log is None:
            raise ValueError("record_every > 0 needs event-log buffers")
        if record_every > 0 and "rows" not in log:
            # column arrays (PvtEventLog) on the device: the library stages the records and unpacks them
            el = self._columns_struct(log)
            check(self.lib.pvt_trace_device(self.handle, rays_ref, C.byref(params), C.byref(tl),
                                            C.byref(el), C.c_void_p(stream)), "pvt_trace_device")
            return
        rec_ref = None
        if record_every > 0:
            rec_ref = C.byref(PvtEventRecords(addr_ptr(log["counts"].data_ptr(), C.c_int32),
                                              addr_ptr(log["rows"].data_ptr(), C.c_uint64)))
        check(self.lib.pvt_trace_device_records(self.handle, rays_ref, C.byref(params), C.byref(tl),
                                                rec_ref, C.c_void_p(stream)), "pvt_trace_device_records")

    def carry_pending(self, stream=None):
        """True when photons parked by the last launch on `stream` (`carry_out=True`) wait to be resumed."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        return bool(self.lib.pvt_scene_carry_pending(self.handle, C.c_void_p(stream)))

    def trim(self):
        """Before the scene is put aside for reuse: free staging buffers, forget parked photons (pvt_scene_trim).
        Refused while a live pipeline still owns one of the scene's streams: its parked photons would vanish."""
        live = self.stream_owners()
        if live:
            raise RuntimeError(f"{len(live)} BundlePipeline(s) still hold streams of this scene; close them first "
                               "(trimming would drop the photons their jobs have parked)")
        check(self.lib.pvt_scene_trim(self.handle), "pvt_scene_trim")

    # -- who a stream's parked photons belong to ------------------------------------------------------------------
    def stream_owners(self):
        """The live pipelines that own a stream of this scene."""
        with self._owners_lock:
            owners = {}
            for handle, ref in list(self._stream_owners.items()):
                owner = ref()
                if owner is None:
                    del self._stream_owners[handle]   # its job died with it
                else:
                    owners[id(owner)] = owner
            return list(owners.values())

    def claim_stream(self, owner, stream_handle):
        """`owner` (a BundlePipeline) starts a job on `stream_handle`.  torch hands out streams from a pool of ~32 per
        device, so the handle may be one an EARLIER pipeline used: if that pipeline is gone (closed, or dropped) whatever
        it left parked belongs to a dead job and is discarded; if it is still alive the two jobs would resume each
        other's photons, and the claim is refused."""
        import weakref

        with self._owners_lock:
            ref = self._stream_owners.get(stream_handle)
            other = ref() if ref is not None else None
            if other is not None and other is not owner:
                raise RuntimeError(
                    "this HIP stream already carries the job of another live BundlePipeline on the same scene (torch reuses "
                    "stream handles); close() that pipeline first, or give each pipeline a scene of its own")
            self._stream_owners[stream_handle] = weakref.ref(owner)
        if self.carry_pending(stream_handle):
            self.carry_discard(stream_handle)

    def release_stream(self, owner, stream_handle):
        with self._owners_lock:
            ref = self._stream_owners.get(stream_handle)
            if ref is not None and ref() in (owner, None):
                del self._stream_owners[stream_handle]

    def carry_discard(self, stream=None):
        """Forget the photons parked on `stream` (an abandoned job)."""
        import torch

        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib.pvt_scene_carry_discard(self.handle, C.c_void_p(stream)), "pvt_scene_carry_discard")

    def emit(self, n_rays, emit_seed, ray_offset=0, stream=None):
        """Device-side emission only -> (positions, directions, wavelengths) CUDA tensors."""
        import torch

        if not self.has_emitter:
            raise ValueError("scene has no emitter tables")
        dev = torch.device("cuda", self.device)
        pos = torch.empty((n_rays, 3), dtype=torch.float64, device=dev)
        direc = torch.empty((n_rays, 3), dtype=torch.float64, device=dev)
        wl = torch.empty(n_rays, dtype=torch.float64, device=dev)
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        params = trace_params(n_rays, 0, ray_offset, emit_seed, 0, 0, 0, 0)
        check(self.lib.pvt_emit_device(self.handle, C.byref(params), C.c_void_p(pos.data_ptr()),
                                       C.c_void_p(direc.data_ptr()), C.c_void_p(wl.data_ptr()),
                                       C.c_void_p(stream)), "pvt_emit_device")
        return pos, direc, wl
