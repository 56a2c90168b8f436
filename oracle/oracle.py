"""Python front of the CPU referee (oracle/pvt_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never by the product package.

`trace_bundle` has the signature and result dict of the reference's
``_kernel.trace_bundle`` (pvtrace/engine/_kernel.pyx:903-1115) plus the
`ray_offset` / `math_mode` knobs, so tests can diff it key by key against both
the reference kernel (math_mode=0, here only) and the HIP engine (math_mode=1).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from pvtrace_amd.engine import native as N

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libpvt_oracle.so")

MATH_LIBM = 0       # bit-identical to the reference kernel
MATH_PORTABLE = 1   # bit-identical to the HIP kernel

_lib = None


def build(force=False):
    src = os.path.join(HERE, "pvt_oracle.c")
    deps = [src, os.path.join(HERE, "..", "include", "pvtrace_hip.h"),
            os.path.join(HERE, "..", "pvtrace_amd", "csrc", "pvt_math.h")]
    stale = force or not os.path.exists(LIB) or any(
        os.path.exists(d) and os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps
    )
    if stale:
        subprocess.check_call(["make", "-C", HERE, "-B", "libpvt_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return LIB


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.pvt_oracle_trace.argtypes = [
            C.POINTER(N.PvtSceneTables), C.POINTER(N.PvtRays), C.POINTER(N.PvtTraceParams),
            C.POINTER(N.PvtTallies), C.POINTER(N.PvtEventLog), C.c_int, C.c_int]
        L.pvt_oracle_trace.restype = C.c_int
        L.pvt_oracle_emit.argtypes = [
            C.POINTER(N.PvtEmitterTables), C.POINTER(N.PvtTraceParams), C.c_void_p, C.c_void_p,
            C.c_void_p, C.c_int]
        L.pvt_oracle_emit.restype = C.c_int
        L.pvt_oracle_math.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_long]
        L.pvt_oracle_math.restype = None
        L.pvt_oracle_fresnel_reflectivity.argtypes = [C.c_double, C.c_double, C.c_double, C.c_int]
        L.pvt_oracle_fresnel_reflectivity.restype = C.c_double
        L.pvt_oracle_interp.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        L.pvt_oracle_interp.restype = C.c_double
        L.pvt_oracle_step_lookup.argtypes = [C.c_double, C.c_void_p, C.c_void_p, C.c_int]
        L.pvt_oracle_step_lookup.restype = C.c_double
        L.pvt_oracle_intersect.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pvt_oracle_intersect.restype = C.c_int
        L.pvt_oracle_normal.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pvt_oracle_mesh_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int]
        L.pvt_oracle_mesh_hits.restype = C.c_int
        L.pvt_oracle_uniforms.argtypes = [C.c_uint64, C.c_void_p, C.c_int]
        L.pvt_oracle_surface.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pvt_oracle_last_steps.restype = C.c_long
        L.pvt_oracle_phase.argtypes = [C.c_int, C.c_double, C.c_uint64, C.c_int, C.c_void_p]
        L.pvt_oracle_fresnel_refract.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.pvt_oracle_specular_reflect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def alloc_outputs(compiled, n_rays, record_every, max_events):
    """Zeroed host output arrays in the reference's shapes/dtypes."""
    nrec = int(compiled.rec_node.shape[0])
    n_recorded = N.num_recorded(n_rays, record_every)
    rows = n_recorded * max_events
    out = {
        "counts": np.zeros(max(n_recorded, 1), dtype=np.int32),
        "rec_distinct": np.zeros(max(nrec, 1), dtype=np.int64),
        "rec_crossings": np.zeros(max(nrec, 1), dtype=np.int64),
        "rec_sums": np.zeros(max(nrec, 1) * 8, dtype=np.float64),
        "rec_bins": np.zeros(max(int(compiled.total_bins), 1), dtype=np.int64),
    }
    for name, dtype, width in N.EVENT_LOG_COLUMNS:
        out[name] = np.zeros(max(rows, 1) * width, dtype=dtype)
    return out, n_recorded, rows


def finish_outputs(compiled, out, n_recorded, rows):
    """Trim/reshape to exactly what _kernel.trace_bundle returns (:1097-1115)."""
    nrec = int(compiled.rec_node.shape[0])
    data = {
        "counts": out["counts"][:n_recorded],
        "rec_distinct": out["rec_distinct"][:nrec],
        "rec_crossings": out["rec_crossings"][:nrec],
        "rec_sums": out["rec_sums"][: nrec * 8].reshape(nrec, 4, 2),
        "rec_bins": out["rec_bins"][: int(compiled.total_bins)],
    }
    for name, _, width in N.EVENT_LOG_COLUMNS:
        col = out[name][: rows * width]
        data[name] = col.reshape(rows, 3) if width == 3 else col
    return data


def structs_for_outputs(out):
    tl = N.PvtTallies(N.np_ptr(out["rec_distinct"]), N.np_ptr(out["rec_crossings"]),
                      N.np_ptr(out["rec_sums"]), N.np_ptr(out["rec_bins"]))
    el = N.PvtEventLog()
    el.counts = N.np_ptr(out["counts"])
    for name, _, _ in N.EVENT_LOG_COLUMNS:
        setattr(el, name, N.np_ptr(out[name]))
    return tl, el


def trace_bundle(compiled, positions, directions, wavelengths, seed, maxsteps, max_events,
                 emit_method, num_threads, record_every, ray_offset=0, math_mode=MATH_LIBM):
    L = lib()
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    dirs = np.ascontiguousarray(directions, dtype=np.float64)
    wl = np.ascontiguousarray(wavelengths, dtype=np.float64)
    n = pos.shape[0]
    st, keep = N.scene_tables_struct(compiled)
    out, n_recorded, rows = alloc_outputs(compiled, n, record_every, max_events)
    tl, el = structs_for_outputs(out)
    rays = N.PvtRays(N.np_ptr(pos), N.np_ptr(dirs), N.np_ptr(wl))
    params = N.trace_params(n, seed, ray_offset, 0, record_every, maxsteps, max_events, emit_method)
    code = L.pvt_oracle_trace(C.byref(st), C.byref(rays), C.byref(params), C.byref(tl),
                              C.byref(el), int(num_threads), int(math_mode))
    if code == -2:
        raise ValueError("Engine supports at most 128 geometry nodes.")
    if code != 0:
        raise RuntimeError(f"pvt_oracle_trace failed: {code}")
    return finish_outputs(compiled, out, n_recorded, rows)


def last_steps():
    """Trips of the photon loop (`count`, _kernel.pyx:655) summed over the rays of this thread's last trace_bundle."""
    return int(lib().pvt_oracle_last_steps())


def emit(emitter, n_rays, emit_seed, ray_offset=0, math_mode=MATH_PORTABLE):
    """Per-ray-stream emission, mirror of the device emitter -> (pos, dir, wl)."""
    L = lib()
    st, keep = N.emitter_tables_struct(emitter)
    pos = np.zeros((n_rays, 3)); dirs = np.zeros((n_rays, 3)); wl = np.zeros(n_rays)
    params = N.trace_params(n_rays, 0, ray_offset, emit_seed, 0, 0, 0, 0)
    code = L.pvt_oracle_emit(C.byref(st), C.byref(params), pos.ctypes.data, dirs.ctypes.data,
                             wl.ctypes.data, int(math_mode))
    if code != 0:
        raise RuntimeError(f"pvt_oracle_emit failed: {code}")
    return pos, dirs, wl


MATH_FN = {"log": 0, "sin": 1, "cos": 2, "asin": 3, "acos": 4, "sqrt": 5, "rcp": 6,
           "sincos_product": 7, "uniform2": 8, "ratio": 9,
           "div_c": 10, "div_n": 11, "div_hist": 12, "div_any": 13,
           "sin2pi": 14, "cos2pi": 15, "sqrt1m2": 16, "rcp_normal": 17, "div_normal": 18, "sqrt_normal": 19}


def math(fn, x, math_mode=MATH_PORTABLE):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib().pvt_oracle_math(MATH_FN[fn], int(math_mode), x.ctypes.data, y.ctypes.data, x.size)
    return y


def uniforms(seed, n):
    out = np.zeros(n)
    lib().pvt_oracle_uniforms(C.c_uint64(seed & ((1 << 64) - 1)), out.ctypes.data, n)
    return out


def fresnel_reflectivity(angle, n1, n2, math_mode=MATH_LIBM):
    return lib().pvt_oracle_fresnel_reflectivity(angle, n1, n2, math_mode)


def fresnel_refract(d, nflipped, n1, n2):
    d = np.ascontiguousarray(d, dtype=np.float64); nf = np.ascontiguousarray(nflipped, dtype=np.float64)
    out = np.zeros(3)
    lib().pvt_oracle_fresnel_refract(d.ctypes.data, nf.ctypes.data, n1, n2, out.ctypes.data)
    return out


def specular_reflect(d, normal):
    d = np.ascontiguousarray(d, dtype=np.float64); nm = np.ascontiguousarray(normal, dtype=np.float64)
    out = np.zeros(3)
    lib().pvt_oracle_specular_reflect(d.ctypes.data, nm.ctypes.data, out.ctypes.data)
    return out


def surface(geom_type, params, point, direction, n1, n2, math_mode=MATH_LIBM):
    """The Fresnel surface branch at `point` of an untransformed shape -> (outward normal, reflectivity, reflected
    direction, refracted direction -- NaN beyond the critical angle)."""
    prm = np.zeros(4); prm[:len(params)] = params
    p = np.ascontiguousarray(point, dtype=np.float64); d = np.ascontiguousarray(direction, dtype=np.float64)
    nrm, refl, trans = np.zeros(3), np.zeros(3), np.zeros(3)
    r = C.c_double(0.0)
    lib().pvt_oracle_surface(int(geom_type), prm.ctypes.data, p.ctypes.data, d.ctypes.data, float(n1), float(n2),
                             int(math_mode), nrm.ctypes.data, C.addressof(r), refl.ctypes.data, trans.ctypes.data)
    return nrm, r.value, refl, trans


def phase(phase_type, param, seed, math_mode=MATH_LIBM):
    """Direction a component's phase function samples from the stream of `seed` (its draws: uniforms(seed, 2))."""
    out = np.zeros(3)
    lib().pvt_oracle_phase(int(phase_type), float(param), int(seed), int(math_mode), out.ctypes.data)
    return out


def interp(x, xs, ys):
    xs = np.ascontiguousarray(xs, dtype=np.float64); ys = np.ascontiguousarray(ys, dtype=np.float64)
    return lib().pvt_oracle_interp(float(x), xs.ctypes.data, ys.ctypes.data, xs.size)


def step_lookup(x, xs, ys):
    xs = np.ascontiguousarray(xs, dtype=np.float64); ys = np.ascontiguousarray(ys, dtype=np.float64)
    return lib().pvt_oracle_step_lookup(float(x), xs.ctypes.data, ys.ctypes.data, xs.size)


def intersect(geom_type, params, origin, direction):
    prm = np.zeros(4); prm[: len(params)] = params
    o = np.ascontiguousarray(origin, dtype=np.float64); d = np.ascontiguousarray(direction, dtype=np.float64)
    ts = np.zeros(8)
    n = lib().pvt_oracle_intersect(geom_type, prm.ctypes.data, o.ctypes.data, d.ctypes.data, ts.ctypes.data)
    return ts[:n].copy()


def normal(geom_type, params, point):
    prm = np.zeros(4); prm[: len(params)] = params
    p = np.ascontiguousarray(point, dtype=np.float64)
    out = np.zeros(3)
    lib().pvt_oracle_normal(geom_type, prm.ctypes.data, p.ctypes.data, out.ctypes.data)
    return out


def mesh_hits(vertices, faces, origin, direction):
    """(t, face) of every forward crossing of a ray with a triangle list, in face order."""
    v = np.ascontiguousarray(vertices, dtype=np.float64); f = np.ascontiguousarray(faces, dtype=np.int32)
    o = np.ascontiguousarray(origin, dtype=np.float64); d = np.ascontiguousarray(direction, dtype=np.float64)
    cap = max(len(f), 1)
    ts = np.zeros(cap); tris = np.zeros(cap, dtype=np.int32)
    n = lib().pvt_oracle_mesh_hits(v.ctypes.data, f.ctypes.data, len(f), o.ctypes.data, d.ctypes.data,
                                   ts.ctypes.data, tris.ctypes.data, cap)
    return ts[:n].copy(), tris[:n].copy()


# ---------------------------------------------------------------------------
# Reference kernel (this container only): oracle/_ref via oracle/build_ref.py

def reference_kernel():
    """The reference's own compiled kernel module, or None when unavailable."""
    from oracle import build_ref

    if not build_ref.ref_available():
        return None
    return build_ref.load()


def reference_trace_bundle(compiled, positions, directions, wavelengths, seed, maxsteps,
                           max_events, emit_method, num_threads, record_every):
    """Drive the REFERENCE kernel with our tables (it duck-types `compiled`,
    _kernel.pyx:929-1017).  Coating tables are ignored by it."""
    kernel = reference_kernel()
    if kernel is None:
        raise RuntimeError("reference kernel unavailable")
    return kernel.trace_bundle(
        compiled, np.ascontiguousarray(positions, dtype=np.float64),
        np.ascontiguousarray(directions, dtype=np.float64),
        np.ascontiguousarray(wavelengths, dtype=np.float64),
        int(seed), int(maxsteps), int(max_events), int(emit_method), int(num_threads),
        int(record_every))
