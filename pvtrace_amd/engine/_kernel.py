"""`trace_bundle` — same name, arguments and result dict as the reference's
native entry point ``pvtrace.engine._kernel.trace_bundle``
(pvtrace/engine/_kernel.pyx:903-1115), executed by the HIP engine.

This path goes through the *host-buffer* C entry `pvt_trace_bundle` (numpy in,
numpy out, no torch anywhere): it is exactly what a pvtrace maintainer would
bind to swap engines.  `engine.simulate` uses the
device-resident entry instead so tallies can stay on the GPU for the RCCL
all-reduce.
"""
import ctypes as C

import numpy as np

from pvtrace_amd.engine import native as N


def _host_outputs(compiled, n_rays, record_every, max_events):
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


def trace_bundle(compiled, positions, directions, wavelengths, seed, maxsteps, max_events,
                 emit_method, num_threads, record_every, *, device=0, devices=None, ray_offset=0,
                 emitter=None, emit_seed=0, timing=None, flags=0):
    """Trace a bundle on the GPU; returns the reference's result dict.

    `num_threads` is accepted for signature compatibility and ignored.  With
    `positions is None` and an `emit.EmitterTables` in `emitter`, rays are
    sampled on the device (`wavelengths` must then be the ray count).  `devices` (a list of GPU
    ids, repeats allowed) splits the bundle over several GPUs inside this one call
    (`pvt_trace_bundle_multi`); the result is independent of the list."""
    lib = N.load_library()
    st, keep = N.scene_tables_struct(compiled)
    if positions is None:
        n = int(wavelengths)
        rays_ref = None
    else:
        pos = np.ascontiguousarray(positions, dtype=np.float64)
        dirs = np.ascontiguousarray(directions, dtype=np.float64)
        wl = np.ascontiguousarray(wavelengths, dtype=np.float64)
        n = pos.shape[0]
        if pos.shape != (n, 3) or dirs.shape != (n, 3) or wl.shape != (n,):
            raise ValueError("positions/directions must be (n,3) and wavelengths (n,)")
        rays_ref = C.byref(N.PvtRays(N.np_ptr(pos), N.np_ptr(dirs), N.np_ptr(wl)))
    em_ref = None
    if emitter is not None:
        est, ekeep = N.emitter_tables_struct(emitter)
        em_ref = C.byref(est)
    out, n_recorded, rows = _host_outputs(compiled, n, record_every, max_events)
    tl = N.PvtTallies(N.np_ptr(out["rec_distinct"]), N.np_ptr(out["rec_crossings"]),
                      N.np_ptr(out["rec_sums"]), N.np_ptr(out["rec_bins"]))
    el = N.PvtEventLog()
    el.counts = N.np_ptr(out["counts"])
    for name, _, _ in N.EVENT_LOG_COLUMNS:
        setattr(el, name, N.np_ptr(out[name]))
    params = N.trace_params(n, seed, ray_offset, emit_seed, record_every, maxsteps, max_events,
                            emit_method, flags=flags)
    ms = C.c_double(0.0)
    if devices is None:
        code = lib.pvt_trace_bundle(C.byref(st), em_ref, rays_ref, C.byref(params), C.byref(tl),
                                    C.byref(el) if record_every > 0 else None, int(device),
                                    C.byref(ms))
    else:
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        code = lib.pvt_trace_bundle_multi(C.byref(st), em_ref, rays_ref, C.byref(params), C.byref(tl),
                                          C.byref(el) if record_every > 0 else None, ids, len(devices),
                                          C.byref(ms))
    N.check(code, "pvt_trace_bundle")
    if timing is not None:
        timing["kernel_ms"] = ms.value
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


def trace_bundle_sets(compiled, positions, directions, wavelengths, seed, maxsteps, emit_method, bundle,
                      *, device=0, ray_offset=0):
    """Tally-mode trace of consecutive bundles of `bundle` rays in ONE launch (host buffers; the C entry
    `pvt_trace_bundle` with `PvtTraceParams.tally_bundle`) -> one dict of rec_* arrays per bundle."""
    lib = N.load_library()
    st, keep = N.scene_tables_struct(compiled)
    pos = np.ascontiguousarray(positions, dtype=np.float64)
    dirs = np.ascontiguousarray(directions, dtype=np.float64)
    wl = np.ascontiguousarray(wavelengths, dtype=np.float64)
    n = pos.shape[0]
    sets = -(-n // int(bundle))
    nrec, nbins = int(compiled.rec_node.shape[0]), int(compiled.total_bins)
    pad = max(nrec, 1)
    stride_i, stride_d = 2 * pad + max(nbins, 1), pad * 8
    ints = np.zeros(sets * stride_i, dtype=np.int64)
    sums = np.zeros(sets * stride_d, dtype=np.float64)
    tl = N.PvtTallies(N.np_ptr(ints), N.np_ptr(ints[pad:]), N.np_ptr(sums), N.np_ptr(ints[2 * pad:]))
    params = N.trace_params(n, seed, ray_offset, 0, 0, maxsteps, 2, emit_method, 0, bundle, stride_i, stride_d)
    rays = N.PvtRays(N.np_ptr(pos), N.np_ptr(dirs), N.np_ptr(wl))
    N.check(lib.pvt_trace_bundle(C.byref(st), None, C.byref(rays), C.byref(params), C.byref(tl), None,
                                 int(device), None), "pvt_trace_bundle")
    ints, sums = ints.reshape(sets, stride_i), sums.reshape(sets, stride_d)
    return [{"rec_distinct": ints[j, :nrec], "rec_crossings": ints[j, pad:pad + nrec],
             "rec_bins": ints[j, 2 * pad:2 * pad + nbins], "rec_sums": sums[j, :nrec * 8].reshape(nrec, 4, 2)}
            for j in range(sets)]
