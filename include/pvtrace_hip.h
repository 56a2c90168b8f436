/* pvtrace_hip.h — C ABI of the MI355X photon-tracing engine (libpvtrace_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of pvtrace: the native call
 *     _kernel.trace_bundle(compiled, positions, directions, wavelengths, seed,
 *                          maxsteps, max_events, emit_method, num_threads,
 *                          record_every) -> dict
 * (reference pvtrace/engine/_kernel.pyx:903-1115), which engine.simulate()
 * (pvtrace/engine/api.py:197-246) wraps.  Plain pointers and sizes only; no
 * torch / numpy / HIP types appear in any signature.  INTEGRATION.md shows the
 * ctypes stub a pvtrace maintainer would add.
 *
 * Table layouts, dtypes and tag values are exactly the reference's
 * CompiledScene (pvtrace/engine/compiler.py:25-50, :75-204) so the same arrays
 * can be handed to either engine.  All matrices are row-major 4x4 doubles.
 */
#ifndef PVTRACE_HIP_H
#define PVTRACE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVT_ABI_VERSION 13

/* limits (reference _kernel.pyx:65-68) */
#define PVT_MAX_NODES 128
#define PVT_MAX_RECORDERS 256
#define PVT_MAX_HITS 512

/* event codes == pvtrace.light.event.Event (reference light/event.py:7-16) */
enum {
    PVT_EV_GENERATE = 0, PVT_EV_REFLECT = 1, PVT_EV_TRANSMIT = 2, PVT_EV_ABSORB = 3,
    PVT_EV_NONRADIATIVE = 4, PVT_EV_SCATTER = 5, PVT_EV_EMIT = 6, PVT_EV_EXIT = 7,
    PVT_EV_REACT = 8, PVT_EV_KILL = 9
};
/* geometry / surface / component / phase / emit-method tags (compiler.py:25-48) */
enum { PVT_GEOM_BOX = 0, PVT_GEOM_SPHERE = 1, PVT_GEOM_CYLINDER = 2, PVT_GEOM_MESH = 3 };
enum { PVT_SURF_FRESNEL = 0, PVT_SURF_NULL = 1 };
enum { PVT_COMP_ABSORBER = 0, PVT_COMP_SCATTERER = 1, PVT_COMP_LUMINOPHORE = 2, PVT_COMP_REACTOR = 3 };
enum { PVT_PHASE_ISOTROPIC = 0, PVT_PHASE_HG = 1, PVT_PHASE_CONE = 2,
       /* EXTENSION (the reference engine rejects it, compiler.py:300-310; its Python path and its scene-spec parser have it,
        * material/utils.py:176-186, cli/parse.py:166-167): cosine-weighted about +z, theta = asin(sqrt(p1)), phi = 2 pi p2 */
       PVT_PHASE_LAMBERTIAN = 3 };
enum { PVT_EMIT_KT = 0, PVT_EMIT_REDSHIFT = 1, PVT_EMIT_FULL = 2 };
/* recorder selectors (engine/recorder.py:45-53) */
enum {
    PVT_REC_ENTERING = 0, PVT_REC_ESCAPING = 1, PVT_REC_REFLECTED = 2, PVT_REC_LOST = 3,
    PVT_REC_REACTED = 4, PVT_REC_KILLED = 5, PVT_REC_EXIT = 6
};
/* error codes (negative returns) */
enum {
    PVT_OK = 0,
    PVT_ERR_INVALID = -1,     /* bad argument; maps to ValueError            */
    PVT_ERR_TOO_MANY_NODES = -2, /* > PVT_MAX_NODES; reference raises ValueError (_kernel.pyx:929-930) */
    PVT_ERR_HIP = -3,         /* HIP runtime failure; see pvt_last_error()   */
    PVT_ERR_NO_DEVICE = -4
};

/* ---- scene tables: HOST pointers, read once by pvt_scene_create ---------
 * Field-for-field the attributes _kernel.trace_bundle reads off `compiled`
 * (_kernel.pyx:933-1017), plus the coating extension (coat_*), which the
 * reference engine has no counterpart for (its compiler rejects non-Fresnel
 * delegates, compiler.py:237-247).  n_coatings may be 0 with NULL coat_* rows. */
typedef struct PvtSceneTables {
    int32_t n_nodes, root_id, n_components, n_abs, n_ems;
    int32_t n_recorders, n_hists, total_bins, n_coatings, reserved0;
    /* per node */
    const int32_t* geom_type;
    const double* geom_params;      /* (n_nodes,4) box: sx,sy,sz  sphere: r  cyl: length, radius */
    const double* local_to_world;   /* (n_nodes,4,4) */
    const double* world_to_local;   /* (n_nodes,4,4) */
    const double* refractive_index;
    const int32_t* surface_type;
    const int32_t* comp_start;
    const int32_t* comp_count;
    const int32_t* coat_start;
    const int32_t* coat_count;
    /* per component */
    const int32_t* comp_type;
    const double* comp_qy;
    const double* comp_tau_rad;
    const double* comp_tau_nr;
    const int32_t* comp_phase_type;
    const double* comp_phase_param;
    const int32_t* comp_abs_start;
    const int32_t* comp_abs_n;
    const int32_t* comp_ems_start;
    const int32_t* comp_ems_n;
    /* pooled spectra */
    const double* abs_x;            /* (n_abs) */
    const double* abs_y;
    const double* ems_x;            /* (n_ems) */
    const double* ems_cdf;
    /* recorders */
    const int32_t* rec_node;
    const int32_t* rec_event;
    const int32_t* rec_has_facet;
    const double* rec_facet;        /* (max(n_recorders,1),3) */
    const double* rec_atol;
    const int32_t* rec_hist_start;
    const int32_t* rec_hist_n;
    /* histograms */
    const int32_t* hist_prop_a;
    const int32_t* hist_prop_b;     /* -1: 1-D */
    const int32_t* hist_na;
    const int32_t* hist_nb;
    const double* hist_lo_a;
    const double* hist_hi_a;
    const double* hist_lo_b;
    const double* hist_hi_b;
    const int32_t* hist_offset;
    /* coatings (extension) */
    const double* coat_facet;       /* (n_coatings,3) local-frame outward normal */
    const double* coat_lo;          /* (n_coatings,3) open local AABB, -inf/inf = unbounded */
    const double* coat_hi;
    const double* coat_reflectivity;/* <0: keep Fresnel */
    const int32_t* coat_reflect_mode;   /* 0 specular, 1 lambertian */
    const int32_t* coat_transmit_mode;  /* 0 Fresnel refraction, 1 index matched */
    /* recorder source filter (extension; NULL = no filter): 0 any, 1 photons emitted by a
     * light (source id < 0), 2 by any component, 3 by component rec_source_id */
    const int32_t* rec_source_mode;
    const int32_t* rec_source_id;
    /* histogram-sampled spectra (extension; NULL = all interpolated).  The reference engine
     * rejects hist=True distributions (compiler.py:274-279, :313-317); semantics follow the
     * Python Distribution's hist branch (material/distribution.py:82-84, :127-129, :171-176):
     * value(x) = y[#{x_i < x}], sample(p) = x[#{cdf_i < p}], no interpolation. */
    const int32_t* comp_abs_hist;
    const int32_t* comp_ems_hist;
    /* triangle meshes (extension; geom_type PVT_GEOM_MESH).  The reference engine rejects
     * Mesh nodes (compiler.py:220-223); its Python tracer traces them through trimesh
     * (geometry/mesh.py:44-61).  Semantics here: every forward crossing (t > EPS) of the
     * node-local ray with a face is a hit, exactly as for the analytic shapes, so the
     * container rule (_kernel.pyx:684-714) applies unchanged; the normal of a hit is the
     * face normal (geometry/mesh.py:63-86).  The ray/triangle test is the watertight
     * shear-and-edge-function test with a half-plane tie rule for exact zeros, so a ray
     * through a shared edge or vertex crosses the surface exactly once.  Vertices are in
     * the node's local frame (already centred on the centre of mass, mesh.py:17). */
    int32_t n_mesh_vertices, n_mesh_faces;
    const int32_t* mesh_face_start; /* (n_nodes) first face of the node's mesh, 0 if none */
    const int32_t* mesh_face_count; /* (n_nodes) 0 for non-mesh nodes */
    const double* mesh_vertices;    /* (n_mesh_vertices,3) pooled */
    const int32_t* mesh_faces;      /* (n_mesh_faces,3) indices into the pooled vertices */
    const double* mesh_normals;     /* (n_mesh_faces,3) outward unit face normals */
} PvtSceneTables;

/* ---- optional device-side emission (replaces the Python/numpy emitter,
 * reference pvtrace/engine/emit.py:22-134).  Ray i is emitted by light
 * i % n_lights (scene.emit round-robin, scene/scene.py:141-151) from its own
 * RNG stream keyed by (emit_seed, global ray index).                        */
enum { PVT_WL_CONSTANT = 0, PVT_WL_SPECTRUM = 1,
       PVT_WL_SPECTRUM_HIST = 2 /* histogram-sampled Distribution (material/distribution.py:171-176): x[#{cdf_i < u}], no interpolation */ };
enum { PVT_POS_POINT = 0, PVT_POS_RECT = 1, PVT_POS_CIRCLE = 2, PVT_POS_CUBE = 3 };
enum { PVT_DIR_Z = 0, PVT_DIR_CONE = 1, PVT_DIR_ISOTROPIC = 2, PVT_DIR_LAMBERTIAN = 3, PVT_DIR_HG = 4 };
typedef struct PvtEmitterTables {
    int32_t n_lights, n_spec;
    const int32_t* wl_type;
    const double* wl_value;         /* constant wavelength (nm) */
    const int32_t* wl_spec_start;   /* into spec_x / spec_cdf */
    const int32_t* wl_spec_n;
    const int32_t* pos_type;
    const double* pos_param;        /* (n_lights,3) half-extents / radius */
    const int32_t* dir_type;
    const double* dir_param;        /* theta_max or g */
    const double* light_to_world;   /* (n_lights,4,4) */
    const double* spec_x;           /* pooled inverse-CDF tables */
    const double* spec_cdf;
} PvtEmitterTables;

typedef struct PvtTraceParams {
    int64_t n_rays;         /* rays in this bundle                                   */
    uint64_t seed;          /* ray i traces with RNG stream seed + ray_offset + i    */
    uint64_t ray_offset;    /* global index of ray 0 (bundle streaming / GPU shard)  */
    uint64_t emit_seed;     /* device emission only                                  */
    int64_t record_every;   /* every k-th ray keeps a full history; 0 = none         */
    int32_t maxsteps;
    int32_t max_events;
    int32_t emit_method;    /* PVT_EMIT_*                                            */
    int32_t workgroups_per_cu; /* persistent workgroups launched per CU; 0 = default (4: one launch
                                * fills the chip).  Launches that overlap on several streams run best
                                * with fewer (2 with three in flight): each then holds fewer CU slots
                                * while it drains, and a workgroup amortises its drain over more photons */
    /* A stream of equal bundles in ONE launch (the reference's simulate_stream, api.py:249-264, hands
     * its kernel one bundle per call; a 50 000-photon bundle is far too little to occupy an MI355X).
     * With tally_bundle = m > 0 (tally mode only: record_every == 0) the n_rays rays are the
     * concatenation of bundles of m rays (the last may be shorter): ray i still uses the stream
     * seed + ray_offset + i — exactly the seeds of the streamed bundles — but the rays of bundle
     * j = i / m are tallied into set j of the tally arrays, set j starting tally_stride_i64 int64
     * elements (rec_distinct, rec_crossings, rec_bins alike) and tally_stride_f64 doubles (rec_sums)
     * after set j-1.  0 = the launch is one bundle. */
    int64_t tally_bundle;
    int64_t tally_stride_i64;
    int64_t tally_stride_f64;
    int64_t flags;          /* PVT_FLAG_* */
} PvtTraceParams;

/* PvtTraceParams.flags */
enum {
    /* pvt_trace_device: do NOT write the fill values into the rows of the event log that no event reached (for the
     * reference's defaults, 128 rows per ray of which a ray writes a dozen, that is most of the log).  Rows beyond
     * counts[j] of recorded ray j are then undefined; only counts[] is cleared.  For callers that read the
     * written rows only. */
    PVT_FLAG_NO_LOG_PREFILL = 1,
    /* A STREAM of tally bundles whose totals are wanted, not each bundle's own (record_every == 0, no tally sets):
     * the launch does not trace its last photons to completion.  A wave that finds no new ray parks its live
     * photons -- complete state: position, direction, wavelength, path, clock, RNG stream, step count, source,
     * first-crossing mask -- in a buffer of the scene that belongs to the HIP stream, and retires; the NEXT
     * launch on that stream (same scene, same maxsteps / emit_method; it adds to the tallies IT is given) hands
     * them to its lanes before its own rays.  Histories are unchanged bit for bit (a photon's draws depend on its
     * stream alone), integer tallies summed over the launches are identical; what disappears is the tail of
     * every launch, where a few long histories keep a few lanes busy (a fifth of all wave-iterations of a 10^6-
     * photon launch of the headline scene).  A launch WITHOUT the flag finishes everything, what it resumed
     * included; with n_rays == 0 it is the closing flush of a job.  pvt_scene_carry_pending() tells whether
     * photons are waiting.  The resuming launch must ask for the maxsteps / emit_method of the launch that parked
     * them (PVT_ERR_INVALID otherwise: the photons would silently change rules) and is never narrower than it (the
     * library widens its grid if need be); pvt_scene_carry_discard() drops parked photons of a job that was
     * abandoned.  The host-buffer entries (pvt_trace_bundle, pvt_trace_bundle_multi) ignore the flag: their scene
     * lives for one call.  (The reference has no counterpart: its bundles end when their slowest ray ends.) */
    PVT_FLAG_CARRY_OUT = 2
};

/* initial rays, world frame (exactly trace_bundle's three array arguments) */
typedef struct PvtRays {
    const double* position;   /* (n,3) */
    const double* direction;  /* (n,3) */
    const double* wavelength; /* (n)   */
} PvtRays;

/* recorder accumulators; the trace ADDS into them (caller zeroes) */
typedef struct PvtTallies {
    int64_t* rec_distinct;   /* (n_recorders)       */
    int64_t* rec_crossings;  /* (n_recorders)       */
    double* rec_sums;        /* (n_recorders,4,2)   */
    int64_t* rec_bins;       /* (total_bins)        */
} PvtTallies;

/* event log: rows = ceil(n/record_every)*max_events; row of event k of
 * recorded ray j is j*max_events+k (same packing as _kernel.pyx:1035-1047).
 * pvt_trace_* pre-fills kind/position/... with 0 and the id columns with -1.
 * (The kernel itself writes PvtEventRecords, below; these columns are made from them by a
 * second, coalesced pass.) */
typedef struct PvtEventLog {
    int32_t* counts;         /* (n_recorded) */
    uint8_t* kind;
    int32_t* hit;
    int32_t* container;
    int32_t* adjacent;
    int32_t* component;
    int32_t* source;
    double* position;        /* (rows,3) */
    double* direction;       /* (rows,3) */
    double* normal;          /* (rows,3) */
    double* wavelength;
    double* travelled;
    double* duration;
} PvtEventLog;

/* event RECORDS: the form the kernel writes.  One event = one 128-byte row (16 x uint64, little endian):
 *   word 0  hit (int32, low half) | container (high half)     word 1  adjacent | component
 *   word 2  source | kind (low byte of the high half)         words 3-5 position, 6-8 direction,
 *   9-11 normal (zeros when the event has none), 12 wavelength, 13 travelled, 14 duration (doubles)
 *   word 15 the row index itself
 * Row of event k of recorded ray j = j*max_events + k, the reference's packing (_kernel.pyx:1035-1047) with
 * the thirteen columns of a row side by side: a lane that follows one ray then writes ONE full cache line
 * per event instead of thirteen scattered column elements (4.2 x less HBM write traffic, measured).  Rows
 * k >= counts[j] are never written and hold whatever the buffer held.  pvt_unpack_records_device turns
 * records into the column arrays of PvtEventLog. */
typedef struct PvtEventRecords {
    int32_t* counts;         /* (n_recorded) events written per recorded ray */
    uint64_t* rows;          /* (n_recorded * max_events, 16) */
} PvtEventRecords;

typedef struct PvtScene PvtScene;   /* opaque: tables resident in HBM on one GPU */

int pvt_abi_version(void);
const char* pvt_last_error(void);
int pvt_device_count(void);

/* Pack the tables and upload them once to `device`. */
int pvt_scene_create(const PvtSceneTables* tables, int device, PvtScene** out);
/* Attach / replace the device-side emitter of a scene (optional). */
int pvt_scene_set_emitter(PvtScene* scene, const PvtEmitterTables* emitter);
void pvt_scene_destroy(PvtScene* scene);

/* Device-resident entry: every pointer in rays / tallies / log is a DEVICE
 * pointer owned by the caller (e.g. torch tensors) on the scene's GPU;
 * `stream` is a hipStream_t (NULL = default stream).  rays == NULL selects
 * device-side emission.  log may be NULL when record_every == 0.
 * Asynchronous: returns after enqueueing. */
int pvt_trace_device(PvtScene* scene, const PvtRays* rays, const PvtTraceParams* params,
                     const PvtTallies* tallies, const PvtEventLog* log, void* stream);
/* (With column arrays the kernel's 128-byte event records are staged in a buffer of the scene, one per stream that
 * asked for it, at most 1 GiB -- larger logs are traced over consecutive ray ranges, same results -- kept until
 * the scene is destroyed: device memory on top of the caller's 117 bytes per row.  pvt_trace_device_records has no
 * such buffer.) */

/* Same launch, the event log kept as RECORDS in caller-owned DEVICE memory (no staging, no unpack pass):
 * what a caller wants who reads only the rows that were written (the Python engine.simulate() does).
 * `records` may be NULL when record_every == 0. */
int pvt_trace_device_records(PvtScene* scene, const PvtRays* rays, const PvtTraceParams* params,
                             const PvtTallies* tallies, const PvtEventRecords* records, void* stream);

/* 1 when photons parked by the last launch on `stream` (PVT_FLAG_CARRY_OUT) wait to be resumed, else 0. */
int pvt_scene_carry_pending(PvtScene* scene, void* stream);
/* Forget the photons parked on `stream` (a job abandoned half-way: an exception between two bundles, a consumer that
 * went away).  The next launch on the stream then starts from its own rays alone. */
int pvt_scene_carry_discard(PvtScene* scene, void* stream);
/* A scene that is put aside for later (a cache of resident scenes): frees the staging buffers of pvt_trace_device (up to
 * 1 GiB per stream that asked for column arrays) and forgets parked photons on every stream.  The tables stay. */
int pvt_scene_trim(PvtScene* scene);

/* Records -> column arrays (all DEVICE pointers), one coalesced pass; `prefill` != 0 also writes the
 * reference's fill values (0 / -1) into the rows no event reached, else those rows are left alone.
 * `out->counts` is not written (the counts are `records->counts`). */
int pvt_unpack_records_device(const PvtEventRecords* records, int64_t n_recorded, int32_t max_events,
                              const PvtEventLog* out, int prefill, void* stream);

/* Host-buffer entry — the literal replacement for _kernel.trace_bundle: all
 * pointers are HOST memory; uploads, traces, downloads, synchronises.
 * `kernel_ms` (nullable) receives the HIP-event time of the trace kernel. */
int pvt_trace_bundle(const PvtSceneTables* tables, const PvtEmitterTables* emitter,
                     const PvtRays* rays, const PvtTraceParams* params,
                     const PvtTallies* tallies, const PvtEventLog* log, int device,
                     double* kernel_ms);

/* (The rays are uploaded in chunks and a chunk is traced while the next one crosses PCIe; results do not depend on the
 * split.  The device block that held the rays and tallies -- up to 1 GiB, one per device -- is kept for the next
 * host-buffer call on that device; pvt_release_cached_memory() frees what is kept.  The reference's trace_bundle keeps
 * nothing between calls either way: _kernel.pyx:1035-1066 allocates per call.) */
void pvt_release_cached_memory(void);

/* In-process multi-GPU form of pvt_trace_bundle: the bundle is split over
 * `devices[0 .. n_devices)` by contiguous index range — shard g traces the global indices
 * pvt_shard_range(n_rays, g, n_devices, record_every) with ray_offset advanced accordingly, so every ray
 * keeps the RNG stream seed + ray_offset + global index and the result does not depend on the device
 * list.  One host thread, one resident scene and one stream per entry of `devices` (an id may appear
 * several times: its shards then share that GPU).  The tallies are summed on the devices (RCCL over
 * xGMI) when the entries are different GPUs, else on the host: integer tallies exactly, the f64 moment sums up to
 * the order of addition; each shard writes the rows of its own rays into the caller's event log (inner shard
 * boundaries are multiples of record_every, so the shards' logs together ARE the single-device log).
 * `kernel_ms` (nullable) receives the longest shard's kernel time.  The reference splits a bundle over
 * OpenMP threads instead (_kernel.pyx:1074-1095); there as here the output is independent of the split. */
int pvt_trace_bundle_multi(const PvtSceneTables* tables, const PvtEmitterTables* emitter,
                           const PvtRays* rays, const PvtTraceParams* params,
                           const PvtTallies* tallies, const PvtEventLog* log,
                           const int* devices, int n_devices, double* kernel_ms);

/* How the last pvt_trace_bundle_multi of THIS thread summed the shards' tallies: 0 none yet, 1 on the host,
 * 2 on the devices with RCCL (ncclCommInitAll over the device list, ncclReduce to the first shard; taken when every
 * entry of the list is a different GPU and librccl can be loaded; PVT_MULTI_REDUCE=host|rccl overrides). */
int pvt_last_multi_reduce(void);

/* [start, stop) of shard `shard` of `n_shards` over n_rays rays; inner boundaries are rounded down to
 * multiples of `align` (pass record_every; <= 1 means no alignment).  Pure host arithmetic. */
int pvt_shard_range(int64_t n_rays, int shard, int n_shards, int64_t align, int64_t* start, int64_t* stop);

/* Device emission only (fills caller-owned DEVICE arrays); used by tests. */
int pvt_emit_device(PvtScene* scene, const PvtTraceParams* params, double* position,
                    double* direction, double* wavelength, void* stream);

/* Self-test: y[i] = f(x[i]) evaluated ON THE DEVICE (host buffers in/out).
 * fn: 0 log, 1 sin, 2 cos, 3 asin, 4 acos, 5 sqrt, 6 1/x, 7 sin*cos via sincos,
 * 8 second xoshiro256+ uniform of stream (uint64)x, 9 x/(x+3), 10-13 known-divisor divisions,
 * 14/15 sin/cos(2 pi x) and 16 sqrt((1-x)(1+x)) (composed functions of pvt_math.h), 17-19 the short
 * 1/x, x/y and sqrt(x) sequences of the kernel (operands in their normal ranges).  Lets the tests prove the bit-reproducibility premise of
 * csrc/pvt_math.h on gfx950. */
int pvt_selftest_math(int fn, const double* x_host, double* y_host, int64_t n, int device);

/* Step counters of the scene since its creation (or the last reset), summed over every launch on every stream --
 * always on, two scalar adds per wave and trip of the photon loop:
 *   out[0] wave-iterations (trips in which a wave stepped its lanes)      out[1] lane-steps (live lanes summed over them)
 *   out[2] photons finished one step early by the fused exit             out[3] waves retired
 * out[1] + out[2] is the reference's loop count (`count`, _kernel.pyx:655) summed over the photons traced so far --
 * exactly, when no photon is parked between launches (PVT_FLAG_CARRY_OUT) at the time of the call; out[1] / (64 out[0])
 * is the fraction of lanes that held a live photon when a wave stepped.  The caller synchronises the streams it
 * launched on first (the copy only orders after the null stream); `reset` != 0 clears the counters afterwards. */
int pvt_scene_counters(PvtScene* scene, uint64_t* out, int reset);

/* The clocks the scene's launches ran at, read ON THE GPU (v12; no reference counterpart -- `elapsed` there is
 * time.perf_counter around the call, pvtrace/engine/api.py:232-245).  Every workgroup reads the constant 100 MHz clock and
 * the shader clock when it starts and when its last wave leaves:
 *   out[0] shader-clock cycles and out[1] 100 MHz ticks, both summed over the lives of all workgroups since the scene's
 *   creation or the last reset of pvt_scene_counters (which clears these too): 100 MHz x out[0] / out[1] is the shader
 *   clock the launches ran at, overlapping launches included.  Synchronise first, as for pvt_scene_counters. */
int pvt_scene_clock(PvtScene* scene, uint64_t* out);

/* GPU-side span of the LAST launch on `stream` (v12): out[0] = 100 MHz time at which its workgroup 0 started, out[1] = the
 * latest time at which one of its workgroups left; (out[1] - out[0]) x 10 ns is the launch's duration as the GPU saw it.
 * A pair of HIP events around the call also counts whatever the host does between recording them (a descheduled thread
 * reads as kernel time: profiles/r06_e2e_outlier.txt).  Synchronises the stream. */
int pvt_scene_launch_span(PvtScene* scene, void* stream, uint64_t* out);

/* Launch geometry actually used by the last trace on this scene (diagnostics). */
int pvt_scene_launch_info(PvtScene* scene, int32_t* grid, int32_t* block, int32_t* lds_bytes);

/* Host-only check of the triangle BVH the library builds for mesh node `node` (no GPU needed):
 * every face appears in exactly one leaf, lies inside the boxes of its leaf and of all its
 * ancestors, the children of a record are a pair on one 64-byte line, a left child's skip link is its
 * sibling and a right child's its parent's; and the copy of the tree's top
 * levels that the trace kernel reads from LDS (cursors and links that name either array), made at five
 * sizes, walks the same records in the same order with the same successors after a hit and after a miss.
 * Returns PVT_OK and the node / leaf counts and the tree depth, or PVT_ERR_INVALID with pvt_last_error(). */
int pvt_mesh_bvh_check(const PvtSceneTables* tables, int32_t node, int32_t* n_bvh_nodes,
                       int32_t* n_leaves, int32_t* depth);

/* Host-only view of the NODE GRID the library builds for scenes of many nodes (no GPU needed).  The reference
 * intersects every node in every step (_kernel.pyx:666-680); for such scenes the trace kernel instead files every
 * node but the root under the cells of a uniform grid touched by its bounding box grown by a margin, and each
 * photon tests only the nodes filed under the cells its ray passes (results unchanged bit for bit).
 * dims[3] = cells per axis, all 0 when the scene gets no grid (few nodes, meshes, non-rigid poses; the plain node loop
 * then serves it); lo[3] / cell[3] = the grid's corner and cell edges, `guard` = the margin, `odd` = a cylinder is
 * filed (the walk's early exit is then more cautious).  `masks` (nullable) receives, per cell (x fastest), two 64-bit
 * words whose bit n says node n is filed there, up to `masks_cap` words. */
int pvt_node_grid_plan(const PvtSceneTables* tables, int32_t* dims, double* lo, double* cell, double* guard,
                       int32_t* odd, uint64_t* masks, int64_t masks_cap);

#ifdef __cplusplus
}
#endif
#endif /* PVTRACE_HIP_H */
