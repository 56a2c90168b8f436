// pvt_trace_kernel.h -- device side of the photon engine: the scene-record layout shared
// with the host packer (enums + `Lay` + `KArgs`), the per-ray RNG, table lookups, the emitter
// and the trace kernel itself.  Included once, by pvt_trace.hip (which holds the design
// overview, the host-side packing and the C ABI).
#pragma once
// Developer switches (build-time; everything else that used to be one is a constant now -- the experiments behind them are
// closed, docs/history.md): PVT_STATS (tools/: -DPVT_STATS=1 builds print per-launch lane statistics), PVT_TIMELINE (per-wave
// stamps), PVT_COUNTERS / PVT_TAIL_CALL / PVT_TAIL_ALPHA (round 5's A/B switches: profiles/r05_tail_ab.txt), and
// PVT_DEV_VARIANTS in pvt_trace.hip (fast developer builds of a few variants).
constexpr int kMeshWaves = 4;   // waves per SIMD the mesh variants are held to (registers: 512 / waves; LDS per workgroup: 160 KB / waves)
#ifndef PVT_STATS
#define PVT_STATS 0
#endif
#ifndef PVT_COUNTERS
#define PVT_COUNTERS 1   // (0: a developer build without the always-on step counters, to price them: tools/gpu_ab.sh)
#endif
// Developer-only: every wave writes wall-clock stamps (start, tables staged, first step, cursor dry, end) and its
// iteration count to KArgs::timeline (tools/gpu_wave_timeline.py)
#ifndef PVT_TIMELINE
#define PVT_TIMELINE 0
#endif

namespace {

constexpr int kBlock = 256;          // 4 wavefronts
constexpr int kChunk = 64;           // rays a wave seeds and hands out at a time
constexpr int kEndClaim = 64;   // rays per claim in the last two rounds of a launch (see the refill)
constexpr int kClaim = kChunk;   // rays a wave claims per cursor atomic: every wave of a launch adds to the SAME
                                     // word, and a device-scope atomic that returns a value costs the wave a round trip
                                     // through the fabric; claiming several chunks at once takes the cursor off the critical
                                     // path -- measured (round 3): 1, 2 and 4 chunks per claim give the same throughput at
                                     // 10^6 and 4 10^6 photons per launch, serial and pipelined: the cursor is NOT a bottleneck
constexpr int kWaves = kBlock / 64;
constexpr int kCursorSlots = 64;     // distinct HIP streams that may trace one scene concurrently
constexpr int kMaxSets = 1024;       // tally sets (bundles of a stream) one launch may serve
constexpr int kRecPlain = 1 << 30;   // flag on a recorder id in the candidate tables: matches without reading its row (host: scene_create)
constexpr int kTallyQ = 64;          // first crossings a wave parks before it computes their statistics together
constexpr int kXSlots = 72;          // LDS photon-state slots used to repack a draining workgroup (>= 64: the
                                     // last stage packs the survivors into one wave; >= 69 so that the 8 KB of
                                     // per-wave seed pools fit in the same region); small enough that FIVE
                                     // workgroups' LDS fit a CU, so a fresh workgroup of the next launch can
                                     // start while draining ones still hold theirs (+6 % on pipelined bundles)
// workgroup control words in LDS
enum { CTL_EXHAUSTED = 0, CTL_DONE = 1, CTL_IN = 2, CTL_LIVE = 6, CTL_COUNT = 16, CTL_CLOCK = 22, CTL_WORDS = 26 };   // (CTL_COUNT: three u64 sums, see KArgs::counters; CTL_CLOCK: two u64 stamps, see KArgs::stamp)
constexpr double kEps = 2.220446049250313e-13;       // _kernel.pyx:29
constexpr double kAlphaZero = 1e-8;                  // :32
constexpr double kCcm = 2.99792458e10;               // :33
constexpr double kPi = 3.14159265358979323846;
constexpr unsigned long long kEmitSalt = 0xA5A5A5A55A5A5A5Aull;

// Scene blobs are arrays of fixed-stride RECORDS (one double blob, one int32 blob), so a
// table element is addressed as base + index*stride + field with compile-time strides
// and fields; only the eight record bases below live in SGPRs (node records start at 0).
// Node records are what a photon reads of the node it is in / hits / tests, and what the intersection loop reads of
// EVERY node: 64 bytes (one scalar load in the wave-uniform loop; a stride of 72 bytes, meant to spread the per-lane
// reads of the grid walk over the LDS banks, gained 1 % there and cost the headline scene an integer multiply per
// record address).
// Rotations, refractive-index reciprocals and critical angles live in small side tables indexed by CLASS (nodes with
// bit-identical rotations / refractive indices share an entry), so a scene of 120 tiles costs 12 KB of LDS, not 56.
enum { ND_T = 0, ND_PARAMS = 3, ND_BITS = 6, ND_N = 7, ND = 8 };  // node doubles: translation of world->local, three shape
                                                                 // parameters (no shape has four), one word of {bit 0: the
                                                                 // rotation is the identity, bit for bit; bits 8-15: geometry
                                                                 // type; high half: rotation class}, refractive index
enum { NI_SURF = 0, NI_CSTART, NI_CCOUNT, NI_CREC, NI_KSTART, NI_KCOUNT, NI_MESH, NI_CAND, NI_NCLS, NI };  // node ints (NI_CSTART: id of the
                                                                 // node's first component, what events and recorders name;
                                                                 // NI_CREC: its first component RECORD -- identical components
                                                                 // share records; NI_MESH: BVH root, -1 = none; NI_CAND: block of
                                                                 // the recorder candidate tables, -1 = nobody listens to the node;
                                                                 // NI_NCLS: refractive-index class)
enum { RT_W2L = 0, RT_L2W = 9, RT = 18 };                         // Lay::rot_d records: the 3x3 blocks of world->local and local->world
enum { CD_QY = 0, CD_TAU_RAD, CD_TAU_NR, CD_PHASE, CD_ABS_SCALE, CD_EMS_SCALE_X, CD_EMS_SCALE_C,
       CD_ABS_RCP, CD_EMS_RCP_X, CD_EMS_RCP_C, CD_ABS_W, CD_EMS_W, CD };   // component doubles (*_RCP: RN(1/spacing) of an evenly spaced table, else NaN;
                                                            // *_W: the spacing w when additionally xs[i] == xs[0] + i*w bit for bit, else NaN)
enum { CI_TYPE = 0, CI_PHASE, CI_ABS_X, CI_ABS_Y, CI_ABS_N, CI_EMS_X, CI_EMS_CDF, CI_EMS_N,
       CI_ABS_G, CI_EMS_GX, CI_EMS_GC, CI_ABS_HIST, CI_EMS_HIST, CI };  // *_G*: guide tables; *_HIST: step tables
enum { RD_FACET = 0, RD_ATOL = 3, RD = 4 };                                     // recorder
enum { RI_NODE = 0, RI_EVENT, RI_HAS_FACET, RI_HSTART, RI_HN, RI_SRC_MODE, RI_SRC_ID, RI };
enum { HD_LO_A = 0, HD_HI_A, HD_LO_B, HD_HI_B, HD_RA, HD_RB, HD };               // histogram (RA/RB: RN(1/(hi-lo)), NaN = divide)
enum { HI_PA = 0, HI_PB, HI_NA, HI_NB, HI_OFF, HI };
enum { KD_FACET = 0, KD_LO = 3, KD_HI = 6, KD_REFL = 9, KD = 10 };              // coating
enum { KI_RMODE = 0, KI_TMODE, KI };

struct Lay {  // record bases (elements) inside the blobs; spectra follow the records and are
              // addressed by absolute offsets stored in the component records
    int comp_d, rec_d, hist_d, coat_d;
    int comp_i, rec_i, hist_i, coat_i;
    int cand_i;     // per node that carries recorders (NI_CAND) 7 x {start, count, bin[6]}: recorders that can fire
                    // for a (node, selector): a list to walk + facet recorders found by normal bin
    int cand_list;  // recorder ids, ascending within each (node, selector)
    int crit_d;     // (n_cls x n_cls) critical angles asin(n[a]/n[c]) (+inf where n[a] >= n[c]) by refractive-index
                    // class, or -1 when the scene has too many distinct indices for the table
    int ccrit_d;    // same shape: the cosine below which pvt_acos(cosine) exceeds that angle (host-proven
                    // threshold, NaN where it could not be proven, -inf where there is no critical angle)
    int grid_d;     // the node grid of scenes with many nodes (-1 = none; GRID variants): {lo[3], hi[3], cell[3], 1/cell[3],
                    // guard, one word of {nx | ny << 8 | nz << 16 | 64-bit words per cell << 24 | odd << 28}}, then per
                    // cell the bit mask of the nodes filed under it (x fastest)
    int rot_d;      // rotation classes x RT doubles (every node has one; the identity's is only read by the Lambertian branch)
    int ncls_d;     // refractive-index classes x {n, RN(1/n)}
    int n_cls;      // refractive-index classes (side of the crit tables)
    int by_node;    // 1: a scene of few nodes -- index classes and recorder candidate blocks are numbered like the nodes, so the
                    // lanes index the tables by node without reading NI_NCLS / NI_CAND first (one dependent LDS read fewer
                    // in the surface branch and in the tally: the headline scene)
};

struct EmitOff {  // emitter blobs (global only; read once per photon)
    int wl_value, pos_param, dir_param, l2w, spec_x, spec_cdf;         // doubles
    int wl_type, wl_spec_start, wl_spec_n, pos_type, dir_type;        // int32
};

struct KArgs {
    const double* gd;   // scene double blob (HBM)
    const int* gi;      // scene int blob
    const double* ed;   // emitter blobs (may be null)
    const int* ei;
    const pvt::BvhNode* bvh;    // triangle meshes (null when the scene has none): stay in HBM/L2
    const pvt::MeshTri* tris;
    Lay lay;
    EmitOff eoff;
    int nd, ni;         // blob lengths
    int nd_lds, ni_lds; // (variants whose tables do not fit LDS) heads of the blobs that are staged all the same: everything
                        // but the spectra and their guide tables (0: nothing)
    int n_nodes, root, n_rec, total_bins, n_coat, n_lights;
    // rays in (null -> device emission)
    const double* pos;
    const double* dir;
    const double* wl;
    unsigned int n_rays;
    unsigned int* cursor;   // [0] ray cursor, [1] claim cursor of resumed photons, [2] photons parked; (PVT_STATS builds) [2..] u64 counters
    unsigned int* cursor_next;   // the block the NEXT launch on this stream will use: cleared by this launch (no memset
                                 // kernel between launches: a one-workgroup fill cannot start while persistent workgroups
                                 // hold every slot of the chip, and the launch behind it waits with it); null = leave alone
    unsigned long long seed;       // + ray_offset folded in by the host
    unsigned long long emit_seed;  // + nothing; global index added per ray
    unsigned long long ray_offset;
    int maxsteps, max_events, emit_method;
    long long record_every;
    // outputs
    long long* rec_distinct;
    long long* rec_crossings;
    double* rec_sums;
    long long* rec_bins;
    // event log of the recorded rays: one 128-byte RECORD per event (see log_row), record of event k of recorded
    // ray j at row j*max_events + k; log_counts[j] = events written.  The reference's column arrays are made from
    // these by unpack_log_kernel.
    unsigned long long* log_rows;
    int* log_counts;
    // tally sets (0 = the launch is one bundle): rays [j*set_size, (j+1)*set_size) are bundle j of a stream of
    // equal bundles; a workgroup serves ONE set (its own ray cursor, its own slice of the tally arrays)
    unsigned int set_size;
    int wgs_per_set;
    long long set_stride_i, set_stride_d;   // elements between consecutive sets in rec_distinct/crossings/bins and rec_sums
    // The root is visited LAST and only by the lanes that need its exact distance (0 = off; 1 = box root,
    // 2 = sphere root, `lazy_k` = 1/(2 radius)); see the node loop
    int lazy_root;
    double lazy_k;
    // 1: the scene is one unrotated box inside a lazy root with an empty, unobserved medium; a photon that leaves
    // the box's surface outwards, provably clear of it, is finished (see the surface branch)
    int fuse_exit;
    int bins_in_lds;
    int xslots;   // photon-state slots in LDS for drain-phase consolidation (0 = off)
    int tq_pos;   // 1: a histogram reads x, y or z -- queued first crossings carry the local position too
    // Photons carried from launch to launch of a stream of bundles (tally launches; PVT_FLAG_CARRY_OUT): a wave
    // that finds the ray cursor dry PARKS its live photons -- complete state, one record of kCarryStride u64 words
    // per photon -- in `carry_out` (count in cursor[2], one atomic per wave) and retires; the next
    // launch on the stream hands them to its lanes first (`carry_in`, count `*carry_in_count`, claimed through
    // cursor[1]) before it touches its own rays.  A launch then has no drain phase: the low-occupancy iterations at
    // the end of every bundle (a fifth of all wave-iterations of a 10^6-photon launch) run once per JOB.
    unsigned long long* carry_in;
    unsigned long long* carry_out;
    const unsigned int* carry_in_count;
    unsigned int carry_cap;
    int carry_flags;   // 1: resume parked photons first   2: park at exhaustion
    unsigned long long* timeline;   // (PVT_TIMELINE builds) 8 words per wave
    // Device emission (EMIT variants): a wave samples the 64 rays of a chunk when it CLAIMS the chunk -- every lane busy,
    // and the sampler's code (a fifth of the kernel's text) out of the step loop -- and parks them in its own
    // [7][64] doubles here (position, direction, wavelength; global memory, L2-resident); refilled lanes read theirs.
    double* emit_pool;
    // Mesh walks: byte offset in LDS of the workgroup's [kMeshQ][kBlock] words of noted leaves (-1: none)
    int meshq_off;
    // Mesh walks: the TOP of every tree (pvt_bvh.h: stage_top) is copied to LDS when a launch starts -- `top_n` records
    // from `bvh_top` to byte offset `top_off`; cursors with pvt::kTopFlag set index that copy
    int top_off, top_n;
    const pvt::BvhNode* bvh_top;
    // Step counters of the scene, always on (pvt_scene_counters): 64 rows (blockIdx & 63) of eight u64 words
    //   [0] wave-iterations: trips of the photon loop in which a wave stepped its lanes
    //   [1] lane-steps: live lanes summed over those trips = the reference's loop count `_kernel.pyx:655` summed over the
    //       photons, except for [2]
    //   [2] fused exits: photons finished one step early by the fused exit (their last, empty step is not run)
    //   [3] waves retired
    //   [4] shader-clock cycles and [5] 100 MHz ticks, summed over the workgroups' lives (see `stamp`)
    // Kept per LANE in three vector registers (the loop is short of scalar ones): one add per trip, one when a photon
    // ends (its own step count `count` is what is added), one in the fused exit; summed in LDS when a wave retires, four
    // atomics per workgroup.  null = off.
    unsigned long long* counters;
    // Clocks of the launch, always on, read on the GPU itself (a pair of HIP events also times whatever the HOST does between
    // recording them: a descheduled thread reads as kernel time).  Thread 0 of every workgroup reads the constant 100 MHz
    // clock (s_memrealtime) and the shader clock (s_memtime) when the workgroup has staged its tables, its last wave reads
    // them again when it leaves: rows [4] / [5] of `counters` sum the shader cycles / the 100 MHz ticks of the workgroups'
    // lives (their ratio is the shader clock the launches ran at), and `stamp` -- two words per HIP stream -- gets the
    // 100 MHz time at which workgroup 0 started (a plain store) and the latest at which a workgroup left (atomic max):
    // after the stream is synchronised, the GPU-side span of its last launch (pvt_scene_launch_span).  null = off.
    unsigned long long* stamp;
    // The tail function's own lazy root (same codes as `lazy_root`, which is 0 when somebody looks at where photons leave
    // the scene -- an event log, an `exit` recorder): there the root's crossing is skipped only when another crossing is known
    // to lie before it, and worked out exactly when it is the nearest one -- among a lone wave's few photons hardly ever.
    int lazy_tail;
};
constexpr int kMeshQ = 8;    // leaves a lane notes before its triangles are tested
constexpr int kCarryBase = 14;     // u64 words of a parked photon before its seen-mask
constexpr int kCarryStride = 18;   // words per parked photon (room for the four-word mask of scenes with > 64 recorders)

// ------------------------------------------------------------------ RNG
struct Rng {
    unsigned long long s0, s1, s2, s3;
};
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& st) {
    st += 0x9E3779B97F4A7C15ull;
    unsigned long long z = st;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ void rng_seed(Rng& r, unsigned long long seed) {
    unsigned long long st = seed;
    r.s0 = splitmix64(st);
    r.s1 = splitmix64(st);
    r.s2 = splitmix64(st);
    r.s3 = splitmix64(st);
}
__device__ __forceinline__ double rng_uniform(Rng& r) {
    unsigned long long result = r.s0 + r.s3;
    unsigned long long t = r.s1 << 17;
    r.s2 ^= r.s0;
    r.s3 ^= r.s1;
    r.s1 ^= r.s2;
    r.s0 ^= r.s3;
    r.s2 ^= t;
    r.s3 = (r.s3 << 45) | (r.s3 >> 19);
    // (double)(result >> 11) * 2^-53, as the reference writes it, is exact at every step (a 53-bit integer,
    // then a power of two); so is this split into the top 32 and the next 21 bits, one conversion shorter
    const double hi = (double)(unsigned int)(result >> 32), lo = (double)(unsigned int)((unsigned int)result >> 11);
    return __builtin_fma(lo, 1.0 / 9007199254740992.0, hi * (1.0 / 4294967296.0));
}

// RN(1/x) for a finite normal x whose reciprocal is normal too: the compiler's IEEE division sequence
// (reciprocal estimate, two Newton steps, quotient, residual, final fused correction) without the operand
// scaling and the special-case fix-up that only matter outside that range -- same intermediate values, same
// result, four instructions fewer.  Zero, infinities and NaN come out as garbage (callers never use them).
// RN(x/y) likewise, for finite normal y with a normal reciprocal and x zero or normal with |x| > 1e-280, the
// quotient neither overflowing nor subnormal (wavelengths, Fresnel amplitudes, free paths: many orders of
// magnitude inside): three instructions fewer than the general sequence.
__device__ __forceinline__ double div_normal(double x, double y) {
    double r = __builtin_amdgcn_rcp(y);
    double e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-y, r, 1.0);
    r = __builtin_fma(r, e, r);
    const double q = x * r;
    e = __builtin_fma(-y, q, x);
    return __builtin_fma(e, r, q);
}
// RN(sqrt(x)) for x zero, or normal and not below 2^-767 (NaN for negative x, like the library): the compiler's
// IEEE square-root sequence (reciprocal-root estimate, one coupled Newton step, two fused residual corrections,
// zeros and +inf passed through) without the scaling of tiny operands.  The operands here are 1 - c^2 and
// (1 - c)(1 + c) for cosines and sines: zero, or 2^-53 and above.
__device__ __forceinline__ double sqrt_normal(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;   // -0, +0, +inf
}
__device__ __forceinline__ double sqrt1m2_normal(double c) { return sqrt_normal((1.0 - c) * (1.0 + c)); }   // pvt_sqrt1m2
__device__ __forceinline__ double rcp_normal(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(e, r, r);
}

// ------------------------------------------------------------ table access
// TAB_LDS (0 / 1 / 2): divergent reads come from global memory / from the LDS copy of the blobs / records and class
// tables from LDS (the heads of the blobs, KArgs::nd_lds) and the spectra with their guide tables -- too large to
// stage -- from global memory; uniform reads always come
// from the global blob so the compiler can use scalar loads.
// Read-only scene blobs viewed through the CONSTANT address space: a load whose
// address is wave-uniform then becomes an s_load through the scalar cache (SGPR result,
// no VGPR address arithmetic) instead of a 64-lane global_load of one address.  The
// blobs are never written while a trace kernel runs, which is what makes this legal.
typedef const __attribute__((address_space(4))) double* CDoubles;
typedef const __attribute__((address_space(4))) int* CInts;

template <int TAB_LDS>
struct Tables {
    CDoubles gd;
    CInts gi;
    const double* ld;  // LDS copies (TAB_LDS != 0)
    const int* li;
    const double* __restrict__ hd;
    const int* __restrict__ hi;
    __device__ __forceinline__ double du(int i) const { return gd[i]; }  // uniform index
    __device__ __forceinline__ int iu(int i) const { return gi[i]; }
    // per-lane index: records and class tables (dv / iv), spectra and their guide tables (sd / si)
    __device__ __forceinline__ double dv(int i) const { return TAB_LDS != 0 ? ld[i] : hd[i]; }
    __device__ __forceinline__ int iv(int i) const { return TAB_LDS != 0 ? li[i] : hi[i]; }
    __device__ __forceinline__ double sd(int i) const { return TAB_LDS == 1 ? ld[i] : hd[i]; }
    __device__ __forceinline__ int si(int i) const { return TAB_LDS == 1 ? li[i] : hi[i]; }
};

struct V3 {
    double x, y, z;
};
__device__ __forceinline__ double dot3(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// RN(x / y) for a divisor known in advance, given z = RN(1/y) computed once on the host:
// five multiply-adds instead of the ~25-instruction IEEE division expansion.  q1 = RN(x z) is
// within 1.5 ulp of x/y; one residual correction (r = x - q y, exact up to a rounding that
// cannot matter at that magnitude) makes q2 the rounding of a value within 2^-52 ulp of x/y, hence
// a faithful quotient; Markstein's theorem (z correctly rounded, q faithful, residual by FMA)
// then makes the second correction EXACTLY RN(x/y).  Preconditions: y, z finite non-zero
// normal, x finite, no underflow of the quotient (durations, indices and bin coordinates are
// many orders of magnitude inside the normal range).  A zero x may come back with the other
// sign (every use adds it or truncates it).  `tests/test_gpu_parity.py` checks the sequence
// against the host's division on random and adversarial operands.
__device__ __forceinline__ double div_known(double x, double y, double z) {
    double q = x * z;
    double r = __builtin_fma(-q, y, x);
    q = __builtin_fma(r, z, q);
    r = __builtin_fma(-q, y, x);
    return __builtin_fma(r, z, q);
}
constexpr double kRcpCcm = 1.0 / kCcm;   // correctly rounded by the compiler

// np.interp-like clamped interpolation (_kernel.pyx:219-238).  The reference bisects the
// whole table for `lo` = the largest index with xs[lo] <= x; that index is unique, so any
// search that finds it gives identical results.  Here a host-built guide table (n buckets of
// equal width over [xs[0], xs[n-1]], entry b = largest i with xs[i] <= left edge of bucket b)
// brackets the answer to a bucket first, the bracket is VALIDATED against the table (falls
// back to the full range if rounding put x in a neighbouring bucket), and the same bisection
// runs inside the bracket: typically 0-1 steps instead of ~log2(n) dependent LDS reads.
// `w` (NaN = no): the abscissae are xs[0] + i*w bit for bit AND every interval has the bits
// of w (both checked by the host), so the reference's index is found by arithmetic — no table
// walk, no dependent LDS reads — and validated against the (computed) neighbours.  `yw` likewise for
// the ordinates (the inverse-CDF lookup returns wavelengths of an evenly spaced grid).
template <int TAB_LDS>
__device__ __forceinline__ double interp_clamped(const Tables<TAB_LDS>& T, double x, int xs, int ys, int n,
                                                 int guide, double scale, int hist, double rcp,
                                                 double w = __builtin_nan(""), double yw = __builtin_nan("")) {
    auto end = [&](int i) { return T.sd(i); };
    if (n == 1) return end(ys);
    // a table on a proven even grid (w, yw) is stored as its first value: the last one is computed, same bits
    const bool even = !hist && w == w;
    const double x0 = end(xs), xl = even ? x0 + (double)(n - 1) * w : end(xs + n - 1);
    if (x <= x0) return end(ys);
    if (hist ? x > xl : x >= xl) return yw == yw ? end(ys) + (double)(n - 1) * yw : end(ys + n - 1);  // step tables search x == xl (plateaus)
    if (even) {
        // the host has checked, with this very sequence of operations, that it lands on the reference's
        // bisection index for every x of the table (pvt_trace.hip: even_w)
        int i = (int)((x - x0) * rcp);   // in [0, n-1] for x0 < x < xl; n-1 (x a rounding below xl) is repaired below
        double xlo = x0 + (double)i * w, xhi = x0 + (double)(i + 1) * w;
        // the product is within an ulp or two of the true quotient: only an x within rounding of a grid point can
        // come out one interval off, and the wave skips the repair unless one of its lanes holds such an x
        if (__ballot(x < xlo || !(x < xhi)) != 0ull) {
            if (x < xlo) { i -= 1; xhi = xlo; xlo = x0 + (double)i * w; }
            else if (!(x < xhi)) { i += 1; xlo = xhi; xhi = x0 + (double)(i + 1) * w; }
        }
        const double ylo = T.sd(ys + i), yhi = T.sd(ys + i + 1);
        return ylo + div_known((yhi - ylo) * (x - xlo), xhi - xlo, rcp);
    }
    int b = (int)((x - x0) * scale);
    b = b < 0 ? 0 : (b > n - 2 ? n - 2 : b);
    int lo = T.si(guide + b), hi = T.si(guide + b + 1) + 1;
    if (hi > n - 1) hi = n - 1;
    // the abscissae travel with the indices, so nothing is re-read after the search
    double xlo = T.sd(xs + lo), xhi = T.sd(xs + hi);
    if (hist) {
        // histogram-sampled table (extension; Python Distribution's hist branch): the value of
        // the first abscissa >= x, i.e. ys[#{xs_i < x}] — same bracket, strict comparison
        if (!(xlo < x)) lo = 0;          // xs[0] < x is known here
        if (!(x <= xhi)) hi = n - 1;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (T.sd(xs + mid) < x) lo = mid; else hi = mid;
        }
        return T.sd(ys + hi);
    }
    if (__ballot(!(xlo <= x) || !(x < xhi)) != 0ull) {   // (rounding put some lane's x in a neighbouring bucket: rare)
        if (!(xlo <= x)) { lo = 0; xlo = x0; }
        if (!(x < xhi)) { hi = n - 1; xhi = xl; }
    }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        const double xm = T.sd(xs + mid);
        if (xm <= x) { lo = mid; xlo = xm; } else { hi = mid; xhi = xm; }
    }
    double ylo, yhi;
    if (yw == yw) {   // ordinates of an evenly spaced grid: computed, same bits as the table's
        const double y0 = end(ys);
        ylo = y0 + (double)lo * yw; yhi = y0 + (double)hi * yw;
    } else {
        ylo = T.sd(ys + lo); yhi = T.sd(ys + hi);
    }
    if (xhi == xlo) return ylo;
    // evenly spaced abscissae (every interval has the same bits, checked by the host): the
    // divisor is known in advance, see div_known
    const double num = (yhi - ylo) * (x - xlo), width = xhi - xlo;
    return ylo + (rcp == rcp ? div_known(num, width, rcp) : num / width);
}
// same, tables in global memory (emitter spectra)
__device__ __forceinline__ double interp_global(const double* xs, const double* ys, int n, double x) {
    if (n == 1) return ys[0];
    if (x <= xs[0]) return ys[0];
    if (x >= xs[n - 1]) return ys[n - 1];
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (xs[mid] <= x) lo = mid; else hi = mid;
    }
    if (xs[hi] == xs[lo]) return ys[lo];
    return ys[lo] + (ys[hi] - ys[lo]) * (x - xs[lo]) / (xs[hi] - xs[lo]);
}

__device__ __forceinline__ V3 sphere_direction(double theta, double phi) {
    double st, ct, sp, cp;
    pvt_sincos(theta, &st, &ct);
    pvt_sincos(phi, &sp, &cp);
    return V3{st * cp, st * sp, ct};
}

// Cosine-weighted direction about +z from two draws (material/utils.py:176-186: theta = asin(sqrt(p1)), phi = 2 pi p2,
// then sin / cos of both): the compositions evaluated directly, as everywhere else in portable arithmetic --
// sin(asin s) = s, cos(asin s) = pvt_sqrt1m2(s), sin / cos(2 pi u) = pvt_sincos2pi(u).
__device__ __forceinline__ V3 lambert_direction(double p1, double p2) {
    const double st = pvt_sqrt(p1), ct = sqrt1m2_normal(st);
    double sp, cp;
    pvt_sincos2pi(p2, &sp, &cp);
    return V3{st * cp, st * sp, ct};
}

// phase functions (_kernel.pyx:455-476); draw order is part of the contract
__device__ __forceinline__ V3 sample_phase(int type, double param, Rng& rng) {
    // the polar angle is sampled through its cosine (HG, isotropic) or its sine (cone); the other one is the
    // composition sin(acos c) = cos(asin c) = pvt_sqrt1m2(c) (pvt_math.h) -- no acos / asin / sincos of theta
    // (the azimuth is 2 pi `turn`: its sine and cosine come from pvt_sincos2pi, exact quadrant reduction)
    double turn, cos_t, sin_t;
    if (type == PVT_PHASE_HG && pvt_fabs(param) >= kEps) {
        double g = param;
        double g1 = rng_uniform(rng);
        double s = 2.0 * g1 - 1.0;
        double q = (1.0 - g * g) / (1.0 + g * s);
        cos_t = 1.0 / (2.0 * g) * (1.0 + g * g - q * q);
        turn = rng_uniform(rng);
        sin_t = pvt_sqrt1m2(cos_t);
    } else if (type == PVT_PHASE_CONE) {
        double g1 = rng_uniform(rng), g2 = rng_uniform(rng);
        sin_t = pvt_sqrt(g1) * pvt_sin(param);
        turn = g2;
        cos_t = pvt_sqrt1m2(sin_t);
    } else {
        double g1 = rng_uniform(rng), g2 = rng_uniform(rng);
        turn = g1;
        cos_t = 2.0 * g2 - 1.0;
        sin_t = pvt_sqrt1m2(cos_t);
    }
    double sp, cp;
    pvt_sincos2pi(turn, &sp, &cp);
    return V3{sin_t * cp, sin_t * sp, cos_t};
}

// ----------------------------------------------------------- emission
// One ray from its own stream (mirrors oracle pvt_oracle_emit; distributions of
// reference pvtrace/engine/emit.py:22-89).
__device__ __forceinline__ void emit_one(const KArgs& A, unsigned long long gi, V3& pos, V3& dir, double& wl) {
    const double* ed = A.ed;
    const int* ei = A.ei;
    const EmitOff& E = A.eoff;
    Rng rng;
    rng_seed(rng, (A.emit_seed + gi) ^ kEmitSalt);
    int li = (int)(gi % (unsigned long long)A.n_lights);
    const int wt = ei[E.wl_type + li];
    if (wt == PVT_WL_SPECTRUM || wt == PVT_WL_SPECTRUM_HIST) {
        double u = rng_uniform(rng);
        int s = ei[E.wl_spec_start + li];
        const int n = ei[E.wl_spec_n + li];
        if (wt == PVT_WL_SPECTRUM_HIST) {
            // histogram-sampled spectrum: the abscissa at index #{cdf_i < u} (numpy.searchsorted, left), the last one
            // when u lies above the table (pvtrace/material/distribution.py:171-176)
            const double* cdf = ed + E.spec_cdf + s;
            int lo = 0, hi = n;   // cdf[i] < u for i < lo, cdf[i] >= u for i >= hi
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] < u) lo = mid + 1; else hi = mid;
            }
            wl = ed[E.spec_x + s + (lo < n ? lo : n - 1)];
        } else {
            wl = interp_global(ed + E.spec_cdf + s, ed + E.spec_x + s, n, u);
        }
    } else {
        wl = ed[E.wl_value + li];
    }
    V3 lp{0.0, 0.0, 0.0}, ld{0.0, 0.0, 1.0};
    const double* pp = ed + E.pos_param + li * 3;
    int pt = ei[E.pos_type + li];
    if (pt == PVT_POS_RECT) {
        lp.x = -pp[0] + 2.0 * pp[0] * rng_uniform(rng);
        lp.y = -pp[1] + 2.0 * pp[1] * rng_uniform(rng);
    } else if (pt == PVT_POS_CIRCLE) {
        double ang = 2.0 * kPi * rng_uniform(rng);
        double rad = pvt_sqrt(rng_uniform(rng)) * pp[0];
        double s, c;
        pvt_sincos(ang, &s, &c);
        lp.x = rad * c;
        lp.y = rad * s;
    } else if (pt == PVT_POS_CUBE) {
        lp.x = -pp[0] + 2.0 * pp[0] * rng_uniform(rng);
        lp.y = -pp[1] + 2.0 * pp[1] * rng_uniform(rng);
        lp.z = -pp[2] + 2.0 * pp[2] * rng_uniform(rng);
    }
    double prm = ed[E.dir_param + li];
    int dt = ei[E.dir_type + li];
    if (dt == PVT_DIR_CONE) ld = sample_phase(PVT_PHASE_CONE, prm, rng);
    else if (dt == PVT_DIR_ISOTROPIC) ld = sample_phase(PVT_PHASE_ISOTROPIC, 0.0, rng);
    else if (dt == PVT_DIR_HG) ld = sample_phase(PVT_PHASE_HG, prm, rng);
    else if (dt == PVT_DIR_LAMBERTIAN) {
        double p1 = rng_uniform(rng), p2 = rng_uniform(rng);
        ld = lambert_direction(p1, p2);
    }
    const double* m = ed + E.l2w + li * 16;
    pos.x = m[0] * lp.x + m[1] * lp.y + m[2] * lp.z + m[3];
    pos.y = m[4] * lp.x + m[5] * lp.y + m[6] * lp.z + m[7];
    pos.z = m[8] * lp.x + m[9] * lp.y + m[10] * lp.z + m[11];
    dir.x = m[0] * ld.x + m[1] * ld.y + m[2] * ld.z;
    dir.y = m[4] * ld.x + m[5] * ld.y + m[6] * ld.z;
    dir.z = m[8] * ld.x + m[9] * ld.y + m[10] * ld.z;
}

// The sampler as a FUNCTION (trace kernels, device emission): called by the wave that claims a chunk of rays, once per 64
// rays -- its 15 KB of code then sit beside the trace kernel's text instead of inside it (every EMIT variant as large
// as its array-input twin plus a call), at the price of the caller's live registers going through scratch around the call.
__device__ __attribute__((noinline)) void emit_chunk(const KArgs* A, unsigned long long gi, double* pool) {
    V3 ep, ed;
    double ew;
    emit_one(*A, gi, ep, ed, ew);
    pool[0] = ep.x; pool[64] = ep.y; pool[128] = ep.z;
    pool[192] = ed.x; pool[256] = ed.y; pool[320] = ed.z;
    pool[384] = ew;
}

__global__ void __launch_bounds__(kBlock) emit_kernel(KArgs A, double* opos, double* odir, double* owl) {
    unsigned int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= A.n_rays) return;
    V3 p, d;
    double wl;
    emit_one(A, A.ray_offset + i, p, d, wl);
    opos[i * 3] = p.x; opos[i * 3 + 1] = p.y; opos[i * 3 + 2] = p.z;
    odir[i * 3] = d.x; odir[i * 3 + 1] = d.y; odir[i * 3 + 2] = d.z;
    owl[i] = wl;
}

// Self-test hook: evaluates one elementary function per element on the device so the
// tests can prove the premise of the whole parity scheme (IEEE divide / sqrt and the
// pvt_math.h functions produce the same bits on gfx950 as on the host).
__global__ void __launch_bounds__(kBlock) math_kernel(int fn, const double* x, double* y, long long n) {
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double v = x[i], r;
    switch (fn) {
        case 0: r = pvt_log(v); break;
        case 1: r = pvt_sin(v); break;
        case 2: r = pvt_cos(v); break;
        case 3: r = pvt_asin(v); break;
        case 4: r = pvt_acos(v); break;
        case 5: r = pvt_sqrt(v); break;
        case 6: r = 1.0 / v; break;
        case 7: { double s, c; pvt_sincos(v, &s, &c); r = s * c; break; }
        case 8: { Rng g; rng_seed(g, (unsigned long long)v); rng_uniform(g); r = rng_uniform(g); break; }
        case 9: r = v / (v + 3.0); break;
        case 10: r = div_known(v, kCcm, kRcpCcm); break;
        case 11: r = div_known(v, 1.5, 1.0 / 1.5); break;
        case 12: r = div_known(v, 800.0 - 400.0, 1.0 / (800.0 - 400.0)); break;
        case 14: { double sn, cs; pvt_sincos2pi(v, &sn, &cs); r = sn; break; }
        case 15: { double sn, cs; pvt_sincos2pi(v, &sn, &cs); r = cs; break; }
        case 16: r = pvt_sqrt1m2(v); break;
        case 17: r = rcp_normal(v); break;
        case 18: r = div_normal(v, v * 0.7310585786300049 + 0.25); break;
        case 19: r = sqrt_normal(v); break;
        default: { double d = v * 0.7310585786300049 + 0.25; r = div_known(v, d, 1.0 / d); break; }
    }
    y[i] = r;
}

// ----------------------------------------------------------- event log
// One event = one 128-byte record = one cache line, written by its lane with eight 16-byte stores:
//   words (u64)  0: hit | container<<32   1: adjacent | component<<32   2: source | kind<<32
//                3-5 position   6-8 direction   9-11 normal (zeros when the event has none)
//                12 wavelength  13 travelled  14 duration  15 row index (j*max_events + k)
// The reference keeps thirteen column arrays indexed [ray][event] (_kernel.pyx:562-597, :1035-1047); written
// from 64 lanes that follow 64 different rays, every 1-, 4-, 8- and 24-byte column store lands in a memory
// sector of its own (measured: 4.2 x the algorithmic bytes reach HBM).  A record is one full line instead; the
// column arrays, for callers that want them on the device, are made by unpack_log_kernel below with
// consecutive lanes on consecutive rows.
constexpr int kRecWords = 16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 pack_dd(double a, double b) {
    const unsigned long long ua = pvt_d2u(a), ub = pvt_d2u(b);
    return u32x4{(unsigned int)ua, (unsigned int)(ua >> 32), (unsigned int)ub, (unsigned int)(ub >> 32)};
}
// (VIA_A: inside a called function -- the tail function -- the kernel-argument segment pointer is not to be had from the
// intrinsic (measured: garbage in a callee, ROCm 7.0); the caller's `A` already IS that segment, handed down explicitly)
template <bool RECORD, bool VIA_A = false>
__device__ __forceinline__ void log_row(const KArgs& A, int rec_slot, int& nev, int kind, int hit,
                                        int container, int adjacent, int component, int source,
                                        const V3& pos, const V3& dir, bool has_normal, const V3& nrm,
                                        double wl, double travelled, double duration) {
    if constexpr (RECORD) {
        if (rec_slot < 0 || nev >= A.max_events) return;   // rec_slot: index of the recorded ray, -1 = not recorded
        const long long row = (long long)rec_slot * A.max_events + nev;
        // the log pointer is read from the kernel-argument segment at the point of use (scalar load) instead of
        // living in two scalar registers across the whole loop
        const __attribute__((address_space(4))) KArgs* ak =
            VIA_A ? (const __attribute__((address_space(4))) KArgs*)(unsigned long long)&A
                  : (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ak));
        u32x4* dst = reinterpret_cast<u32x4*>(ak->log_rows + row * kRecWords);
        const unsigned long long p0 = pvt_d2u(pos.x);
        const V3 n = has_normal ? nrm : V3{0.0, 0.0, 0.0};
        dst[0] = u32x4{(unsigned int)hit, (unsigned int)container, (unsigned int)adjacent, (unsigned int)component};
        dst[1] = u32x4{(unsigned int)source, (unsigned int)kind, (unsigned int)p0, (unsigned int)(p0 >> 32)};
        dst[2] = pack_dd(pos.y, pos.z);
        dst[3] = pack_dd(dir.x, dir.y);
        dst[4] = pack_dd(dir.z, n.x);
        dst[5] = pack_dd(n.y, n.z);
        dst[6] = pack_dd(wl, travelled);
        dst[7] = pack_dd(duration, pvt_u2d((unsigned long long)row));
        nev += 1;
    }
}

// Records -> the reference's column arrays (PvtEventLog), one thread per row, consecutive lanes on consecutive
// rows of the same recorded ray, so every column is written in contiguous runs.  Rows a ray did not write
// (k >= counts[j]) get the reference's fill values (0, ids -1; _kernel.pyx:1035-1047) when `prefill`, else they
// are left alone.  grid.x walks the recorded rays (`rays_per_block` each), grid.y the events in chunks of 256.
__global__ void __launch_bounds__(kBlock) unpack_log_kernel(const unsigned long long* __restrict__ rows_in,
                                                            const int* __restrict__ counts, PvtEventLog out,
                                                            long long n_recorded, int max_events, int rays_per_block,
                                                            int prefill) {
    const int t = threadIdx.x;
    const int jr = rays_per_block > 1 ? t / max_events : 0;
    const int k = (rays_per_block > 1 ? t - jr * max_events : t) + (int)blockIdx.y * kBlock;
    const long long j = (long long)blockIdx.x * rays_per_block + jr;
    if (jr >= rays_per_block || j >= n_recorded || k >= max_events) return;
    const bool valid = k < counts[j];
    if (!valid && !prefill) return;
    const long long row = j * max_events + k;
    u32x4 q[8];
    if (valid) {
        const u32x4* src = reinterpret_cast<const u32x4*>(rows_in + row * kRecWords);
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = __builtin_nontemporal_load(src + i);
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) q[i] = u32x4{0u, 0u, 0u, 0u};
        q[0] = u32x4{~0u, ~0u, ~0u, ~0u};
        q[1].x = ~0u;
    }
    auto dd = [&](int word) -> double {   // u64 word `word` of the record as a double
        const u32x4 v = q[word >> 1];
        const unsigned long long u = (word & 1) ? ((unsigned long long)v.w << 32 | v.z) : ((unsigned long long)v.y << 32 | v.x);
        return pvt_u2d(u);
    };
    out.hit[row] = (int)q[0].x; out.container[row] = (int)q[0].y;
    out.adjacent[row] = (int)q[0].z; out.component[row] = (int)q[0].w;
    out.source[row] = (int)q[1].x; out.kind[row] = (uint8_t)q[1].y;
    out.position[row * 3] = dd(3); out.position[row * 3 + 1] = dd(4); out.position[row * 3 + 2] = dd(5);
    out.direction[row * 3] = dd(6); out.direction[row * 3 + 1] = dd(7); out.direction[row * 3 + 2] = dd(8);
    out.normal[row * 3] = dd(9); out.normal[row * 3 + 1] = dd(10); out.normal[row * 3 + 2] = dd(11);
    out.wavelength[row] = dd(12); out.travelled[row] = dd(13); out.duration[row] = dd(14);
}

// Records -> PACKED records: the counts[j] rows of recorded ray j, rays one after the other (`first[j]` = rows of the
// rays before j, an exclusive prefix sum made by the host); same thread mapping as unpack_log_kernel.  What the
// host-buffer entry moves over PCIe.
__global__ void __launch_bounds__(kBlock) pack_log_kernel(const unsigned long long* __restrict__ rows_in,
                                                          const int* __restrict__ counts, const long long* __restrict__ first,
                                                          unsigned long long* __restrict__ packed, long long n_recorded,
                                                          int max_events, int rays_per_block) {
    const int t = threadIdx.x;
    const int jr = rays_per_block > 1 ? t / max_events : 0;
    const int k = (rays_per_block > 1 ? t - jr * max_events : t) + (int)blockIdx.y * kBlock;
    const long long j = (long long)blockIdx.x * rays_per_block + jr;
    if (jr >= rays_per_block || j >= n_recorded || k >= max_events || k >= counts[j]) return;
    const u32x4* src = reinterpret_cast<const u32x4*>(rows_in + (j * max_events + k) * kRecWords);
    u32x4* dst = reinterpret_cast<u32x4*>(packed + (first[j] + k) * kRecWords);
#pragma unroll
    for (int i = 0; i < 8; i++) dst[i] = __builtin_nontemporal_load(src + i);
}

// LDS accumulator layout (per workgroup), after the table copies:
//   f64 sums[n_rec*8] | u64 cross[n_rec] | u32 distinct[n_rec] | u32 bins[total_bins] (if they fit)
struct Accum {
    unsigned int* cross;
    unsigned int* distinct;
    double* sums;
    unsigned int* bins;  // null -> straight to global
};

template <int SEENW>
struct Seen {
    unsigned long long w[SEENW];
};

// --------------------------------------------------------------- kernel
// MESH: the scene has triangle-mesh nodes.  The BVH walk costs ~35 VGPRs, so scenes made of
// analytic shapes run the variant compiled without it (one more wave per SIMD).
// GRID: scenes of many nodes -- every lane finds the nodes its ray can cross through a uniform grid (see the node loop).
// TAIL: the same loop as a FUNCTION for the last wave of a draining workgroup (`tail_run`, below): no rays to claim, no
// rendezvous -- it takes the `tail_total` photons its caller left in the exchange buffer and steps them until none is left.
#ifndef PVT_TAIL_CALL
#define PVT_TAIL_CALL 1
#endif
#ifndef PVT_TAIL_HOIST
#define PVT_TAIL_HOIST 1   // (0: a developer build whose tail function reads the hit node's record where the kernels' loop does)
#endif
#ifndef PVT_TAIL_LAZY
#define PVT_TAIL_LAZY 1   // (0: a developer build whose tail function intersects the root wherever the kernels' loop does)
#endif
#ifndef PVT_TAIL_ALPHA
#define PVT_TAIL_ALPHA 1   // (0: a developer build whose tail function looks the absorption coefficients up in every step)
#endif
template <bool RECORD, int TAB_LDS, int SEENW, bool MESH, bool GRID>
__device__ void tail_run(const KArgs* kernel_args, int total, unsigned int lds);

// A wave leaves its workgroup: the LAST one to do so adds the workgroup's accumulators (LDS) to the launch's outputs -- one
// global atomic per non-zero slot -- and its step counters to the scene's.  (No closing barrier: retiring waves must never
// be counted by the drain-phase rendezvous barriers of the waves still running.)  A function of its own, given nothing but
// the kernel's argument pointer (made a scalar and a pointer to constant memory again, as in tail_run): where the
// accumulators lie in LDS is worked out here, from the same quantities trace_body lays them out by.
template <int TAB_LDS>
__device__ __attribute__((noinline)) void leave_workgroup(const KArgs* kernel_args) {
    extern __shared__ double smem[];
    const unsigned long long bits = (unsigned long long)kernel_args;
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)bits), hi = __builtin_amdgcn_readfirstlane((unsigned int)(bits >> 32));
    const __attribute__((address_space(4))) KArgs* ak = (const __attribute__((address_space(4))) KArgs*)(((unsigned long long)hi << 32) | lo);
    const KArgs& A = *(const KArgs*)ak;
    const int lane = threadIdx.x & 63;
    // (the layout of trace_body: tables | sums | crossings | distinct | bins | control words)
    const int nd_lds = TAB_LDS == 1 ? A.nd : TAB_LDS == 2 ? A.nd_lds : 0;
    const int ni_stage = TAB_LDS == 1 ? A.ni : TAB_LDS == 2 ? A.ni_lds : 0;
    const int ni_lds = (ni_stage + 1) & ~1;
    int* lds_i = reinterpret_cast<int*>(smem + nd_lds);
    double* acc_sums = reinterpret_cast<double*>(lds_i + ni_lds);
    unsigned long long* acc_cross = reinterpret_cast<unsigned long long*>(acc_sums + A.n_rec * 8);
    unsigned int* acc_distinct = reinterpret_cast<unsigned int*>(acc_cross + A.n_rec);
    unsigned int* acc_bins = acc_distinct + ((A.n_rec + 1) & ~1);
    int* ctl = reinterpret_cast<int*>(acc_bins + ((A.bins_in_lds ? A.total_bins : 0) + 1 & ~1));
    const unsigned int set = A.set_size ? blockIdx.x / (unsigned int)A.wgs_per_set : 0u;
    __threadfence_block();
    int order = 0;
    if (lane == 0) order = atomicAdd(&ctl[CTL_DONE], 1);
    order = __builtin_amdgcn_readfirstlane(order);
    if (order != kWaves - 1) return;
    __threadfence_block();
    {
        unsigned long long* ctr = A.counters;
        if (ctr && lane < 4) {
            const unsigned long long* cnt = reinterpret_cast<const unsigned long long*>(ctl + CTL_COUNT);
            const unsigned long long v = lane == 3 ? (unsigned long long)kWaves : cnt[lane];
            if (v) atomicAdd(ctr + (blockIdx.x & 63u) * 8u + lane, v);
        }
        if (lane == 0) {   // the workgroup's life on both clocks (KArgs::stamp)
            const unsigned long long* t0 = reinterpret_cast<const unsigned long long*>(ctl + CTL_CLOCK);
            const unsigned long long real = wall_clock64(), cyc = __builtin_readcyclecounter();
            if (ctr) {
                atomicAdd(ctr + (blockIdx.x & 63u) * 8u + 4u, cyc - t0[1]);
                atomicAdd(ctr + (blockIdx.x & 63u) * 8u + 5u, real - t0[0]);
            }
            if (A.stamp) atomicMax(A.stamp + 1, real);
        }
    }
    unsigned long long* const out_distinct = reinterpret_cast<unsigned long long*>(A.rec_distinct) + (long long)set * A.set_stride_i;
    unsigned long long* const out_crossings = reinterpret_cast<unsigned long long*>(A.rec_crossings) + (long long)set * A.set_stride_i;
    unsigned long long* const out_bins = reinterpret_cast<unsigned long long*>(A.rec_bins) + (long long)set * A.set_stride_i;
    double* const out_sums = A.rec_sums + (long long)set * A.set_stride_d;
    for (int i = lane; i < A.n_rec; i += 64) {
        const unsigned long long c = acc_cross[i];
        const unsigned int d = acc_distinct[i];
        if (c) atomicAdd(out_crossings + i, c);
        if (d) atomicAdd(out_distinct + i, (unsigned long long)d);
    }
    for (int i = lane; i < A.n_rec * 8; i += 64) {
        double v = acc_sums[i];
        if (v != 0.0) atomicAdd(out_sums + i, v);
    }
    if (A.bins_in_lds)
        for (int i = lane; i < A.total_bins; i += 64) {
            unsigned int v = acc_bins[i];
            if (v) atomicAdd(out_bins + i, (unsigned long long)v);
        }
}

template <bool RECORD, int TAB_LDS, int SEENW, bool EMIT, bool MESH, bool GRID = false, bool TAIL = false>
__device__ __forceinline__ void trace_body(const KArgs& A, int tail_total = 0, unsigned int tail_lds = 0u) {
    extern __shared__ double smem_of_kernel[];
    // (In a called function the address of the kernel's dynamic LDS is looked up in a table in memory wherever it is used --
    // six scalar loads and waits per step of the tail function, measured in its ISA; the kernel hands it over instead.)
    double* const smem = TAIL ? (double*)(__attribute__((address_space(3))) double*)(unsigned long long)tail_lds
                              : (double*)smem_of_kernel;
    // A wave alone waits out every scalar-cache round trip too: what the loop asks of the kernel's arguments time and again
    // is held in registers by the tail function (the kernels' own loop re-reads them where it needs them: a scalar load
    // costs it nothing, a register does).
    auto held = [](int v) __attribute__((always_inline)) -> int {
        if constexpr (TAIL) asm volatile("" : "+s"(v));
        return v;
    };
    const int k_root = held(A.root), k_maxsteps = held(A.maxsteps), k_nodes = held(A.n_nodes);
#if PVT_TIMELINE
    unsigned long long tl_t[6] = {(unsigned long long)wall_clock64(), 0, 0, 0, 0, 0};
    unsigned long long tl_iters = 0;
    const unsigned long long tl_c0 = __builtin_readcyclecounter();   // shader clock (s_memtime); wall_clock64 is the constant 100 MHz one
#endif
    Lay L = A.lay;
    if constexpr (TAIL) {   // (every field the step reads, resident)
        L.comp_d = held(L.comp_d); L.rec_d = held(L.rec_d); L.hist_d = held(L.hist_d); L.coat_d = held(L.coat_d);
        L.comp_i = held(L.comp_i); L.rec_i = held(L.rec_i); L.hist_i = held(L.hist_i); L.coat_i = held(L.coat_i);
        L.cand_i = held(L.cand_i); L.cand_list = held(L.cand_list); L.crit_d = held(L.crit_d); L.ccrit_d = held(L.ccrit_d);
        L.grid_d = held(L.grid_d); L.rot_d = held(L.rot_d); L.ncls_d = held(L.ncls_d); L.n_cls = held(L.n_cls);
    }
    // Launch constants that the loop only asks yes/no questions of, in ONE scalar register.  Kept as separate
    // conditions each becomes a 64-bit lane mask that the allocator holds (spills) for the whole loop; `uf(bit)`
    // re-derives the answer from the word where it is asked (the empty asm keeps the compiler from hoisting it).
    enum { UF_COATED = 0, UF_FUSE_EXIT, UF_CRIT, UF_HAS_REC, UF_TQ_POS, UF_BINS_LDS, UF_EMIT_FULL, UF_EMIT_KT, UF_LAZY1, UF_LAZY2, UF_BY_NODE,
           UF_TAIL_LAZY1, UF_TAIL_LAZY2 };
    unsigned int uflags_ =
        (A.n_coat > 0 ? 1u << UF_COATED : 0u) | (A.fuse_exit != 0 ? 1u << UF_FUSE_EXIT : 0u) | (L.crit_d >= 0 ? 1u << UF_CRIT : 0u) |
        (A.n_rec > 0 ? 1u << UF_HAS_REC : 0u) | (A.tq_pos ? 1u << UF_TQ_POS : 0u) | (A.bins_in_lds ? 1u << UF_BINS_LDS : 0u) |
        (A.emit_method == PVT_EMIT_FULL ? 1u << UF_EMIT_FULL : 0u) | (A.emit_method == PVT_EMIT_KT ? 1u << UF_EMIT_KT : 0u) |
        (A.lazy_root == 1 ? 1u << UF_LAZY1 : 0u) | (A.lazy_root == 2 ? 1u << UF_LAZY2 : 0u) | (L.by_node ? 1u << UF_BY_NODE : 0u);
    if constexpr (TAIL && PVT_TAIL_LAZY) {   // (only where the launch itself has no lazy root: see KArgs::lazy_tail)
        if (A.lazy_root == 0) uflags_ |= (A.lazy_tail == 1 ? 1u << UF_TAIL_LAZY1 : 0u) | (A.lazy_tail == 2 ? 1u << UF_TAIL_LAZY2 : 0u);
    }
    const unsigned int uflags = uflags_;
    auto uf = [&](int bit) -> bool {
        unsigned int f = uflags;
        asm volatile("" : "+s"(f));
        return ((f >> bit) & 1u) != 0u;
    };

    // ---- stage tables + zero accumulators --------------------------------
    double* lds_d = smem;
    const int nd_lds = TAB_LDS == 1 ? A.nd : TAB_LDS == 2 ? A.nd_lds : 0;
    const int ni_stage = TAB_LDS == 1 ? A.ni : TAB_LDS == 2 ? A.ni_lds : 0;
    const int ni_lds = (ni_stage + 1) & ~1;  // keep 8-byte alignment after the ints
    int* lds_i = reinterpret_cast<int*>(lds_d + nd_lds);
    double* acc_sums = reinterpret_cast<double*>(lds_i + ni_lds);
    // crossings are 64-bit: a trapped photon crosses surfaces up to `maxsteps` times, and a persistent
    // workgroup of a 2^31-photon launch sees millions of photons (distinct rays and bins are bounded by them)
    unsigned long long* acc_cross = reinterpret_cast<unsigned long long*>(acc_sums + A.n_rec * 8);
    unsigned int* acc_distinct = reinterpret_cast<unsigned int*>(acc_cross + A.n_rec);
    unsigned int* acc_bins = acc_distinct + ((A.n_rec + 1) & ~1);   // (keeps what follows 8-byte aligned)
    int* ctl = reinterpret_cast<int*>(acc_bins + ((A.bins_in_lds ? A.total_bins : 0) + 1 & ~1));
    unsigned long long* xbuf = reinterpret_cast<unsigned long long*>(ctl + CTL_WORDS);  // [14 + SEENW (+1 when RECORD)][xslots] u64 words
    // per-wave queue of first crossings awaiting their statistics: [4 (+3 with positions)][kTallyQ] doubles
    // + [kTallyQ] recorder ids
    constexpr int kXWords = 14 + SEENW + (RECORD ? 1 : 0);
    const int tq_doubles = A.tq_pos ? 7 : 4;
    double* const tq_d = reinterpret_cast<double*>(xbuf + kXWords * A.xslots) + (threadIdx.x >> 6) * (tq_doubles * kTallyQ);
    int* const tq_r = reinterpret_cast<int*>(reinterpret_cast<double*>(xbuf + kXWords * A.xslots) + kWaves * tq_doubles * kTallyQ)
                      + (threadIdx.x >> 6) * kTallyQ;
    int tq_n = 0;   // wave-uniform
    if constexpr (!TAIL) {   // (the tail function finds the workgroup's LDS as its caller left it)
    if (threadIdx.x < CTL_WORDS) ctl[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 4 && A.cursor_next) A.cursor_next[threadIdx.x] = 0u;
    if constexpr (MESH) {   // the top of the triangle trees (two 16-byte words per record)
        typedef unsigned int Word4 __attribute__((ext_vector_type(4)));
        Word4* dst = reinterpret_cast<Word4*>(reinterpret_cast<char*>(smem) + A.top_off);
        const Word4* src = reinterpret_cast<const Word4*>(A.bvh_top);
        for (int i = threadIdx.x; i < A.top_n * 2; i += kBlock) dst[i] = src[i];
    }
    if constexpr (TAB_LDS != 0) {
        for (int i = threadIdx.x; i < nd_lds; i += kBlock) lds_d[i] = A.gd[i];
        for (int i = threadIdx.x; i < ni_stage; i += kBlock) lds_i[i] = A.gi[i];
    }
    for (int i = threadIdx.x; i < A.n_rec * 8; i += kBlock) acc_sums[i] = 0.0;
    for (int i = threadIdx.x; i < A.n_rec; i += kBlock) { acc_cross[i] = 0ull; acc_distinct[i] = 0u; }
    if (A.bins_in_lds)
        for (int i = threadIdx.x; i < A.total_bins; i += kBlock) acc_bins[i] = 0u;
    __syncthreads();
    if (threadIdx.x == 0) {   // (KArgs::stamp; read again by the last wave to leave, in leave_workgroup)
        unsigned long long* t0 = reinterpret_cast<unsigned long long*>(ctl + CTL_CLOCK);
        t0[0] = wall_clock64(); t0[1] = __builtin_readcyclecounter();
        if (blockIdx.x == 0 && A.stamp) A.stamp[0] = t0[0];
    }
    }
#if PVT_TIMELINE
    tl_t[1] = wall_clock64();
#endif

    Tables<TAB_LDS> T{(CDoubles)A.gd, (CInts)A.gi, TAB_LDS != 0 ? lds_d : A.gd, TAB_LDS != 0 ? lds_i : A.gi, A.gd, A.gi};

    // a node's flag word, read by the lane (see ND_BITS)
    auto node_bits = [&](int node) -> unsigned long long { return pvt_d2u(T.dv(node * ND + ND_BITS)); };
    auto node_ident = [](unsigned long long bits) -> bool { return (bits & 1ull) != 0; };
    auto node_geom = [](unsigned long long bits) -> int { return (int)(((unsigned int)bits >> 8) & 0xffu); };
    auto node_rot = [&](unsigned long long bits) -> int { return L.rot_d + (int)(unsigned int)(bits >> 32) * RT; };

    // Forward crossings (t > kEps) of one analytic shape by the ray (o, d) in the shape's own frame, each handed to
    // `fold` in the reference's order (_kernel.pyx:245-345).  `inv`: 1/d per axis (boxes only).  The shape parameters
    // are scalars in the wave-uniform node loop and per-lane values in the grid walk: same operations either way.
    auto shape_hits = [](int gt, double g0, double g1, double g2, const V3& o, const V3& d, double inv0, double inv1, double inv2,
                         auto&& fold) __attribute__((always_inline)) {
        if (gt == PVT_GEOM_BOX) {  // slab test (:245-276)
            double tmin = -INFINITY, tmax = INFINITY;
            bool miss = false;
            const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, gpar[3] = {g0, g1, g2}, inv[3] = {inv0, inv1, inv2};
            // a ray parallel to a pair of faces (a direction component below 1e-300) takes the reference's
            // inside/outside test for that axis; the wave only runs the general form when a lane holds one
            if (__ballot(pvt_fabs(dd[0]) < 1e-300 || pvt_fabs(dd[1]) < 1e-300 || pvt_fabs(dd[2]) < 1e-300) == 0ull) {
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    const double sz = gpar[a];
                    const double ta = (-0.5 * sz - oo[a]) * inv[a], tb = (0.5 * sz - oo[a]) * inv[a];
                    tmin = __builtin_fmax(tmin, __builtin_fmin(ta, tb));
                    tmax = __builtin_fmin(tmax, __builtin_fmax(ta, tb));
                }
            } else
#pragma unroll
            for (int a = 0; a < 3; a++) {
                double sz = gpar[a];
                double lo = -0.5 * sz, hi = 0.5 * sz;
                if (pvt_fabs(dd[a]) < 1e-300) {
                    if (oo[a] < lo || oo[a] > hi) miss = true;
                } else {
                    // (the reference swaps ta, tb into order and keeps the largest entry / smallest exit
                    // distance: min and max of finite numbers, which is what these are)
                    const double ta = (lo - oo[a]) * inv[a], tb = (hi - oo[a]) * inv[a];
                    tmin = __builtin_fmax(tmin, __builtin_fmin(ta, tb));
                    tmax = __builtin_fmin(tmax, __builtin_fmax(ta, tb));
                }
            }
            if (!miss && !(tmax < tmin)) {
                if (tmin > kEps) fold(tmin);
                if (tmax > kEps) fold(tmax);
            }
        } else if (gt == PVT_GEOM_SPHERE) {  // (:279-298)
            double radius = g0;
            double a = dot3(d, d), b = 2.0 * dot3(d, o), c = dot3(o, o) - radius * radius;
            double disc = b * b - 4.0 * a * c;
            if (!(disc < 0.0)) {
                double sq = pvt_sqrt(disc);
                double t = (-b - sq) / (2.0 * a);
                if (t > kEps) fold(t);
                t = (-b + sq) / (2.0 * a);
                if (t > kEps) fold(t);
            }
        } else {  // capped z cylinder (:301-345)
            double half = 0.5 * g0, radius = g1;
            double a = d.x * d.x + d.y * d.y;
            if (a > 1e-300) {
                double b = 2.0 * (o.x * d.x + o.y * d.y);
                double c = o.x * o.x + o.y * o.y - radius * radius;
                double disc = b * b - 4.0 * a * c;
                if (disc >= 0.0) {
                    double sq = pvt_sqrt(disc);
                    double t = (-b - sq) / (2.0 * a);
                    double z = o.z + t * d.z;
                    if (z > -half && z < half && t > kEps) fold(t);
                    t = (-b + sq) / (2.0 * a);
                    z = o.z + t * d.z;
                    if (z > -half && z < half && t > kEps) fold(t);
                }
            }
            if (pvt_fabs(d.z) > 1e-300) {
                double t = (-half - o.z) / d.z;
                double x = o.x + t * d.x, y = o.y + t * d.y;
                if (x * x + y * y <= radius * radius && t > kEps) fold(t);
                t = (half - o.z) / d.z;
                x = o.x + t * d.x;
                y = o.y + t * d.y;
                if (x * x + y * y <= radius * radius && t > kEps) fold(t);
            }
        }
    };

    const int lane = threadIdx.x & 63;
    // rank of this lane among the set bits of a wave mask: the bits below it, counted by the mbcnt pair
    auto rank_in = [](unsigned long long mask) -> unsigned int {
        return __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
    };

    // ---- the rays this workgroup draws from: the whole launch, or its tally set (wave-uniform) ------
    const unsigned int set = A.set_size ? blockIdx.x / (unsigned int)A.wgs_per_set : 0u;
    const unsigned int ray_lo = set * A.set_size;
    const unsigned int n_local = A.set_size ? (A.n_rays - ray_lo < A.set_size ? A.n_rays - ray_lo : A.set_size) : A.n_rays;
    unsigned int* const cursor = A.cursor + set;

    // ---- per-photon state --------------------------------------------------
    bool alive = false;
    V3 pos{0, 0, 0}, dir{0, 0, 1};
    double wl = 0.0, travelled = 0.0, duration = 0.0;
    Rng rng{0, 0, 0, 0};
    int count = 0, source = -1, nev = 0;
    unsigned int c_iters = 0u, c_steps = 0u, c_fused = 0u;   // this lane's share of the step counters (KArgs::counters)
    // (tail function) the absorption coefficients this lane's photon met last: container, wavelength bits, sum, first term
    // -- kept in LDS, in the exchange buffer the function was handed its photons through (free once they are read): four
    // words per lane, read together (one wait) where two table lookups stood; in registers they were seven more than the
    // function has (the hit node's record, below, went to scratch for them: PVT_TAIL_ALPHA 2 keeps that variant)
    int ac_node = -1;
    unsigned long long ac_wl = 0ull;
    double ac_alpha = 0.0, ac_pre0 = 0.0;
    unsigned long long* const ac_lds = xbuf + lane;   // [4][64] words: container, wavelength bits, sum, first term
    // (tail function) the record of the node this step's nearest crossing lies on, and the Fresnel constants of the pair
    // (container, adjacent), read in one go right after the node loop
    V3 hr_t{0, 0, 0}, hr_g{0, 0, 0};
    unsigned long long hr_bits = 0ull;
    int hr_surf = 0;
    double hr_n2 = 0.0, hr_rn2 = 0.0, hr_cc = 0.0;
    constexpr bool kHoist = TAIL && PVT_TAIL_HOIST;
    int rec_slot = -1;   // recorded rays: index among them (row block rec_slot * max_events), else -1
    Seen<SEENW> seen;
#pragma unroll
    for (int w = 0; w < SEENW; w++) seen.w[w] = 0ull;

#if PVT_STATS
    unsigned long long st_iters = 0, st_lane_steps = 0, st_drain_iters = 0, st_drain_lane_steps = 0;
    unsigned long long st_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_mark = 0;
#define PVT_MARK(k) do { unsigned long long now_ = __builtin_readcyclecounter(); if (ws & WS_SOLO) st_t[k] += now_ - st_mark; st_mark = now_; } while (0)
    unsigned long long st_c[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // bulk-phase lane counts per section
    unsigned long long st_g[4] = {0, 0, 0, 0};   // grid walk: cell rounds, node-test trips (per wave), node tests, cell steps (per lane)
#define PVT_COUNT(k, pred) do { unsigned long long b_ = __ballot(pred); if (!(ws & WS_EXHAUSTED)) st_c[k] += __popcll(b_); } while (0)
#else
#define PVT_MARK(k) do {} while (0)
#define PVT_COUNT(k, pred) do {} while (0)
#endif
    // wave-uniform ray window claimed from the global cursor
    // (the waves of a launch -- of a tally set -- are numbered; wave k owns rays [k kClaim, (k+1) kClaim) and parked
    // photons [64 k, 64 k + 64) without asking, everything beyond is claimed from the cursors)
    const unsigned int wgs_in_set = A.set_size ? (unsigned int)A.wgs_per_set : gridDim.x;
    const unsigned int waves_in_set = wgs_in_set * kWaves;
    const unsigned int wave_in_set = (blockIdx.x - set * (A.set_size ? (unsigned int)A.wgs_per_set : 0u)) * kWaves +
                                     (unsigned int)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned int w_first = wave_in_set * (unsigned int)kClaim;
    unsigned int w_next = w_first, w_end = w_first, w_base = 0;
    unsigned int w_claim_end = w_first < n_local ? (n_local - w_first < (unsigned int)kClaim ? n_local : w_first + kClaim) : w_first;
    unsigned int c_next = 0, c_end = 0;   // window of parked photons
    // The wave's own yes/no state, one bit each in ONE scalar register (as separate flags each is a 64-bit lane mask
    // held -- or spilled -- across the whole loop): the ray cursor is dry; that has been counted in the workgroup;
    // in the drain rendezvous; last wave standing; parked photons may still be waiting; first claim of them
    enum { WS_EXHAUSTED = 1, WS_COUNTED = 2, WS_REGIME = 4, WS_SOLO = 8, WS_CARRY_IN = 16, WS_C_FIRST = 32 };
    unsigned int ws = WS_C_FIRST | ((!RECORD && (A.carry_flags & 1) != 0) ? WS_CARRY_IN : 0u);
    // Per-wave pool of ready-made RNG states for the claimed chunk (4 x 64 u64 = 2 KB).  It lives
    // in the photon-exchange buffer: that buffer is first used when ALL waves of the workgroup
    // have run the cursor dry, i.e. after every pool has been consumed.
    const bool seed_pool = A.xslots * (14 + SEENW) >= kWaves * 4 * 64;
    const int pool_at = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * 4 * 64;   // wave-uniform
    // drain-phase consolidation state (all wave-uniform)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    int members = 0, parity = 0;
    int tail_n = 0;   // photons this wave hands to the tail function when it leaves the loop (0: none)
    // (mesh scenes never repack a draining workgroup -- KArgs::xslots is 0 there -- and the history variant of the grid walk
    // has no register to spare for the call: both keep their last wave in this loop)
    constexpr bool kTailCall = PVT_TAIL_CALL && !TAIL && !MESH && !(GRID && RECORD);

    // Statistics of the parked first crossings, one lane per record: the recorder's distinct count, the
    // eight running sums (the angle is acos of the parked cosine; 1.0 -> exactly 0 for events without a
    // normal) and its histograms.  Runs when the queue cannot take another trip, i.e. with ~60 busy lanes
    // instead of the handful that cross a recorder in any one step, and once more when the wave retires.
    auto tally_flush = [&]() {
        if (lane < tq_n) {
            const int r = tq_r[lane];
            const double q_wl = tq_d[lane], q_angle = pvt_acos(tq_d[kTallyQ + lane]);
            const double q_duration = tq_d[2 * kTallyQ + lane], q_travelled = tq_d[3 * kTallyQ + lane];
            const int ri = L.rec_i + r * RI;
            atomicAdd(&acc_distinct[r], 1u);
            double* sp = acc_sums + r * 8;
            atomicAdd(&sp[0], q_wl); atomicAdd(&sp[1], q_wl * q_wl);
            atomicAdd(&sp[2], q_angle); atomicAdd(&sp[3], q_angle * q_angle);
            atomicAdd(&sp[4], q_duration); atomicAdd(&sp[5], q_duration * q_duration);
            atomicAdd(&sp[6], q_travelled); atomicAdd(&sp[7], q_travelled * q_travelled);
            const int h0 = T.iv(ri + RI_HSTART), h1 = h0 + T.iv(ri + RI_HN);
            for (int h = h0; h < h1; h++) {
                const int hi_ = L.hist_i + h * HI, hd_ = L.hist_d + h * HD;
                const int pa = T.iv(hi_ + HI_PA), pb = T.iv(hi_ + HI_PB);
                const int na = T.iv(hi_ + HI_NA), nb = T.iv(hi_ + HI_NB);
                auto prop = [&](int pr) -> double {
                    if (pr < 4) return pr == 0 ? q_wl : pr == 1 ? q_angle : pr == 2 ? q_duration : q_travelled;
                    return tq_d[pr * kTallyQ + lane];   // x, y, z in the recorder node's frame (A.tq_pos)
                };
                const double la = T.dv(hd_ + HD_LO_A), ha = T.dv(hd_ + HD_HI_A);
                const double ra = T.dv(hd_ + HD_RA);
                const double qa = ra == ra ? div_known(prop(pa) - la, ha - la, ra) : (prop(pa) - la) / (ha - la);
                const int ia = (int)(qa * na);
                if (ia < 0 || ia >= na) continue;
                int slot = T.iv(hi_ + HI_OFF) + ia;
                if (pb >= 0) {
                    const double lb = T.dv(hd_ + HD_LO_B), hb = T.dv(hd_ + HD_HI_B);
                    const double rb = T.dv(hd_ + HD_RB);
                    const double qb = rb == rb ? div_known(prop(pb) - lb, hb - lb, rb) : (prop(pb) - lb) / (hb - lb);
                    const int ib = (int)(qb * nb);
                    if (ib < 0 || ib >= nb) continue;
                    slot = T.iv(hi_ + HI_OFF) + ia * nb + ib;
                }
                if (uf(UF_BINS_LDS)) atomicAdd(&acc_bins[slot], 1u);
                else atomicAdd(reinterpret_cast<unsigned long long*>(A.rec_bins) + (long long)set * A.set_stride_i + slot, 1ull);
            }
        }
        tq_n = 0;
    };
    if constexpr (TAIL) {   // the photons the caller left in the exchange buffer, one per lane
        constexpr int X = kXSlots;
        const int slot = lane;
        ws = WS_EXHAUSTED | WS_COUNTED | WS_SOLO;
        alive = slot < tail_total;
        if (alive) {
            pos = V3{pvt_u2d(xbuf[0 * X + slot]), pvt_u2d(xbuf[1 * X + slot]), pvt_u2d(xbuf[2 * X + slot])};
            dir = V3{pvt_u2d(xbuf[3 * X + slot]), pvt_u2d(xbuf[4 * X + slot]), pvt_u2d(xbuf[5 * X + slot])};
            wl = pvt_u2d(xbuf[6 * X + slot]); travelled = pvt_u2d(xbuf[7 * X + slot]); duration = pvt_u2d(xbuf[8 * X + slot]);
            rng.s0 = xbuf[9 * X + slot]; rng.s1 = xbuf[10 * X + slot]; rng.s2 = xbuf[11 * X + slot]; rng.s3 = xbuf[12 * X + slot];
            const unsigned long long cs_ = xbuf[13 * X + slot];
            count = (int)(unsigned int)cs_;
            source = (int)(unsigned int)(cs_ >> 32);
#pragma unroll
            for (int w = 0; w < SEENW; w++) seen.w[w] = xbuf[(14 + w) * X + slot];
            if constexpr (RECORD) {
                const unsigned long long rn_ = xbuf[(14 + SEENW) * X + slot];
                rec_slot = (int)(unsigned int)rn_;
                nev = (int)(unsigned int)(rn_ >> 32);
            }
        }
    }
    if constexpr (TAIL && PVT_TAIL_ALPHA == 1) ac_lds[0] = ~0ull;   // (nothing known yet; after the reads above, same wave: in order)
    for (;;) {
        if constexpr (TAIL) {
            if (__ballot(alive) == 0ull) break;
        }
        if constexpr (!TAIL) {
        // ================= refill dead lanes ==============================
        unsigned long long need = __ballot(!alive);
        if constexpr (!RECORD) {
            // photons parked by the previous launch on this stream come first (see KArgs::carry_in).  They are handed
            // out like rays, 64 at a time: every wave starts with the slice its index names (no atomic -- the previous
            // launch parked at most 64 per wave, so with an equal or larger grid that is all of them); only what lies
            // beyond is claimed through the claim cursor, one atomic per 64.  (A claim per refill, for just the lanes
            // that died in the last step, put a queue of ~10^5 same-address atomics in front of every step: measured
            // 21 us per iteration instead of 11.)
            if ((ws & WS_CARRY_IN) && need != 0ull) {
                const __attribute__((address_space(4))) KArgs* ak =
                    (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
                asm volatile("" : "+s"(ak));
                const unsigned int have = *ak->carry_in_count;
                if (c_next >= c_end) {
                    unsigned int b = ~0u;
                    if (ws & WS_C_FIRST) {
                        ws &= ~(unsigned int)WS_C_FIRST;
                        b = wave_in_set * 64u;
                    } else if (have > waves_in_set * 64u) {
                        if (lane == 0) b = atomicAdd(cursor + 1, 64u);
                        b = __builtin_amdgcn_readfirstlane(b) + waves_in_set * 64u;
                    }
                    if (b >= have) {
                        ws &= ~(unsigned int)WS_CARRY_IN;
                    } else {
                        c_next = b;
                        c_end = have - b < 64u ? have : b + 64u;
                    }
                }
                if (ws & WS_CARRY_IN) {
                    const unsigned int want = __popcll(need), got = c_end - c_next < want ? c_end - c_next : want;
                    const unsigned int rank = rank_in(need);
                    if (!alive && rank < got) {
                        // photon-major records: one base address, every word at an immediate offset
                        const unsigned long long* src = ak->carry_in + (unsigned long long)(c_next + rank) * kCarryStride;
                        pos = V3{pvt_u2d(src[0]), pvt_u2d(src[1]), pvt_u2d(src[2])};
                        dir = V3{pvt_u2d(src[3]), pvt_u2d(src[4]), pvt_u2d(src[5])};
                        wl = pvt_u2d(src[6]); travelled = pvt_u2d(src[7]); duration = pvt_u2d(src[8]);
                        rng.s0 = src[9]; rng.s1 = src[10]; rng.s2 = src[11]; rng.s3 = src[12];
                        const unsigned long long cs_ = src[13];
                        count = (int)(unsigned int)cs_;
                        source = (int)(unsigned int)(cs_ >> 32);
#pragma unroll
                        for (int w = 0; w < SEENW; w++) seen.w[w] = src[kCarryBase + w];
                        nev = 0;
                        alive = true;
                    }
                    c_next += got;
                    need = __ballot(!alive);
                }
            }
        }
        for (int pass = 0; pass < 2 && need != 0ull; pass++) {
            if (w_next >= w_end) {
                if (ws & WS_EXHAUSTED) break;
                unsigned int b = w_end;          // next chunk of the rays this wave has claimed ...
                if (b >= w_claim_end) {          // ... or a new claim (the first one is the wave's own index: no atomic;
                                                 // 4096 waves starting together on ONE word cost the launch ~50 us)
                    // (Smaller claims towards the end of the rays -- a wave that takes the last 64 rays runs some six
                    // iterations longer than its neighbours, which found the cursor dry -- were measured and NOT kept:
                    // with claims of 32 / 16 rays in the last two rounds the pipelined bench lost 4 % / 13 %, the extra
                    // same-address atomics cost more than the ragged end.  Between two claims of a wave every other
                    // wave claims about once, so the cursor stands near w_claim_end + waves x 64 now.)
                    const unsigned int claim = (n_local - w_claim_end > 2u * waves_in_set * (unsigned int)kClaim || w_claim_end > n_local)
                                                   ? (unsigned int)kClaim : (unsigned int)kEndClaim;
                    unsigned int c = 0;
                    if (lane == 0) c = atomicAdd(cursor, claim);
                    c = __builtin_amdgcn_readfirstlane(c) + waves_in_set * (unsigned int)kClaim;
                    if (c >= n_local) { ws |= WS_EXHAUSTED; break; }
                    b = c;
                    w_claim_end = (n_local - c < claim) ? n_local : c + claim;
                }
                w_next = b;
                w_end = (w_claim_end - b < (unsigned int)kChunk) ? w_claim_end : b + kChunk;
                w_base = b;
                if constexpr (EMIT) {
                    // sample the whole chunk now (see KArgs::emit_pool); the stores are read back by lanes of this very
                    // wave, later in program order
                    const __attribute__((address_space(4))) KArgs* ak =
                        (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(ak));
                    unsigned int lane_here = (unsigned int)lane;
                    asm volatile("" : "+v"(lane_here));
                    if (b + lane_here < w_end) {
                        double* pool = ak->emit_pool + ((unsigned long long)blockIdx.x * kWaves + (unsigned long long)wave) * (7 * 64) + lane_here;
                        const unsigned long long gi_ = A.ray_offset + (unsigned long long)ray_lo + (unsigned long long)b + (unsigned long long)lane_here;
                        emit_chunk((const KArgs*)ak, gi_, pool);
                    }
                }
                if (seed_pool) {
                    // Seed the whole chunk NOW, with every lane busy, instead of inside each later
                    // refill with only the dead lanes active (8 64-bit multiplies per seed): lane l
                    // prepares the stream of ray b + l and parks it in this wave's slice of LDS.
                    // (the lane number behind an opaque copy: hoisted out of the loop, seed + ray_lo + lane would sit in two
                    // vector registers for the whole kernel, which has none to spare)
                    unsigned int lane_here = (unsigned int)lane;
                    asm volatile("" : "+v"(lane_here));
                    unsigned long long st = A.seed + (unsigned long long)ray_lo + (unsigned long long)b + (unsigned long long)lane_here;
#pragma unroll
                    for (int k = 0; k < 4; k++) {   // one word at a time: keeps the live range short
                        xbuf[pool_at + k * 64 + lane] = splitmix64(st);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            unsigned int avail = w_end - w_next;
            unsigned int rank = rank_in(need);
            unsigned int want = __popcll(need);
            if (!alive && rank < avail) {
                const unsigned int il = w_next + rank;   // index within the set
                const unsigned int i = ray_lo + il;      // index within the launch
                if constexpr (EMIT) {
                    const __attribute__((address_space(4))) KArgs* ak =
                        (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(ak));
                    const double* pool = ak->emit_pool + ((unsigned long long)blockIdx.x * kWaves + (unsigned long long)wave) * (7 * 64) + (il - w_base);
                    pos = V3{pool[0], pool[64], pool[128]};
                    dir = V3{pool[192], pool[256], pool[320]};
                    wl = pool[384];
                } else {
                    // The three ray pointers are read from the kernel-argument segment HERE, through a pointer
                    // the compiler cannot trace back to it: kept as loop invariants they would sit in six
                    // scalar registers the loop is short of (spilled to VGPR lanes, each use a VALU
                    // v_readlane); scalar loads cost the vector pipe nothing.
                    const __attribute__((address_space(4))) KArgs* ak =
                        (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(ak));
                    const double* rp = ak->pos;
                    const double* rd = ak->dir;
                    const double* rw = ak->wl;
                    pos = V3{rp[i * 3ull], rp[i * 3ull + 1], rp[i * 3ull + 2]};
                    dir = V3{rd[i * 3ull], rd[i * 3ull + 1], rd[i * 3ull + 2]};
                    wl = rw[i];
                }
                if (seed_pool) {
                    const int k = pool_at + (int)(il - w_base);
                    rng.s0 = xbuf[k]; rng.s1 = xbuf[k + 64]; rng.s2 = xbuf[k + 128]; rng.s3 = xbuf[k + 192];
                } else {
                    rng_seed(rng, A.seed + (unsigned long long)i);
                }
                travelled = 0.0;
                duration = 0.0;
                count = 0;
                source = -1;
                nev = 0;
#pragma unroll
                for (int w = 0; w < SEENW; w++) seen.w[w] = 0ull;
                alive = true;
                if constexpr (RECORD) {
                    rec_slot = -1;
                    if (A.record_every > 0 && (long long)i % A.record_every == 0) rec_slot = (int)((long long)i / A.record_every);
                    log_row<RECORD, TAIL>(A, rec_slot, nev, PVT_EV_GENERATE, -1, -1, -1, -1, source, pos, dir, false,
                                    pos, wl, travelled, duration);
                }
            }
            w_next += (want < avail) ? want : avail;
            need = __ballot(!alive);
        }
        // ================= drain: repack the workgroup's survivors ==========
        // Once the ray cursor is dry a wave can only lose lanes; with 4 waves per workgroup
        // each at a handful of live lanes, most issue slots would be spent on idle lanes
        // (measured: 55 % of all wave-iterations of a 10^6-photon launch ran at 11 live lanes).
        // So when ALL waves of the workgroup are draining they rendezvous each iteration
        // (s_barrier), publish their live counts, and as soon as the survivors fit into fewer
        // waves every live photon's state is moved through LDS into the lowest waves and the
        // emptied waves retire.  Which lane carries a photon never affects its history (RNG
        // stream, seen-mask and log rows travel with it), so results stay bit-identical.
        if constexpr (!RECORD) {
            if ((ws & WS_EXHAUSTED) && (A.carry_flags & 2)) {
                // no rays left for this wave: its live photons are parked for the next launch on the stream
                const unsigned long long live_mask = __ballot(alive);
                if (live_mask != 0ull) {
                    const __attribute__((address_space(4))) KArgs* ak =
                        (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
                    asm volatile("" : "+s"(ak));
                    unsigned int b = 0;
                    if (lane == 0) b = atomicAdd(cursor + 2, (unsigned int)__popcll(live_mask));
                    b = __builtin_amdgcn_readfirstlane(b);
                    const unsigned int at = b + rank_in(live_mask);
                    if (alive && at < ak->carry_cap) {
                        unsigned long long* dst = ak->carry_out + (unsigned long long)at * kCarryStride;
                        dst[0] = pvt_d2u(pos.x); dst[1] = pvt_d2u(pos.y); dst[2] = pvt_d2u(pos.z);
                        dst[3] = pvt_d2u(dir.x); dst[4] = pvt_d2u(dir.y); dst[5] = pvt_d2u(dir.z);
                        dst[6] = pvt_d2u(wl); dst[7] = pvt_d2u(travelled); dst[8] = pvt_d2u(duration);
                        dst[9] = rng.s0; dst[10] = rng.s1; dst[11] = rng.s2; dst[12] = rng.s3;
                        dst[13] = (unsigned long long)(unsigned int)count | ((unsigned long long)(unsigned int)source << 32);
#pragma unroll
                        for (int w = 0; w < SEENW; w++) dst[kCarryBase + w] = seen.w[w];
                    }
                    alive = false;   // (the wave leaves through the "nothing alive, no rays left" exit below)
                }
            }
        }
        if ((ws & (WS_EXHAUSTED | WS_COUNTED)) == WS_EXHAUSTED) {
            ws |= WS_COUNTED;
            if (lane == 0) atomicAdd(&ctl[CTL_EXHAUSTED], 1);
        }
        if (!(ws & WS_REGIME)) {
            if (__ballot(alive) == 0ull) break;  // wave drained and cursor exhausted
            // (values read back from LDS are made scalars explicitly -- readfirstlane -- so that the state they
            // decide stays in scalar registers)
            if (A.xslots > 0 && (ws & (WS_EXHAUSTED | WS_SOLO)) == WS_EXHAUSTED &&
                __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctl[CTL_EXHAUSTED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == kWaves) {
                ws |= WS_REGIME;
                if (lane == 0) ctl[CTL_IN + wave] = 1;
                __syncthreads();  // R0: everybody still running is now in lock-step
                members = __builtin_amdgcn_readfirstlane((ctl[CTL_IN] ? 1 : 0) | (ctl[CTL_IN + 1] ? 2 : 0) | (ctl[CTL_IN + 2] ? 4 : 0) |
                                                         (ctl[CTL_IN + 3] ? 8 : 0));
            }
        }
        if (ws & WS_REGIME) {
            const unsigned long long live_mask = __ballot(alive);
            const int live = __popcll(live_mask);
            int* live_tab = ctl + CTL_LIVE + parity * kWaves;  // double-buffered by iteration parity
            parity ^= 1;
            if (lane == 0) live_tab[wave] = live;
            __syncthreads();  // A
            int l[kWaves], total = 0, nw = 0, before = 0, pos_in_set = 0;
#pragma unroll
            for (int w = 0; w < kWaves; w++) {
                l[w] = ((members >> w) & 1) ? __builtin_amdgcn_readfirstlane(live_tab[w]) : 0;
                if (l[w] == 0) members &= ~(1 << w);  // that wave retires now (it sees its own 0)
                else {
                    if (w < wave) { before += l[w]; pos_in_set += 1; }
                    total += l[w];
                    nw += 1;
                }
            }
            if (live == 0) break;
            if (nw == 1) {
                ws = (ws & ~(unsigned int)WS_REGIME) | WS_SOLO;  // last wave standing: no more rendezvous
                // ... and no more rays: the rest of its photons' histories runs in the tail function (the loop without
                // the refill and the rendezvous, a function of its own: it cannot move this loop's registers); the
                // photons go through the exchange buffer, as in a consolidation
                if constexpr (kTailCall) {
                    constexpr int X = kXSlots;
                    if (alive) {
                        const int slot = (int)rank_in(live_mask);
                        xbuf[0 * X + slot] = pvt_d2u(pos.x); xbuf[1 * X + slot] = pvt_d2u(pos.y); xbuf[2 * X + slot] = pvt_d2u(pos.z);
                        xbuf[3 * X + slot] = pvt_d2u(dir.x); xbuf[4 * X + slot] = pvt_d2u(dir.y); xbuf[5 * X + slot] = pvt_d2u(dir.z);
                        xbuf[6 * X + slot] = pvt_d2u(wl); xbuf[7 * X + slot] = pvt_d2u(travelled); xbuf[8 * X + slot] = pvt_d2u(duration);
                        xbuf[9 * X + slot] = rng.s0; xbuf[10 * X + slot] = rng.s1; xbuf[11 * X + slot] = rng.s2; xbuf[12 * X + slot] = rng.s3;
                        xbuf[13 * X + slot] = (unsigned long long)(unsigned int)count | ((unsigned long long)(unsigned int)source << 32);
#pragma unroll
                        for (int w = 0; w < SEENW; w++) xbuf[(14 + w) * X + slot] = seen.w[w];
                        if constexpr (RECORD) {
                            xbuf[(14 + SEENW) * X + slot] = (unsigned long long)(unsigned int)rec_slot | ((unsigned long long)(unsigned int)nev << 32);
                        }
                    }
                    tail_n = live;   // (the call itself comes after the loop: nothing of the loop is live across it)
                    alive = false;
                    break;
                }
            } else if (total <= 64 * (nw - 1) && total <= A.xslots) {
                constexpr int X = kXSlots;   // (A.xslots is 0 or kXSlots: constant offsets in the LDS instructions)
                if (alive) {
                    const int slot = before + (int)rank_in(live_mask);
                    xbuf[0 * X + slot] = pvt_d2u(pos.x); xbuf[1 * X + slot] = pvt_d2u(pos.y); xbuf[2 * X + slot] = pvt_d2u(pos.z);
                    xbuf[3 * X + slot] = pvt_d2u(dir.x); xbuf[4 * X + slot] = pvt_d2u(dir.y); xbuf[5 * X + slot] = pvt_d2u(dir.z);
                    xbuf[6 * X + slot] = pvt_d2u(wl); xbuf[7 * X + slot] = pvt_d2u(travelled); xbuf[8 * X + slot] = pvt_d2u(duration);
                    xbuf[9 * X + slot] = rng.s0; xbuf[10 * X + slot] = rng.s1; xbuf[11 * X + slot] = rng.s2; xbuf[12 * X + slot] = rng.s3;
                    xbuf[13 * X + slot] = (unsigned long long)(unsigned int)count | ((unsigned long long)(unsigned int)source << 32);
#pragma unroll
                    for (int w = 0; w < SEENW; w++) xbuf[(14 + w) * X + slot] = seen.w[w];
                    if constexpr (RECORD) {
                        xbuf[(14 + SEENW) * X + slot] = (unsigned long long)(unsigned int)rec_slot | ((unsigned long long)(unsigned int)nev << 32);
                    }
                }
                __syncthreads();  // B
                const int keep = (total + 63) >> 6;  // waves that stay, lowest ids of the set
                if (pos_in_set >= keep) { alive = false; break; }
                if constexpr (kTailCall) {
                    if (keep == 1) {   // the survivors fit one wave: this one, in the tail function (see above)
                        tail_n = total;
                        alive = false;
                        break;
                    }
                }
                const int slot = pos_in_set * 64 + lane;
                alive = slot < total;
                if (alive) {
                    pos = V3{pvt_u2d(xbuf[0 * X + slot]), pvt_u2d(xbuf[1 * X + slot]), pvt_u2d(xbuf[2 * X + slot])};
                    dir = V3{pvt_u2d(xbuf[3 * X + slot]), pvt_u2d(xbuf[4 * X + slot]), pvt_u2d(xbuf[5 * X + slot])};
                    wl = pvt_u2d(xbuf[6 * X + slot]); travelled = pvt_u2d(xbuf[7 * X + slot]); duration = pvt_u2d(xbuf[8 * X + slot]);
                    rng.s0 = xbuf[9 * X + slot]; rng.s1 = xbuf[10 * X + slot]; rng.s2 = xbuf[11 * X + slot]; rng.s3 = xbuf[12 * X + slot];
                    const unsigned long long cs_ = xbuf[13 * X + slot];
                    count = (int)(unsigned int)cs_;
                    source = (int)(unsigned int)(cs_ >> 32);
#pragma unroll
                    for (int w = 0; w < SEENW; w++) seen.w[w] = xbuf[(14 + w) * X + slot];
                    if constexpr (RECORD) {
                        const unsigned long long rn_ = xbuf[(14 + SEENW) * X + slot];
                        rec_slot = (int)(unsigned int)rn_;
                        nev = (int)(unsigned int)(rn_ >> 32);
                    }
                }
                // the set shrinks to its `keep` lowest members
                int kept = 0, m2 = 0;
#pragma unroll
                for (int w = 0; w < kWaves; w++)
                    if (((members >> w) & 1) && kept < keep) { m2 |= 1 << w; kept += 1; }
                members = m2;
                if (keep == 1) ws = (ws & ~(unsigned int)WS_REGIME) | WS_SOLO;
            }
        }
        }   // (!TAIL)
#if PVT_STATS
        {
            unsigned long long live = __popcll(__ballot(alive));
            st_iters += 1; st_lane_steps += live;
            if (ws & WS_EXHAUSTED) { st_drain_iters += 1; st_drain_lane_steps += live; }
        }
#endif

#if PVT_COUNTERS
        c_iters += 1u;
#endif
#if PVT_TIMELINE
        if (tl_iters == 0) tl_t[2] = wall_clock64();
        if ((ws & WS_EXHAUSTED) && tl_t[3] == 0) tl_t[3] = wall_clock64();
        tl_iters += 1;
#endif
        PVT_MARK(0);  // refill + drain bookkeeping
        PVT_COUNT(0, alive);
        // ================= one step for every live lane ==================
        // deferred event of this step
        int ev_kind = -1, ev_hit = -1, ev_container = -1, ev_adjacent = -1, ev_component = -1;
        bool ev_normal = false, terminal = false;
        int t_sel = -1, t_node = -1;
        bool t_normal = false;
        double t_cos = 1.0;   // cosine of the angle the tallies record (acos is taken when a recorder needs it)
        V3 nrm{0, 0, 0};
        int tri1 = -1;   // triangle record of the nearest crossing when it lies on a mesh
        bool em = false;   // re-emitted in this step

        // ---- stage 1: where does the ray go?  (every live lane) ------------------------------------
        // Lane classes for the rest of the step; the divergent bodies below are keyed on them.
        enum { CLS_NONE = 0, CLS_SURF, CLS_EXIT, CLS_ABS };
        int cls = CLS_NONE;
        int hit = -1, container = -1, adjacent = -1;
        double t0 = 0.0;
        bool pend = false;   // reaches the exit / absorption / surface decision
        if (alive) {
            count += 1;
            bool budget_kill = false;
            if constexpr (RECORD) budget_kill = (rec_slot >= 0 && nev >= A.max_events - 1);
            if (budget_kill) {
                // event budget exhausted: KILL row, no tally (_kernel.pyx:658-663)
                ev_kind = PVT_EV_KILL;
                terminal = true;
            } else {
                // ---- intersect every node, fold nearest/second/container ----
                int nhits = 0, n1 = -1, n2 = -1, cnode = -1;
                tri1 = -1;
                double t1 = INFINITY, t2 = INFINITY, cbest = INFINITY;
                int c2node = -1;            // (mesh scenes) the next-nearest node that holds the ray
                double c2best = INFINITY;
                V3 d{0, 0, 0};
                double inv[3] = {0, 0, 0};
                bool inv_ok = false;   // wave-uniform
                int rot = -1;          // rotation class of `d` (wave-uniform)
                // The root last.  Every photon is inside the root, which therefore contributes exactly one
                // forward crossing, farther than any crossing of a node that lies strictly inside it (the
                // host has checked that with a margin: no ties, so the visiting order cannot matter).  All the
                // rest of the step needs from that crossing is its ORDER among the others -- unless it is the
                // nearest one (the photon leaves the scene) and somebody looks at where it leaves (an event
                // log, an `exit` recorder), in which case this launch was not given `lazy_root`.  So the lanes
                // compare the crossings they have with a cheap lower bound of the root's distance -- the
                // distance to the root's nearest face -- and only the lanes it cannot decide for pay for the
                // root's intersection; in a typical scene (a 5 cm slab in a 5 m world) none ever does.
                // (tail function: also where exits are looked at -- `lazy_exact`: the root's own distance is worked out whenever
                // it is the nearest crossing, i.e. the photon leaves the scene)
                const bool lazy_exact = TAIL && PVT_TAIL_LAZY && !MESH && (uf(UF_TAIL_LAZY1) || uf(UF_TAIL_LAZY2));
                const int lazy_root = MESH ? 0 : (lazy_exact ? (uf(UF_TAIL_LAZY1) ? 1 : 2)
                                                              : (RECORD ? 0 : (uf(UF_LAZY1) ? 1 : (uf(UF_LAZY2) ? 2 : 0))));   // wave-uniform
                // Grid scenes fold a crossing by the key (t, node): what the reference's first-minimum scans over its hit
                // list (nodes ascending) come to, whatever the order the nodes are visited in.  Written as selects of
                // VALUES: as branches that assign, the compiler merges the assignments into stores through a selected
                // POINTER, and nearest / second-nearest then live in scratch memory.
                // (Every captured variable is read ONCE, up front, and written once at the end: a lambda's body is optimised
                // before it is inlined, while its captures are still pointers, and a choice between two of them read in
                // different branches becomes a load through a chosen pointer -- which pins them to memory for good.)
                auto fold_by_key = [&](double t, int node, int& nl, double& tfirst) __attribute__((always_inline)) {
                    const double a1 = t1, a2 = t2, tf = tfirst;
                    const int m1 = n1, m2 = n2, have = nhits, l = nl;
                    const bool none = have == 0;
                    const bool nearest = none || t < a1 || (t == a1 && node < m1);
                    const bool second = !nearest && (m2 < 0 || t < a2 || (t == a2 && node < m2));
                    const bool shift = nearest && !none;
                    tfirst = l == 0 ? t : tf;
                    nl = l + 1;
                    t1 = nearest ? t : a1;
                    n1 = nearest ? node : m1;
                    t2 = shift ? a1 : (second ? t : a2);
                    n2 = shift ? m1 : (second ? node : m2);
                    nhits = have + 1;
                };
                if constexpr (GRID) {
                    // ---- scenes of many nodes: every lane looks up ITS candidates in a uniform grid over the nodes -----
                    // The reference intersects every node in every step (_kernel.pyx:666-680).  A node the ray does not
                    // cross contributes nothing to that loop, and all the step needs of the others is the nearest and
                    // the second-nearest crossing and the nearest node crossed exactly once.  The host files every
                    // node but the root under the cells of a grid that its bounding box (grown by a margin far above
                    // rounding) touches; a lane walks the cells its ray passes, front to back (3-D DDA), and runs the
                    // reference's own intersection arithmetic on the nodes it finds there, each once.  Crossings are
                    // folded by (t, node) -- the order the reference's first-minimum scans imply -- so the visiting
                    // order cannot matter.  The walk ends where the ray leaves the grid, or earlier: once two
                    // crossings lie nearer than the end of the cells visited (by `guard`, the margin again), every
                    // node not yet seen has all its crossings beyond them -- it cannot be the nearest or the second
                    // -- and lies clear of the photon, so a convex box or sphere is crossed twice or not at all and
                    // cannot hold the ray either.  (A cylinder grazed at the rim of a cap may come out with ONE
                    // crossing in the reference's arithmetic, _kernel.pyx:301-345: scenes with cylinders -- `odd` --
                    // additionally walk on until the cells visited reach past the container found so far.)
                    const int gb = L.grid_d;
                    const unsigned long long gbits = pvt_d2u(T.du(gb + 13));   // nx | ny << 8 | nz << 16 | words << 24 | odd << 28
                    const double guard = T.du(gb + 12);
                    // (everything the walk keeps is a named scalar: a local array that is selected from or passed by reference
                    // ends up in scratch memory, and a scratch access is a trip to the memory system in the middle of the walk)
                    const double invw0 = rcp_normal(dir.x), invw1 = rcp_normal(dir.y), invw2 = rcp_normal(dir.z);
                    const double invw[3] = {invw0, invw1, invw2};   // (set-up only, fully unrolled)
                    const double pw[3] = {pos.x, pos.y, pos.z}, dw[3] = {dir.x, dir.y, dir.z};
                    // the ray against the grid's box (a direction component this small moves the photon by less than
                    // rounding over the whole scene: treated as parallel)
                    double t_in = 0.0, t_out = INFINITY;
                    bool walk = true;
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        const double lo = T.du(gb + a), hi = T.du(gb + 3 + a);
                        if (pvt_fabs(dw[a]) < 1e-20) {
                            if (pw[a] < lo || pw[a] > hi) walk = false;
                        } else {
                            const double ta = (lo - pw[a]) * invw[a], tb = (hi - pw[a]) * invw[a];
                            t_in = __builtin_fmax(t_in, __builtin_fmin(ta, tb));
                            t_out = __builtin_fmin(t_out, __builtin_fmax(ta, tb));
                        }
                    }
                    if (!(t_in <= t_out)) walk = false;
                    // first cell, the distance at which the ray leaves it along each axis, steps left along each axis
                    double tm0 = INFINITY, tm1 = INFINITY, tm2 = INFINITY;
                    unsigned int remv = 0;   // steps left: x | y << 8 | z << 16; bits 24-26: the ray runs towards lower indices
                    int ci = 0;
                    {
                        int stride = 1;
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            const int na = (int)((unsigned int)(gbits >> (8 * a)) & 0xffu);
                            const double lo = T.du(gb + a), cell = T.du(gb + 6 + a), rcell = T.du(gb + 9 + a);
                            const double at = pw[a] + dw[a] * t_in;
                            int c = (int)((at - lo) * rcell);
                            c = c < 0 ? 0 : (c > na - 1 ? na - 1 : c);
                            const bool par = pvt_fabs(dw[a]) < 1e-20, neg = dw[a] < 0.0;
                            const double tma = par ? INFINITY : ((lo + (double)(c + (neg ? 0 : 1)) * cell) - pw[a]) * invw[a];
                            if (a == 0) tm0 = tma; else if (a == 1) tm1 = tma; else tm2 = tma;
                            remv |= (unsigned int)(neg ? c : na - 1 - c) << (8 * a);
                            if (neg) remv |= 1u << (24 + a);
                            ci += c * stride;
                            stride *= na;
                        }
                    }
                    const int words = (int)((unsigned int)(gbits >> 24) & 0xfu);
                    const bool odd = ((gbits >> 28) & 1ull) != 0;
                    unsigned long long seen_lo = 0ull, seen_hi = 0ull;   // nodes this lane has tested in this step
                    // (always_inline: a closure that is CALLED holds pointers to the variables it captured, which then live in
                    // scratch memory)
                    // one node, tested by this lane alone (the records come from LDS with per-lane addresses)
                    // One node, tested by this lane alone (records from LDS with per-lane addresses).  The root comes last
                    // and goes through here too -- with the lazy shortcut of the plain node loop below, when the launch
                    // has it -- so the intersection code exists once in the kernel's text.
                    auto visit = [&](int node) __attribute__((always_inline)) {
                        const int hn = node * ND;
                        const double tx = T.dv(hn + ND_T), ty = T.dv(hn + ND_T + 1), tz = T.dv(hn + ND_T + 2);
                        const double g0 = T.dv(hn + ND_PARAMS), g1 = T.dv(hn + ND_PARAMS + 1), g2 = T.dv(hn + ND_PARAMS + 2);
                        const unsigned long long hb = pvt_d2u(T.dv(hn + ND_BITS));
                        int gt = (int)(((unsigned int)hb >> 8) & 0xffu);
                        V3 o, dl;
                        double il0, il1, il2;
                        if ((hb & 1ull) != 0) {
                            o.x = pos.x + tx; o.y = pos.y + ty; o.z = pos.z + tz;
                            dl = dir;
                            il0 = invw0; il1 = invw1; il2 = invw2;
                        } else {
                            const int rm = L.rot_d + (int)(unsigned int)(hb >> 32) * RT + RT_W2L;
                            o.x = T.dv(rm + 0) * pos.x + T.dv(rm + 1) * pos.y + T.dv(rm + 2) * pos.z + tx;
                            o.y = T.dv(rm + 3) * pos.x + T.dv(rm + 4) * pos.y + T.dv(rm + 5) * pos.z + ty;
                            o.z = T.dv(rm + 6) * pos.x + T.dv(rm + 7) * pos.y + T.dv(rm + 8) * pos.z + tz;
                            dl.x = T.dv(rm + 0) * dir.x + T.dv(rm + 1) * dir.y + T.dv(rm + 2) * dir.z;
                            dl.y = T.dv(rm + 3) * dir.x + T.dv(rm + 4) * dir.y + T.dv(rm + 5) * dir.z;
                            dl.z = T.dv(rm + 6) * dir.x + T.dv(rm + 7) * dir.y + T.dv(rm + 8) * dir.z;
                            il0 = rcp_normal(dl.x); il1 = rcp_normal(dl.y); il2 = rcp_normal(dl.z);
                        }
                        bool root_known = false;
                        if (lazy_root && node == k_root) {   // (see the plain node loop: same bound, same conditions)
                            double bound;
                            if (lazy_root == 1) {
                                bound = __builtin_fmin(__builtin_fmin(0.5 * g0 - pvt_fabs(o.x), 0.5 * g1 - pvt_fabs(o.y)), 0.5 * g2 - pvt_fabs(o.z));
                            } else {
                                bound = (g0 * g0 - dot3(o, o)) * A.lazy_k;
                            }
                            const bool undecided = !(bound > 0.0) || (nhits > 0 && !(t1 < bound)) || (nhits >= 2 && !(t2 < bound)) ||
                                                   (cnode >= 0 && !(cbest < bound)) || (lazy_exact && nhits == 0);
                            {   // one more crossing, behind all the others: nearest only if there is no other
                                const double a1 = t1, a2 = t2;
                                const int m1 = n1, m2 = n2, have = nhits, cn = cnode;
                                const bool known = !undecided;
                                t1 = known && have == 0 ? bound : a1; n1 = known && have == 0 ? node : m1;
                                t2 = known && have == 1 ? INFINITY : a2; n2 = known && have == 1 ? node : m2;
                                nhits = have + (known ? 1 : 0);
                                cnode = known && cn < 0 ? node : cn;   // crossed once: it holds the ray unless a nearer node does
                                root_known = known;
                                gt = known ? -1 : gt;   // nothing to intersect
                            }
                        }
                        int nl = 0;
                        double tfirst = 0.0;
                        auto fold = [&](double t) __attribute__((always_inline)) { fold_by_key(t, node, nl, tfirst); };
                        if (gt >= 0) shape_hits(gt, g0, g1, g2, o, dl, il0, il1, il2, fold);
                        {
                            const double cb = cbest;
                            const int cn = cnode;
                            const bool holds = nl == 1 && !root_known && (tfirst < cb || (tfirst == cb && node < cn));
                            cbest = holds ? tfirst : cb;
                            cnode = holds ? node : cn;
                        }
                    };
                    auto next_cell = [&]() __attribute__((always_inline)) {   // leave the current cell: done, or on to the next one
                        const double t_cell = __builtin_fmin(tm0, __builtin_fmin(tm1, tm2));   // the ray leaves the cell here
                        const bool enough = nhits >= 2 && t2 + guard < t_cell && (!odd || (cnode >= 0 && cbest + guard < t_cell));
                        const int ax = (tm0 <= tm1 && tm0 <= tm2) ? 0 : (tm1 <= tm2 ? 1 : 2);
                        const unsigned int left = (remv >> (8 * ax)) & 0xffu;
                        if (enough || left == 0u || !(t_cell < INFINITY)) {
                            walk = false;
                        } else {
                            remv -= 1u << (8 * ax);
                            const int nx = (int)((unsigned int)gbits & 0xffu), ny = (int)((unsigned int)(gbits >> 8) & 0xffu);
                            const int stride = ax == 0 ? 1 : (ax == 1 ? nx : nx * ny);
                            ci += ((remv >> (24 + ax)) & 1u) ? -stride : stride;
                            if (ax == 0) tm0 = __builtin_fma(pvt_fabs(invw0), T.du(gb + 6), tm0);
                            else if (ax == 1) tm1 = __builtin_fma(pvt_fabs(invw1), T.du(gb + 7), tm1);
                            else tm2 = __builtin_fma(pvt_fabs(invw2), T.du(gb + 8), tm2);
                        }
                    };
                    // Every lane at its own pace.  A trip: a lane whose current cell holds nothing it has not tested moves on to
                    // its next cell (or out of the walk); then it tests ONE node -- of the cell it is in now or, the walk
                    // over, the root.  (Both halves run in every trip of a wave anyway, its lanes being at different points.)
                    bool root_todo = true;
                    auto untested = [&](unsigned long long& m_lo, unsigned long long& m_hi) __attribute__((always_inline)) {
                        const int at = gb + 14 + ci * words;
                        m_lo = pvt_d2u(T.dv(at)) & ~seen_lo;
                        m_hi = words > 1 ? pvt_d2u(T.dv(at + 1)) & ~seen_hi : 0ull;
                    };
                    for (;;) {
                        if (__ballot(walk || root_todo) == 0ull) break;
                        unsigned long long m_lo = 0ull, m_hi = 0ull;
                        if (walk) {
                            untested(m_lo, m_hi);
                            if ((m_lo | m_hi) == 0ull) {
                                next_cell();
                                if (walk) untested(m_lo, m_hi);
                            }
                        }
                        bool test = false;
                        int node = k_root;
                        if (m_lo != 0ull) { node = __builtin_ctzll(m_lo); seen_lo |= m_lo & (0ull - m_lo); test = true; }
                        else if (m_hi != 0ull) { node = 64 + __builtin_ctzll(m_hi); seen_hi |= m_hi & (0ull - m_hi); test = true; }
                        else if (!walk && root_todo) { root_todo = false; test = true; }
#if PVT_STATS
                        st_g[0] += 1; st_g[1] += 1; st_g[2] += __popcll(__ballot(test)); st_g[3] += __popcll(__ballot(!test));
#endif
                        if (test) visit(node);
                    }
                }
                // (grid scenes have visited every node they need by now)
                for (int k = 0; k < (GRID ? 0 : k_nodes); k++) {
                    const int node = !lazy_root ? k : (k == k_nodes - 1 ? k_root : (k < k_root ? k : k + 1));
                    // An unrotated node (identity rotation, bit for bit -- the usual case) only translates:
                    // 1*x + 0*y + 0*z + t equals x + t up to the sign of a zero, which no comparison,
                    // quotient or stored value below can see.
                    // the node's 64-byte record: one scalar load, all of it in SGPRs
                    const int hn = node * ND;
                    const double tx = T.du(hn + ND_T), ty = T.du(hn + ND_T + 1), tz = T.du(hn + ND_T + 2);
                    const double gpar[3] = {T.du(hn + ND_PARAMS), T.du(hn + ND_PARAMS + 1), T.du(hn + ND_PARAMS + 2)};
                    const unsigned long long hbits = pvt_d2u(T.du(hn + ND_BITS));
                    const bool ident = (hbits & 1ull) != 0;   // wave-uniform
                    const int rc = (int)(unsigned int)(hbits >> 32);   // rotation class
                    V3 o;
                    if (ident) {
                        o.x = pos.x + tx; o.y = pos.y + ty; o.z = pos.z + tz;
                    } else {
                        const int rm = L.rot_d + rc * RT + RT_W2L;
                        o.x = T.du(rm + 0) * pos.x + T.du(rm + 1) * pos.y + T.du(rm + 2) * pos.z + tx;
                        o.y = T.du(rm + 3) * pos.x + T.du(rm + 4) * pos.y + T.du(rm + 5) * pos.z + ty;
                        o.z = T.du(rm + 6) * pos.x + T.du(rm + 7) * pos.y + T.du(rm + 8) * pos.z + tz;
                    }
                    // Nodes whose world->local rotations are bit-identical (the host files them under the
                    // first such node) see the same local direction: it and its reciprocals are reused.
                    if (rc != rot) {
                        if (ident) {
                            d = dir;
                        } else {
                            const int rm = L.rot_d + rc * RT + RT_W2L;
                            d.x = T.du(rm + 0) * dir.x + T.du(rm + 1) * dir.y + T.du(rm + 2) * dir.z;
                            d.y = T.du(rm + 3) * dir.x + T.du(rm + 4) * dir.y + T.du(rm + 5) * dir.z;
                            d.z = T.du(rm + 6) * dir.x + T.du(rm + 7) * dir.y + T.du(rm + 8) * dir.z;
                        }
                        rot = rc;
                        inv_ok = false;
                    }
                const int gt = (int)(((unsigned int)hbits >> 8) & 0xffu);
                // Hits are folded as they are found, in the reference's (node, k)
                // order, so no per-ray hit list exists; the tie-breaks equal the
                // reference's argmin scans over its hit arrays (:684-714).
                int nl = 0;
                double tfirst = 0.0;
                bool root_known = false;   // this lane's root crossing is accounted for without its distance
                if (lazy_root && node == k_root) {
                    double bound;   // <= distance to the root's surface along any direction
                    if (lazy_root == 1) {
                        const double mx = 0.5 * gpar[0] - pvt_fabs(o.x), my = 0.5 * gpar[1] - pvt_fabs(o.y),
                                     mz = 0.5 * gpar[2] - pvt_fabs(o.z);
                        bound = __builtin_fmin(__builtin_fmin(mx, my), mz);
                    } else {   // (R^2 - |o|^2) / (2R) <= R - |o|
                        const double radius = gpar[0];
                        bound = (radius * radius - dot3(o, o)) * A.lazy_k;
                    }
                    const bool undecided = !(bound > 0.0) || (nhits > 0 && !(t1 < bound)) || (nhits >= 2 && !(t2 < bound)) ||
                                           (cnode >= 0 && !(cbest < bound)) || (lazy_exact && nhits == 0);
                    if (!undecided) {
                        // one more crossing, behind all the others: nearest only if there is no other
                        if (nhits == 0) { t1 = bound; n1 = node; }
                        else if (nhits == 1) { t2 = INFINITY; n2 = node; }
                        nhits += 1;
                        if (cnode < 0) cnode = node;   // crossed once: it holds the ray unless a nearer node does
                        root_known = true;
                    }
                }
                auto fold = [&](double t) __attribute__((always_inline)) {
                    if (nl == 0) tfirst = t;
                    nl += 1;
                    if (nhits == 0) { t1 = t; n1 = node; }
                    else if (t < t1) { t2 = t1; n2 = n1; t1 = t; n1 = node; }
                    else if (n2 < 0 || t < t2) { t2 = t; n2 = node; }
                    nhits += 1;
                };
                if (root_known) {
                    // nothing to intersect
                } else if (MESH && gt == PVT_GEOM_MESH) {
                    // EXTENSION (no reference counterpart, see include/pvtrace_hip.h): every
                    // forward crossing of the node's triangles, found by a stack-free walk of
                    // the BVH (pvt_bvh.h).  Crossings of one mesh are ordered by
                    // (t, face) so the result does not depend on the walk order.
                    const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
                    const double ax = pvt_fabs(d.x), ay = pvt_fabs(d.y), az = pvt_fabs(d.z);
                    const int kz = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
                    int kx = kz == 2 ? 0 : kz + 1, ky = kx == 2 ? 0 : kx + 1;
                    auto pick = [](const double* v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : v[2]); };
                    const double dz = pick(dd, kz);
                    if (dz < 0.0) { int tmp = kx; kx = ky; ky = tmp; }
                    // (the shear constants feed the exact triangle test: the short division sequences give RN(x / dz) and RN(1 / dz)
                    // bit for bit when dz -- the dominant direction component -- is an ordinary number and the numerators are zero
                    // or not tiny, which a wave checks once; anything else takes the general divisions)
                    double shx, shy, shz;
                    {
                        const double sx_ = pick(dd, kx), sy_ = pick(dd, ky);
                        auto ordinary = [](double x) { return x == 0.0 || (pvt_fabs(x) > 1e-280 && pvt_fabs(x) < 1e280); };
                        if (__ballot(!(pvt_fabs(dz) > 1e-100 && pvt_fabs(dz) < 1e100) || !ordinary(sx_) || !ordinary(sy_)) == 0ull) {
                            shx = div_normal(sx_, dz); shy = div_normal(sy_, dz); shz = rcp_normal(dz);
                        } else {
                            shx = sx_ / dz; shy = sy_ / dz; shz = 1.0 / dz;
                        }
                    }
                    // Box culling only has to be conservative (a false hit costs a triangle test, a
                    // false miss would lose a crossing).  A ray parallel to a slab gets a huge finite
                    // reciprocal instead of inf: inside the slab the two plane distances then have
                    // opposite signs (interval covers everything), outside the same sign (pushed out
                    // of range, or a harmless false hit), and 0 * inf = NaN can never arise.
                    double minv[3];
#pragma unroll
                    for (int a = 0; a < 3; a++) minv[a] = pvt_fabs(dd[a]) < 1e-300 ? 1e300 : rcp_normal(dd[a]);   // (culling only)
                    long long f1 = -1, f2 = -1;   // faces of this node's entries in (t1, t2)
                    int i = T.iu(node * NI + NI_MESH);
                    // (the two table pointers in registers for the walk: the build keeps loop invariants where they are used --
                    // good for the photon loop as a whole, but here that is a scalar load of the kernel argument, and a wait
                    // for it, in every iteration of the walk)
                    const pvt::BvhNode* bvh = A.bvh;
                    const pvt::MeshTri* tris = A.tris;
                    asm volatile("" : "+s"(bvh), "+s"(tris));
                    // The walk waits for its records more than it computes (a wave of the 20 480-face ball: 62 % of its cycles,
                    // ~1 000 per box): the top levels of every tree -- where every ray passes -- are read from a copy in LDS
                    // (pvt_bvh.h: stage_top).  A cursor is the index of a record in global memory, or kTopFlag | slot in
                    // that copy; skip links are stored in the same form, the walk is over when the cursor equals the
                    // tree's end.
                    const pvt::BvhNode* const ltop = reinterpret_cast<const pvt::BvhNode*>(reinterpret_cast<const char*>(smem) + A.top_off);
                    // (records in global memory through a buffer descriptor: the address of a record is its byte offset, one
                    // shift, instead of a 64-bit multiply-add, and the loads do not count as LDS traffic to wait for)
                    const __amdgpu_buffer_rsrc_t brec = __builtin_amdgcn_make_buffer_rsrc(const_cast<pvt::BvhNode*>(bvh), 0, -1, 0x00020000);
                    auto record_at = [&](int index) __attribute__((always_inline)) {
                        typedef unsigned int Word4 __attribute__((ext_vector_type(4)));
                        struct Pair { Word4 a, b; } w;
                        w.a = __builtin_amdgcn_raw_buffer_load_b128(brec, index << 5, 0, 0);
                        w.b = __builtin_amdgcn_raw_buffer_load_b128(brec, (index << 5) + 16, 0, 0);
                        return __builtin_bit_cast(pvt::BvhNode, w);
                    };
                    // (the root's record -- its box, the tree's end, where its copy is -- through the scalar cache: the node is the wave's)
                    const __attribute__((address_space(4))) pvt::BvhNode* const rootp =
                        (const __attribute__((address_space(4))) pvt::BvhNode*)(A.bvh + i);
                    const float rlo[3] = {rootp->lo[0], rootp->lo[1], rootp->lo[2]}, rhi[3] = {rootp->hi[0], rootp->hi[1], rootp->hi[2]};
                    const int end = rootp->skip, rlink = rootp->link;
                    const int n_records = end - i;   // (before the cursor moves to the copy)
                    pvt::BvhNode b;
                    // The boxes are tested in f32, relative to the mesh's centre (the node's parameter slots) and from where
                    // the ray ENTERS the root box -- every quantity then has the mesh's own size, the test's rounding displaces
                    // a plane by < 5e-7 of the mesh's diagonal, and the host pads every box by 4e-6 of it (pvt_bvh.h).
                    float oi[3], invf[3];   // (a plane's distance is fma(plane, 1/d, -(o/d)): o/d once per walk)
                    {
                        double t_near = 0.0, t_far = INFINITY;
                        const double cc[3] = {gpar[0], gpar[1], gpar[2]};
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            const double ta = (((double)rlo[a] + cc[a]) - oo[a]) * minv[a], tb = (((double)rhi[a] + cc[a]) - oo[a]) * minv[a];
                            t_near = __builtin_fmax(t_near, __builtin_fmin(ta, tb));
                            t_far = __builtin_fmin(t_far, __builtin_fmax(ta, tb));
                        }
                        if (rlink >= 0 && (rlink & pvt::kTopFlag) != 0) {   // the tree has a copy in LDS: the root's slot precedes its children's
                            i = rlink - 1;
                            b = ltop[i & pvt::kIndexMask];
                        } else {
                            b = record_at(i);
                        }
                        if (t_far < t_near) i = end;   // the ray misses the mesh's box altogether
#pragma unroll
                        for (int a = 0; a < 3; a++) {
                            invf[a] = pvt_fabs(dd[a]) < 1e-20 ? 1e20f : (float)minv[a];   // (a component this small: the ray is parallel to the pair of planes)
                            oi[a] = (float)((oo[a] - cc[a]) + dd[a] * t_near) * invf[a];
                        }
                    }
                    // The triangle tests are DEFERRED: a lane that reaches a leaf only notes it (four slots in registers) and
                    // walks on until its slots are full; when every lane of the wave has stopped -- slots full or walk
                    // over -- the notes are worked off together, every lane on one of its leaves at once, and the walks
                    // resume.  Tested inside the walk, the watertight
                    // test (185 vector instructions against the 50 of a box) ran in four iterations out of five for
                    // the one or two lanes that happened to stand at a leaf.  (Crossings are ordered by (t, face): the
                    // order they are found in does not matter.)
                    // (A mesh of a few triangles -- a tree of a handful of nodes, leaves of up to eight triangles -- has no
                    // walk to speak of to keep going: its leaves are tested as they are met.  Wave-uniform: the node is.)
                    // (the notes live in LDS, one column per lane: in registers they were a shift register whose copies the
                    // compiler carried through every path of the walk -- 14 moves per box)
                    unsigned int* const mq = reinterpret_cast<unsigned int*>(reinterpret_cast<char*>(smem) + A.meshq_off) + threadIdx.x;
                    int qn = 0;
                    const int qcap = n_records <= 16 ? 1 : kMeshQ;
                    for (;;) {
                        // ---- walk: every lane goes on until its walk is over or its slots are full
                        for (;;) {
                            const bool go = i != end && qn < qcap;
                            if (__ballot(go) == 0ull || __popcll(__ballot(i != end && qn >= qcap)) >= 4) break;   // (lanes with full leaf slots that end a walk phase: 1 / 4 / 16 measured within 3 %)
#if PVT_STATS
                            if (MESH) { st_g[0] += 1; st_g[1] += __popcll(__ballot(go)); }
#endif
                            if (go) {
                                float tmin = 0.0f, tmax = INFINITY;
#pragma unroll
                                for (int a = 0; a < 3; a++) {
                                    const float ta = __builtin_fmaf(b.lo[a], invf[a], -oi[a]), tb = __builtin_fmaf(b.hi[a], invf[a], -oi[a]);
                                    tmin = __builtin_fmaxf(tmin, __builtin_fminf(ta, tb));
                                    tmax = __builtin_fminf(tmax, __builtin_fmaxf(ta, tb));
                                }
                                // the successor: after a miss and after a leaf the skip link, after a hit of an inner record its first
                                // child (whose own skip link is its sibling, in the same 64-byte line)
                                int next = b.skip;
                                if (!(tmax < tmin)) {
                                    if (b.link < 0) {   // a leaf: its triangle later (the newest note first: any order will do)
                                        mq[qn * kBlock] = (unsigned int)b.link;
                                        qn += 1;
                                    } else {
                                        next = b.link;
                                    }
                                }
                                i = next;
                                if (i != end) {
                                    if (i & pvt::kTopFlag) b = ltop[i & pvt::kIndexMask];
                                    else b = record_at(i);
                                }
                            }
                        }
                        // ---- the noted leaves, every lane on one of its own at a time
                        while (__ballot(qn > 0) != 0ull) {
                            if (qn > 0) {
                                qn -= 1;
                                const int leaf = (int)mq[qn * kBlock];
                                constexpr int tn = 1;   // (one triangle per leaf: pvt_bvh.h)
                                const int tri_start = leaf & pvt::kIndexMask;
                                const pvt::MeshTri* tr = tris + tri_start;
                                for (int k = 0; k < tn; k++, tr++) {
#if PVT_STATS
                                    if (MESH) { st_g[2] += 1; st_g[3] += __popcll(__ballot(true)); }
#endif
                                    double va[3], vb[3], vc[3];
#pragma unroll
                                    for (int a = 0; a < 3; a++) {
                                        va[a] = tr->v[a] - oo[a]; vb[a] = tr->v[3 + a] - oo[a]; vc[a] = tr->v[6 + a] - oo[a];
                                    }
                                    const double az_ = pick(va, kz), bz_ = pick(vb, kz), cz_ = pick(vc, kz);
                                    const double axs = pick(va, kx) - shx * az_, ays = pick(va, ky) - shy * az_;
                                    const double bxs = pick(vb, kx) - shx * bz_, bys = pick(vb, ky) - shy * bz_;
                                    const double cxs = pick(vc, kx) - shx * cz_, cys = pick(vc, ky) - shy * cz_;
                                    const double u = cxs * bys - cys * bxs;
                                    const double v = axs * cys - ays * cxs;
                                    const double w = bxs * ays - bys * axs;
                                    // (one flag instead of a chain of exits: every lane of the trip is on a triangle of its own, and each
                                    // early exit was a pair of exec-mask instructions and a branch for all of them)
                                    const double det = u + v + w;
                                    bool ok = !((u < 0.0 || v < 0.0 || w < 0.0) && (u > 0.0 || v > 0.0 || w > 0.0)) && det != 0.0;
                                    // exact zeros of an edge function (a ray through an edge or a vertex): the ownership rule, only when a
                                    // lane of the wave has one
                                    if (__ballot(ok && (u == 0.0 || v == 0.0 || w == 0.0)) != 0ull) {
                                        const double sg = det < 0.0 ? -1.0 : 1.0;
                                        auto owned = [](double gx, double gy) { return gx > 0.0 || (gx == 0.0 && gy > 0.0); };
                                        if (u == 0.0 && !owned(sg * (cys - bys), sg * (bxs - cxs))) ok = false;
                                        if (v == 0.0 && !owned(sg * (ays - cys), sg * (cxs - axs))) ok = false;
                                        if (w == 0.0 && !owned(sg * (bys - ays), sg * (axs - bxs))) ok = false;
                                    }
                                    if (!ok) continue;
                                    const double t = (u * (shz * az_) + v * (shz * bz_) + w * (shz * cz_)) / det;
                                    if (!(t > kEps)) continue;
                                    const long long face = tr->face;
                                    const int tri = tri_start + k;
                                    if (nl == 0 || t < tfirst) tfirst = t;
                                    nl += 1;
                                    if (nhits == 0) { t1 = t; n1 = node; tri1 = tri; f1 = face; }
                                    else if (t < t1 || (t == t1 && f1 >= 0 && face < f1)) {
                                        t2 = t1; n2 = n1; f2 = f1; t1 = t; n1 = node; tri1 = tri; f1 = face;
                                    } else if (n2 < 0 || t < t2 || (t == t2 && f2 >= 0 && face < f2)) { t2 = t; n2 = node; f2 = face; }
                                    nhits += 1;
                                }
                            }
                        }
                        if (__ballot(i != end) == 0ull) break;
                    }
                } else if (gt == PVT_GEOM_BOX) {  // slab test (_kernel.pyx:245-276)
                double tmin = -INFINITY, tmax = INFINITY;
                bool miss = false;
                const double oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
                if (!inv_ok) {   // 1/d per axis, shared by consecutive nodes whose rotations have the same bits
#pragma unroll
                    for (int a = 0; a < 3; a++) inv[a] = rcp_normal(dd[a]);   // 1/d (garbage below 1e-300: never used)
                    inv_ok = true;
                }
                // a ray parallel to a pair of faces (a direction component below 1e-300) takes the reference's
                // inside/outside test for that axis; the wave only runs the general form when a lane holds one
                if (__ballot(pvt_fabs(dd[0]) < 1e-300 || pvt_fabs(dd[1]) < 1e-300 || pvt_fabs(dd[2]) < 1e-300) == 0ull) {
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        const double sz = gpar[a];
                        const double ta = (-0.5 * sz - oo[a]) * inv[a], tb = (0.5 * sz - oo[a]) * inv[a];
                        tmin = __builtin_fmax(tmin, __builtin_fmin(ta, tb));
                        tmax = __builtin_fmin(tmax, __builtin_fmax(ta, tb));
                    }
                } else
#pragma unroll
                for (int a = 0; a < 3; a++) {
                    double sz = gpar[a];
                    double lo = -0.5 * sz, hi = 0.5 * sz;
                    if (pvt_fabs(dd[a]) < 1e-300) {
                        if (oo[a] < lo || oo[a] > hi) miss = true;
                    } else {
                        // (the reference swaps ta, tb into order and keeps the largest entry / smallest exit
                        // distance: min and max of finite numbers, which is what these are)
                        const double ta = (lo - oo[a]) * inv[a], tb = (hi - oo[a]) * inv[a];
                        tmin = __builtin_fmax(tmin, __builtin_fmin(ta, tb));
                        tmax = __builtin_fmin(tmax, __builtin_fmax(ta, tb));
                    }
                }
                if (!miss && !(tmax < tmin)) {
                    if (tmin > kEps) fold(tmin);
                    if (tmax > kEps) fold(tmax);
                }
            } else if (gt == PVT_GEOM_SPHERE) {  // (:279-298)
                    double radius = gpar[0];
                    double a = dot3(d, d), b = 2.0 * dot3(d, o), c = dot3(o, o) - radius * radius;
                    double disc = b * b - 4.0 * a * c;
                    if (!(disc < 0.0)) {
                        double sq = pvt_sqrt(disc);
                        double t = (-b - sq) / (2.0 * a);
                        if (t > kEps) fold(t);
                        t = (-b + sq) / (2.0 * a);
                        if (t > kEps) fold(t);
                    }
                } else {  // capped z cylinder (:301-345)
                    double half = 0.5 * gpar[0], radius = gpar[1];
                    double a = d.x * d.x + d.y * d.y;
                    if (a > 1e-300) {
                        double b = 2.0 * (o.x * d.x + o.y * d.y);
                        double c = o.x * o.x + o.y * o.y - radius * radius;
                        double disc = b * b - 4.0 * a * c;
                        if (disc >= 0.0) {
                            double sq = pvt_sqrt(disc);
                            double t = (-b - sq) / (2.0 * a);
                            double z = o.z + t * d.z;
                            if (z > -half && z < half && t > kEps) fold(t);
                            t = (-b + sq) / (2.0 * a);
                            z = o.z + t * d.z;
                            if (z > -half && z < half && t > kEps) fold(t);
                        }
                    }
                    if (pvt_fabs(d.z) > 1e-300) {
                        double t = (-half - o.z) / d.z;
                        double x = o.x + t * d.x, y = o.y + t * d.y;
                        if (x * x + y * y <= radius * radius && t > kEps) fold(t);
                        t = (half - o.z) / d.z;
                        x = o.x + t * d.x;
                        y = o.y + t * d.y;
                        if (x * x + y * y <= radius * radius && t > kEps) fold(t);
                    }
                }
                    // The container is the nearest node the ray starts inside of: crossed exactly once for the
                    // reference's convex shapes (:696-706); a triangle mesh may be non-convex, so it holds the
                    // ray when it is crossed an ODD number of times (extension; Mesh.contains semantics,
                    // pvtrace/geometry/mesh.py:29-32)
                    const bool holds = (MESH && gt == PVT_GEOM_MESH) ? (nl & 1) != 0 : nl == 1;
                    if (holds && !root_known) {
                        if (tfirst < cbest) {
                            if constexpr (MESH) { c2best = cbest; c2node = cnode; }
                            cbest = tfirst; cnode = node;
                        } else if (MESH && tfirst < c2best) { c2best = tfirst; c2node = node; }
                    }
                }

                PVT_MARK(1);  // node loop
                if (nhits == 0) {
                    terminal = true;  // nothing ahead: the ray vanishes silently (:681-682)
                } else {
                    hit = n1;
                    t0 = t1;
                    if (nhits == 1) { container = hit; adjacent = -1; }
                    else {
                        container = (cnode >= 0) ? cnode : hit;
                        adjacent = (container == hit) ? n2 : hit;
                        if constexpr (MESH) {
                            // leaving a mesh: the second-nearest crossing may be the same (non-convex) mesh
                            // again; what lies beyond the surface is the next node that holds the ray
                            if (container == hit && c2node >= 0 && node_geom(node_bits(hit)) == PVT_GEOM_MESH) adjacent = c2node;
                        }
                    }
                    ev_container = container;
                    if (count > k_maxsteps) {  // (:716-723)
                        ev_kind = PVT_EV_KILL;
                        terminal = true;
                        t_sel = PVT_REC_KILLED; t_node = container;
                    } else {
                        pend = true;
                    }
                }
            }
        }
        // ---- container properties + volume absorption coefficient (:746-760) -----------------------
        // (node and component records are per-lane LDS reads)
        // alpha = sum of the components' coefficients; the running partial sums ARE the cumulative
        // thresholds the reference recomputes when it picks the absorbing component (:768-781): the first
        // one is kept (it decides for containers of two components; more are re-walked when a lane needs it)
        double alpha = 0.0, pre0 = 0.0, n_container = 1.0;
        int cbase = 0, ccount = 0, crec = 0;   // first component id, count, first component record (identical components share records)
        if (pend) {
            n_container = T.dv(container * ND + ND_N);
            cbase = T.iv(container * NI + NI_CSTART); ccount = T.iv(container * NI + NI_CCOUNT);
            if (uf(UF_BY_NODE)) crec = cbase;   // (scenes of few nodes keep one record per component id)
            else crec = T.iv(container * NI + NI_CREC);
            // (tail function: a wave alone on its SIMD waits out every LDS round trip in full, and the records the rest of the
            // step needs -- the hit node's, the far side's refractive index, the pair's Fresnel constants -- are all
            // addressed by what the node loop has just found: read together HERE they cost one wait instead of five spread
            // over the step; the same words, so the same bits)
            if constexpr (TAIL && PVT_TAIL_HOIST) {
                const int hn = hit * ND;
                hr_t = V3{T.dv(hn + ND_T), T.dv(hn + ND_T + 1), T.dv(hn + ND_T + 2)};
                hr_g = V3{T.dv(hn + ND_PARAMS), T.dv(hn + ND_PARAMS + 1), T.dv(hn + ND_PARAMS + 2)};
                hr_bits = pvt_d2u(T.dv(hn + ND_BITS));
                hr_surf = T.iv(hit * NI + NI_SURF);
                if (adjacent >= 0) {
                    int kc = container, ka = adjacent;
                    if (!uf(UF_BY_NODE)) { kc = T.iv(container * NI + NI_NCLS); ka = T.iv(adjacent * NI + NI_NCLS); }
                    hr_n2 = T.dv(adjacent * ND + ND_N);
                    hr_rn2 = T.dv(L.ncls_d + ka * 2 + 1);
                    hr_cc = uf(UF_CRIT) ? T.dv(L.ccrit_d + kc * L.n_cls + ka) : __builtin_nan("");
                }
            }
            // (tail function: a photon that bounces inside one body keeps its wavelength, and with it the coefficients of
            // the step before -- the same table values, read once instead of once per bounce; not in the kernels' own loop,
            // where the seven registers cost more than the lookups: docs/history.md, round 4)
            bool known = false;
            if constexpr (TAIL && PVT_TAIL_ALPHA == 1) {
                ac_node = (int)(unsigned int)ac_lds[0]; ac_wl = ac_lds[64];
                ac_alpha = pvt_u2d(ac_lds[128]); ac_pre0 = pvt_u2d(ac_lds[192]);
            }
            if constexpr (TAIL && PVT_TAIL_ALPHA) known = container == ac_node && pvt_d2u(wl) == ac_wl;
            if (hit != k_root && !known) {
                for (int k = 0; k < ccount; k++) {
                    const int ci = L.comp_i + (crec + k) * CI, cd = L.comp_d + (crec + k) * CD;
                    alpha += interp_clamped<TAB_LDS>(T, wl, T.iv(ci + CI_ABS_X), T.iv(ci + CI_ABS_Y), T.iv(ci + CI_ABS_N),
                                                            T.iv(ci + CI_ABS_G), T.dv(cd + CD_ABS_SCALE), T.iv(ci + CI_ABS_HIST),
                                                            T.dv(cd + CD_ABS_RCP), T.dv(cd + CD_ABS_W));
                    if (k == 0) pre0 = alpha;
                }
                if constexpr (TAIL && PVT_TAIL_ALPHA == 1) {
                    ac_lds[0] = (unsigned long long)(unsigned int)container; ac_lds[64] = pvt_d2u(wl);
                    ac_lds[128] = pvt_d2u(alpha); ac_lds[192] = pvt_d2u(pre0);
                } else if constexpr (TAIL) { ac_node = container; ac_wl = pvt_d2u(wl); ac_alpha = alpha; ac_pre0 = pre0; }
            }
            if constexpr (TAIL) {
                if (known && hit != k_root) { alpha = ac_alpha; pre0 = ac_pre0; }
            }
        }
        int comp = -1;
        if (pend) {
            // free path in the container (no draw when the photon leaves the scene or the medium is clear), then
            // ONE advance for all three outcomes: to the absorption point if that comes first, else to the surface
            double depth = INFINITY;
            if (hit != k_root && alpha > kAlphaZero) depth = div_normal(-pvt_log(1.0 - rng_uniform(rng)), alpha);
            {
                const double adv = __builtin_fmin(depth, t0);   // (depth == t0 is a surface event: same value)
                pos.x = pos.x + dir.x * adv; pos.y = pos.y + dir.y * adv; pos.z = pos.z + dir.z * adv;
                travelled += adv;
                duration += div_known(adv * n_container, kCcm, kRcpCcm);
            }
            if (hit == k_root) {  // leaves the scene (:728-744)
                ev_kind = PVT_EV_EXIT; ev_hit = hit; ev_adjacent = adjacent;
                terminal = true;
                t_sel = PVT_REC_EXIT; t_node = hit; t_normal = true;
                cls = CLS_EXIT;
            } else {
                if (depth < t0) {  // absorbed (:762-832)
                    const double target = rng_uniform(rng) * alpha;
                    comp = cbase;
                    if (ccount <= 2) {
                        // (the reference walks the cumulative coefficients; with two components the first
                        // partial sum decides, and the last component takes what is left, :768-781)
                        comp = (ccount == 2 && !(target <= pre0)) ? cbase + 1 : cbase;
                    } else {
                        double running = 0.0;
                        for (int k = 0; k < ccount; k++) {
                            const int ci = L.comp_i + (crec + k) * CI, cd = L.comp_d + (crec + k) * CD;
                            running += interp_clamped<TAB_LDS>(T, wl, T.iv(ci + CI_ABS_X), T.iv(ci + CI_ABS_Y), T.iv(ci + CI_ABS_N),
                                                                      T.iv(ci + CI_ABS_G), T.dv(cd + CD_ABS_SCALE), T.iv(ci + CI_ABS_HIST),
                                                                      T.dv(cd + CD_ABS_RCP), T.dv(cd + CD_ABS_W));
                            if (target <= running) { comp = cbase + k; break; }
                        }
                    }
                    log_row<RECORD, TAIL>(A, rec_slot, nev, PVT_EV_ABSORB, -1, container, -1, comp, source, pos,
                                    dir, false, pos, wl, travelled, duration);
                    ev_component = comp;
                    cls = CLS_ABS;
                } else {
                    // ---- surface interaction (:834-895) ---------
                    ev_hit = hit;
                    if (adjacent < 0) {  // malformed scene (:840-845)
                        ev_kind = PVT_EV_KILL;
                        terminal = true;
                    } else {
                        ev_adjacent = adjacent;
                        t_node = hit; t_normal = true; ev_normal = true;
                        cls = CLS_SURF;
                    }
                }
            }
        }
        // ---- the absorbing component decides (:783-832): one pass per distinct component of the wave,
        // its record in SGPRs
        {
            const int cu = comp;                       // the component's id (what the event and `source` name) ...
            const bool mine = cls == CLS_ABS;
            const int cr = mine ? crec + (comp - cbase) : 0;   // ... and its record
            const int ci = L.comp_i + cr * CI, cd = L.comp_d + cr * CD;
            const int ctype = T.iv(ci + CI_TYPE);
            if (mine) {
                bool radiative = false;
                if (ctype == PVT_COMP_SCATTERER || ctype == PVT_COMP_LUMINOPHORE)
                    radiative = rng_uniform(rng) < T.dv(cd + CD_QY);
                double tau = 0.0;
                if (radiative) {
                    // phase function (_kernel.pyx:455-476): the draws, in the reference's order, give the
                    // cosine (or sine) of the polar angle; its partner is the composition pvt_sqrt1m2; the
                    // azimuth is 2 pi `em_turn` (pvt_sincos2pi); the new direction is written here, before the
                    // event is logged at the end of the step
                    double em_s, em_c, em_turn;
                    const int pt = T.iv(ci + CI_PHASE);
                    const double pp = T.dv(cd + CD_PHASE);
                    if (pt == PVT_PHASE_HG && pvt_fabs(pp) >= kEps) {
                        double g1 = rng_uniform(rng);
                        double sg = 2.0 * g1 - 1.0;
                        double q = (1.0 - pp * pp) / (1.0 + pp * sg);
                        em_c = 1.0 / (2.0 * pp) * (1.0 + pp * pp - q * q);
                        em_turn = rng_uniform(rng);
                        em_s = sqrt1m2_normal(em_c);
                    } else if (pt == PVT_PHASE_CONE) {
                        double g1 = rng_uniform(rng), g2 = rng_uniform(rng);
                        em_s = pvt_sqrt(g1) * pvt_sin(pp);
                        em_turn = g2;
                        em_c = sqrt1m2_normal(em_s);
                    } else {
                        double g1 = rng_uniform(rng), g2 = rng_uniform(rng);
                        em_turn = g1;
                        em_c = 2.0 * g2 - 1.0;
                        em_s = sqrt1m2_normal(em_c);
                    }
                    {
                        double sp, cp;
                        pvt_sincos2pi(em_turn, &sp, &cp);
                        dir = V3{em_s * cp, em_s * sp, em_c};
                    }
                    em = true;
                    source = cu;
                    if (ctype == PVT_COMP_LUMINOPHORE) {
                        const int ex = T.iv(ci + CI_EMS_X), ec = T.iv(ci + CI_EMS_CDF), en = T.iv(ci + CI_EMS_N);
                        const int eh = T.iv(ci + CI_EMS_HIST);
                        const double ew = T.dv(cd + CD_EMS_W);
                        double p1;
                        if (uf(UF_EMIT_FULL)) {
                            p1 = 0.0;
                        } else {
                            double e_nm = wl;
                            if (uf(UF_EMIT_KT)) {
                                const double kb_ev = 1.380649e-23 / 1.60217662e-19;
                                double e_ev = div_normal(1240.0, e_nm) + 1.5 * kb_ev * 300.0;
                                e_nm = div_normal(1240.0, e_ev);
                            }
                            p1 = interp_clamped<TAB_LDS>(T, e_nm, ex, ec, en, T.iv(ci + CI_EMS_GX), T.dv(cd + CD_EMS_SCALE_X),
                                                                              eh, T.dv(cd + CD_EMS_RCP_X), ew);
                        }
                        double gamma = p1 + (1.0 - p1) * rng_uniform(rng);
                        wl = interp_clamped<TAB_LDS>(T, gamma, ec, ex, en, T.iv(ci + CI_EMS_GC), T.dv(cd + CD_EMS_SCALE_C), eh,
                                                                    T.dv(cd + CD_EMS_RCP_C), __builtin_nan(""), eh ? __builtin_nan("") : ew);
                        tau = T.dv(cd + CD_TAU_RAD);
                        ev_kind = PVT_EV_EMIT;
                    } else {
                        ev_kind = PVT_EV_SCATTER;
                    }
                } else {
                    tau = T.dv(cd + CD_TAU_NR);
                    if (ctype == PVT_COMP_REACTOR) { ev_kind = PVT_EV_REACT; t_sel = PVT_REC_REACTED; }
                    else { ev_kind = PVT_EV_NONRADIATIVE; t_sel = PVT_REC_LOST; }
                    t_node = container;
                    terminal = true;
                }
                // radiative / non-radiative lifetime: the last draw of either branch
                if (tau > 0.0) duration += -pvt_log(1.0 - rng_uniform(rng)) * tau;
            }
        }

        PVT_MARK(2);  // classification + absorption + emission draws
        PVT_COUNT(1, alive && ev_component >= 0);            // absorbed
        PVT_COUNT(2, em);                                    // re-emitted
        PVT_COUNT(3, alive && t_normal && ev_kind != PVT_EV_EXIT);   // surface
        PVT_COUNT(4, alive && ev_kind == PVT_EV_EXIT);       // exit
        PVT_COUNT(5, alive && t_sel >= 0);                   // has a tally selector already (terminal kinds)
        PVT_COUNT(6, alive && terminal);
        // ---- local point + outward world normal of the node the event refers to.
        // Shared by EXIT and surface events (re-converged: one copy of the code).
        // Both are pure functions of (t_node, pos, tri1), so the coating / Lambertian code and the
        // x,y,z histogram axes RECOMPUTE them where needed (same arithmetic, same bits) instead of
        // keeping 12 VGPRs alive across the transcendental sites, the register-pressure peak.
        auto local_point = [&]() -> V3 {
            if constexpr (kHoist) {
                if (t_normal) {   // (an event with a normal refers to the node that was hit: its record is in registers)
                    if (node_ident(hr_bits)) return V3{pos.x + hr_t.x, pos.y + hr_t.y, pos.z + hr_t.z};
                    const int m = node_rot(hr_bits) + RT_W2L;
                    return V3{T.dv(m + 0) * pos.x + T.dv(m + 1) * pos.y + T.dv(m + 2) * pos.z + hr_t.x,
                              T.dv(m + 3) * pos.x + T.dv(m + 4) * pos.y + T.dv(m + 5) * pos.z + hr_t.y,
                              T.dv(m + 6) * pos.x + T.dv(m + 7) * pos.y + T.dv(m + 8) * pos.z + hr_t.z};
                }
            }
            const int tr = t_node * ND + ND_T;
            const unsigned long long tb = node_bits(t_node);
            if (node_ident(tb))   // unrotated node: translate only
                return V3{pos.x + T.dv(tr), pos.y + T.dv(tr + 1), pos.z + T.dv(tr + 2)};
            const int m = node_rot(tb) + RT_W2L;
            return V3{T.dv(m + 0) * pos.x + T.dv(m + 1) * pos.y + T.dv(m + 2) * pos.z + T.dv(tr),
                      T.dv(m + 3) * pos.x + T.dv(m + 4) * pos.y + T.dv(m + 5) * pos.z + T.dv(tr + 1),
                      T.dv(m + 6) * pos.x + T.dv(m + 7) * pos.y + T.dv(m + 8) * pos.z + T.dv(tr + 2)};
        };
        auto local_normal = [&](const V3& lp) -> V3 {   // outward normal (_kernel.pyx:359-400)
            const int gp = t_node * ND + ND_PARAMS;
            // (every caller stands at a surface event: the node is the one that was hit -- tail function: its record is in registers)
            const int gt = kHoist ? node_geom(hr_bits) : node_geom(node_bits(t_node));
            auto param = [&](int a) -> double {
                if constexpr (kHoist) return a == 0 ? hr_g.x : (a == 1 ? hr_g.y : hr_g.z);
                return T.dv(gp + a);
            };
            if (MESH && gt == PVT_GEOM_MESH) {   // face normal of the crossed triangle (geometry/mesh.py:63-86)
                const pvt::MeshTri* tr = A.tris + tri1;
                return V3{tr->n[0], tr->n[1], tr->n[2]};
            }
            if (gt == PVT_GEOM_BOX) {
                // The reference scans -x, +x, -y, +y, -z, +z for the face nearest to the point, a later face
                // winning only when strictly nearer (_kernel.pyx:359-377).  Per axis that is the smaller of
                // |p + h| and |p - h| (the + face only when strictly smaller), and across axes the first
                // smallest: the same six distances and the same comparisons, without the running triple.
                const double hx = 0.5 * param(0), hy = 0.5 * param(1), hz = 0.5 * param(2);
                const double mx = pvt_fabs(lp.x - (-1.0) * hx), px = pvt_fabs(lp.x - hx);
                const double my = pvt_fabs(lp.y - (-1.0) * hy), py = pvt_fabs(lp.y - hy);
                const double mz = pvt_fabs(lp.z - (-1.0) * hz), pz = pvt_fabs(lp.z - hz);
                const double cx = __builtin_fmin(px, mx), cy = __builtin_fmin(py, my), cz = __builtin_fmin(pz, mz);   // (no NaNs here)
                const bool use_y = cy < cx;
                const double cxy = __builtin_fmin(cy, cx);
                const bool use_z = cz < cxy;
                const double sx = px < mx ? 1.0 : -1.0, sy = py < my ? 1.0 : -1.0, sz = pz < mz ? 1.0 : -1.0;
                return V3{(!use_y && !use_z) ? sx : 0.0, (use_y && !use_z) ? sy : 0.0, use_z ? sz : 0.0};
            }
            if (gt == PVT_GEOM_SPHERE) {
                double mag = pvt_sqrt(dot3(lp, lp));
                return V3{lp.x / mag, lp.y / mag, lp.z / mag};
            }
            double half = 0.5 * param(0);
            double tol = 1e-8 + 1e-5 * pvt_fabs(half);
            if (pvt_fabs(lp.z + half) <= tol) return V3{0.0, 0.0, -1.0};
            if (pvt_fabs(lp.z - half) <= tol) return V3{0.0, 0.0, 1.0};
            double r = pvt_sqrt(lp.x * lp.x + lp.y * lp.y);
            return V3{lp.x / r, lp.y / r, 0.0};
        };
        if (alive && t_normal) {
            const V3 nloc = local_normal(local_point());
            const unsigned long long tb = kHoist ? hr_bits : node_bits(t_node);
            if (node_ident(tb)) {
                nrm = nloc;
            } else {
                const int q = node_rot(tb) + RT_L2W;
                nrm.x = T.dv(q + 0) * nloc.x + T.dv(q + 1) * nloc.y + T.dv(q + 2) * nloc.z;
                nrm.y = T.dv(q + 3) * nloc.x + T.dv(q + 4) * nloc.y + T.dv(q + 5) * nloc.z;
                nrm.z = T.dv(q + 6) * nloc.x + T.dv(q + 7) * nloc.y + T.dv(q + 8) * nloc.z;
            }
        }

        PVT_MARK(3);  // frame + normal
        // ---- incidence geometry -----------------------------------------------------
        // Fresnel's formulas take the cosine of the incidence angle (the dot product itself) and its sine (the
        // composition pvt_sqrt1m2); the comparison with the critical angle is a comparison of cosines (host-
        // proven threshold); the ANGLE is only what recorders accumulate, so its acos is taken when a first
        // crossing is tallied (a queue of 64 per wave, below).
        const bool surf = alive && t_normal && ev_kind != PVT_EV_EXIT;
        bool flip = false;       // the normal is used flipped along the ray (nf = -nrm)
        double ddot = 0.0;       // nf . dir, before the clamp
        double ac_arg = 1.0;
        if (alive && t_normal) {
            if (ev_kind == PVT_EV_EXIT) {
                ac_arg = __builtin_fmin(pvt_fabs(dot3(nrm, dir)), 1.0);
            } else {
                ddot = dot3(nrm, dir);
                if (ddot < 0.0) {   // flip the normal along the ray; its dot product is then the negation, bit for bit
                    flip = true;
                    ddot = -ddot;
                }
                ac_arg = __builtin_fmin(ddot, 1.0);   // (non-negative after the flip; the reference also clamps at -1)
            }
        }
        if (alive && t_normal) t_cos = ac_arg;
        const bool fres = surf && (kHoist ? hr_surf : T.iv(ev_hit * NI + NI_SURF)) == PVT_SURF_FRESNEL;
        const double c1 = ac_arg, s1 = fres ? sqrt1m2_normal(ac_arg) : 0.0;   // cos / sin of the incidence angle

        PVT_MARK(4);  // cosines of the incidence / exit angle
        if (surf) {
            // ---- Fresnel / coating decision at the surface (:865-895) ------------
            const int hit = ev_hit, container = ev_container, adjacent = ev_adjacent;
            double r = 0.0, n1 = 0.0, n2 = 0.0, rn2 = 0.0;
            if (fres) {  // unpolarised Fresnel, 1.0 beyond the critical angle (:406-419)
                // (tail function: the container's index was read for the clock, the far side's and the pair's constants with
                // the hit node's record -- all right after the node loop)
                n1 = kHoist ? n_container : T.dv(container * ND + ND_N);
                n2 = kHoist ? hr_n2 : T.dv(adjacent * ND + ND_N);
                // what depends on the refractive indices alone is tabulated per pair of index CLASSES
                int kc = container, ka = adjacent;
                if (!kHoist && !uf(UF_BY_NODE)) { kc = T.iv(container * NI + NI_NCLS); ka = T.iv(adjacent * NI + NI_NCLS); }
                const int ncls_d = L.ncls_d, n_cls = L.n_cls;
                rn2 = kHoist ? hr_rn2 : T.dv(ncls_d + ka * 2 + 1);
                // critical angle asin(n2/n1): a function of the node pair, tabulated by the host
                // with the same pvt_asin (small scenes), else computed here
                bool tir;
                // (the reference compares acos(c1) with the critical angle; the host has turned that into a
                // comparison of c1 itself wherever it could prove the two agree for every double)
                const bool crit_tab = uf(UF_CRIT);
                const double cc = kHoist ? hr_cc : (crit_tab ? T.dv(L.ccrit_d + kc * n_cls + ka) : __builtin_nan(""));
                if (cc == cc) tir = c1 < cc;
                else {   // (rare: the threshold could not be proven, or the scene has too many indices for the tables)
                    if (kHoist && !uf(UF_BY_NODE)) { kc = T.iv(container * NI + NI_NCLS); ka = T.iv(adjacent * NI + NI_NCLS); }
                    if (crit_tab) tir = pvt_acos(c1) > T.dv(L.crit_d + kc * n_cls + ka);
                    else tir = n2 < n1 && pvt_acos(c1) > pvt_asin(div_known(n2, n1, T.dv(ncls_d + kc * 2 + 1)));
                }
                if (tir) {
                    r = 1.0;
                } else {
                    double q = div_known(n1, n2, rn2) * s1;
                    double k = sqrt_normal(1.0 - q * q);
                    double rs1 = n1 * c1 - n2 * k, rs2 = n1 * c1 + n2 * k;
                    const double as = div_normal(rs1, rs2);   // (the reference writes each quotient twice)
                    double rs = as * as;
                    double rp1 = n1 * k - n2 * c1, rp2 = n1 * k + n2 * c1;
                    const double ap = div_normal(rp1, rp2);
                    double rp = ap * ap;
                    r = 0.5 * (rs + rp);
                }
            }
            int coat = -1;
            if (uf(UF_COATED) && fres) {
                const int cs = T.iv(hit * NI + NI_KSTART), ce = cs + T.iv(hit * NI + NI_KCOUNT);
                // (only the lanes whose node carries coatings; on an unrotated node the normal in the node's frame IS the
                // world normal computed above, bit for bit -- the same local_normal, not rotated -- so it is not recomputed)
                V3 lpos{0, 0, 0}, nloc = nrm;
                if (cs < ce) {
                    lpos = local_point();
                    if (!node_ident(kHoist ? hr_bits : node_bits(hit))) nloc = local_normal(lpos);
                }
                const double nl3[3] = {nloc.x, nloc.y, nloc.z}, pl3[3] = {lpos.x, lpos.y, lpos.z};
                for (int c = cs; c < ce && coat < 0; c++) {
                    bool ok = true;
#pragma unroll
                    for (int a = 0; a < 3; a++) {
                        double f = T.dv(L.coat_d + c * KD + KD_FACET + a);
                        if (pvt_fabs(nl3[a] - f) > 1e-8 + 1e-5 * pvt_fabs(f)) ok = false;
                        if (!(pl3[a] > T.dv(L.coat_d + c * KD + KD_LO + a) && pl3[a] < T.dv(L.coat_d + c * KD + KD_HI + a))) ok = false;
                    }
                    if (ok) coat = c;
                }
                if (coat >= 0) {
                    // a coating sets the reflectivity -- except beyond the critical angle when it
                    // transmits by Fresnel refraction: no refracted ray exists there
                    double cr = T.dv(L.coat_d + coat * KD + KD_REFL);
                    if (cr >= 0.0 && !(r == 1.0 && T.iv(L.coat_i + coat * KI + KI_TMODE) != 1)) r = cr;
                }
            }
            double u = 1.0;
            if (r > 0.0) u = rng_uniform(rng);
            if (u < r) {
                bool lamb = false;
                if (coat >= 0) lamb = T.iv(L.coat_i + coat * KI + KI_RMODE) == 1;
                if (lamb) {
                    // cosine-weighted about the incoming side's normal, in the node frame
                    double side = dot3(nrm, dir) < 0.0 ? 1.0 : -1.0;
                    V3 nloc = nrm;   // (unrotated node: the world normal is the local one)
                    if (!node_ident(node_bits(hit))) nloc = local_normal(local_point());
                    V3 mm{side * nloc.x, side * nloc.y, side * nloc.z};
                    double p1 = rng_uniform(rng), p2 = rng_uniform(rng);
                    V3 sd = lambert_direction(p1, p2);
                    double sign = mm.z < 0.0 ? -1.0 : 1.0;
                    double a = -1.0 / (sign + mm.z);
                    double b = mm.x * mm.y * a;
                    V3 t1v{1.0 + sign * mm.x * mm.x * a, sign * b, -sign * mm.x};
                    V3 t2v{b, sign + mm.y * mm.y * a, -mm.y};
                    V3 dl{sd.x * t1v.x + sd.y * t2v.x + sd.z * mm.x, sd.x * t1v.y + sd.y * t2v.y + sd.z * mm.y,
                          sd.x * t1v.z + sd.y * t2v.z + sd.z * mm.z};
                    const int q = node_rot(node_bits(hit)) + RT_L2W;
                    dir.x = T.dv(q + 0) * dl.x + T.dv(q + 1) * dl.y + T.dv(q + 2) * dl.z;
                    dir.y = T.dv(q + 3) * dl.x + T.dv(q + 4) * dl.y + T.dv(q + 5) * dl.z;
                    dir.z = T.dv(q + 6) * dl.x + T.dv(q + 7) * dl.y + T.dv(q + 8) * dl.z;
                } else {  // specular (:422-433): nf is nrm flipped along dir, dd their dot product (formed above)
                    const V3 nf{flip ? -nrm.x : nrm.x, flip ? -nrm.y : nrm.y, flip ? -nrm.z : nrm.z};
                    const double dd = ddot;
                    dir = V3{dir.x - 2.0 * dd * nf.x, dir.y - 2.0 * dd * nf.y, dir.z - 2.0 * dd * nf.z};
                }
                ev_kind = PVT_EV_REFLECT;
                t_sel = (container != hit) ? PVT_REC_REFLECTED : -1;
            } else {
                bool matched = false;
                if (coat >= 0) matched = T.iv(L.coat_i + coat * KI + KI_TMODE) == 1;
                if (fres && !matched) {  // Snell, vector form (:436-446)
                    double n = div_known(n1, n2, rn2);
                    const V3 nf{flip ? -nrm.x : nrm.x, flip ? -nrm.y : nrm.y, flip ? -nrm.z : nrm.z};
                    const double dd = ddot;   // dir . nf
                    double c = sqrt_normal(1.0 - n * n * (1.0 - dd * dd));
                    double sign = dd < 0.0 ? -1.0 : 1.0;
                    double k = sign * (c - sign * n * dd);
                    dir = V3{n * dir.x + k * nf.x, n * dir.y + k * nf.y, n * dir.z + k * nf.z};
                }
                ev_kind = PVT_EV_TRANSMIT;
                t_sel = (container == hit) ? PVT_REC_ESCAPING : PVT_REC_ENTERING;
            }
            // Fused exit (tally launches of a scene that is ONE unrotated box in an empty, unobserved world; the
            // host sets the flag).  A photon that has just been sent away from the box's surface on the OUTSIDE
            // can only leave the scene: its next step would find no crossing of the (convex) box, one crossing of
            // the root, draw nothing (the world has no absorber), log nothing and fire no recorder (nobody
            // listens to the root) -- so it ends here and its lane is refilled a step earlier.  "No crossing of
            // the box" is not taken on trust: the reference's slab test would see, on the face's axis, the
            // distances (h - s*o_a) / |d_a| =: g / dn to the plane the photon stands on (a rounding error that may
            // have either sign) and a negative one to the plane behind it, so the box's far distance is at most
            // g / dn -- below kEps, hence ignored (_kernel.pyx:271-276), whenever g <= kEps/2 * dn; g is formed
            // with the very operations the next step would use (o = pos + t, h = 0.5 * size).  A photon that
            // cannot be cleared this way (grazing departures) simply takes its next step.
            if (!RECORD && !MESH && uf(UF_FUSE_EXIT) && !terminal && count < k_maxsteps &&
                (ev_kind == PVT_EV_REFLECT ? container == k_root : adjacent == k_root)) {
                const V3 lp = local_point();
                const int gp = hit * ND + ND_PARAMS;
                const double h = kHoist ? 0.5 * (pvt_fabs(nrm.x) * hr_g.x + pvt_fabs(nrm.y) * hr_g.y + pvt_fabs(nrm.z) * hr_g.z)
                                        : 0.5 * (pvt_fabs(nrm.x) * T.dv(gp) + pvt_fabs(nrm.y) * T.dv(gp + 1) + pvt_fabs(nrm.z) * T.dv(gp + 2));
                const double g = h - dot3(nrm, lp), dn = dot3(nrm, dir);
                if (dn > 0.0 && g <= (0.5 * kEps) * dn) {
                    terminal = true;
#if PVT_COUNTERS
                    c_fused += 1u;
#endif
                }
            }
        }

        PVT_MARK(5);  // fresnel / reflect / refract
        // ================= deferred event: log row + tallies ==============
        if (alive && ev_kind >= 0)
            log_row<RECORD, TAIL>(A, rec_slot, nev, ev_kind, ev_hit, ev_container, ev_adjacent, ev_component, source, pos,
                            dir, ev_normal, nrm, wl, travelled, duration);

        // Lane-parallel tally.  Each lane walks the (host-precomputed) list of recorders that
        // can fire for ITS (node, selector) — different lanes handle different recorders in
        // the same trip — reading the recorder rows from LDS with per-lane addresses and
        // adding into the workgroup accumulators with per-lane LDS atomics (hardware
        // serialises same-address lanes; no wave-uniform recorder loop, no scalar-load
        // chains, no software scan).  Recorder order per lane is ascending, as in the
        // reference's loop (_kernel.pyx:517-556).
        if (uf(UF_HAS_REC)) {
            // Facet recorders whose facets have distinct dominant axes (the usual "one recorder
            // per box face") are found in O(1): the host files each under the bin (dominant axis,
            // sign) of its facet, the lane looks up the bin of ITS normal and verifies that one
            // recorder with the full tolerance test (its first trip).  Everything else is walked.
            int cs = 0, cn = 0, rbin = -1;
            if (alive && t_sel >= 0) {
                int cb = t_node;       // the node's block of the candidate tables
                bool listened = true;
                if (!uf(UF_BY_NODE)) {   // (scenes of many nodes: only the nodes somebody listens to have one)
                    cb = T.iv(t_node * NI + NI_CAND);
                    listened = cb >= 0;
                    cb = listened ? cb : 0;
                }
                const int key = L.cand_i + (cb * 7 + t_sel) * 8;
                if (listened) {
                    cs = T.iv(key);
                    cn = T.iv(key + 1);
                }
                if (listened && t_normal) {
                    const double ax = pvt_fabs(nrm.x), ay = pvt_fabs(nrm.y), az = pvt_fabs(nrm.z);
                    int b = (ax >= ay && ax >= az) ? (nrm.x > 0.0 ? 1 : 0)
                          : (ay >= az)             ? (nrm.y > 0.0 ? 3 : 2)
                                                   : (nrm.z > 0.0 ? 5 : 4);
                    rbin = T.iv(key + 2 + b);
                }
            }
            // trip t of a lane: its bin recorder first (if any), then its list -- so lanes served by
            // the bin table and lanes served by a list share the same trips
            const int nb = rbin >= 0 ? 1 : 0, ntrips = nb + cn;
            for (int j = 0;; j++) {
                const unsigned long long trip = __ballot(j < ntrips);
                if (trip == 0ull) break;
                if (tq_n + __popcll(trip) > kTallyQ) tally_flush();   // room for one record per lane of this trip
                bool push = false;
                int push_r = 0;
                if (j < ntrips) {
                    const int entry = j < nb ? rbin : T.iv(L.cand_list + cs + j - nb);
                    const int r = entry & (kRecPlain - 1);
                    const int ri = L.rec_i + r * RI;
                    bool match = true;
                    const bool plain = (entry & kRecPlain) != 0;   // nothing to check (see kRecPlain)
                    const int smode = plain ? 0 : T.iv(ri + RI_SRC_MODE);  // source filter (extension)
                    if (smode != 0) {
                        if (smode == 1) match = source < 0;
                        else if (smode == 2) match = source >= 0;
                        else match = source == T.iv(ri + RI_SRC_ID);
                    }
                    if (!plain && match && T.iv(ri + RI_HAS_FACET) != 0) {
                        const int rd = L.rec_d + r * RD;
                        const double atol = T.dv(rd + RD_ATOL);
                        if (!t_normal) match = false;
                        else if (pvt_fabs(T.dv(rd + RD_FACET) - nrm.x) > atol) match = false;
                        else if (pvt_fabs(T.dv(rd + RD_FACET + 1) - nrm.y) > atol) match = false;
                        else if (pvt_fabs(T.dv(rd + RD_FACET + 2) - nrm.z) > atol) match = false;
                    }
                    if (match) {
                        atomicAdd(&acc_cross[r], 1ull);
                        const unsigned long long bit = 1ull << (r & 63);
                        bool first;
                        if constexpr (SEENW == 1) {
                            first = !(seen.w[0] & bit);
                            seen.w[0] |= bit;
                        } else {
                            const int w = r >> 6;
                            const unsigned long long cur = w == 0 ? seen.w[0] : w == 1 ? seen.w[1] : w == 2 ? seen.w[2] : seen.w[3];
                            first = !(cur & bit);
                            if (w == 0) seen.w[0] |= bit; else if (w == 1) seen.w[1] |= bit;
                            else if (w == 2) seen.w[2] |= bit; else seen.w[3] |= bit;
                        }
                        push = first;
                        push_r = r;
                    }
                }
                // A first crossing is parked (recorder, wavelength, cosine, times[, local position]); the queue is
                // wave-private, so plain LDS stores at ranks of a ballot are all it takes.
                const unsigned long long pm = __ballot(push);
                if (pm != 0ull) {
                    const int at = tq_n + (int)rank_in(pm);
                    if (push) {
                        tq_r[at] = push_r;
                        tq_d[at] = wl; tq_d[kTallyQ + at] = t_cos; tq_d[2 * kTallyQ + at] = duration; tq_d[3 * kTallyQ + at] = travelled;
                        if (uf(UF_TQ_POS)) {
                            const V3 lpos = local_point();   // position in the recorder node's frame
                            tq_d[4 * kTallyQ + at] = lpos.x; tq_d[5 * kTallyQ + at] = lpos.y; tq_d[6 * kTallyQ + at] = lpos.z;
                        }
                    }
                    tq_n += __popcll(pm);
                }
            }
        }

        PVT_MARK(6);  // log + tally
        if (alive && terminal) {
            if constexpr (RECORD) {
                if (rec_slot >= 0) A.log_counts[rec_slot] = nev;
            }
#if PVT_COUNTERS
            c_steps += (unsigned int)count;   // the trips this photon took, in whichever launches and lanes
#endif
            alive = false;
        }
    }

#if PVT_STATS
    if (lane == 0) {
        unsigned long long* c = reinterpret_cast<unsigned long long*>(A.cursor) + 1;
        atomicAdd(c + 0, st_iters); atomicAdd(c + 1, st_lane_steps);
        atomicAdd(c + 2, st_drain_iters); atomicAdd(c + 3, st_drain_lane_steps);
        atomicAdd(c + 4, 1ull);
        for (int k = 0; k < 7; k++) atomicAdd(c + 8 + k, st_t[k]);
        atomicAdd(c + 15, (ws & WS_SOLO) ? 1ull : 0ull);
        for (int k = 0; k < 8; k++) atomicAdd(c + 16 + k, st_c[k]);
        for (int k = 0; k < 4; k++) atomicAdd(c + 24 + k, st_g[k]);
    }
#endif
#if PVT_TIMELINE
    if (!TAIL && A.timeline && lane == 0) {
        unsigned long long* o = A.timeline + ((unsigned long long)blockIdx.x * kWaves + (threadIdx.x >> 6)) * 8;
        o[0] = tl_t[0]; o[1] = tl_t[1]; o[2] = tl_t[2]; o[3] = tl_t[3]; o[4] = wall_clock64(); o[5] = tl_iters;
        // (word 7: valid | the wave ended as the last of its workgroup << 1 | HW_ID register << 32: wave slot, SIMD, CU, SE)
        o[6] = __builtin_readcyclecounter() - tl_c0;
        // (the XCC_ID register, hwreg 20, says which of the eight dies: bits 2-5 of this word)
        o[7] = 1ull | ((ws & WS_SOLO) || tail_n > 0 ? 2ull : 0ull) | ((unsigned long long)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u) << 2) |
               ((unsigned long long)__builtin_amdgcn_s_getreg(4 | (31 << 11)) << 32);
    }
#endif
    tally_flush();   // the first crossings still parked
    {   // this wave's share of the step counters, into the workgroup's sums
        unsigned long long* const cnt = reinterpret_cast<unsigned long long*>(ctl + CTL_COUNT);
        if (lane == 0) atomicAdd(cnt, (unsigned long long)c_iters);
        atomicAdd(cnt + 1, (unsigned long long)c_steps);
        atomicAdd(cnt + 2, (unsigned long long)c_fused);
    }
#if PVT_TIMELINE
    if constexpr (TAIL) {   // (the caller has written the wave's record: word 1 becomes when this function was entered, 4 when it
                            // returns, 5 gets its trips << 32)
        if (A.timeline && lane == 0) {
            unsigned long long* o = A.timeline + ((unsigned long long)blockIdx.x * kWaves + (threadIdx.x >> 6)) * 8;
            o[1] = tl_t[0]; o[4] = wall_clock64(); o[5] |= tl_iters << 32;
        }
    }
#endif
    if constexpr (TAIL) return;   // (the caller leaves the workgroup)
    // The rest of its photons' histories, when this wave was the last of a draining workgroup (see the drain), then the
    // workgroup's epilogue -- both as FUNCTIONS, called here at the very end: nothing of the loop above is live across either
    // call, so the loop's register allocation does not know of them (a value that were live across a call would be spilled
    // where it is defined and reloaded where it is used, the loop included), and what the epilogue needs -- the
    // accumulators' places in LDS, the tally set, the output pointers -- is worked out inside it instead of being held
    // (or spilled) for the kernel's whole life.
    {
        const __attribute__((address_space(4))) KArgs* ak =
            (const __attribute__((address_space(4))) KArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        if constexpr (kTailCall) {
            if (tail_n > 0) {
                // (behind an opaque copy: every caller passes the same address expression, which interprocedural constant
                // propagation would otherwise put back into the callee -- table lookups and all)
                unsigned int lds_at = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) double*)smem_of_kernel;
                asm volatile("" : "+s"(lds_at));
                tail_run<RECORD, TAB_LDS, SEENW, MESH, GRID>((const KArgs*)ak, tail_n, lds_at);
            }
        }
        leave_workgroup<TAB_LDS>((const KArgs*)ak);
    }

}

// The last wave of a draining workgroup finishes its photons here (see the drain in trace_body): the step loop alone, as a
// function -- the kernels' own loop keeps its registers whatever this one needs, and what is specific to a wave that runs
// alone on its SIMD (every step is latency, nothing overlaps) can be done here without a price on the bulk.
template <bool RECORD, int TAB_LDS, int SEENW, bool MESH, bool GRID>
__device__ __attribute__((noinline)) void tail_run(const KArgs* kernel_args, int total, unsigned int lds) {
    // The kernel's arguments, read where the kernel itself reads them.  The pointer arrives in vector registers: it is made
    // a scalar again (readfirstlane) and a pointer into the constant address space, so that the fields come through the
    // scalar cache and what is decided by them stays wave-uniform.  (The kernel-argument intrinsic itself returns garbage
    // inside a called function -- build/probe, ROCm 7.0 -- hence the explicit argument.)
    const unsigned long long bits = (unsigned long long)kernel_args;
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)bits), hi = __builtin_amdgcn_readfirstlane((unsigned int)(bits >> 32));
    const __attribute__((address_space(4))) KArgs* ak = (const __attribute__((address_space(4))) KArgs*)(((unsigned long long)hi << 32) | lo);
    trace_body<RECORD, TAB_LDS, SEENW, false, MESH, GRID, true>(*(const KArgs*)ak, __builtin_amdgcn_readfirstlane(total),
                                                                __builtin_amdgcn_readfirstlane(lds));
}

// Entry points.  Every variant runs four waves per SIMD.  Scenes of analytic shapes, tally launches and
// history-keeping ones alike, need 108-118 registers, none spilled (the build switches machine-level loop-invariant
// code motion off, which had kept every constant of the loop in a register of its own -- with it the history variants
// needed 144 registers and ran three waves; at four they are 3-10 % faster).  Mesh variants would take 144-154
// registers: held to 128 they park about sixty values in scratch around the BVH walk (photon state the walk does not
// touch), and the fourth wave is worth more than that costs -- the walk is a chain of dependent loads: +8 ... +19 % in
// the pipelined stream, 3.1 -> 2.7 ms for a single 10^6-photon launch on the 327 680-face ball (five waves: worse).
template <bool RECORD, int TAB_LDS, int SEENW, bool EMIT, bool MESH>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(kMeshWaves, kMeshWaves))) trace_kernel(KArgs A) {
    trace_body<RECORD, TAB_LDS, SEENW, EMIT, MESH>(A);
}
template <bool RECORD, int TAB_LDS, int SEENW, bool EMIT>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) trace_kernel_w4(KArgs A) {
    trace_body<RECORD, TAB_LDS, SEENW, EMIT, false>(A);
}
template <bool RECORD, int SEENW, bool EMIT>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) trace_kernel_grid(KArgs A) {
    trace_body<RECORD, 1, SEENW, EMIT, false, true>(A);
}

}  // namespace
