// pvt_trace.hip — the photon loop of pvtrace on CDNA4 (gfx950), and the C ABI
// declared in include/pvtrace_hip.h.
//
// What it replaces: reference pvtrace/engine/_kernel.pyx trace_one / trace_bundle
// (:603-1115), i.e. the per-photon while-alive loop behind engine.simulate().
// This is not a translation of that file: the reference walks one ray per CPU
// thread through per-thread scratch arrays; here
//
//   * one wavefront LANE owns one live photon; a persistent workgroup keeps all
//     64 lanes of each wave busy by refilling dead lanes from a global ray
//     cursor (wave-level ballot + prefix rank, one atomic per 64 rays);
//   * the scene tables (a few KB) are packed once into two blobs in HBM and
//     staged into LDS per workgroup; wave-uniform reads (the node loop, the
//     recorder loop) go through the scalar cache from the global copy, lane-
//     divergent reads (spectra binary search, per-lane node rows) hit LDS;
//   * hit classification is streaming (nearest / second nearest / container are
//     folded while the nodes are intersected) — no per-ray hit arrays;
//   * every lane's event (row to log, recorder to tally) is deferred to ONE
//     re-converged block at the end of the step instead of being emitted from
//     each divergent branch;
//   * recorders accumulate in LDS (integer + f64 atomics) and are flushed with
//     one global atomic per slot per workgroup; the statistics of first
//     crossings (angle, sums, histograms) are parked 64 per wave and computed
//     together;
//   * what the host can prove about a scene is decided once, not per photon:
//     arithmetic table indices on even grids (the abscissae are not even stored),
//     cosine thresholds for total internal reflection, the world node visited
//     lazily, photons that can only leave the scene ended where they leave the
//     last box (tally launches; each with a proof obligation checked per lane
//     where one is needed);
//   * all arithmetic is FP64 with FMA contraction off and the transcendental
//     functions of pvt_math.h, so a photon's whole history is bit-identical to
//     the CPU referee (oracle/pvt_oracle.c, math_mode 1).
//
// No MFMA: there is no dense contraction anywhere on this path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <chrono>
#include <dlfcn.h>
#include <string>
#include <thread>
#include <mutex>
#include <vector>

#include "../../include/pvtrace_hip.h"
#include "pvt_math.h"
#include "pvt_bvh.h"

#include "pvt_trace_kernel.h"

namespace {

// ============================================================== host side
thread_local std::string g_error;

int fail(int code, const std::string& msg) {
    g_error = msg;
    return code;
}
#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(PVT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

#if PVT_TIMELINE
constexpr int kTlLaunches = 48, kTlWgs = 1024;
constexpr size_t kTlBytes = (size_t)kTlLaunches * kTlWgs * kWaves * 8 * 8;
unsigned long long* g_timeline = nullptr;
int g_tl_launch = 0, g_tl_grid[kTlLaunches] = {0};
long long g_tl_n[kTlLaunches] = {0};
void timeline_dump();
#endif

template <class T>
int push(std::vector<T>& blob, const T* src, size_t n) {
    int at = (int)blob.size();
    if (n && src) blob.insert(blob.end(), src, src + n);
    return at;
}

}  // namespace

struct PvtScene {
    int device = 0;
    Lay lay{};
    EmitOff eoff{};
    int nd = 0, ni = 0;
    int nd_small = 0, ni_small = 0;   // ... of which everything but the spectra / their guide tables (the blobs' heads)
    int n_nodes = 0, root = 0, n_rec = 0, total_bins = 0, n_coat = 0, n_lights = 0;
    double* d_gd = nullptr;
    int* d_gi = nullptr;
    double* d_ed = nullptr;
    int* d_ei = nullptr;
    pvt::BvhNode* d_bvh = nullptr;      // triangle meshes: BVH nodes + gathered triangles
    pvt::MeshTri* d_tris = nullptr;
    pvt::BvhNode* d_bvh_top = nullptr;  // the top levels of the trees as a workgroup copies them to LDS (pvt_bvh.h: stage_top)
    int top_n = 0;                      // ... records of it
    int meshq = 0;                      // leaves a lane of a mesh walk notes in LDS before their triangles are tested (1 or kMeshQ)
    unsigned int* d_set_cursor = nullptr;   // kCursorSlots x kMaxSets cursors: launches with tally sets
    unsigned long long* d_counters = nullptr;   // step counters and clocks: 64 rows x 8 words (KArgs::counters, pvt_scene_counters / _clock)
    unsigned long long* d_stamp = nullptr;      // kCursorSlots x {start, end} of a stream's last launch, 100 MHz ticks (KArgs::stamp)
    unsigned int* d_cursor = nullptr;   // kCursorSlots cursors (64 B apart), one per stream: launches on
                                        // different streams may overlap, each needs its own
    std::mutex slot_mutex;              // launches on one stream are ordered and may share a cursor;
    std::vector<hipStream_t> slot_of;   // index = cursor slot owned by that stream
    // pvt_trace_device with column arrays: the kernel's 128-byte event records are staged here (one buffer per
    // stream slot, grown on demand, at most `stage_limit` bytes: larger logs are traced in several launches)
    std::vector<unsigned long long*> stage;
    std::vector<size_t> stage_bytes;
    size_t stage_limit = (size_t)1 << 30;
    // photons carried from launch to launch of a stream (PVT_FLAG_CARRY_OUT; see KArgs::carry_in): per stream slot
    // two buffers (the launch that resumes one may park into the other) and which of them holds parked photons
    struct Carry {
        unsigned long long* buf[2] = {nullptr, nullptr};
        int parity = 0;            // the NEXT launch parks into buf[parity]
        int phase = 0;             // ... and uses cursor block `phase` of the slot's three (see trace_launch)
        bool pending = false;      // buf[parity ^ 1] holds photons parked by the previous launch
        long long bound = 0;       // at most this many
        int maxsteps = 0, emit_method = 0;   // ... under these rules: the launch that resumes them must trace by the same
    };
    std::vector<Carry> carry;
    // device emission: per stream slot, [7][64] doubles per wave of the launch (KArgs::emit_pool), grown on demand
    std::vector<double*> emit_pool;
    std::vector<size_t> emit_pool_bytes;
    unsigned int carry_cap = 0;    // photons one buffer holds (a launch of more lanes than that does not park)
    int num_cu = 0;
    int last_grid = 0, last_lds = 0;
    size_t lds_limit = 0;
    int lazy_root = 0;              // 1 box / 2 sphere root that strictly contains every other node (see the kernel's node loop)
    double lazy_k = 0.0;            // sphere root: 1 / (2 radius)
    bool exit_observed = false;     // a recorder listens to (root, exit)
    bool grid = false;              // the scene has a node grid (many nodes; see plan_node_grid)
    int grid_dims[3] = {0, 0, 0};
    bool fuse_exit = false;         // see scene_create: photons leaving the only child's surface outwards are done
    bool hist_reads_position = false;   // a histogram axis is x, y or z
    bool consolidate = true;        // developer switches (environment), read once at scene creation
    double dev_blocks_per_cu = 0.0;
};

namespace {
struct LdsPlan { size_t bytes; bool tab_lds, small_lds; int bins_in_lds, xslots, tq_pos; bool ok; };
LdsPlan plan_lds(const PvtScene* s, bool record);   // (defined with the launch code)
}  // namespace

extern "C" {

int pvt_abi_version(void) { return PVT_ABI_VERSION; }
const char* pvt_last_error(void) { return g_error.c_str(); }

int pvt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// The surface branch asks "is the incidence angle beyond the critical angle?", which the reference evaluates as
// acos(c) > crit (c = the clamped cosine in [0, 1]).  pvt_acos falls as c grows, so there is a threshold c* with
// pvt_acos(c) > crit  <=>  c < c*: found here by bisection over the doubles of [0, 1] with the very pvt_acos the
// device runs, then CHECKED -- pvt_acos is accurate to under an ulp but need not be monotone to the last bit, so
// the 1024 doubles either side of the boundary are all evaluated; farther away the angle differs from crit by
// hundreds of ulps (|d acos / dc| >= 1) and the sign of the comparison cannot depend on the rounding.  NaN = no
// threshold could be proven (the kernel then evaluates the reference's expression); -inf = never total reflection.
static double cosine_threshold(double crit) {
    if (!(crit < INFINITY)) return -INFINITY;
    auto beyond = [&](double c) { return pvt_acos(c) > crit; };
    if (!beyond(0.0)) return -INFINITY;     // (not reachable for crit = asin(x) < pi/2; kept for safety)
    if (beyond(1.0)) return NAN;
    auto bits = [](double v) { uint64_t u; std::memcpy(&u, &v, 8); return u; };
    auto from = [](uint64_t u) { double v; std::memcpy(&v, &u, 8); return v; };
    uint64_t lo = bits(0.0), hi = bits(1.0);   // beyond(lo), !beyond(hi); non-negative doubles order like their bits
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (beyond(from(mid))) lo = mid; else hi = mid;
    }
    for (uint64_t k = 1; k <= 1024; k++) {
        if (lo >= k && !beyond(from(lo - k))) return NAN;
        if (hi + k <= bits(1.0) && beyond(from(hi + k))) return NAN;
    }
    return from(hi);   // the smallest cosine that is NOT beyond the critical angle
}

// The node grid of scenes with many nodes (kernel: GRID variants, the walk in the node loop).  Every node but the root
// is filed under the cells that its world-space bounding box, grown by 2m, touches; m = 1e-6 of the scene's extent, many
// orders of magnitude above the rounding of any distance the intersection arithmetic forms (1e-16 of it per
// operation).  What the kernel's early exit relies on, with that margin:
//   * a crossing the reference's arithmetic reports for a node lies inside that node's box grown by m, so some cell the
//     walk has visited by then (the walk's own rounding: 1e-13 of the extent) holds the node;
//   * a node filed under none of the cells visited so far stands clear of the photon by more than m, so a box or a
//     sphere (radius >= 1e-5 of the extent, checked here) is crossed twice or not at all -- never once.
// Returns false (no grid: the plain node loop serves the scene) for scenes it cannot vouch for: few nodes, meshes,
// non-rigid or inconsistent poses, degenerate shapes.
// Negative controls of the grid tests (tests/test_gpu_grid.py, tests/test_node_grid.py) are environment switches read
// when a scene is created, and they produce WRONG physics on purpose: whoever has one set gets told, loudly, every time.
bool dev_switch(const char* name) {
    if (!getenv(name)) return false;
    fprintf(stderr, "[pvtrace_hip] WARNING: %s is set -- the node grid of this scene is built WRONG on purpose (a test's negative "
                    "control); unset it for real work\n", name);
    return true;
}

struct NodeGrid {
    int n[3] = {1, 1, 1};
    double lo[3], hi[3], cell[3], guard = 0.0;
    int words = 1;
    bool odd = false;
    std::vector<unsigned long long> masks;
};
static bool plan_node_grid(const PvtSceneTables* t, NodeGrid* g) {
    const int N = t->n_nodes, root = t->root_id;
    int min_nodes = 8;
    if (const char* env = getenv("PVT_GRID_MIN_NODES")) min_nodes = atoi(env);
    if (getenv("PVT_NO_GRID") || N < min_nodes || N < 3) return false;
    std::vector<double> blo((size_t)N * 3), bhi((size_t)N * 3);
    double extent = 0.0;
    for (int n = 0; n < N; n++) {
        if (t->geom_type[n] == PVT_GEOM_MESH) return false;
        const double* w = t->world_to_local + n * 16;
        const double* l = t->local_to_world + n * 16;
        const double* gp = t->geom_params + n * 4;
        double h[3];
        switch (t->geom_type[n]) {
            case PVT_GEOM_BOX: h[0] = 0.5 * gp[0]; h[1] = 0.5 * gp[1]; h[2] = 0.5 * gp[2]; break;
            case PVT_GEOM_SPHERE: h[0] = h[1] = h[2] = gp[0]; break;
            default: h[0] = h[1] = gp[1]; h[2] = 0.5 * gp[0]; break;   // cylinder about z
        }
        for (int a = 0; a < 3; a++)
            if (!(std::isfinite(h[a]) && h[a] > 0.0)) return false;
        // rigid and consistent: world->local is a rotation plus a translation, local->world its inverse
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                double rr = 0.0, wl = 0.0;
                for (int k = 0; k < 3; k++) { rr += w[r * 4 + k] * w[c * 4 + k]; wl += w[r * 4 + k] * l[k * 4 + c]; }
                if (!(std::fabs(rr - (r == c ? 1.0 : 0.0)) < 1e-9) || !(std::fabs(wl - (r == c ? 1.0 : 0.0)) < 1e-9)) return false;
            }
        double back = 0.0;   // world->local of the node's own origin must be the zero vector
        for (int r = 0; r < 3; r++) {
            const double v = w[r * 4] * l[3] + w[r * 4 + 1] * l[7] + w[r * 4 + 2] * l[11] + w[r * 4 + 3];
            back = std::fmax(back, std::fabs(v));
        }
        for (int a = 0; a < 3; a++) {
            const double c = l[a * 4 + 3];
            double hw = t->geom_type[n] == PVT_GEOM_SPHERE ? h[0]
                                                           : std::fabs(l[a * 4]) * h[0] + std::fabs(l[a * 4 + 1]) * h[1] + std::fabs(l[a * 4 + 2]) * h[2];
            hw *= 1.0 + 1e-9;
            if (!std::isfinite(c) || !std::isfinite(hw)) return false;
            blo[(size_t)n * 3 + a] = c - hw; bhi[(size_t)n * 3 + a] = c + hw;
            extent = std::fmax(extent, std::fabs(c) + hw);
        }
        if (!(back <= 1e-9 * (1.0 + extent))) return false;
    }
    if (!(extent > 0.0) || !std::isfinite(extent)) return false;
    const double m = 1e-6 * extent;
    for (int n = 0; n < N; n++) {
        if (n == root || t->geom_type[n] == PVT_GEOM_BOX) continue;
        const double radius = t->geom_type[n] == PVT_GEOM_SPHERE ? t->geom_params[n * 4] : t->geom_params[n * 4 + 1];
        if (!(radius >= 1e-5 * extent)) return false;
        if (t->geom_type[n] == PVT_GEOM_CYLINDER) g->odd = true;
    }
    // (negative controls of tests/test_gpu_grid.py: file the boxes a centimetre too small / leave the walk as soon as
    // any two crossings are known -- results must then differ from the referee's)
    const double grow = dev_switch("PVT_GRID_DEV_SHRINK") ? -1.0 : 2.0 * m;
    for (int a = 0; a < 3; a++) { g->lo[a] = INFINITY; g->hi[a] = -INFINITY; }
    for (int n = 0; n < N; n++) {
        if (n == root) continue;
        for (int a = 0; a < 3; a++) {
            blo[(size_t)n * 3 + a] -= grow; bhi[(size_t)n * 3 + a] += grow;
            g->lo[a] = std::fmin(g->lo[a], blo[(size_t)n * 3 + a] - m);
            g->hi[a] = std::fmax(g->hi[a], bhi[(size_t)n * 3 + a] + m);
        }
    }
    // ---- resolution.  What a photon pays for is the nodes it tests and the cells it steps through, and both depend on
    // how the cells fall on the nodes: on an array of 6 x 6 tiles a 6 x 6 grid (one tile per cell) traces 2.26e9
    // photons/s, 8 x 8 and 15 x 15 grids 1.58e9, 11 x 11 1.93e9 (measured).  So the resolution is CHOSEN: starting from
    // about one cubic cell per node, each axis in turn tries other counts, and a candidate is priced by walking a fixed
    // set of sample rays through it -- the kernel's walk with the nodes' boxes standing in for the shapes: cells
    // visited, nodes tested, exit once two crossings lie before the end of the cells visited.  A wave waits for its
    // slowest lane, so the price is the mean over the dearest quarter of the rays.
    const int W = N > 64 ? 2 : 1;
    auto file_nodes = [&](const int (&dims)[3], double (&cell)[3], std::vector<unsigned long long>& masks) {
        for (int a = 0; a < 3; a++) cell[a] = (g->hi[a] - g->lo[a]) / dims[a];
        masks.assign((size_t)dims[0] * dims[1] * dims[2] * W, 0ull);
        for (int n = 0; n < N; n++) {
            if (n == root) continue;
            int c0[3], c1[3];
            for (int a = 0; a < 3; a++) {   // cells touched, one cell more on either side when a face lies within m of a cell wall
                c0[a] = (int)std::floor((blo[(size_t)n * 3 + a] - m - g->lo[a]) / cell[a]);
                c1[a] = (int)std::floor((bhi[(size_t)n * 3 + a] + m - g->lo[a]) / cell[a]);
                c0[a] = c0[a] < 0 ? 0 : c0[a];
                c1[a] = c1[a] > dims[a] - 1 ? dims[a] - 1 : c1[a];
            }
            for (int z = c0[2]; z <= c1[2]; z++)
                for (int y = c0[1]; y <= c1[1]; y++)
                    for (int x = c0[0]; x <= c1[0]; x++)
                        masks[(((size_t)z * dims[1] + y) * dims[0] + x) * W + (n >> 6)] |= 1ull << (n & 63);
        }
    };
    // sample rays (fixed pseudo-random sequence: the same scene always gets the same grid): from inside a node's box,
    // from a face of one, from anywhere in the grid's box; directions isotropic
    struct Ray { double o[3], d[3]; };
    std::vector<Ray> rays;
    {
        unsigned long long st = 0x9E3779B97F4A7C15ull;
        auto uni = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
        std::vector<int> others;
        for (int n = 0; n < N; n++) if (n != root) others.push_back(n);
        for (int k = 0; k < 384; k++) {
            Ray r;
            const int n = others[(size_t)(uni() * others.size()) % others.size()];
            for (int a = 0; a < 3; a++) {
                const double lo = k % 4 == 3 ? g->lo[a] : blo[(size_t)n * 3 + a], hi = k % 4 == 3 ? g->hi[a] : bhi[(size_t)n * 3 + a];
                r.o[a] = lo + uni() * (hi - lo);
            }
            if (k % 4 == 2) { const int a = (int)(uni() * 3) % 3; r.o[a] = uni() < 0.5 ? blo[(size_t)n * 3 + a] + grow : bhi[(size_t)n * 3 + a] - grow; }
            double z = 2.0 * uni() - 1.0, ph = 6.283185307179586 * uni(), s = std::sqrt(1.0 - z * z);
            r.d[0] = s * std::cos(ph); r.d[1] = s * std::sin(ph); r.d[2] = z;
            rays.push_back(r);
        }
    }
    auto price = [&](const int (&dims)[3]) -> double {
        double cell[3];
        std::vector<unsigned long long> masks;
        file_nodes(dims, cell, masks);
        std::vector<double> cost;
        for (const Ray& r : rays) {
            double t_in = 0.0, t_out = INFINITY;
            bool walk = true;
            for (int a = 0; a < 3; a++) {
                if (std::fabs(r.d[a]) < 1e-20) { if (r.o[a] < g->lo[a] || r.o[a] > g->hi[a]) walk = false; continue; }
                const double ta = (g->lo[a] - r.o[a]) / r.d[a], tb = (g->hi[a] - r.o[a]) / r.d[a];
                t_in = std::fmax(t_in, std::fmin(ta, tb)); t_out = std::fmin(t_out, std::fmax(ta, tb));
            }
            if (!(t_in <= t_out)) walk = false;
            int c[3] = {0, 0, 0};
            double tm[3] = {INFINITY, INFINITY, INFINITY};
            for (int a = 0; a < 3 && walk; a++) {
                c[a] = (int)((r.o[a] + r.d[a] * t_in - g->lo[a]) / cell[a]);
                c[a] = c[a] < 0 ? 0 : (c[a] > dims[a] - 1 ? dims[a] - 1 : c[a]);
                if (std::fabs(r.d[a]) >= 1e-20) tm[a] = (g->lo[a] + (c[a] + (r.d[a] < 0 ? 0 : 1)) * cell[a] - r.o[a]) / r.d[a];
            }
            unsigned long long seen[2] = {0ull, 0ull};
            int cells = 0, tests = 0, nh = 0;
            double t1 = INFINITY, t2 = INFINITY;
            while (walk) {
                cells += 1;
                const unsigned long long* mk = &masks[(((size_t)c[2] * dims[1] + c[1]) * dims[0] + c[0]) * W];
                for (int w = 0; w < W; w++) {
                    unsigned long long fresh = mk[w] & ~seen[w];
                    seen[w] |= mk[w];
                    while (fresh) {
                        const int n = w * 64 + __builtin_ctzll(fresh);
                        fresh &= fresh - 1;
                        tests += 1;
                        double te = -INFINITY, tx = INFINITY;   // the ray against the node's box
                        bool miss = false;
                        for (int a = 0; a < 3; a++) {
                            const double lo = blo[(size_t)n * 3 + a], hi = bhi[(size_t)n * 3 + a];
                            if (std::fabs(r.d[a]) < 1e-20) { if (r.o[a] < lo || r.o[a] > hi) miss = true; continue; }
                            const double ta = (lo - r.o[a]) / r.d[a], tb = (hi - r.o[a]) / r.d[a];
                            te = std::fmax(te, std::fmin(ta, tb)); tx = std::fmin(tx, std::fmax(ta, tb));
                        }
                        if (miss || tx < te || !(tx > 0.0)) continue;
                        const double ts[2] = {te, tx};
                        for (int q = te > 0.0 ? 0 : 1; q < 2; q++) {
                            if (ts[q] < t1) { t2 = t1; t1 = ts[q]; } else if (ts[q] < t2) t2 = ts[q];
                            nh += 1;
                        }
                    }
                }
                const double t_cell = std::fmin(tm[0], std::fmin(tm[1], tm[2]));
                const int ax = (tm[0] <= tm[1] && tm[0] <= tm[2]) ? 0 : (tm[1] <= tm[2] ? 1 : 2);
                const int nxt = c[ax] + (r.d[ax] < 0 ? -1 : 1);
                if ((nh >= 2 && t2 + m < t_cell) || !(t_cell < INFINITY) || nxt < 0 || nxt >= dims[ax]) break;
                c[ax] = nxt;
                tm[ax] += cell[ax] / std::fabs(r.d[ax]);
            }
            // a trip of the kernel's walk moves a lane on by one cell AND tests one node
            cost.push_back((double)(cells > tests ? cells : tests) + 0.25 * (cells + tests));
        }
        std::sort(cost.begin(), cost.end());
        double sum = 0.0;
        const size_t from = cost.size() - cost.size() / 4;
        for (size_t i = from; i < cost.size(); i++) sum += cost[i];
        return sum / (double)(cost.size() - from);
    };
    double ext[3], vol = 1.0;
    for (int a = 0; a < 3; a++) { ext[a] = g->hi[a] - g->lo[a]; vol *= ext[a]; }
    constexpr int kMaxCells = 512;   // 8 KB of masks in LDS at two words per cell
    {   // start: about one cell per node, as cubic as the extent allows
        double target = std::fmin((double)kMaxCells, std::fmax(8.0, 1.0 * (N - 1)));
        // (developer sweep; never more cells than the mask table's share of LDS holds)
        if (const char* env = getenv("PVT_GRID_CELLS")) target = std::fmin((double)kMaxCells, std::fmax(1.0, atof(env)));
        double side = std::cbrt(vol / target);
        for (int pass = 0; pass < 200; pass++) {
            long long cells = 1;
            for (int a = 0; a < 3; a++) {
                g->n[a] = (int)std::fmin(64.0, std::fmax(1.0, std::floor(ext[a] / side + 0.5)));
                cells *= g->n[a];
            }
            if ((double)cells <= target * 1.25 && cells <= kMaxCells) break;
            side *= 1.05;
        }
    }
    if (!getenv("PVT_GRID_CELLS") && !getenv("PVT_GRID_NO_TUNING")) {
        double best = price(g->n);
        for (int sweep = 0; sweep < 2; sweep++)
            for (int a = 0; a < 3; a++) {
                const int n0 = g->n[a];
                int pick = n0;
                for (int v = std::max(1, n0 / 2); v <= std::min(64, 2 * n0 + 1); v++) {
                    if (v == n0) continue;
                    int dims[3] = {g->n[0], g->n[1], g->n[2]};
                    dims[a] = v;
                    if ((long long)dims[0] * dims[1] * dims[2] > kMaxCells) break;
                    const double p = price(dims);
                    if (p < best * 0.98) { best = p; pick = v; }   // (a clear gain only: ties keep the coarser grid)
                }
                g->n[a] = pick;
            }
    }
    g->words = W;
    g->guard = dev_switch("PVT_GRID_DEV_GUARD") ? -1e30 : m;
    file_nodes(g->n, g->cell, g->masks);
    return true;
}

int pvt_scene_create(const PvtSceneTables* t, int device, PvtScene** out) {
    if (!t || !out) return fail(PVT_ERR_INVALID, "null argument");
    if (t->n_nodes <= 0) return fail(PVT_ERR_INVALID, "scene has no nodes");
    if (t->n_nodes > PVT_MAX_NODES) return fail(PVT_ERR_TOO_MANY_NODES, "more than 128 geometry nodes");
    if (t->n_recorders > PVT_MAX_RECORDERS) return fail(PVT_ERR_INVALID, "more than 256 recorders");
    if (pvt_device_count() <= device) return fail(PVT_ERR_NO_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));

    const int N = t->n_nodes, C = t->n_components, R = t->n_recorders, H = t->n_hists, K = t->n_coatings;
    std::vector<pvt::BvhNode> bvh_nodes;
    std::vector<pvt::MeshTri> bvh_tris;
    std::vector<int> bvh_roots;
    for (int n = 0; n < N; n++) {
        const int g = t->geom_type[n];
        if (g < PVT_GEOM_BOX || g > PVT_GEOM_MESH) return fail(PVT_ERR_INVALID, "unknown geometry type");
        if (g != PVT_GEOM_MESH) continue;
        if (!t->mesh_face_start || !t->mesh_face_count || !t->mesh_vertices || !t->mesh_faces || !t->mesh_normals)
            return fail(PVT_ERR_INVALID, "mesh node without mesh tables");
        const long long f0 = t->mesh_face_start[n], fc = t->mesh_face_count[n];
        if (fc <= 0 || f0 < 0 || f0 + fc > t->n_mesh_faces) return fail(PVT_ERR_INVALID, "mesh face range out of bounds");
        if (t->n_mesh_faces >= (1 << 27)) return fail(PVT_ERR_INVALID, "more than 2^27 mesh faces in one scene");
        for (long long k = 3 * f0; k < 3 * (f0 + fc); k++)
            if (t->mesh_faces[k] < 0 || t->mesh_faces[k] >= t->n_mesh_vertices)
                return fail(PVT_ERR_INVALID, "mesh face indexes a missing vertex");
    }
    // ---- classes: what many nodes have in common is stored once (see the enums next to struct Lay) ----
    // unrotated: the 3x3 blocks of both matrices of a node are the identity, bit for bit (+0.0 off the diagonal)
    auto unrotated = [&](int n) {
        const double one = 1.0, zero = 0.0;
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                const double* want = r == c ? &one : &zero;
                if (std::memcmp(&t->world_to_local[n * 16 + r * 4 + c], want, 8) != 0) return false;
                if (std::memcmp(&t->local_to_world[n * 16 + r * 4 + c], want, 8) != 0) return false;
            }
        return true;
    };
    // rotation classes: nodes whose two 3x3 blocks have the same bits share a record (and, in the wave-uniform
    // node loop, the local direction and its reciprocals)
    std::vector<int> rot_class(N), rot_first;
    for (int n = 0; n < N; n++) {
        int cls = -1;
        for (size_t e = 0; e < rot_first.size() && cls < 0; e++) {
            bool same = true;
            for (int r = 0; r < 3 && same; r++)
                for (int c = 0; c < 3 && same; c++)
                    same = std::memcmp(&t->world_to_local[n * 16 + r * 4 + c], &t->world_to_local[rot_first[e] * 16 + r * 4 + c], 8) == 0 &&
                           std::memcmp(&t->local_to_world[n * 16 + r * 4 + c], &t->local_to_world[rot_first[e] * 16 + r * 4 + c], 8) == 0;
            if (same) cls = (int)e;
        }
        if (cls < 0) { cls = (int)rot_first.size(); rot_first.push_back(n); }
        rot_class[n] = cls;
    }
    const int Q = (int)rot_first.size();
    // refractive-index classes (bit-identical indices)
    bool index_ok = true;   // refractive indices the known-divisor division is proven for
    std::vector<int> idx_class(N), idx_first;
    for (int n = 0; n < N; n++) {
        const double v = t->refractive_index[n];
        if (!(std::isfinite(v) && v > 1e-100 && v < 1e100)) index_ok = false;
        int cls = -1;
        for (size_t e = 0; e < idx_first.size() && cls < 0; e++)
            if (std::memcmp(&t->refractive_index[idx_first[e]], &v, 8) == 0) cls = (int)e;
        if (cls < 0) { cls = (int)idx_first.size(); idx_first.push_back(n); }
        idx_class[n] = cls;
    }
    if (!index_ok) return fail(PVT_ERR_INVALID, "refractive indices must be finite and positive");
    // scenes of few nodes: classes, component records and candidate blocks numbered like the nodes / the reference's ids
    // (Lay::by_node: the lanes index the tables without reading NI_NCLS / NI_CREC / NI_CAND first)
    const bool by_node = N <= 16;
    if (by_node) {
        idx_first.resize((size_t)N);
        for (int n = 0; n < N; n++) { idx_class[n] = n; idx_first[(size_t)n] = n; }
    }
    const int M = (int)idx_first.size();

    // Spectra, packed per DISTINCT table.  RN(1/spacing) when EVERY interval of the abscissae has the same bits and
    // the ordinates keep the quotient inside div_known's domain (no -0.0, no extreme magnitudes):
    auto even_rcp = [](const double* xs, const double* ys, int n) -> double {
        if (n < 2) return NAN;
        const double w = xs[1] - xs[0];
        if (!(w > 1e-100 && w < 1e100)) return NAN;
        for (int i = 1; i + 1 < n; i++) if (xs[i + 1] - xs[i] != w) return NAN;
        for (int i = 0; i < n; i++) {
            if (ys[i] == 0.0 && std::signbit(ys[i])) return NAN;
            if (!(std::fabs(ys[i]) < 1e100)) return NAN;
            if (i > 0 && ys[i] != ys[i - 1] && std::fabs(ys[i] - ys[i - 1]) < 1e-100) return NAN;
        }
        return 1.0 / w;
    };
    // The spacing itself when, additionally, xs[i] == xs[0] + i*w bit for bit AND the kernel's arithmetic
    // (i = int((x - xs[0]) * rcp), one repair step against the computed neighbours) provably lands on the
    // reference's bisection index for every x inside the table: the raw index is monotone in x, so it is enough
    // that every abscissa and its two neighbouring doubles come out right (checked here with the device's own
    // sequence of operations).  Such a table is stored as its first abscissa alone, without a guide table.
    auto even_w = [](const double* xs, int n, double rcp) -> double {
        if (!(rcp == rcp) || n < 2 || n > (1 << 24)) return NAN;
        const double w = xs[1] - xs[0];
        auto grid = [&](int i) { volatile double prod = (double)i * w; volatile double at = xs[0] + prod; return (double)at; };
        for (int i = 0; i < n; i++)
            if (grid(i) != xs[i]) return NAN;   // two roundings, never contracted
        auto lands = [&](double x, int want) {
            volatile double diff = x - xs[0];
            volatile double quot = diff * rcp;
            int i = (int)quot;
            i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
            double xlo = grid(i);
            if (x < xlo) { i -= 1; xlo = grid(i); }
            double xhi = grid(i + 1);
            if (!(x < xhi)) { i += 1; xlo = xhi; xhi = grid(i + 1); }
            return i == want && xlo <= x && x < xhi;
        };
        for (int i = 0; i < n; i++) {   // x0 < x < xl is all the even path ever sees
            const double below = std::nextafter(xs[i], -INFINITY), above = std::nextafter(xs[i], INFINITY);
            if (i > 0 && !lands(below, i - 1)) return NAN;
            if (i > 0 && i < n - 1 && !lands(xs[i], i)) return NAN;
            if (i < n - 1 && !lands(above, i)) return NAN;
        }
        return w;
    };
    // The reference keeps one set of tables per component of every node (compiler.py:160-215); a scene of many nodes
    // made of the same material repeats them.  Here a table that has the bits of an earlier one (abscissae, ordinates,
    // sampling mode) is that earlier one: the 121 tiles of an LSC array share ONE absorption and ONE emission table.
    auto abs_hist = [&](int c) { return t->comp_abs_hist && t->comp_abs_hist[c] ? 1 : 0; };
    auto ems_hist = [&](int c) { return t->comp_ems_hist && t->comp_ems_hist[c] ? 1 : 0; };
    auto same_abs = [&](int c, int e) {
        const int n = t->comp_abs_n[c];
        return n == t->comp_abs_n[e] && abs_hist(c) == abs_hist(e) &&
               std::memcmp(t->abs_x + t->comp_abs_start[c], t->abs_x + t->comp_abs_start[e], (size_t)n * 8) == 0 &&
               std::memcmp(t->abs_y + t->comp_abs_start[c], t->abs_y + t->comp_abs_start[e], (size_t)n * 8) == 0;
    };
    auto same_ems = [&](int c, int e) {
        const int n = t->comp_ems_n[c];
        return n == t->comp_ems_n[e] && ems_hist(c) == ems_hist(e) &&
               std::memcmp(t->ems_x + t->comp_ems_start[c], t->ems_x + t->comp_ems_start[e], (size_t)n * 8) == 0 &&
               std::memcmp(t->ems_cdf + t->comp_ems_start[c], t->ems_cdf + t->comp_ems_start[e], (size_t)n * 8) == 0;
    };
    std::vector<int> abs_of(C), ems_of(C);   // the component whose tables component c uses (itself: it owns them)
    {
        std::vector<int> abs_owners, ems_owners;
        for (int c = 0; c < C; c++) {
            abs_of[c] = ems_of[c] = c;
            for (int e : abs_owners) if (same_abs(c, e)) { abs_of[c] = e; break; }
            for (int e : ems_owners) if (same_ems(c, e)) { ems_of[c] = e; break; }
            if (abs_of[c] == c) abs_owners.push_back(c);
            if (ems_of[c] == c) ems_owners.push_back(c);
        }
    }
    // component RECORDS: the components of a node are a run of records; a node whose run has the contents of an
    // earlier node's run shares it (NI_CREC).  Component IDS (events, `source`, recorder filters) stay the reference's.
    auto same_component = [&](int c, int e) {
        return t->comp_type[c] == t->comp_type[e] && t->comp_phase_type[c] == t->comp_phase_type[e] &&
               std::memcmp(&t->comp_qy[c], &t->comp_qy[e], 8) == 0 && std::memcmp(&t->comp_tau_rad[c], &t->comp_tau_rad[e], 8) == 0 &&
               std::memcmp(&t->comp_tau_nr[c], &t->comp_tau_nr[e], 8) == 0 &&
               std::memcmp(&t->comp_phase_param[c], &t->comp_phase_param[e], 8) == 0 &&
               abs_of[c] == abs_of[e] && ems_of[c] == ems_of[e];
    };
    std::vector<int> node_crec(N, 0), rec_comp;   // rec_comp[r] = the component id whose fields record r holds
    for (int n = 0; n < N; n++) {
        const int c0 = t->comp_start[n], cc = t->comp_count[n];
        if (cc < 0 || c0 < 0 || c0 + cc > C) return fail(PVT_ERR_INVALID, "component range of a node out of bounds");
        int found = -1;
        if (by_node) found = c0;   // (one record per component id: NI_CREC == NI_CSTART, which the kernel relies on)
        for (int e = 0; e < n && found < 0 && !by_node; e++) {
            if (t->comp_count[e] != cc) continue;
            bool same = true;
            for (int k = 0; k < cc && same; k++) same = same_component(c0 + k, t->comp_start[e] + k);
            if (same) found = node_crec[e];
        }
        if (found < 0) {
            found = (int)rec_comp.size();
            for (int k = 0; k < cc; k++) rec_comp.push_back(c0 + k);
        }
        node_crec[n] = found;
    }
    if (by_node) {
        rec_comp.resize((size_t)C);
        for (int c = 0; c < C; c++) rec_comp[(size_t)c] = c;
    }
    const int CR = (int)rec_comp.size();
    // recorder candidate blocks: only for the nodes somebody listens to
    std::vector<int> node_cand(N, -1);
    int n_cand = 0;
    for (int r = 0; r < R; r++) {
        const int n = t->rec_node[r];
        if (n < 0 || n >= N) return fail(PVT_ERR_INVALID, "recorder on a missing node");
        if (!by_node && node_cand[n] < 0) node_cand[n] = n_cand++;
    }
    if (by_node) {
        for (int n = 0; n < N; n++) node_cand[n] = n;
        n_cand = N;
    }

    // fixed-stride records, then the pooled spectra
    Lay lay{};
    lay.comp_d = N * ND;
    lay.rec_d = lay.comp_d + CR * CD;
    lay.hist_d = lay.rec_d + R * RD;
    lay.coat_d = lay.hist_d + H * HD;
    // The small tables come first in the blob -- records, then critical angles, rotation classes, index classes and the
    // node grid -- and the spectra last: when a scene's spectra are too large for LDS, a workgroup still stages everything
    // before `spec_d` (KArgs::nd_lds; the guide tables are the tail of the int blob in the same way).
    const int small_d = lay.coat_d + K * KD;
    constexpr int kCritClasses = 16;
    lay.n_cls = M;
    lay.crit_d = M <= kCritClasses ? small_d : -1;
    lay.ccrit_d = lay.crit_d >= 0 ? lay.crit_d + M * M : -1;
    lay.rot_d = small_d + (lay.crit_d >= 0 ? 2 * M * M : 0);
    lay.ncls_d = lay.rot_d + Q * RT;
    lay.by_node = by_node ? 1 : 0;
    NodeGrid grid;
    const bool has_grid = plan_node_grid(t, &grid);
    lay.grid_d = has_grid ? lay.ncls_d + M * 2 : -1;
    const int spec_d = lay.ncls_d + M * 2 + (has_grid ? 14 + (int)grid.masks.size() : 0);
    std::vector<double> c_abs_rcp(C), c_abs_w(C), c_ems_rcp_x(C), c_ems_rcp_c(C), c_ems_w(C);
    std::vector<int> c_abs_x(C), c_abs_y(C), c_ems_x(C), c_ems_c(C), c_abs_g(C), c_ems_gx(C), c_ems_gc(C);
    int spec_len = 0, guide_len = 0;
    for (int c = 0; c < C; c++) {
        const double* ax = t->abs_x + t->comp_abs_start[c];
        const double* ay = t->abs_y + t->comp_abs_start[c];
        const double* ex = t->ems_x + t->comp_ems_start[c];
        const double* ec = t->ems_cdf + t->comp_ems_start[c];
        const int an = t->comp_abs_n[c], en = t->comp_ems_n[c];
        if (abs_of[c] != c) {
            const int e = abs_of[c];
            c_abs_rcp[c] = c_abs_rcp[e]; c_abs_w[c] = c_abs_w[e]; c_abs_x[c] = c_abs_x[e]; c_abs_y[c] = c_abs_y[e]; c_abs_g[c] = c_abs_g[e];
        } else {
            c_abs_rcp[c] = even_rcp(ax, ay, an);
            c_abs_w[c] = abs_hist(c) ? NAN : even_w(ax, an, c_abs_rcp[c]);
            const bool abs_compact = c_abs_w[c] == c_abs_w[c];
            c_abs_x[c] = spec_d + spec_len; spec_len += abs_compact ? (an > 0 ? 1 : 0) : an;
            c_abs_y[c] = spec_d + spec_len; spec_len += an;
            c_abs_g[c] = abs_compact ? -1 : guide_len; guide_len += abs_compact ? 0 : an;
        }
        if (ems_of[c] != c) {
            const int e = ems_of[c];
            c_ems_rcp_x[c] = c_ems_rcp_x[e]; c_ems_rcp_c[c] = c_ems_rcp_c[e]; c_ems_w[c] = c_ems_w[e];
            c_ems_x[c] = c_ems_x[e]; c_ems_c[c] = c_ems_c[e]; c_ems_gx[c] = c_ems_gx[e]; c_ems_gc[c] = c_ems_gc[e];
        } else {
            c_ems_rcp_x[c] = even_rcp(ex, ec, en);
            c_ems_rcp_c[c] = even_rcp(ec, ex, en);
            c_ems_w[c] = ems_hist(c) ? NAN : even_w(ex, en, c_ems_rcp_x[c]);
            const bool ems_compact = c_ems_w[c] == c_ems_w[c];
            c_ems_x[c] = spec_d + spec_len; spec_len += ems_compact ? (en > 0 ? 1 : 0) : en;
            c_ems_c[c] = spec_d + spec_len; spec_len += en;
            c_ems_gx[c] = ems_compact ? -1 : guide_len; guide_len += ems_compact ? 0 : en;
            c_ems_gc[c] = guide_len; guide_len += en;
        }
    }
    const int spec_end = spec_d + spec_len;
    std::vector<double> gd((size_t)spec_end + 1, 0.0);
    if (has_grid) {
        double* d = gd.data() + lay.grid_d;
        for (int a = 0; a < 3; a++) { d[a] = grid.lo[a]; d[3 + a] = grid.hi[a]; d[6 + a] = grid.cell[a]; d[9 + a] = 1.0 / grid.cell[a]; }
        d[12] = grid.guard;
        const unsigned long long bits = (unsigned long long)grid.n[0] | ((unsigned long long)grid.n[1] << 8) | ((unsigned long long)grid.n[2] << 16) |
                                        ((unsigned long long)grid.words << 24) | ((unsigned long long)(grid.odd ? 1 : 0) << 28);
        std::memcpy(&d[13], &bits, 8);
        std::memcpy(&d[14], grid.masks.data(), grid.masks.size() * 8);
    }
    if (lay.crit_d >= 0)
        for (int c = 0; c < M; c++)
            for (int a = 0; a < M; a++) {
                const double n1 = t->refractive_index[idx_first[c]], n2 = t->refractive_index[idx_first[a]];
                const double crit = n2 < n1 ? pvt_asin(n2 / n1) : INFINITY;   // same pvt_asin as the device
                gd[lay.crit_d + c * M + a] = crit;
                gd[lay.ccrit_d + c * M + a] = cosine_threshold(crit);
            }
    for (int q = 0; q < Q; q++) {
        double* d = gd.data() + lay.rot_d + q * RT;
        const int n = rot_first[q];
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
                d[RT_W2L + r * 3 + c] = t->world_to_local[n * 16 + r * 4 + c];
                d[RT_L2W + r * 3 + c] = t->local_to_world[n * 16 + r * 4 + c];
            }
    }
    for (int m = 0; m < M; m++) {
        gd[lay.ncls_d + m * 2] = t->refractive_index[idx_first[m]];
        gd[lay.ncls_d + m * 2 + 1] = 1.0 / t->refractive_index[idx_first[m]];
    }
    lay.comp_i = N * NI;
    lay.rec_i = lay.comp_i + CR * CI;
    lay.hist_i = lay.rec_i + R * RI;
    lay.coat_i = lay.hist_i + H * HI;
    lay.cand_i = lay.coat_i + K * KI;
    lay.cand_list = lay.cand_i + n_cand * 7 * 8;
    const int guide0 = lay.cand_list + R;  // guide tables: one entry per table point, per searched array
    std::vector<int> gi((size_t)guide0 + (size_t)guide_len + 1, 0);
    // guide[b] = largest i <= n-2 with xs[i] <= xs[0] + b*(xs[n-1]-xs[0])/(n-1), b = 0..n-1
    auto build_guide = [&](const double* xs, int n, int at, double* scale) {
        *scale = 0.0;
        if (n < 2 || !(xs[n - 1] > xs[0])) return;
        const int Kb = n - 1;
        *scale = (double)Kb / (xs[n - 1] - xs[0]);
        int i = 0;
        for (int b = 0; b <= Kb; b++) {
            const double edge = xs[0] + (double)b * ((xs[n - 1] - xs[0]) / (double)Kb);
            while (i + 1 <= n - 2 && xs[i + 1] <= edge) i++;
            gi[at + b] = i;
        }
    };
    {   // recorders grouped by the (node, selector) they listen to.  A facet recorder whose facet
        // has a clearly dominant component, alone in its (axis, sign) bin, goes to the bin table;
        // the rest (no facet, oblique facets, bin collisions) to the walked list, ascending id.
        int at = 0;
        for (int node_of_key = 0; node_of_key < N; node_of_key++) {
          if (node_cand[node_of_key] < 0) continue;
          for (int sel = 0; sel < 7; sel++) {
            const int key = node_of_key * 7 + sel;
            int* rec = gi.data() + lay.cand_i + (node_cand[node_of_key] * 7 + sel) * 8;
            rec[0] = at;
            int owner[6] = {-1, -1, -1, -1, -1, -1};
            bool clash[6] = {false, false, false, false, false, false};
            auto bin_of = [&](int r) -> int {
                if (!t->rec_has_facet[r]) return -1;
                const double* f = t->rec_facet + r * 3;
                const double a[3] = {std::fabs(f[0]), std::fabs(f[1]), std::fabs(f[2])};
                int k = (a[0] >= a[1] && a[0] >= a[2]) ? 0 : (a[1] >= a[2] ? 1 : 2);
                const double other = std::fmax(a[(k + 1) % 3], a[(k + 2) % 3]);
                // any normal within atol of the facet must have the same dominant axis and sign
                if (!(a[k] - other > 4.0 * t->rec_atol[r] + 1e-9) || !(a[k] > 2.0 * t->rec_atol[r])) return -1;
                return k * 2 + (f[k] > 0.0 ? 1 : 0);
            };
            for (int r = 0; r < R; r++) {
                if (t->rec_node[r] * 7 + t->rec_event[r] != key) continue;
                int b = bin_of(r);
                if (b >= 0) { if (owner[b] >= 0) clash[b] = true; else owner[b] = r; }
            }
            // kRecPlain on an entry: the lane need not read the recorder's row at all -- no source filter, and either no
            // facet, or a facet that IS the bin's axis (exactly +-1 on it, zeros elsewhere) on an unrotated box, whose
            // world normals are exactly such unit vectors: |facet - normal| is exactly 0 for every normal of the bin
            const bool exact_normals = t->geom_type[node_of_key] == PVT_GEOM_BOX && unrotated(node_of_key);
            auto unfiltered = [&](int r) { return !t->rec_source_mode || t->rec_source_mode[r] == 0; };
            auto axis_facet = [&](int r, int b) {
                const double* f = t->rec_facet + r * 3;
                for (int a = 0; a < 3; a++)
                    if (f[a] != (a == b / 2 ? (b % 2 ? 1.0 : -1.0) : 0.0)) return false;
                return t->rec_atol[r] >= 0.0;
            };
            for (int b = 0; b < 6; b++) {
                rec[2 + b] = (owner[b] >= 0 && !clash[b]) ? owner[b] : -1;
                if (rec[2 + b] >= 0 && exact_normals && unfiltered(owner[b]) && axis_facet(owner[b], b)) rec[2 + b] |= kRecPlain;
            }
            for (int r = 0; r < R; r++) {
                if (t->rec_node[r] * 7 + t->rec_event[r] != key) continue;
                int b = bin_of(r);
                if (b >= 0 && !clash[b]) continue;  // served by the bin table
                gi[lay.cand_list + at++] = r | ((unfiltered(r) && !t->rec_has_facet[r]) ? kRecPlain : 0);
            }
            rec[1] = at - rec[0];
          }
        }
    }
    for (int n = 0; n < N; n++) {
        double* d = gd.data() + n * ND;
        for (int r = 0; r < 3; r++) d[ND_T + r] = t->world_to_local[n * 16 + r * 4 + 3];
        for (int c = 0; c < 3; c++) d[ND_PARAMS + c] = t->geom_params[n * 4 + c];
        const unsigned long long bits = (unsigned long long)(unsigned int)((unrotated(n) ? 1 : 0) | (t->geom_type[n] << 8)) |
                                        ((unsigned long long)(unsigned int)rot_class[n] << 32);
        std::memcpy(&d[ND_BITS], &bits, 8);
        d[ND_N] = t->refractive_index[n];
        int* q = gi.data() + n * NI;
        q[NI_SURF] = t->surface_type[n];
        q[NI_CSTART] = t->comp_start[n];
        q[NI_CCOUNT] = t->comp_count[n];
        q[NI_CREC] = node_crec[n];
        q[NI_KSTART] = K > 0 ? t->coat_start[n] : 0;
        q[NI_KCOUNT] = K > 0 ? t->coat_count[n] : 0;
        q[NI_MESH] = -1;
        q[NI_CAND] = node_cand[n];
        q[NI_NCLS] = idx_class[n];
        if (t->geom_type[n] == PVT_GEOM_MESH) {
            const int f0 = t->mesh_face_start[n], fc = t->mesh_face_count[n];
            double centre[3];
            q[NI_MESH] = pvt::BvhBuilder(t->mesh_vertices, t->mesh_faces, t->mesh_normals, bvh_nodes, bvh_tris)
                             .add_mesh(f0, fc, centre);
            bvh_roots.push_back(q[NI_MESH]);
            for (int c = 0; c < 3; c++) d[ND_PARAMS + c] = centre[c];   // a mesh has no shape parameters: the point its boxes are relative to
        }
    }
    for (int rc = 0; rc < CR; rc++) {
        const int c = rec_comp[rc];
        double* d = gd.data() + lay.comp_d + rc * CD;
        d[CD_QY] = t->comp_qy[c];
        d[CD_TAU_RAD] = t->comp_tau_rad[c];
        d[CD_TAU_NR] = t->comp_tau_nr[c];
        d[CD_PHASE] = t->comp_phase_param[c];
        int* q = gi.data() + lay.comp_i + rc * CI;
        q[CI_TYPE] = t->comp_type[c];
        q[CI_PHASE] = t->comp_phase_type[c];
        if (t->comp_phase_type[c] == PVT_PHASE_LAMBERTIAN) {
            // The Lambertian phase function, theta = asin(sqrt(p1)), IS the cone's theta = asin(sqrt(p1) sin(theta_max)) at
            // theta_max = pi/2 -- same two draws in the same order -- provided sin(pi/2) is the double 1.0 in the
            // kernel's arithmetic (x * 1.0 is exact); checked here with the very function the kernel calls.
            const double half_pi = 1.5707963267948966;
            if (pvt_sin(half_pi) != 1.0) return fail(PVT_ERR_INVALID, "pvt_sin(pi/2) != 1: the Lambertian phase function cannot be lowered to a cone");
            d[CD_PHASE] = half_pi;
            q[CI_PHASE] = PVT_PHASE_CONE;
        }
        q[CI_ABS_X] = c_abs_x[c];   // absolute offsets into the double blob
        q[CI_ABS_Y] = c_abs_y[c];
        q[CI_ABS_N] = t->comp_abs_n[c];
        q[CI_EMS_X] = c_ems_x[c];
        q[CI_EMS_CDF] = c_ems_c[c];
        q[CI_EMS_N] = t->comp_ems_n[c];
        q[CI_ABS_HIST] = abs_hist(c);
        q[CI_EMS_HIST] = ems_hist(c);
        // guide tables only for the arrays that are searched (see even_w); -1 is never dereferenced
        q[CI_ABS_G] = c_abs_g[c] < 0 ? -1 : guide0 + c_abs_g[c];
        q[CI_EMS_GX] = c_ems_gx[c] < 0 ? -1 : guide0 + c_ems_gx[c];
        q[CI_EMS_GC] = guide0 + c_ems_gc[c];
        d[CD_ABS_RCP] = c_abs_rcp[c];
        d[CD_EMS_RCP_X] = c_ems_rcp_x[c];
        d[CD_EMS_RCP_C] = c_ems_rcp_c[c];
        d[CD_ABS_W] = c_abs_w[c];
        d[CD_EMS_W] = c_ems_w[c];
        // the guide scales are functions of the tables alone: computed (and the tables written) by whoever owns them,
        // which is always a component with a record of its own or an earlier one -- fill in from the owner below
    }
    // the tables themselves, once per owner (a compact table keeps its first abscissa only), and their guide tables
    std::vector<double> abs_scale(C, 0.0), ems_scale_x(C, 0.0), ems_scale_c(C, 0.0);
    for (int c = 0; c < C; c++) {
        const double* ax = t->abs_x + t->comp_abs_start[c];
        const double* ex = t->ems_x + t->comp_ems_start[c];
        const double* ec = t->ems_cdf + t->comp_ems_start[c];
        const int an = t->comp_abs_n[c], en = t->comp_ems_n[c];
        if (abs_of[c] == c) {
            const bool abs_compact = c_abs_g[c] < 0;
            if (!abs_compact) build_guide(ax, an, guide0 + c_abs_g[c], &abs_scale[c]);
            for (int i = 0; i < (abs_compact ? (an > 0 ? 1 : 0) : an); i++) gd[c_abs_x[c] + i] = ax[i];
            for (int i = 0; i < an; i++) gd[c_abs_y[c] + i] = t->abs_y[t->comp_abs_start[c] + i];
        } else {
            abs_scale[c] = abs_scale[abs_of[c]];
        }
        if (ems_of[c] == c) {
            const bool ems_compact = c_ems_gx[c] < 0;
            if (!ems_compact) build_guide(ex, en, guide0 + c_ems_gx[c], &ems_scale_x[c]);
            build_guide(ec, en, guide0 + c_ems_gc[c], &ems_scale_c[c]);
            for (int i = 0; i < (ems_compact ? (en > 0 ? 1 : 0) : en); i++) gd[c_ems_x[c] + i] = ex[i];
            for (int i = 0; i < en; i++) gd[c_ems_c[c] + i] = ec[i];
        } else {
            ems_scale_x[c] = ems_scale_x[ems_of[c]];
            ems_scale_c[c] = ems_scale_c[ems_of[c]];
        }
    }
    for (int rc = 0; rc < CR; rc++) {
        const int c = rec_comp[rc];
        double* d = gd.data() + lay.comp_d + rc * CD;
        d[CD_ABS_SCALE] = abs_scale[c];
        d[CD_EMS_SCALE_X] = ems_scale_x[c];
        d[CD_EMS_SCALE_C] = ems_scale_c[c];
    }
    for (int r = 0; r < R; r++) {
        double* d = gd.data() + lay.rec_d + r * RD;
        for (int a = 0; a < 3; a++) d[RD_FACET + a] = t->rec_facet[r * 3 + a];
        d[RD_ATOL] = t->rec_atol[r];
        int* q = gi.data() + lay.rec_i + r * RI;
        q[RI_NODE] = t->rec_node[r];
        q[RI_EVENT] = t->rec_event[r];
        q[RI_HAS_FACET] = t->rec_has_facet[r];
        q[RI_HSTART] = t->rec_hist_start[r];
        q[RI_HN] = t->rec_hist_n[r];
        q[RI_SRC_MODE] = t->rec_source_mode ? t->rec_source_mode[r] : 0;
        q[RI_SRC_ID] = t->rec_source_id ? t->rec_source_id[r] : -1;
    }
    for (int h = 0; h < H; h++) {
        double* d = gd.data() + lay.hist_d + h * HD;
        d[HD_LO_A] = t->hist_lo_a[h]; d[HD_HI_A] = t->hist_hi_a[h];
        d[HD_LO_B] = t->hist_lo_b[h]; d[HD_HI_B] = t->hist_hi_b[h];
        auto rcp_or_nan = [](double width) {   // NaN: the kernel divides for real
            return (std::isfinite(width) && std::fabs(width) > 1e-290 && std::fabs(width) < 1e290) ? 1.0 / width : NAN;
        };
        d[HD_RA] = rcp_or_nan(t->hist_hi_a[h] - t->hist_lo_a[h]);
        d[HD_RB] = rcp_or_nan(t->hist_hi_b[h] - t->hist_lo_b[h]);
        int* q = gi.data() + lay.hist_i + h * HI;
        q[HI_PA] = t->hist_prop_a[h]; q[HI_PB] = t->hist_prop_b[h];
        q[HI_NA] = t->hist_na[h]; q[HI_NB] = t->hist_nb[h]; q[HI_OFF] = t->hist_offset[h];
    }
    for (int k = 0; k < K; k++) {
        double* d = gd.data() + lay.coat_d + k * KD;
        for (int a = 0; a < 3; a++) {
            d[KD_FACET + a] = t->coat_facet[k * 3 + a];
            d[KD_LO + a] = t->coat_lo[k * 3 + a];
            d[KD_HI + a] = t->coat_hi[k * 3 + a];
        }
        d[KD_REFL] = t->coat_reflectivity[k];
        int* q = gi.data() + lay.coat_i + k * KI;
        q[KI_RMODE] = t->coat_reflect_mode[k];
        q[KI_TMODE] = t->coat_transmit_mode[k];
    }

    // Lazy root (kernel node loop): the root is a box or a sphere and every other node lies strictly inside it,
    // its bounding sphere clearing the root's surface by a margin -- then a ray from inside the root meets every
    // other node's surface strictly before the root's.
    int lazy_root = 0;
    double lazy_k = 0.0;
    bool exit_observed = false;
    for (int r = 0; r < R; r++)
        if (t->rec_node[r] == t->root_id && t->rec_event[r] == PVT_REC_EXIT) exit_observed = true;
    if (!getenv("PVT_NO_LAZY_ROOT") && (t->geom_type[t->root_id] == PVT_GEOM_BOX || t->geom_type[t->root_id] == PVT_GEOM_SPHERE)) {
        const int root = t->root_id;
        const double* w2l = t->world_to_local + root * 16;
        const double* rp = t->geom_params + root * 4;
        const bool box = t->geom_type[root] == PVT_GEOM_BOX;
        const double scale = box ? std::fmax(rp[0], std::fmax(rp[1], rp[2])) : rp[0];
        const double margin = 1e-6 * scale + 1e-9;
        bool inside = std::isfinite(scale) && scale > 0.0;
        for (int n = 0; n < N && inside; n++) {
            if (n == root) continue;
            const double* l2w = t->local_to_world + n * 16;
            const double* gp = t->geom_params + n * 4;
            double radius;   // of a sphere about the node's origin that holds the whole shape
            switch (t->geom_type[n]) {
                case PVT_GEOM_BOX: radius = 0.5 * std::sqrt(gp[0] * gp[0] + gp[1] * gp[1] + gp[2] * gp[2]); break;
                case PVT_GEOM_SPHERE: radius = gp[0]; break;
                case PVT_GEOM_CYLINDER: radius = std::sqrt(gp[1] * gp[1] + 0.25 * gp[0] * gp[0]); break;
                default: radius = INFINITY; break;   // (mesh scenes never take this path)
            }
            radius *= 1.0 + 1e-12;
            double c[3];   // the node's origin in the root's frame
            for (int a = 0; a < 3; a++)
                c[a] = w2l[a * 4] * l2w[3] + w2l[a * 4 + 1] * l2w[7] + w2l[a * 4 + 2] * l2w[11] + w2l[a * 4 + 3];
            if (box) {
                for (int a = 0; a < 3; a++)
                    if (!(0.5 * rp[a] - std::fabs(c[a]) - radius > margin)) inside = false;
            } else {
                if (!(rp[0] - std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) - radius > margin)) inside = false;
            }
        }
        if (inside) {
            lazy_root = box ? 1 : 2;
            lazy_k = box ? 0.0 : 1.0 / (2.0 * rp[0]);
        }
    }
    // Fused exit (kernel surface branch): the scene is ONE unrotated box inside a lazy root whose medium neither
    // absorbs nor is listened to -- a photon that leaves the box's surface outwards can only leave the scene.
    bool fuse_exit = false;
    if (lazy_root && N == 2 && !getenv("PVT_NO_FUSED_EXIT")) {
        const int child = 1 - t->root_id;
        bool ok = t->geom_type[child] == PVT_GEOM_BOX && unrotated(child) && t->comp_count[t->root_id] == 0;
        for (int r = 0; r < R; r++)
            if (t->rec_node[r] == t->root_id) ok = false;
        fuse_exit = ok;
    }

    // owned until every upload has succeeded: a failing HIP call must not leak the scene
    struct Owner {
        PvtScene* p;
        ~Owner() { if (p) pvt_scene_destroy(p); }
    } owner{new PvtScene()};
    PvtScene* s = owner.p;
    s->stage.reserve(kCursorSlots); s->stage_bytes.reserve(kCursorSlots); s->carry.reserve(kCursorSlots);   // (references stay valid)
    s->emit_pool.reserve(kCursorSlots); s->emit_pool_bytes.reserve(kCursorSlots);
    s->device = device;
    s->lay = lay;
    s->nd = (int)gd.size();
    s->ni = (int)gi.size();
    s->nd_small = spec_d;
    s->ni_small = guide0;
    s->n_nodes = N;
    s->root = t->root_id;
    s->n_rec = R;
    s->total_bins = t->total_bins;
    s->n_coat = K;
    s->lazy_root = lazy_root;
    s->lazy_k = lazy_k;
    s->exit_observed = exit_observed;
    s->fuse_exit = fuse_exit;
    s->grid = has_grid;
    for (int a = 0; a < 3; a++) s->grid_dims[a] = has_grid ? grid.n[a] : 0;
    for (int h = 0; h < H; h++)
        if (t->hist_prop_a[h] >= 4 || t->hist_prop_b[h] >= 4) s->hist_reads_position = true;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    s->num_cu = prop.multiProcessorCount;
    s->lds_limit = prop.sharedMemPerBlock;
    s->consolidate = getenv("PVT_NO_CONSOLIDATE") == nullptr;
    if (const char* env = getenv("PVT_BLOCKS_PER_CU")) s->dev_blocks_per_cu = atof(env);
    if (const char* env = getenv("PVT_STAGE_BYTES")) s->stage_limit = (size_t)atoll(env);   // (tests: force several launches)
    HIP_TRY(hipMalloc(&s->d_gd, gd.size() * sizeof(double)));
    HIP_TRY(hipMalloc(&s->d_gi, gi.size() * sizeof(int)));
    HIP_TRY(hipMalloc(&s->d_cursor, 64 * kCursorSlots + 256));   // + room for the PVT_STATS counters
    HIP_TRY(hipMemset(s->d_cursor, 0, 64 * kCursorSlots + 256));
    HIP_TRY(hipMalloc(&s->d_set_cursor, (size_t)kCursorSlots * kMaxSets * 4));
    HIP_TRY(hipMalloc(&s->d_counters, 64 * 8 * 8));
    HIP_TRY(hipMemset(s->d_counters, 0, 64 * 8 * 8));
    HIP_TRY(hipMalloc(&s->d_stamp, (size_t)kCursorSlots * 2 * 8));
    HIP_TRY(hipMemset(s->d_stamp, 0, (size_t)kCursorSlots * 2 * 8));
    if (bvh_nodes.size() >= ((size_t)1 << 26) || bvh_tris.size() >= ((size_t)1 << 26))
        return fail(PVT_ERR_INVALID, "meshes too large: the walk's cursors and leaf references hold 2^26 records / triangles");
    if (!bvh_nodes.empty()) {
        // a lane of a walk notes kMeshQ leaves before their triangles are tested -- one, in trees of a handful of records
        // (the kernel applies the same rule per tree)
        s->meshq = 1;
        for (int r : bvh_roots)
            if (bvh_nodes[r].skip - r > 16) s->meshq = kMeshQ;   // (15 records and the unused one after the root)
        // The top levels of the trees go to LDS (pvt_bvh.h: stage_top), as many records as leave four workgroups per CU
        // -- the mesh variants' four waves per SIMD -- their LDS in either kind of launch (tallies / histories).
        {
            const LdsPlan tally = plan_lds(s, false), hist = plan_lds(s, true);
            const size_t other = (tally.bytes > hist.bytes ? tally.bytes : hist.bytes) + (size_t)s->meshq * kBlock * 4;
            const size_t per_wg = (size_t)39 * 1024 * 4 / kMeshWaves;
            size_t room = tally.ok && hist.ok && other < per_wg ? per_wg - other : 0;
            if (room > 32 * 1024) room = 32 * 1024;
            if (const char* env = getenv("PVT_MESH_TOP_BYTES")) {   // (developer override, never past the room computed above; 0: no copy)
                const size_t asked = (size_t)atoll(env);
                room = asked < room ? asked : room;
            }
            std::vector<pvt::BvhNode> top;
            pvt::stage_top(bvh_nodes, bvh_roots, room / sizeof(pvt::BvhNode), top);
            s->top_n = (int)top.size();
            // what was planned without the copy must still hold with it (plan_lds reserves the copy first): same table
            // placement, everything inside the LDS of a workgroup
            const LdsPlan tally2 = plan_lds(s, false), hist2 = plan_lds(s, true);
            const size_t with_top = (size_t)s->meshq * kBlock * 4 + (size_t)s->top_n * sizeof(pvt::BvhNode);
            if (tally2.ok != tally.ok || hist2.ok != hist.ok || tally2.tab_lds != tally.tab_lds || hist2.tab_lds != hist.tab_lds ||
                tally2.bytes + with_top > s->lds_limit || hist2.bytes + with_top > s->lds_limit)
                return fail(PVT_ERR_INVALID, "internal: the LDS plan of a mesh scene changed when the copy of its trees' top levels was added");
            if (s->top_n > 0) {
                HIP_TRY(hipMalloc(&s->d_bvh_top, top.size() * sizeof(pvt::BvhNode)));
                HIP_TRY(hipMemcpy(s->d_bvh_top, top.data(), top.size() * sizeof(pvt::BvhNode), hipMemcpyHostToDevice));
            }
        }
        bvh_nodes.push_back(pvt::BvhNode{});   // (one record past the end, kept for walks that fetch ahead)
        HIP_TRY(hipMalloc(&s->d_bvh, bvh_nodes.size() * sizeof(pvt::BvhNode)));
        HIP_TRY(hipMalloc(&s->d_tris, bvh_tris.size() * sizeof(pvt::MeshTri)));
        HIP_TRY(hipMemcpy(s->d_bvh, bvh_nodes.data(), bvh_nodes.size() * sizeof(pvt::BvhNode), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_tris, bvh_tris.data(), bvh_tris.size() * sizeof(pvt::MeshTri), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(s->d_gd, gd.data(), gd.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_gi, gi.data(), gi.size() * sizeof(int), hipMemcpyHostToDevice));
    owner.p = nullptr;
    *out = s;
    return PVT_OK;
}

int pvt_scene_set_emitter(PvtScene* s, const PvtEmitterTables* e) {
    if (!s || !e || e->n_lights <= 0) return fail(PVT_ERR_INVALID, "bad emitter");
    HIP_TRY(hipSetDevice(s->device));
    std::vector<double> ed;
    std::vector<int> ei;
    EmitOff o{};
    const int Lt = e->n_lights;
    o.wl_value = push(ed, e->wl_value, Lt);
    o.pos_param = push(ed, e->pos_param, (size_t)Lt * 3);
    o.dir_param = push(ed, e->dir_param, Lt);
    o.l2w = push(ed, e->light_to_world, (size_t)Lt * 16);
    o.spec_x = push(ed, e->spec_x, e->n_spec);
    o.spec_cdf = push(ed, e->spec_cdf, e->n_spec);
    o.wl_type = push(ei, e->wl_type, Lt);
    o.wl_spec_start = push(ei, e->wl_spec_start, Lt);
    o.wl_spec_n = push(ei, e->wl_spec_n, Lt);
    o.pos_type = push(ei, e->pos_type, Lt);
    o.dir_type = push(ei, e->dir_type, Lt);
    ed.push_back(0.0);
    ei.push_back(0);
    if (s->d_ed) { (void)hipFree(s->d_ed); s->d_ed = nullptr; }
    if (s->d_ei) { (void)hipFree(s->d_ei); s->d_ei = nullptr; }
    HIP_TRY(hipMalloc(&s->d_ed, ed.size() * sizeof(double)));
    HIP_TRY(hipMalloc(&s->d_ei, ei.size() * sizeof(int)));
    HIP_TRY(hipMemcpy(s->d_ed, ed.data(), ed.size() * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->d_ei, ei.data(), ei.size() * sizeof(int), hipMemcpyHostToDevice));
    s->eoff = o;
    s->n_lights = Lt;
    return PVT_OK;
}

void pvt_scene_destroy(PvtScene* s) {
    if (!s) return;
#if PVT_TIMELINE
    timeline_dump();
#endif
    (void)hipSetDevice(s->device);
    if (s->d_gd) (void)hipFree(s->d_gd);
    if (s->d_gi) (void)hipFree(s->d_gi);
    if (s->d_ed) (void)hipFree(s->d_ed);
    if (s->d_ei) (void)hipFree(s->d_ei);
    if (s->d_cursor) (void)hipFree(s->d_cursor);
    if (s->d_set_cursor) (void)hipFree(s->d_set_cursor);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->d_stamp) (void)hipFree(s->d_stamp);
    if (s->d_bvh) (void)hipFree(s->d_bvh);
    if (s->d_bvh_top) (void)hipFree(s->d_bvh_top);
    if (s->d_tris) (void)hipFree(s->d_tris);
    for (auto* b : s->stage) if (b) (void)hipFree(b);
    for (auto& c : s->carry) for (auto* b : c.buf) if (b) (void)hipFree(b);
    for (auto* b : s->emit_pool) if (b) (void)hipFree(b);
    delete s;
}

}  // extern "C"

namespace {

KArgs base_args(const PvtScene* s, const PvtTraceParams* p) {
    KArgs a{};
    a.gd = s->d_gd; a.gi = s->d_gi; a.ed = s->d_ed; a.ei = s->d_ei;
    a.bvh = s->d_bvh; a.tris = s->d_tris;
    a.lay = s->lay; a.eoff = s->eoff;
    a.nd = s->nd; a.ni = s->ni;
    a.nd_lds = 0; a.ni_lds = 0;
    a.n_nodes = s->n_nodes; a.root = s->root; a.n_rec = s->n_rec; a.total_bins = s->total_bins;
    a.n_coat = s->n_coat; a.n_lights = s->n_lights;
    a.n_rays = (unsigned int)p->n_rays;
    a.cursor = s->d_cursor;
    a.counters = s->d_counters;
    a.seed = p->seed + p->ray_offset;
    a.emit_seed = p->emit_seed;
    a.ray_offset = p->ray_offset;
    a.maxsteps = p->maxsteps; a.max_events = p->max_events; a.emit_method = p->emit_method;
    a.record_every = p->record_every;
    return a;
}

#ifndef PVT_DEV_VARIANTS
#define PVT_DEV_VARIANTS 0   // developer builds: only the analytic, array-input, <=64-recorder variants (fast compile)
#endif
template <bool RECORD, int TAB_LDS, int SEENW>
hipError_t launch_variant(bool emit, int grid, size_t lds, hipStream_t st, const KArgs& a) {
    const bool mesh = a.bvh != nullptr;
    if constexpr (TAB_LDS == 1 && (!PVT_DEV_VARIANTS || SEENW == 1) && PVT_DEV_VARIANTS != 2) {
        if (a.lay.grid_d >= 0 && !mesh && (!PVT_DEV_VARIANTS || !emit)) {   // many nodes: per-lane walk of the node grid
            if (emit && !PVT_DEV_VARIANTS) hipLaunchKernelGGL((trace_kernel_grid<RECORD, SEENW, !PVT_DEV_VARIANTS>), dim3(grid), dim3(kBlock), lds, st, a);
            else hipLaunchKernelGGL((trace_kernel_grid<RECORD, SEENW, false>), dim3(grid), dim3(kBlock), lds, st, a);
            return hipGetLastError();
        }
    }
#if PVT_DEV_VARIANTS == 2   // developer builds of the MESH variants only (tables in LDS, <= 64 recorders, array input)
    if (emit || !mesh || TAB_LDS != 1 || SEENW != 1) return hipErrorNotSupported;
    if constexpr (TAB_LDS == 1 && SEENW == 1) hipLaunchKernelGGL((trace_kernel<RECORD, 1, 1, false, true>), dim3(grid), dim3(kBlock), lds, st, a);
#elif PVT_DEV_VARIANTS
    if (emit || mesh || TAB_LDS != 1 || SEENW != 1) return hipErrorNotSupported;
    if constexpr (TAB_LDS == 1 && SEENW == 1) {
        if constexpr (RECORD) hipLaunchKernelGGL((trace_kernel_w4<true, 1, 1, false>), dim3(grid), dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((trace_kernel_w4<false, 1, 1, false>), dim3(grid), dim3(kBlock), lds, st, a);
    }
#else
    // (mesh scenes stage all their tables or none: plan_lds never asks for the heads alone there)
    constexpr int MESH_TAB = TAB_LDS == 2 ? 0 : TAB_LDS;
    if (mesh && TAB_LDS == 2) return hipErrorNotSupported;
    if (emit) {
        if (mesh) hipLaunchKernelGGL((trace_kernel<RECORD, MESH_TAB, SEENW, true, true>), dim3(grid), dim3(kBlock), lds, st, a);
        else if constexpr (RECORD) hipLaunchKernelGGL((trace_kernel_w4<true, TAB_LDS, SEENW, true>), dim3(grid), dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((trace_kernel_w4<false, TAB_LDS, SEENW, true>), dim3(grid), dim3(kBlock), lds, st, a);
    } else {
        if (mesh) hipLaunchKernelGGL((trace_kernel<RECORD, MESH_TAB, SEENW, false, true>), dim3(grid), dim3(kBlock), lds, st, a);
        else if constexpr (RECORD) hipLaunchKernelGGL((trace_kernel_w4<true, TAB_LDS, SEENW, false>), dim3(grid), dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((trace_kernel_w4<false, TAB_LDS, SEENW, false>), dim3(grid), dim3(kBlock), lds, st, a);
    }
#endif
    return hipGetLastError();
}

template <bool RECORD, int TAB_LDS>
hipError_t launch_seen(int n_rec, bool emit, int grid, size_t lds, hipStream_t st, const KArgs& a) {
    if (n_rec <= 64) return launch_variant<RECORD, TAB_LDS, 1>(emit, grid, lds, st, a);
    return launch_variant<RECORD, TAB_LDS, 4>(emit, grid, lds, st, a);
}

}  // namespace

extern "C" {

}  // extern "C"

namespace {

// cursor / staging slot of a stream (launches on one stream are ordered and share it)
int slot_of_stream(PvtScene* s, hipStream_t st) {
    std::lock_guard<std::mutex> lock(s->slot_mutex);
    size_t slot = 0;
    while (slot < s->slot_of.size() && s->slot_of[slot] != st) slot++;
    if (slot == s->slot_of.size()) {
        if (slot >= (size_t)kCursorSlots) return -1;
        s->slot_of.push_back(st);
        s->stage.push_back(nullptr);
        s->stage_bytes.push_back(0);
        s->carry.emplace_back();
        s->emit_pool.push_back(nullptr);
        s->emit_pool_bytes.push_back(0);
    }
    return (int)slot;
}

int check_trace_args(PvtScene* s, const PvtRays* rays, const PvtTraceParams* p, const PvtTallies* tl) {
    if (!s || !p || !tl) return fail(PVT_ERR_INVALID, "null argument");
    if (p->n_rays < 0 || p->n_rays > 0x7fffffffLL)
        return fail(PVT_ERR_INVALID, "n_rays must fit in 31 bits per bundle");
    if (p->max_events < 2 && p->record_every > 0) return fail(PVT_ERR_INVALID, "max_events must be >= 2");
    if (!rays && !s->d_ed && p->n_rays > 0) return fail(PVT_ERR_INVALID, "no rays and no emitter");
    if ((p->flags & PVT_FLAG_CARRY_OUT) && (p->record_every > 0 || p->tally_bundle > 0))
        return fail(PVT_ERR_INVALID, "PVT_FLAG_CARRY_OUT is for plain tally launches (record_every == 0, no tally sets)");
    if (p->tally_bundle > 0 && p->record_every > 0) return fail(PVT_ERR_INVALID, "tally_bundle needs record_every == 0");
    return PVT_OK;
}

// Enqueue ONE trace kernel.  `log_rows` / `log_counts`: the event records of the recorded rays (device memory,
// null when record_every == 0); counts are cleared here.
// LDS of a launch, before what mesh walks add: recorder accumulators + control words, tables if they fit, bins if
// they fit, then the drain-phase consolidation buffer (kXSlots photon states, also the seed pools) within 40 KB.
// `s->meshq` and `s->top_n` (mesh scenes) are reserved first, so that what is decided here still fits with them.
LdsPlan plan_lds(const PvtScene* s, bool record) {
    LdsPlan lp{0, false, false, 0, 0, 0, true};
    const size_t reserved = (size_t)s->meshq * kBlock * 4 + (size_t)s->top_n * sizeof(pvt::BvhNode);
    const size_t acc_bytes = (size_t)s->n_rec * (8 * 8 + 8) + (size_t)((s->n_rec + 1) & ~1) * 4 + CTL_WORDS * 4;
    const size_t tab_bytes = (size_t)s->nd * 8 + (size_t)((s->ni + 1) & ~1) * 4;
    const size_t bins_bytes = ((size_t)s->total_bins * 4 + 7) & ~(size_t)7;
    const size_t lds_limit = s->lds_limit - reserved;
    const size_t budget = (64 * 1024 < s->lds_limit ? 64 * 1024 : s->lds_limit) - reserved;  // keep >= 2 workgroups per CU
    // per-wave queues of first crossings awaiting their statistics (kernel: tally_flush)
    lp.tq_pos = s->hist_reads_position ? 1 : 0;
    const size_t tq_bytes = (size_t)kWaves * kTallyQ * ((lp.tq_pos ? 7 : 4) * 8 + 4);
    if (acc_bytes + tq_bytes > lds_limit) { lp.ok = false; return lp; }
    // (PVT_TABLES: developer / test switch -- "global": no tables in LDS, "heads": never the spectra)
    const char* force = getenv("PVT_TABLES");
    lp.tab_lds = acc_bytes + tq_bytes + tab_bytes <= budget && !force;
    // spectra too large for LDS: everything else -- node, component and recorder records, class tables -- still is staged
    // (the blobs' heads; the spectra and their guide tables are read from global memory)
    const size_t small_bytes = (size_t)s->nd_small * 8 + (size_t)((s->ni_small + 1) & ~1) * 4;
    lp.small_lds = !lp.tab_lds && s->meshq == 0 && acc_bytes + tq_bytes + small_bytes <= 40 * 1024 && acc_bytes + tq_bytes + small_bytes <= budget &&
                   !(force && force[0] == 'g');
    size_t lds = acc_bytes + tq_bytes + (lp.tab_lds ? tab_bytes : lp.small_lds ? small_bytes : 0);
    lp.bins_in_lds = (lds + bins_bytes <= budget) ? 1 : 0;
    if (lp.bins_in_lds) lds += bins_bytes;
    const size_t xw = 14 + (s->n_rec <= 64 ? 1 : 4) + (record ? 1 : 0);
    const size_t xbytes = (size_t)kXSlots * xw * 8;
    // (mesh scenes do without: measured, repacking a draining workgroup buys their launches nothing, and the 9 KB hold
    // another level of the trees' top)
    if (s->consolidate && s->meshq == 0 && lds + xbytes <= 40 * 1024 && lds + xbytes <= lds_limit) {
        lp.xslots = kXSlots;
        lds += xbytes;
    }
    lp.bytes = (lds + 15) & ~(size_t)15;
    return lp;
}

int trace_launch(PvtScene* s, const PvtRays* rays, const PvtTraceParams* p, const PvtTallies* tl,
                 unsigned long long* log_rows, int* log_counts, hipStream_t st) {
    const int slot = slot_of_stream(s, st);
    if (slot < 0)
        return fail(PVT_ERR_INVALID, "more than 64 HIP streams are tracing this scene; create one scene per group of streams");
    PvtScene::Carry& carry = s->carry[(size_t)slot];
    const bool carry_in = carry.pending;
    if (p->n_rays == 0 && !carry_in) return PVT_OK;
    if (carry_in && (p->record_every > 0 || p->tally_bundle > 0))
        return fail(PVT_ERR_INVALID, "photons parked by the previous launch on this stream (PVT_FLAG_CARRY_OUT) are waiting: "
                                     "finish them with a plain tally launch (n_rays may be 0) first");
    if (carry_in && (p->maxsteps != carry.maxsteps || p->emit_method != carry.emit_method))
        return fail(PVT_ERR_INVALID, "photons parked by the previous launch on this stream were traced with maxsteps " +
                                     std::to_string(carry.maxsteps) + ", emit_method " + std::to_string(carry.emit_method) +
                                     "; this launch asks for " + std::to_string(p->maxsteps) + ", " + std::to_string(p->emit_method) +
                                     ": finish them under their own rules first (a launch with n_rays = 0 and the old values), or drop "
                                     "them with pvt_scene_carry_discard");
    long long n_sets = 0;
    if (p->tally_bundle > 0) {
        n_sets = (p->n_rays + p->tally_bundle - 1) / p->tally_bundle;
        if (p->tally_bundle > 0x7fffffffLL || n_sets > kMaxSets)
            return fail(PVT_ERR_INVALID, "at most 1024 tally sets per launch");
        if (p->tally_stride_i64 < 0 || p->tally_stride_f64 < 0) return fail(PVT_ERR_INVALID, "negative tally stride");
    }
    HIP_TRY(hipSetDevice(s->device));

    KArgs a = base_args(s, p);
    a.stamp = s->d_stamp + 2 * (size_t)slot;
    if (rays) { a.pos = rays->position; a.dir = rays->direction; a.wl = rays->wavelength; }
    a.rec_distinct = reinterpret_cast<long long*>(tl->rec_distinct);
    a.rec_crossings = reinterpret_cast<long long*>(tl->rec_crossings);
    a.rec_sums = tl->rec_sums;
    a.rec_bins = reinterpret_cast<long long*>(tl->rec_bins);
    const bool record = p->record_every > 0;
    // nobody looks at where a photon leaves the scene: the root's distance is only needed to ORDER crossings
    a.lazy_root = (!record && !s->exit_observed && !s->d_bvh) ? s->lazy_root : 0;
    a.lazy_tail = s->d_bvh ? 0 : s->lazy_root;
    a.fuse_exit = (a.lazy_root && s->fuse_exit) ? 1 : 0;
    a.lazy_k = s->lazy_k;
    if (record) {
        if (!log_rows || !log_counts) return fail(PVT_ERR_INVALID, "record_every > 0 needs an event log");
        a.log_rows = log_rows;
        a.log_counts = log_counts;
        const size_t nrec = (size_t)((p->n_rays + p->record_every - 1) / p->record_every);
        HIP_TRY(hipMemsetAsync(log_counts, 0, nrec * 4, st));
    }
    // the cursor belongs to the stream: two launches can only overlap on different streams.  A slot holds THREE
    // blocks {ray cursor, claim cursor of the resumed photons, count of the photons parked} used in rotation:
    // launch L counts in block L mod 3, reads how many photons its predecessor parked from block L-1, and clears
    // block L+1 for its successor (all zero at scene creation) -- so no fill kernel sits between two launches
    unsigned int* const blocks = s->d_cursor + 16 * slot;
    if (n_sets) {
        a.cursor = s->d_set_cursor + (size_t)kMaxSets * slot;
        HIP_TRY(hipMemsetAsync(a.cursor, 0, (size_t)n_sets * 4, st));
    } else {
        a.cursor = blocks + 4 * carry.phase;
        a.cursor_next = blocks + 4 * ((carry.phase + 1) % 3);
    }
#if PVT_STATS
    static unsigned long long* g_stats = nullptr;
    if (!g_stats) (void)hipMalloc(&g_stats, 256);
    (void)hipMemsetAsync(g_stats, 0, 256, st);
    a.cursor = reinterpret_cast<unsigned int*>(g_stats);  // dev build: counters live in their own buffer
    a.cursor_next = nullptr;
#endif

    const LdsPlan lp = plan_lds(s, record);
    if (!lp.ok) return fail(PVT_ERR_INVALID, "recorder accumulators exceed LDS");
    a.tq_pos = lp.tq_pos;
    a.bins_in_lds = lp.bins_in_lds;
    a.xslots = lp.xslots;
    const bool tab_lds = lp.tab_lds;
    if (lp.small_lds) { a.nd_lds = s->nd_small; a.ni_lds = s->ni_small; }
    size_t lds = lp.bytes;
    a.meshq_off = -1;
    a.top_off = 0; a.top_n = 0; a.bvh_top = nullptr;
    if (a.bvh) {   // mesh walks: the lanes' noted leaves, then the copy of the trees' top levels
        a.meshq_off = (int)lds;
        lds += (size_t)s->meshq * kBlock * 4;
        a.top_off = (int)lds; a.top_n = s->top_n; a.bvh_top = s->d_bvh_top;
        lds += (size_t)s->top_n * sizeof(pvt::BvhNode);
        if (lds > s->lds_limit) return fail(PVT_ERR_INVALID, "mesh staging exceeds LDS");
    }

    // persistent grid: enough workgroups to fill every CU a few times over,
    // never more than the rays can feed
    long long blocks_for_rays = (p->n_rays + (carry_in ? carry.bound : 0) + kBlock - 1) / kBlock;
    double per_cu = p->workgroups_per_cu > 0 ? (double)p->workgroups_per_cu : 4.0;
    if (s->dev_blocks_per_cu > 0) per_cu = s->dev_blocks_per_cu;   // developer override, read once per scene
    long long grid = (long long)((double)s->num_cu * per_cu);
    if (grid > blocks_for_rays) grid = blocks_for_rays;
    if (grid < 1) grid = 1;
    // Every wave resumes the slice of the parked photons that its index names (64 per wave, the kernel's refill); the
    // launch that parked them put at most 64 per wave of ITS grid: this launch must not be narrower, or slices beyond
    // its last wave would depend on the claim cursor being asked often enough (workgroups_per_cu 4, then 1)
    if (carry_in && grid * kBlock < carry.bound) grid = (carry.bound + kBlock - 1) / kBlock;
    if (n_sets) {   // a workgroup serves one set: the same total, split evenly over the sets
        long long per_set = grid / n_sets;
        const long long for_rays = (p->tally_bundle + kBlock - 1) / kBlock;
        if (per_set > for_rays) per_set = for_rays;
        if (per_set < 1) per_set = 1;
        grid = per_set * n_sets;
        a.set_size = (unsigned int)p->tally_bundle;
        a.wgs_per_set = (int)per_set;
        a.set_stride_i = p->tally_stride_i64;
        a.set_stride_d = p->tally_stride_f64;
    }
    s->last_grid = (int)grid;
    s->last_lds = (int)lds;

    // carried photons
    bool carry_out = (p->flags & PVT_FLAG_CARRY_OUT) != 0 && !record && !n_sets;
    if (carry_in || carry_out) {
        if (!s->carry_cap) s->carry_cap = (unsigned int)((long long)s->num_cu * 4 * kBlock);
        if (grid * kBlock > (long long)s->carry_cap) carry_out = false;   // an unusually wide launch finishes its own photons
        const size_t bytes = (size_t)s->carry_cap * kCarryStride * 8;
        for (int q = 0; q < 2; q++)
            if (!carry.buf[q]) HIP_TRY(hipMalloc(&carry.buf[q], bytes));
        a.carry_cap = s->carry_cap;
        a.carry_in = carry.buf[carry.parity ^ 1];
        a.carry_out = carry.buf[carry.parity];
        a.carry_in_count = blocks + 4 * ((carry.phase + 2) % 3) + 2;
        a.carry_flags = (carry_in ? 1 : 0) | (carry_out ? 2 : 0);
    }

#if PVT_TIMELINE
    // developer build: the first kTlLaunches launches after PVT_TIMELINE_FROM each get a region of the stamp buffer;
    // nothing is synchronised (launches overlap as usual); pvt_timeline_dump() writes the file
    if (getenv("PVT_TIMELINE_FILE")) {
        if (!g_timeline) { (void)hipMalloc(&g_timeline, kTlBytes); (void)hipMemset(g_timeline, 0, kTlBytes); }
        const int from = getenv("PVT_TIMELINE_FROM") ? atoi(getenv("PVT_TIMELINE_FROM")) : 0;
        const int k = g_tl_launch++ - from;
        if (k >= 0 && k < kTlLaunches && grid <= kTlWgs) {
            a.timeline = g_timeline + (size_t)k * kTlWgs * kWaves * 8;
            g_tl_grid[k] = (int)grid; g_tl_n[k] = p->n_rays;
        }
    }
#endif
    const bool emit = rays == nullptr && s->d_ed != nullptr;
    if (emit) {
        const size_t need = (size_t)grid * kWaves * 7 * 64 * sizeof(double);
        if (s->emit_pool_bytes[(size_t)slot] < need) {
            // (growth is rare -- the grid of a stream's launches is stable -- and the launch before this one on the stream may
            // still be reading the old buffer)
            if (s->emit_pool[(size_t)slot]) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(s->emit_pool[(size_t)slot]); s->emit_pool[(size_t)slot] = nullptr; s->emit_pool_bytes[(size_t)slot] = 0; }
            const size_t bytes = need < ((size_t)s->num_cu * 4 * kWaves * 7 * 64 * sizeof(double)) ? (size_t)s->num_cu * 4 * kWaves * 7 * 64 * sizeof(double) : need;
            HIP_TRY(hipMalloc(&s->emit_pool[(size_t)slot], bytes));
            s->emit_pool_bytes[(size_t)slot] = bytes;
        }
        a.emit_pool = s->emit_pool[(size_t)slot];
    }
    hipError_t e;
    if (record) {
        e = tab_lds ? launch_seen<true, 1>(s->n_rec, emit, (int)grid, lds, st, a)
            : lp.small_lds ? launch_seen<true, 2>(s->n_rec, emit, (int)grid, lds, st, a)
                           : launch_seen<true, 0>(s->n_rec, emit, (int)grid, lds, st, a);
    } else {
        e = tab_lds ? launch_seen<false, 1>(s->n_rec, emit, (int)grid, lds, st, a)
            : lp.small_lds ? launch_seen<false, 2>(s->n_rec, emit, (int)grid, lds, st, a)
                           : launch_seen<false, 0>(s->n_rec, emit, (int)grid, lds, st, a);
    }
    if (e != hipSuccess) return fail(PVT_ERR_HIP, std::string("trace_kernel launch: ") + hipGetErrorString(e));
    if (!n_sets) carry.phase = (carry.phase + 1) % 3;
    if (carry_in || carry_out) {
        carry.pending = carry_out;
        carry.bound = carry_out ? grid * kBlock : 0;
        carry.maxsteps = p->maxsteps;
        carry.emit_method = p->emit_method;
        carry.parity ^= 1;
    }
#if PVT_STATS
    {
        unsigned long long c[32];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(c, a.cursor, 256, hipMemcpyDeviceToHost);
        fprintf(stderr, "[pvt stats] waves %llu  wave-iterations %llu (drain %llu)  lane-steps %llu (drain %llu)  "
                "mean live lanes/iter %.1f (bulk %.1f, drain %.1f)  iters/wave %.1f (drain %.1f)\n",
                c[5], c[1], c[3], c[2], c[4], (double)c[2] / c[1], (double)(c[2] - c[4]) / (double)(c[1] - c[3] + 1e-9),
                (double)c[4] / (c[3] + 1e-9), (double)c[1] / c[5], (double)c[3] / c[5]);
        fprintf(stderr, "[pvt stats] solo-wave cycles: refill %llu nodes %llu absorb %llu frame %llu trig %llu surface %llu tally %llu (solo waves %llu)\n",
                c[9], c[10], c[11], c[12], c[13], c[14], c[15], c[16]);
        const double bi = (double)(c[1] - c[3]) + 1e-9;
        if (a.lay.grid_d >= 0)
            fprintf(stderr, "[pvt stats] grid walk per wave-iteration: cell rounds %.2f  node-test trips %.2f  (lanes per trip %.1f, lanes per cell round %.1f)\n",
                    (double)c[25] / c[1], (double)c[26] / c[1], (double)c[27] / (c[26] + 1e-9), (double)c[28] / (c[25] + 1e-9));
        if (a.bvh)
            fprintf(stderr, "[pvt stats] mesh walk per wave-iteration: box trips %.2f (lanes per trip %.1f)  triangle trips %.2f (lanes per trip %.1f);  per lane-step: boxes %.2f triangles %.2f\n",
                    (double)c[25] / c[1], (double)c[26] / (c[25] + 1e-9), (double)c[27] / c[1], (double)c[28] / (c[27] + 1e-9),
                    (double)c[26] / c[2], (double)c[28] / c[2]);
        fprintf(stderr, "[pvt stats] bulk lanes/iteration: live %.1f absorbed %.1f re-emitted %.1f surface %.1f exit %.1f terminal-selector %.1f terminal %.1f\n",
                c[17] / bi, c[18] / bi, c[19] / bi, c[20] / bi, c[21] / bi, c[22] / bi, c[23] / bi);
    }
#endif
    return PVT_OK;
}

#if PVT_TIMELINE
void timeline_dump() {
    const char* path = getenv("PVT_TIMELINE_FILE");
    if (!path || !g_timeline) return;
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> host(kTlBytes / 8);
    (void)hipMemcpy(host.data(), g_timeline, kTlBytes, hipMemcpyDeviceToHost);
    if (FILE* fp = fopen(path, "wb")) {
        for (int k = 0; k < kTlLaunches; k++) {
            if (!g_tl_grid[k]) continue;
            unsigned long long head[4] = {0xABCDull, (unsigned long long)k, (unsigned long long)g_tl_grid[k], (unsigned long long)g_tl_n[k]};
            fwrite(head, 8, 4, fp);
            fwrite(host.data() + (size_t)k * kTlWgs * kWaves * 8, 8, (size_t)g_tl_grid[k] * kWaves * 8, fp);
        }
        fclose(fp);
    }
}
#endif

int unpack_launch(const unsigned long long* rows, const int* counts, long long n_recorded, int max_events,
                  const PvtEventLog* out, bool prefill, hipStream_t st) {
    if (n_recorded <= 0) return PVT_OK;
    const int rays_per_block = max_events >= kBlock ? 1 : kBlock / max_events;
    const long long gx = (n_recorded + rays_per_block - 1) / rays_per_block;
    const int gy = rays_per_block > 1 ? 1 : (max_events + kBlock - 1) / kBlock;
    if (gx > 0x7fffffffLL) return fail(PVT_ERR_INVALID, "too many recorded rays for one unpack launch");
    hipLaunchKernelGGL(unpack_log_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, st, rows, counts, *out,
                       n_recorded, max_events, rays_per_block, prefill ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PVT_ERR_HIP, std::string("unpack_log_kernel launch: ") + hipGetErrorString(e));
    return PVT_OK;
}

}  // namespace

extern "C" {

int pvt_trace_device_records(PvtScene* s, const PvtRays* rays, const PvtTraceParams* p, const PvtTallies* tl,
                             const PvtEventRecords* rec, void* stream) {
    int rc = check_trace_args(s, rays, p, tl);
    if (rc != PVT_OK) return rc;
    if (p->record_every > 0 && (!rec || !rec->rows || !rec->counts))
        return fail(PVT_ERR_INVALID, "record_every > 0 needs event records");
    return trace_launch(s, rays, p, tl, rec ? reinterpret_cast<unsigned long long*>(rec->rows) : nullptr, rec ? rec->counts : nullptr,
                        reinterpret_cast<hipStream_t>(stream));
}

int pvt_scene_carry_pending(PvtScene* s, void* stream) {
    if (!s) return 0;
    std::lock_guard<std::mutex> lock(s->slot_mutex);
    for (size_t k = 0; k < s->slot_of.size(); k++)
        if (s->slot_of[k] == reinterpret_cast<hipStream_t>(stream)) return s->carry[k].pending ? 1 : 0;
    return 0;
}

int pvt_scene_carry_discard(PvtScene* s, void* stream) {
    if (!s) return fail(PVT_ERR_INVALID, "null scene");
    std::lock_guard<std::mutex> lock(s->slot_mutex);
    for (size_t k = 0; k < s->slot_of.size(); k++)
        if (s->slot_of[k] == reinterpret_cast<hipStream_t>(stream)) {
            // the parked photons are simply forgotten: the next launch on the stream starts from its own rays (the cursor
            // blocks keep rotating; the count the abandoned launch left behind is never read)
            s->carry[k].pending = false;
            s->carry[k].bound = 0;
        }
    return PVT_OK;
}

int pvt_scene_trim(PvtScene* s) {
    if (!s) return fail(PVT_ERR_INVALID, "null scene");
    std::lock_guard<std::mutex> lock(s->slot_mutex);
    bool any = false;
    for (auto* b : s->stage) any = any || b != nullptr;
    if (any) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipDeviceSynchronize());   // (an unpack pass may still read a staging buffer)
        for (size_t k = 0; k < s->stage.size(); k++)
            if (s->stage[k]) { (void)hipFree(s->stage[k]); s->stage[k] = nullptr; s->stage_bytes[k] = 0; }
    }
    for (auto& c : s->carry) { c.pending = false; c.bound = 0; }   // a scene put aside carries nobody's photons
    return PVT_OK;
}

int pvt_unpack_records_device(const PvtEventRecords* rec, int64_t n_recorded, int32_t max_events,
                              const PvtEventLog* out, int prefill, void* stream) {
    if (!rec || !out || !rec->rows || !rec->counts || n_recorded < 0 || max_events < 1)
        return fail(PVT_ERR_INVALID, "bad unpack arguments");
    return unpack_launch(reinterpret_cast<const unsigned long long*>(rec->rows), rec->counts, n_recorded, max_events, out, prefill != 0,
                         reinterpret_cast<hipStream_t>(stream));
}

// Column arrays on the device: the records are staged in a buffer of the scene (one per stream) and unpacked by
// a second kernel, which also writes the reference's fill values into the rows no event reached.  A log too
// large for the staging limit is traced in several launches over consecutive ray ranges (each a multiple of
// record_every, so the rows and RNG streams are those of the single launch).
int pvt_trace_device(PvtScene* s, const PvtRays* rays, const PvtTraceParams* p, const PvtTallies* tl,
                     const PvtEventLog* log, void* stream) {
    int rc = check_trace_args(s, rays, p, tl);
    if (rc != PVT_OK) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (p->record_every <= 0) return trace_launch(s, rays, p, tl, nullptr, nullptr, st);
    if (!log) return fail(PVT_ERR_INVALID, "record_every > 0 needs an event log");
    if (p->n_rays == 0) return PVT_OK;
    HIP_TRY(hipSetDevice(s->device));
    const int slot = slot_of_stream(s, st);
    if (slot < 0) return fail(PVT_ERR_INVALID, "more than 64 HIP streams are tracing this scene; create one scene per group of streams");
    const long long every = p->record_every, me = p->max_events;
    const long long nrec = (p->n_rays + every - 1) / every;
    const size_t row_bytes = (size_t)kRecWords * 8;
    // recorded rays per launch under the staging limit
    long long per_launch = (long long)(s->stage_limit / (row_bytes * (size_t)me));
    if (per_launch < 1) per_launch = 1;
    if (per_launch > nrec) per_launch = nrec;
    const size_t need = (size_t)per_launch * (size_t)me * row_bytes;
    if (s->stage_bytes[slot] < need) {
        // (a stream-ordered free would do; growth is rare and the buffer may still be read by an unpack in flight)
        if (s->stage[slot]) { HIP_TRY(hipStreamSynchronize(st)); (void)hipFree(s->stage[slot]); s->stage[slot] = nullptr; s->stage_bytes[slot] = 0; }
        HIP_TRY(hipMalloc(&s->stage[slot], need));
        s->stage_bytes[slot] = need;
    }
    const bool prefill = !(p->flags & PVT_FLAG_NO_LOG_PREFILL);
    for (long long j0 = 0; j0 < nrec; j0 += per_launch) {
        const long long j1 = j0 + per_launch < nrec ? j0 + per_launch : nrec;
        const long long r0 = j0 * every, r1 = j1 * every < p->n_rays ? j1 * every : p->n_rays;
        PvtTraceParams q = *p;
        q.n_rays = r1 - r0;
        q.ray_offset = p->ray_offset + (uint64_t)r0;
        PvtRays sub{};
        if (rays) sub = PvtRays{rays->position + 3 * r0, rays->direction + 3 * r0, rays->wavelength + r0};
        rc = trace_launch(s, rays ? &sub : nullptr, &q, tl, s->stage[slot], log->counts + j0, st);
        if (rc != PVT_OK) return rc;
        const long long row0 = j0 * me;
        PvtEventLog out{log->counts + j0, log->kind + row0, log->hit + row0, log->container + row0, log->adjacent + row0,
                        log->component + row0, log->source + row0, log->position + 3 * row0, log->direction + 3 * row0,
                        log->normal + 3 * row0, log->wavelength + row0, log->travelled + row0, log->duration + row0};
        if (prefill) {
            // the fill values go down as plain memsets (full-width stores: measured 1.6 x the rate of filling from
            // the unpack kernel's per-row stores); the unpack pass then writes the rows that hold events
            const size_t rows = (size_t)(j1 - j0) * (size_t)me;
            HIP_TRY(hipMemsetAsync(out.kind, 0, rows, st));
            HIP_TRY(hipMemsetAsync(out.hit, 0xFF, rows * 4, st));
            HIP_TRY(hipMemsetAsync(out.container, 0xFF, rows * 4, st));
            HIP_TRY(hipMemsetAsync(out.adjacent, 0xFF, rows * 4, st));
            HIP_TRY(hipMemsetAsync(out.component, 0xFF, rows * 4, st));
            HIP_TRY(hipMemsetAsync(out.source, 0xFF, rows * 4, st));
            HIP_TRY(hipMemsetAsync(out.position, 0, rows * 24, st));
            HIP_TRY(hipMemsetAsync(out.direction, 0, rows * 24, st));
            HIP_TRY(hipMemsetAsync(out.normal, 0, rows * 24, st));
            HIP_TRY(hipMemsetAsync(out.wavelength, 0, rows * 8, st));
            HIP_TRY(hipMemsetAsync(out.travelled, 0, rows * 8, st));
            HIP_TRY(hipMemsetAsync(out.duration, 0, rows * 8, st));
        }
        rc = unpack_launch(s->stage[slot], log->counts + j0, j1 - j0, (int)me, &out, false, st);
        if (rc != PVT_OK) return rc;
    }
    return PVT_OK;
}

int pvt_emit_device(PvtScene* s, const PvtTraceParams* p, double* position, double* direction,
                    double* wavelength, void* stream) {
    if (!s || !p || !s->d_ed) return fail(PVT_ERR_INVALID, "scene has no emitter");
    if (p->n_rays <= 0) return PVT_OK;
    if (p->n_rays > 0x7fffffffLL) return fail(PVT_ERR_INVALID, "n_rays must fit in 31 bits per bundle");
    HIP_TRY(hipSetDevice(s->device));
    KArgs a = base_args(s, p);
    int grid = (int)((p->n_rays + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(emit_kernel, dim3(grid), dim3(kBlock), 0, reinterpret_cast<hipStream_t>(stream), a,
                       position, direction, wavelength);
    HIP_TRY(hipGetLastError());
    return PVT_OK;
}

int pvt_selftest_math(int fn, const double* x_host, double* y_host, int64_t n, int device) {
    if (!x_host || !y_host || n < 0) return fail(PVT_ERR_INVALID, "bad argument");
    if (n == 0) return PVT_OK;
    if (pvt_device_count() <= device) return fail(PVT_ERR_NO_DEVICE, "no such HIP device");
    HIP_TRY(hipSetDevice(device));
    double *dx = nullptr, *dy = nullptr;
    HIP_TRY(hipMalloc(&dx, (size_t)n * 8));
    HIP_TRY(hipMalloc(&dy, (size_t)n * 8));
    HIP_TRY(hipMemcpy(dx, x_host, (size_t)n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(math_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, nullptr, fn, dx, dy,
                       (long long)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(y_host, dy, (size_t)n * 8, hipMemcpyDeviceToHost));
    (void)hipFree(dx);
    (void)hipFree(dy);
    return PVT_OK;
}

int pvt_scene_counters(PvtScene* s, uint64_t* out, int reset) {
    if (!s || !out) return fail(PVT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    unsigned long long rows[64 * 8];
    HIP_TRY(hipMemcpy(rows, s->d_counters, sizeof rows, hipMemcpyDeviceToHost));   // (orders after the launches on the null stream only)
    for (int k = 0; k < 4; k++) out[k] = 0;
    for (int r = 0; r < 64; r++)
        for (int k = 0; k < 4; k++) out[k] += rows[r * 8 + k];
    if (reset) HIP_TRY(hipMemset(s->d_counters, 0, sizeof rows));
    return PVT_OK;
}

int pvt_scene_clock(PvtScene* s, uint64_t* out) {
    if (!s || !out) return fail(PVT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    unsigned long long rows[64 * 8];
    HIP_TRY(hipMemcpy(rows, s->d_counters, sizeof rows, hipMemcpyDeviceToHost));
    out[0] = out[1] = 0;
    for (int r = 0; r < 64; r++) { out[0] += rows[r * 8 + 4]; out[1] += rows[r * 8 + 5]; }
    return PVT_OK;
}

int pvt_scene_launch_span(PvtScene* s, void* stream, uint64_t* out) {
    if (!s || !out) return fail(PVT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    out[0] = out[1] = 0;
    size_t k = 0;
    {
        std::lock_guard<std::mutex> lock(s->slot_mutex);
        while (k < s->slot_of.size() && s->slot_of[k] != reinterpret_cast<hipStream_t>(stream)) k++;
        if (k == s->slot_of.size()) return fail(PVT_ERR_INVALID, "no launch on this stream yet");
    }
    HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
    HIP_TRY(hipMemcpy(out, s->d_stamp + 2 * k, 16, hipMemcpyDeviceToHost));
    return PVT_OK;
}

int pvt_scene_launch_info(PvtScene* s, int32_t* grid, int32_t* block, int32_t* lds_bytes) {
    if (!s) return fail(PVT_ERR_INVALID, "null scene");
    if (grid) *grid = s->last_grid;
    if (block) *block = kBlock;
    if (lds_bytes) *lds_bytes = s->last_lds;
    return PVT_OK;
}

int pvt_mesh_bvh_check(const PvtSceneTables* t, int32_t node, int32_t* n_bvh_nodes, int32_t* n_leaves,
                       int32_t* depth_out) {
    if (!t || node < 0 || node >= t->n_nodes || t->geom_type[node] != PVT_GEOM_MESH)
        return fail(PVT_ERR_INVALID, "not a mesh node");
    const int f0 = t->mesh_face_start[node], fc = t->mesh_face_count[node];
    if (fc <= 0 || f0 < 0 || f0 + fc > t->n_mesh_faces) return fail(PVT_ERR_INVALID, "mesh face range out of bounds");
    std::vector<pvt::BvhNode> nodes;
    std::vector<pvt::MeshTri> tris;
    double centre[3];
    const int root = pvt::BvhBuilder(t->mesh_vertices, t->mesh_faces, t->mesh_normals, nodes, tris).add_mesh(f0, fc, centre);
    if (root != 0 || nodes.size() < 2 || nodes[0].skip != (int)nodes.size()) return fail(PVT_ERR_INVALID, "root skip link");
    std::vector<int> seen(fc, 0);
    int leaves = 0, records = 0, max_depth = 0;
    // from the root down (an explicit stack of {record, parent, depth, the skip link it must carry})
    struct Open { int id, parent, depth, skip; };
    std::vector<Open> todo{{0, -1, 1, (int)nodes.size()}};
    std::vector<char> visited(nodes.size(), 0);
    while (!todo.empty()) {
        const Open o = todo.back();
        todo.pop_back();
        if (o.id < 0 || o.id >= (int)nodes.size() || visited[o.id]++) return fail(PVT_ERR_INVALID, "child link out of range or shared");
        const pvt::BvhNode& b = nodes[o.id];
        records += 1;
        max_depth = std::max(max_depth, o.depth);
        if (b.skip != o.skip) return fail(PVT_ERR_INVALID, "skip link: a left child's is its sibling, a right child's its parent's");
        for (int a = 0; a < 3; a++) {
            if (!(b.lo[a] <= b.hi[a])) return fail(PVT_ERR_INVALID, "empty box");
            if (o.parent >= 0 && (b.lo[a] < nodes[o.parent].lo[a] || b.hi[a] > nodes[o.parent].hi[a]))
                return fail(PVT_ERR_INVALID, "child box not inside its parent");
        }
        if (b.link < 0) {
            leaves += 1;
            const int tri = b.link & pvt::kIndexMask;
            if (tri >= (int)tris.size()) return fail(PVT_ERR_INVALID, "triangle record out of range");
            const pvt::MeshTri& tr = tris[tri];
            const long long local = tr.face - f0;
            if (local < 0 || local >= fc || seen[local]++) return fail(PVT_ERR_INVALID, "face missing or duplicated");
            for (int c = 0; c < 3; c++)
                for (int a = 0; a < 3; a++) {
                    if (tr.v[3 * c + a] != t->mesh_vertices[3 * (size_t)t->mesh_faces[3 * (size_t)tr.face + c] + a])
                        return fail(PVT_ERR_INVALID, "gathered vertex differs from the table");
                    if (tr.v[3 * c + a] - centre[a] < b.lo[a] || tr.v[3 * c + a] - centre[a] > b.hi[a])
                        return fail(PVT_ERR_INVALID, "triangle outside its leaf box");
                }
        } else {
            const int c = b.link;
            if (c <= o.id || (c & 1) != 0) return fail(PVT_ERR_INVALID, "children not a pair on one 64-byte line after their parent");
            todo.push_back({c + 1, o.id, o.depth + 1, o.skip});
            todo.push_back({c, o.id, o.depth + 1, c + 1});
        }
    }
    for (int k = 0; k < fc; k++) if (seen[k] != 1) return fail(PVT_ERR_INVALID, "face missing from the tree");
    if (records + 1 != (int)nodes.size() || visited[1]) return fail(PVT_ERR_INVALID, "records outside the tree");   // (one unused record follows the root)
    // the copy of the top levels for LDS (pvt_bvh.h: stage_top), at several budgets: a walk through the cursors that hits
    // every box must name the records the plain walk names, in its order, and a miss must lead where the plain skip link leads
    auto replay = [](const std::vector<pvt::BvhNode>& plain, const std::vector<pvt::BvhNode>& staged,
                     const std::vector<pvt::BvhNode>& top, int root) -> const char* {
        const int end = plain[root].skip;
        std::vector<int> cursor_of((size_t)(end - root) + 1, -1);
        cursor_of[(size_t)(end - root)] = end;
        std::vector<int> order;
        int c = pvt::first_cursor(staged, root);
        for (int i = root; i != end; i = pvt::next_cursor(plain[i], true)) {
            if (c == end) return "staged walk ends early";
            if ((c & pvt::kTopFlag) ? (size_t)(c & pvt::kIndexMask) >= top.size() : (c < root || c >= end)) return "cursor out of range";
            if (cursor_of[(size_t)(i - root)] >= 0) return "plain walk visits a record twice";
            cursor_of[(size_t)(i - root)] = c;
            order.push_back(i);
            const pvt::BvhNode& rec = pvt::at_cursor(staged, top, c);
            for (int a = 0; a < 3; a++)
                if (rec.lo[a] != plain[i].lo[a] || rec.hi[a] != plain[i].hi[a]) return "staged walk out of order";
            if ((plain[i].link < 0) != (rec.link < 0) || (plain[i].link < 0 && rec.link != plain[i].link)) return "staged leaf differs";
            c = pvt::next_cursor(rec, true);
            if (order.size() > (size_t)(end - root)) return "plain walk does not end";
        }
        if (c != end) return "staged walk does not end at the tree's end";
        for (int i : order) {
            const pvt::BvhNode& rec = pvt::at_cursor(staged, top, cursor_of[(size_t)(i - root)]);
            if (pvt::next_cursor(rec, false) != cursor_of[(size_t)(plain[i].skip - root)]) return "staged skip link differs";
        }
        return nullptr;
    };
    for (size_t budget : {(size_t)3, (size_t)7, (size_t)64, (size_t)768, nodes.size()}) {
        std::vector<pvt::BvhNode> staged = nodes, top;
        pvt::stage_top(staged, std::vector<int>{0}, budget, top);
        if (top.size() > budget) return fail(PVT_ERR_INVALID, "copy of the top levels exceeds its budget");
        if (const char* what = replay(nodes, staged, top, 0)) return fail(PVT_ERR_INVALID, what);
    }
    // ... and ALL the scene's meshes staged together, as scene creation does (the budget is shared out, slots run on from
    // tree to tree): every tree still walks like its plain self
    {
        std::vector<pvt::BvhNode> all;
        std::vector<pvt::MeshTri> all_tris;
        std::vector<int> roots;
        for (int n = 0; n < t->n_nodes; n++)
            if (t->geom_type[n] == PVT_GEOM_MESH)
                roots.push_back(pvt::BvhBuilder(t->mesh_vertices, t->mesh_faces, t->mesh_normals, all, all_tris)
                                    .add_mesh(t->mesh_face_start[n], t->mesh_face_count[n]));
        for (size_t budget : {(size_t)0, (size_t)10, (size_t)100, (size_t)512, all.size()}) {
            std::vector<pvt::BvhNode> staged = all, top;
            pvt::stage_top(staged, roots, budget, top);
            if (top.size() > budget) return fail(PVT_ERR_INVALID, "shared copy of the top levels exceeds its budget");
            for (int r : roots)
                if (const char* what = replay(all, staged, top, r)) return fail(PVT_ERR_INVALID, what);
        }
    }
    if (n_bvh_nodes) *n_bvh_nodes = (int32_t)records;
    if (n_leaves) *n_leaves = leaves;
    if (depth_out) *depth_out = max_depth;
    return PVT_OK;
}

}  // extern "C"

namespace {

thread_local int g_last_multi_reduce = 0;   // 0 none yet, 1 host sum, 2 RCCL on the devices (pvt_last_multi_reduce)

// RCCL, loaded at run time (no link dependency: a process that already holds an RCCL -- PyTorch's -- keeps using
// that copy): ncclReduce of `counts[k]` elements of bufs[k][r] (rank r = entry r of `devs`) into rank 0, all in
// one group.
// Returns 0 = summed on the devices, 1 = RCCL unavailable (nothing enqueued: the caller may sum on the host),
// 2 = a call failed AFTER reduces had been enqueued (shard 0's buffers may hold partial sums: an error, no fall-back).
// Communicators are kept per device list (ncclCommInitAll costs far more than reducing a few KB).
int rccl_reduce_to_first(const std::vector<int>& devs, const std::vector<void*> (&bufs)[4], const size_t (&counts)[4],
                         const bool (&is_f64)[4], std::string* why) {
    typedef int (*InitAll)(void**, int, const int*);
    typedef int (*Reduce)(const void*, void*, size_t, int, int, int, void*, hipStream_t);
    typedef int (*Void)();
    static void* lib = nullptr;
    static InitAll init_all = nullptr; static Reduce reduce = nullptr; static Void group_start = nullptr, group_end = nullptr;
    static std::vector<std::pair<std::vector<int>, std::vector<void*>>> comm_cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    if (!lib) {
        const char* names[] = {getenv("PVT_RCCL_LIB"), "librccl.so", "librccl.so.1"};
        for (const char* nm : names) if (nm && !lib) lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);   // a copy already mapped
        for (const char* nm : names) if (nm && !lib) lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (!lib) { *why = std::string("dlopen(librccl): ") + dlerror(); return 1; }
        init_all = (InitAll)dlsym(lib, "ncclCommInitAll"); reduce = (Reduce)dlsym(lib, "ncclReduce");
        group_start = (Void)dlsym(lib, "ncclGroupStart"); group_end = (Void)dlsym(lib, "ncclGroupEnd");
    }
    if (!init_all || !reduce || !group_start || !group_end) { *why = "RCCL symbols missing"; return 1; }
    const int n = (int)devs.size();
    std::vector<void*>* comms = nullptr;
    for (auto& entry : comm_cache) if (entry.first == devs) comms = &entry.second;
    if (!comms) {
        std::vector<void*> fresh((size_t)n, nullptr);
        if (int e = init_all(fresh.data(), n, devs.data())) { *why = "ncclCommInitAll failed with " + std::to_string(e); return 1; }
        comm_cache.emplace_back(devs, fresh);   // (kept for the life of the process)
        comms = &comm_cache.back().second;
    }
    constexpr int kInt64 = 4, kFloat64 = 8, kSum = 0;   // rccl.h: ncclInt64, ncclFloat64, ncclSum
    if (group_start()) { *why = "ncclGroupStart failed"; return 1; }
    bool ok = true;
    for (int k = 0; k < 4 && ok; k++) {
        if (!counts[k]) continue;
        for (int r = 0; r < n && ok; r++) {
            (void)hipSetDevice(devs[(size_t)r]);
            if (reduce(bufs[k][(size_t)r], bufs[k][(size_t)r], counts[k], is_f64[k] ? kFloat64 : kInt64, kSum, 0, (*comms)[(size_t)r], nullptr)) ok = false;
        }
    }
    if (group_end()) ok = false;
    for (int r = 0; r < n; r++) {
        (void)hipSetDevice(devs[(size_t)r]);
        if (hipStreamSynchronize(nullptr) != hipSuccess) ok = false;
    }
    if (!ok) { *why = "an RCCL call failed after reduces had been enqueued"; return 2; }
    return 0;
}

constexpr size_t kChunkRays = 524288;   // rays per upload chunk of a host-buffer bundle (28 MB: ~0.5 ms of PCIe; a chunk costs ~70 us of host time)

// PVT_HOST_PHASES=1 (developer switch): where the milliseconds of a host-buffer call go, on stderr
struct PhaseClock {
    bool on = getenv("PVT_HOST_PHASES") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    std::string line;
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        char buf[64];
        snprintf(buf, sizeof buf, " %s %.3f", what, std::chrono::duration<double, std::milli>(now - last).count());
        line += buf;
        last = now;
    }
    void report(size_t n, size_t chunks) const {
        if (on) fprintf(stderr, "[pvt host phases] n %zu chunks %zu ms:%s\n", n, chunks, line.c_str());
    }
};

// The device block of a host-buffer call (rays + tallies) is kept for the next call on the device: freeing 56 MB and
// allocating them again is a quarter of a millisecond per call, an eighth of a 10^6-photon pvt_trace_bundle
// (profiles/r06_host_bundle.txt).  ONE block per device, at most kArenaKeep bytes, handed to one call at a time (a call
// that finds it taken, or too small, allocates its own); pvt_release_cached_memory() frees what is kept.
constexpr size_t kArenaKeep = (size_t)1 << 30;
struct ArenaCache {
    struct Kept { int device; void* ptr; size_t bytes; };
    static std::mutex& mutex() { static std::mutex m; return m; }
    static std::vector<Kept>& kept() { static std::vector<Kept> k; return k; }
    static void* take(int device, size_t bytes, size_t* got) {
        std::lock_guard<std::mutex> lock(mutex());
        auto& k = kept();
        for (size_t i = 0; i < k.size(); i++)
            if (k[i].device == device && k[i].bytes >= bytes) {
                void* p = k[i].ptr;
                *got = k[i].bytes;
                k.erase(k.begin() + (long)i);
                return p;
            }
        return nullptr;
    }
    // -> the block the caller must free itself (the one handed in, or the smaller one it displaced), or null
    static void* give(int device, void* ptr, size_t bytes) {
        if (bytes > kArenaKeep || getenv("PVT_NO_HOST_CACHE")) return ptr;
        std::lock_guard<std::mutex> lock(mutex());
        auto& k = kept();
        for (size_t i = 0; i < k.size(); i++)
            if (k[i].device == device) {
                if (k[i].bytes >= bytes) return ptr;
                void* old = k[i].ptr;
                k[i] = Kept{device, ptr, bytes};
                return old;
            }
        k.push_back(Kept{device, ptr, bytes});
        return nullptr;
    }
    static void release_all() {
        std::vector<Kept> mine;
        {
            std::lock_guard<std::mutex> lock(mutex());
            mine.swap(kept());
        }
        for (auto& b : mine) { (void)hipSetDevice(b.device); (void)hipFree(b.ptr); }
    }
};

// The streams and events of one host-buffer call, taken from a per-device pool (creating and destroying four streams
// per call costs more than tracing a small bundle) and handed back by the destructor.
struct StreamSet {
    hipStream_t copy = nullptr, trace[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev0 = nullptr, ev1 = nullptr, arrived = nullptr;
    int n_trace = 0, device = -1;
    struct Pooled { int device; hipStream_t copy, trace[3]; hipEvent_t ev0, ev1, arrived; };
    static std::mutex& mutex() { static std::mutex m; return m; }
    static std::vector<Pooled>& pool() { static std::vector<Pooled> p; return p; }

    hipError_t open(int dev, int want_trace) {
        device = dev;
        n_trace = want_trace;
        {
            std::lock_guard<std::mutex> lock(mutex());
            auto& p = pool();
            for (size_t k = 0; k < p.size(); k++)
                if (p[k].device == dev) {
                    copy = p[k].copy; ev0 = p[k].ev0; ev1 = p[k].ev1; arrived = p[k].arrived;
                    for (int q = 0; q < 3; q++) trace[q] = p[k].trace[q];
                    p.erase(p.begin() + (long)k);
                    return hipSuccess;
                }
        }
        hipError_t e = hipStreamCreateWithFlags(&copy, hipStreamNonBlocking);
        for (int q = 0; q < 3 && e == hipSuccess; q++) e = hipStreamCreateWithFlags(&trace[q], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&ev0);
        if (e == hipSuccess) e = hipEventCreate(&ev1);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&arrived, hipEventDisableTiming);
        return e;
    }
    ~StreamSet() {
        if (!copy || !trace[0] || !trace[1] || !trace[2] || !ev0 || !ev1 || !arrived) return;   // (a failed open: leaked, once)
        // nothing of this call may still be running on streams the next call will be handed
        (void)hipStreamSynchronize(copy);
        for (int q = 0; q < 3; q++) (void)hipStreamSynchronize(trace[q]);
        std::lock_guard<std::mutex> lock(mutex());
        pool().push_back(Pooled{device, copy, {trace[0], trace[1], trace[2]}, ev0, ev1, arrived});
    }
};

// One host-buffer bundle on one device, in phases (pvt_trace_bundle runs them back to back; pvt_trace_bundle_multi
// runs one per device and can sum the tallies on the devices in between): upload + trace, fetch the tallies, fetch
// the event log.
struct HostBundle {
    PvtScene* scene = nullptr;
    std::vector<void*> bufs;
    const PvtSceneTables* tables = nullptr;
    PvtTraceParams params{};            // (a copy: a shard's parameters live as long as its bundle)
    const PvtTraceParams* p = &params;
    size_t nR = 0, nB = 0, R = 1, B = 1, n_sets = 1, si = 0, sd = 0;
    void *t_i[3] = {nullptr, nullptr, nullptr}, *t_d = nullptr;   // distinct | crossings | bins, sums (device)
    unsigned long long* rows = nullptr;                             // event records (device)
    int* counts = nullptr;
    size_t nrec = 0;
    double ms = 0.0;
    PhaseClock clock;
    size_t n_chunks = 1;
    char* arena = nullptr;              // tallies | rays | counts of the log, one block (ArenaCache)
    size_t arena_bytes = 0, tally_bytes = 0;
    int arena_device = -1;

    ~HostBundle() {
        if (scene) (void)hipSetDevice(scene->device);
        for (void* b : bufs) (void)hipFree(b);
        if (arena)
            if (void* mine = ArenaCache::give(arena_device, arena, arena_bytes)) (void)hipFree(mine);
        clock.mark("free");
        if (scene) pvt_scene_destroy(scene);
        clock.mark("scene-destroy");
        clock.report((size_t)params.n_rays, n_chunks);
    }
    hipError_t dalloc(size_t bytes, void** out) {
        hipError_t e = hipMalloc(out, bytes ? bytes : 8);
        if (e == hipSuccess) bufs.push_back(*out);
        return e;
    }
    hipError_t move(void* dst, const void* src, size_t pitch_elems, size_t width_elems, hipMemcpyKind kind) const {
        if (width_elems == 0) return hipSuccess;
        if (n_sets == 1) return hipMemcpy(dst, src, width_elems * 8, kind);
        return hipMemcpy2D(dst, pitch_elems * 8, src, pitch_elems * 8, width_elems * 8, n_sets, kind);
    }

    // scene + rays + tallies (seeded with `seed`, or zero) on `device`, trace enqueued and finished.
    //
    // The rays cross PCIe in CHUNKS and a chunk is traced while the next one is still on its way: the copies go down
    // one stream, each chunk's launch waits for its own copy and runs on one of three trace streams used in turn (the
    // drain tail of one launch -- a few long histories -- is covered by the bulk of the next), ray i keeps the
    // stream seed + ray_offset + i whatever chunk carries it, every launch adds into the same tallies with atomics and
    // writes the event-log rows of its own rays (inner chunk boundaries are multiples of record_every).  One upload
    // followed by one launch left the GPU idle for the whole upload: 56 MB at the link's ~50 GB/s is 1.1 ms, the trace
    // of those 10^6 photons 0.65 ms (profiles/r06_host_io.txt).
    int trace(const PvtSceneTables* tb, const PvtEmitterTables* emitter, const PvtRays* rays, const PvtTraceParams* pp,
              const PvtTallies* seed, bool want_log, int device) {
        tables = tb;
        params = *pp;
        params.flags &= ~(int64_t)PVT_FLAG_CARRY_OUT;   // a scene that lives for one call has no next launch to carry photons to
        int rc = pvt_scene_create(tables, device, &scene);
        if (rc != PVT_OK) return rc;
        if (emitter) {
            rc = pvt_scene_set_emitter(scene, emitter);
            if (rc != PVT_OK) return rc;
        }
        clock.mark("scene");
        const size_t n = (size_t)p->n_rays;
        nR = (size_t)tables->n_recorders; nB = (size_t)tables->total_bins;
        R = nR > 0 ? nR : 1; B = nB > 0 ? nB : 1;
        // tally sets (PvtTraceParams.tally_bundle): the device copies keep the caller's strides, slices move as
        // 2-D copies (one row per set)
        n_sets = p->tally_bundle > 0 ? (size_t)((p->n_rays + p->tally_bundle - 1) / p->tally_bundle) : 1;
        si = n_sets > 1 ? (size_t)p->tally_stride_i64 : 0; sd = n_sets > 1 ? (size_t)p->tally_stride_f64 : 0;
        if (n_sets > 1 && (si < R || si < B || sd < R * 8)) return fail(PVT_ERR_INVALID, "tally strides smaller than a set");
        if (p->record_every > 0 && !want_log) return fail(PVT_ERR_INVALID, "record_every > 0 needs an event log");
        // ONE allocation for the rays, the tallies and the counts of the log (allocating and freeing device memory is a
        // large part of a small call); the event records, which can be gigabytes, have their own
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t bytes_i[3] = {((n_sets - 1) * si + R) * 8, ((n_sets - 1) * si + R) * 8, ((n_sets - 1) * si + B) * 8};
        const size_t bytes_d = ((n_sets - 1) * sd + R * 8) * 8;
        nrec = p->record_every > 0 ? (size_t)((p->n_rays + p->record_every - 1) / p->record_every) : 0;
        tally_bytes = up(bytes_i[0]) + up(bytes_i[1]) + up(bytes_i[2]) + up(bytes_d);
        const size_t ray_bytes = rays ? up(n * 24) * 2 + up(n * 8) : 0;
        const size_t need = tally_bytes + ray_bytes + up(nrec * 4);
        arena_device = device;
        arena = (char*)ArenaCache::take(device, need, &arena_bytes);
        if (!arena) {
            HIP_TRY(hipMalloc((void**)&arena, need));
            arena_bytes = need;
        }
        char* at = arena;
        for (int k = 0; k < 3; k++) { t_i[k] = at; at += up(bytes_i[k]); }
        t_d = at; at += up(bytes_d);
        char *dp = nullptr, *dd = nullptr, *dw = nullptr;
        if (rays) { dp = at; at += up(n * 24); dd = at; at += up(n * 24); dw = at; at += up(n * 8); }
        if (nrec) {
            counts = (int*)at; at += up(nrec * 4);
            void* b1;
            HIP_TRY(dalloc(nrec * (size_t)p->max_events * kRecWords * 8, &b1));
            rows = (unsigned long long*)b1;
        }
        clock.mark("alloc");
        // The trace ADDS into the caller's tallies: the device copies start from them.  A caller who hands in zeros (the
        // reference's trace_bundle always starts from zero, _kernel.pyx:1035-1047) costs a fill on the first trace stream,
        // no transfer; the other trace streams wait for that fill.
        bool seeded = false;
        if (seed) {
            auto any = [&](const void* base, size_t stride_elems, size_t width_elems) {
                for (size_t j = 0; j < n_sets && !seeded; j++) {
                    const unsigned long long* w = (const unsigned long long*)base + j * stride_elems;
                    for (size_t q = 0; q < width_elems; q++)
                        if (w[q]) { seeded = true; break; }
                }
            };
            any(seed->rec_distinct, si, nR); any(seed->rec_crossings, si, nR);
            any(seed->rec_sums, sd, nR * 8); any(seed->rec_bins, si, nB);
        }
        if (seeded) {
            HIP_TRY(hipMemset(arena, 0, tally_bytes));
            HIP_TRY(move(t_i[0], seed->rec_distinct, si, nR, hipMemcpyHostToDevice));
            HIP_TRY(move(t_i[1], seed->rec_crossings, si, nR, hipMemcpyHostToDevice));
            HIP_TRY(move(t_d, seed->rec_sums, sd, nR * 8, hipMemcpyHostToDevice));
            HIP_TRY(move(t_i[2], seed->rec_bins, si, nB, hipMemcpyHostToDevice));
            HIP_TRY(hipDeviceSynchronize());   // (these ran on the null stream; the streams below do not wait for it)
        }
        PvtTallies dt{(int64_t*)t_i[0], (int64_t*)t_i[1], (double*)t_d, (int64_t*)t_i[2]};

        // chunks: about kChunkRays rays each, inner boundaries on multiples of record_every; a launch with tally sets
        // indexes its sets from its first ray and is not split
        size_t chunk = n;
        if (rays && n_sets == 1 && n > 0) {
            size_t want = n >= kChunkRays + kChunkRays / 2 ? kChunkRays : n;
            if (const char* env = getenv("PVT_HOST_CHUNK_RAYS")) {   // (tests: 0 = one upload and one launch, k = chunks of k rays)
                const long long k = atoll(env);
                want = k > 0 ? (size_t)k : n;
            }
            if (p->record_every > 1) {   // inner boundaries on multiples of record_every: a chunk's recorded rays are whole rows of the log
                const size_t re = (size_t)p->record_every;
                want = (want + re - 1) / re * re;
            }
            chunk = want < n ? want : n;
        }
        n_chunks = n ? (n + chunk - 1) / chunk : 1;
        StreamSet ss;
        HIP_TRY(ss.open(device, n_chunks > 1 ? 3 : 1));
        if (!seeded) {
            HIP_TRY(hipMemsetAsync(arena, 0, tally_bytes, ss.trace[0]));
            HIP_TRY(hipEventRecord(ss.arrived, ss.trace[0]));
            for (int k = 1; k < ss.n_trace; k++) HIP_TRY(hipStreamWaitEvent(ss.trace[k], ss.arrived, 0));
        }
        clock.mark("tallies");
        HIP_TRY(hipEventRecord(ss.ev0, ss.trace[0]));
        for (size_t k = 0; k < n_chunks && n > 0; k++) {   // (an empty bundle launches nothing)
            const size_t lo = k * chunk, hi = lo + chunk < n ? lo + chunk : n;
            hipStream_t st = ss.trace[k % ss.n_trace];
            PvtRays drays{};
            if (rays) {
                HIP_TRY(hipMemcpyAsync(dp + lo * 24, rays->position + lo * 3, (hi - lo) * 24, hipMemcpyHostToDevice, ss.copy));
                HIP_TRY(hipMemcpyAsync(dd + lo * 24, rays->direction + lo * 3, (hi - lo) * 24, hipMemcpyHostToDevice, ss.copy));
                HIP_TRY(hipMemcpyAsync(dw + lo * 8, rays->wavelength + lo, (hi - lo) * 8, hipMemcpyHostToDevice, ss.copy));
                HIP_TRY(hipEventRecord(ss.arrived, ss.copy));
                HIP_TRY(hipStreamWaitEvent(st, ss.arrived, 0));
                drays = PvtRays{(const double*)(dp + lo * 24), (const double*)(dd + lo * 24), (const double*)(dw + lo * 8)};
            }
            PvtTraceParams pk = params;
            pk.n_rays = (int64_t)(hi - lo);
            pk.ray_offset = params.ray_offset + (uint64_t)lo;
            PvtEventRecords rec{};
            if (nrec) {
                const size_t j0 = lo / (size_t)p->record_every;   // (lo is a multiple of record_every)
                rec = PvtEventRecords{counts + j0, reinterpret_cast<uint64_t*>(rows + j0 * (size_t)p->max_events * kRecWords)};
            }
            rc = pvt_trace_device_records(scene, rays ? &drays : nullptr, &pk, &dt, nrec ? &rec : nullptr, st);
            if (rc != PVT_OK) return rc;
        }
        clock.mark("enqueue");
        for (int k = 1; k < ss.n_trace; k++) {   // the last event waits for every trace stream
            HIP_TRY(hipEventRecord(ss.arrived, ss.trace[k]));
            HIP_TRY(hipStreamWaitEvent(ss.trace[0], ss.arrived, 0));
        }
        HIP_TRY(hipEventRecord(ss.ev1, ss.trace[0]));
        HIP_TRY(hipEventSynchronize(ss.ev1));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, ss.ev0, ss.ev1));
        ms = t;   // (several chunks: from the start of the upload to the end of the last launch)
        clock.mark("wait");
        return PVT_OK;
    }

    int fetch_tallies(const PvtTallies* out) {
        HIP_TRY(hipSetDevice(scene->device));
        if (n_sets == 1) {   // the four arrays lie side by side in the block: one transfer, four host copies
            std::vector<char> host(tally_bytes);
            HIP_TRY(hipMemcpy(host.data(), arena, tally_bytes, hipMemcpyDeviceToHost));
            std::memcpy(out->rec_distinct, host.data() + ((char*)t_i[0] - arena), nR * 8);
            std::memcpy(out->rec_crossings, host.data() + ((char*)t_i[1] - arena), nR * 8);
            std::memcpy(out->rec_sums, host.data() + ((char*)t_d - arena), nR * 8 * 8);
            std::memcpy(out->rec_bins, host.data() + ((char*)t_i[2] - arena), nB * 8);
            clock.mark("fetch-tallies");
            return PVT_OK;
        }
        HIP_TRY(move(out->rec_distinct, t_i[0], si, nR, hipMemcpyDeviceToHost));
        HIP_TRY(move(out->rec_crossings, t_i[1], si, nR, hipMemcpyDeviceToHost));
        HIP_TRY(move(out->rec_sums, t_d, sd, nR * 8, hipMemcpyDeviceToHost));
        HIP_TRY(move(out->rec_bins, t_i[2], si, nB, hipMemcpyDeviceToHost));
        clock.mark("fetch-tallies");
        return PVT_OK;
    }

    // The event log: the caller's column arrays get the reference's fill values on the host (memset / fill), the
    // records the rays wrote cross PCIe PACKED -- `counts[j]` rows of recorded ray j, gathered on the device -- and
    // are scattered into the columns by this thread.  (Moving the dense columns costs 117 bytes x max_events per
    // recorded ray over PCIe, of which a ray writes a dozen rows: 15 GB for the reference's README case.)
    int fetch_log(const PvtEventLog* log) {
        if (!(p->record_every > 0)) return PVT_OK;
        HIP_TRY(hipSetDevice(scene->device));
        const size_t me = (size_t)p->max_events, total_rows = nrec * me;
        HIP_TRY(hipMemcpy(log->counts, counts, nrec * 4, hipMemcpyDeviceToHost));
        std::vector<long long> first(nrec + 1, 0);
        for (size_t j = 0; j < nrec; j++) first[j + 1] = first[j] + log->counts[j];
        const size_t used = (size_t)first[nrec];
        std::memset(log->kind, 0, total_rows);
        std::memset(log->hit, 0xFF, total_rows * 4); std::memset(log->container, 0xFF, total_rows * 4);
        std::memset(log->adjacent, 0xFF, total_rows * 4); std::memset(log->component, 0xFF, total_rows * 4);
        std::memset(log->source, 0xFF, total_rows * 4);
        std::memset(log->position, 0, total_rows * 24); std::memset(log->direction, 0, total_rows * 24);
        std::memset(log->normal, 0, total_rows * 24);
        std::memset(log->wavelength, 0, total_rows * 8); std::memset(log->travelled, 0, total_rows * 8);
        std::memset(log->duration, 0, total_rows * 8);
        if (!used) return PVT_OK;
        void *d_first, *d_packed;
        HIP_TRY(dalloc((nrec + 1) * 8, &d_first));
        HIP_TRY(dalloc(used * kRecWords * 8, &d_packed));
        HIP_TRY(hipMemcpy(d_first, first.data(), (nrec + 1) * 8, hipMemcpyHostToDevice));
        const int rays_per_block = me >= (size_t)kBlock ? 1 : (int)((size_t)kBlock / me);
        const long long gx = ((long long)nrec + rays_per_block - 1) / rays_per_block;
        const int gy = rays_per_block > 1 ? 1 : (int)((me + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(pack_log_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kBlock), 0, nullptr, rows, counts,
                           (const long long*)d_first, (unsigned long long*)d_packed, (long long)nrec, (int)me, rays_per_block);
        HIP_TRY(hipGetLastError());
        std::vector<unsigned long long> packed(used * kRecWords);
        HIP_TRY(hipMemcpy(packed.data(), d_packed, used * kRecWords * 8, hipMemcpyDeviceToHost));
        for (size_t j = 0; j < nrec; j++) {
            const unsigned long long* r = packed.data() + (size_t)first[j] * kRecWords;
            size_t row = j * me;
            for (int k = 0; k < log->counts[j]; k++, r += kRecWords, row++) {
                log->hit[row] = (int32_t)(uint32_t)r[0]; log->container[row] = (int32_t)(uint32_t)(r[0] >> 32);
                log->adjacent[row] = (int32_t)(uint32_t)r[1]; log->component[row] = (int32_t)(uint32_t)(r[1] >> 32);
                log->source[row] = (int32_t)(uint32_t)r[2]; log->kind[row] = (uint8_t)(r[2] >> 32);
                std::memcpy(log->position + 3 * row, r + 3, 24);
                std::memcpy(log->direction + 3 * row, r + 6, 24);
                std::memcpy(log->normal + 3 * row, r + 9, 24);
                std::memcpy(log->wavelength + row, r + 12, 8);
                std::memcpy(log->travelled + row, r + 13, 8);
                std::memcpy(log->duration + row, r + 14, 8);
            }
        }
        return PVT_OK;
    }
};

}  // namespace

extern "C" {

// Host-buffer entry: the literal stand-in for _kernel.trace_bundle.
int pvt_trace_bundle(const PvtSceneTables* tables, const PvtEmitterTables* emitter, const PvtRays* rays,
                     const PvtTraceParams* p, const PvtTallies* tl, const PvtEventLog* log, int device,
                     double* kernel_ms) {
    if (!tables || !p || !tl) return fail(PVT_ERR_INVALID, "null argument");
    HostBundle hb;
    int rc = hb.trace(tables, emitter, rays, p, tl, log != nullptr, device);
    if (rc != PVT_OK) return rc;
    if (kernel_ms) *kernel_ms = hb.ms;
    rc = hb.fetch_tallies(tl);
    if (rc != PVT_OK) return rc;
    return hb.fetch_log(log);
}

void pvt_release_cached_memory(void) { ArenaCache::release_all(); }

int pvt_shard_range(int64_t n_rays, int shard, int n_shards, int64_t align, int64_t* start, int64_t* stop) {
    if (n_rays < 0 || n_shards <= 0 || shard < 0 || shard >= n_shards || !start || !stop)
        return fail(PVT_ERR_INVALID, "bad shard arguments");
    if (align < 1) align = 1;
    auto edge = [&](int r) -> int64_t {
        if (r <= 0) return 0;
        if (r >= n_shards) return n_rays;
        // n_rays * r fits: n_rays < 2^31 per bundle in practice, and __int128 keeps the general case exact
        const int64_t e = (int64_t)(((__int128)n_rays * r) / n_shards);
        return (e / align) * align;
    };
    *start = edge(shard);
    *stop = edge(shard + 1);
    return PVT_OK;
}

int pvt_trace_bundle_multi(const PvtSceneTables* tables, const PvtEmitterTables* emitter, const PvtRays* rays,
                           const PvtTraceParams* p, const PvtTallies* tl, const PvtEventLog* log,
                           const int* devices, int n_devices, double* kernel_ms) {
    if (!tables || !p || !tl || !devices || n_devices <= 0) return fail(PVT_ERR_INVALID, "null argument");
    if (p->record_every > 0 && !log) return fail(PVT_ERR_INVALID, "record_every > 0 needs an event log");
    const int ndev = pvt_device_count();
    for (int g = 0; g < n_devices; g++)
        if (devices[g] < 0 || devices[g] >= ndev) return fail(PVT_ERR_NO_DEVICE, "no such HIP device in the device list");
    const char* how = getenv("PVT_MULTI_REDUCE");   // "host": never RCCL; "rccl": also for a single device (self-test)
    const bool force_rccl = how && std::string(how) == "rccl", never_rccl = how && std::string(how) == "host";
    if (n_devices == 1 && !force_rccl) return pvt_trace_bundle(tables, emitter, rays, p, tl, log, devices[0], kernel_ms);
    // the shards below own ONE tally set each (R / B / R*8 elements) and shard edges ignore bundle boundaries
    if (p->tally_bundle > 0)
        return fail(PVT_ERR_INVALID, "tally_bundle is not supported over a device list; trace the group on one device");

    const size_t nR = (size_t)tables->n_recorders, nB = (size_t)tables->total_bins;
    struct Shard {
        int rc = PVT_OK;
        std::string error;
        bool traced = false;
        HostBundle hb;
    };
    std::vector<Shard> shards((size_t)n_devices);
    auto parallel = [&](auto&& body) {   // one host thread per device-list entry
        std::vector<std::thread> workers;
        for (int g = 0; g < n_devices; g++)
            workers.emplace_back([&, g]() {
                Shard& sh = shards[(size_t)g];
                if (sh.rc == PVT_OK) sh.rc = body(sh, g);
                if (sh.rc != PVT_OK && sh.error.empty()) sh.error = g_error;   // thread-local: carry it to the caller's thread
            });
        for (auto& w : workers) w.join();
    };
    auto first_error = [&]() -> int {
        for (int g = 0; g < n_devices; g++)
            if (shards[(size_t)g].rc != PVT_OK)
                return fail(shards[(size_t)g].rc, "shard " + std::to_string(g) + " on device " + std::to_string(devices[g]) + ": " + shards[(size_t)g].error);
        return PVT_OK;
    };
    // 1. every shard: its own scene, rays, zeroed tallies and event records on its device; traced to completion
    parallel([&](Shard& sh, int g) -> int {
        int64_t start = 0, stop = 0;
        int rc = pvt_shard_range(p->n_rays, g, n_devices, p->record_every, &start, &stop);
        if (rc != PVT_OK || stop <= start) return rc;
        PvtTraceParams q = *p;
        q.n_rays = stop - start;
        q.ray_offset = p->ray_offset + (uint64_t)start;
        PvtRays r{};
        if (rays) r = PvtRays{rays->position + 3 * start, rays->direction + 3 * start, rays->wavelength + start};
        // (the shard's params must outlive the bundle: keep a copy inside it)
        sh.traced = true;
        return sh.hb.trace(tables, emitter, rays ? &r : nullptr, &q, nullptr, log != nullptr, devices[g]);
    });
    int rc = first_error();
    // 2. the tallies: summed ON THE DEVICES with RCCL (one communicator per device-list entry, ncclReduce to the
    //    first shard) when every entry is a different GPU and the library can be loaded -- else on the host
    bool on_device = false;
    std::vector<int> live;
    for (int g = 0; g < n_devices; g++) if (shards[(size_t)g].traced) live.push_back(g);
    if (rc == PVT_OK && !never_rccl && !live.empty() && (live.size() > 1 || force_rccl)) {
        bool distinct = true;
        for (size_t x = 0; x < live.size(); x++)
            for (size_t y = x + 1; y < live.size(); y++)
                if (devices[live[x]] == devices[live[y]]) distinct = false;
        if (distinct) {
            std::vector<int> devs;
            std::vector<void*> bufs[4];
            for (int g : live) {
                devs.push_back(devices[g]);
                HostBundle& hb = shards[(size_t)g].hb;
                bufs[0].push_back(hb.t_i[0]); bufs[1].push_back(hb.t_i[1]); bufs[2].push_back(hb.t_i[2]); bufs[3].push_back(hb.t_d);
            }
            const size_t counts[4] = {nR, nR, nB, nR * 8};
            const bool is_f64[4] = {false, false, false, true};
            std::string why;
            const int how_it_went = rccl_reduce_to_first(devs, bufs, counts, is_f64, &why);
            on_device = how_it_went == 0;
            if (how_it_went == 2) rc = fail(PVT_ERR_HIP, "summing the shards' tallies with RCCL failed midway (" + why + "); the first shard's buffers "
                                                        "may hold partial sums, so nothing is returned");
            else if (!on_device && force_rccl) rc = fail(PVT_ERR_HIP, "RCCL reduce requested (PVT_MULTI_REDUCE=rccl) but unavailable: " + why);
        }
    }
    g_last_multi_reduce = on_device ? 2 : 1;
    // 3. tallies to the host (one shard's after a device-side sum, every shard's otherwise), added to the caller's
    if (rc == PVT_OK) {
        std::vector<int64_t> distinct(nR ? nR : 1), crossings(nR ? nR : 1), bins(nB ? nB : 1);
        std::vector<double> sums(nR ? nR * 8 : 1);
        for (size_t x = 0; x < live.size() && rc == PVT_OK; x++) {
            if (on_device && x > 0) break;
            PvtTallies t{distinct.data(), crossings.data(), sums.data(), bins.data()};
            rc = shards[(size_t)live[x]].hb.fetch_tallies(&t);
            if (rc != PVT_OK) break;
            for (size_t i = 0; i < nR; i++) { tl->rec_distinct[i] += distinct[i]; tl->rec_crossings[i] += crossings[i]; }
            for (size_t i = 0; i < nR * 8; i++) tl->rec_sums[i] += sums[i];
            for (size_t i = 0; i < nB; i++) tl->rec_bins[i] += bins[i];
        }
    }
    // 4. each shard's rows of the event log, into the caller's arrays (inner shard edges are multiples of record_every)
    if (rc == PVT_OK && p->record_every > 0) {
        parallel([&](Shard& sh, int g) -> int {
            if (!sh.traced) return PVT_OK;
            int64_t start = 0, stop = 0;
            (void)pvt_shard_range(p->n_rays, g, n_devices, p->record_every, &start, &stop);
            const int64_t j0 = start / p->record_every, row0 = j0 * p->max_events;
            PvtEventLog l{log->counts + j0, log->kind + row0, log->hit + row0, log->container + row0,
                          log->adjacent + row0, log->component + row0, log->source + row0,
                          log->position + 3 * row0, log->direction + 3 * row0, log->normal + 3 * row0,
                          log->wavelength + row0, log->travelled + row0, log->duration + row0};
            return sh.hb.fetch_log(&l);
        });
        rc = first_error();
    }
    double longest = 0.0;
    for (int g : live) {
        if (shards[(size_t)g].hb.ms > longest) longest = shards[(size_t)g].hb.ms;
    }
    if (kernel_ms) *kernel_ms = longest;
    return rc;
}

int pvt_last_multi_reduce(void) { return g_last_multi_reduce; }

int pvt_node_grid_plan(const PvtSceneTables* t, int32_t* dims, double* lo, double* cell, double* guard, int32_t* odd,
                       uint64_t* masks, int64_t masks_cap) {
    if (!t || !dims || t->n_nodes <= 0 || t->n_nodes > PVT_MAX_NODES) return fail(PVT_ERR_INVALID, "bad argument");
    NodeGrid g;
    const bool on = plan_node_grid(t, &g);
    for (int a = 0; a < 3; a++) {
        dims[a] = on ? g.n[a] : 0;
        if (lo) lo[a] = on ? g.lo[a] : 0.0;
        if (cell) cell[a] = on ? g.cell[a] : 0.0;
    }
    if (guard) *guard = on ? g.guard : 0.0;
    if (odd) *odd = on && g.odd ? 1 : 0;
    if (on && masks) {
        const size_t cells = (size_t)g.n[0] * g.n[1] * g.n[2];
        for (size_t c = 0; c < cells; c++)
            for (int w = 0; w < 2; w++)
                if ((int64_t)(c * 2 + w) < masks_cap) masks[c * 2 + w] = w < g.words ? g.masks[c * g.words + w] : 0ull;
    }
    return PVT_OK;
}

}  // extern "C"
