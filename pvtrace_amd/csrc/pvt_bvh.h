// Triangle BVH of PVT_GEOM_MESH nodes: host-side construction and the walk both the trace kernel (per lane) and
// the host-side self-check run.
//
// A photon needs EVERY forward crossing of a mesh (the container rule counts them, _kernel.pyx:684-714), so
// front-to-back ordering buys nothing; what a per-lane walk on the GPU pays for is the CHAIN OF DEPENDENT LOADS from
// the root to a leaf (a 327 680-face ball: 21 levels of a binary tree, each a trip to L2 or further) and registers.
// Hence:
//   * 4-wide nodes (a binary surface-area-heuristic tree with its levels collapsed in pairs): one 128-byte line
//     holds the boxes of four children, half as many levels;
//   * nodes in depth-first order with a `skip` link (first node after the subtree), as before, so there is no
//     per-lane stack in scratch or LDS -- and a 64-bit TRAIL instead: 4 bits per level remember which children of
//     the current ancestors were hit, so a node reached by falling off the end of its elder sibling's subtree knows
//     from its own header (level, slot in parent) whether its parent's test let it in, without re-reading the parent;
//   * a hit child that is a leaf is tested right away (leaves are slots of their parent, not nodes); of the hit
//     inner children the walk jumps to the first, the others are met in depth-first order;
//   * triangles split hot / cold: the three vertices (72 bytes, read by every test) apart from the face normal and
//     the face id (32 bytes, read for the crossings that are found).
// Boxes are f32 rounded OUTWARDS after padding by 1e-7 of the mesh diagonal: culling is only a filter and must be
// conservative with respect to the (differently rounded) watertight f64 triangle test, so that the set of crossings
// found through the tree equals the set a brute-force loop over the faces finds (which is what oracle/pvt_oracle.c
// does) bit for bit.  Which tree is built never changes a result: crossings are ordered by (t, face).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define PVT_BVH_HD __host__ __device__ __forceinline__
#else
#define PVT_BVH_HD inline
#endif

namespace pvt {

struct BvhNode {          // 128 bytes = one cache line
    float lo[4][3], hi[4][3];   // the children's boxes; an empty slot has lo = +inf, hi = -inf
    int child[4];         // > 0: index of the child's node;  < 0: leaf, -(1 + ((first triangle << 4) | count));  0: empty
    int skip;             // first node after this node's subtree
    short level, slot;    // depth (root 0, at most kMaxLevels - 1) and slot in the parent
    int pad[2];
};
static_assert(sizeof(BvhNode) == 128, "one node per cache line");
struct MeshTri {          // hot: 72 bytes
    double v[9];          // three vertices, node-local frame
};
struct MeshTriCold {      // 32 bytes
    double n[3];          // outward unit face normal
    long long face;       // index in the scene's pooled face table (tie-break key, diagnostics)
};
constexpr int kMaxLevels = 16;   // 4 trail bits each in a 64-bit word

// Leaf size: the watertight triangle test costs several box tests, so large meshes get one
// triangle per leaf (measured on MI355X: 20 480-face ball 3.2 ms vs 4.5 ms with 4 per leaf);
// for a handful of faces the tree is not worth walking and leaves hold up to 8.
constexpr int kSmallMesh = 32, kSmallLeaf = 8, kLargeLeaf = 1;

// Slab test of one child box.  Box culling only has to be conservative (a false hit costs a triangle test, a false
// miss would lose a crossing).  `minv` = 1/d per axis, a huge finite number for a direction component below 1e-300:
// inside the slab the two plane distances then have opposite signs (interval covers everything), outside the same
// sign (pushed out of range, or a harmless false hit), and 0 * inf = NaN can never arise.
// `om` = oo * minv per axis: the plane distances are ONE fused multiply-add each, (plane * minv) - om.  The fused
// rounding differs from (plane - o) * minv in the last bits only, which the padding of the boxes covers a billion
// times over -- culling is the one place of the engine where the last bit cannot matter.
PVT_BVH_HD bool bvh_box_hit(const float* lo, const float* hi, const double* om, const double* minv) {
    double tmin = -INFINITY, tmax = INFINITY;
    for (int a = 0; a < 3; a++) {
        const double ta = __builtin_fma((double)lo[a], minv[a], -om[a]), tb = __builtin_fma((double)hi[a], minv[a], -om[a]);
        tmin = fmax(tmin, fmin(ta, tb));
        tmax = fmin(tmax, fmax(ta, tb));
    }
    return !(tmax < tmin || tmax < 0.0);
}

// The walk: calls leaf(first triangle, count) for every leaf whose box (and whose ancestors' boxes) the ray hits;
// tick() runs at the top of every iteration (the kernel empties its queue of pending triangle tests there: at most
// four leaves are reported between two ticks).
struct BvhNoTick { PVT_BVH_HD void operator()() const {} };
template <class Leaf, class Tick = BvhNoTick>
PVT_BVH_HD void bvh_walk(const BvhNode* nodes, int root, const double* om, const double* minv, Leaf&& leaf, Tick&& tick = Tick()) {
    int i = root;
    const int end = nodes[root].skip;
    unsigned long long trail = 0ull;   // bit 4 L + k: child k of the current ancestor at level L was hit (inner children)
    while (i < end) {
        tick();
        const BvhNode& N = nodes[i];
        const int level = N.level;
        if (level > 0 && !((trail >> (4 * (level - 1) + N.slot)) & 1ull)) {   // the parent's test left this child out
            i = N.skip;
            continue;
        }
        unsigned int inner = 0u;
        int first = 0;
        for (int k = 0; k < 4; k++) {
            const int c = N.child[k];
            if (c == 0 || !bvh_box_hit(N.lo[k], N.hi[k], om, minv)) continue;
            if (c < 0) {
                leaf((-c - 1) >> 4, (-c - 1) & 15);
            } else {
                inner |= 1u << k;
                if (!first) first = c;
            }
        }
        trail = (trail & ~(0xFull << (4 * level))) | ((unsigned long long)inner << (4 * level));
        i = first ? first : N.skip;
    }
}

class BvhBuilder {
public:
    BvhBuilder(const double* vertices, const int32_t* faces, const double* normals,
               std::vector<BvhNode>& nodes, std::vector<MeshTri>& tris, std::vector<MeshTriCold>& cold)
        : v_(vertices), f_(faces), n_(normals), nodes_(nodes), tris_(tris), cold_(cold) {}

    // Adds the BVH of faces [f0, f0 + count) and returns the index of its root node.
    int add_mesh(int f0, int count) {
        order_.resize(count);
        std::iota(order_.begin(), order_.end(), f0);
        cx_.resize(3 * (size_t)count);
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < count; k++) {
            const int32_t* idx = f_ + 3 * (size_t)(f0 + k);
            for (int a = 0; a < 3; a++) {
                double s = 0.0;
                for (int c = 0; c < 3; c++) {
                    double x = v_[3 * (size_t)idx[c] + a];
                    s += x;
                    lo[a] = std::min(lo[a], x);
                    hi[a] = std::max(hi[a], x);
                }
                cx_[3 * (size_t)k + a] = s / 3.0;
            }
        }
        double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) +
                                (hi[2] - lo[2]) * (hi[2] - lo[2]));
        pad_ = 1e-7 * diag + 1e-300;
        f0_ = f0;
        leaf_ = count <= kSmallMesh ? kSmallLeaf : kLargeLeaf;
        // a binary tree first (surface-area heuristic; median splits if that comes out too deep for the trail) ...
        for (int attempt = 0; attempt < 2; attempt++) {
            bin_.clear();
            std::iota(order_.begin(), order_.end(), f0);
            build_binary(0, count, attempt == 1);
            if (wide_depth(0) <= kMaxLevels) break;
        }
        // ... then its levels collapsed into 4-wide nodes, emitted depth first
        const int root = (int)nodes_.size();
        emit(0, 0, 0);
        return root;
    }

private:
    struct Bin {              // node of the intermediate binary tree over order_[begin, end)
        int begin, end, left, right;   // children: indices into bin_, -1 for a leaf
        double lo[3], hi[3];
    };

    void bounds(int begin, int end, double* lo, double* hi) const {
        for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
        for (int k = begin; k < end; k++) {
            const int32_t* idx = f_ + 3 * (size_t)order_[k];
            for (int c = 0; c < 3; c++)
                for (int a = 0; a < 3; a++) {
                    double x = v_[3 * (size_t)idx[c] + a];
                    lo[a] = std::min(lo[a], x);
                    hi[a] = std::max(hi[a], x);
                }
        }
    }
    int build_binary(int begin, int end, bool median_only) {
        const int me = (int)bin_.size();
        bin_.push_back(Bin{begin, end, -1, -1, {0, 0, 0}, {0, 0, 0}});
        bounds(begin, end, bin_[me].lo, bin_[me].hi);
        if (end - begin <= leaf_) {
            std::sort(order_.begin() + begin, order_.begin() + end);   // face order inside a leaf
            return me;
        }
        // Binned surface-area heuristic: for each axis the centroids fall into kBins bins; the split plane
        // between two bins that minimises  area(left) * n_left + area(right) * n_right  wins (the expected
        // number of triangle tests of a random ray).  Degenerate cases (all centroids in one bin on every
        // axis) fall back to the median along the widest axis.
        double clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = begin; k < end; k++)
            for (int a = 0; a < 3; a++) {
                double c = cx_[3 * (size_t)(order_[k] - f0_) + a];
                clo[a] = std::min(clo[a], c);
                chi[a] = std::max(chi[a], c);
            }
        int axis = 0;
        if (chi[1] - clo[1] > chi[axis] - clo[axis]) axis = 1;
        if (chi[2] - clo[2] > chi[axis] - clo[axis]) axis = 2;
        int mid = (begin + end) / 2;
        double best_cost = INFINITY, best_plane = 0.0;
        int best_axis = -1;
        constexpr int kBins = 16;
        auto half_area = [](const double* l, const double* h) {
            const double dx = h[0] - l[0], dy = h[1] - l[1], dz = h[2] - l[2];
            return dx * dy + dy * dz + dz * dx;
        };
        for (int a = 0; a < 3 && end - begin > 4 && !median_only; a++) {
            const double span = chi[a] - clo[a];
            if (!(span > 0.0)) continue;
            int cnt[kBins] = {0};
            double blo[kBins][3], bhi[kBins][3];
            for (int b = 0; b < kBins; b++)
                for (int c = 0; c < 3; c++) { blo[b][c] = INFINITY; bhi[b][c] = -INFINITY; }
            for (int k = begin; k < end; k++) {
                const int face = order_[k];
                int b = (int)((cx_[3 * (size_t)(face - f0_) + a] - clo[a]) / span * kBins);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                cnt[b] += 1;
                const int32_t* idx = f_ + 3 * (size_t)face;
                for (int c = 0; c < 3; c++)
                    for (int e = 0; e < 3; e++) {
                        const double x = v_[3 * (size_t)idx[c] + e];
                        blo[b][e] = std::min(blo[b][e], x);
                        bhi[b][e] = std::max(bhi[b][e], x);
                    }
            }
            // right-to-left suffix boxes, then a left-to-right sweep
            double rlo[kBins][3], rhi[kBins][3];
            int rcnt[kBins];
            double l3[3] = {INFINITY, INFINITY, INFINITY}, h3[3] = {-INFINITY, -INFINITY, -INFINITY};
            int n = 0;
            for (int b = kBins - 1; b >= 0; b--) {
                for (int e = 0; e < 3; e++) { l3[e] = std::min(l3[e], blo[b][e]); h3[e] = std::max(h3[e], bhi[b][e]); }
                n += cnt[b];
                for (int e = 0; e < 3; e++) { rlo[b][e] = l3[e]; rhi[b][e] = h3[e]; }
                rcnt[b] = n;
            }
            for (int e = 0; e < 3; e++) { l3[e] = INFINITY; h3[e] = -INFINITY; }
            n = 0;
            for (int b = 0; b + 1 < kBins; b++) {
                for (int e = 0; e < 3; e++) { l3[e] = std::min(l3[e], blo[b][e]); h3[e] = std::max(h3[e], bhi[b][e]); }
                n += cnt[b];
                if (n == 0 || rcnt[b + 1] == 0) continue;
                const double cost = half_area(l3, h3) * n + half_area(rlo[b + 1], rhi[b + 1]) * rcnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_plane = clo[a] + span * (b + 1) / kBins; }
            }
        }
        if (best_axis >= 0) {
            auto left = [&](int face) {
                const double span = chi[best_axis] - clo[best_axis];
                int b = (int)((cx_[3 * (size_t)(face - f0_) + best_axis] - clo[best_axis]) / span * kBins);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return clo[best_axis] + span * (b + 1) / kBins <= best_plane;
            };
            mid = (int)(std::stable_partition(order_.begin() + begin, order_.begin() + end, left) - order_.begin());
        }
        if (best_axis < 0 || mid == begin || mid == end) {
            mid = (begin + end) / 2;
            std::nth_element(order_.begin() + begin, order_.begin() + mid, order_.begin() + end,
                             [&](int a, int b) {
                                 double ca = cx_[3 * (size_t)(a - f0_) + axis], cb = cx_[3 * (size_t)(b - f0_) + axis];
                                 return ca < cb || (ca == cb && a < b);
                             });
        }
        const int l = build_binary(begin, mid, median_only);
        const int r = build_binary(mid, end, median_only);
        bin_[me].left = l;
        bin_[me].right = r;
        return me;
    }
    // the (up to four) binary nodes that become the children of the wide node made from binary node b: its two
    // children, then -- while there is room -- the inner one with the largest box replaced by ITS two children
    void gather(int b, int* kids, int& n) const {
        n = 0;
        if (bin_[b].left < 0) { kids[n++] = b; return; }    // (a mesh of one leaf: the root holds it)
        kids[n++] = bin_[b].left;
        kids[n++] = bin_[b].right;
        while (n < 4) {
            int pick = -1;
            double best = -1.0;
            for (int k = 0; k < n; k++) {
                const Bin& c = bin_[kids[k]];
                if (c.left < 0) continue;
                const double dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
                const double area = dx * dy + dy * dz + dz * dx;
                if (area > best) { best = area; pick = k; }
            }
            if (pick < 0) break;
            const int c = kids[pick];
            kids[pick] = bin_[c].left;              // keeps depth-first order close to the binary tree's
            for (int k = n; k > pick + 1; k--) kids[k] = kids[k - 1];
            kids[pick + 1] = bin_[c].right;
            n += 1;
        }
    }
    int wide_depth(int b) const {
        int kids[4], n;
        gather(b, kids, n);
        int d = 0;
        for (int k = 0; k < n; k++)
            if (bin_[kids[k]].left >= 0) d = std::max(d, wide_depth(kids[k]));
        return d + 1;
    }
    void emit(int b, int level, int slot) {
        const int me = (int)nodes_.size();
        nodes_.push_back(BvhNode{});
        nodes_[me].level = (short)level;
        nodes_[me].slot = (short)slot;
        int kids[4], n;
        gather(b, kids, n);
        for (int k = 0; k < 4; k++) {
            for (int a = 0; a < 3; a++) { nodes_[me].lo[k][a] = INFINITY; nodes_[me].hi[k][a] = -INFINITY; }
            nodes_[me].child[k] = 0;
        }
        for (int k = 0; k < n; k++) {
            const Bin c = bin_[kids[k]];
            for (int a = 0; a < 3; a++) {
                nodes_[me].lo[k][a] = std::nextafter((float)(c.lo[a] - pad_), -INFINITY);   // (float) rounds to nearest:
                nodes_[me].hi[k][a] = std::nextafter((float)(c.hi[a] + pad_), INFINITY);    // one more step outwards
            }
            if (c.left < 0) {      // a leaf: its triangles, gathered in face order
                nodes_[me].child[k] = -(1 + (((int)tris_.size() << 4) | (c.end - c.begin)));
                for (int q = c.begin; q < c.end; q++) {
                    const int face = order_[q];
                    const int32_t* idx = f_ + 3 * (size_t)face;
                    MeshTri t{};
                    MeshTriCold u{};
                    for (int v = 0; v < 3; v++)
                        for (int a = 0; a < 3; a++) t.v[3 * v + a] = v_[3 * (size_t)idx[v] + a];
                    for (int a = 0; a < 3; a++) u.n[a] = n_[3 * (size_t)face + a];
                    u.face = face;
                    tris_.push_back(t);
                    cold_.push_back(u);
                }
            } else {
                nodes_[me].child[k] = (int)nodes_.size();
                emit(kids[k], level + 1, k);
            }
        }
        nodes_[me].skip = (int)nodes_.size();
    }

    const double* v_;
    const int32_t* f_;
    const double* n_;
    std::vector<BvhNode>& nodes_;
    std::vector<MeshTri>& tris_;
    std::vector<MeshTriCold>& cold_;
    std::vector<Bin> bin_;
    std::vector<int> order_;
    std::vector<double> cx_;
    double pad_ = 0.0;
    int f0_ = 0;
    int leaf_ = kLargeLeaf;
};

}  // namespace pvt
