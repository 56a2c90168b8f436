// Host-side construction of the triangle BVH the trace kernel walks for PVT_GEOM_MESH nodes.
//
// Layout is made for a per-lane, stack-free walk on the GPU: every record carries two links -- `skip`, the
// record to go to when its box is missed or its subtree is done, and `link`, its first child (inner records) or
// its triangle (leaves) -- so the traversal is `i = hit && inner ? link[i] : skip[i]` with no per-lane stack in
// scratch or LDS.  The two children of a record are stored NEXT TO EACH OTHER, as a 64-byte-aligned pair: the
// walk visits both whenever it hits their parent (the left one's skip link is the right one), so the cache line
// fetched for the left child serves the right one as well.  (Until round 4 the records were in depth-first order
// with the left child implied at i + 1: one and a half lines per parent instead of one, and the right child always a
// fresh request to L2.)  A photon needs EVERY forward crossing of a mesh (the container rule counts them,
// _kernel.pyx:684-714), so front-to-back ordering buys nothing and a fixed order is free.
// The tree is built with a binned surface-area heuristic.  Leaves hold one triangle, pre-gathered (vertices + face
// normal + face id) in a 104-byte record.  Records are 32 bytes (f32 boxes rounded outwards): culling is only a
// filter, the triangle test itself stays f64.
//
// The boxes are stored RELATIVE TO THE CENTRE of the mesh's bounding box and padded by 4e-6 of its diagonal:
// the kernel tests them in f32 against a ray re-originated at its entry into the root box, so every
// quantity of the test is of the mesh's own size and the test's rounding displaces a plane by at most ~5e-7
// of the diagonal (four roundings of 2^-24 relative, on differences of at most two diagonals).  Culling must
// be conservative with respect to the exact (f64) watertight triangle test, so that the set of crossings
// found through the BVH equals the set a brute-force loop over the faces finds (which is what
// oracle/pvt_oracle.c does) bit for bit: the padding is eight times the rounding.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

namespace pvt {

struct alignas(16) BvhNode {      // 32 bytes: a pair of siblings per 64-byte line, half the traffic of f64 boxes
    float lo[3], hi[3];   // box relative to the mesh's centre, padded, rounded OUTWARDS to f32: still conservative
    int skip;             // where to go when this subtree is culled or finished (a cursor, see kTopFlag)
    int link;             // leaves: kLeafFlag | triangle record; inner records: the first child (a cursor), its sibling follows it
};
constexpr int kLeafFlag = (int)0x80000000;   // (a leaf's link is negative)
constexpr int kTopFlag = 1 << 30;            // a cursor with this bit names a slot of the copy in LDS (stage_top), else a record in global memory
constexpr int kIndexMask = kTopFlag - 1;
struct MeshTri {      // 104 bytes
    double v[9];      // three vertices, node-local frame
    double n[3];      // outward unit face normal
    long long face;   // index in the scene's pooled face table (tie-break key, diagnostics)
};

// Leaf size: the watertight triangle test costs several box tests (185 vector instructions against 40, and the boxes of
// a small tree are read from LDS), so every leaf holds ONE triangle.  (Rounds 1-3 gave meshes of up to 32 faces leaves of
// eight; with round 4's walk one triangle per leaf is 24 % faster on the 12-triangle slab and 52 % on the L-shaped prism:
// profiles/r04_mesh_walk_series.txt.)

class BvhBuilder {
public:
    BvhBuilder(const double* vertices, const int32_t* faces, const double* normals,
               std::vector<BvhNode>& nodes, std::vector<MeshTri>& tris)
        : v_(vertices), f_(faces), n_(normals), nodes_(nodes), tris_(tris) {}

    // Adds the BVH of faces [f0, f0 + count) and returns the index of its root node; `centre` (nullable) receives
    // the point the boxes are relative to.
    int add_mesh(int f0, int count, double* centre = nullptr) {
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
        pad_ = 4e-6 * diag + 1e-300;
        for (int a = 0; a < 3; a++) {
            c_[a] = 0.5 * (lo[a] + hi[a]);
            if (centre) centre[a] = c_[a];
        }
        f0_ = f0;
        // the root at an even index with an unused record after it, so that every pair of siblings shares a 64-byte line
        if (nodes_.size() & 1) nodes_.push_back(unused());
        const int root = (int)nodes_.size();
        nodes_.push_back(BvhNode{});
        nodes_.push_back(unused());
        build(root, 0, count);
        link_skips(root, (int)nodes_.size());   // the tree's end: one past its last record
        nodes_[root + 1].skip = (int)nodes_.size();
        return root;
    }

private:
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
    static BvhNode unused() {   // (never walked; an empty box and links that end a walk, should one ever get here)
        BvhNode b{};
        for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
        b.link = kLeafFlag;
        return b;
    }
    // skip links, from the top: the left child's is its sibling, the right child's is its parent's
    void link_skips(int me, int skip) {
        nodes_[me].skip = skip;
        if (nodes_[me].link >= 0) {
            const int c = nodes_[me].link;
            link_skips(c, c + 1);
            link_skips(c + 1, skip);
        }
    }
    void build(int me, int begin, int end) {
        double lo[3], hi[3];
        bounds(begin, end, lo, hi);
        for (int a = 0; a < 3; a++) {
            nodes_[me].lo[a] = std::nextafter((float)(lo[a] - pad_ - c_[a]), -INFINITY);   // (float) rounds to nearest:
            nodes_[me].hi[a] = std::nextafter((float)(hi[a] + pad_ - c_[a]), INFINITY);    // one more step outwards
        }
        if (end - begin <= 1) {
            nodes_[me].link = kLeafFlag | (int)tris_.size();
            const int face = order_[begin];
            const int32_t* idx = f_ + 3 * (size_t)face;
            MeshTri t{};
            for (int c = 0; c < 3; c++)
                for (int a = 0; a < 3; a++) t.v[3 * c + a] = v_[3 * (size_t)idx[c] + a];
            for (int a = 0; a < 3; a++) t.n[a] = n_[3 * (size_t)face + a];
            t.face = face;
            tris_.push_back(t);
        } else {
            // Binned surface-area heuristic: for each axis the centroids fall into kBins bins; the split plane
            // between two bins that minimises  area(left) * n_left + area(right) * n_right  wins (the expected
            // number of triangle tests of a random ray).  Degenerate cases (all centroids in one bin on every
            // axis) fall back to the median along the widest axis.  Which tree is built never changes a
            // result: a photon collects EVERY crossing and orders them by (t, face).
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
            for (int a = 0; a < 3 && end - begin > 4; a++) {
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
            const int c = (int)nodes_.size();   // the children, side by side
            nodes_.push_back(BvhNode{});
            nodes_.push_back(BvhNode{});
            nodes_[me].link = c;
            build(c, begin, mid);
            build(c + 1, mid, end);
        }
    }

    const double* v_;
    const int32_t* f_;
    const double* n_;
    std::vector<BvhNode>& nodes_;
    std::vector<MeshTri>& tris_;
    std::vector<int> order_;
    std::vector<double> cx_;
    double pad_ = 0.0;
    double c_[3] = {0.0, 0.0, 0.0};
    int f0_ = 0;
};

// ---- the top of the trees, for LDS ------------------------------------------------------------------------
// The kernel's walk waits for its records far longer than it computes with them, and every ray passes the top
// levels of a tree.  `stage_top` picks, for every tree, the levels that fit a budget of records (small trees
// whole, the rest sharing what is left), copies them -- in the order they have in `nodes`, so siblings stay side by
// side -- to `top`, the image a workgroup loads into LDS, and re-writes the links of BOTH arrays as cursors: a
// record's index in `nodes`, or kTopFlag | slot in `top` when the record has a copy.  The tree's end stays what it
// was (the root's skip link: the walk is over when the cursor equals it).  A skip link leads to a sibling or to an
// ancestor's sibling -- never deeper than its source -- so links INTO the copy come from everywhere, links out of
// it only from the child links of its last level.  The root's record in `nodes` is never walked when a copy exists,
// but it is where the kernel learns of it: its child link then names slot + 1 of the root's own copy.
// Trees without a copy (no budget, or a single leaf) are left as they were.  Results never depend on any of this.
inline void stage_top(std::vector<BvhNode>& nodes, const std::vector<int>& roots, size_t budget, std::vector<BvhNode>& top) {
    top.clear();
    struct Tree { int root, end; std::vector<int> depth; std::vector<size_t> per_level; };
    std::vector<Tree> trees;
    for (int r : roots) {
        if (nodes[r].link < 0) continue;   // a single leaf
        Tree t{r, nodes[r].skip, std::vector<int>((size_t)(nodes[r].skip - r), -1), {}};
        std::vector<int> todo{r};
        t.depth[0] = 0;
        while (!todo.empty()) {
            const int i = todo.back();
            todo.pop_back();
            const size_t d = (size_t)t.depth[i - r];
            if (t.per_level.size() <= d) t.per_level.resize(d + 1, 0);
            t.per_level[d] += 1;
            if (nodes[i].link >= 0)
                for (int c = nodes[i].link; c < nodes[i].link + 2; c++) { t.depth[c - r] = (int)d + 1; todo.push_back(c); }
        }
        trees.push_back(std::move(t));
    }
    // small trees first: what they leave of their share goes to the larger ones
    std::vector<size_t> by_size(trees.size());
    std::iota(by_size.begin(), by_size.end(), (size_t)0);
    std::sort(by_size.begin(), by_size.end(), [&](size_t a, size_t b) {
        const int na = trees[a].end - trees[a].root, nb = trees[b].end - trees[b].root;
        return na < nb || (na == nb && a < b);
    });
    std::vector<int> last_level(trees.size(), -1);
    size_t left = budget;
    for (size_t k = 0; k < by_size.size(); k++) {
        const Tree& t = trees[by_size[k]];
        const size_t share = left / (by_size.size() - k);
        size_t count = 0;
        int L = -1;
        for (size_t d = 0; d < t.per_level.size() && count + t.per_level[d] <= share; d++) { count += t.per_level[d]; L = (int)d; }
        if (L < 1) continue;   // (the root alone is not worth a copy)
        last_level[by_size[k]] = L;
        left -= count;
    }
    for (size_t k = 0; k < trees.size(); k++) {   // (slots in scene order)
        const Tree& t = trees[k];
        const int L = last_level[k];
        if (L < 0) continue;
        std::vector<int> slot((size_t)(t.end - t.root), -1);
        int next = (int)top.size();
        for (int i = t.root; i < t.end; i++)
            if (t.depth[i - t.root] >= 0 && t.depth[i - t.root] <= L) slot[i - t.root] = next++;
        auto cursor = [&](int target) { return target == t.end || slot[target - t.root] < 0 ? target : (kTopFlag | slot[target - t.root]); };
        for (int i = t.root; i < t.end; i++) {
            if (t.depth[i - t.root] < 0) continue;   // (the unused record after the root)
            nodes[i].skip = cursor(nodes[i].skip);
            if (nodes[i].link >= 0) nodes[i].link = cursor(nodes[i].link);
            if (slot[i - t.root] >= 0) top.push_back(nodes[i]);
        }
    }
}

// What the kernel's walk does with a cursor, restated for the host-side check of `stage_top` (pvt_mesh_bvh_check):
// the record a cursor names, and the cursor after it for a hit / a miss.
inline const BvhNode& at_cursor(const std::vector<BvhNode>& nodes, const std::vector<BvhNode>& top, int cursor) {
    return (cursor & kTopFlag) ? top[cursor & kIndexMask] : nodes[cursor];
}
inline int next_cursor(const BvhNode& b, bool hit) { return hit && b.link >= 0 ? b.link : b.skip; }
// where a walk of the tree rooted at record `root` starts: the root's copy when there is one (see stage_top)
inline int first_cursor(const std::vector<BvhNode>& nodes, int root) {
    const int link = nodes[root].link;
    return link >= 0 && (link & kTopFlag) ? (kTopFlag | ((link & kIndexMask) - 1)) : root;
}

}  // namespace pvt
