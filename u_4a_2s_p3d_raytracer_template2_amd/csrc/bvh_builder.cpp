// bvh_builder.cpp -- see bvh_builder.h.
#include "bvh_builder.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <limits>

namespace p3d {
namespace {

struct Box {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = FLT_MAX; hi[a] = -FLT_MAX; } }
    void grow(const float* l, const float* h) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    void grow(const Box& b) { grow(b.lo, b.hi); }
    float half_area() const {
        float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0 || dy < 0 || dz < 0) return 0.0f;
        return dx * dy + dx * dz + dy * dz;
    }
};

struct Builder {
    std::vector<BuildPrim>& P;
    const BvhOptions& opt;
    std::vector<NodePair>& nodes;
    std::vector<uint32_t>& refs;
    BvhStats& st;
    double sah = 0.0;
    float root_area = 1.0f;

    Builder(std::vector<BuildPrim>& p, const BvhOptions& o, std::vector<NodePair>& n,
            std::vector<uint32_t>& r, BvhStats& s) : P(p), opt(o), nodes(n), refs(r), st(s) {}

    static float centroid(const BuildPrim& p, int a) { return 0.5f * (p.lo[a] + p.hi[a]); }

    int32_t make_leaf(size_t b, size_t e, uint32_t depth, const Box& box) {
        std::sort(P.begin() + b, P.begin() + e,
                  [](const BuildPrim& x, const BuildPrim& y) { return x.ref < y.ref; });   // kind-major: the
        // lanes of a wave then mostly run the same intersector in the same leaf-loop iteration
        uint32_t first = (uint32_t)refs.size();
        for (size_t i = b; i < e; i++) refs.push_back(P[i].ref);
        st.n_leaves++;
        st.max_depth = std::max(st.max_depth, depth);
        sah += (double)box.half_area() / root_area * opt.cost_intersect * (double)(e - b);
        uint32_t code = (first << 3) | (uint32_t)(e - b - 1);
        return (int32_t)~code;
    }

    // returns child reference, fills `box` with the bounds of [b,e)
    int32_t build(size_t b, size_t e, uint32_t depth, Box& box) {
        box.reset();
        Box cb; cb.reset();
        for (size_t i = b; i < e; i++) {
            box.grow(P[i].lo, P[i].hi);
            float c[3] = {centroid(P[i], 0), centroid(P[i], 1), centroid(P[i], 2)};
            cb.grow(c, c);
        }
        size_t n = e - b;
        if (n == 1) return make_leaf(b, e, depth, box);

        // ---- binned SAH over the three axes
        const int NB = (int)std::min<uint32_t>(std::max<uint32_t>(opt.bins, 2), 64);
        float best_cost = FLT_MAX; int best_axis = -1, best_bin = -1;
        float parent_area = box.half_area();
        for (int a = 0; a < 3; a++) {
            float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.0f)) continue;
            Box bb[64]; uint32_t cnt[64];
            for (int k = 0; k < NB; k++) { bb[k].reset(); cnt[k] = 0; }
            float scale = (float)NB / ext;
            for (size_t i = b; i < e; i++) {
                int k = (int)((centroid(P[i], a) - cb.lo[a]) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                bb[k].grow(P[i].lo, P[i].hi); cnt[k]++;
            }
            float right_area[64]; uint32_t right_cnt[64];
            Box acc; acc.reset(); uint32_t c = 0;
            for (int k = NB - 1; k >= 1; k--) {
                acc.grow(bb[k]); c += cnt[k];
                right_area[k] = acc.half_area(); right_cnt[k] = c;
            }
            acc.reset(); c = 0;
            for (int k = 0; k < NB - 1; k++) {
                acc.grow(bb[k]); c += cnt[k];
                if (c == 0 || right_cnt[k + 1] == 0) continue;
                float cost = acc.half_area() * (float)c + right_area[k + 1] * (float)right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = k; }
            }
        }
        float split_cost = FLT_MAX;
        if (best_axis >= 0 && parent_area > 0.0f)
            split_cost = opt.cost_traverse + opt.cost_intersect * best_cost / parent_area;
        float leaf_cost = opt.cost_intersect * (float)n;
        if (n <= opt.leaf_max && leaf_cost <= split_cost) return make_leaf(b, e, depth, box);

        size_t mid;
        if (best_axis < 0 || depth > 56) {
            // coincident centroids (or a pathologically deep tree): object-median split in
            // scene order so the tree depth stays O(log n)
            mid = b + n / 2;
            if (best_axis >= 0)
                std::nth_element(P.begin() + b, P.begin() + mid, P.begin() + e,
                                 [&](const BuildPrim& x, const BuildPrim& y) {
                                     return centroid(x, best_axis) < centroid(y, best_axis);
                                 });
        } else {
            float ext = cb.hi[best_axis] - cb.lo[best_axis];
            float scale = (float)NB / ext;
            float lo = cb.lo[best_axis];
            int a = best_axis, kb = best_bin;
            auto it = std::partition(P.begin() + b, P.begin() + e, [&](const BuildPrim& x) {
                int k = (int)((centroid(x, a) - lo) * scale);
                k = std::min(std::max(k, 0), NB - 1);
                return k <= kb;
            });
            mid = (size_t)(it - P.begin());
            if (mid == b || mid == e) mid = b + n / 2;
        }

        int32_t idx = (int32_t)nodes.size();
        nodes.push_back(NodePair());
        sah += (double)parent_area / root_area * opt.cost_traverse;
        Box lb, rb;
        int32_t lc = build(b, mid, depth + 1, lb);
        int32_t rc = build(mid, e, depth + 1, rb);
        NodePair& nd = nodes[idx];
        nd.lo0[0] = lb.lo[0]; nd.lo0[1] = lb.lo[1]; nd.lo0[2] = lb.lo[2]; nd.hi0x = lb.hi[0];
        nd.hi0yz[0] = lb.hi[1]; nd.hi0yz[1] = lb.hi[2]; nd.lo1xy[0] = rb.lo[0]; nd.lo1xy[1] = rb.lo[1];
        nd.lo1z = rb.lo[2]; nd.hi1[0] = rb.hi[0]; nd.hi1[1] = rb.hi[1]; nd.hi1[2] = rb.hi[2];
        nd.child0 = lc; nd.child1 = rc; nd.pad0 = nd.pad1 = 0;
        return idx;
    }
};

void set_absent(NodePair& nd, int which) {
    float q = std::numeric_limits<float>::quiet_NaN();
    if (which == 0) {
        nd.lo0[0] = nd.lo0[1] = nd.lo0[2] = nd.hi0x = nd.hi0yz[0] = nd.hi0yz[1] = q;
        nd.child0 = ~0;   // leaf code 0: never reached (NaN box)
    } else {
        nd.lo1xy[0] = nd.lo1xy[1] = nd.lo1z = nd.hi1[0] = nd.hi1[1] = nd.hi1[2] = q;
        nd.child1 = ~0;
    }
}

}  // namespace

void build_bvh(std::vector<BuildPrim>& prims, const BvhOptions& opt, std::vector<NodePair>& nodes,
               std::vector<uint32_t>& leaf_refs, BvhStats& stats) {
    nodes.clear(); leaf_refs.clear(); stats = BvhStats();
    Builder B(prims, opt, nodes, leaf_refs, stats);
    if (prims.empty()) {
        NodePair root = NodePair();
        set_absent(root, 0); set_absent(root, 1);
        nodes.push_back(root);
        leaf_refs.push_back(0);
        stats.n_nodes = 1;
        return;
    }
    Box all; all.reset();
    for (auto& p : prims) all.grow(p.lo, p.hi);
    B.root_area = std::max(all.half_area(), 1e-30f);
    Box box;
    // reserve slot 0 for the root so that "nodes[0] is the root" holds even if the whole
    // scene fits one leaf
    size_t n = prims.size();
    BvhOptions o = opt;
    int32_t r = B.build(0, n, 1, box);
    if (r < 0) {
        NodePair root = NodePair();
        root.lo0[0] = box.lo[0]; root.lo0[1] = box.lo[1]; root.lo0[2] = box.lo[2];
        root.hi0x = box.hi[0]; root.hi0yz[0] = box.hi[1]; root.hi0yz[1] = box.hi[2];
        root.child0 = r;
        set_absent(root, 1);
        nodes.push_back(root);
    }
    // build() is depth-first with the root pushed first, so when r >= 0 it is index 0
    stats.n_nodes = (uint32_t)nodes.size();
    stats.n_leaf_refs = (uint32_t)leaf_refs.size();
    stats.sah_cost = (float)B.sah;
    (void)o;
}

}  // namespace p3d
