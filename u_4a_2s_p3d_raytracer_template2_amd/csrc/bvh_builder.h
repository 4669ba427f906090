// bvh_builder.h -- host-side binned-SAH BVH2 builder producing the flat NodePair array the
// traversal kernels walk.  Replaces BVH::Build / build_recursive (RT/bvh.cpp:28-158); the
// tree is deliberately NOT the reference's median split: its closest-hit result is discarded
// by the reference itself, so only "conservative boxes" is required for parity.
#ifndef P3D_BVH_BUILDER_H
#define P3D_BVH_BUILDER_H

#include <cstdint>
#include <vector>

#include "p3d_device_types.h"

namespace p3d {

struct BuildPrim {
    float lo[3], hi[3];     // padded bounds (conservative against float rounding)
    uint32_t ref;           // kind << 30 | index in the kind's array
    uint32_t scene_id;      // position in scene order (tie-break key)
};

struct BvhStats {
    uint32_t n_nodes = 0, n_leaves = 0, max_depth = 0, n_leaf_refs = 0;
    float sah_cost = 0.0f;
};

struct BvhOptions {
    uint32_t leaf_max = 4;
    uint32_t bins = 16;
    float cost_traverse = 1.2f;   // one NodePair visit = two slab tests
    float cost_intersect = 1.0f;
};

// nodes[0] is always an inner node (the root); leaf_refs is grouped per leaf, kind-major inside a
// leaf (ascending ref).
void build_bvh(std::vector<BuildPrim>& prims, const BvhOptions& opt,
               std::vector<NodePair>& nodes, std::vector<uint32_t>& leaf_refs, BvhStats& stats);

}  // namespace p3d
#endif
