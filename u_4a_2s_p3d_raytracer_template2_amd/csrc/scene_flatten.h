// scene_flatten.h -- p3d_scene_desc (scene order, loader form) -> device records + padded
// builder bounds.  Shared by p3d_scene_create and the host-only BVH probe so that CPU tests
// exercise exactly what is uploaded.
#ifndef P3D_SCENE_FLATTEN_H
#define P3D_SCENE_FLATTEN_H

#include <string>
#include <vector>

#include "bvh_builder.h"
#include "p3d_device_types.h"
#include "p3d_hip.h"

namespace p3d {

struct FlatScene {
    std::vector<SphereRec> spheres; std::vector<PrimMeta> sphere_meta;
    std::vector<TriRec> tris; std::vector<BoxRec> boxes;
    std::vector<PlaneRec> planes; std::vector<PrimMeta> plane_meta;
    std::vector<MaterialRec> materials; std::vector<LightRec> lights;
    std::vector<BuildPrim> build_prims;      // bounded primitives with PADDED bounds
};

// returns an empty string on success, otherwise the reason the description is invalid
std::string flatten_scene(const p3d_scene_desc& d, FlatScene& out);

// The builders emit leaves as runs of a reference list (kind << 30 | index).  Walking that costs a dependent
// load and a kind switch per primitive, so before upload the primitive arrays are PERMUTED INTO LEAF ORDER and
// every leaf becomes a LeafRec of three typed runs; nodes' leaf children are rewritten to ~(leaf index), absent
// children to leaf 0 (empty).  Records carry their scene id and material, so the order of the arrays is free.
// map_*[old index] = new index (for whoever holds references in the old numbering: the uniform grid).
struct TypedLeaves {
    std::vector<LeafRec> leaves;
    std::vector<uint32_t> map_sph, map_tri, map_box;
    bool overflow = false;            // more leaf records than a reference can index
};
// direct: single-type leaves are named by their reference (kLeafTris / kLeafSpheres) instead of getting a LeafRec
void type_leaves(std::vector<NodePair>& nodes, const std::vector<uint32_t>& refs, FlatScene& F, TypedLeaves& out, bool direct);

// (Node order: the builders emit nodes depth first.  Treelets of 4 / 8 nodes per 128-byte line were built and measured in
//  round 3 -- within 1 % on every scene, profiles/r03_exp03_sharing_wg_levers.txt -- and taken out again.)
// 32-byte node pairs for scenes read from HBM (QNode): 16-bit plane codes on a grid over the boxes of all nodes,
// plane = base + code * scale per axis.  Every coded box contains its f32 box with one code of margin on each side
// (lo: floor - 1, hi: ceil + 1, evaluated in double against the f32 base / scale the device uses), so a slab test on the
// coded box is conservative wherever one on the f32 box was.  An absent child (NaN box) becomes the point box at
// code 0, outside every real box; its reference is the empty leaf.
struct QuantisedNodes { std::vector<QNode> nodes; float scale[3], base[3]; };
void quantise_nodes(const std::vector<NodePair>& nodes, QuantisedNodes& out);

}  // namespace p3d
#endif
