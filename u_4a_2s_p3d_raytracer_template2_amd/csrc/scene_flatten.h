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

}  // namespace p3d
#endif
