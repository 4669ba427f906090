// grid_builder.h -- the reference's uniform grid (Grid::Build, RT/grid.cpp:30-98) as flat arrays for the device:
// same bounding box, same cell-count formula, same cell populations in the same (scene) order.  Unlike the
// BVH, whose shape is free because the reference discards its closest-hit result, the grid's
// shape is part of GRID mode's observable behaviour: hits are accepted per cell (RT/grid.cpp:265-309).
#ifndef P3D_GRID_BUILDER_H
#define P3D_GRID_BUILDER_H

#include <cstdint>
#include <vector>

#include "p3d_hip.h"

namespace p3d {

struct GridHost {
    int32_t n[3] = {0, 0, 0};
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    std::vector<uint32_t> cell_start;     // n[0]*n[1]*n[2] + 1 offsets into items
    std::vector<uint32_t> items;          // primitive refs (kind << 30 | index within the kind), scene order per cell
};

// one entry per primitive of the scene, in scene order: the bounding box Object::GetBoundingBox() returns
// (RT/scene.cpp:42-44,180-186,194-196; planes: the default [-1,1]^3 of RT/scene.h:75) and the
// primitive's device reference
struct GridPrim { float lo[3], hi[3]; uint32_t ref; };

// bounding boxes of a scene description in scene order, with the reference's float arithmetic
void grid_prims_from_desc(const p3d_scene_desc& d, std::vector<GridPrim>& out);
// false: the reference's cell-count formula asks for more than 2^31 cells (nothing is built)
bool build_grid(const std::vector<GridPrim>& prims, GridHost& out);

}  // namespace p3d
#endif
