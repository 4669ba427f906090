#include "grid_builder.h"

#include <algorithm>
#include <cfloat>
#include <cmath>

#include "p3d_device_types.h"

namespace p3d {

static const float kEps = 0.001f;                                        // RT/macros.h:1

void grid_prims_from_desc(const p3d_scene_desc& d, std::vector<GridPrim>& out) {
    out.clear();
    out.reserve(d.n_prims);
    uint32_t n_sph = 0, n_tri = 0, n_box = 0, n_pln = 0;
    for (uint32_t i = 0; i < d.n_prims; i++) {
        const float* v = d.prim_data + 12 * (size_t)i;
        GridPrim g;
        switch (d.prim_type[i]) {
        case P3D_SPHERE:                                                 // RT/scene.cpp:180-186
            for (int a = 0; a < 3; a++) { g.lo[a] = v[a] - v[3]; g.hi[a] = v[a] + v[3]; }
            g.ref = (0u << kRefKindShift) | n_sph++;
            break;
        case P3D_TRIANGLE:                                               // RT/scene.cpp:26-39: min/max, then -= / += EPSILON
            for (int a = 0; a < 3; a++) {
                float lo = std::min(std::min(v[a], v[3 + a]), v[6 + a]);
                float hi = std::max(std::max(v[a], v[3 + a]), v[6 + a]);
                g.lo[a] = lo - kEps; g.hi[a] = hi + kEps;
            }
            g.ref = (1u << kRefKindShift) | n_tri++;
            break;
        case P3D_BOX:                                                    // RT/scene.cpp:194-196
            for (int a = 0; a < 3; a++) { g.lo[a] = v[a]; g.hi[a] = v[3 + a]; }
            g.ref = (2u << kRefKindShift) | n_box++;
            break;
        default:                                                         // Plane: Object::GetBoundingBox(), RT/scene.h:75
            for (int a = 0; a < 3; a++) { g.lo[a] = -1.0f; g.hi[a] = 1.0f; }
            g.ref = (3u << kRefKindShift) | n_pln++;
            break;
        }
        out.push_back(g);
    }
}

static inline double dclamp(double x, double mn, double mx) { return x < mn ? mn : (x > mx ? mx : x); }   // RT/maths.h:50-53

bool build_grid(const std::vector<GridPrim>& prims, GridHost& G) {       // RT/grid.cpp:30-98
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (const GridPrim& p : prims)
        for (int a = 0; a < 3; a++) {                                    // AABB::extend, RT/boundingBox.cpp:52-60
            if (mn[a] > p.lo[a]) mn[a] = p.lo[a];
            if (mx[a] < p.hi[a]) mx[a] = p.hi[a];
        }
    for (int a = 0; a < 3; a++) { mn[a] -= kEps; mx[a] += kEps; G.mn[a] = mn[a]; G.mx[a] = mx[a]; }
    const double wx = mx[0] - mn[0], wy = mx[1] - mn[1], wz = mx[2] - mn[2];
    const double s = pow((int)prims.size() / (wx * wy * wz), 0.3333333);
    const float m = 2.0f;                                                // RT/rayAccelerator.h:29
    const double fx = m * wx * s + 1, fy = m * wy * s + 1, fz = m * wz * s + 1;
    // No primitives (the box is still inverted: widths of -inf, s = 0, NaN counts) or coordinates that are not
    // numbers: the reference's int conversion of such a count is undefined behaviour.  Here: ONE empty cell and the
    // box as it stands -- an inverted box is missed by every ray (grid_init), which is what an empty grid amounts to.
    if (prims.empty() || !(fx >= 1.0 && fy >= 1.0 && fz >= 1.0) || !std::isfinite(fx) || !std::isfinite(fy) || !std::isfinite(fz)) {
        G.n[0] = G.n[1] = G.n[2] = 1;
        G.cell_start.assign(2, 0u);
        G.items.clear();
        return true;
    }
    if (fx * fy * fz > 2147483647.0) return false;                       // more cells than the device's 32-bit cell index
    const int nx = (int)fx, ny = (int)fy, nz = (int)fz;
    G.n[0] = nx; G.n[1] = ny; G.n[2] = nz;
    const size_t cells = (size_t)nx * ny * nz;
    struct Range { int x0, x1, y0, y1, z0, z1; };
    std::vector<Range> rng(prims.size());
    G.cell_start.assign(cells + 1, 0u);
    for (size_t i = 0; i < prims.size(); i++) {
        const GridPrim& p = prims[i];
        Range r;
        r.x0 = dclamp((p.lo[0] - mn[0]) * nx / (mx[0] - mn[0]), 0, nx - 1);
        r.y0 = dclamp((p.lo[1] - mn[1]) * ny / (mx[1] - mn[1]), 0, ny - 1);
        r.z0 = dclamp((p.lo[2] - mn[2]) * nz / (mx[2] - mn[2]), 0, nz - 1);
        r.x1 = dclamp((p.hi[0] - mn[0]) * nx / (mx[0] - mn[0]), 0, nx - 1);
        r.y1 = dclamp((p.hi[1] - mn[1]) * ny / (mx[1] - mn[1]), 0, ny - 1);
        r.z1 = dclamp((p.hi[2] - mn[2]) * nz / (mx[2] - mn[2]), 0, nz - 1);
        rng[i] = r;
        for (int iz = r.z0; iz <= r.z1; iz++)
            for (int iy = r.y0; iy <= r.y1; iy++)
                for (int ix = r.x0; ix <= r.x1; ix++) G.cell_start[(size_t)ix + (size_t)nx * iy + (size_t)nx * ny * iz + 1]++;
    }
    for (size_t c = 0; c < cells; c++) G.cell_start[c + 1] += G.cell_start[c];
    G.items.assign(G.cell_start[cells], 0u);
    std::vector<uint32_t> fill(G.cell_start.begin(), G.cell_start.end() - 1);
    for (size_t i = 0; i < prims.size(); i++) {                          // scene order inside every cell (push_back order)
        const Range& r = rng[i];
        for (int iz = r.z0; iz <= r.z1; iz++)
            for (int iy = r.y0; iy <= r.y1; iy++)
                for (int ix = r.x0; ix <= r.x1; ix++) G.items[fill[(size_t)ix + (size_t)nx * iy + (size_t)nx * ny * iz]++] = prims[i].ref;
    }
    return true;
}

}  // namespace p3d
