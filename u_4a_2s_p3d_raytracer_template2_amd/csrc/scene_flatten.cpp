#include "scene_flatten.h"

#include <algorithm>
#include <cmath>

namespace p3d {

// Builder bounds are padded by max(1e-3, 1e-5 * |coordinate|): hits are decided by the
// reference's primitive arithmetic, whose rounded hit point can sit a few ulps outside the
// exact geometry, while the slab test must never cull a primitive brute force would hit.
static void pad(BuildPrim& b) {
    float m = 0.0f;
    for (int a = 0; a < 3; a++) m = std::max(m, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
    float p = std::max(1e-3f, 1e-5f * m);
    for (int a = 0; a < 3; a++) { b.lo[a] -= p; b.hi[a] += p; }
}

std::string flatten_scene(const p3d_scene_desc& d, FlatScene& F) {
    if (d.n_prims && (!d.prim_type || !d.prim_data || !d.prim_material)) return "primitive arrays missing";
    if ((d.n_materials && !d.materials) || (d.n_lights && !d.lights)) return "material/light arrays missing";
    F = FlatScene();
    F.build_prims.reserve(d.n_prims);
    for (uint32_t i = 0; i < d.n_prims; i++) {
        const float* v = d.prim_data + 12 * (size_t)i;
        uint32_t mat = d.prim_material[i];
        if (mat >= d.n_materials) return "primitive references a missing material";
        BuildPrim b; b.scene_id = i;
        switch (d.prim_type[i]) {
        case P3D_SPHERE: {
            float r = std::fabs(v[3]);
            for (int a = 0; a < 3; a++) { b.lo[a] = v[a] - r; b.hi[a] = v[a] + r; }
            b.ref = (0u << kRefKindShift) | (uint32_t)F.spheres.size();
            F.spheres.push_back(SphereRec{v[0], v[1], v[2], v[3]});
            F.sphere_meta.push_back(PrimMeta{i, mat});
            pad(b); F.build_prims.push_back(b);
            break;
        }
        case P3D_TRIANGLE: {
            TriRec t;
            for (int a = 0; a < 3; a++) {
                t.p0[a] = v[a];
                t.e1[a] = v[3 + a] - v[a];        // points[1] - points[0], RT/scene.cpp:62
                t.e2[a] = v[6 + a] - v[a];        // points[2] - points[0], RT/scene.cpp:63
                b.lo[a] = std::min(v[a], std::min(v[3 + a], v[6 + a]));
                b.hi[a] = std::max(v[a], std::max(v[3 + a], v[6 + a]));
            }
            t.scene_id = i; t.material = mat; t.pad = 0;
            b.ref = (1u << kRefKindShift) | (uint32_t)F.tris.size();
            F.tris.push_back(t);
            pad(b); F.build_prims.push_back(b);
            break;
        }
        case P3D_BOX: {
            BoxRec x;
            for (int a = 0; a < 3; a++) {
                x.mn[a] = v[a]; x.mx[a] = v[3 + a];
                b.lo[a] = std::min(v[a], v[3 + a]); b.hi[a] = std::max(v[a], v[3 + a]);
            }
            x.scene_id = i; x.material = mat;
            b.ref = (2u << kRefKindShift) | (uint32_t)F.boxes.size();
            F.boxes.push_back(x);
            pad(b); F.build_prims.push_back(b);
            break;
        }
        case P3D_PLANE:   // unbounded: tested outside the BVH
            F.planes.push_back(PlaneRec{v[0], v[1], v[2], v[3]});
            F.plane_meta.push_back(PrimMeta{i, mat});
            break;
        default: return "unknown primitive type";
        }
    }
    if (F.spheres.size() > kRefIndexMask || F.tris.size() > kRefIndexMask || F.build_prims.size() >= (1u << 28))
        return "too many primitives";
    F.materials.resize(d.n_materials);
    for (uint32_t i = 0; i < d.n_materials; i++) {
        const float* m = d.materials + 12 * (size_t)i;
        F.materials[i] = MaterialRec{{m[0], m[1], m[2]}, m[3], {m[4], m[5], m[6]}, m[7], m[8], m[9], m[10], m[11]};
    }
    F.lights.resize(d.n_lights);
    for (uint32_t i = 0; i < d.n_lights; i++) {
        const float* l = d.lights + 6 * (size_t)i;
        F.lights[i] = LightRec{{l[0], l[1], l[2]}, 0.0f, {l[3], l[4], l[5]}, 0.0f};
    }
    return std::string();
}

}  // namespace p3d
