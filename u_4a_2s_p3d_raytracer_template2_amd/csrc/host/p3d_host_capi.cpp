// p3d_host_capi.cpp -- flat C shim over the C++ host layer (p3d_scene.h) so that the Python
// test / bench harness (ctypes) can drive the same loader, camera and sample generator the
// C++ front end uses.  These p3dh_* functions are conveniences ABOVE the drop-in boundary;
// the boundary itself is include/p3d_hip.h.
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../bvh_builder.h"
#include "../scene_flatten.h"
#include "../grid_builder.h"
#include "p3d_scene.h"

using namespace p3d_host;

struct p3dh_scene {
    Scene scene;
    Scene::Flat flat;
};

extern "C" {

p3dh_scene* p3dh_scene_load(const char* path) {
    p3dh_scene* h = new p3dh_scene();
    if (!h->scene.load_p3f(path)) { delete h; return nullptr; }
    h->scene.flatten(h->flat);
    return h;
}
void p3dh_scene_free(p3dh_scene* h) { delete h; }

// out[0..7] = n_prims, n_lights, n_materials, res_x, res_y, accel, spp, parse_ok
void p3dh_scene_info(const p3dh_scene* h, int32_t* out) {
    out[0] = (int32_t)h->flat.desc.n_prims; out[1] = (int32_t)h->flat.desc.n_lights;
    out[2] = (int32_t)h->flat.desc.n_materials;
    out[3] = h->scene.GetCamera()->GetResX(); out[4] = h->scene.GetCamera()->GetResY();
    out[5] = (int32_t)h->scene.GetAccelStruct(); out[6] = (int32_t)h->scene.GetSamplesPerPixel();
    out[7] = h->scene.parse_error().empty() ? 1 : 0;
}
void p3dh_scene_set_resolution(p3dh_scene* h, int32_t w, int32_t hh) { h->scene.GetCamera()->SetResolution(w, hh); }
void p3dh_scene_set_eye(p3dh_scene* h, float x, float y, float z) { h->scene.GetCamera()->SetEye(Vector(x, y, z)); }
// the flattened arrays stay owned by the handle
void p3dh_scene_desc(const p3dh_scene* h, p3d_scene_desc* out) { *out = h->flat.desc; }
void p3dh_scene_camera(const p3dh_scene* h, p3d_camera* out) { h->scene.GetCamera()->describe(out); }
void p3dh_primary_ray(const p3dh_scene* h, float px, float py, float* o3, float* d3) {
    Ray r = h->scene.GetCamera()->PrimaryRay(Vector(px, py, 0));
    o3[0] = r.origin.x; o3[1] = r.origin.y; o3[2] = r.origin.z;
    d3[0] = r.direction.x; d3[1] = r.direction.y; d3[2] = r.direction.z;
}
void p3dh_generate_samples(uint32_t seed, int32_t res_x, int32_t res_y, int32_t spp, float aperture, float* out) {
    generate_samples(seed, res_x, res_y, spp, aperture, out);
}

// ---- host-only BVH build, for tests that run without a GPU
struct p3dh_bvh {
    std::vector<p3d::NodePair> nodes;
    std::vector<uint32_t> refs;
    std::vector<p3d::BuildPrim> prims;   // padded bounds, in the builder's final order
    p3d::BvhStats stats;
};
// saveImgFile() replacement, exposed for the CPU-side tests
int p3dh_save_png(const char* path, const uint8_t* img_Data, int32_t w, int32_t h) { return save_png(path, img_Data, w, h); }

p3dh_bvh* p3dh_bvh_build(const p3d_scene_desc* d, uint32_t leaf_max) {
    p3d::FlatScene F;
    if (!p3d::flatten_scene(*d, F).empty()) return nullptr;
    p3dh_bvh* b = new p3dh_bvh();
    b->prims = F.build_prims;
    p3d::BvhOptions o;
    if (leaf_max) o.leaf_max = leaf_max;
    p3d::build_bvh(b->prims, o, b->nodes, b->refs, b->stats);
    return b;
}
void p3dh_bvh_free(p3dh_bvh* b) { delete b; }
// out[0..4] = n_nodes, n_leaf_refs, n_leaves, max_depth, n_prims
void p3dh_bvh_info(const p3dh_bvh* b, uint32_t* out) {
    out[0] = (uint32_t)b->nodes.size(); out[1] = (uint32_t)b->refs.size(); out[2] = b->stats.n_leaves;
    out[3] = b->stats.max_depth; out[4] = (uint32_t)b->prims.size();
}
// nodes16: 16 dwords per node exactly as uploaded; refs: leaf reference list
void p3dh_bvh_dump(const p3dh_bvh* b, uint32_t* nodes16, uint32_t* refs) {
    memcpy(nodes16, b->nodes.data(), b->nodes.size() * sizeof(p3d::NodePair));
    memcpy(refs, b->refs.data(), b->refs.size() * sizeof(uint32_t));
}

// the 32-byte node pairs kernels that read the scene from HBM walk (csrc/scene_flatten.cpp: quantise_nodes):
// qnodes8 = 8 dwords per node, scale3 / base3 = the de-quantisation constants (plane = base + code * scale)
void p3dh_bvh_quantise(const p3dh_bvh* b, uint32_t* qnodes8, float* scale3, float* base3) {
    p3d::QuantisedNodes Q;
    p3d::quantise_nodes(b->nodes, Q);
    memcpy(qnodes8, Q.nodes.data(), Q.nodes.size() * sizeof(p3d::QNode));
    memcpy(scale3, Q.scale, sizeof Q.scale); memcpy(base3, Q.base, sizeof Q.base);
}

// ---- the triangle normals the device shades with (computed on the host by flatten_scene), scene order of the
// triangles; returns their number.  For the CPU-side parity test against the reference's known answers.
int64_t p3dh_triangle_normals(const p3d_scene_desc* d, float* out3, uint64_t cap) {
    p3d::FlatScene F;
    if (!p3d::flatten_scene(*d, F).empty()) return -1;
    for (size_t i = 0; i < F.tris.size() && i < cap; i++) memcpy(out3 + 3 * i, F.tris[i].n, 12);
    return (int64_t)F.tris.size();
}

// ---- host-only grid build (the reference's Grid::Build layout, csrc/grid_builder.cpp), for tests without a GPU
// dims[3]; counts: cells' populations (nx*ny*nz) or NULL; returns the number of cells, or -1
int64_t p3dh_grid_build(const p3d_scene_desc* d, int32_t* dims, uint32_t* counts, uint64_t counts_cap) {
    std::vector<p3d::GridPrim> prims;
    p3d::grid_prims_from_desc(*d, prims);
    p3d::GridHost g;
    if (!p3d::build_grid(prims, g)) return -1;
    dims[0] = g.n[0]; dims[1] = g.n[1]; dims[2] = g.n[2];
    const size_t cells = g.cell_start.size() - 1;
    if (counts) for (size_t c = 0; c < cells && c < counts_cap; c++) counts[c] = g.cell_start[c + 1] - g.cell_start[c];
    return (int64_t)cells;
}

}  // extern "C"
