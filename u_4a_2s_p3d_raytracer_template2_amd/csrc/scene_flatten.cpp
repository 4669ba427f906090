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
            t.scene_id = i; t.material = mat; t.pad = 0; t.pad2 = 0;
            {   // Triangle's stored normal (RT/scene.cpp:16-25: component formulas, then normalize()) as rayTracing()
                // uses it: getNormal(point).normalize() (RT/main.cpp:587) -- normalised twice, in float, in this order.
                // Same IEEE operations the device used to repeat for every hit (-ffp-contract=off on both sides).
                const float* V = t.e1; const float* W = t.e2;
                float n[3] = {(V[1] * W[2]) - (V[2] * W[1]), (V[2] * W[0]) - (V[0] * W[2]), (V[0] * W[1]) - (V[1] * W[0])};
                for (int pass = 0; pass < 2; pass++) {
                    const float l = 1.0f / std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);     // RT/vector.cpp:66-71
                    n[0] *= l; n[1] *= l; n[2] *= l;
                }
                t.n[0] = n[0]; t.n[1] = n[1]; t.n[2] = n[2];
            }
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

void type_leaves(std::vector<NodePair>& nodes, const std::vector<uint32_t>& refs, FlatScene& F, TypedLeaves& T, bool direct) {
    const uint32_t kNone = 0xFFFFFFFFu;
    T.leaves.clear();
    T.leaves.push_back(LeafRec{0, 0, 0, 0});                      // leaf 0: empty
    T.map_sph.assign(F.spheres.size(), kNone); T.map_tri.assign(F.tris.size(), kNone); T.map_box.assign(F.boxes.size(), kNone);
    std::vector<SphereRec> sph; std::vector<PrimMeta> sph_meta; std::vector<TriRec> tri; std::vector<BoxRec> box;
    sph.reserve(F.spheres.size()); sph_meta.reserve(F.spheres.size()); tri.reserve(F.tris.size()); box.reserve(F.boxes.size());
    auto convert = [&](int32_t& child, float probe) {
        if (child >= 0) return;
        if (probe != probe) { child = ~0; return; }               // absent child (NaN box): the empty leaf
        const uint32_t code = ~(uint32_t)child, first = code >> 3, n = (code & 7u) + 1u;
        LeafRec L{(uint32_t)tri.size(), (uint32_t)sph.size(), (uint32_t)box.size(), 0};
        uint32_t nt = 0, ns = 0, nb = 0;
        for (uint32_t i = first; i < first + n && i < refs.size(); i++) {
            const uint32_t kind = refs[i] >> kRefKindShift, idx = refs[i] & kRefIndexMask;
            if (kind == 0u) { T.map_sph[idx] = (uint32_t)sph.size(); sph.push_back(F.spheres[idx]); sph_meta.push_back(F.sphere_meta[idx]); ns++; }
            else if (kind == 1u) { T.map_tri[idx] = (uint32_t)tri.size(); tri.push_back(F.tris[idx]); nt++; }
            else { T.map_box[idx] = (uint32_t)box.size(); box.push_back(F.boxes[idx]); nb++; }
        }
        L.counts = nt | (ns << 8) | (nb << 16);
        if (direct && nb == 0 && (nt == 0) != (ns == 0)) {        // one run of one type: the reference names it
            const uint32_t run_first = nt ? L.tri_first : L.sph_first, run = nt ? nt : ns;
            if (run_first <= kLeafFirstMask && run <= 16u) {
                child = (int32_t)(0x80000000u | ((nt ? kLeafTris : kLeafSpheres) << kLeafKindShift) | ((run - 1u) << kLeafCountShift) | run_first);
                return;
            }
        }
        if (T.leaves.size() >= (1u << kLeafKindShift)) { T.overflow = true; return; }   // (2^29 mixed leaves: never in practice)
        child = ~(int32_t)T.leaves.size();
        T.leaves.push_back(L);
    };
    for (NodePair& nd : nodes) { convert(nd.child0, nd.lo0[0]); convert(nd.child1, nd.lo1xy[0]); }
    // primitives no leaf holds (left out of the tree by cull_never_hit) keep a place behind the others
    for (size_t i = 0; i < F.spheres.size(); i++)
        if (T.map_sph[i] == kNone) { T.map_sph[i] = (uint32_t)sph.size(); sph.push_back(F.spheres[i]); sph_meta.push_back(F.sphere_meta[i]); }
    for (size_t i = 0; i < F.tris.size(); i++)
        if (T.map_tri[i] == kNone) { T.map_tri[i] = (uint32_t)tri.size(); tri.push_back(F.tris[i]); }
    for (size_t i = 0; i < F.boxes.size(); i++)
        if (T.map_box[i] == kNone) { T.map_box[i] = (uint32_t)box.size(); box.push_back(F.boxes[i]); }
    F.spheres.swap(sph); F.sphere_meta.swap(sph_meta); F.tris.swap(tri); F.boxes.swap(box);
}

void quantise_nodes(const std::vector<NodePair>& nodes, QuantisedNodes& Q) {
    auto child_box = [](const NodePair& n, int c, float lo[3], float hi[3]) {
        if (c == 0) { lo[0] = n.lo0[0]; lo[1] = n.lo0[1]; lo[2] = n.lo0[2]; hi[0] = n.hi0x; hi[1] = n.hi0yz[0]; hi[2] = n.hi0yz[1]; }
        else { lo[0] = n.lo1xy[0]; lo[1] = n.lo1xy[1]; lo[2] = n.lo1z; hi[0] = n.hi1[0]; hi[1] = n.hi1[1]; hi[2] = n.hi1[2]; }
    };
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (const NodePair& n : nodes)
        for (int c = 0; c < 2; c++) {
            float lo[3], hi[3];
            child_box(n, c, lo, hi);
            if (!(lo[0] <= hi[0])) continue;                       // absent child (NaN bounds)
            for (int a = 0; a < 3; a++) { mn[a] = std::min<double>(mn[a], lo[a]); mx[a] = std::max<double>(mx[a], hi[a]); }
        }
    for (int a = 0; a < 3; a++) {
        if (mn[a] > mx[a]) { mn[a] = 0.0; mx[a] = 1.0; }           // no box at all
        // all boxes land in codes [2, 65533]: room for the one-code margin without clamping
        const double ext = std::max(mx[a] - mn[a], 1e-6 * std::max({std::fabs(mn[a]), std::fabs(mx[a]), 1.0}));
        Q.scale[a] = (float)(ext / 65530.0);
        if (!(Q.scale[a] > 0.0f)) Q.scale[a] = 1e-30f;
        Q.base[a] = (float)(mn[a] - 2.5 * (double)Q.scale[a]);
    }
    Q.nodes.resize(nodes.size());
    for (size_t i = 0; i < nodes.size(); i++) {
        const NodePair& n = nodes[i];
        uint32_t code[2][3];
        for (int c = 0; c < 2; c++) {
            float lo[3], hi[3];
            child_box(n, c, lo, hi);
            for (int a = 0; a < 3; a++) {
                if (!(lo[0] <= hi[0])) { code[c][a] = 0u; continue; }
                const double b = Q.base[a], sc = Q.scale[a];
                double ql = std::floor(((double)lo[a] - b) / sc) - 1.0, qh = std::ceil(((double)hi[a] - b) / sc) + 1.0;
                ql = std::min(std::max(ql, 0.0), 65535.0); qh = std::min(std::max(qh, 0.0), 65535.0);
                code[c][a] = (uint32_t)ql | ((uint32_t)qh << 16);
            }
        }
        QNode& q = Q.nodes[i];
        q.x0 = code[0][0]; q.y0 = code[0][1]; q.z0 = code[0][2]; q.child0 = n.child0;
        q.x1 = code[1][0]; q.y1 = code[1][1]; q.z1 = code[1][2]; q.child1 = n.child1;
    }
}

}  // namespace p3d
