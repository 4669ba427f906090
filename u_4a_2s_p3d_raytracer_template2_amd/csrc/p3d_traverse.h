// p3d_traverse.h -- per-lane BVH2 traversal (closest hit with scene-order tie break, any hit)
// over the flat NodePair array, with the traversal stack in LDS laid out [slot][lane].
// The slab test is ours and conservative (padded boxes); everything that decides a hit or a
// distance is the reference's primitive arithmetic from p3d_device_math.h.
#ifndef P3D_TRAVERSE_H
#define P3D_TRAVERSE_H

#include "p3d_device_math.h"
#include "p3d_device_types.h"

namespace p3d {

#define P3D_DONE ((int32_t)0x80000000)

// ------------------------------------------------------------------ per-lane work counters
struct Ctr { uint32_t closest, shadow, box, sph, tri, aab, pln; };

// ------------------------------------------------------------------ BVH traversal
struct Hit {
    float t;
    uint32_t ref;        // kind<<30 | index ; 0xFFFFFFFF = miss
    uint32_t sid;        // scene index
    uint32_t mat;
};

// ------------------------------------------------------------------ scene views
// All per-lane-indexed scene data lives in ONE blob of 16-byte quads (LaunchParams::blob,
// section offsets in quads).  GlobalScene reads it from HBM/L2/L1 with global_load_dwordx4;
// LdsScene reads the workgroup's LDS copy (small scenes: ds_read_b128 moves 4x the bytes per
// clock of the vector L1 return path and same-address lanes broadcast).
extern __shared__ __attribute__((aligned(16))) uint32_t p3d_lds[];

struct SceneOffsets { uint32_t nodes, leaves, spheres, sphere_meta, tris, tri_normals, boxes, mats, tri_quads; };

struct GlobalScene {
    const float4* q; SceneOffsets o;
    const uint4* qn;                  // quantised node pairs (QNode, p3d_device_types.h): two quads per node
    float qs[3], qb[3];               // plane = qb + code * qs
    __device__ __forceinline__ float4 ld4(uint32_t i) const { return q[i]; }
    __device__ __forceinline__ uint32_t ld1(uint32_t quad, uint32_t dw) const {
        return reinterpret_cast<const uint32_t*>(q + quad)[dw];
    }
    __device__ __forceinline__ uint2 ld2(uint32_t quad, uint32_t pair) const {
        return reinterpret_cast<const uint2*>(q + quad)[pair];
    }
};
struct LdsScene {
    SceneOffsets o;
    __device__ __forceinline__ float4 ld4(uint32_t i) const { return reinterpret_cast<const float4*>(p3d_lds)[i]; }
    __device__ __forceinline__ uint32_t ld1(uint32_t quad, uint32_t dw) const { return p3d_lds[quad * 4 + dw]; }
    __device__ __forceinline__ uint2 ld2(uint32_t quad, uint32_t pair) const {
        return reinterpret_cast<const uint2*>(p3d_lds)[quad * 2 + pair];
    }
};
// which triangle test a scene kind uses (p3d_device_math.h: hit_triangle<FAST_RCP>)
template <class SV> struct FastRcp { static constexpr bool value = false; };
template <> struct FastRcp<GlobalScene> { static constexpr bool value = true; };
// a leaf = three typed ranges into the primitive arrays, which the host stores in LEAF ORDER (LeafRec):
// x = first triangle, y = first sphere, z = first box, w = counts (triangles | spheres << 8 | boxes << 16).
// `code` = a negative child reference (p3d_device_types.h: kLeaf*).  In scenes read from HBM a leaf that is one run of
// triangles or of spheres -- nearly all of them -- is named by the reference itself: no record to fetch between the node
// and its primitives (one dependent fetch less per leaf visit).  Mixed leaves and leaves with boxes go through their
// LeafRec, and so does every leaf of a scene small enough for LDS (the host emits no direct references there: the
// record is one LDS read, cheaper than decoding in kernels that are bound by instruction issue).
__device__ __forceinline__ uint4 sv_leaf(const GlobalScene& sv, int32_t code) {
    const uint32_t c = (uint32_t)code, kind = (c >> kLeafKindShift) & 3u;
    uint4 L = make_uint4(c & kLeafFirstMask, c & kLeafFirstMask, 0u, 0u);
    const uint32_t n = ((c >> kLeafCountShift) & 15u) + 1u;
    L.w = kind == kLeafTris ? n : (n << 8);
    if (kind == kLeafIndirect) {
        const float4 t = sv.ld4(sv.o.leaves + ~c);
        L = make_uint4(__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w));
    }
    return L;
}
__device__ __forceinline__ uint4 sv_leaf(const LdsScene& sv, int32_t code) {
    const float4 t = sv.ld4(sv.o.leaves + ~(uint32_t)code);
    return make_uint4(__float_as_uint(t.x), __float_as_uint(t.y), __float_as_uint(t.z), __float_as_uint(t.w));
}
template <class SV> __device__ __forceinline__ float4 sv_sphere(const SV& sv, uint32_t i) { return sv.ld4(sv.o.spheres + i); }
template <class SV> __device__ __forceinline__ PrimMeta sv_sphere_meta(const SV& sv, uint32_t i) {
    uint2 m = sv.ld2(sv.o.sphere_meta, i); PrimMeta r; r.scene_id = m.x; r.material = m.y; return r;
}
template <class SV> __device__ __forceinline__ void sv_tri(const SV& sv, uint32_t i, float4& a, float4& b, float4& c) {
    uint32_t q = sv.o.tris + i * sv.o.tri_quads; a = sv.ld4(q); b = sv.ld4(q + 1); c = sv.ld4(q + 2);
}
template <class SV> __device__ __forceinline__ float4 sv_tri_normal(const SV& sv, uint32_t i) { return sv.ld4(sv.o.tri_normals + i); }
template <class SV> __device__ __forceinline__ void sv_box(const SV& sv, uint32_t i, float4& a, float4& b) {
    uint32_t q = sv.o.boxes + i * 2u; a = sv.ld4(q); b = sv.ld4(q + 1);
}
template <class SV> __device__ __forceinline__ void sv_mat(const SV& sv, uint32_t m, float4& a, float4& b, float4& c) {
    uint32_t q = sv.o.mats + m * 3u; a = sv.ld4(q); b = sv.ld4(q + 1); c = sv.ld4(q + 2);
}

// A lane's traversal stack lives in its wave's LDS region, laid out [slot][lane].  An entry is a
// node ref plus the ray's entry distance into that node (popped entries farther than the best hit
// are skipped without touching memory).  Two layouts behind one interface:
//  WideStack  (the LDS-scene stack until late in round 2, kept for comparison builds) 8-byte entries {ref, t as f32}:
//             one ds_write_b64 / ds_read_b64, popped entries farther than the best hit skipped in a loop.
//  SlimStack  (round 1; superseded by RefStack below, kept for comparison builds: -DP3D_HBM_STACK=SlimStack
//             -DP3D_HBM_STACK_DWORDS=96u) 6-byte entries: refs in one [slot][lane] plane, t truncated to its upper
//             16 bits in a second one.  Scenes read from HBM -- a depth-25 stack is 12.8 KB per wave wide, 9.6 KB
//             slim, which is the difference between 12 and 16 resident waves per CU for kernels that
//             wait on memory most of the time.  Truncation moves a positive t towards zero and best.t
//             is never negative, so a stored distance never exceeds the true one by more than it may:
//             the skip test stays conservative (it only ever visits a node it could have skipped).
struct TravStack {
    uint32_t* region;    // this wave's stack region
    uint32_t lane, slots;
};
constexpr uint32_t kWideStackDwords = 128, kSlimStackDwords = 96;    // per slot per wave

struct WideStack {
    uint2* base; int sp;
    __device__ __forceinline__ explicit WideStack(const TravStack& r)
        : base(reinterpret_cast<uint2*>(r.region) + r.lane), sp(0) {}
    __device__ __forceinline__ void push(int32_t node, float t) {
        base[sp * 64] = make_uint2((uint32_t)node, __float_as_uint(t)); sp++;
    }
    // next entry not farther than tmax, or false when the stack is empty
    __device__ __forceinline__ bool pop(float tmax, int32_t& node) {
        while (sp > 0) {
            sp--;
            const uint2 e = base[sp * 64];
            if (__uint_as_float(e.y) <= tmax) { node = (int32_t)e.x; return true; }
        }
        return false;
    }
    __device__ __forceinline__ bool pop(int32_t& node) {
        if (sp == 0) return false;
        sp--; node = (int32_t)base[sp * 64].x;
        return true;
    }
};
struct SlimStack {
    uint32_t* refs; uint16_t* dist; int sp;
    __device__ __forceinline__ explicit SlimStack(const TravStack& r)
        : refs(r.region + r.lane), dist(reinterpret_cast<uint16_t*>(r.region + r.slots * 64) + r.lane), sp(0) {}
    __device__ __forceinline__ void push(int32_t node, float t) {
        refs[sp * 64] = (uint32_t)node; dist[sp * 64] = (uint16_t)(__float_as_uint(t) >> 16); sp++;
    }
    __device__ __forceinline__ bool pop(float tmax, int32_t& node) {
        while (sp > 0) {
            sp--;
            const float t = __uint_as_float((uint32_t)dist[sp * 64] << 16);
            if (t <= tmax) { node = (int32_t)refs[sp * 64]; return true; }
        }
        return false;
    }
    __device__ __forceinline__ bool pop(int32_t& node) {
        if (sp == 0) return false;
        sp--; node = (int32_t)refs[sp * 64];
        return true;
    }
};
// 4-byte entries: node refs only.  A popped node is always visited; if it lies behind the best hit so far its
// children fail their slab tests (one wasted fetch), which is the price for 6.4 instead of 9.6 KB of LDS per wave
// at tree depth 25 -- scenes read from HBM are bound by how many waves are in flight to hide fetch latency.
struct RefStack {
    uint32_t* refs; int sp;
    __device__ __forceinline__ explicit RefStack(const TravStack& r) : refs(r.region + r.lane), sp(0) {}
    __device__ __forceinline__ void push(int32_t node, float) { refs[sp * 64] = (uint32_t)node; sp++; }
    __device__ __forceinline__ bool pop(float, int32_t& node) { return pop(node); }
    __device__ __forceinline__ bool pop(int32_t& node) {
        if (sp == 0) return false;
        sp--; node = (int32_t)refs[sp * 64];
        return true;
    }
};
// measured, 1920x1080 depth 4: dragon 2.12 -> 1.88 ms (tree), 10^6 primitives 3.32 -> 3.19 ms (wavefront) and
// 5.3 -> 4.3 ms (tile) against the 6-byte SlimStack: more resident waves beat pop-time pruning
#ifndef P3D_HBM_STACK
#define P3D_HBM_STACK RefStack
#endif
template <class SV> struct StackOf;
// scenes in LDS: reference-only slots too since round 2 (the pruning pop loop of WideStack compiled to ~25 scalar
// instructions per pop; without it the level-1 kernel runs 377 -> 360 SALU per wave, config 2 0.1279 -> 0.1263 ms,
// config 4 4.91 -> 4.81 ms); comparison build: -DP3D_LDS_STACK=WideStack -DP3D_LDS_STACK_DWORDS=128u
#ifndef P3D_LDS_STACK
#define P3D_LDS_STACK RefStack
#endif
template <> struct StackOf<LdsScene> { typedef P3D_LDS_STACK type; };
template <> struct StackOf<GlobalScene> { typedef P3D_HBM_STACK type; };

struct SlabRay { float kx, ky, kz, ix, iy, iz; };     // i = 1/d, k = -o/d: a plane's distance is fma(plane, i, k)

// 1 / d for the slab test, clamped to +-1e30 (one v_med3).  With an INFINITE reciprocal (d == 0) the fused form below
// gives NaN for one plane and an infinity for the other whenever the box spans the coordinate origin on that axis, and
// a ray inside that slab was culled; with 1e30 both distances are finite (or properly signed infinities) and the axis
// behaves like any other: the slab's two distances keep their signs, and their size is far beyond any hit distance.
__device__ __forceinline__ float slab_rcp(float d) {
#ifdef P3D_SLAB_RCP_UNCLAMPED       // (diagnostic build: shows tests/test_gpu_random_scenes.py::test_rays_with_a_zero_direction_component failing)
    return __builtin_amdgcn_rcpf(d);
#else
    return __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(d), -1e30f, 1e30f);
#endif
}
// Scenes in LDS: f32 boxes, plane distance = fma(plane, i, -o * i)
__device__ __forceinline__ SlabRay make_slab(const LdsScene&, const Ray& r) {
    SlabRay s;
    // 1-ulp hardware reciprocals are enough here: the slab test only has to be conservative
    // (boxes are padded by >= 1e-3, far above a relative 1e-7), it never decides a hit
    s.ix = slab_rcp(r.d.x); s.iy = slab_rcp(r.d.y); s.iz = slab_rcp(r.d.z);
    s.kx = -r.o.x * s.ix; s.ky = -r.o.y * s.iy; s.kz = -r.o.z * s.iz;
    return s;
}
// Scenes read from HBM: 16-bit plane codes, plane = qb + code * qs, so the distance is fma(code, qs * i, (qb - o) * i):
// the same one FMA per plane, with the de-quantisation folded into the per-ray constants
__device__ __forceinline__ SlabRay make_slab(const GlobalScene& g, const Ray& r) {
    SlabRay s;
    const float ix = slab_rcp(r.d.x), iy = slab_rcp(r.d.y), iz = slab_rcp(r.d.z);
    s.ix = g.qs[0] * ix; s.iy = g.qs[1] * iy; s.iz = g.qs[2] * iz;
    s.kx = (g.qb[0] - r.o.x) * ix; s.ky = (g.qb[1] - r.o.y) * iy; s.kz = (g.qb[2] - r.o.z) * iz;
    return s;
}
// conservative slab test against a padded box; returns entry distance in tn.  One fused multiply-add per plane
// (this arithmetic is ours, not the reference's: it only has to be conservative).  A direction component of zero
// makes i infinite and a distance NaN where (plane - o) * i would have been an infinity: fminf / fmaxf drop NaNs,
// so that axis is then ignored -- a few more visits for axis-parallel rays, never a missed box.
__device__ __forceinline__ bool slab(const SlabRay& s, float lx, float ly, float lz, float hx,
                                     float hy, float hz, float tlimit, float& tn) {
    float ax = __builtin_fmaf(lx, s.ix, s.kx), bx = __builtin_fmaf(hx, s.ix, s.kx);
    float ay = __builtin_fmaf(ly, s.iy, s.ky), by = __builtin_fmaf(hy, s.iy, s.ky);
    float az = __builtin_fmaf(lz, s.iz, s.kz), bz = __builtin_fmaf(hz, s.iz, s.kz);
    float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    t1 = t1 * 1.0000005f + 1e-30f;
    tn = t0;
    // (t0 <= t1 && t0 <= tlimit) as ONE comparison against min(t1, tlimit): a mask AND on the scalar unit less per child.
    // t0 itself stays in the comparison, so a ray with NaNs in it (t0 and t1 both NaN) still misses every box -- folding
    // t1 >= 0 in as max(t0, 0) would not: fmax drops the NaN (profiles/r02_experiments_traversal.txt).
    return (t0 <= fminf(t1, tlimit)) && (t1 >= 0.0f);
}

// One visit of node pair `n`: both children's slab tests (hit flags, entry distances) and their references.
__device__ __forceinline__ void node_test(const LdsScene& sv, const SlabRay& s, int32_t n, float tlimit, bool& h0, bool& h1,
                                          float& tn0, float& tn1, int32_t& c0, int32_t& c1) {
    const uint32_t b = sv.o.nodes + (uint32_t)n * 4u;
    const float4 q0 = sv.ld4(b), q1 = sv.ld4(b + 1), q2 = sv.ld4(b + 2), q3 = sv.ld4(b + 3);
    h0 = slab(s, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tlimit, tn0);
    h1 = slab(s, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tlimit, tn1);
    c0 = __float_as_int(q3.x); c1 = __float_as_int(q3.y);
}
// 32-byte node pairs: per child three dwords of (lo | hi << 16) plane codes + the reference -- half the fetches of the
// f32 node (these walks are bound by the L1-miss path, tools/ubench/gather_rate.hip), and two to four nodes of a
// depth-first run share a cache line.  The codes convert exactly (integers < 2^16); the boxes only grew (host:
// quantise_nodes), and the slab test only has to be conservative.
__device__ __forceinline__ void node_test(const GlobalScene& sv, const SlabRay& s, int32_t n, float tlimit, bool& h0, bool& h1,
                                          float& tn0, float& tn1, int32_t& c0, int32_t& c1) {
    const uint4 a = sv.qn[(uint32_t)n * 2u], b = sv.qn[(uint32_t)n * 2u + 1u];
    h0 = slab(s, (float)(a.x & 0xFFFFu), (float)(a.y & 0xFFFFu), (float)(a.z & 0xFFFFu),
              (float)(a.x >> 16), (float)(a.y >> 16), (float)(a.z >> 16), tlimit, tn0);
    h1 = slab(s, (float)(b.x & 0xFFFFu), (float)(b.y & 0xFFFFu), (float)(b.z & 0xFFFFu),
              (float)(b.x >> 16), (float)(b.y >> 16), (float)(b.z >> 16), tlimit, tn1);
    c0 = (int32_t)a.w; c1 = (int32_t)b.w;
}

// "t < closest_t" in scene order == nearest, lowest scene index on ties
__device__ __forceinline__ void take_closer(Hit& best, bool h, float t, uint32_t ref, uint32_t sid, uint32_t mat) {
    if (h && (t < best.t || (t == best.t && sid < best.sid))) { best.t = t; best.ref = ref; best.sid = sid; best.mat = mat; }
}

// One leaf.  No per-primitive reference to fetch and no kind to dispatch on: the leaf record names a run of
// triangles, a run of spheres and a run of boxes in arrays the host stored in leaf order, so a test is one
// round trip to the primitive's record (the reference list cost a dependent load + a kind switch per primitive).
template <bool COUNT, class SV>
__device__ __forceinline__ void leaf_closest(const LaunchParams& P, const SV& sv, const Ray& r, int32_t leaf,
                                             Hit& best, Ctr& ctr) {
    const uint4 L = sv_leaf(sv, leaf);
    const uint32_t nt = L.w & 0xFFu, ns = (L.w >> 8) & 0xFFu, nb = L.w >> 16;
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < nt; i++) {
        float4 a, b, c;
        sv_tri(sv, L.x + i, a, b, c);
        if (COUNT) ctr.tri++;
        float t;
        const bool h = hit_triangle<FastRcp<SV>::value>(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), t);
        take_closer(best, h, t, (1u << kRefKindShift) | (L.x + i), __float_as_uint(a.w), __float_as_uint(b.w));
    }
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < ns; i++) {
        const float4 sp = sv_sphere(sv, L.y + i);
        const PrimMeta m = sv_sphere_meta(sv, L.y + i);
        if (COUNT) ctr.sph++;
        float t;
        const bool h = hit_sphere(r, mk(sp.x, sp.y, sp.z), sp.w, t);
        take_closer(best, h, t, (0u << kRefKindShift) | (L.y + i), m.scene_id, m.material);
    }
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < nb; i++) {
        float4 a, b; V3 nn;
        sv_box(sv, L.z + i, a, b);
        if (COUNT) ctr.aab++;
        float t;
        const bool h = hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
        take_closer(best, h, t, (2u << kRefKindShift) | (L.z + i), __float_as_uint(a.w), __float_as_uint(b.w));
    }
}

// Shape of the per-lane walk loop.  Scenes in LDS: "while-while" (inner loop over nodes, then one leaf) -- fewest
// instructions, and those kernels are bound by instruction issue.  Scenes read from HBM: ONE flat loop in which every
// lane takes a node step or a leaf step per iteration ("if-if"): a lane at a leaf no longer waits for the other lanes to
// finish descending, so the dependent chain of a wave is as long as its slowest lane's, not the sum of the phases.
// Measured: dragon (tree schedule) 1.58 -> 1.26 ms, 10^6 random primitives 2.74 -> 2.67 ms; config 2 with the flat
// loop 353 -> 374 SALU per wave and 0.1273 -> 0.1278 ms, which is why LDS scenes keep while-while.
template <class SV> struct FlatWalk { static constexpr bool value = false; };
template <> struct FlatWalk<GlobalScene> { static constexpr bool value = true; };

// closest hit over planes (unbounded, outside the BVH) + BVH
template <bool COUNT, class SV>
__device__ __forceinline__ Hit closest_hit(const LaunchParams& P, const SV& sv, const Ray& r, TravStack region, Ctr& ctr) {
    Hit best; best.t = 3.402823466e+38f; best.ref = 0xFFFFFFFFu; best.sid = 0xFFFFFFFFu; best.mat = 0;
    if (COUNT) ctr.closest++;
    for (uint32_t i = 0; i < P.n_planes; i++) {
        PlaneRec pl = P.planes[i];
        float t;
        if (COUNT) ctr.pln++;
        if (hit_plane(r, mk(pl.nx, pl.ny, pl.nz), pl.d, t)) {
            PrimMeta m = P.plane_meta[i];
            if (t < best.t || (t == best.t && m.scene_id < best.sid)) {
                best.t = t; best.ref = (3u << kRefKindShift) | i; best.sid = m.scene_id; best.mat = m.material;
            }
        }
    }
    SlabRay s = make_slab(sv, r);
    typename StackOf<SV>::type st(region);
    int32_t cur = 0;
    if (FlatWalk<SV>::value) {
    while (cur != P3D_DONE) {
        if (cur >= 0) {
            float tn0, tn1; bool h0, h1; int32_t c0, c1;
            node_test(sv, s, cur, best.t, h0, h1, tn0, tn1, c0, c1);
            if (COUNT) ctr.box += 2;
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                int32_t nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
                float fart = swap ? tn0 : tn1;
                st.push(farc, fart);
                cur = nearc;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else if (!st.pop(best.t, cur)) cur = P3D_DONE;
        } else {
            leaf_closest<COUNT>(P, sv, r, cur, best, ctr);
            if (!st.pop(best.t, cur)) cur = P3D_DONE;
        }
    }
    return best;
    }
    while (cur != P3D_DONE) {
        while (cur >= 0) {
            float tn0, tn1; bool h0, h1; int32_t c0, c1;
            node_test(sv, s, cur, best.t, h0, h1, tn0, tn1, c0, c1);
            if (COUNT) ctr.box += 2;
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                int32_t nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
                float fart = swap ? tn0 : tn1;
                st.push(farc, fart);
                cur = nearc;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else if (!st.pop(best.t, cur)) cur = P3D_DONE;
        }
        if (cur != P3D_DONE) {
            leaf_closest<COUNT>(P, sv, r, cur, best, ctr);
            if (!st.pop(best.t, cur)) cur = P3D_DONE;
        }
    }
    return best;
}

// any hit with t < tmax (tmax = +inf, bounded == false: "any intercepts() at all", the
// NONE-mode shadow loop of RT/main.cpp:480-487)
template <bool COUNT, class SV>
__device__ __forceinline__ bool leaf_any(const LaunchParams& P, const SV& sv, const Ray& r, int32_t leaf, bool bounded,
                                         float tmax, Ctr& ctr) {
    const uint4 L = sv_leaf(sv, leaf);
    const uint32_t nt = L.w & 0xFFu, ns = (L.w >> 8) & 0xFFu, nb = L.w >> 16;
    bool occluded = false;
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < nt; i++) {
        float4 a, b, c;
        sv_tri(sv, L.x + i, a, b, c);
        if (COUNT) ctr.tri++;
        float t;
        const bool h = hit_triangle<FastRcp<SV>::value>(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), t);
        occluded = occluded | (h && (!bounded || t < tmax));
    }
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < ns; i++) {
        const float4 sp = sv_sphere(sv, L.y + i);
        if (COUNT) ctr.sph++;
        float t;
        const bool h = hit_sphere(r, mk(sp.x, sp.y, sp.z), sp.w, t);
        occluded = occluded | (h && (!bounded || t < tmax));
    }
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < nb; i++) {
        float4 a, b; V3 nn;
        sv_box(sv, L.z + i, a, b);
        if (COUNT) ctr.aab++;
        float t;
        const bool h = hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
        occluded = occluded | (h && (!bounded || t < tmax));
    }
    return occluded;
}

template <bool COUNT, class SV>
__device__ __forceinline__ bool any_hit(const LaunchParams& P, const SV& sv, const Ray& r, bool bounded, float tmax,
                                        TravStack region, Ctr& ctr) {
    if (COUNT) ctr.shadow++;
    if (P.n_planes) {
        // planes: always candidates without an accelerator; behind the default [-1,1]^3 box
        // in the reference's BVH / grid
        bool gate = !bounded || ref_unit_box_hit(r);
        if (gate) {
            for (uint32_t i = 0; i < P.n_planes; i++) {
                PlaneRec pl = P.planes[i];
                float t;
                if (COUNT) ctr.pln++;
                if (hit_plane(r, mk(pl.nx, pl.ny, pl.nz), pl.d, t) && (!bounded || t < tmax)) return true;
            }
        }
    }
    SlabRay s = make_slab(sv, r);
    float tlimit = bounded ? tmax : 3.402823466e+38f;
    typename StackOf<SV>::type st(region);
    int32_t cur = 0;
    if (FlatWalk<SV>::value) {
    bool occluded = false;
    while (cur != P3D_DONE) {
        if (cur >= 0) {
            float tn0, tn1; bool h0, h1; int32_t c0, c1;
            node_test(sv, s, cur, tlimit, h0, h1, tn0, tn1, c0, c1);
            if (COUNT) ctr.box += 2;
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                st.push(swap ? c0 : c1, 0.0f);
                cur = swap ? c1 : c0;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else if (!st.pop(cur)) cur = P3D_DONE;
        } else {
            if (leaf_any<COUNT>(P, sv, r, cur, bounded, tmax, ctr)) { occluded = true; cur = P3D_DONE; }
            else if (!st.pop(cur)) cur = P3D_DONE;
        }
    }
    return occluded;
    }
    while (cur != P3D_DONE) {
        while (cur >= 0) {
            float tn0, tn1; bool h0, h1; int32_t c0, c1;
            node_test(sv, s, cur, tlimit, h0, h1, tn0, tn1, c0, c1);
            if (COUNT) ctr.box += 2;
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                st.push(swap ? c0 : c1, 0.0f);
                cur = swap ? c1 : c0;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else if (!st.pop(cur)) cur = P3D_DONE;
        }
        if (cur != P3D_DONE) {
            if (leaf_any<COUNT>(P, sv, r, cur, bounded, tmax, ctr)) return true;
            if (!st.pop(cur)) cur = P3D_DONE;
        }
    }
    return false;
}


// ------------------------------------------------------------------ work-sharing per-lane walk (scenes read from HBM)
// What the per-wave timelines of round 3 showed (profiles/r03_timelines.txt): on scenes read from HBM a launch lasts as
// long as its slowest WAVES, and a wave as long as its slowest LANE -- on the dragon (100 000 triangles the reference's
// epsilon makes invisible, SURVEY Q7: a ray through them never finds the hit that would end its walk) single waves live
// 1 ms of a 1.3 ms launch with a handful of lanes walking and 50-60 idle, while the chip runs a third full.
// So the lanes of a wave SHARE their walks.  All 64 lanes stay in one wave-uniform loop; a lane that has finished (or
// never had a ray) is idle.  When enough lanes are idle and some lane has pending subtrees on its stack, each idle lane
// takes the BOTTOM entry (the subtree nearest the root: the biggest piece of pending work) of one such lane's stack,
// together with that lane's ray (ds_bpermute), and walks it with its own stack as a helper of that ray's OWNER.  Results
// are merged through the wave's LDS: closest hit = minimum over (t, scene id) of everything any lane found for the owner
// (64-bit ds_min: the reference's "nearest, lowest scene index on ties" does not depend on visit order), any
// hit = OR.  Helpers prune with the owner's best distance so far (read from LDS at every node step), so work stolen
// ahead of a near hit dies quickly.  Same hits, same bits; only the test COUNTS differ from the private walk (a helper
// may test a box the owner would have pruned a moment later).
//
// Per-wave LDS ("share region", behind the wave's stack region): [0,64) donor table of a steal round, [64,192) the
// owners' best keys (u64: t bits << 32 | scene id), [192,320) ref + material of the key's holder, [320,384) any-hit flags.
constexpr unsigned long long kShareNoHit = 0x7F7FFFFFFFFFFFFFull;      // t = FLT_MAX, id = none

__device__ __forceinline__ void wave_lds_fence() {
    // LDS operations of one wave execute in program order; this only keeps the compiler from moving them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint32_t share_lane_rank(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ float shfl_f(float v, uint32_t src) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src << 2), __float_as_int(v)));
}
__device__ __forceinline__ uint32_t shfl_u(uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
}

// One steal round (all 64 lanes, converged): pairs the k-th idle lane with the k-th lane that has a pending stack
// entry.  Returns true in the lanes that took work; their `cur` is the stolen node and ray / slab constants / owner
// are the victim's.  `limit` travels too (closest hit: the victim's best distance; any hit: its distance bound).
__device__ __forceinline__ bool steal_round(uint32_t* stack_region, uint32_t* share, uint32_t lane, uint64_t busy, uint64_t donors,
                                            int32_t& cur, int& sp, int& bot, Ray& r, SlabRay& s, float& limit, uint32_t& owner) {
    const uint64_t idle = ~busy;
    const bool is_idle = ((idle >> lane) & 1ull) != 0, is_donor = ((donors >> lane) & 1ull) != 0;
    const uint32_t n_i = (uint32_t)__popcll(idle), n_d = (uint32_t)__popcll(donors), n = n_i < n_d ? n_i : n_d;
    const uint32_t rank_i = share_lane_rank(idle), rank_d = share_lane_rank(donors);
    volatile uint32_t* tab = share;
    const bool robbed = is_donor && rank_d < n;
    if (robbed) tab[rank_d] = lane;
    wave_lds_fence();
    const bool thief = is_idle && rank_i < n;
    const uint32_t victim = thief ? tab[rank_i] : lane;
    const int v_bot = (int)shfl_u((uint32_t)bot, victim);
    const Ray vr = {mk(shfl_f(r.o.x, victim), shfl_f(r.o.y, victim), shfl_f(r.o.z, victim)),
                    mk(shfl_f(r.d.x, victim), shfl_f(r.d.y, victim), shfl_f(r.d.z, victim))};
    const SlabRay vs = {shfl_f(s.kx, victim), shfl_f(s.ky, victim), shfl_f(s.kz, victim),
                        shfl_f(s.ix, victim), shfl_f(s.iy, victim), shfl_f(s.iz, victim)};
    const float v_limit = shfl_f(limit, victim);
    const uint32_t v_owner = shfl_u(owner, victim);
    if (thief) {
        cur = (int32_t)reinterpret_cast<volatile uint32_t*>(stack_region)[(uint32_t)v_bot * 64u + victim];
        sp = 0; bot = 0; r = vr; s = vs; limit = v_limit; owner = v_owner;
    }
    if (robbed) { bot++; if (bot == sp) { bot = 0; sp = 0; } }
    wave_lds_fence();
    return thief;
}

// publish a lane's best hit for the ray it works on: key by 64-bit minimum, ref / material by whoever holds the minimum
__device__ __forceinline__ unsigned long long share_key(const Hit& h) {
    // (t + 0.0f: -0.0f and +0.0f compare equal in the reference's "t < closest_t", so they must give one key)
    return ((unsigned long long)__float_as_uint(h.t + 0.0f) << 32) | (unsigned long long)h.sid;
}
__device__ __forceinline__ void share_publish(uint32_t* share, uint32_t owner, const Hit& h) {
    if (h.ref == 0xFFFFFFFFu) return;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(share + 64);
    const unsigned long long key = share_key(h);
    atomicMin(&keys[owner], key);
    if (reinterpret_cast<volatile unsigned long long*>(keys)[owner] == key)
        reinterpret_cast<volatile unsigned long long*>(share + 192)[owner] = (unsigned long long)h.ref | ((unsigned long long)h.mat << 32);
}

// closest hit, work shared among the lanes of the wave.  Must be called by all 64 lanes together; `active` = this lane
// has a ray.  Same result as closest_hit().
template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_shared(const LaunchParams& P, const GlobalScene& sv, const Ray& ray, bool active,
                                                  TravStack region, uint32_t* share, Ctr& ctr) {
    Hit best; best.t = 3.402823466e+38f; best.ref = 0xFFFFFFFFu; best.sid = 0xFFFFFFFFu; best.mat = 0;
    if (active) {
        if (COUNT) ctr.closest++;
        for (uint32_t i = 0; i < P.n_planes; i++) {
            PlaneRec pl = P.planes[i];
            float t;
            if (COUNT) ctr.pln++;
            if (hit_plane(ray, mk(pl.nx, pl.ny, pl.nz), pl.d, t)) {
                PrimMeta m = P.plane_meta[i];
                if (t < best.t || (t == best.t && m.scene_id < best.sid)) {
                    best.t = t; best.ref = (3u << kRefKindShift) | i; best.sid = m.scene_id; best.mat = m.material;
                }
            }
        }
    }
    const uint32_t lane = region.lane;
    Ray r = ray;
    SlabRay s = make_slab(sv, r);
    uint32_t* refs = region.region + lane;
    int sp = 0, bot = 0;
    int32_t cur = active ? 0 : P3D_DONE;
    uint32_t owner = lane;
    bool shared = false;                       // wave-uniform: some lane of this wave has taken over work in this query
    volatile uint32_t* key_hi = share + 64;    // high dword of keys[o] at [2 * o + 1]
    const uint32_t min_idle = P.share_min_idle;
    for (;;) {
        const uint64_t busy = __ballot(cur != P3D_DONE);
        if (busy == 0) break;
        if (64u - (uint32_t)__popcll(busy) >= min_idle) {
            const uint64_t donors = __ballot(cur != P3D_DONE && sp > bot);
            if (donors != 0) {
                if (!shared) {
                    reinterpret_cast<volatile unsigned long long*>(share + 64)[lane] = kShareNoHit;
                    shared = true;
                    wave_lds_fence();
                }
                // an idle lane's result so far belongs to the ray it worked on: hand it in before taking other work
                if (cur == P3D_DONE) { share_publish(share, owner, best); best.ref = 0xFFFFFFFFu; }
                wave_lds_fence();
                float limit = best.t;
                const bool took = steal_round(region.region, share, lane, busy, donors, cur, sp, bot, r, s, limit, owner);
                if (took) { best.t = limit; best.sid = 0xFFFFFFFFu; best.mat = 0; }
            }
        }
        if (cur >= 0) {
            float tl = best.t;
            if (shared) tl = fminf(tl, __uint_as_float(key_hi[2u * owner + 1u]));   // what any lane found for this ray so far
            float tn0, tn1; bool h0, h1; int32_t c0, c1;
            node_test(sv, s, cur, tl, h0, h1, tn0, tn1, c0, c1);
            if (COUNT) ctr.box += 2;
            if (h0 && h1) {
                const bool swap = tn1 < tn0;
                refs[sp * 64] = (uint32_t)(swap ? c0 : c1); sp++;
                cur = swap ? c1 : c0;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else if (sp > bot) { sp--; cur = (int32_t)refs[sp * 64]; if (sp == bot) { sp = 0; bot = 0; } }
            else cur = P3D_DONE;
        } else if (cur != P3D_DONE) {
            const float before = best.t; const uint32_t sid_before = best.sid;
            leaf_closest<COUNT>(P, sv, r, cur, best, ctr);
            if (shared && (best.t != before || best.sid != sid_before))          // let the other lanes of this ray prune with it
                atomicMin(&reinterpret_cast<unsigned long long*>(share + 64)[owner], share_key(best));
            if (sp > bot) { sp--; cur = (int32_t)refs[sp * 64]; if (sp == bot) { sp = 0; bot = 0; } }
            else cur = P3D_DONE;
        }
    }
    if (shared) {
        wave_lds_fence();
        share_publish(share, owner, best);
        wave_lds_fence();
        const unsigned long long k = reinterpret_cast<volatile unsigned long long*>(share + 64)[lane];
        const unsigned long long aq = reinterpret_cast<volatile unsigned long long*>(share + 192)[lane];
        const uint2 a = make_uint2((uint32_t)aq, (uint32_t)(aq >> 32));
        best.t = 3.402823466e+38f; best.ref = 0xFFFFFFFFu; best.sid = 0xFFFFFFFFu; best.mat = 0;
        if (active && k != kShareNoHit) { best.t = __uint_as_float((uint32_t)(k >> 32)); best.sid = (uint32_t)k; best.ref = a.x; best.mat = a.y; }
        wave_lds_fence();
    }
    return best;
}

// any hit with t < tmax (bounded) or at all, work shared like closest_hit_shared().  All 64 lanes together; `need` =
// this lane has a shadow ray.  Same result as any_hit().
template <bool COUNT>
__device__ __forceinline__ bool any_hit_shared(const LaunchParams& P, const GlobalScene& sv, const Ray& ray, bool need, bool bounded,
                                               float tmax, TravStack region, uint32_t* share, Ctr& ctr) {
    bool occ_own = false;
    if (need) {
        if (COUNT) ctr.shadow++;
        if (P.n_planes) {
            const bool gate = !bounded || ref_unit_box_hit(ray);
            for (uint32_t i = 0; i < P.n_planes; i++) {
                PlaneRec pl = P.planes[i];
                float t;
                if (gate && !occ_own) {
                    if (COUNT) ctr.pln++;
                    if (hit_plane(ray, mk(pl.nx, pl.ny, pl.nz), pl.d, t) && (!bounded || t < tmax)) occ_own = true;
                }
            }
        }
    }
    const uint32_t lane = region.lane;
    Ray r = ray;
    SlabRay s = make_slab(sv, r);
    float limit = bounded ? tmax : 3.402823466e+38f;
    uint32_t* refs = region.region + lane;
    int sp = 0, bot = 0;
    int32_t cur = (need && !occ_own) ? 0 : P3D_DONE;
    uint32_t owner = lane;
    bool shared = false;
    volatile uint32_t* flags = share + 320;
    const uint32_t min_idle = P.share_min_idle;
    for (;;) {
        const uint64_t busy = __ballot(cur != P3D_DONE);
        if (busy == 0) break;
        if (64u - (uint32_t)__popcll(busy) >= min_idle) {
            const uint64_t donors = __ballot(cur != P3D_DONE && sp > bot);
            if (donors != 0) {
                if (!shared) { flags[lane] = occ_own ? 1u : 0u; shared = true; wave_lds_fence(); }
                // a ray some lane has found an occluder for needs no more work
                if (cur != P3D_DONE && flags[owner] != 0u) { cur = P3D_DONE; sp = 0; bot = 0; }
                const uint64_t busy2 = __ballot(cur != P3D_DONE), donors2 = __ballot(cur != P3D_DONE && sp > bot);
                if (donors2 != 0) (void)steal_round(region.region, share, lane, busy2, donors2, cur, sp, bot, r, s, limit, owner);
            }
        }
        if (cur >= 0) {
            float tn0, tn1; bool h0, h1; int32_t c0, c1;
            node_test(sv, s, cur, limit, h0, h1, tn0, tn1, c0, c1);
            if (COUNT) ctr.box += 2;
            if (h0 && h1) {
                const bool swap = tn1 < tn0;
                refs[sp * 64] = (uint32_t)(swap ? c0 : c1); sp++;
                cur = swap ? c1 : c0;
            } else if (h0) cur = c0;
            else if (h1) cur = c1;
            else if (sp > bot) { sp--; cur = (int32_t)refs[sp * 64]; if (sp == bot) { sp = 0; bot = 0; } }
            else cur = P3D_DONE;
        } else if (cur != P3D_DONE) {
            if (leaf_any<COUNT>(P, sv, r, cur, bounded, limit, ctr)) {
                if (owner == lane) occ_own = true;
                if (shared) flags[owner] = 1u;
                cur = P3D_DONE; sp = 0; bot = 0;
            } else if (sp > bot) { sp--; cur = (int32_t)refs[sp * 64]; if (sp == bot) { sp = 0; bot = 0; } }
            else cur = P3D_DONE;
        }
    }
    if (shared) {
        wave_lds_fence();
        occ_own = occ_own || flags[lane] != 0u;
        wave_lds_fence();
    }
    return need && occ_own;
}


// ------------------------------------------------------------------ uniform grid (accel 1, RT/grid.cpp)
// The reference's GRID mode is NOT "the same hits through another structure": Grid::Traverse finds the
// closest hit PER CELL and accepts it only if it lies before the cell's exit (RT/grid.cpp:265-309), planes
// sit in the cells their default [-1,1]^3 box covers and are no closest-hit candidates anywhere
// else, and a shadow ray that misses the grid's box counts as SHADOWED (RT/grid.cpp:327-328).  So GRID mode
// walks a real grid here: the reference's cell-count formula and cell lists (built by grid_builder.cpp with
// the reference's float arithmetic), its slab clip and Amanatides-Woo set-up evaluated in the same order --
// the DDA state in double precision like the reference's -- and, per cell, the primitives in scene order.
// Per lane: cells are a few primitives each and rays diverge after the first step.
template <bool COUNT, class SV>
__device__ __forceinline__ bool prim_test(const LaunchParams& P, const SV& sv, const Ray& r, uint32_t ref, float& t,
                                          uint32_t& sid, uint32_t& mat, Ctr& ctr) {
    const uint32_t kind = ref >> kRefKindShift, idx = ref & kRefIndexMask;
    if (kind == 1u) {
        float4 a, b, c;
        sv_tri(sv, idx, a, b, c);
        if (COUNT) ctr.tri++;
        sid = __float_as_uint(a.w); mat = __float_as_uint(b.w);
        return hit_triangle<FastRcp<SV>::value>(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), t);
    } else if (kind == 0u) {
        const float4 s = sv_sphere(sv, idx);
        const PrimMeta m = sv_sphere_meta(sv, idx);
        if (COUNT) ctr.sph++;
        sid = m.scene_id; mat = m.material;
        return hit_sphere(r, mk(s.x, s.y, s.z), s.w, t);
    } else if (kind == 2u) {
        float4 a, b; V3 nn;
        sv_box(sv, idx, a, b);
        if (COUNT) ctr.aab++;
        sid = __float_as_uint(a.w); mat = __float_as_uint(b.w);
        return hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
    }
    const PlaneRec pl = P.planes[idx];
    const PrimMeta m = P.plane_meta[idx];
    if (COUNT) ctr.pln++;
    sid = m.scene_id; mat = m.material;
    return hit_plane(r, mk(pl.nx, pl.ny, pl.nz), pl.d, t);
}

struct GridWalk {
    int ix, iy, iz, ix_step, iy_step, iz_step, ix_stop, iy_stop, iz_stop;
    double tx_next, ty_next, tz_next, dtx, dty, dtz;
};
__device__ __forceinline__ int grid_index(float v, int n) {                // (int)clamp(v, 0, n - 1), RT/maths.h:50-53
    const double x = (double)v, hi = (double)(n - 1);
    return (int)(x < 0.0 ? 0.0 : (x > hi ? hi : x));
}
// Grid::Init_Traverse, RT/grid.cpp:101-245.  false = the ray misses the grid's bounding box.
__device__ __forceinline__ bool grid_init(const LaunchParams& P, const Ray& ray, GridWalk& g) {
    const float ox = ray.o.x, oy = ray.o.y, oz = ray.o.z, dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    const float x0 = P.grid_min[0], y0 = P.grid_min[1], z0 = P.grid_min[2];
    const float x1 = P.grid_max[0], y1 = P.grid_max[1], z1 = P.grid_max[2];
    const int nx = P.grid_n[0], ny = P.grid_n[1], nz = P.grid_n[2];
    float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max, t0, t1;
    const float a = fdiv(1.0f, dx);
    if (a >= 0.0f) { tx_min = (x0 - ox) * a; tx_max = (x1 - ox) * a; } else { tx_min = (x1 - ox) * a; tx_max = (x0 - ox) * a; }
    const float b = fdiv(1.0f, dy);
    if (b >= 0.0f) { ty_min = (y0 - oy) * b; ty_max = (y1 - oy) * b; } else { ty_min = (y1 - oy) * b; ty_max = (y0 - oy) * b; }
    const float c = fdiv(1.0f, dz);
    if (c >= 0.0f) { tz_min = (z0 - oz) * c; tz_max = (z1 - oz) * c; } else { tz_min = (z1 - oz) * c; tz_max = (z0 - oz) * c; }
    if (tx_min > ty_min) t0 = tx_min; else t0 = ty_min;
    if (tz_min > t0) t0 = tz_min;
    if (tx_max < ty_max) t1 = tx_max; else t1 = ty_max;
    if (tz_max < t1) t1 = tz_max;
    if (t0 > t1 || t1 < 0.0f) return false;
    const bool inside = (ox > x0 && ox < x1) && (oy > y0 && oy < y1) && (oz > z0 && oz < z1);   // AABB::isInside (strict)
    float px = ox, py = oy, pz = oz;
    if (!inside) { px = ox + dx * t0; py = oy + dy * t0; pz = oz + dz * t0; }
    g.ix = grid_index(fdiv((px - x0) * (float)nx, x1 - x0), nx);
    g.iy = grid_index(fdiv((py - y0) * (float)ny, y1 - y0), ny);
    g.iz = grid_index(fdiv((pz - z0) * (float)nz, z1 - z0), nz);
    g.dtx = (double)fdiv(tx_max - tx_min, (float)nx);
    g.dty = (double)fdiv(ty_max - ty_min, (float)ny);
    g.dtz = (double)fdiv(tz_max - tz_min, (float)nz);
    if (dx > 0.0f) { g.tx_next = (double)tx_min + (double)(g.ix + 1) * g.dtx; g.ix_step = 1; g.ix_stop = nx; }
    else           { g.tx_next = (double)tx_min + (double)(nx - g.ix) * g.dtx; g.ix_step = -1; g.ix_stop = -1; }
    if (dx == 0.0f) g.tx_next = (double)3.402823466e+38f;
    if (dy > 0.0f) { g.ty_next = (double)ty_min + (double)(g.iy + 1) * g.dty; g.iy_step = 1; g.iy_stop = ny; }
    else           { g.ty_next = (double)ty_min + (double)(ny - g.iy) * g.dty; g.iy_step = -1; g.iy_stop = -1; }
    if (dy == 0.0f) g.ty_next = (double)3.402823466e+38f;
    if (dz > 0.0f) { g.tz_next = (double)tz_min + (double)(g.iz + 1) * g.dtz; g.iz_step = 1; g.iz_stop = nz; }
    else           { g.tz_next = (double)tz_min + (double)(nz - g.iz) * g.dtz; g.iz_step = -1; g.iz_stop = -1; }
    if (dz == 0.0f) g.tz_next = (double)3.402823466e+38f;
    return true;
}
// one DDA step (RT/grid.cpp:282-308 without the acceptance test): returns the exit distance of the cell
// being left in t_exit, false when the walk leaves the grid
__device__ __forceinline__ bool grid_step(GridWalk& g) {
    if (g.tx_next < g.ty_next && g.tx_next < g.tz_next) { g.tx_next += g.dtx; g.ix += g.ix_step; return g.ix != g.ix_stop; }
    if (g.ty_next < g.tz_next) { g.ty_next += g.dty; g.iy += g.iy_step; return g.iy != g.iy_stop; }
    g.tz_next += g.dtz; g.iz += g.iz_step; return g.iz != g.iz_stop;
}
__device__ __forceinline__ double grid_cell_exit(const GridWalk& g) {
    if (g.tx_next < g.ty_next && g.tx_next < g.tz_next) return g.tx_next;
    if (g.ty_next < g.tz_next) return g.ty_next;
    return g.tz_next;
}
// Grid::Traverse(Ray&, Object**, Vector&), RT/grid.cpp:248-310
template <bool COUNT, class SV>
__device__ __forceinline__ Hit grid_closest(const LaunchParams& P, const SV& sv, const Ray& r, Ctr& ctr) {
    Hit best; best.t = 3.402823466e+38f; best.ref = 0xFFFFFFFFu; best.sid = 0xFFFFFFFFu; best.mat = 0;
    if (COUNT) ctr.closest++;
    GridWalk g;
    if (!grid_init(P, r, g)) return best;
    for (;;) {
        const uint32_t cell = (uint32_t)g.ix + (uint32_t)P.grid_n[0] * ((uint32_t)g.iy + (uint32_t)P.grid_n[1] * (uint32_t)g.iz);
        const uint32_t i0 = P.grid_cells[cell], i1 = P.grid_cells[cell + 1];
        Hit c = best; c.t = 3.402823466e+38f;                   // closestDistance restarts in every cell
        for (uint32_t i = i0; i < i1; i++) {                    // scene order inside a cell: "distance < closestDistance"
            const uint32_t ref = P.grid_items[i];
            float t; uint32_t sid, mat;
            if (prim_test<COUNT>(P, sv, r, ref, t, sid, mat, ctr) && t < c.t) { c.t = t; c.ref = ref; c.sid = sid; c.mat = mat; }
        }
        if ((double)c.t < grid_cell_exit(g)) return c;           // accepted only before the cell's exit
        if (!grid_step(g)) return best;
    }
}
// Grid::Traverse(Ray&), RT/grid.cpp:313-361: r.d is normalised, length = |L|.  A ray that misses the
// grid's box is reported as occluded, like in the reference.
template <bool COUNT, class SV>
__device__ __forceinline__ bool grid_any(const LaunchParams& P, const SV& sv, const Ray& r, float length, Ctr& ctr) {
    if (COUNT) ctr.shadow++;
    GridWalk g;
    if (!grid_init(P, r, g)) return true;
    for (;;) {
        const uint32_t cell = (uint32_t)g.ix + (uint32_t)P.grid_n[0] * ((uint32_t)g.iy + (uint32_t)P.grid_n[1] * (uint32_t)g.iz);
        const uint32_t i0 = P.grid_cells[cell], i1 = P.grid_cells[cell + 1];
        for (uint32_t i = i0; i < i1; i++) {
            float t; uint32_t sid, mat;
            if (prim_test<COUNT>(P, sv, r, P.grid_items[i], t, sid, mat, ctr) && t < length) return true;
        }
        if (!grid_step(g)) return false;
    }
}

// ------------------------------------------------------------------ wave-wide (packet) traversal
// For small trees the 64 rays of a wave walk the BVH TOGETHER: the node index is wave-uniform,
// so node and primitive records are fetched once per wave (broadcast LDS reads or scalar
// loads), the traversal stack is one LDS dword per entry for the whole wave, primitive-kind
// dispatch is a scalar branch, and lanes only differ in their exec bit.  A node is visited when
// ANY lane's slab test passes (ballot).  Incoherent rays make the wave visit the union of the
// lanes' paths, which for a tree of a few nodes is still far cheaper than 64 private walks
// with divergent loops; large scenes use the per-lane walk above.  Results are identical:
// the hit rule (nearest t, lowest scene index on ties) does not depend on visit order.
// Both functions must be called by all lanes of the wave together (converged), with
// `active` = this lane has a ray.
struct WaveStack {
    int32_t* base;       // this WAVE's region (same address in every lane)
    __device__ __forceinline__ void push(int sp, int32_t node) { base[sp] = node; }
    __device__ __forceinline__ int32_t at(int sp) const { return __builtin_amdgcn_readfirstlane(base[sp]); }
};

template <bool COUNT, class SV>
__device__ __forceinline__ void leaf_closest_packet(const LaunchParams& P, const SV& sv, const Ray& r, bool live,
                                                    int32_t leaf, Hit& best, Ctr& ctr) {
    // the leaf is wave-uniform: its record and its primitives are fetched once for the wave (same address in
    // every lane: a broadcast LDS read or one cache line), the counts are scalar loop bounds
    const uint4 Lv = sv_leaf(sv, leaf);
    const uint32_t tri0 = __builtin_amdgcn_readfirstlane(Lv.x), sph0 = __builtin_amdgcn_readfirstlane(Lv.y),
                   box0 = __builtin_amdgcn_readfirstlane(Lv.z), cnt = __builtin_amdgcn_readfirstlane(Lv.w);
    const uint32_t nt = cnt & 0xFFu, ns = (cnt >> 8) & 0xFFu, nb = cnt >> 16;
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < nt; i++) {
        float4 a, b, c;
        sv_tri(sv, tri0 + i, a, b, c);
        if (live) {
            if (COUNT) ctr.tri++;
            float t;
            const bool h = hit_triangle<true>(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), t);
            take_closer(best, h, t, (1u << kRefKindShift) | (tri0 + i), __float_as_uint(a.w), __float_as_uint(b.w));
        }
    }
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < ns; i++) {
        const float4 sp = sv_sphere(sv, sph0 + i);
        const PrimMeta m = sv_sphere_meta(sv, sph0 + i);
        if (live) {
            if (COUNT) ctr.sph++;
            float t;
            const bool h = hit_sphere(r, mk(sp.x, sp.y, sp.z), sp.w, t);
            take_closer(best, h, t, (0u << kRefKindShift) | (sph0 + i), m.scene_id, m.material);
        }
    }
    _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < nb; i++) {
        float4 a, b;
        sv_box(sv, box0 + i, a, b);
        if (live) {
            if (COUNT) ctr.aab++;
            float t; V3 nn;
            const bool h = hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
            take_closer(best, h, t, (2u << kRefKindShift) | (box0 + i), __float_as_uint(a.w), __float_as_uint(b.w));
        }
    }
}

template <bool COUNT, class SV>
__device__ __forceinline__ Hit closest_hit_packet(const LaunchParams& P, const SV& sv, const Ray& r, bool active,
                                                  WaveStack ws, Ctr& ctr) {
    Hit best; best.t = 3.402823466e+38f; best.ref = 0xFFFFFFFFu; best.sid = 0xFFFFFFFFu; best.mat = 0;
    if (COUNT && active) ctr.closest++;
    for (uint32_t i = 0; i < P.n_planes; i++) {
        PlaneRec pl = P.planes[i];
        PrimMeta m = P.plane_meta[i];
        float t;
        if (active) {
            if (COUNT) ctr.pln++;
            if (hit_plane(r, mk(pl.nx, pl.ny, pl.nz), pl.d, t) &&
                (t < best.t || (t == best.t && m.scene_id < best.sid))) {
                best.t = t; best.ref = (3u << kRefKindShift) | i; best.sid = m.scene_id; best.mat = m.material;
            }
        }
    }
    if (__ballot(active) == 0) return best;
    SlabRay s = make_slab(sv, r);
    int sp = 0;
    int32_t cur = 0;
    for (;;) {
        float tn0 = 0.0f, tn1 = 0.0f; bool t0, t1; int32_t cv0, cv1;
        node_test(sv, s, cur, best.t, t0, t1, tn0, tn1, cv0, cv1);
        const bool h0 = active && t0, h1 = active && t1;
        if (COUNT && active) ctr.box += 2;
        uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
        const int32_t c0 = __builtin_amdgcn_readfirstlane(cv0), c1 = __builtin_amdgcn_readfirstlane(cv1);
        if (m0 && c0 < 0) { leaf_closest_packet<COUNT>(P, sv, r, h0, c0, best, ctr); m0 = 0; }
        if (m1 && c1 < 0) { leaf_closest_packet<COUNT>(P, sv, r, h1, c1, best, ctr); m1 = 0; }
        if (m0 && m1) {
            // near child first as seen by the first lane that wants both (any order is correct)
            const uint64_t both = m0 & m1;
            const int f = both ? (int)__builtin_ctzll(both) : (int)__builtin_ctzll(m0);
            const bool swap = __shfl(tn1, f) < __shfl(tn0, f);
            ws.push(sp, swap ? c0 : c1); sp++;
            cur = swap ? c1 : c0;
        } else if (m0) cur = c0;
        else if (m1) cur = c1;
        else {
            if (sp == 0) break;
            sp--; cur = ws.at(sp);
        }
    }
    return best;
}

// per-lane result: occluded (any primitive hit, with t < tmax when bounded)
template <bool COUNT, class SV>
__device__ __forceinline__ bool any_hit_packet(const LaunchParams& P, const SV& sv, const Ray& r, bool active,
                                               bool bounded, float tmax, WaveStack ws, Ctr& ctr) {
    bool occ = false;
    if (COUNT && active) ctr.shadow++;
    if (P.n_planes) {
        const bool gate = active && (!bounded || ref_unit_box_hit(r));
        for (uint32_t i = 0; i < P.n_planes; i++) {
            PlaneRec pl = P.planes[i];
            float t;
            if (gate && !occ) {
                if (COUNT) ctr.pln++;
                if (hit_plane(r, mk(pl.nx, pl.ny, pl.nz), pl.d, t) && (!bounded || t < tmax)) occ = true;
            }
        }
    }
    if (__ballot(active && !occ) == 0) return occ;
    SlabRay s = make_slab(sv, r);
    const float tlimit = bounded ? tmax : 3.402823466e+38f;
    int sp = 0;
    int32_t cur = 0;
    for (;;) {
        float tn0, tn1; bool t0, t1; int32_t cv0, cv1;
        const bool live = active && !occ;
        node_test(sv, s, cur, tlimit, t0, t1, tn0, tn1, cv0, cv1);
        const bool h0 = live && t0, h1 = live && t1;
        if (COUNT && live) ctr.box += 2;
        uint64_t m0 = __ballot(h0), m1 = __ballot(h1);
        const int32_t c0 = __builtin_amdgcn_readfirstlane(cv0), c1 = __builtin_amdgcn_readfirstlane(cv1);
        for (int side = 0; side < 2; side++) {
            const int32_t c = side ? c1 : c0;
            const uint64_t m = side ? m1 : m0;
            if (!(m && c < 0)) continue;
            const bool hl = (side ? h1 : h0) && !occ;
            const uint4 Lv = sv_leaf(sv, c);
            const uint32_t tri0 = __builtin_amdgcn_readfirstlane(Lv.x), sph0 = __builtin_amdgcn_readfirstlane(Lv.y),
                           box0 = __builtin_amdgcn_readfirstlane(Lv.z), cnt = __builtin_amdgcn_readfirstlane(Lv.w);
            const uint32_t nt = cnt & 0xFFu, ns = (cnt >> 8) & 0xFFu, nb = cnt >> 16;
            _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < nt; i++) {
                float4 a, b, cc;
                sv_tri(sv, tri0 + i, a, b, cc);
                if (hl && !occ) {
                    if (COUNT) ctr.tri++;
                    float t;
                    const bool h = hit_triangle<true>(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(cc.x, cc.y, cc.z), t);
                    occ = occ | (h && (!bounded || t < tmax));
                }
            }
            _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < ns; i++) {
                const float4 sp4 = sv_sphere(sv, sph0 + i);
                if (hl && !occ) {
                    if (COUNT) ctr.sph++;
                    float t;
                    const bool h = hit_sphere(r, mk(sp4.x, sp4.y, sp4.z), sp4.w, t);
                    occ = occ | (h && (!bounded || t < tmax));
                }
            }
            _Pragma("clang loop vectorize(disable) unroll(disable)")
    for (uint32_t i = 0; i < nb; i++) {
                float4 a, b;
                sv_box(sv, box0 + i, a, b);
                if (hl && !occ) {
                    if (COUNT) ctr.aab++;
                    float t; V3 nn;
                    const bool h = hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
                    occ = occ | (h && (!bounded || t < tmax));
                }
            }
        }
        if (m0 && c0 < 0) m0 = 0;
        if (m1 && c1 < 0) m1 = 0;
        if (__ballot(active && !occ) == 0) break;
        if (m0 && m1) { ws.push(sp, c1); sp++; cur = c0; }
        else if (m0) cur = c0;
        else if (m1) cur = c1;
        else {
            if (sp == 0) break;
            sp--; cur = ws.at(sp);
        }
    }
    return occ;
}

}  // namespace p3d
#endif
