// p3d_shade.h -- one node of the reference's shading recursion (rayTracing(),
// RT/main.cpp:530-721, and processLight(), RT/main.cpp:471-526) as a non-recursive function:
// shade_hit() returns the node's direct colour, its mixing weight KR and the child rays it
// spawns; the caller (wavefront level kernels or the single-launch tree kernel) decides how the
// children are scheduled and combines them with combine_node() in the reference's order.
#ifndef P3D_SHADE_H
#define P3D_SHADE_H

#include "p3d_traverse.h"
#include "p3d_powf.h"

namespace p3d {

// getNormal(point).normalize() of the hit primitive (RT/main.cpp:587-589)
template <class SV>
__device__ __forceinline__ V3 prim_normal(const LaunchParams& P, const SV& sv, uint32_t ref, const Ray& r, V3 point) {
    uint32_t kind = ref >> kRefKindShift, idx = ref & kRefIndexMask;
    if (kind == 0u) {                                                   // RT/scene.cpp:174-178
        float4 s = sv_sphere(sv, idx);
        V3 n = normalized(sub(point, mk(s.x, s.y, s.z)));
        return normalized(n);
    } else if (kind == 1u) {                                            // RT/scene.cpp:10-25,46-49
        const float4 n = sv_tri_normal(sv, idx);                        // normalised twice on the host (scene_flatten.cpp)
        return mk(n.x, n.y, n.z);
    } else if (kind == 2u) {
        float4 a, b;
        sv_box(sv, idx, a, b);
        float t; V3 nn = mk(0.0f, 0.0f, 0.0f);
        hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
        return normalized(nn);
    } else {                                                            // RT/scene.cpp:143-146
        PlaneRec pl = P.planes[idx];
        return normalized(mk(pl.nx, pl.ny, pl.nz));
    }
}

struct Mtl { V3 diff; float kd; V3 spec; float ks; float shine, T, ior, refl; };
template <class SV>
__device__ __forceinline__ Mtl load_material(const SV& sv, uint32_t m) {
    float4 a, b, c;
    sv_mat(sv, m, a, b, c);
    Mtl r; r.diff = mk(a.x, a.y, a.z); r.kd = a.w; r.spec = mk(b.x, b.y, b.z); r.ks = b.w;
    r.shine = c.x; r.T = c.y; r.ior = c.z; r.refl = c.w;
    return r;
}


// Traversal context of a lane: its private stack (per-lane walk) and its wave's shared stack
// (packet walk).  WALK selects the walk at compile time: 0 = per-lane BVH walk, 1 = wave-wide (packet) BVH
// walk -- identical results -- and 2 = the reference's uniform grid (GRID mode, accel 1: its own semantics).
struct TravCtx { TravStack lane; WaveStack wave; uint32_t* share; };
// WALK_SHARED (scenes read from HBM only): the per-lane walk with the lanes of a wave sharing their pending subtrees
// (p3d_traverse.h: closest_hit_shared).  Like the packet walk it must be reached by all lanes of the wave together.
enum { WALK_LANE = 0, WALK_PACKET = 1, WALK_GRID = 2, WALK_SHARED = 3 };

template <bool COUNT> __device__ __forceinline__ Hit closest_shared(const LaunchParams& P, const GlobalScene& sv, const Ray& ray, bool active,
                                                                    const TravCtx& tc, Ctr& ctr) {
    return closest_hit_shared<COUNT>(P, sv, ray, active, tc.lane, tc.share, ctr);
}
template <bool COUNT> __device__ __forceinline__ Hit closest_shared(const LaunchParams& P, const LdsScene& sv, const Ray& ray, bool active,
                                                                    const TravCtx& tc, Ctr& ctr) {      // (never selected: LDS scenes keep the private walk)
    Hit h; h.t = 3.402823466e+38f; h.ref = 0xFFFFFFFFu; h.sid = 0xFFFFFFFFu; h.mat = 0;
    if (active) h = closest_hit<COUNT>(P, sv, ray, tc.lane, ctr);
    return h;
}
template <bool COUNT> __device__ __forceinline__ bool any_shared(const LaunchParams& P, const GlobalScene& sv, const Ray& sr, bool need, bool bounded,
                                                                 float length, const TravCtx& tc, Ctr& ctr) {
    return any_hit_shared<COUNT>(P, sv, sr, need, bounded, length, tc.lane, tc.share, ctr);
}
template <bool COUNT> __device__ __forceinline__ bool any_shared(const LaunchParams& P, const LdsScene& sv, const Ray& sr, bool need, bool bounded,
                                                                 float length, const TravCtx& tc, Ctr& ctr) {
    return need ? any_hit<COUNT>(P, sv, sr, bounded, length, tc.lane, ctr) : false;
}

template <bool COUNT, int WALK, class SV>
__device__ __forceinline__ Hit find_closest(const LaunchParams& P, const SV& sv, const Ray& ray, bool active,
                                            const TravCtx& tc, Ctr& ctr) {
    if (WALK == WALK_PACKET) return closest_hit_packet<COUNT>(P, sv, ray, active, tc.wave, ctr);
    if (WALK == WALK_SHARED) return closest_shared<COUNT>(P, sv, ray, active, tc, ctr);
    Hit h; h.t = 3.402823466e+38f; h.ref = 0xFFFFFFFFu; h.sid = 0xFFFFFFFFu; h.mat = 0;
    if (active) h = (WALK == WALK_GRID) ? grid_closest<COUNT>(P, sv, ray, ctr) : closest_hit<COUNT>(P, sv, ray, tc.lane, ctr);
    return h;
}

// Shadow query of processLight() (RT/main.cpp:476-510).  `need` = this lane builds a shadow
// ray (L.N > 0).  NONE: un-normalised direction, no distance bound; BVH/GRID: normalised
// direction and t < |L| (BVH::Traverse(Ray&), RT/bvh.cpp:351-352).
template <bool COUNT, int WALK, class SV>
__device__ __forceinline__ bool light_occluded(const LaunchParams& P, const SV& sv, V3 L, V3 precise, bool need,
                                               const TravCtx& tc, Ctr& ctr) {
    Ray sr; sr.o = precise; sr.d = L;
    float length = 0.0f;
    const bool bounded = WALK == WALK_GRID || P.accel != 0;
    if (bounded && need) { length = vlen(sr.d); sr.d = normalized(sr.d); }
    if (WALK == WALK_PACKET) return any_hit_packet<COUNT>(P, sv, sr, need, bounded, length, tc.wave, ctr);
    if (WALK == WALK_SHARED) return any_shared<COUNT>(P, sv, sr, need, bounded, length, tc, ctr);
    if (WALK == WALK_GRID) return need ? grid_any<COUNT>(P, sv, sr, length, ctr) : false;   // Grid::Traverse(Ray&), RT/grid.cpp:313
    return need ? any_hit<COUNT>(P, sv, sr, bounded, length, tc.lane, ctr) : false;
}
// where a kernel reads powf's tables and coefficients (the head of the scene blob): the LDS copy of the scene, or the blob
__device__ __forceinline__ PowTabLds pow_tab(const LdsScene&) { return PowTabLds(); }
__device__ __forceinline__ PowTabGlobal pow_tab(const GlobalScene& g) { return PowTabGlobal(g.q); }

// Blinn-Phong term of one unoccluded light, RT/main.cpp:512-525
template <class SV>
__device__ __forceinline__ void light_term(const SV& sv, V3 L, V3 lcol, V3& color, const Mtl& M, const Ray& ray, V3 normal) {
    L = normalized(L);        // (the same value light_occluded() formed for a bounded shadow ray: CSE'd when inlined)
    V3 H = normalized(add(L, mul(ray.d, -1.0f)));
    float VdotN = dot(H, normal);
    float d1 = dot(normal, L);
    float max1 = (0.0f < d1) ? d1 : 0.0f;                // std::max(0.0f, x)
    float max2 = (0.0f < VdotN) ? VdotN : 0.0f;
    V3 diff = mul(cmul(lcol, M.diff), max1);
    if (M.ks == 0.0f && M.shine >= 0.0f) {
        // spec * Ks * 0.4 is (finite * 0) * 0.4 = 0: the sum is unchanged (up to the sign of a zero),
        // and powf is the most expensive call of the whole term
        color = add(color, mul(diff, M.kd));
        return;
    }
    // the host libm's powf, bit for bit (p3d_powf.h); max2 is +0 or positive
    V3 spec = mul(cmul(lcol, M.spec), p3d_powf_nonneg(max2, M.shine, pow_tab(sv)));
    color = add(color, add(mul(diff, M.kd), mul(mul(spec, M.ks), 0.4f)));
}

// What one rayTracing() invocation produces before its recursive calls return.
struct NodeOut {
    bool terminal;          // true: `ret` is this invocation's return value already
    V3 ret;                 // valid when terminal
    V3 color;               // direct lighting sum (the reference's `color` before recursion)
    float KR;
    uint32_t mat;
    bool has_refl, has_refr;
    Ray refl, refr;         // child rays; the reflection child keeps ior_1, the refraction child gets newIor
    float newIor;
    uint32_t rng_refl, rng_refr;   // the children's random-stream keys (stochastic features only)
};

// ---- random numbers of the distribution-ray-tracing features
// The reference draws from libc rand() inside the recursion, in pixel order: a serial stream no
// parallel renderer can reproduce.  Here every rayTracing() invocation owns a 32-bit key -- the root's
// is hash(seed, pixel, sample), a child's is hash(parent key, which child) -- and its k-th draw is
// hash(key, k): the image is a pure function of (scene, camera, seed), independent of schedule,
// sharding and launch order.  Parity with the reference is statistical (tests/test_gpu_distribution.py).
__device__ __forceinline__ uint32_t rng_mix(uint32_t a, uint32_t b) {
    uint32_t h = (a ^ 0x9E3779B9u) * 0x85EBCA6Bu + b;
    h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ float rng_u01(uint32_t key, uint32_t k) {       // [0, 1), 24 bits like rand_float()
    return (float)(rng_mix(key, k) >> 8) * (1.0f / 16777216.0f);
}
constexpr uint32_t kRngRefl = 0xA511E9B3u, kRngRefr = 0x63D83595u, kRngFuzzy = 0x40000000u;

// position of light `light` as this invocation sees it: jittered over the reference's 0.5 x 0.5
// area light, one stratum per pixel sample, when soft shadows run with anti-aliasing
// (RT/main.cpp:621); the light's own position otherwise
template <bool STOCH>
__device__ __forceinline__ V3 light_position(const LaunchParams& P, float4 lpos, uint32_t light, uint32_t rng, int sample) {
    if (STOCH && (P.features & kFeatSoftJitter)) {
        const float offx = (float)(sample / P.spp), offy = (float)(sample % P.spp);   // RT/main.cpp:779-780
        const float jx = fdiv(offx + rng_u01(rng, 2u * light), (float)P.spp);
        const float jy = fdiv(offy + rng_u01(rng, 2u * light + 1u), (float)P.spp);
        return mk(lpos.x + 0.5f * jx, lpos.y + 0.5f * jy, lpos.z);
    }
    return mk(lpos.x, lpos.y, lpos.z);
}

// Scene::GetSkyboxColor, RT/scene.cpp:383-461, same expressions in the same types: "double invMa = 1 / ma" is a float
// division widened afterwards, s and t are formed in double and rounded once, the two clamping lines (:450,452) are
// expression statements without effect, u8tofloat divides by 255.99f (RT/maths.h:120-123).  Bit-exact against the
// reference's object code through the oracle (tests/test_oracle_vs_ref.py, tests/test_gpu_distribution.py).  A direction
// with NaNs indexes outside the image in the reference (wild read); here it returns black.
__device__ __forceinline__ V3 skybox_color(const LaunchParams& P, V3 c) {
    float ma; int side;
    if (fabsf(c.x) > fabsf(c.y)) { ma = fabsf(c.x); side = c.x >= 0.0f ? 1 : 0; }     // LEFT at X = +1, RIGHT at X = -1
    else { ma = fabsf(c.y); side = c.y >= 0.0f ? 2 : 3; }                            // TOP / BOTTOM
    if (fabsf(c.z) > ma) { ma = fabsf(c.z); side = c.z >= 0.0f ? 4 : 5; }            // FRONT / BACK
    const float sc = side == 0 ? -c.z : side == 1 ? c.z : side == 5 ? c.x : -c.x;
    const float tc = side == 2 ? -c.z : side == 3 ? c.z : c.y;
    const double invMa = (double)fdiv(1.0f, ma);
    const float s = (float)(((double)sc * invMa + 1.0) / 2.0);
    const float t = (float)(((double)tc * invMa + 1.0) / 2.0);
    const uint32_t width = P.sky_w[side], height = P.sky_h[side], bpp = P.sky_bpp[side];
    const uint32_t xp = (uint32_t)(int)((float)(width - 1u) * s);
    const uint32_t yp = (uint32_t)(int)((float)(height - 1u) * t);
    if (!(xp < width && yp < height)) return mk(0.0f, 0.0f, 0.0f);
    const uint8_t* px = P.sky + P.sky_off[side] + ((size_t)yp * width + xp) * bpp;
    return mk(fdiv((float)px[0], 255.99f), fdiv((float)px[1], 255.99f), fdiv((float)px[2], 255.99f));
}
// what a ray that hits nothing returns: bgColor (RT/main.cpp:582), or the cube map with P3D_FEATURE_SKYBOX
template <bool STOCH>
__device__ __forceinline__ V3 miss_color(const LaunchParams& P, const Ray& ray) {
    if (STOCH && (P.features & kFeatSky)) return skybox_color(P, ray.d);
    return mk(P.bg[0], P.bg[1], P.bg[2]);
}

// colour returned by a node once its children returned refl_ret / refr_ret (zero when the
// child was never traced): "color += reflection_color * KR * specColor + refraction_color *
// (1 - KR)", RT/main.cpp:719, same association.
__device__ __forceinline__ V3 combine_node(V3 color, float KR, V3 spec, V3 refl_ret, V3 refr_ret) {
    return add(color, add(cmul(mul(refl_ret, KR), spec), mul(refr_ret, 1.0f - KR)));
}

// rayTracing(ray, depth, ior_1) up to (not including) its recursive calls.  Written so that
// the shadow queries sit in wave-uniform control flow (the packet walk needs every lane of the
// wave to arrive together): lanes without a ray or without a hit carry live == false /
// hit == false through the light loop instead of leaving early.
template <bool COUNT, int WALK, class SV, bool STOCH = false>
__device__ __forceinline__ NodeOut shade_hit(const LaunchParams& P, const SV& sv, const Ray& ray, const Hit& h,
                                             bool live, int depth, float ior_1, const TravCtx& tc, Ctr& ctr,
                                             uint32_t rng = 0u, int sample_override = -1) {
    // pixel sample this invocation belongs to: a launch parameter in the wavefront schedule (one launch per
    // sample pass), a loop variable in the tile schedule
    const int sample = sample_override >= 0 ? sample_override : P.wf_sample;
    NodeOut o;
    o.terminal = true; o.KR = 0.0f; o.mat = h.mat; o.has_refl = false; o.has_refr = false; o.newIor = 1.0f;
    o.rng_refl = STOCH ? rng_mix(rng, kRngRefl) : 0u; o.rng_refr = STOCH ? rng_mix(rng, kRngRefr) : 0u;
    o.color = mk(0.0f, 0.0f, 0.0f);
    o.ret = o.color;
    o.refl.o = o.color; o.refl.d = o.color; o.refr.o = o.color; o.refr.d = o.color;
#ifdef P3D_DEBUG_SKIP
    if (P.dbg_skip == 1u) { o.ret = mk(h.t, 0.0f, 0.0f); return o; }
#endif
    const bool hit = live && h.ref != 0xFFFFFFFFu;
    if (__ballot(hit) == 0) {                     // whole wave missed (sky tiles): nothing to light
        o.ret = miss_color<STOCH>(P, ray);
        return o;
    }
    V3 hit_point = o.color, normal = o.color, precise = o.color;
    if (hit) {                                                           // RT/main.cpp:587-590
        hit_point = add(ray.o, mul(ray.d, h.t));
        normal = prim_normal(P, sv, h.ref, ray, hit_point);
        precise = add(hit_point, mul(normal, P3D_EPS));
        // getNormal(precise_hit_point): only a sphere's normal depends on the point
        if ((h.ref >> kRefKindShift) == 0u) normal = prim_normal(P, sv, h.ref, ray, precise);
    }
    V3 color = mk(0.0f, 0.0f, 0.0f);
    // lights in groups of 64: first every shadow query of the group (bit i = light i occluded),
    // then the shading terms in light order -- same sums, same order as the reference's loop
    for (uint32_t l0 = 0; l0 < P.n_lights; l0 += 64) {
        const uint32_t ln = (P.n_lights - l0 < 64u) ? (P.n_lights - l0) : 64u;
        uint64_t occluded = 0;
        for (uint32_t i = 0; i < ln; i++) {
            const float4 lpos = reinterpret_cast<const float4*>(P.lights + l0 + i)[0];
            V3 L = sub(light_position<STOCH>(P, lpos, l0 + i, rng, sample), hit_point);
#ifdef P3D_DEBUG_SKIP
            const bool need = hit && dot(L, normal) > 0.0f && P.dbg_skip != 2u;
#else
            const bool need = hit && dot(L, normal) > 0.0f;              // RT/main.cpp:476
#endif
            if (light_occluded<COUNT, WALK>(P, sv, L, precise, need, tc, ctr)) occluded |= (1ull << i);
        }
        if (hit) {
            Mtl Ml = load_material(sv, h.mat);
            for (uint32_t i = 0; i < ln; i++) {
                if (occluded & (1ull << i)) continue;
                const float4* lp = reinterpret_cast<const float4*>(P.lights + l0 + i);
                float4 lpos = lp[0], lcol = lp[1];
                V3 L = sub(light_position<STOCH>(P, lpos, l0 + i, rng, sample), hit_point);
                light_term(sv, L, mk(lcol.x, lcol.y, lcol.z), color, Ml, ray, normal);
            }
        }
    }
    if (!hit) {
        o.ret = miss_color<STOCH>(P, ray);
        return o;
    }
    Mtl M = load_material(sv, h.mat);
    V3 Vv = mul(ray.d, -1.0f);
    o.color = color;
    if (depth >= P.max_depth) {                                          // RT/main.cpp:632-634
        o.ret = clampc(color);
        return o;
    }
    bool inside = false;
    if (dot(ray.d, normal) > 0.0f) { normal = mul(normal, -1.0f); inside = true; }
    if (M.refl > 0.0f) {                                                 // RT/main.cpp:646-667
        V3 rdir = sub(ray.d, mul(mul(normal, dot(ray.d, normal)), 2.0f));
        o.refl.o = precise;
        if (STOCH && (P.features & kFeatFuzzy)) {                        // RT/main.cpp:651-660
            // rnd_unit_sphere(), RT/maths.h:98-104: rejection sampling of the unit ball
            V3 p; uint32_t k = kRngFuzzy;
            do {
                p = sub(mul(mk(rng_u01(rng, k), rng_u01(rng, k + 1u), rng_u01(rng, k + 2u)), 2.0f), mk(1.0f, 1.0f, 1.0f));
                k += 3u;
            } while (dot(p, p) >= 1.0f);
            V3 sphere_center = add(rdir, precise);
            V3 sphere_offset = add(sphere_center, mul(p, 0.3f));          // roughness 0.3
            V3 fuzzy = normalized(sub(sphere_offset, precise));
            // a rejected fuzzy direction leaves the mirror direction UN-normalised, as in the reference
            o.refl.d = dot(fuzzy, normal) > 0.0f ? fuzzy : rdir;
        } else {
            o.refl.d = normalized(rdir);
        }
        o.has_refl = true;
    }
    float KR;
    if (M.T != 0.0f) {                                                   // RT/main.cpp:671-713
        float R0 = 1.0f, R1 = 1.0f;
        V3 viewnormal = mul(normal, dot(normal, Vv));
        V3 viewtangent = sub(viewnormal, Vv);
        float nn = inside ? ior_1 : fdiv(ior_1, M.ior);
        float cos_i = vlen(viewnormal);
        float sin_t = nn * vlen(viewtangent);
        float insqrt = (float)(1.0 - (double)sin_t * (double)sin_t);     // pow(float, 2) is double
        if (insqrt >= 0.0f) {
            float cos_t = fsqrt(insqrt);
            V3 rfr = add(mul(normalized(viewtangent), sin_t), normalized(mul(normal, cos_t)));
            o.refr.o = add(hit_point, mul(rfr, 0.001f));
            o.refr.d = rfr;
            o.newIor = inside ? 1.0f : M.ior;
            o.has_refr = true;
            float den = ior_1 * cos_i + o.newIor * cos_t;
            float q0 = fabsf(fdiv(ior_1 * cos_i - o.newIor * cos_t, den));
            float q1 = fabsf(fdiv(ior_1 * cos_t - o.newIor * cos_i, den));
            R0 = (float)((double)q0 * (double)q0);
            R1 = (float)((double)q1 * (double)q1);
        }
        KR = 0.0f * (R0 + R1);                                           // 1 / 2 * (R0 + R1), SURVEY Q5
    } else {
        KR = M.ks;
    }
    o.KR = KR;
    if (o.has_refl || o.has_refr) {
        o.terminal = false;
    } else {
        V3 zero = mk(0.0f, 0.0f, 0.0f);
        o.ret = combine_node(color, KR, M.spec, zero, zero);
    }
    return o;
}

// Camera::PrimaryRay, RT/camera.h:91-108
// the same ray from the frame constants: fx = px/res_x - 0.5f, fy = py/res_y - 0.5f come from
// the per-column / per-row tables, uw = u*w, vh = v*h, vz = n*(-plane_dist) from the host
__device__ __forceinline__ Ray primary_ray_tab(const LaunchParams& P, int x, int y) {
    const float fx = P.ray_fx[x], fy = P.ray_fy[y];
    V3 vX = mul(mk(P.uw[0], P.uw[1], P.uw[2]), fx);
    V3 vY = mul(mk(P.vh[0], P.vh[1], P.vh[2]), fy);
    Ray r; r.o = mk(P.eye[0], P.eye[1], P.eye[2]);
    r.d = normalized(add(add(vX, vY), mk(P.vz[0], P.vz[1], P.vz[2])));
    return r;
}
__device__ __forceinline__ Ray primary_ray(const LaunchParams& P, float px, float py) {
    V3 u = mk(P.u[0], P.u[1], P.u[2]), v = mk(P.v[0], P.v[1], P.v[2]), n = mk(P.n[0], P.n[1], P.n[2]);
    V3 vX = mul(mul(u, P.w), fdiv(px, (float)P.res_x) - 0.5f);
    V3 vY = mul(mul(v, P.h), fdiv(py, (float)P.res_y) - 0.5f);
    V3 vZ = mul(n, -P.plane_dist);
    Ray r; r.o = mk(P.eye[0], P.eye[1], P.eye[2]);
    r.d = normalized(add(add(vX, vY), vZ));
    return r;
}
// Camera::PrimaryRay(lens, pixel), RT/camera.h:110-127
__device__ __forceinline__ Ray primary_ray_lens(const LaunchParams& P, float lx, float ly, float px, float py) {
    V3 u = mk(P.u[0], P.u[1], P.u[2]), v = mk(P.v[0], P.v[1], P.v[2]), n = mk(P.n[0], P.n[1], P.n[2]);
    float ppx = P.w * (fdiv(px, (float)P.res_x) - 0.5f) * P.focal_ratio;
    float ppy = P.h * (fdiv(py, (float)P.res_y) - 0.5f) * P.focal_ratio;
    V3 dir = add(add(mul(u, ppx - lx), mul(v, ppy - ly)), mul(n, -P.focal_ratio * P.plane_dist));
    Ray r;
    r.d = normalized(dir);
    r.o = add(add(mk(P.eye[0], P.eye[1], P.eye[2]), mul(u, lx)), mul(v, ly));
    return r;
}

}  // namespace p3d
#endif
