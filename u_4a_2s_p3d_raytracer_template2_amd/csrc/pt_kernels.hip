// pt_kernels.hip -- the reference's Shadertoy path tracer (PT/P3D_RT.glsl + PT/common.glsl,
// PT/ = /root/reference/GPU_PathTracer_template/) as one gfx950 kernel: a lane is a pixel and
// runs ALL requested frames of mainImage() in registers (the shader's only inter-frame state is
// the pixel's own previous value in buffer A, so nothing has to go through memory between
// frames).  The 100 procedural small spheres are ray-independent except for the motion-blur
// offset, so each workgroup derives their table once into LDS (same integer hash, same float
// expressions as the shader evaluates per ray) and every hit_world() walks that table with
// broadcast LDS reads; the pixel RNG is still advanced once per moving-sphere candidate per call,
// exactly like the shader.  GLSL built-ins at their specification formulas; compiled with
// -ffp-contract=off.  PARITY UNPINNED (see include/p3d_pathtracer.h).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "p3d_pathtracer.h"

namespace p3dpt {

struct f2 { float x, y; };
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 F3(float a, float b, float c) { f3 r; r.x = a; r.y = b; r.z = c; return r; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return F3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 operator/(f3 a, float s) { return F3(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross3(f3 a, f3 b) { return F3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
__device__ __forceinline__ float len3(f3 a) { return __builtin_sqrtf(dot3(a, a)); }
__device__ __forceinline__ f3 norm3(f3 a) { return a / len3(a); }
__device__ __forceinline__ f3 mix3(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
__device__ __forceinline__ f3 pow3(f3 a, float e) { return F3(powf(a.x, e), powf(a.y, e), powf(a.z, e)); }

#define PT_PI 3.14159265358979f      // PT/common.glsl:1
#define PT_EPS 0.001f                // PT/common.glsl:2

// ---- integer hash RNG, PT/common.glsl:29-54 (exact in uint32)
__device__ __forceinline__ uint32_t base_hash(uint32_t px, uint32_t py) {
    uint32_t qx = 1103515245U * ((px >> 1U) ^ py);
    uint32_t qy = 1103515245U * ((py >> 1U) ^ px);
    uint32_t h = 1103515245U * (qx ^ (qy >> 3U));
    return h ^ (h >> 16);
}
__device__ __forceinline__ uint32_t next_hash(float& seed) {     // vec2(seed += 0.1, seed += 0.1): left to right
    seed += 0.1f; const float a = seed;
    seed += 0.1f; const float b = seed;
    return base_hash(__float_as_uint(a), __float_as_uint(b));
}
__device__ __forceinline__ float hash1(float& seed) { return (float)next_hash(seed) / (float)0xffffffffU; }
__device__ __forceinline__ f2 hash2(float& seed) {
    const uint32_t n = next_hash(seed);
    f2 r; r.x = (float)(n & 0x7fffffffU) / (float)0x7fffffff; r.y = (float)((n * 48271U) & 0x7fffffffU) / (float)0x7fffffff;
    return r;
}
__device__ __forceinline__ f3 hash3(float& seed) {
    const uint32_t n = next_hash(seed);
    return F3((float)(n & 0x7fffffffU) / (float)0x7fffffff, (float)((n * 16807U) & 0x7fffffffU) / (float)0x7fffffff,
              (float)((n * 48271U) & 0x7fffffffU) / (float)0x7fffffff);
}
__device__ __forceinline__ f2 random_in_unit_disk(float& seed) {                  // :71-76
    const f2 h = hash2(seed);
    const float phi = h.y * 6.28318530718f;
    const float r = __builtin_sqrtf(h.x * 1.0f);
    float sn, cs;
    sincosf(phi, &sn, &cs);                      // one argument reduction for both (the same values as sinf / cosf)
    f2 o; o.x = r * sn; o.y = r * cs;
    return o;
}
__device__ __forceinline__ f3 random_in_unit_sphere(float& seed) {                // :78-84
    const f3 h = hash3(seed) * F3(2.0f, 6.28318530718f, 1.0f) - F3(1.0f, 0.0f, 0.0f);
    const float phi = h.y;
    const float r = powf(h.z, 1.0f / 3.0f);
    const float s = __builtin_sqrtf(1.0f - h.x * h.x);
    float sn, cs;
    sincosf(phi, &sn, &cs);
    return F3(r * (s * sn), r * (s * cs), r * h.x);
}

struct Ray { f3 o, d; float t; };
struct Cam { f3 eye, u, v, n; float width, height, lensRadius, planeDist, focusDist, time0, time1; };
enum { MT_DIFFUSE = 0, MT_METAL = 1, MT_DIALECTRIC = 2 };
struct Mat { int type; f3 albedo, spec; float rough, refIdx; f3 refract; };
struct Rec { f3 pos, normal; float t; Mat m; };

__device__ __forceinline__ Mat diffuse_mat(f3 albedo) { Mat m; m.type = MT_DIFFUSE; m.albedo = albedo; m.spec = F3(0, 0, 0); m.rough = 1.0f; m.refIdx = 1.0f; m.refract = F3(0, 0, 0); return m; }
__device__ __forceinline__ Mat metal_mat(f3 spec, float rough) { Mat m; m.type = MT_METAL; m.albedo = F3(0, 0, 0); m.spec = spec; m.rough = rough; m.refIdx = 0.0f; m.refract = F3(0, 0, 0); return m; }
__device__ __forceinline__ Mat glass_mat(f3 refract, float idx, float rough) { Mat m; m.type = MT_DIALECTRIC; m.albedo = F3(1, 1, 1); m.spec = F3(0.04f, 0.04f, 0.04f); m.refIdx = idx; m.refract = refract; m.rough = rough; return m; }

// ---- primitives, PT/common.glsl:334-500
// Written without per-lane early returns (as the Whitted intersectors, p3d_device_math.h): every lane evaluates the
// shader's expressions in the shader's order under an `ok` predicate; lanes the shader would have returned from compute
// values nobody uses.  A divergent `if` is three scalar instructions on a kernel bound by scalar issue; each test keeps one
// wave-level exit after its cheap rejections and one branch around the record update.  Same arithmetic, same bits.
__device__ __forceinline__ bool hit_triangle(f3 v0, f3 v1, f3 v2, const Ray& r, float tmin, float tmax, Rec& rec) {
    const f3 e1 = v1 - v0, e2 = v2 - v0;
    const f3 pv = cross3(r.d, e2);
    const float det = dot3(pv, e1);
    bool ok = !(det > -0.0000001f && det < 0.0000001f);
    const float inv = 1.0f / det;
    const f3 tv = r.o - v0;
    const float u = inv * dot3(tv, pv);
    ok = ok && !(u < 0.0f || u > 1.0f);
    if (__ballot(ok) == 0) return false;
    const f3 qv = cross3(tv, e1);
    const float v = inv * dot3(r.d, qv);
    ok = ok && !(v < 0.0f || v > 1.0f);              // (sic) PT/common.glsl:364
    const float t = inv * dot3(e2, qv);
    const bool hit = ok && t < tmax && t > tmin;
    if (hit) { rec.t = t; rec.normal = norm3(cross3(e1, e2)); rec.pos = r.o + r.d * t; }
    return hit;
}
// static and moving spheres share the test once the centre is known; `moving` selects
// (pos - c) / radius (hit_movingSphere) instead of normalize(pos - c) (hit_sphere)
__device__ __forceinline__ bool hit_sphere(f3 c, float radius, bool moving, const Ray& r, float tmin, float tmax, Rec& rec) {
    const f3 L = r.o - c;
    const float b = dot3(L, r.d);
    const float cc = dot3(L, L) - radius * radius;
    const float disc = b * b - cc;
    const bool ok = !(cc > 0.0f && b > 0.0f) && !(disc < 0.0f);
    if (__ballot(ok) == 0) return false;
    const float sq = __builtin_sqrtf(disc);          // NaN on the bypassed lanes: every comparison below is false for them
    const float t0 = -b - sq;
    const float t = (t0 < 0.0f) ? (-b + sq) : t0;
    const bool hit = ok && t < tmax && t > tmin;
    if (hit) {
        rec.t = t; rec.pos = r.o + r.d * t;
        if (radius >= 0.0f) rec.normal = moving ? ((rec.pos - c) / radius) : norm3(rec.pos - c);
        else rec.normal = norm3(c - rec.pos);
    }
    return hit;
}

// LDS table of the 10x10 procedural spheres (PT/P3D_RT.glsl:88-178): centre, class, hash seed
struct SmallSphere { float cx, cy, cz; int cls; float seed; };   // cls: -1 absent, 0 moving, 1 diffuse, 2 metal, 3 fuzzy metal, 4 glass
__device__ __forceinline__ void build_small_spheres(SmallSphere* tab) {
    for (int i = threadIdx.x; i < 100; i += blockDim.x) {
        const int x = i / 10 - 5, y = i % 10 - 5;
        const float fx = (float)x, fy = (float)y;
        float seed = fx + fy / 1000.0f;
        const f3 r1 = hash3(seed);
        const f3 c = F3(fx + 0.9f * r1.x, 0.2f, fy + 0.9f * r1.y);
        SmallSphere s; s.cx = c.x; s.cy = c.y; s.cz = c.z; s.seed = seed;
        if (!(len3(c - F3(4.0f, 0.2f, 0.0f)) > 0.9f)) s.cls = -1;
        else if (r1.z < 0.3f) s.cls = 0;
        else if (r1.z < 0.5f) s.cls = 1;
        else if (r1.z < 0.7f) s.cls = 2;
        else if (r1.z < 0.9f) s.cls = 3;
        else s.cls = 4;
        tab[i] = s;
    }
}
__device__ __forceinline__ Mat small_sphere_material(int cls, float seed) {
    if (cls <= 1) { const f3 a = hash3(seed); const f3 b = hash3(seed); return diffuse_mat(a * b); }
    if (cls == 2) return metal_mat((hash3(seed) + F3(1.0f, 1.0f, 1.0f)) * 0.5f, 0.0f);
    if (cls == 3) { const f3 a = (hash3(seed) + F3(1.0f, 1.0f, 1.0f)) * 0.5f; const float rg = hash1(seed); return metal_mat(a, rg); }
    return glass_mat(hash3(seed), 1.2f, 0.0f);
}

// hit_world, PT/P3D_RT.glsl:12-180
// which of the 100 cells hold a sphere at all / a moving one (class 0): wave-uniform bit sets
struct SphereSets { uint64_t present_lo, present_hi, moving_lo, moving_hi; };

__device__ __forceinline__ bool hit_world(const SmallSphere* tab, const SphereSets& sets, int* wave_union, float& gSeed, const Ray& r, float tmin, float tmax, Rec& rec) {
    bool hit = false;
    rec.t = tmax;
    if (hit_triangle(F3(-10.0f, -0.01f, 10.0f), F3(10.0f, -0.01f, 10.0f), F3(-10.0f, -0.01f, -10.0f), r, tmin, rec.t, rec)) { hit = true; rec.m = diffuse_mat(F3(0.2f, 0.2f, 0.2f)); }
    if (hit_triangle(F3(-10.0f, -0.01f, -10.0f), F3(10.0f, -0.01f, 10.0f), F3(10.0f, -0.01f, -10.0f), r, tmin, rec.t, rec)) { hit = true; rec.m = diffuse_mat(F3(0.2f, 0.2f, 0.2f)); }
    if (hit_sphere(F3(-4.0f, 1.0f, 0.0f), 1.0f, false, r, tmin, rec.t, rec)) { hit = true; rec.m = diffuse_mat(F3(0.4f, 0.2f, 0.1f)); }
    if (hit_sphere(F3(4.0f, 1.0f, 0.0f), 1.0f, false, r, tmin, rec.t, rec)) { hit = true; rec.m = metal_mat(F3(0.7f, 0.6f, 0.5f), 0.0f); }
    if (hit_sphere(F3(0.0f, 1.0f, 0.0f), 1.0f, false, r, tmin, rec.t, rec)) { hit = true; rec.m = glass_mat(F3(0, 0, 0), 1.333f, 0.0f); }
    if (hit_sphere(F3(0.0f, 1.0f, 0.0f), -0.5f, false, r, tmin, rec.t, rec)) { hit = true; rec.m = glass_mat(F3(0, 0, 0), 1.333f, 0.0f); }
    // Small spheres: conservative WAVE-LEVEL culling that never changes the result.  A sphere of the
    // 10x10 field lives in y in [0, 0.9] (radius 0.2, motion blur lifts a centre by at most 0.5) and its
    // centre in [fx, fx+0.9] x [fz, fz+0.9].  Each lane bounds the cells its ray segment [tmin, rec.t]
    // can touch inside that slab (margins of 0.05+ dwarf any rounding); the wave takes the union of
    // the 64 ranges (butterfly min/max) and tests only those cells -- a scalar branch per candidate,
    // no divergence.  Rays whose direction is not unit length (fuzzy metals add an offset without
    // re-normalising, and the shader's sphere test assumes |d| = 1) are not geometric: they keep the
    // whole field.  The pixel RNG is advanced for EVERY moving-sphere candidate, in index order, as
    // the shader does.
    int ix0 = 100, ix1 = -100, iz0 = 100, iz1 = -100;             // empty range
    {
        const float dd = dot3(r.d, r.d);
        if (!(fabsf(dd - 1.0f) < 1e-3f)) { ix0 = -5; ix1 = 4; iz0 = -5; iz1 = 4; }
        else {
            float ta = tmin, tb = rec.t;
            const float ylo = -0.05f, yhi = 0.95f;
            if (r.d.y != 0.0f) {
                const float inv = 1.0f / r.d.y;
                const float t0 = (ylo - r.o.y) * inv, t1 = (yhi - r.o.y) * inv;
                ta = fmaxf(ta, fminf(t0, t1)); tb = fminf(tb, fmaxf(t0, t1));
            } else if (!(r.o.y >= ylo && r.o.y <= yhi)) tb = ta - 1.0f;
            if (ta <= tb) {
                const float xa = r.o.x + r.d.x * ta, xb = r.o.x + r.d.x * tb;
                const float za = r.o.z + r.d.z * ta, zb = r.o.z + r.d.z * tb;
                const float m = 0.25f + 0.01f * (fabsf(ta) + fabsf(tb));      // radius + slack growing with distance
                ix0 = max(-5, (int)floorf(fminf(xa, xb) - m - 0.9f)); ix1 = min(4, (int)floorf(fmaxf(xa, xb) + m));
                iz0 = max(-5, (int)floorf(fminf(za, zb) - m - 0.9f)); iz1 = min(4, (int)floorf(fmaxf(za, zb) + m));
            }
        }
    }
    {   // union over the ACTIVE lanes of the wave.  hit_world() is reached in divergent control flow, so
        // a shuffle butterfly would mix in stale registers of inactive lanes and miss active ones;
        // LDS min/max atomics on this wave's four scratch words are exact for any exec mask (DS
        // operations of one wave execute in issue order; the wave barriers only pin the compiler).
        int* u = wave_union + (threadIdx.x >> 6) * 4;
        const int lane = (int)(threadIdx.x & 63);
        if (lane == __builtin_amdgcn_readfirstlane(lane)) { u[0] = 100; u[1] = -100; u[2] = 100; u[3] = -100; }
        __builtin_amdgcn_wave_barrier();
        atomicMin(&u[0], ix0); atomicMax(&u[1], ix1); atomicMin(&u[2], iz0); atomicMax(&u[3], iz1);
        __builtin_amdgcn_wave_barrier();
        ix0 = __builtin_amdgcn_readfirstlane(((volatile int*)u)[0]); ix1 = __builtin_amdgcn_readfirstlane(((volatile int*)u)[1]);
        iz0 = __builtin_amdgcn_readfirstlane(((volatile int*)u)[2]); iz1 = __builtin_amdgcn_readfirstlane(((volatile int*)u)[3]);
        __builtin_amdgcn_wave_barrier();
    }
    // The cells to look at, as a wave-uniform 100-bit set (two scalar registers pairs): every MOVING
    // sphere (its two RNG draws happen whether or not it is near, in index order) plus the present
    // spheres inside the union rectangle.  Walking the set bits instead of all 100 cells removes the
    // LDS read + scalar branch the culled cells used to cost.
    uint64_t near_lo = 0, near_hi = 0;
    if (ix0 <= ix1 && iz0 <= iz1) {
        const uint64_t row = ((1ull << (iz1 - iz0 + 1)) - 1ull) << (iz0 + 5);          // <= 10 bits
        for (int gx = ix0; gx <= ix1; gx++) {
            const int o = (gx + 5) * 10;
            if (o < 64) { near_lo |= row << o; if (o > 54) near_hi |= row >> (64 - o); }
            else near_hi |= row << (o - 64);
        }
    }
    near_lo &= sets.present_lo; near_hi &= sets.present_hi;
    int best = -1;
    // Index order, as the shader walks the field: a cell inside the rectangle is tested (a moving one after its two draws),
    // a moving sphere outside it only advances the pixel's RNG by its two draws.  There are ~28 moving spheres and a handful
    // of near cells, so the far ones are not walked bit by bit (a dozen scalar instructions each: this kernel was bound by
    // scalar issue) but counted -- popcount of the moving-and-far bits below the next near cell -- and their 2 x count
    // additions run in a counted loop: the same float additions in the same order.
    for (int half = 0; half < 2; half++) {
        const uint64_t near_m = half ? near_hi : near_lo, moving_m = half ? sets.moving_hi : sets.moving_lo;
        const uint64_t far_moving = moving_m & ~near_m;
        uint64_t todo = near_m, below = 0;                       // below: bits under the cells already handled
        for (;;) {
            const int b = todo ? __builtin_ctzll(todo) : 64;
            const uint64_t upto = b < 64 ? ((1ull << b) - 1ull) : ~0ull;      // bits below cell b
            int n_far = __popcll(far_moving & upto & ~below);
            for (; n_far >= 2; n_far -= 2) { gSeed += 0.1f; gSeed += 0.1f; gSeed += 0.1f; gSeed += 0.1f; }
            if (n_far) { gSeed += 0.1f; gSeed += 0.1f; }
            if (b == 64) break;
            todo &= todo - 1;
            below = upto | (1ull << b);
            const int i = half * 64 + b;
            const bool moving = (moving_m >> b) & 1ull;          // wave-uniform
            float h = 0.0f;
            if (moving) {          // motion blur: centre interpolated towards a RANDOM centre1, drawn per call
                gSeed += 0.1f; const float sa = gSeed;
                gSeed += 0.1f; const float sb = gSeed;
                h = (float)base_hash(__float_as_uint(sa), __float_as_uint(sb)) / (float)0xffffffffU;   // hash1(gSeed)
            }
            const SmallSphere s = tab[i];
            f3 c = F3(s.cx, s.cy, s.cz);
            if (moving) {
                const f3 c1 = c + F3(0.0f, h * 0.5f, 0.0f);
                c = c + (c1 - c) * ((r.t - 0.0f) / (1.0f - 0.0f));
            }
            if (hit_sphere(c, 0.2f, moving, r, tmin, rec.t, rec)) { hit = true; best = i; }
        }
    }
    if (best >= 0) rec.m = small_sphere_material(tab[best].cls, tab[best].seed);
    return hit;
}

__device__ __forceinline__ float schlick(float cosine, float r0) { r0 = r0 * r0; return r0 + (1.0f - r0) * powf(1.0f - cosine, 5.0f); }

// scatter, PT/common.glsl:217-324
__device__ __forceinline__ bool scatter(float& gSeed, const Ray& in, const Rec& rec, f3& atten, Ray& out) {
    f3 precise = rec.pos + rec.normal * PT_EPS;
    if (rec.m.type == MT_DIFFUSE) {
        const f3 S = rec.pos + rec.normal + norm3(random_in_unit_sphere(gSeed));
        const f3 dir = norm3(S - rec.pos);
        out.o = precise; out.d = norm3(dir); out.t = in.t;
        atten = rec.m.albedo * fmaxf(dot3(out.d, rec.normal), 0.0f) / PT_PI;
        return true;
    }
    if (rec.m.type == MT_METAL) {
        f3 dir = norm3(in.d - 2.0f * dot3(in.d, rec.normal) * rec.normal);
        dir = dir + rec.m.rough * random_in_unit_sphere(gSeed);
        out.o = precise; out.d = dir; out.t = in.t;
        atten = rec.m.spec;
        return true;
    }
    atten = rec.m.albedo;
    f3 outward; float niOverNt, cosine, etaI, etaT;
    if (dot3(in.d, rec.normal) > 0.0f) {
        outward = -rec.normal; niOverNt = rec.m.refIdx; cosine = dot3(in.d, rec.normal); etaI = rec.m.refIdx; etaT = 1.0f;
    } else {
        outward = rec.normal; niOverNt = 1.0f / rec.m.refIdx; cosine = -dot3(in.d, rec.normal); etaI = 1.0f; etaT = rec.m.refIdx;
    }
    const float r0 = (etaI - etaT) / (etaI + etaT);
    const float k = 1.0f - niOverNt * niOverNt * (1.0f - cosine * cosine);
    const float reflectProb = (k < 0.0f) ? 1.0f : schlick(cosine, r0);
    if (hash1(gSeed) < reflectProb) {
        f3 dir = in.d - 2.0f * dot3(rec.normal, in.d) * rec.normal;           // reflect()
        dir = dir + rec.m.rough * random_in_unit_sphere(gSeed);
        out.o = rec.pos + outward * PT_EPS; out.d = dir; out.t = in.t;         // "normalize(dir);" result unused
    } else {
        f3 refr = norm3(niOverNt * in.d + (niOverNt * cosine - __builtin_sqrtf(k)) * outward);
        refr = mix3(refr, norm3(outward + random_in_unit_sphere(gSeed)), rec.m.rough * rec.m.rough);
        const f3 ab = F3(expf(rec.m.refract.x * -rec.t), expf(rec.m.refract.y * -rec.t), expf(rec.m.refract.z * -rec.t));
        atten = atten * ab;
        out.o = rec.pos - outward * PT_EPS; out.d = refr; out.t = in.t;
    }
    return true;
}

// directlighting, PT/P3D_RT.glsl:182-232
__device__ __forceinline__ f3 direct_lighting(const SmallSphere* tab, const SphereSets& sets, int* wave_union, float& gSeed, f3 lpos, const Ray& r, const Rec& rec) {
    f3 lightDir = norm3(lpos - rec.pos);
    const float dotRec = fmaxf(dot3(rec.normal, lightDir), 0.0f);
    if (!(dotRec > 0.0f)) return F3(0, 0, 0);
    Ray feeler; feeler.o = rec.pos + PT_EPS * rec.normal; feeler.d = lightDir; feeler.t = 0.0f;
    const float size = len3(lightDir);          // (sic) length of the normalised direction
    Rec dummy;
    if (hit_world(tab, sets, wave_union, gSeed, feeler, 0.0f, size, dummy)) return F3(0, 0, 0);
    f3 specCol, diffCol; float shininess, diffuse, specular;
    if (rec.m.type == MT_DIFFUSE) { specCol = F3(0.1f, 0.1f, 0.1f); diffCol = rec.m.albedo; shininess = 10.0f; diffuse = 1.0f; specular = 0.0f; }
    else if (rec.m.type == MT_METAL) { specCol = rec.m.albedo; diffCol = F3(0, 0, 0); shininess = 100.0f; diffuse = 0.0f; specular = 1.0f; }
    else { specCol = F3(0.004f, 0.004f, 0.004f); diffCol = F3(0, 0, 0); shininess = 100.0f; diffuse = 0.0f; specular = 1.0f; }
    lightDir = norm3(lightDir);
    const f3 H = norm3(lightDir - r.d);
    diffCol = diffCol * fmaxf(0.0f, dot3(rec.normal, lightDir));                        // pl.color = (1,1,1)
    specCol = specCol * powf(fmaxf(0.0f, dot3(rec.normal, H)), shininess);
    return diffCol * diffuse + specCol * specular;
}

// rayColor, PT/P3D_RT.glsl:234-282 (MAX_BOUNCES 10, RUSSIAN_ROULETTE false)
__device__ __forceinline__ f3 ray_color(const SmallSphere* tab, const SphereSets& sets, int* wave_union, float& gSeed, Ray r) {
    Rec rec;
    rec.pos = F3(0, 0, 0); rec.normal = F3(0, 0, 0); rec.t = 0.0f; rec.m = diffuse_mat(F3(0, 0, 0));
    f3 col = F3(0, 0, 0), thr = F3(1, 1, 1);
    for (int i = 0; i < 10; ++i) {
        if (hit_world(tab, sets, wave_union, gSeed, r, 0.001f, 10000.0f, rec)) {
            col = col + direct_lighting(tab, sets, wave_union, gSeed, F3(-10.0f, 15.0f, 0.0f), r, rec) * thr;
            col = col + direct_lighting(tab, sets, wave_union, gSeed, F3(8.0f, 15.0f, 3.0f), r, rec) * thr;
            col = col + direct_lighting(tab, sets, wave_union, gSeed, F3(1.0f, 15.0f, -9.0f), r, rec) * thr;
            Ray sr; f3 atten;
            if (scatter(gSeed, r, rec, atten, sr)) { r = sr; thr = thr * atten; }
        } else {
            const float t = 0.8f * (r.d.y + 1.0f);
            col = col + thr * mix3(F3(1, 1, 1), F3(0.5f, 0.7f, 1.0f), t);
            break;
        }
    }
    return col;
}

struct PtLaunch {
    Cam cam;                 // createCamera() result (host, PT/common.glsl:101-128)
    float res_x, res_y;
    int32_t ires_x, ires_y;
    int32_t n_frames, first_frame, frame_stride;
    float time0, dt;
    float* rgba; float* linear;
    // linear sums only (no buffer-A recurrence to carry from frame to frame): the frames of a strip are cut into n_chunks
    // runs of chunk_frames, one wave each, whose partial sums land in partial[chunk][pixel][3]; pt_sum_chunks_kernel adds
    // them in chunk order.  n_chunks <= 1: one wave per strip carries all frames (needed for rgba).
    int32_t n_chunks, chunk_frames;
    float* partial;
};

// One wave (a 16x4 pixel strip) per workgroup: a wave keeps its pixels for all n_frames frames, so its
// run time varies a lot from strip to strip; with four waves per workgroup the wave slots of the early
// finishers stayed reserved until the slowest one was done.  PT_WAVES_PER_EU caps the register budget.
// Round 3: the launch was bound by its LONGEST strips -- 256 frames of a strip full of glass are a serial chain of
// ~200 ms while the average strip needs 10, and the profile showed 1.5 of 5 wave slots per SIMD occupied on average
// (profiles/r03_pathtracer_pmc.json).  Callers that want the linear sums only get the frames of a strip cut into runs
// (PtLaunch::n_chunks): 221 -> 85 ms for 1080p x 256 samples.
#ifndef PT_WAVES_PER_EU
#define PT_WAVES_PER_EU 5
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PT_WAVES_PER_EU, 8))) void pt_frames_kernel(const PtLaunch P) {
    __shared__ SmallSphere tab[100];
    __shared__ int wave_union[4];          // cell range union scratch of this wave (hit_world)
    build_small_spheres(tab);
    __syncthreads();
    SphereSets sets;
    {   // lane i looks at cells i and 64 + i; ballots make the sets wave-uniform
        const int lane = (int)threadIdx.x;
        const int c0 = tab[lane].cls, c1 = lane < 36 ? tab[64 + lane].cls : -1;
        sets.present_lo = __ballot(c0 >= 0); sets.present_hi = __ballot(c1 >= 0);
        sets.moving_lo = __ballot(c0 == 0); sets.moving_hi = __ballot(c1 == 0);
    }
    const int tiles_x = (P.ires_x + 15) / 16;
    const int n_strips = tiles_x * ((P.ires_y + 3) / 4);
    const int strip = (int)blockIdx.x % n_strips, chunk = (int)blockIdx.x / n_strips;     // chunk-major: a strip's runs start far apart
    const int tx = strip % tiles_x, ty = strip / tiles_x;
    const int x = tx * 16 + (threadIdx.x & 15), y = ty * 4 + (threadIdx.x >> 4);
    if (x >= P.ires_x || y >= P.ires_y) return;
    const float fcx = (float)x + 0.5f, fcy = (float)y + 0.5f;                 // gl_FragCoord
    const float pix_hash = (float)base_hash(__float_as_uint(fcx), __float_as_uint(fcy)) / (float)0xffffffffU;
    float prev0 = 0.0f, prev1 = 0.0f, prev2 = 0.0f, prevw = 0.0f;             // buffer A texel of this pixel
    f3 sum = F3(0, 0, 0);
    const int j0 = P.n_chunks > 1 ? chunk * P.chunk_frames : 0;
    const int j1 = P.n_chunks > 1 ? (j0 + P.chunk_frames < P.n_frames ? j0 + P.chunk_frames : P.n_frames) : P.n_frames;
    for (int j = j0; j < j1; j++) {
        const int k = P.first_frame + j * P.frame_stride;
        const float iTime = P.time0 + (float)k * P.dt;
        float gSeed = pix_hash + iTime;                                       // PT/P3D_RT.glsl:288
        const f2 jit = hash2(gSeed);
        const float psx = fcx + jit.x, psy = fcy + jit.y;
        // getRay, PT/common.glsl:130-146
        const f2 d = random_in_unit_disk(gSeed);
        const float lsx = P.cam.lensRadius * d.x, lsy = P.cam.lensRadius * d.y;
        const float time = P.cam.time0 + hash1(gSeed) * (P.cam.time1 - P.cam.time0);
        const float ppx = P.cam.width * (psx / P.res_x - 0.5f) * P.cam.focusDist;
        const float ppy = P.cam.height * (psy / P.res_y - 0.5f) * P.cam.focusDist;
        Ray r;
        r.o = P.cam.eye + P.cam.u * lsx + P.cam.v * lsy;
        r.d = norm3(P.cam.u * (ppx - lsx) + P.cam.v * (ppy - lsy) + P.cam.n * (-P.cam.focusDist * P.cam.planeDist));
        r.t = time;
        f3 color = ray_color(tab, sets, wave_union, gSeed, r);
        sum = sum + color;
        // accumulation, PT/P3D_RT.glsl:345-365 -- only when the caller wants buffer A (six powf per frame and pixel that a
        // request for the linear sums never reads)
        if (P.rgba) {
            const f3 prevLinear = pow3(F3(prev0, prev1, prev2), 2.2f);
            const float w = prevw + 1.0f;
            color = mix3(prevLinear, color, 1.0f / w);
            const f3 g = pow3(color, 1.0f / 2.2f);
            prev0 = g.x; prev1 = g.y; prev2 = g.z; prevw = w;
        }
    }
    const size_t p = (size_t)y * P.ires_x + x;
    if (P.n_chunks > 1) {
        float* dst = P.partial + ((size_t)chunk * P.ires_x * P.ires_y + p) * 3;
        dst[0] = sum.x; dst[1] = sum.y; dst[2] = sum.z;
        return;
    }
    if (P.rgba) reinterpret_cast<float4*>(P.rgba)[p] = make_float4(prev0, prev1, prev2, prevw);
    if (P.linear) { P.linear[3 * p] = sum.x; P.linear[3 * p + 1] = sum.y; P.linear[3 * p + 2] = sum.z; }
}

// linear[i] = partial[0][i] + partial[1][i] + ... in chunk order (the same order on every run: same bits)
__global__ void pt_sum_chunks_kernel(const float* partial, float* linear, size_t n, int n_chunks) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = partial[i];
    for (int c = 1; c < n_chunks; c++) s += partial[(size_t)c * n + i];
    linear[i] = s;
}

__global__ void pt_hash_kernel(uint32_t n, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = base_hash(a[i], b[i]);
}

hipError_t launch_pt_frames(const PtLaunch& P, hipStream_t stream) {
    const int tiles = ((P.ires_x + 15) / 16) * ((P.ires_y + 3) / 4);
    const int chunks = P.n_chunks > 1 ? P.n_chunks : 1;
    hipLaunchKernelGGL(pt_frames_kernel, dim3((unsigned)tiles * (unsigned)chunks), dim3(64), 0, stream, P);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || chunks == 1) return e;
    const size_t n = (size_t)P.ires_x * P.ires_y * 3;
    hipLaunchKernelGGL(pt_sum_chunks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, P.partial, P.linear, n, chunks);
    return hipGetLastError();
}
hipError_t launch_pt_hash(uint32_t n, const uint32_t* a, const uint32_t* b, uint32_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(pt_hash_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, a, b, out);
    return hipGetLastError();
}

}  // namespace p3dpt
