// p3d_device_math.h -- float3 algebra and the four primitive intersectors, written in the
// reference's evaluation order (RT/ = /root/reference/P3D_RayTracer_Template2/).  Must be
// compiled with -ffp-contract=off and without fast-math: these expressions decide hits.
#ifndef P3D_DEVICE_MATH_H
#define P3D_DEVICE_MATH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace p3d {

#define P3D_EPS 0.001f   // RT/macros.h:1

struct V3 { float x, y, z; };

// IEEE-754 correctly rounded float divide / square root.  hipcc keeps both correctly rounded
// by default (-fhip-fp32-correctly-rounded-divide-sqrt, passed explicitly by the build); the
// __fdiv_rn/__fsqrt_rn spellings are avoided because __fsqrt_rn maps to the NATIVE (1 ulp)
// square root in this ROCm's headers.
__device__ __forceinline__ float fdiv(float a, float b) { return a / b; }
__device__ __forceinline__ float fsqrt(float a) { return __builtin_sqrtf(a); }
// 1.0f / x, correctly rounded, in 3 instructions where the compiler's division takes 11: the hardware reciprocal (1 ulp)
// and one Newton step in FMA arithmetic, r = r0 + r0 * (1 - x * r0).  Equal to 1.0f / x for EVERY float whose exponent
// keeps x and 1 / x normal -- checked over all 2^32 bit patterns on the device (p3d_debug_check_rcp,
// tests/test_gpu_exact_math.py); zeros, subnormals, infinities, NaNs and |x| >= 2^126 take the division itself behind a
// wave-level branch.  Triangle::intercepts divides 1.0 by det (via double: the same correctly rounded value) in every test.
__device__ __forceinline__ float frcp(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r0, 1.0f);
    float r = __builtin_fmaf(e, r0, r0);
    const bool plain = ((__float_as_uint(x) & 0x7f800000u) - 0x00800000u) < 0x7e000000u;     // 2^-126 <= |x| < 2^126
    if (__ballot(!plain) != 0) {
        if (!plain) r = 1.0f / x;
    }
    return r;
}

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 mul(V3 a, float f) { return mk(a.x * f, a.y * f, a.z * f); }
__device__ __forceinline__ V3 cmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 u, V3 v) {                       // RT/vector.cpp:85-100
    return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
__device__ __forceinline__ float vlen(V3 a) { return fsqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 normalized(V3 a) {                        // RT/vector.cpp:66-71
    float l = fdiv(1.0f, vlen(a));
    return mk(a.x * l, a.y * l, a.z * l);
}
__device__ __forceinline__ float clamp01(float v) { return (v < 0.0f) ? 0.0f : ((v > 1.0f) ? 1.0f : v); }
__device__ __forceinline__ V3 clampc(V3 c) { return mk(clamp01(c.x), clamp01(c.y), clamp01(c.z)); }
__device__ __forceinline__ uint32_t u8fromfloat(float x) {              // RT/maths.h:113-117
    float s = x * 255.99f;
    return (s >= 255.0f) ? 255u : (uint32_t)(uint8_t)(int)s;
}

struct Ray { V3 o, d; };

// ------------------------------------------------------------------ primitive tests
// Written WITHOUT per-lane early returns: every lane evaluates the reference's expressions in the reference's
// order and carries an `ok` predicate; lanes that the reference would have returned from earlier compute
// garbage that is never used.  On gfx950 a scalar-unit instruction costs about three vector instructions of
// issue time (measured: 5.2 vs 1.7 cycles per wave-instruction per SIMD, tools/ubench/valu_rate.hip), and every
// divergent `if` is three of them (s_and_saveexec / s_cbranch_execz / s_or exec) plus hazard s_nops: the branchy
// form of these tests made the level-1 kernel scalar-bound (459 SALU per wave).  Each test keeps ONE wave-level
// exit -- no active lane can still hit -- taken after its cheapest rejection.  Same arithmetic, same bits.

// Triangle::intercepts, RT/scene.cpp:55-88 (e1, e2 are the stored P1-P0, P2-P0)
// FAST_RCP: 1 / det by frcp() -- the kernels that read the scene from HBM, where the eight vector instructions saved per
// test count (config 3: +3.5 %) and the guard's three scalar ones do not; scenes served from LDS are bound by scalar issue
// and keep the division (with frcp there: config 2 -4.5 %, config 4 -4.5 %; tools/r03/exp36.sh).  Same bits either way.
template <bool FAST_RCP = false>
__device__ __forceinline__ bool hit_triangle(const Ray& r, V3 p0, V3 e1, V3 e2, float& t) {
    const V3 h = cross(r.d, e2);
    const float det = dot(e1, h);
    bool ok = !(det > -P3D_EPS && det < P3D_EPS);
    if (__ballot(ok) == 0) return false;
    const float f = FAST_RCP ? frcp(det) : fdiv(1.0f, det);
    const V3 s = sub(r.o, p0);
    const float u = f * dot(s, h);
    ok = ok & !(u < 0.0f || u > 1.0f);
    const V3 q = cross(s, e1);
    const float v = f * dot(r.d, q);
    ok = ok & !(v < 0.0f || u + v > 1.0f);
    const float t0 = f * dot(e2, q);
    ok = ok & (t0 > P3D_EPS);
    t = t0;
    return ok;
}
// Sphere::intercepts, RT/scene.cpp:149-172
__device__ __forceinline__ bool hit_sphere(const Ray& r, V3 c, float radius, float& t) {
    const V3 L = sub(r.o, c);
    const float a = dot(r.d, r.d);
    const float b = dot(r.d, L) * 2.0f;
    const float cc = dot(L, L) - radius * radius;
    const float delta = b * b - 4.0f * a * cc;
    bool ok = !(delta < 0.0f);
    if (__ballot(ok) == 0) return false;
    const float sq = fsqrt(delta);
    float t0 = fdiv(-b - sq, 2.0f * a);
    float t1 = fdiv(-b + sq, 2.0f * a);
    const bool sw = t0 > t1;                                   // if (t0 > t1) swap(t0, t1)
    const float lo = sw ? t1 : t0, hi = sw ? t0 : t1;
    const bool neg = lo < 0.0f;                                // if (t0 < 0) { t0 = t1; if (t0 < 0) return false; }
    t = neg ? hi : lo;
    ok = ok & !(neg && hi < 0.0f);
    return ok;
}
// aaBox::intercepts, RT/scene.cpp:198-278; nrm = the face normal the reference stores as a
// side effect
__device__ __forceinline__ bool hit_aabox(const Ray& r, V3 mn, V3 mx, float& t, V3& nrm) {
    float aux = fdiv(1.0f, r.d.x);
    const float ax0 = (mn.x - r.o.x) * aux, ax1 = (mx.x - r.o.x) * aux;
    const float tminx = aux >= 0.0f ? ax0 : ax1, tmaxx = aux >= 0.0f ? ax1 : ax0;
    aux = fdiv(1.0f, r.d.y);
    const float ay0 = (mn.y - r.o.y) * aux, ay1 = (mx.y - r.o.y) * aux;
    const float tminy = aux >= 0.0f ? ay0 : ay1, tmaxy = aux >= 0.0f ? ay1 : ay0;
    aux = fdiv(1.0f, r.d.z);
    const float az0 = (mn.z - r.o.z) * aux, az1 = (mx.z - r.o.z) * aux;
    const float tminz = aux >= 0.0f ? az0 : az1, tmaxz = aux >= 0.0f ? az1 : az0;
    // entering face: x if tminx > tminy else y; then z if tminz > that
    const bool inx = tminx > tminy;
    float tIn = inx ? tminx : tminy;
    const bool inz = tminz > tIn;
    const float sIn = (inz ? tminz : tIn) < 0.0f ? -1.0f : 1.0f;         // sign rule of the chosen axis' tmin
    tIn = inz ? tminz : tIn;
    const V3 fIn = mk(inz ? 0.0f : (inx ? sIn : 0.0f), inz ? 0.0f : (inx ? 0.0f : sIn), inz ? sIn : 0.0f);
    const bool outx = tmaxx < tmaxy;
    float tOut = outx ? tmaxx : tmaxy;
    const bool outz = tmaxz < tOut;
    const float sOut = (outz ? tmaxz : tOut) < 0.0f ? -1.0f : 1.0f;
    tOut = outz ? tmaxz : tOut;
    const V3 fOut = mk(outz ? 0.0f : (outx ? sOut : 0.0f), outz ? 0.0f : (outx ? 0.0f : sOut), outz ? sOut : 0.0f);
    const bool ok = tIn < tOut && tOut > P3D_EPS;
    const bool entering = tIn > P3D_EPS;
    t = entering ? tIn : tOut;
    nrm = entering ? fIn : fOut;
    return ok;
}
// Plane::intercepts, RT/scene.cpp:122-141
__device__ __forceinline__ bool hit_plane(const Ray& r, V3 pn, float D, float& t) {
    const float denominator = dot(pn, r.d);
    bool ok = !(fabsf(denominator) < P3D_EPS);
    const float numerator = dot(pn, r.o) + D;
    const float taux = -fdiv(numerator, denominator);
    ok = ok & !(taux <= 0.0f);
    t = taux;
    return ok;
}
// AABB::intercepts, RT/boundingBox.cpp:64-124, only for the default [-1,1]^3 box that
// bounds planes inside the reference's BVH / grid
__device__ __forceinline__ bool ref_unit_box_hit(const Ray& r) {
    float txn, tyn, tzn, txx, tyx, tzx;
    float a = fdiv(1.0f, r.d.x);
    if (a >= 0.0f) { txn = (-1.0f - r.o.x) * a; txx = (1.0f - r.o.x) * a; }
    else           { txn = (1.0f - r.o.x) * a; txx = (-1.0f - r.o.x) * a; }
    float b = fdiv(1.0f, r.d.y);
    if (b >= 0.0f) { tyn = (-1.0f - r.o.y) * b; tyx = (1.0f - r.o.y) * b; }
    else           { tyn = (1.0f - r.o.y) * b; tyx = (-1.0f - r.o.y) * b; }
    float c = fdiv(1.0f, r.d.z);
    if (c >= 0.0f) { tzn = (-1.0f - r.o.z) * c; tzx = (1.0f - r.o.z) * c; }
    else           { tzn = (1.0f - r.o.z) * c; tzx = (-1.0f - r.o.z) * c; }
    float t0 = (txn > tyn) ? ((txn > tzn) ? txn : tzn) : ((tyn > tzn) ? tyn : tzn);
    float t1 = (txx < tyx) ? ((txx < tzx) ? txx : tzx) : ((tyx < tzx) ? tyx : tzx);
    return (t0 < t1 && t1 > 0.0f);
}


}  // namespace p3d
#endif
