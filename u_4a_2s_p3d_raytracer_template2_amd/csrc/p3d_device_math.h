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
// Triangle::intercepts, RT/scene.cpp:55-88 (e1, e2 are the stored P1-P0, P2-P0)
__device__ __forceinline__ bool hit_triangle(const Ray& r, V3 p0, V3 e1, V3 e2, float& t) {
    V3 h = cross(r.d, e2);
    float det = dot(e1, h);
    if (det > -P3D_EPS && det < P3D_EPS) return false;
    float f = fdiv(1.0f, det);
    V3 s = sub(r.o, p0);
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, e1);
    float v = f * dot(r.d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t0 = f * dot(e2, q);
    if (t0 > P3D_EPS) { t = t0; return true; }
    return false;
}
// Sphere::intercepts, RT/scene.cpp:149-172
__device__ __forceinline__ bool hit_sphere(const Ray& r, V3 c, float radius, float& t) {
    V3 L = sub(r.o, c);
    float a = dot(r.d, r.d);
    float b = dot(r.d, L) * 2.0f;
    float cc = dot(L, L) - radius * radius;
    float delta = b * b - 4.0f * a * cc;
    if (delta < 0.0f) return false;
    float sq = fsqrt(delta);
    float t0 = fdiv(-b - sq, 2.0f * a);
    float t1 = fdiv(-b + sq, 2.0f * a);
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < 0.0f) { t0 = t1; if (t0 < 0.0f) return false; }
    t = t0;
    return true;
}
// aaBox::intercepts, RT/scene.cpp:198-278; nrm = the face normal the reference stores as a
// side effect
__device__ __forceinline__ bool hit_aabox(const Ray& r, V3 mn, V3 mx, float& t, V3& nrm) {
    float tminx, tminy, tminz, tmaxx, tmaxy, tmaxz;
    float aux = fdiv(1.0f, r.d.x);
    if (aux >= 0.0f) { tminx = (mn.x - r.o.x) * aux; tmaxx = (mx.x - r.o.x) * aux; }
    else             { tminx = (mx.x - r.o.x) * aux; tmaxx = (mn.x - r.o.x) * aux; }
    aux = fdiv(1.0f, r.d.y);
    if (aux >= 0.0f) { tminy = (mn.y - r.o.y) * aux; tmaxy = (mx.y - r.o.y) * aux; }
    else             { tminy = (mx.y - r.o.y) * aux; tmaxy = (mn.y - r.o.y) * aux; }
    aux = fdiv(1.0f, r.d.z);
    if (aux >= 0.0f) { tminz = (mn.z - r.o.z) * aux; tmaxz = (mx.z - r.o.z) * aux; }
    else             { tminz = (mx.z - r.o.z) * aux; tmaxz = (mn.z - r.o.z) * aux; }
    float tIn, tOut; V3 fIn, fOut;
    if (tminx > tminy) { tIn = tminx; fIn = mk(tminx < 0.0f ? -1.0f : 1.0f, 0.0f, 0.0f); }
    else               { tIn = tminy; fIn = mk(0.0f, tminy < 0.0f ? -1.0f : 1.0f, 0.0f); }
    if (tminz > tIn)   { tIn = tminz; fIn = mk(0.0f, 0.0f, tminz < 0.0f ? -1.0f : 1.0f); }
    if (tmaxx < tmaxy) { tOut = tmaxx; fOut = mk(tmaxx < 0.0f ? -1.0f : 1.0f, 0.0f, 0.0f); }
    else               { tOut = tmaxy; fOut = mk(0.0f, tmaxy < 0.0f ? -1.0f : 1.0f, 0.0f); }
    if (tmaxz < tOut)  { tOut = tmaxz; fOut = mk(0.0f, 0.0f, tmaxz < 0.0f ? -1.0f : 1.0f); }
    if (tIn < tOut && tOut > P3D_EPS) {
        if (tIn > P3D_EPS) { t = tIn; nrm = fIn; }
        else               { t = tOut; nrm = fOut; }
        return true;
    }
    return false;
}
// Plane::intercepts, RT/scene.cpp:122-141
__device__ __forceinline__ bool hit_plane(const Ray& r, V3 pn, float D, float& t) {
    float denominator = dot(pn, r.d);
    if (fabsf(denominator) < P3D_EPS) return false;
    float numerator = dot(pn, r.o) + D;
    float taux = -fdiv(numerator, denominator);
    if (taux <= 0.0f) return false;
    t = taux;
    return true;
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
