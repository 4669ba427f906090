// p3d_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Whitted hot path.
//
// One lane = one pixel sample tree.  A wave owns a 16x4 pixel tile, a 256-thread workgroup a
// 16x16 tile; the blockIdx -> tile map keeps each XCD (own L2) on a contiguous band of the
// image.  The reference's recursion rayTracing() (RT/main.cpp:530-721) runs as an iterative
// post-order machine whose frames and BVH traversal stack live in LDS, laid out
// [slot][lane] so every ds_read/ds_write is bank-conflict free.
//
// Numerics: this file must be compiled with -ffp-contract=off and without fast-math.  Every
// expression that decides a hit, a hit distance or a colour is written in the reference's
// evaluation order (citations RT/ = /root/reference/P3D_RayTracer_Template2/); the
// reference's "1.0 / x" double divides rounded to float equal IEEE float division
// (53 >= 2*24+2 bits, innocuous double rounding).  Only the BVH slab test is ours: it is
// conservative (padded boxes), because the reference's closest hit is brute force.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "p3d_device_types.h"

namespace p3d {

#define P3D_EPS 0.001f   // RT/macros.h:1
#define P3D_DONE ((int32_t)0x80000000)

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

// ------------------------------------------------------------------ per-lane work counters
struct Ctr { uint32_t closest, shadow, box, sph, tri, aab, pln; };

// ------------------------------------------------------------------ BVH traversal
struct Hit {
    float t;
    uint32_t ref;        // kind<<30 | index ; 0xFFFFFFFF = miss
    uint32_t sid;        // scene index
    uint32_t mat;
};

// LDS traversal stack: 8-byte entries {node ref, entry distance}, [slot][lane]
struct TravStack {
    uint2* base;         // points at this lane's slot-0 entry; slot stride = 64 entries
    __device__ __forceinline__ void push(int sp, int32_t node, float t) {
        base[sp * 64] = make_uint2((uint32_t)node, __float_as_uint(t));
    }
    __device__ __forceinline__ uint2 at(int sp) const { return base[sp * 64]; }
};

struct SlabRay { float ox, oy, oz, ix, iy, iz; };

__device__ __forceinline__ SlabRay make_slab(const Ray& r) {
    SlabRay s;
    s.ox = r.o.x; s.oy = r.o.y; s.oz = r.o.z;
    s.ix = fdiv(1.0f, r.d.x); s.iy = fdiv(1.0f, r.d.y); s.iz = fdiv(1.0f, r.d.z);
    return s;
}
// conservative slab test against a padded box; returns entry distance in tn
__device__ __forceinline__ bool slab(const SlabRay& s, float lx, float ly, float lz, float hx,
                                     float hy, float hz, float tlimit, float& tn) {
    float ax = (lx - s.ox) * s.ix, bx = (hx - s.ox) * s.ix;
    float ay = (ly - s.oy) * s.iy, by = (hy - s.oy) * s.iy;
    float az = (lz - s.oz) * s.iz, bz = (hz - s.oz) * s.iz;
    float t0 = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    float t1 = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    t1 = t1 * 1.0000005f + 1e-30f;
    tn = t0;
    return (t0 <= t1) && (t1 >= 0.0f) && (t0 <= tlimit);
}

template <bool COUNT>
__device__ __forceinline__ void leaf_closest(const LaunchParams& P, const Ray& r, int32_t leaf,
                                             Hit& best, Ctr& ctr) {
    uint32_t code = ~(uint32_t)leaf;
    uint32_t first = code >> 3, n = (code & 7u) + 1u;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t ref = P.leaf_refs[first + i];
        uint32_t kind = ref >> kRefKindShift, idx = ref & kRefIndexMask;
        float t; bool h; uint32_t sid = 0, mat = 0;
        if (kind == 1u) {
            const float4* tp = reinterpret_cast<const float4*>(P.tris + idx);
            float4 a = tp[0], b = tp[1], c = tp[2];
            if (COUNT) ctr.tri++;
            h = hit_triangle(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), t);
            sid = __float_as_uint(a.w); mat = __float_as_uint(b.w);
        } else if (kind == 0u) {
            float4 s = *reinterpret_cast<const float4*>(P.spheres + idx);
            if (COUNT) ctr.sph++;
            h = hit_sphere(r, mk(s.x, s.y, s.z), s.w, t);
            if (h && t <= best.t) { PrimMeta m = P.sphere_meta[idx]; sid = m.scene_id; mat = m.material; }
        } else {
            const float4* bp = reinterpret_cast<const float4*>(P.boxes + idx);
            float4 a = bp[0], b = bp[1];
            V3 nn;
            if (COUNT) ctr.aab++;
            h = hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
            sid = __float_as_uint(a.w); mat = __float_as_uint(b.w);
        }
        // "t < closest_t" in scene order == nearest, lowest scene index on ties
        if (h && (t < best.t || (t == best.t && sid < best.sid))) {
            best.t = t; best.ref = ref; best.sid = sid; best.mat = mat;
        }
    }
}

// closest hit over planes (unbounded, outside the BVH) + BVH
template <bool COUNT>
__device__ __forceinline__ Hit closest_hit(const LaunchParams& P, const Ray& r, TravStack st, Ctr& ctr) {
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
    SlabRay s = make_slab(r);
    int sp = 0;
    int32_t cur = 0;
    while (cur != P3D_DONE) {
        while (cur >= 0) {
            const float4* np = reinterpret_cast<const float4*>(P.nodes + cur);
            float4 q0 = np[0], q1 = np[1], q2 = np[2];
            int4 q3 = *reinterpret_cast<const int4*>(np + 3);
            float tn0, tn1;
            bool h0 = slab(s, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, best.t, tn0);
            bool h1 = slab(s, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, best.t, tn1);
            if (COUNT) ctr.box += 2;
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                int32_t nearc = swap ? q3.y : q3.x, farc = swap ? q3.x : q3.y;
                float fart = swap ? tn0 : tn1;
                st.push(sp, farc, fart); sp++;
                cur = nearc;
            } else if (h0) cur = q3.x;
            else if (h1) cur = q3.y;
            else {
                cur = P3D_DONE;
                while (sp > 0) {
                    sp--;
                    uint2 e = st.at(sp);
                    if (__uint_as_float(e.y) <= best.t) { cur = (int32_t)e.x; break; }
                }
            }
        }
        if (cur != P3D_DONE) {
            leaf_closest<COUNT>(P, r, cur, best, ctr);
            cur = P3D_DONE;
            while (sp > 0) {
                sp--;
                uint2 e = st.at(sp);
                if (__uint_as_float(e.y) <= best.t) { cur = (int32_t)e.x; break; }
            }
        }
    }
    return best;
}

// any hit with t < tmax (tmax = +inf, bounded == false: "any intercepts() at all", the
// NONE-mode shadow loop of RT/main.cpp:480-487)
template <bool COUNT>
__device__ __forceinline__ bool leaf_any(const LaunchParams& P, const Ray& r, int32_t leaf, bool bounded,
                                         float tmax, Ctr& ctr) {
    uint32_t code = ~(uint32_t)leaf;
    uint32_t first = code >> 3, n = (code & 7u) + 1u;
    bool occluded = false;
    for (uint32_t i = 0; i < n; i++) {
        uint32_t ref = P.leaf_refs[first + i];
        uint32_t kind = ref >> kRefKindShift, idx = ref & kRefIndexMask;
        float t; bool h;
        if (kind == 1u) {
            const float4* tp = reinterpret_cast<const float4*>(P.tris + idx);
            float4 a = tp[0], b = tp[1], c = tp[2];
            if (COUNT) ctr.tri++;
            h = hit_triangle(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), t);
        } else if (kind == 0u) {
            float4 s = *reinterpret_cast<const float4*>(P.spheres + idx);
            if (COUNT) ctr.sph++;
            h = hit_sphere(r, mk(s.x, s.y, s.z), s.w, t);
        } else {
            const float4* bp = reinterpret_cast<const float4*>(P.boxes + idx);
            float4 a = bp[0], b = bp[1];
            V3 nn;
            if (COUNT) ctr.aab++;
            h = hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
        }
        if (h && (!bounded || t < tmax)) occluded = true;
    }
    return occluded;
}

template <bool COUNT>
__device__ __forceinline__ bool any_hit(const LaunchParams& P, const Ray& r, bool bounded, float tmax,
                                        TravStack st, Ctr& ctr) {
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
    SlabRay s = make_slab(r);
    float tlimit = bounded ? tmax : 3.402823466e+38f;
    int sp = 0;
    int32_t cur = 0;
    while (cur != P3D_DONE) {
        while (cur >= 0) {
            const float4* np = reinterpret_cast<const float4*>(P.nodes + cur);
            float4 q0 = np[0], q1 = np[1], q2 = np[2];
            int4 q3 = *reinterpret_cast<const int4*>(np + 3);
            float tn0, tn1;
            bool h0 = slab(s, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, tlimit, tn0);
            bool h1 = slab(s, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, tlimit, tn1);
            if (COUNT) ctr.box += 2;
            if (h0 && h1) {
                bool swap = tn1 < tn0;
                st.push(sp, swap ? q3.x : q3.y, 0.0f); sp++;
                cur = swap ? q3.y : q3.x;
            } else if (h0) cur = q3.x;
            else if (h1) cur = q3.y;
            else if (sp > 0) { sp--; cur = (int32_t)st.at(sp).x; }
            else cur = P3D_DONE;
        }
        if (cur != P3D_DONE) {
            if (leaf_any<COUNT>(P, r, cur, bounded, tmax, ctr)) return true;
            if (sp > 0) { sp--; cur = (int32_t)st.at(sp).x; }
            else cur = P3D_DONE;
        }
    }
    return false;
}

// ------------------------------------------------------------------ shading
// getNormal(point).normalize() of the hit primitive (RT/main.cpp:587-589)
__device__ __forceinline__ V3 prim_normal(const LaunchParams& P, uint32_t ref, const Ray& r, V3 point) {
    uint32_t kind = ref >> kRefKindShift, idx = ref & kRefIndexMask;
    if (kind == 0u) {                                                   // RT/scene.cpp:174-178
        float4 s = *reinterpret_cast<const float4*>(P.spheres + idx);
        V3 n = normalized(sub(point, mk(s.x, s.y, s.z)));
        return normalized(n);
    } else if (kind == 1u) {                                            // RT/scene.cpp:10-25,46-49
        const float4* tp = reinterpret_cast<const float4*>(P.tris + idx);
        float4 b = tp[1], c = tp[2];
        V3 V = mk(b.x, b.y, b.z), W = mk(c.x, c.y, c.z);
        V3 n = mk((V.y * W.z) - (V.z * W.y), (V.z * W.x) - (V.x * W.z), (V.x * W.y) - (V.y * W.x));
        n = normalized(n);
        return normalized(n);
    } else if (kind == 2u) {
        const float4* bp = reinterpret_cast<const float4*>(P.boxes + idx);
        float4 a = bp[0], b = bp[1];
        float t; V3 nn = mk(0.0f, 0.0f, 0.0f);
        hit_aabox(r, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), t, nn);
        return normalized(nn);
    } else {                                                            // RT/scene.cpp:143-146
        PlaneRec pl = P.planes[idx];
        return normalized(mk(pl.nx, pl.ny, pl.nz));
    }
}

struct Mtl { V3 diff; float kd; V3 spec; float ks; float shine, T, ior, refl; };
__device__ __forceinline__ Mtl load_material(const LaunchParams& P, uint32_t m) {
    const float4* mp = reinterpret_cast<const float4*>(P.materials + m);
    float4 a = mp[0], b = mp[1], c = mp[2];
    Mtl r; r.diff = mk(a.x, a.y, a.z); r.kd = a.w; r.spec = mk(b.x, b.y, b.z); r.ks = b.w;
    r.shine = c.x; r.T = c.y; r.ior = c.z; r.refl = c.w;
    return r;
}

// shade-stack frame in LDS: 12 dwords, [field][lane]
struct Frames {
    uint32_t* base;   // this lane's field-0 of frame-0; field stride 64, frame stride 12*64
    __device__ __forceinline__ uint32_t& f(int frame, int field) { return base[(frame * 12 + field) * 64]; }
    __device__ __forceinline__ void put3(int frame, int field, V3 v) {
        f(frame, field) = __float_as_uint(v.x); f(frame, field + 1) = __float_as_uint(v.y);
        f(frame, field + 2) = __float_as_uint(v.z);
    }
    __device__ __forceinline__ V3 get3(int frame, int field) {
        return mk(__uint_as_float(f(frame, field)), __uint_as_float(f(frame, field + 1)),
                  __uint_as_float(f(frame, field + 2)));
    }
};
enum { FR_C = 0, FR_KR = 3, FR_META = 4, FR_A = 5, FR_RD = 8, FR_IOR = 11 };
#define FR_HAS_REFR 0x40000000u
#define FR_WAIT_REFR 0x80000000u

// processLight(), RT/main.cpp:471-526
template <bool COUNT>
__device__ __forceinline__ void process_light(const LaunchParams& P, V3 L, V3 lcol, V3& color, const Mtl& M,
                                              const Ray& ray, V3 precise, V3 normal, TravStack st, Ctr& ctr) {
    bool insideShadow = false;
    if (dot(L, normal) > 0.0f) {
        Ray sr; sr.o = precise; sr.d = L;
        if (P.accel == 0) {
            insideShadow = any_hit<COUNT>(P, sr, false, 0.0f, st, ctr);
        } else {
            float length = vlen(sr.d);              // BVH::Traverse(Ray&), RT/bvh.cpp:351-352
            sr.d = normalized(sr.d);
            insideShadow = any_hit<COUNT>(P, sr, true, length, st, ctr);
        }
    }
    if (!insideShadow) {
        L = normalized(L);
        V3 H = normalized(add(L, mul(ray.d, -1.0f)));
        float VdotN = dot(H, normal);
        float d1 = dot(normal, L);
        float max1 = (0.0f < d1) ? d1 : 0.0f;        // std::max(0.0f, x)
        float max2 = (0.0f < VdotN) ? VdotN : 0.0f;
        V3 diff = mul(cmul(lcol, M.diff), max1);
        V3 spec = mul(cmul(lcol, M.spec), powf(max2, M.shine));
        color = add(color, add(mul(diff, M.kd), mul(mul(spec, M.ks), 0.4f)));
    }
}

// One primary ray's whole tree: rayTracing(ray, 1, 1.0) of RT/main.cpp:530-721, iterative.
template <bool COUNT>
__device__ __forceinline__ V3 trace_tree(const LaunchParams& P, Ray ray, TravStack st, Frames fr,
                                         int32_t& primary_hit, Ctr& ctr) {
    int fsp = 0;              // frames on the stack == depth - 1
    float ior_1 = 1.0f;
    bool first = true;
    V3 ret = mk(0.0f, 0.0f, 0.0f);
    for (;;) {
        Hit h = closest_hit<COUNT>(P, ray, st, ctr);
        if (first) { primary_hit = (h.ref == 0xFFFFFFFFu) ? -1 : (int32_t)h.sid; first = false; }
        bool descend = false;
        if (h.ref == 0xFFFFFFFFu) {
            ret = mk(P.bg[0], P.bg[1], P.bg[2]);
        } else {
            const int depth = fsp + 1;
            Mtl M = load_material(P, h.mat);
            V3 hit_point = add(ray.o, mul(ray.d, h.t));
            V3 normal = prim_normal(P, h.ref, ray, hit_point);
            V3 precise = add(hit_point, mul(normal, P3D_EPS));
            normal = prim_normal(P, h.ref, ray, precise);
            V3 Vv = mul(ray.d, -1.0f);
            V3 color = mk(0.0f, 0.0f, 0.0f);
            for (uint32_t i = 0; i < P.n_lights; i++) {
                const float4* lp = reinterpret_cast<const float4*>(P.lights + i);
                float4 lpos = lp[0], lcol = lp[1];
                V3 L = sub(mk(lpos.x, lpos.y, lpos.z), hit_point);
                process_light<COUNT>(P, L, mk(lcol.x, lcol.y, lcol.z), color, M, ray, precise, normal, st, ctr);
            }
            if (depth >= P.max_depth) {
                ret = clampc(color);                                     // RT/main.cpp:632-634
            } else {
                bool inside = false;
                if (dot(ray.d, normal) > 0.0f) { normal = mul(normal, -1.0f); inside = true; }
                bool has_refl = M.refl > 0.0f;
                Ray rr; rr.o = precise; rr.d = mk(0.0f, 0.0f, 0.0f);
                if (has_refl) {                                          // RT/main.cpp:646-667
                    V3 rdir = sub(ray.d, mul(mul(normal, dot(ray.d, normal)), 2.0f));
                    rr.d = normalized(rdir);
                }
                float KR; bool has_refr = false; Ray fray; float newIor = 1.0f;
                fray.o = mk(0.0f, 0.0f, 0.0f); fray.d = fray.o;
                if (M.T != 0.0f) {                                       // RT/main.cpp:671-713
                    float R0 = 1.0f, R1 = 1.0f;
                    V3 viewnormal = mul(normal, dot(normal, Vv));
                    V3 viewtangent = sub(viewnormal, Vv);
                    float nn = inside ? ior_1 : fdiv(ior_1, M.ior);
                    float cos_i = vlen(viewnormal);
                    float sin_t = nn * vlen(viewtangent);
                    float insqrt = (float)(1.0 - (double)sin_t * (double)sin_t);   // pow(float,2) is double
                    if (insqrt >= 0.0f) {
                        float cos_t = fsqrt(insqrt);
                        V3 rfr = add(mul(normalized(viewtangent), sin_t), normalized(mul(normal, cos_t)));
                        fray.o = add(hit_point, mul(rfr, 0.001f));
                        fray.d = rfr;
                        newIor = inside ? 1.0f : M.ior;
                        has_refr = true;
                        float den = ior_1 * cos_i + newIor * cos_t;
                        float q0 = fabsf(fdiv(ior_1 * cos_i - newIor * cos_t, den));
                        float q1 = fabsf(fdiv(ior_1 * cos_t - newIor * cos_i, den));
                        R0 = (float)((double)q0 * (double)q0);
                        R1 = (float)((double)q1 * (double)q1);
                    }
                    KR = 0.0f * (R0 + R1);                               // 1 / 2 * (R0 + R1), SURVEY Q5
                } else {
                    KR = M.ks;
                }
                if (has_refl || has_refr) {
                    fr.put3(fsp, FR_C, color);
                    fr.f(fsp, FR_KR) = __float_as_uint(KR);
                    if (has_refl) {
                        fr.f(fsp, FR_META) = h.mat | (has_refr ? FR_HAS_REFR : 0u);
                        fr.put3(fsp, FR_A, fray.o);
                        fr.put3(fsp, FR_RD, fray.d);
                        fr.f(fsp, FR_IOR) = __float_as_uint(newIor);
                        ray = rr;                                        // ior_1 unchanged
                    } else {
                        V3 A = cmul(mul(mk(0.0f, 0.0f, 0.0f), KR), M.spec);
                        fr.f(fsp, FR_META) = h.mat | FR_WAIT_REFR;
                        fr.put3(fsp, FR_A, A);
                        ray = fray; ior_1 = newIor;
                    }
                    fsp++;
                    descend = true;
                } else {
                    V3 zero = mk(0.0f, 0.0f, 0.0f);
                    ret = add(color, add(cmul(mul(zero, KR), M.spec), mul(zero, 1.0f - KR)));
                }
            }
        }
        if (descend) continue;
        // ---- return path: combine into parents (RT/main.cpp:719)
        bool resumed = false;
        while (fsp > 0) {
            int k = fsp - 1;
            uint32_t meta = fr.f(k, FR_META);
            V3 C = fr.get3(k, FR_C);
            float KR = __uint_as_float(fr.f(k, FR_KR));
            if (!(meta & FR_WAIT_REFR)) {
                Mtl M = load_material(P, meta & 0x3FFFFFFFu);
                V3 A = cmul(mul(ret, KR), M.spec);
                if (meta & FR_HAS_REFR) {
                    ray.o = fr.get3(k, FR_A);
                    ray.d = fr.get3(k, FR_RD);
                    ior_1 = __uint_as_float(fr.f(k, FR_IOR));
                    fr.put3(k, FR_A, A);
                    fr.f(k, FR_META) = meta | FR_WAIT_REFR;
                    resumed = true;
                    break;
                }
                ret = add(C, add(A, mul(mk(0.0f, 0.0f, 0.0f), 1.0f - KR)));
            } else {
                V3 A = fr.get3(k, FR_A);
                ret = add(C, add(A, mul(ret, 1.0f - KR)));
            }
            fsp--;
        }
        if (!resumed) return ret;
    }
}

// ------------------------------------------------------------------ the frame kernel
// Camera::PrimaryRay, RT/camera.h:91-108
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

template <bool COUNT>
__global__ __launch_bounds__(kWavesPerGroup * 64) void whitted_frame_kernel(const LaunchParams P) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware tile map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8
    // share one, each XCD has its own L2), so XCD k is given CHUNKS of xcd_chunk consecutive
    // tiles: chunk c goes to XCD c % 8.  xcd_chunk = 1 is the identity map (best load balance:
    // ray-tree depth is very uneven across the image), larger chunks trade balance for L2
    // locality on scenes whose BVH does not fit one L2.
    const int bid = blockIdx.x;
    const int j = bid >> 3;
    const int tile = ((j / P.xcd_chunk) * 8 + (bid & 7)) * P.xcd_chunk + (j % P.xcd_chunk);
    if (tile >= P.n_tiles) return;
    const int tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const int lx = lane & 15, ly = (lane >> 4) + wave * 4;
    const int x = tx * 16 + lx;
    const int row = ty * (kWavesPerGroup * 4) + ly;      // row in the compact local buffer
    const int blk = row / P.row_block;
    const int y = (blk * P.world + P.rank) * P.row_block + (row - blk * P.row_block);
    if (x >= P.res_x || y >= P.res_y) return;            // no barriers below: early exit is safe

    const int wave_dwords = P.trav_stack_entries * 128 + (P.max_depth > 1 ? (P.max_depth - 1) : 1) * 12 * 64;
    uint32_t* wbase = lds + wave * wave_dwords;
    TravStack st; st.base = reinterpret_cast<uint2*>(wbase) + lane;
    Frames fr; fr.base = wbase + P.trav_stack_entries * 128 + lane;

    Ctr ctr = {0, 0, 0, 0, 0, 0, 0};
    V3 color = mk(0.0f, 0.0f, 0.0f);
    int32_t hid = -1;
    if (P.spp == 0) {                                    // RT/main.cpp:756-775
        Ray ray = primary_ray(P, (float)x + 0.5f, (float)y + 0.5f);
        color = clampc(trace_tree<COUNT>(P, ray, st, fr, hid, ctr));
    } else {                                             // RT/main.cpp:776-801
        const int ns = P.spp * P.spp;
        const float4* sp = reinterpret_cast<const float4*>(P.samples) + ((size_t)y * P.res_x + x) * ns;
        for (int s = 0; s < ns; s++) {
            float4 sm = sp[s];
            Ray ray = primary_ray_lens(P, sm.z, sm.w, sm.x, sm.y);
            int32_t h2 = -1;
            V3 c = clampc(trace_tree<COUNT>(P, ray, st, fr, h2, ctr));
            color = add(color, c);
            if (s == 0) hid = h2;
        }
        color = mk(fdiv(color.x, 16.0f), fdiv(color.y, 16.0f), fdiv(color.z, 16.0f));
    }
    const size_t p = (size_t)row * P.res_x + x;
    if (P.rgb8) {
        P.rgb8[3 * p] = (uint8_t)u8fromfloat(color.x);
        P.rgb8[3 * p + 1] = (uint8_t)u8fromfloat(color.y);
        P.rgb8[3 * p + 2] = (uint8_t)u8fromfloat(color.z);
    }
    if (P.rgb32f) { P.rgb32f[3 * p] = color.x; P.rgb32f[3 * p + 1] = color.y; P.rgb32f[3 * p + 2] = color.z; }
    if (P.hit_id) P.hit_id[p] = hid;
    if (COUNT) {
        DeviceCounters* c = P.counters;
        atomicAdd(&c->closest_queries, (unsigned long long)ctr.closest);
        atomicAdd(&c->shadow_queries, (unsigned long long)ctr.shadow);
        atomicAdd(&c->box_tests, (unsigned long long)ctr.box);
        atomicAdd(&c->sphere_tests, (unsigned long long)ctr.sph);
        atomicAdd(&c->tri_tests, (unsigned long long)ctr.tri);
        atomicAdd(&c->aabox_tests, (unsigned long long)ctr.aab);
        atomicAdd(&c->plane_tests, (unsigned long long)ctr.pln);
        atomicAdd(&c->pixels, 1ull);
    }
}

// ------------------------------------------------------------------ rank-0 de-interleave
__global__ void deinterleave_kernel(const uint8_t* __restrict__ gathered, uint8_t* __restrict__ frame,
                                    int res_x, int res_y, int row_block, int world, size_t rank_stride,
                                    int bpp) {
    const size_t row_bytes = (size_t)res_x * bpp;
    const size_t total = row_bytes * res_y;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t y = i / row_bytes, off = i - y * row_bytes;
        int blk = (int)(y / row_block);
        int rank = blk % world, lblk = blk / world;
        size_t lrow = (size_t)lblk * row_block + (y - (size_t)blk * row_block);
        frame[i] = gathered[(size_t)rank * rank_stride + lrow * row_bytes + off];
    }
}

// ------------------------------------------------------------------ unit probe
__global__ void debug_intersect_kernel(uint32_t n, const uint32_t* type, const float* prim12,
                                       const float* origin, const float* dir, int32_t* hit, float* t,
                                       float* normal) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* d = prim12 + 12 * (size_t)i;
    Ray r; r.o = mk(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]);
    r.d = mk(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
    float tt = 0.0f; bool h = false; V3 nn = mk(0.0f, 0.0f, 0.0f);
    switch (type[i]) {
    case 0: {
        V3 c = mk(d[0], d[1], d[2]);
        h = hit_sphere(r, c, d[3], tt);
        if (h) { V3 hp = add(r.o, mul(r.d, tt)); nn = normalized(normalized(sub(hp, c))); }
        break;
    }
    case 1: {
        V3 p0 = mk(d[0], d[1], d[2]), p1 = mk(d[3], d[4], d[5]), p2 = mk(d[6], d[7], d[8]);
        V3 e1 = sub(p1, p0), e2 = sub(p2, p0);
        h = hit_triangle(r, p0, e1, e2, tt);
        if (h) {
            V3 m = mk((e1.y * e2.z) - (e1.z * e2.y), (e1.z * e2.x) - (e1.x * e2.z), (e1.x * e2.y) - (e1.y * e2.x));
            nn = normalized(normalized(m));
        }
        break;
    }
    case 2: {
        V3 f;
        h = hit_aabox(r, mk(d[0], d[1], d[2]), mk(d[3], d[4], d[5]), tt, f);
        if (h) nn = normalized(f);
        break;
    }
    default: {
        V3 pn = mk(d[0], d[1], d[2]);
        h = hit_plane(r, pn, d[3], tt);
        if (h) nn = normalized(pn);
        break;
    }
    }
    hit[i] = h ? 1 : 0;
    t[i] = tt;
    normal[3 * i] = nn.x; normal[3 * i + 1] = nn.y; normal[3 * i + 2] = nn.z;
}

// ------------------------------------------------------------------ launchers (host)
size_t frame_kernel_lds_bytes(const LaunchParams& P) {
    int frames = P.max_depth > 1 ? (P.max_depth - 1) : 1;
    size_t wave_dwords = (size_t)P.trav_stack_entries * 128 + (size_t)frames * 12 * 64;
    return wave_dwords * 4 * kWavesPerGroup;
}

hipError_t launch_frame(const LaunchParams& P, bool count, hipStream_t stream) {
    size_t lds = frame_kernel_lds_bytes(P);
    dim3 grid((unsigned)P.grid_blocks), block(kWavesPerGroup * 64);
    if (count) {
        hipLaunchKernelGGL(whitted_frame_kernel<true>, grid, block, lds, stream, P);
    } else {
        hipLaunchKernelGGL(whitted_frame_kernel<false>, grid, block, lds, stream, P);
    }
    return hipGetLastError();
}

hipError_t prepare_frame_kernels(size_t max_lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(whitted_frame_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(whitted_frame_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
}

hipError_t launch_deinterleave(const void* gathered, void* frame, int res_x, int res_y, int row_block,
                               int world, size_t rank_stride, int bpp, hipStream_t stream) {
    hipLaunchKernelGGL(deinterleave_kernel, dim3(2048), dim3(256), 0, stream,
                       (const uint8_t*)gathered, (uint8_t*)frame, res_x, res_y, row_block, world,
                       rank_stride, bpp);
    return hipGetLastError();
}

hipError_t launch_debug_intersect(uint32_t n, const uint32_t* type, const float* prim12, const float* origin,
                                  const float* dir, int32_t* hit, float* t, float* normal, hipStream_t stream) {
    hipLaunchKernelGGL(debug_intersect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, type, prim12,
                       origin, dir, hit, t, normal);
    return hipGetLastError();
}

}  // namespace p3d
