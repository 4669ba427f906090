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


}  // namespace p3d
#endif
