// p3d_oracle.cpp -- CPU oracle: expression-level restatement of the Whitted hot path of
// P3D_RayTracer_Template2.  TEST INFRASTRUCTURE ONLY (see p3d_oracle.h for who may load it
// and for the pinning status; intersectors/shading/loader are "parity unpinned" against
// reference image bytes because RT/scene.cpp and RT/main.cpp cannot be built here).
//
// Citations: RT/ = /root/reference/P3D_RayTracer_Template2/.  Every quirk of SURVEY.md §0
// (Q1..Q12) that changes pixels is reproduced on purpose and marked where it happens.
//
// Build: g++ -O2 -std=c++14 -ffp-contract=off -fPIC -shared (oracle/Makefile).  All float
// arithmetic is IEEE binary32 evaluated in source order; "1.0 / x" denotes the reference's
// double-precision divides (innocuous double rounding, but kept literal).

#include "p3d_oracle.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stack>
#include <string>
#include <thread>
#include <vector>

namespace {

const float kEps = 0.001f;                       // RT/macros.h:1
const float kPi = 3.141592653589793238462f;      // RT/maths.h:7

// ------------------------------------------------------------------ RT/vector.cpp
struct V3 {
    float x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(float a, float b, float c) : x(a), y(b), z(c) {}
    float get(int axis) const { return axis == 0 ? x : (axis == 1 ? y : z); }
};
inline V3 operator+(const V3& a, const V3& b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(const V3& a, float f) { return V3(a.x * f, a.y * f, a.z * f); }
inline V3 operator/(const V3& a, float f) { return V3(a.x / f, a.y / f, a.z / f); }
inline float dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(const V3& u, const V3& v) {       // RT/vector.cpp:85-100
    return V3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
inline float vlen(const V3& a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
inline V3& normalize(V3& a) {                     // RT/vector.cpp:66-71 (mutates)
    float l = 1.0 / vlen(a);
    a.x *= l; a.y *= l; a.z *= l;
    return a;
}
inline V3 normalized(V3 a) { normalize(a); return a; }

// ------------------------------------------------------------------ RT/color.h
struct Col {
    float r, g, b;
    Col() : r(0), g(0), b(0) {}
    Col(float R, float G, float B) : r(R), g(G), b(B) {}
};
inline Col operator*(const Col& a, float c) { return Col(a.r * c, a.g * c, a.b * c); }
inline Col operator*(const Col& a, const Col& c) { return Col(a.r * c.r, a.g * c.g, a.b * c.b); }
inline Col operator+(const Col& a, const Col& c) { return Col(a.r + c.r, a.g + c.g, a.b + c.b); }
inline Col operator/(const Col& a, float c) { return Col(a.r / c, a.g / c, a.b / c); }
inline float clamp01(float v) { return (v < 0.0) ? 0.0 : ((v > 1.0) ? 1.0 : v); }  // RT/color.h:9
inline Col clampc(const Col& a) { return Col(clamp01(a.r), clamp01(a.g), clamp01(a.b)); }

inline uint8_t u8fromfloat(float x) {             // RT/maths.h:113-117
    return ((x * 255.99f) >= 255.0f ? 255 : (uint8_t)(x * 255.99f));
}
inline float rand_float() {                       // RT/maths.h:67-70
    return ((float)rand() / ((float)RAND_MAX + 1.0));
}

struct RayO { V3 o, d; };

thread_local p3o_counters g_ctr;

// ------------------------------------------------------------------ RT/boundingBox.cpp
struct Box3 {
    V3 mn, mx;
    Box3() : mn(-1.0f, -1.0f, -1.0f), mx(1.0f, 1.0f, 1.0f) {}   // RT/boundingBox.cpp:8-12
    Box3(const V3& a, const V3& b) : mn(a), mx(b) {}
    bool inside(const V3& p) const {              // RT/boundingBox.cpp:41-44 (strict)
        return ((p.x > mn.x && p.x < mx.x) && (p.y > mn.y && p.y < mx.y) &&
                (p.z > mn.z && p.z < mx.z));
    }
    V3 centroid() const { return (mn + mx) / 2; }
    void extend(const Box3& b) {
        if (mn.x > b.mn.x) mn.x = b.mn.x;
        if (mn.y > b.mn.y) mn.y = b.mn.y;
        if (mn.z > b.mn.z) mn.z = b.mn.z;
        if (mx.x < b.mx.x) mx.x = b.mx.x;
        if (mx.y < b.mx.y) mx.y = b.mx.y;
        if (mx.z < b.mx.z) mx.z = b.mx.z;
    }
    bool hit(const RayO& ray, float& t) const {   // RT/boundingBox.cpp:64-124
        g_ctr.aabb_tests++;
        double t0, t1;
        float ox = ray.o.x, oy = ray.o.y, oz = ray.o.z;
        float dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
        float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
        float a = 1.0 / dx;
        if (a >= 0) { tx_min = (mn.x - ox) * a; tx_max = (mx.x - ox) * a; }
        else        { tx_min = (mx.x - ox) * a; tx_max = (mn.x - ox) * a; }
        float b = 1.0 / dy;
        if (b >= 0) { ty_min = (mn.y - oy) * b; ty_max = (mx.y - oy) * b; }
        else        { ty_min = (mx.y - oy) * b; ty_max = (mn.y - oy) * b; }
        float c = 1.0 / dz;
        if (c >= 0) { tz_min = (mn.z - oz) * c; tz_max = (mx.z - oz) * c; }
        else        { tz_min = (mx.z - oz) * c; tz_max = (mn.z - oz) * c; }
        // MAX3 / MIN3 macro shapes of RT/macros.h:5-10
        t0 = (tx_min > ty_min) ? ((tx_min > tz_min) ? tx_min : tz_min)
                               : ((ty_min > tz_min) ? ty_min : tz_min);
        t1 = (tx_max < ty_max) ? ((tx_max < tz_max) ? tx_max : tz_max)
                               : ((ty_max < tz_max) ? ty_max : tz_max);
        t = (t0 < 0) ? t1 : t0;
        return (t0 < t1 && t1 > 0);
    }
};

// ------------------------------------------------------------------ RT/scene.h:23-65
struct Mat {
    Col diff; float kd; Col spec; float ks; float shine; float refl; float T; float ior;
    Mat() : diff(0.2f, 0.2f, 0.2f), kd(0.2f), spec(1.0f, 1.0f, 1.0f), ks(0.8f), shine(20),
            refl(1.0f), T(0.0f), ior(1.0f) {}     // RT/scene.h:27-28
};
struct LightO { V3 pos; Col col; };

// ------------------------------------------------------------------ RT/scene.cpp:10-283
struct Prim {
    int type;
    int material;
    V3 a, b, c;        // sphere: a=center, b.x=radius | tri: points | box: a=min,b=max | plane: a=PN, b.x=D
    V3 nrm;            // tri: unit normal ; box: side-effect normal
    V3 bmin, bmax;     // tri: padded Min/Max
    float raw[12];     // the 12 floats the loader saw (for dumps)

    Box3 bbox() const {
        switch (type) {
        case P3O_TRIANGLE: return Box3(bmin, bmax);                      // RT/scene.cpp:42-44
        case P3O_SPHERE: {                                               // RT/scene.cpp:180-186
            float r = b.x;
            return Box3(V3(a.x - r, a.y - r, a.z - r), V3(a.x + r, a.y + r, a.z + r));
        }
        case P3O_BOX: return Box3(a, b);                                 // RT/scene.cpp:194-196
        default: return Box3();
        }
    }
    V3 centroid() const { return bbox().centroid(); }

    bool hit(const RayO& r, float& t) {
        switch (type) {
        case P3O_TRIANGLE: return hit_tri(r, t);
        case P3O_SPHERE: return hit_sphere(r, t);
        case P3O_BOX: return hit_box(r, t);
        default: return hit_plane(r, t);
        }
    }
    bool hit_tri(const RayO& r, float& t) const {                        // RT/scene.cpp:55-88
        g_ctr.tri_tests++;
        V3 e1 = b - a;
        V3 e2 = c - a;
        V3 h = cross(r.d, e2);
        float det = dot(e1, h);
        if (det > -kEps && det < kEps) return false;
        float f = 1.0 / det;
        V3 s = r.o - a;
        float u = f * dot(s, h);
        if (u < 0.0 || u > 1.0) return false;
        V3 q = cross(s, e1);
        float v = f * dot(r.d, q);
        if (v < 0.0 || u + v > 1.0) return false;
        float t0 = f * dot(e2, q);
        if (t0 > kEps) { t = t0; return true; }
        return false;
    }
    bool hit_sphere(const RayO& r, float& t) const {                     // RT/scene.cpp:149-172
        g_ctr.sphere_tests++;
        float radius = b.x;
        V3 L = r.o - a;
        float qa = dot(r.d, r.d);
        float qb = dot(r.d, L) * 2.0f;
        float qc = dot(L, L) - radius * radius;
        float delta = qb * qb - 4 * qa * qc;
        if (delta < 0) return false;
        float t0 = (-qb - sqrtf(delta)) / (2 * qa);
        float t1 = (-qb + sqrtf(delta)) / (2 * qa);
        if (t0 > t1) std::swap(t0, t1);
        if (t0 < 0) { t0 = t1; if (t0 < 0) return false; }
        t = t0;
        return true;
    }
    bool hit_box(const RayO& ray, float& t) {                            // RT/scene.cpp:198-278
        g_ctr.box_tests++;
        V3 tmin, tmax;
        float tIn, tOut;
        float aux = 1.0f / ray.d.x;
        if (aux >= 0) { tmin.x = (a.x - ray.o.x) * aux; tmax.x = (b.x - ray.o.x) * aux; }
        else          { tmin.x = (b.x - ray.o.x) * aux; tmax.x = (a.x - ray.o.x) * aux; }
        aux = 1.0f / ray.d.y;
        if (aux >= 0) { tmin.y = (a.y - ray.o.y) * aux; tmax.y = (b.y - ray.o.y) * aux; }
        else          { tmin.y = (b.y - ray.o.y) * aux; tmax.y = (a.y - ray.o.y) * aux; }
        aux = 1.0f / ray.d.z;
        if (aux >= 0) { tmin.z = (a.z - ray.o.z) * aux; tmax.z = (b.z - ray.o.z) * aux; }
        else          { tmin.z = (b.z - ray.o.z) * aux; tmax.z = (a.z - ray.o.z) * aux; }
        V3 faceIn, faceOut;
        if (tmin.x > tmin.y) { tIn = tmin.x; faceIn = V3(tmin.x < 0 ? -1 : 1, 0, 0); }
        else                 { tIn = tmin.y; faceIn = V3(0, tmin.y < 0 ? -1 : 1, 0); }
        if (tmin.z > tIn)    { tIn = tmin.z; faceIn = V3(0, 0, tmin.z < 0 ? -1 : 1); }
        if (tmax.x < tmax.y) { tOut = tmax.x; faceOut = V3(tmax.x < 0 ? -1 : 1, 0, 0); }
        else                 { tOut = tmax.y; faceOut = V3(0, tmax.y < 0 ? -1 : 1, 0); }
        if (tmax.z < tOut)   { tOut = tmax.z; faceOut = V3(0, 0, tmax.z < 0 ? -1 : 1); }
        if (tIn < tOut && tOut > kEps) {
            if (tIn > kEps) { t = tIn; nrm = faceIn; }                   // SURVEY Q9 side effect
            else            { t = tOut; nrm = faceOut; }
            return true;
        }
        return false;
    }
    bool hit_plane(const RayO& r, float& t) const {                      // RT/scene.cpp:122-147
        g_ctr.plane_tests++;
        float denominator = dot(a, r.d);
        if (fabsf(denominator) < kEps) return false;
        float numerator = dot(a, r.o) + b.x;
        float taux = -(numerator / denominator);
        if (taux <= 0) return false;
        t = taux;
        return true;
    }
    V3 normal_at(const V3& p) const {
        switch (type) {
        case P3O_SPHERE: { V3 n = p - a; return normalize(n); }          // RT/scene.cpp:174-178
        case P3O_TRIANGLE: return nrm;                                   // RT/scene.cpp:46-49
        case P3O_BOX: return nrm;                                        // RT/scene.cpp:280-283
        default: return a;                                               // RT/scene.cpp:143-146
        }
    }
};

Prim make_sphere(const V3& c, float r) {
    Prim p; p.type = P3O_SPHERE; p.material = -1; p.a = c; p.b = V3(r, 0, 0);
    float raw[12] = {c.x, c.y, c.z, r, 0, 0, 0, 0, 0, 0, 0, 0};
    memcpy(p.raw, raw, sizeof raw);
    return p;
}
Prim make_triangle(const V3& P0, const V3& P1, const V3& P2) {           // RT/scene.cpp:10-40
    Prim p; p.type = P3O_TRIANGLE; p.material = -1; p.a = P0; p.b = P1; p.c = P2;
    V3 V = P1 - P0, W = P2 - P0;
    V3 n(0, 0, 0);
    n.x = (V.y * W.z) - (V.z * W.y);
    n.y = (V.z * W.x) - (V.x * W.z);
    n.z = (V.x * W.y) - (V.y * W.x);
    normalize(n);
    p.nrm = n;
    float x0 = std::min(std::min(P0.x, P1.x), P2.x), y0 = std::min(std::min(P0.y, P1.y), P2.y),
          z0 = std::min(std::min(P0.z, P1.z), P2.z);
    float x1 = std::max(std::max(P0.x, P1.x), P2.x), y1 = std::max(std::max(P0.y, P1.y), P2.y),
          z1 = std::max(std::max(P0.z, P1.z), P2.z);
    p.bmin = V3(x0 - kEps, y0 - kEps, z0 - kEps);
    p.bmax = V3(x1 + kEps, y1 + kEps, z1 + kEps);
    float raw[12] = {P0.x, P0.y, P0.z, P1.x, P1.y, P1.z, P2.x, P2.y, P2.z, 0, 0, 0};
    memcpy(p.raw, raw, sizeof raw);
    return p;
}
Prim make_box(const V3& mn, const V3& mx) {
    Prim p; p.type = P3O_BOX; p.material = -1; p.a = mn; p.b = mx;
    float raw[12] = {mn.x, mn.y, mn.z, mx.x, mx.y, mx.z, 0, 0, 0, 0, 0, 0};
    memcpy(p.raw, raw, sizeof raw);
    return p;
}
Prim make_plane(const V3& P0, const V3& P1, const V3& P2) {              // RT/scene.cpp:95-115
    Prim p; p.type = P3O_PLANE; p.material = -1;
    V3 v21 = P1 - P0, v31 = P2 - P0;
    V3 PN = cross(v21, v31);
    float D = 0.0f;                       // reference leaves D uninitialised when degenerate
    if (vlen(PN) == 0.0) { fprintf(stderr, "DEGENERATED PLANE!\n"); }
    else { normalize(PN); D = dot(PN, P0) * (-1); }
    p.a = PN; p.b = V3(D, 0, 0);
    float raw[12] = {P0.x, P0.y, P0.z, P1.x, P1.y, P1.z, P2.x, P2.y, P2.z, 0, 0, 0};
    memcpy(p.raw, raw, sizeof raw);
    return p;
}

// ------------------------------------------------------------------ RT/camera.h
struct Cam {
    V3 eye, at, up, u, v, n;
    float fovy, plane_dist, focal_ratio, aperture, aperture_ratio, w, h, hither;
    int res_x, res_y;
    void setup(V3 from, V3 At, V3 Up, float angle, float hith, int rx, int ry,
               float ap_ratio, float foc_ratio) {                        // RT/camera.h:35-73
        eye = from; at = At; up = Up; fovy = angle; hither = hith; res_x = rx; res_y = ry;
        focal_ratio = foc_ratio; aperture_ratio = ap_ratio;
        n = eye - at;
        plane_dist = vlen(n);
        n = n / plane_dist;
        u = cross(up, n);
        u = u / vlen(u);
        v = cross(n, u);
        normalize(n);      // "ze = n.normalize()" mutates n after u,v were formed (RT/camera.h:55)
        h = 2 * plane_dist * tanf((kPi * angle / 180) / 2.0f);
        w = ((float)res_x / res_y) * h;
        aperture = ap_ratio * (w / res_x);
    }
    RayO primary(const V3& ps) const {                                   // RT/camera.h:91-108
        V3 vX = u * w * (ps.x / res_x - 0.5f);
        V3 vY = v * h * (ps.y / res_y - 0.5f);
        V3 vZ = n * -plane_dist;
        V3 dir = vX + vY + vZ;
        normalize(dir);
        RayO r; r.o = eye; r.d = dir;
        return r;
    }
    RayO primary_lens(const V3& ls, const V3& ps) const {                // RT/camera.h:110-127
        V3 p(w * (ps.x / res_x - 0.5f) * focal_ratio, h * (ps.y / res_y - 0.5f) * focal_ratio, 0);
        V3 dir = u * (p.x - ls.x) + v * (p.y - ls.y) + n * (-focal_ratio * plane_dist);
        normalize(dir);
        RayO r; r.o = eye + (u * ls.x) + (v * ls.y); r.d = dir;
        return r;
    }
};

// ------------------------------------------------------------------ RT/bvh.cpp restated
struct RefBVH {
    struct Node { Box3 bb; bool leaf; unsigned n_objs; unsigned index; };
    struct Item { int node; float t; };
    int threshold = 2;                                                   // RT/rayAccelerator.h:72
    std::vector<int> order;          // permuted object indices ("objects" vector of the BVH)
    std::vector<Node> nodes;
    std::vector<Item> hit_stack;     // member stack, deliberately persistent
    std::vector<Prim>* prims = nullptr;

    float cen(int slot, int axis) const { return (*prims)[order[slot]].centroid().get(axis); }

    void build(std::vector<Prim>& ps) {                                  // RT/bvh.cpp:28-46
        prims = &ps; order.clear(); nodes.clear(); hit_stack.clear();
        Box3 world(V3(FLT_MAX, FLT_MAX, FLT_MAX), V3(-FLT_MAX, -FLT_MAX, -FLT_MAX));
        for (size_t i = 0; i < ps.size(); i++) { world.extend(ps[i].bbox()); order.push_back((int)i); }
        world.mn.x -= kEps; world.mn.y -= kEps; world.mn.z -= kEps;
        world.mx.x += kEps; world.mx.y += kEps; world.mx.z += kEps;
        Node root; root.bb = world; root.leaf = false; root.n_objs = 0; root.index = 0;
        nodes.push_back(root);
        rec(0, (int)order.size(), 0);
    }
    void rec(int left, int right, int node) {                            // RT/bvh.cpp:48-158
        if ((right - left) <= threshold) {
            nodes[node].leaf = true; nodes[node].index = left; nodes[node].n_objs = right - left;
            return;
        }
        Box3 nb = nodes[node].bb;
        int axis;
        V3 dist = nb.mx - nb.mn;
        if (dist.x >= dist.y && dist.x >= dist.z) axis = 0;
        else if (dist.y >= dist.x && dist.y >= dist.z) axis = 1;
        else axis = 2;
        std::vector<Prim>& P = *prims;
        // same libstdc++ std::sort + same comparator + same input order => same permutation
        std::sort(order.begin() + left, order.begin() + right, [&](int ia, int ib) {
            float ca = P[ia].bbox().centroid().get(axis);
            float cb = P[ib].bbox().centroid().get(axis);
            return ca < cb;
        });
        float mid = (nb.mx.get(axis) + nb.mn.get(axis)) * 0.5f;
        int split;
        if (cen(left, axis) > mid || cen(right - 1, axis) <= mid) {
            mid = 0.0f;
            for (int i = left; i < right; i++) mid += cen(i, axis);
            mid /= (right - left);
        }
        if (cen(left, axis) > mid || cen(right - 1, axis) <= mid) {
            split = left + threshold;
        } else {
            int start = left, end = right, mi;
            while (start != end && start < end) {
                mi = start + (end - start) / 2;
                float mc = cen(mi, axis);
                if (mc <= mid) { start = mi + 1; continue; }
                else if (mc > mid) { end = mi; continue; }
                break;
            }
            for (split = start; split < end; split++)
                if (cen(split, axis) > mid) break;
        }
        Box3 lb(V3(FLT_MAX, FLT_MAX, FLT_MAX), V3(-FLT_MAX, -FLT_MAX, -FLT_MAX));
        Box3 rb = lb;
        for (int j = left; j < split; j++) lb.extend(P[order[j]].bbox());
        for (int j = split; j < right; j++) rb.extend(P[order[j]].bbox());
        Node ln; ln.bb = lb; ln.leaf = false; ln.n_objs = 0; ln.index = 0;
        Node rn = ln; rn.bb = rb;
        nodes[node].leaf = false; nodes[node].index = (unsigned)nodes.size();
        int li = (int)nodes.size();
        nodes.push_back(ln); nodes.push_back(rn);
        rec(left, split, li);
        rec(split, right, li + 1);
    }
    // RT/bvh.cpp:252-346.  hit_obj < 0 means "NULL"; returns the reference's bool.
    bool closest(const RayO& ray, int& hit_obj, V3& hit_point) {
        float tmp, tmin = FLT_MAX;
        bool hit = false;
        int cur = 0;
        if (!nodes[cur].bb.hit(ray, tmp)) return false;
        std::vector<Prim>& P = *prims;
        while (true) {
            if (nodes[cur].leaf) {
                for (unsigned i = nodes[cur].index; i < nodes[cur].index + nodes[cur].n_objs; i++) {
                    int oi = order[i];
                    if (P[oi].hit(ray, tmp) && tmp < tmin) { tmin = tmp; hit_obj = oi; }
                }
                hit = true;          // "hit_obj != NULL" tests the out-pointer: always true (Q3)
            } else {
                int l = nodes[cur].index, r = l + 1;
                float ld, rd;
                bool lh = nodes[l].bb.hit(ray, ld);
                bool rh = nodes[r].bb.hit(ray, rd);
                if (nodes[l].bb.inside(ray.o)) ld = 0;
                if (nodes[r].bb.inside(ray.o)) rd = 0;
                if (lh && ld > tmin) lh = false;
                if (rh && rd > tmin) rh = false;
                if (lh && rh) {
                    if (ld < rd) { cur = l; hit_stack.push_back(Item{r, rd}); }
                    else         { cur = r; hit_stack.push_back(Item{l, ld}); }
                    continue;
                } else if (lh) { cur = l; continue; }
                else if (rh) { cur = r; continue; }
            }
            bool newNode = false;
            while (!hit_stack.empty()) {
                Item it = hit_stack.back(); hit_stack.pop_back();
                if (it.t < tmin) { cur = it.node; newNode = true; break; }
            }
            if (!newNode) break;
        }
        if (hit) { hit_point = ray.o + ray.d * tmin; return true; }
        return false;
    }
    // RT/bvh.cpp:348-416.  Mutates the ray direction like the reference.
    bool shadow(RayO& ray) {
        float tmp;
        double length = vlen(ray.d);
        normalize(ray.d);
        int cur = 0;
        if (!nodes[cur].bb.hit(ray, tmp)) return false;
        std::vector<Prim>& P = *prims;
        while (true) {
            if (nodes[cur].leaf) {
                for (unsigned i = nodes[cur].index; i < nodes[cur].index + nodes[cur].n_objs; i++) {
                    if (P[order[i]].hit(ray, tmp) && tmp < length) return true;   // stack left dirty (Q4)
                }
            } else {
                int l = nodes[cur].index, r = l + 1;
                float ld, rd;
                bool lh = nodes[l].bb.hit(ray, ld);
                bool rh = nodes[r].bb.hit(ray, rd);
                if (lh && rh) {
                    if (ld < rd) { cur = l; hit_stack.push_back(Item{r, rd}); }
                    else         { cur = r; hit_stack.push_back(Item{l, ld}); }
                    continue;
                } else if (lh) { cur = l; continue; }
                else if (rh) { cur = r; continue; }
            }
            if (hit_stack.empty()) break;
            Item it = hit_stack.back(); hit_stack.pop_back();
            cur = it.node;
        }
        return false;
    }
};

// ------------------------------------------------------------------ RT/grid.cpp restated
inline double dclamp(const double x, const double mn, const double mx) {   // RT/maths.h:50-53
    return (x < mn ? mn : (x > mx ? mx : x));
}
struct RefGrid {
    std::vector<std::vector<int> > cells;
    int nx = 0, ny = 0, nz = 0;
    float m = 2.0f;                                                      // RT/rayAccelerator.h:29
    Box3 bbox;
    std::vector<Prim>* prims = nullptr;

    void build(std::vector<Prim>& ps) {                                  // RT/grid.cpp:30-98
        prims = &ps; cells.clear();
        Box3 gb(V3(FLT_MAX, FLT_MAX, FLT_MAX), V3(-FLT_MAX, -FLT_MAX, -FLT_MAX));
        for (size_t i = 0; i < ps.size(); i++) gb.extend(ps[i].bbox());
        gb.mn.x -= kEps; gb.mn.y -= kEps; gb.mn.z -= kEps;
        gb.mx.x += kEps; gb.mx.y += kEps; gb.mx.z += kEps;
        bbox = gb;
        double wx = bbox.mx.x - bbox.mn.x;
        double wy = bbox.mx.y - bbox.mn.y;
        double wz = bbox.mx.z - bbox.mn.z;
        double s = pow((int)ps.size() / (wx * wy * wz), 0.3333333);
        nx = m * wx * s + 1;
        ny = m * wy * s + 1;
        nz = m * wz * s + 1;
        int cellCount = nx * ny * nz;
        cells.resize(cellCount);
        for (size_t oi = 0; oi < ps.size(); oi++) {
            Box3 obb = ps[oi].bbox();
            int ixmin = dclamp((obb.mn.x - bbox.mn.x) * nx / (bbox.mx.x - bbox.mn.x), 0, nx - 1);
            int iymin = dclamp((obb.mn.y - bbox.mn.y) * ny / (bbox.mx.y - bbox.mn.y), 0, ny - 1);
            int izmin = dclamp((obb.mn.z - bbox.mn.z) * nz / (bbox.mx.z - bbox.mn.z), 0, nz - 1);
            int ixmax = dclamp((obb.mx.x - bbox.mn.x) * nx / (bbox.mx.x - bbox.mn.x), 0, nx - 1);
            int iymax = dclamp((obb.mx.y - bbox.mn.y) * ny / (bbox.mx.y - bbox.mn.y), 0, ny - 1);
            int izmax = dclamp((obb.mx.z - bbox.mn.z) * nz / (bbox.mx.z - bbox.mn.z), 0, nz - 1);
            for (int iz = izmin; iz <= izmax; iz++)
                for (int iy = iymin; iy <= iymax; iy++)
                    for (int ix = ixmin; ix <= ixmax; ix++)
                        cells[ix + nx * iy + nx * ny * iz].push_back((int)oi);
        }
    }
    bool init(const RayO& ray, int& ix, int& iy, int& iz, double& dtx, double& dty, double& dtz,
              double& tx_next, double& ty_next, double& tz_next, int& ix_step, int& iy_step,
              int& iz_step, int& ix_stop, int& iy_stop, int& iz_stop) const {   // RT/grid.cpp:101-245
        float t0, t1;
        float ox = ray.o.x, oy = ray.o.y, oz = ray.o.z;
        float dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
        float x0 = bbox.mn.x, y0 = bbox.mn.y, z0 = bbox.mn.z;
        float x1 = bbox.mx.x, y1 = bbox.mx.y, z1 = bbox.mx.z;
        float tx_min, ty_min, tz_min, tx_max, ty_max, tz_max;
        float a = 1.0 / dx;
        if (a >= 0) { tx_min = (x0 - ox) * a; tx_max = (x1 - ox) * a; }
        else        { tx_min = (x1 - ox) * a; tx_max = (x0 - ox) * a; }
        float b = 1.0 / dy;
        if (b >= 0) { ty_min = (y0 - oy) * b; ty_max = (y1 - oy) * b; }
        else        { ty_min = (y1 - oy) * b; ty_max = (y0 - oy) * b; }
        float c = 1.0 / dz;
        if (c >= 0) { tz_min = (z0 - oz) * c; tz_max = (z1 - oz) * c; }
        else        { tz_min = (z1 - oz) * c; tz_max = (z0 - oz) * c; }
        if (tx_min > ty_min) t0 = tx_min; else t0 = ty_min;
        if (tz_min > t0) t0 = tz_min;
        if (tx_max < ty_max) t1 = tx_max; else t1 = ty_max;
        if (tz_max < t1) t1 = tz_max;
        if (t0 > t1 || t1 < 0) return false;
        if (bbox.inside(ray.o)) {
            ix = dclamp((ox - x0) * nx / (x1 - x0), 0, nx - 1);
            iy = dclamp((oy - y0) * ny / (y1 - y0), 0, ny - 1);
            iz = dclamp((oz - z0) * nz / (z1 - z0), 0, nz - 1);
        } else {
            V3 p = ray.o + ray.d * t0;
            ix = dclamp((p.x - x0) * nx / (x1 - x0), 0, nx - 1);
            iy = dclamp((p.y - y0) * ny / (y1 - y0), 0, ny - 1);
            iz = dclamp((p.z - z0) * nz / (z1 - z0), 0, nz - 1);
        }
        dtx = (tx_max - tx_min) / nx;
        dty = (ty_max - ty_min) / ny;
        dtz = (tz_max - tz_min) / nz;
        if (dx > 0) { tx_next = tx_min + (ix + 1) * dtx; ix_step = +1; ix_stop = nx; }
        else        { tx_next = tx_min + (nx - ix) * dtx; ix_step = -1; ix_stop = -1; }
        if (dx == 0.0) tx_next = FLT_MAX;
        if (dy > 0) { ty_next = ty_min + (iy + 1) * dty; iy_step = +1; iy_stop = ny; }
        else        { ty_next = ty_min + (ny - iy) * dty; iy_step = -1; iy_stop = -1; }
        if (dy == 0.0) ty_next = FLT_MAX;
        if (dz > 0) { tz_next = tz_min + (iz + 1) * dtz; iz_step = +1; iz_stop = nz; }
        else        { tz_next = tz_min + (nz - iz) * dtz; iz_step = -1; iz_stop = -1; }
        if (dz == 0.0) tz_next = FLT_MAX;
        return true;
    }
    bool closest(const RayO& ray, int& hit_obj, V3& hit_point) {          // RT/grid.cpp:248-310
        int ix, iy, iz, ix_step, iy_step, iz_step, ix_stop, iy_stop, iz_stop;
        double tx_next, ty_next, tz_next, dtx, dty, dtz;
        if (!init(ray, ix, iy, iz, dtx, dty, dtz, tx_next, ty_next, tz_next, ix_step, iy_step,
                  iz_step, ix_stop, iy_stop, iz_stop)) return false;
        std::vector<Prim>& P = *prims;
        float closestDistance; int closestObj = -1; float distance;
        while (true) {
            const std::vector<int>& objs = cells[ix + nx * iy + nx * ny * iz];
            closestDistance = FLT_MAX;
            for (int oi : objs)
                if (P[oi].hit(ray, distance) && distance < closestDistance) {
                    closestDistance = distance; closestObj = oi;
                }
            if (tx_next < ty_next && tx_next < tz_next) {
                if (closestDistance < tx_next) { hit_obj = closestObj; hit_point = ray.o + ray.d * closestDistance; return true; }
                tx_next += dtx; ix += ix_step; if (ix == ix_stop) return false;
            } else if (ty_next < tz_next) {
                if (closestDistance < ty_next) { hit_obj = closestObj; hit_point = ray.o + ray.d * closestDistance; return true; }
                ty_next += dty; iy += iy_step; if (iy == iy_stop) return false;
            } else {
                if (closestDistance < tz_next) { hit_obj = closestObj; hit_point = ray.o + ray.d * closestDistance; return true; }
                tz_next += dtz; iz += iz_step; if (iz == iz_stop) return false;
            }
        }
    }
    bool shadow(RayO& ray) {                                             // RT/grid.cpp:313-361
        double length = vlen(ray.d);
        normalize(ray.d);
        int ix, iy, iz, ix_step, iy_step, iz_step, ix_stop, iy_stop, iz_stop;
        double tx_next, ty_next, tz_next, dtx, dty, dtz;
        if (!init(ray, ix, iy, iz, dtx, dty, dtz, tx_next, ty_next, tz_next, ix_step, iy_step,
                  iz_step, ix_stop, iy_stop, iz_stop)) return true;      // "miss the box = shadowed"
        std::vector<Prim>& P = *prims;
        float distance;
        while (true) {
            const std::vector<int>& objs = cells[ix + nx * iy + nx * ny * iz];
            for (int oi : objs)
                if (P[oi].hit(ray, distance) && distance < length) return true;
            if (tx_next < ty_next && tx_next < tz_next) {
                tx_next += dtx; ix += ix_step; if (ix == ix_stop) return false;
            } else if (ty_next < tz_next) {
                ty_next += dty; iy += iy_step; if (iy == iy_stop) return false;
            } else {
                tz_next += dtz; iz += iz_step; if (iz == iz_stop) return false;
            }
        }
    }
};

}  // namespace

// ------------------------------------------------------------------ scene container
struct p3o_scene {
    std::vector<Prim> prims;
    std::vector<Mat> mats;
    std::vector<LightO> lights;
    Cam cam;
    bool has_cam = false;
    Col bg;
    unsigned spp = 0;
    int accel = 0;
    bool parse_ok = true;
    RefBVH bvh; bool bvh_built = false;
    RefGrid grid; bool grid_built = false;
    // Scene::skybox_img (RT/scene.h:190-195): right, left, top, bottom, front, back
    struct Face { std::vector<uint8_t> img; unsigned resX = 0, resY = 0, BPP = 3; } skybox_img[6];
};

// Scene::GetSkyboxColor, RT/scene.cpp:383-461 (dead code in the reference: nothing calls it).  Same
// expressions in the same types: "double invMa = 1 / ma" is a FLOAT division widened afterwards; s and t are formed in
// double and rounded once; the two clamping lines (:450,452) are expression statements without effect; u8tofloat
// divides by 255.99f (RT/maths.h:120-123).
static Col skybox_color(const p3o_scene* sc, const V3& dir) {
    const V3 c = dir;                                   // "skybox indexed by the ray direction"
    float ma; int side;
    if (fabs(c.x) > fabs(c.y)) { ma = fabs(c.x); side = c.x >= 0 ? 1 : 0; }       // LEFT at X = +1, RIGHT at X = -1
    else { ma = fabs(c.y); side = c.y >= 0 ? 2 : 3; }                             // TOP / BOTTOM
    if (fabs(c.z) > ma) { ma = fabs(c.z); side = c.z >= 0 ? 4 : 5; }              // FRONT / BACK
    float scx = 0, tcx = 0;
    switch (side) {
    case 0: scx = -c.z; tcx = c.y; break;
    case 1: scx = c.z; tcx = c.y; break;
    case 2: scx = -c.x; tcx = -c.z; break;
    case 3: scx = -c.x; tcx = c.z; break;
    case 4: scx = -c.x; tcx = c.y; break;
    case 5: scx = c.x; tcx = c.y; break;
    }
    double invMa = 1 / ma;
    float s = (scx * invMa + 1) / 2;
    float t = (tcx * invMa + 1) / 2;
    const p3o_scene::Face& F = sc->skybox_img[side];
    unsigned width = F.resX, height = F.resY, bytesperpixel = F.BPP;
    unsigned xp = int((width - 1) * s);
    unsigned yp = int((height - 1) * t);
    size_t at = ((size_t)yp * width + xp) * bytesperpixel;
    if (at + 2 >= F.img.size()) return Col(0, 0, 0);    // (a direction with NaNs indexes outside the image: the reference would read wild memory)
    auto u8tofloat = [](uint8_t x) { return (float)(x / 255.99f); };
    return Col(u8tofloat(F.img[at]), u8tofloat(F.img[at + 1]), u8tofloat(F.img[at + 2]));
}

namespace {

// ---- tracer state: one per thread (the reference has one, globally)
V3 rnd_unit_sphere() {                                                   // RT/maths.h:98-104
    V3 p;
    do {
        p = V3(rand_float(), rand_float(), rand_float()) * 2 - V3(1.0, 1.0, 1.0);
    } while (dot(p, p) >= 1.0);
    return p;
}

struct Tracer {
    p3o_scene* sc;
    std::vector<Prim>* prims;     // thread-private copy when threads>1 (box normal side effects)
    RefBVH* bvh;
    RefGrid* grid;
    int accel;
    int max_depth;
    bool break_fixed;
    int32_t last_primary_hit;
    // distribution-ray-tracing switches (compile-time false in the reference, RT/main.cpp:40-45)
    bool soft_shadow = false, fuzzy_reflection = false;
    bool skybox = false;                           // a miss returns Scene::GetSkyboxColor(ray) instead of the background colour
    int spp = 0;                                   // globalSamplesPerPixel; ANTI_ALIASING == spp > 0 (RT/main.cpp:943)
    int offset_for_shadowx = 0, offset_for_shadowy = 0;                  // RT/main.cpp:101,779-780

    Prim& obj(int i) { g_ctr.get_object++; return (*prims)[i]; }         // RT/scene.cpp:307-312

    // RT/main.cpp:471-526
    void processLight(V3& L, const Col& lightColor, Col& color, const Mat& material,
                      const RayO& ray, const V3& precise, const V3& normal) {
        float closest_t = FLT_MAX;
        bool insideShadow = false;
        if (dot(L, normal) > 0) {
            RayO shadowRay; shadowRay.o = precise; shadowRay.d = L;
            g_ctr.rays++; g_ctr.shadow_queries++;
            int n = (int)prims->size();
            switch (accel) {
            case 1: if (grid->shadow(shadowRay)) insideShadow = true; break;
            case 2: if (bvh->shadow(shadowRay)) insideShadow = true; break;
            default:   // NONE: un-normalised direction, no distance bound
                for (int i = 0; i < n; i++)
                    if (obj(i).hit(shadowRay, closest_t)) { insideShadow = true; break; }
                break;
            }
        }
        if (!insideShadow) {
            normalize(L);
            V3 H = L + (ray.d * -1);
            normalize(H);
            float VdotN = dot(H, normal);
            float max1 = std::max(0.0f, dot(normal, L));
            float max2 = std::max(0.0f, VdotN);
            Col diff = (lightColor * material.diff) * max1;
            Col spec = (lightColor * material.spec) * powf(max2, material.shine);
            color = color + ((diff * material.kd) + (spec * material.ks * 0.4f));
        }
    }

    // RT/main.cpp:530-721
    Col rayTracing(RayO ray, int depth, float ior_1, bool primary) {
        g_ctr.closest_queries++;
        int n = (int)prims->size();
        float closest_t = FLT_MAX;
        float t = FLT_MAX;
        int closest = -1;
        V3 hit_point;
        Col color(0.0f, 0.0f, 0.0f);
        bool brute = true;
        if (accel == 1) {
            if (!grid->closest(ray, closest, hit_point)) closest = -1;
            brute = false;
        } else if (accel == 2) {
            if (!bvh->closest(ray, closest, hit_point)) closest = -1;
            brute = !break_fixed;           // missing "break" => falls into default
        }
        if (brute) {
            for (int i = 0; i < n; i++) {
                Prim& o = obj(i);
                if (o.hit(ray, t) && t < closest_t) { closest_t = t; closest = i; }
            }
            if (closest >= 0) hit_point = ray.o + ray.d * closest_t;
        }
        if (primary) last_primary_hit = closest;
        if (closest < 0) return skybox ? skybox_color(sc, ray.d) : sc->bg;   // SURVEY Q8 (the reference: always bgColor)

        Prim& O = (*prims)[closest];
        const Mat& M = sc->mats[O.material];
        V3 normal = O.normal_at(hit_point); normalize(normal);
        V3 precise = hit_point + normal * kEps;
        normal = O.normal_at(precise); normalize(normal);
        V3 V = ray.d * (-1);

        for (size_t i = 0; i < sc->lights.size(); ++i) {
            const auto& light = sc->lights[i];
            if (soft_shadow) {                                           // RT/main.cpp:598-625
                float shadow = 0.5f;
                if (spp == 0) {
                    float distance = shadow / 4;
                    float cur_x = light.pos.x - distance * shadow * 4;
                    float cur_y = light.pos.y - distance * shadow * 4;
                    Col avg_col = light.col / (4 * 4);
                    for (int a = 0; a < 4; a++) {
                        for (int b = 0; b < 4; b++) {
                            V3 position(cur_x, cur_y, light.pos.z);
                            V3 L = position - hit_point;
                            processLight(L, avg_col, color, M, ray, precise, normal);
                            cur_x += distance;
                        }
                        cur_y += distance;
                        cur_x = light.pos.x - distance * shadow * 4;
                    }
                } else {
                    // same expression shape as the reference (g++ evaluates the constructor
                    // arguments right to left, like sampleUnitDisk below)
                    V3 position(light.pos.x + shadow * ((offset_for_shadowx + rand_float()) / spp),
                                light.pos.y + shadow * ((offset_for_shadowy + rand_float()) / spp), light.pos.z);
                    V3 L = position - hit_point;
                    processLight(L, light.col, color, M, ray, precise, normal);
                }
            } else {
                V3 L = light.pos - hit_point;
                processLight(L, light.col, color, M, ray, precise, normal);
            }
        }
        if (depth >= max_depth) return clampc(color);

        Col reflection_color(0.0f, 0.0f, 0.0f), refraction_color(0.0f, 0.0f, 0.0f);
        bool inside = false;
        if (dot(ray.d, normal) > 0) { normal = normal * -1; inside = true; }

        if (M.refl > 0 && depth < max_depth) {
            V3 rdir = ray.d - (normal * dot(ray.d, normal) * 2);
            if (fuzzy_reflection) {                                      // RT/main.cpp:651-660
                V3 sphere_center = rdir + precise;
                V3 sphere_offset = sphere_center + rnd_unit_sphere() * 0.3f;
                V3 fuzzy = sphere_offset - precise;
                normalize(fuzzy);
                if (dot(fuzzy, normal) > 0) rdir = fuzzy;        // otherwise the un-normalised mirror direction
            } else {
                normalize(rdir);
            }
            RayO rr; rr.o = precise; rr.d = rdir;
            g_ctr.rays++;
            reflection_color = rayTracing(rr, depth + 1, ior_1, false);
        }
        float KR;
        if (M.T != 0) {
            float R0 = 1.0f, R1 = 1.0f;
            V3 viewnormal = normal * dot(normal, V);
            V3 viewtangent = viewnormal - V;
            float nn = inside ? ior_1 : ior_1 / M.ior;
            float cos_i = vlen(viewnormal);
            float sin_t = nn * vlen(viewtangent);
            float insqrt = 1 - pow(sin_t, 2);          // pow(float,int) -> double
            if (insqrt >= 0) {
                float cos_t = sqrtf(insqrt);
                V3 nc = normal * cos_t;
                V3 rfr = normalize(viewtangent) * sin_t + normalize(nc);
                V3 org = hit_point + rfr * 0.001f;
                RayO fr; fr.o = org; fr.d = rfr;
                g_ctr.rays++;
                float newIor = inside ? 1.0f : M.ior;
                refraction_color = rayTracing(fr, depth + 1, newIor, false);
                R0 = pow(fabsf((ior_1 * cos_i - newIor * cos_t) / (ior_1 * cos_i + newIor * cos_t)), 2);
                R1 = pow(fabsf((ior_1 * cos_t - newIor * cos_i) / (ior_1 * cos_i + newIor * cos_t)), 2);
            }
            KR = 1 / 2 * (R0 + R1);                    // integer 1/2 == 0
        } else {
            KR = M.ks;
        }
        color = color + (reflection_color * KR * M.spec + refraction_color * (1 - KR));
        return color;
    }
};

V3 sampleUnitDisk() {                                                    // RT/main.cpp:724-730
    V3 p;
    do {
        // same expression shape as the reference so g++ picks the same argument order
        p = V3(rand_float(), rand_float(), 0.0) * 2 - V3(1.0, 1.0, 0.0);
    } while (dot(p, p) >= 1.0);
    return p;
}

void add_ctr(p3o_counters& a, const p3o_counters& b) {
    a.rays += b.rays; a.closest_queries += b.closest_queries; a.shadow_queries += b.shadow_queries;
    a.aabb_tests += b.aabb_tests; a.sphere_tests += b.sphere_tests; a.tri_tests += b.tri_tests;
    a.box_tests += b.box_tests; a.plane_tests += b.plane_tests; a.get_object += b.get_object;
}

// RT/main.cpp:732-832, rows [y0,y1)
void render_rows(Tracer& T, int y0, int y1, unsigned spp, uint8_t* rgb8, float* rgb32f,
                 int32_t* hit_id) {
    p3o_scene* sc = T.sc;
    const Cam& cam = sc->cam;
    int W = cam.res_x;
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < W; x++) {
            Col color;
            V3 pixel;
            int32_t hid = -1;
            if (spp == 0) {
                pixel.x = x + 0.5f; pixel.y = y + 0.5f;
                RayO ray = cam.primary(pixel);
                g_ctr.rays++;
                color = clampc(T.rayTracing(ray, 1, 1.0, true));
                hid = T.last_primary_hit;
            } else {
                for (unsigned i = 0; i < spp; i++)
                    for (unsigned j = 0; j < spp; j++) {
                        T.offset_for_shadowx = (int)i; T.offset_for_shadowy = (int)j;
                        pixel.x = x + (i + rand_float()) / spp;
                        pixel.y = y + (j + rand_float()) / spp;
                        V3 lens = sampleUnitDisk() * cam.aperture;
                        RayO ray = cam.primary_lens(lens, pixel);
                        g_ctr.rays++;
                        color = color + clampc(T.rayTracing(ray, 1, 1.0, true));
                        if (i == 0 && j == 0) hid = T.last_primary_hit;
                    }
                color = color / (4 * 4);
            }
            size_t p = (size_t)y * W + x;
            if (rgb8) { rgb8[3 * p] = u8fromfloat(color.r); rgb8[3 * p + 1] = u8fromfloat(color.g); rgb8[3 * p + 2] = u8fromfloat(color.b); }
            if (rgb32f) { rgb32f[3 * p] = color.r; rgb32f[3 * p + 1] = color.g; rgb32f[3 * p + 2] = color.b; }
            if (hit_id) hit_id[p] = hid;
        }
    }
}

// ---- .p3f loader (RT/scene.cpp:476-675 grammar)
struct Tok {
    std::vector<std::string> t; size_t i = 0;
    bool more() const { return i < t.size(); }
    std::string next() { return i < t.size() ? t[i++] : std::string(); }
    float f() { return strtof(next().c_str(), nullptr); }
    double d() { return strtod(next().c_str(), nullptr); }
    long l() { return strtol(next().c_str(), nullptr, 10); }
    V3 v() { float a = f(), b = f(), c = f(); return V3(a, b, c); }
    Col c() { float a = f(), b = f(), c2 = f(); return Col(a, b, c2); }
};

bool load_p3f(p3o_scene* sc, const char* path) {
    std::ifstream file(path, std::ios::in);
    if (!file) return false;
    // tokenise by lines so '#' can drop the rest of its line exactly like file.ignore()
    Tok tk;
    std::string line;
    std::vector<std::vector<std::string> > lines;
    while (std::getline(file, line)) {
        std::istringstream is(line); std::string w; std::vector<std::string> ws;
        while (is >> w) ws.push_back(w);
        lines.push_back(ws);
    }
    // flatten but remember line boundaries for comment handling
    std::vector<size_t> tok_line;
    for (size_t li = 0; li < lines.size(); li++)
        for (auto& w : lines[li]) { tk.t.push_back(w); tok_line.push_back(li); }
    int material = -1;
    V3 from, at, up; float fov = 45, hither = 0.01f, ap = 0, foc = 1; int xres = 512, yres = 512;
    while (tk.more()) {
        size_t cmd_idx = tk.i;
        std::string cmd = tk.next();
        if (cmd == "accel") { sc->accel = (int)(unsigned)tk.l(); }
        else if (cmd == "spp") { sc->spp = (unsigned)tk.l(); }
        else if (cmd == "f") {
            Mat m;
            m.diff = tk.c(); double Kd = tk.d(); m.spec = tk.c();
            double Ks = tk.d(), Shine = tk.d(), T = tk.d(), ior = tk.d();
            m.kd = Kd; m.ks = Ks; m.shine = Shine; m.refl = Ks; m.T = T; m.ior = ior;   // RT/scene.h:30-32
            sc->mats.push_back(m); material = (int)sc->mats.size() - 1;
        } else if (cmd == "s") {
            V3 c = tk.v(); float r = tk.f();
            Prim p = make_sphere(c, r); p.material = material; sc->prims.push_back(p);
        } else if (cmd == "box") {
            V3 a = tk.v(), b = tk.v();
            Prim p = make_box(a, b); p.material = material; sc->prims.push_back(p);
        } else if (cmd == "p") {
            unsigned nv = (unsigned)tk.l();
            if (nv == 3) {
                V3 a = tk.v(), b = tk.v(), c = tk.v();
                Prim p = make_triangle(a, b, c); p.material = material; sc->prims.push_back(p);
            } else { fprintf(stderr, "Unsupported number of vertices.\n"); sc->parse_ok = false; break; }
        } else if (cmd == "mesh") {
            unsigned nv = (unsigned)tk.l(), nf = (unsigned)tk.l();
            std::vector<V3> vs(nv);
            for (unsigned i = 0; i < nv; i++) vs[i] = tk.v();
            for (unsigned i = 0; i < nf; i++) {
                unsigned P0 = (unsigned)tk.l(), P1 = (unsigned)tk.l(), P2 = (unsigned)tk.l();
                if (P0 > 0) { P0 -= 1; P1 -= 1; P2 -= 1; }
                else { P0 += nv; P1 += nv; P2 += nv; }
                Prim p = make_triangle(vs[P0], vs[P1], vs[P2]); p.material = material;
                sc->prims.push_back(p);
            }
        } else if (cmd == "pl") {
            V3 a = tk.v(), b = tk.v(), c = tk.v();
            Prim p = make_plane(a, b, c); p.material = material; sc->prims.push_back(p);
        } else if (cmd == "l") {
            LightO l; l.pos = tk.v(); l.col = tk.c(); sc->lights.push_back(l);
        } else if (cmd == "v") {
            tk.next(); from = tk.v();
            tk.next(); at = tk.v();
            tk.next(); up = tk.v();
            tk.next(); fov = tk.f();
            tk.next(); hither = tk.f();
            tk.next(); xres = (int)tk.l(); yres = (int)tk.l();
            tk.next(); ap = tk.f();
            tk.next(); foc = tk.f();
            sc->cam.setup(from, at, up, fov, hither, xres, yres, ap, foc);
            sc->has_cam = true;
        } else if (cmd == "bclr") { sc->bg = tk.c(); }
        else if (cmd == "env") { tk.next(); }
        else if (cmd[0] == '#') {
            size_t li = tok_line[cmd_idx];
            while (tk.i < tk.t.size() && tok_line[tk.i] == li) tk.i++;
        } else {
            fprintf(stderr, "unknown command '%s'.\n", cmd.c_str());
            sc->parse_ok = false;
            break;
        }
    }
    // objects declared before any "f" have an uninitialised material pointer in the
    // reference (UB); the oracle gives them Material()'s defaults.
    bool need_default = false;
    for (auto& p : sc->prims) if (p.material < 0) need_default = true;
    if (need_default) {
        sc->mats.push_back(Mat());
        for (auto& p : sc->prims) if (p.material < 0) p.material = (int)sc->mats.size() - 1;
    }
    return sc->has_cam;
}

void ensure_accel(p3o_scene* sc, int accel) {
    if (accel == 2 && !sc->bvh_built) { sc->bvh.build(sc->prims); sc->bvh_built = true; }
    if (accel == 1 && !sc->grid_built) { sc->grid.build(sc->prims); sc->grid_built = true; }
}

}  // namespace

// ------------------------------------------------------------------ C interface
extern "C" {

p3o_scene* p3o_scene_load(const char* path) {
    p3o_scene* sc = new p3o_scene();
    if (!load_p3f(sc, path)) { delete sc; return nullptr; }
    return sc;
}
void p3o_scene_free(p3o_scene* sc) { delete sc; }

void p3o_scene_info(const p3o_scene* sc, int32_t* out) {
    out[0] = (int32_t)sc->prims.size(); out[1] = (int32_t)sc->lights.size();
    out[2] = (int32_t)sc->mats.size(); out[3] = sc->cam.res_x; out[4] = sc->cam.res_y;
    out[5] = sc->accel; out[6] = (int32_t)sc->spp; out[7] = sc->parse_ok ? 1 : 0;
}
void p3o_scene_set_resolution(p3o_scene* sc, int32_t w, int32_t h) {
    Cam& c = sc->cam;
    c.setup(c.eye, c.at, c.up, c.fovy, c.hither, w, h, c.aperture_ratio, c.focal_ratio);
}
void p3o_scene_prims(const p3o_scene* sc, int32_t* type, float* data12, int32_t* material) {
    for (size_t i = 0; i < sc->prims.size(); i++) {
        type[i] = sc->prims[i].type; material[i] = sc->prims[i].material;
        memcpy(data12 + 12 * i, sc->prims[i].raw, 12 * sizeof(float));
    }
}
void p3o_scene_materials(const p3o_scene* sc, float* o) {
    for (size_t i = 0; i < sc->mats.size(); i++) {
        const Mat& m = sc->mats[i];
        float v[12] = {m.diff.r, m.diff.g, m.diff.b, m.kd, m.spec.r, m.spec.g, m.spec.b, m.ks,
                       m.shine, m.T, m.ior, m.refl};
        memcpy(o + 12 * i, v, sizeof v);
    }
}
void p3o_scene_lights(const p3o_scene* sc, float* o) {
    for (size_t i = 0; i < sc->lights.size(); i++) {
        const LightO& l = sc->lights[i];
        float v[6] = {l.pos.x, l.pos.y, l.pos.z, l.col.r, l.col.g, l.col.b};
        memcpy(o + 6 * i, v, sizeof v);
    }
}
void p3o_scene_bg(const p3o_scene* sc, float* o) { o[0] = sc->bg.r; o[1] = sc->bg.g; o[2] = sc->bg.b; }
void p3o_scene_camera(const p3o_scene* sc, float* o) {
    const Cam& c = sc->cam;
    float v[19] = {c.eye.x, c.eye.y, c.eye.z, c.u.x, c.u.y, c.u.z, c.v.x, c.v.y, c.v.z,
                   c.n.x, c.n.y, c.n.z, c.w, c.h, c.plane_dist, c.aperture, c.focal_ratio,
                   (float)c.res_x, (float)c.res_y};
    memcpy(o, v, sizeof v);
}

int p3o_render(p3o_scene* sc, const p3o_params* prm, uint8_t* rgb8, float* rgb32f,
               int32_t* hit_id, p3o_counters* ctr) {
    if (!sc || !prm) return -1;
    int accel = prm->accel < 0 ? sc->accel : prm->accel;
    unsigned spp = prm->spp < 0 ? sc->spp : (unsigned)prm->spp;
    int H = sc->cam.res_y;
    int y0 = prm->y0 > 0 ? prm->y0 : 0;
    int y1 = prm->y1 > 0 ? std::min(prm->y1, H) : H;
    ensure_accel(sc, accel);
    int threads = prm->threads > 1 ? prm->threads : 1;
    if (spp != 0 || prm->fuzzy_reflection) threads = 1;   // libc rand() stream is consumed in pixel order
    p3o_counters total; memset(&total, 0, sizeof total);
    if (threads == 1) {
        memset(&g_ctr, 0, sizeof g_ctr);
        srand(prm->seed);                                                 // RT/main.cpp:747 (every frame)
        Tracer T; T.sc = sc; T.prims = &sc->prims; T.bvh = &sc->bvh; T.grid = &sc->grid;
        T.accel = accel; T.max_depth = prm->max_depth; T.break_fixed = prm->break_fixed != 0;
        T.soft_shadow = prm->soft_shadow != 0; T.fuzzy_reflection = prm->fuzzy_reflection != 0; T.spp = (int)spp; T.skybox = prm->skybox != 0;
        T.last_primary_hit = -1;
        sc->bvh.hit_stack.clear();
        render_rows(T, y0, y1, spp, rgb8, rgb32f, hit_id);
        total = g_ctr;
    } else {
        std::vector<std::thread> pool;
        std::vector<p3o_counters> ctrs(threads);
        // interleaved 8-row blocks: image cost is very uneven (sky vs glass)
        for (int ti = 0; ti < threads; ti++) {
            pool.emplace_back([&, ti]() {
                memset(&g_ctr, 0, sizeof g_ctr);
                std::vector<Prim> priv = sc->prims;
                RefBVH bvh = sc->bvh; bvh.prims = &priv; bvh.hit_stack.clear();
                RefGrid grid = sc->grid; grid.prims = &priv;
                Tracer T; T.sc = sc; T.prims = &priv; T.bvh = &bvh; T.grid = &grid;
                T.accel = accel; T.max_depth = prm->max_depth; T.break_fixed = prm->break_fixed != 0;
        T.soft_shadow = prm->soft_shadow != 0; T.fuzzy_reflection = prm->fuzzy_reflection != 0; T.spp = (int)spp; T.skybox = prm->skybox != 0;
                T.last_primary_hit = -1;
                const int blk = 8;
                for (int b = y0 / blk; b * blk < y1; b++) {
                    if (b % threads != ti) continue;
                    int r0 = std::max(b * blk, y0), r1 = std::min((b + 1) * blk, y1);
                    render_rows(T, r0, r1, 0, rgb8, rgb32f, hit_id);
                }
                ctrs[ti] = g_ctr;
            });
        }
        for (auto& th : pool) th.join();
        for (auto& c : ctrs) add_ctr(total, c);
    }
    if (ctr) *ctr = total;
    return 0;
}

void p3o_scene_set_skybox(p3o_scene* sc, const uint8_t* const faces[6], const uint32_t* res_x, const uint32_t* res_y,
                          const uint32_t* bytes_per_pixel) {
    for (int i = 0; i < 6; i++) {
        p3o_scene::Face& F = sc->skybox_img[i];
        F.resX = res_x[i]; F.resY = res_y[i]; F.BPP = bytes_per_pixel[i];
        F.img.assign(faces[i], faces[i] + (size_t)F.resX * F.resY * F.BPP);
    }
}
void p3o_skybox_color(const p3o_scene* sc, const float* d, float* rgb3) {
    Col c = skybox_color(sc, V3(d[0], d[1], d[2]));
    rgb3[0] = c.r; rgb3[1] = c.g; rgb3[2] = c.b;
}

// one rayTracing(ray, 1, 1.0) call (RT/main.cpp:530) on an arbitrary ray, for the shading KATs
void p3o_trace(p3o_scene* sc, int accel, int max_depth, int soft_shadow, const float* o, const float* d,
               float* rgb3) {
    ensure_accel(sc, accel);
    memset(&g_ctr, 0, sizeof g_ctr);
    Tracer T; T.sc = sc; T.prims = &sc->prims; T.bvh = &sc->bvh; T.grid = &sc->grid;
    T.accel = accel; T.max_depth = max_depth; T.break_fixed = false;
    T.soft_shadow = soft_shadow != 0; T.fuzzy_reflection = false; T.spp = 0;
    T.last_primary_hit = -1;
    RayO r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    Col c = T.rayTracing(r, 1, 1.0, true);
    rgb3[0] = c.r; rgb3[1] = c.g; rgb3[2] = c.b;
}

// ---- KATs
static Prim prim_from12(int type, const float* d) {
    switch (type) {
    case P3O_SPHERE: return make_sphere(V3(d[0], d[1], d[2]), d[3]);
    case P3O_TRIANGLE: return make_triangle(V3(d[0], d[1], d[2]), V3(d[3], d[4], d[5]), V3(d[6], d[7], d[8]));
    case P3O_BOX: return make_box(V3(d[0], d[1], d[2]), V3(d[3], d[4], d[5]));
    default: return make_plane(V3(d[0], d[1], d[2]), V3(d[3], d[4], d[5]), V3(d[6], d[7], d[8]));
    }
}
int p3o_intersect(int type, const float* prim12, const float* o, const float* d, float* t_out,
                  float* nrm) {
    Prim p = prim_from12(type, prim12);
    RayO r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    float t = FLT_MAX;
    bool h = p.hit(r, t);
    if (h) {
        *t_out = t;
        if (nrm) {
            V3 hp = r.o + r.d * t;
            V3 n = p.normal_at(hp); normalize(n);
            nrm[0] = n.x; nrm[1] = n.y; nrm[2] = n.z;
        }
    }
    return h ? 1 : 0;
}
int p3o_aabb_intercepts(const float* mn, const float* mx, const float* o, const float* d, float* t_out) {
    Box3 b(V3(mn[0], mn[1], mn[2]), V3(mx[0], mx[1], mx[2]));
    RayO r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    float t = 0; bool h = b.hit(r, t); *t_out = t; return h ? 1 : 0;
}
void p3o_prim_bbox(int type, const float* prim12, float* mn, float* mx) {
    Prim p = prim_from12(type, prim12); Box3 b = p.bbox();
    mn[0] = b.mn.x; mn[1] = b.mn.y; mn[2] = b.mn.z; mx[0] = b.mx.x; mx[1] = b.mx.y; mx[2] = b.mx.z;
}
void p3o_normalize(float* v) { V3 a(v[0], v[1], v[2]); normalize(a); v[0] = a.x; v[1] = a.y; v[2] = a.z; }
void p3o_primary_ray(const p3o_scene* sc, float px, float py, float* o, float* d) {
    RayO r = sc->cam.primary(V3(px, py, 0));
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
}
void p3o_primary_ray_lens(const p3o_scene* sc, float lx, float ly, float px, float py, float* o, float* d) {
    RayO r = sc->cam.primary_lens(V3(lx, ly, 0), V3(px, py, 0));
    o[0] = r.o.x; o[1] = r.o.y; o[2] = r.o.z; d[0] = r.d.x; d[1] = r.d.y; d[2] = r.d.z;
}
uint8_t p3o_u8fromfloat(float x) { return u8fromfloat(x); }
void p3o_rand_floats(uint32_t seed, int32_t n, float* out) {
    srand(seed);
    for (int i = 0; i < n; i++) out[i] = rand_float();
}

int32_t p3o_refbvh_node_count(p3o_scene* sc) { ensure_accel(sc, 2); return (int32_t)sc->bvh.nodes.size(); }
void p3o_refbvh_dump(p3o_scene* sc, float* nodes8, int32_t* n_objs, int32_t* order) {
    ensure_accel(sc, 2);
    for (size_t i = 0; i < sc->bvh.nodes.size(); i++) {
        const RefBVH::Node& nd = sc->bvh.nodes[i];
        float v[8] = {nd.bb.mn.x, nd.bb.mn.y, nd.bb.mn.z, nd.bb.mx.x, nd.bb.mx.y, nd.bb.mx.z,
                      nd.leaf ? 1.0f : 0.0f, (float)nd.index};
        memcpy(nodes8 + 8 * i, v, sizeof v);
        n_objs[i] = nd.leaf ? (int32_t)nd.n_objs : 0;
    }
    for (size_t i = 0; i < sc->bvh.order.size(); i++) order[i] = sc->bvh.order[i];
}
int p3o_refbvh_shadow(p3o_scene* sc, const float* o, const float* d) {
    ensure_accel(sc, 2);
    RayO r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    return sc->bvh.shadow(r) ? 1 : 0;
}
int p3o_refbvh_closest(p3o_scene* sc, const float* o, const float* d, int32_t* obj, float* t) {
    ensure_accel(sc, 2);
    RayO r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    int h = -1; V3 hp;
    bool ok = sc->bvh.closest(r, h, hp);
    *obj = h;
    if (h >= 0) { float tt = FLT_MAX; sc->prims[h].hit(r, tt); *t = tt; }
    return ok ? 1 : 0;
}
void p3o_refgrid_dims(p3o_scene* sc, int32_t* nxyz, int32_t* cell_counts) {
    ensure_accel(sc, 1);
    nxyz[0] = sc->grid.nx; nxyz[1] = sc->grid.ny; nxyz[2] = sc->grid.nz;
    if (cell_counts)
        for (size_t i = 0; i < sc->grid.cells.size(); i++) cell_counts[i] = (int32_t)sc->grid.cells[i].size();
}
int p3o_refgrid_shadow(p3o_scene* sc, const float* o, const float* d) {
    ensure_accel(sc, 1);
    RayO r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    return sc->grid.shadow(r) ? 1 : 0;
}
int p3o_refgrid_closest(p3o_scene* sc, const float* o, const float* d, int32_t* obj, float* t) {
    ensure_accel(sc, 1);
    RayO r; r.o = V3(o[0], o[1], o[2]); r.d = V3(d[0], d[1], d[2]);
    int h = -1; V3 hp;
    bool ok = sc->grid.closest(r, h, hp);
    *obj = ok ? h : -1;
    if (ok && h >= 0) { float tt = FLT_MAX; sc->prims[h].hit(r, tt); *t = tt; }
    return ok ? 1 : 0;
}

}  // extern "C"
