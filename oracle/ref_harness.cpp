// ref_harness.cpp -- glue that exposes the REFERENCE'S OWN compiled objects to the tests.
//
// oracle/Makefile compiles RT/vector.cpp, RT/boundingBox.cpp, RT/bvh.cpp and RT/grid.cpp
// unchanged, in place from /root/reference, and links them with this file into
// oracle/_ref/libp3d_ref.so (git-ignored; travels to the GPU box as a built artefact).
// Nothing here restates reference logic: every function below forwards to the reference's
// Vector / AABB / Camera / BVH / Grid / maths.h / color.h code.  The only non-reference
// pieces are (a) `int Ray::nextId`, whose definition lives in the unbuildable RT/main.cpp:89,
// and (b) HObj, an Object subclass whose intersect/normal/bbox virtuals forward to the oracle's
// restated primitives (RT/scene.cpp, where the reference's bodies live, cannot be built here),
// so that the reference's BVH::Build/Traverse and Grid::Build/Traverse can be driven.
//
// TEST INFRASTRUCTURE ONLY.

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <queue>
#include <random>
#include <stack>
#include <vector>

// private members of Camera / BVH / Grid are read for structural comparison
#define private public
#define protected public
#define class struct      // BVH's nested node type is implicitly private (RT/rayAccelerator.h:39-52)
#include "rayAccelerator.h"   // pulls scene.h, camera.h, vector.h, ray.h, boundingBox.h, color.h, maths.h
#undef class
#undef private
#undef protected
#include "macros.h"

#include "p3d_oracle.h"

int Ray::nextId = 0;   // RT/main.cpp:89 (that file is not built)

namespace {

class HObj : public Object {
public:
    int id;
    int type;
    float d[12];
    Vector last_normal;
    bool intercepts(Ray& r, float& dist) override {
        float o3[3] = {r.origin.x, r.origin.y, r.origin.z};
        float d3[3] = {r.direction.x, r.direction.y, r.direction.z};
        float t = 0, n[3] = {0, 0, 0};
        if (p3o_intersect(type, d, o3, d3, &t, n)) {
            dist = t;
            last_normal = Vector(n[0], n[1], n[2]);
            return true;
        }
        return false;
    }
    Vector getNormal(Vector) override { return last_normal; }
    AABB GetBoundingBox() override {
        if (type == P3O_PLANE) return Object::GetBoundingBox();   // default box
        float mn[3], mx[3];
        p3o_prim_bbox(type, d, mn, mx);
        Vector a(mn[0], mn[1], mn[2]), b(mx[0], mx[1], mx[2]);
        return AABB(a, b);
    }
};

struct Accel {
    std::vector<HObj*> objs;
    BVH* bvh = nullptr;
    Grid* grid = nullptr;
    int index_of(Object* o) const { return static_cast<HObj*>(o)->id; }
};

Vector V(const float* p) { return Vector(p[0], p[1], p[2]); }
void put(float* o, Vector v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

}  // namespace

extern "C" {

// out: add3 sub3 cross3 scaled3(a*b.x) div3(a/b.x) dot len(a) normalized(a)3  = 20 floats
void ref_vec_ops(const float* a3, const float* b3, float* out20) {
    Vector a = V(a3), b = V(b3);
    put(out20 + 0, a + b);
    put(out20 + 3, a - b);
    put(out20 + 6, a % b);
    put(out20 + 9, a * b3[0]);
    put(out20 + 12, a / b3[0]);
    out20[15] = a * b;
    out20[16] = a.length();
    Vector c = a;
    c.normalize();
    put(out20 + 17, c);
}

int ref_aabb_intercepts(const float* mn, const float* mx, const float* o, const float* d, float* t) {
    AABB box(V(mn), V(mx));
    Ray r(V(o), V(d));
    float tt = 0;
    bool h = box.intercepts(r, tt);
    *t = tt;
    return h ? 1 : 0;
}
int ref_aabb_inside(const float* mn, const float* mx, const float* p) {
    AABB box(V(mn), V(mx));
    return box.isInside(V(p)) ? 1 : 0;
}
void ref_aabb_centroid(const float* mn, const float* mx, float* c) {
    AABB box(V(mn), V(mx));
    put(c, box.centroid());
}

// cam9 = from3 at3 up3 ; cam6 = angle hither resx resy aperture_ratio focal_ratio
// derived19 laid out like p3o_scene_camera
void* ref_camera_new(const float* cam9, const float* cam6, float* derived19) {
    Camera* c = new Camera(V(cam9), V(cam9 + 3), V(cam9 + 6), cam6[0], cam6[1], 100.0 * cam6[1],
                           (int)cam6[2], (int)cam6[3], cam6[4], cam6[5]);
    if (derived19) {
        put(derived19, c->eye); put(derived19 + 3, c->u); put(derived19 + 6, c->v);
        put(derived19 + 9, c->n);
        derived19[12] = c->w; derived19[13] = c->h; derived19[14] = c->plane_dist;
        derived19[15] = c->aperture; derived19[16] = c->focal_ratio;
        derived19[17] = (float)c->res_x; derived19[18] = (float)c->res_y;
    }
    return c;
}
void ref_camera_free(void* c) { delete (Camera*)c; }
void ref_camera_ray(void* cam, float px, float py, float* o, float* d) {
    Vector ps(px, py, 0);
    Ray r = ((Camera*)cam)->PrimaryRay(ps);
    put(o, r.origin); put(d, r.direction);
}
void ref_camera_ray_lens(void* cam, float lx, float ly, float px, float py, float* o, float* d) {
    Vector ls(lx, ly, 0), ps(px, py, 0);
    Ray r = ((Camera*)cam)->PrimaryRay(ls, ps);
    put(o, r.origin); put(d, r.direction);
}

uint8_t ref_u8fromfloat(float x) { return u8fromfloat(x); }
void ref_rand_floats(unsigned seed, int n, float* out) {
    set_rand_seed(seed);
    for (int i = 0; i < n; i++) out[i] = rand_float();
}
void ref_color_ops(const float* a3, const float* b3, float* out12) {
    Color a(a3[0], a3[1], a3[2]), b(b3[0], b3[1], b3[2]);
    Color c = a.clamp();
    out12[0] = c.r(); out12[1] = c.g(); out12[2] = c.b();
    Color m = a * b;
    out12[3] = m.r(); out12[4] = m.g(); out12[5] = m.b();
    Color s = a * b3[0];
    out12[6] = s.r(); out12[7] = s.g(); out12[8] = s.b();
    Color q = a / b3[0];
    out12[9] = q.r(); out12[10] = q.g(); out12[11] = q.b();
}

void* ref_accel_new(int n, const int* type, const float* data12) {
    Accel* a = new Accel();
    for (int i = 0; i < n; i++) {
        HObj* o = new HObj();
        o->id = i;
        o->type = type[i];
        memcpy(o->d, data12 + 12 * i, sizeof(o->d));
        a->objs.push_back(o);
    }
    return a;
}
void ref_accel_free(void* h) {
    Accel* a = (Accel*)h;
    for (auto* o : a->objs) delete o;
    delete a->bvh; delete a->grid; delete a;
}
int ref_bvh_build(void* h) {
    Accel* a = (Accel*)h;
    std::vector<Object*> objs(a->objs.begin(), a->objs.end());
    a->bvh = new BVH();
    a->bvh->Build(objs);                               // RT/bvh.cpp:28
    return (int)a->bvh->nodes.size();
}
void ref_bvh_dump(void* h, float* nodes8, int* n_objs, int* order) {
    Accel* a = (Accel*)h;
    for (size_t i = 0; i < a->bvh->nodes.size(); i++) {
        BVH::BVHNode* nd = a->bvh->nodes[i];
        AABB& b = nd->getAABB();
        float v[8] = {b.min.x, b.min.y, b.min.z, b.max.x, b.max.y, b.max.z,
                      nd->isLeaf() ? 1.0f : 0.0f, (float)nd->getIndex()};
        memcpy(nodes8 + 8 * i, v, sizeof v);
        n_objs[i] = nd->isLeaf() ? (int)nd->getNObjs() : 0;
    }
    for (size_t i = 0; i < a->bvh->objects.size(); i++) order[i] = a->index_of(a->bvh->objects[i]);
}
int ref_bvh_shadow(void* h, const float* o, const float* d) {
    Accel* a = (Accel*)h;
    Ray r(V(o), V(d));
    return a->bvh->Traverse(r) ? 1 : 0;                 // RT/bvh.cpp:348
}
int ref_bvh_closest(void* h, const float* o, const float* d, int* obj, float* t) {
    Accel* a = (Accel*)h;
    Ray r(V(o), V(d));
    Object* ho = NULL;
    Vector hp;
    bool ok = a->bvh->Traverse(r, &ho, hp);            // RT/bvh.cpp:252
    *obj = ho ? a->index_of(ho) : -1;
    if (ho) { float tt = FLT_MAX; ho->intercepts(r, tt); *t = tt; }
    return ok ? 1 : 0;
}
int ref_bvh_stack_size(void* h) { return (int)((Accel*)h)->bvh->hit_stack.size(); }

void ref_grid_build(void* h, int* nxyz) {
    Accel* a = (Accel*)h;
    std::vector<Object*> objs(a->objs.begin(), a->objs.end());
    a->grid = new Grid();
    a->grid->Build(objs);                              // RT/grid.cpp:30
    nxyz[0] = a->grid->nx; nxyz[1] = a->grid->ny; nxyz[2] = a->grid->nz;
}
void ref_grid_cell_counts(void* h, int* counts) {
    Accel* a = (Accel*)h;
    for (size_t i = 0; i < a->grid->cells.size(); i++) counts[i] = (int)a->grid->cells[i].size();
}
int ref_grid_shadow(void* h, const float* o, const float* d) {
    Accel* a = (Accel*)h;
    Ray r(V(o), V(d));
    return a->grid->Traverse(r) ? 1 : 0;               // RT/grid.cpp:313
}
int ref_grid_closest(void* h, const float* o, const float* d, int* obj, float* t) {
    Accel* a = (Accel*)h;
    Ray r(V(o), V(d));
    Object* ho = NULL;
    Vector hp;
    bool ok = a->grid->Traverse(r, &ho, hp);           // RT/grid.cpp:248
    *obj = (ok && ho) ? a->index_of(ho) : -1;
    if (ok && ho) { float tt = FLT_MAX; ho->intercepts(r, tt); *t = tt; }
    return ok ? 1 : 0;
}

}  // extern "C"
