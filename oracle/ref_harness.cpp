// ref_harness.cpp -- glue that exposes the REFERENCE'S OWN compiled objects to the tests.
//
// oracle/Makefile compiles, unchanged and from where they lie under /root/reference,
//   RT/vector.cpp, RT/boundingBox.cpp, RT/bvh.cpp, RT/grid.cpp            (whole files)
//   RT/scene.cpp:1-331     (Triangle/Plane/Sphere/aaBox bodies, Scene accessors)
//   RT/main.cpp:9-15,18,20-33,35-104,471-730  (globals, processLight, rayTracing, sampleUnitDisk;
//                          line 34 `#define MAX_DEPTH 4` is replaced by -DMAX_DEPTH=N so that the
//                          depth-2 and depth-6 BASELINE configs can be rendered too)
// (the line ranges are streamed from the reference files into g++'s stdin; no reference text is
// written into this repository) and links them with this file into oracle/_ref/libp3d_ref[_dN].so
// (git-ignored; travels to the GPU box as a built artefact).  The rest of those two files
// (viewer shell, DevIL skybox/PNG code, .p3f loader with its LoadSkybox call, the MSVC-only
// create_random_scene) needs libraries this image lacks and is not built.
//
// Nothing here restates reference arithmetic: scenes are built from the reference's own
// Sphere/Triangle/aaBox/Plane/Material/Light/Camera/Scene classes, the accelerators by BVH::Build /
// Grid::Build as init_scene() does (RT/main.cpp:912-936), and every ray goes through the
// reference's rayTracing().  The one thing written out here is the pixel loop of renderScene()
// (RT/main.cpp:749-805): that function also calls GL/GLUT/DevIL and cannot be linked.
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
#include <unordered_map>
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

#ifndef REF_MAX_DEPTH
#define REF_MAX_DEPTH 4
#endif

// ---- the reference's globals and functions (RT/main.cpp:40-45,86-103,471,530,724), defined in the
//      object compiled from main.cpp's line ranges
extern bool ANTI_ALIASING, SOFT_SHADOW, DEPTH_OF_FIELD, FUZZY_REFLECTION, MOTION_BLUR, SCHLICK_APPROX;
extern Scene* scene;
extern Grid* grid_ptr;
extern BVH* bvh_ptr;
extern accelerator Accel_Struct;
extern int RES_X, RES_Y;
extern int offset_for_shadowx, offset_for_shadowy;
extern int globalSamplesPerPixel;
Color rayTracing(Ray ray, int depth, float ior_1);
Vector sampleUnitDisk(void);

namespace {

enum { T_SPHERE = 0, T_TRIANGLE = 1, T_BOX = 2, T_PLANE = 3 };   // = P3O_* of p3d_oracle.h

Vector V(const float* p) { return Vector(p[0], p[1], p[2]); }
void put(float* o, Vector v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }

// prim12 in loader form: sphere c3,r | triangle p0,p1,p2 | box min3,max3 | plane p0,p1,p2
Object* make_object(int type, const float* d) {
    Vector a = V(d), b = V(d + 3), c = V(d + 6);
    switch (type) {
    case T_SPHERE: return new Sphere(a, d[3]);              // RT/scene.cpp:513-523 (loader call)
    case T_TRIANGLE: return new Triangle(a, b, c);          // RT/scene.cpp:535-554
    case T_BOX: return new aaBox(a, b);                     // RT/scene.cpp:525-534
    default: return new Plane(a, b, c);                     // RT/scene.cpp:587-596
    }
}

struct Accel {
    std::vector<Object*> objs;
    std::unordered_map<Object*, int> index;
    BVH* bvh = nullptr;
    Grid* grid = nullptr;
    bool own_objs = true;
    int index_of(Object* o) const { auto it = index.find(o); return it == index.end() ? -1 : it->second; }
};

struct RScene {
    Scene* sc = nullptr;                 // owns the objects (RT/scene.cpp:288-295)
    std::vector<Material*> mats;
    Accel acc;
};

}  // namespace

extern "C" {

int ref_max_depth(void) { return REF_MAX_DEPTH; }

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

// ---- the reference's intersectors (RT/scene.cpp:55-88,122-147,149-172,198-278) on its own objects.
// normal = getNormal(o + d*t).normalize(), what rayTracing() computes first (RT/main.cpp:587).
int ref_intersect(int type, const float* prim12, const float* o, const float* d, float* t_out, float* nrm3) {
    Object* ob = make_object(type, prim12);
    Ray r(V(o), V(d));
    float t = FLT_MAX;
    bool h = ob->intercepts(r, t);
    if (h) {
        *t_out = t;
        if (nrm3) {
            Vector hp = r.origin + r.direction * t;
            put(nrm3, ob->getNormal(hp).normalize());
        }
    }
    delete ob;
    return h ? 1 : 0;
}
void ref_prim_bbox(int type, const float* prim12, float* mn, float* mx) {
    Object* ob = make_object(type, prim12);
    AABB b = ob->GetBoundingBox();
    put(mn, b.min); put(mx, b.max);
    delete ob;
}

// cam9 = from3 at3 up3 ; cam6 = angle hither resx resy aperture_ratio focal_ratio
// derived19 laid out like p3o_scene_camera
void* ref_camera_new(const float* cam9, const float* cam6, float* derived19) {
    Camera* c = new Camera(V(cam9), V(cam9 + 3), V(cam9 + 6), cam6[0], cam6[1], 100.0 * cam6[1],
                           (int)cam6[2], (int)cam6[3], cam6[4], cam6[5]);     // RT/scene.cpp:641
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

// ---- accelerators over the reference's own primitives
void* ref_accel_new(int n, const int* type, const float* data12) {
    Accel* a = new Accel();
    for (int i = 0; i < n; i++) {
        Object* o = make_object(type[i], data12 + 12 * i);
        a->index[o] = i;
        a->objs.push_back(o);
    }
    return a;
}
void ref_accel_free(void* h) {
    Accel* a = (Accel*)h;
    if (a->own_objs) for (auto* o : a->objs) delete o;
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

// ---- whole scenes through the reference's Scene / Material / Light / Camera classes.
// material[i] indexes mat12 (12 floats: diff rgb, Kd, spec rgb, Ks, shine, T, ior, pad), -1 = none;
// lights: 6 floats each (pos3, rgb); cam9/cam6 as ref_camera_new.
void* ref_scene_new(int n, const int* type, const float* data12, const int* material,
                    int n_mat, const float* mat12, int n_lights, const float* light6,
                    const float* bg3, const float* cam9, const float* cam6) {
    RScene* rs = new RScene();
    rs->sc = new Scene();
    for (int m = 0; m < n_mat; m++) {
        const float* p = mat12 + 12 * m;
        Color cd(p[0], p[1], p[2]), cs(p[4], p[5], p[6]);
        rs->mats.push_back(new Material(cd, p[3], cs, p[7], p[8], p[9], p[10]));   // RT/scene.cpp:503-511
    }
    for (int i = 0; i < n; i++) {
        Object* o = make_object(type[i], data12 + 12 * i);
        o->SetMaterial(material[i] >= 0 ? rs->mats[material[i]] : NULL);
        rs->sc->addObject(o);
        rs->acc.index[o] = i;
        rs->acc.objs.push_back(o);
    }
    rs->acc.own_objs = false;
    for (int l = 0; l < n_lights; l++) {
        Vector p = V(light6 + 6 * l);
        Color c(light6[6 * l + 3], light6[6 * l + 4], light6[6 * l + 5]);
        rs->sc->addLight(new Light(p, c));                                          // RT/scene.cpp:598-607
    }
    rs->sc->SetBackgroundColor(Color(bg3[0], bg3[1], bg3[2]));
    rs->sc->SetCamera((Camera*)ref_camera_new(cam9, cam6, NULL));
    return rs;
}
void ref_scene_free(void* h) {
    RScene* rs = (RScene*)h;
    delete rs->acc.bvh; delete rs->acc.grid;
    rs->acc.bvh = nullptr; rs->acc.grid = nullptr;
    delete rs->sc->GetCamera();
    for (int l = 0; l < rs->sc->getNumLights(); l++) delete rs->sc->getLight(l);
    delete rs->sc;
    for (auto* m : rs->mats) delete m;
    delete rs;
}

// Render one frame with the reference's rayTracing().  accel 0/1/2 (RT/scene.h:18); spp as in the
// .p3f `spp` line (0 = Whitted; n = n*n jittered samples + thin lens, RT/main.cpp:939-946); seed
// replaces time(NULL) in set_rand_seed() (RT/main.cpp:747).  soft / fuzzy set the reference's
// SOFT_SHADOW / FUZZY_REFLECTION globals.  Outputs like p3o_render: rgb8 = img_Data (bottom row
// first), rgb32f = the colour handed to u8fromfloat, hit_id = scene index of the primary hit
// (first sample when spp > 0) found by the reference's own closest-hit code, rays = Ray::nextId delta.
// Rows [y0,y1) only when y1 > 0 (the other rows of the outputs are left untouched).
int ref_scene_render(void* h, int accel, int spp, unsigned seed, int soft, int fuzzy, int y0, int y1,
                     uint8_t* rgb8, float* rgb32f, int* hit_id, unsigned long long* rays) {
    RScene* rs = (RScene*)h;
    scene = rs->sc;
    Camera* cam = scene->GetCamera();
    RES_X = cam->GetResX();
    RES_Y = cam->GetResY();
    Accel_Struct = (accelerator)accel;
    // init_scene(), RT/main.cpp:912-936
    std::vector<Object*> objs(rs->acc.objs.begin(), rs->acc.objs.end());
    if (Accel_Struct == GRID_ACC) {
        if (!rs->acc.grid) { rs->acc.grid = new Grid(); rs->acc.grid->Build(objs); }
        grid_ptr = rs->acc.grid;
    } else if (Accel_Struct == BVH_ACC) {
        if (!rs->acc.bvh) { rs->acc.bvh = new BVH(); rs->acc.bvh->Build(objs); }
        bvh_ptr = rs->acc.bvh;
        while (!bvh_ptr->hit_stack.empty()) bvh_ptr->hit_stack.pop();   // fresh object per run
    }
    globalSamplesPerPixel = spp;                                         // RT/main.cpp:938-946
    ANTI_ALIASING = spp != 0;
    DEPTH_OF_FIELD = spp != 0;
    SOFT_SHADOW = soft != 0;
    FUZZY_REFLECTION = fuzzy != 0;
    MOTION_BLUR = false;

    set_rand_seed(seed);                                                 // RT/main.cpp:747
    int id0 = Ray::nextId;
    if (y0 < 0) y0 = 0;
    if (y1 <= 0 || y1 > RES_Y) y1 = RES_Y;       // rows [y0,y1) only: a test-time shortcut for slow scenes
    size_t counter = (size_t)3 * y0 * RES_X;
    for (int y = y0; y < y1; y++) {                                      // RT/main.cpp:749-805
        for (int x = 0; x < RES_X; x++) {
            Color color;
            Vector pixel;
            int hid = -1;
            auto probe = [&](Ray& ray) {
                // scene index of the closest hit, by the reference's own code for this mode:
                // brute force in scene order (RT/main.cpp:567-574, also BVH mode: SURVEY Q1) or
                // Grid::Traverse (RT/main.cpp:555-559)
                if (Accel_Struct == GRID_ACC) {
                    Object* ho = NULL; Vector hp;
                    if (grid_ptr->Traverse(ray, &ho, hp) && ho) return rs->acc.index_of(ho);
                    return -1;
                }
                int best = -1; float closest_t = FLT_MAX;
                for (int i = 0; i < scene->getNumObjects(); i++) {
                    float t = FLT_MAX;
                    if (scene->getObject(i)->intercepts(ray, t) && t < closest_t) { closest_t = t; best = i; }
                }
                return best;
            };
            if (!ANTI_ALIASING) {
                pixel.x = x + 0.5f;
                pixel.y = y + 0.5f;
                Ray ray = cam->PrimaryRay(pixel, MOTION_BLUR);
                hid = probe(ray);
                color = rayTracing(ray, 1, 1.0).clamp();
            } else {
                for (int i = 0; i < globalSamplesPerPixel; i++) {
                    for (int j = 0; j < globalSamplesPerPixel; j++) {
                        offset_for_shadowx = i;
                        offset_for_shadowy = j;
                        pixel.x = x + (i + rand_float()) / globalSamplesPerPixel;
                        pixel.y = y + (j + rand_float()) / globalSamplesPerPixel;
                        Vector cameralens;
                        float aperture = cam->GetAperture();
                        cameralens = sampleUnitDisk() * aperture;
                        Ray ray = cam->PrimaryRay(cameralens, pixel);
                        if (i == 0 && j == 0) hid = probe(ray);
                        color += rayTracing(ray, 1, 1.0).clamp();
                    }
                }
                color = color / (4 * 4);
            }
            if (rgb8) {
                rgb8[counter] = u8fromfloat(static_cast<float>(color.r()));
                rgb8[counter + 1] = u8fromfloat(static_cast<float>(color.g()));
                rgb8[counter + 2] = u8fromfloat(static_cast<float>(color.b()));
            }
            if (rgb32f) { rgb32f[counter] = color.r(); rgb32f[counter + 1] = color.g(); rgb32f[counter + 2] = color.b(); }
            if (hit_id) hit_id[counter / 3] = hid;
            counter += 3;
        }
    }
    if (rays) *rays = (unsigned long long)(Ray::nextId - id0);   // the hit-id probe reuses `ray`: no extra ids
    return 0;
}

// one rayTracing() call on an arbitrary ray (depth 1, ior 1), for shading known-answer tests
void ref_scene_trace(void* h, int accel, int soft, const float* o, const float* d, float* rgb3) {
    RScene* rs = (RScene*)h;
    scene = rs->sc;
    Accel_Struct = (accelerator)accel;
    std::vector<Object*> objs(rs->acc.objs.begin(), rs->acc.objs.end());
    if (Accel_Struct == GRID_ACC) {
        if (!rs->acc.grid) { rs->acc.grid = new Grid(); rs->acc.grid->Build(objs); }
        grid_ptr = rs->acc.grid;
    } else if (Accel_Struct == BVH_ACC) {
        if (!rs->acc.bvh) { rs->acc.bvh = new BVH(); rs->acc.bvh->Build(objs); }
        bvh_ptr = rs->acc.bvh;
    }
    globalSamplesPerPixel = 0; ANTI_ALIASING = false; DEPTH_OF_FIELD = false;
    SOFT_SHADOW = soft != 0; FUZZY_REFLECTION = false; MOTION_BLUR = false;
    Ray ray(V(o), V(d));
    Color c = rayTracing(ray, 1, 1.0);
    rgb3[0] = c.r(); rgb3[1] = c.g(); rgb3[2] = c.b();
}

}  // extern "C"
