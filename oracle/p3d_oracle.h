/*
 * p3d_oracle.h -- C interface of the CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * The oracle is a plain-C++ restatement of the per-pixel Whitted hot path of
 * P3D_RayTracer_Template2 (citations RT/ = /root/reference/P3D_RayTracer_Template2/).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (u_4a_2s_p3d_raytracer_template2_amd/) never includes, links or calls it.
 *
 * Pinning status: PINNED against the reference's own object code.  oracle/Makefile compiles
 * RT/vector.cpp, boundingBox.cpp, bvh.cpp, grid.cpp, RT/scene.cpp:1-331 and RT/main.cpp:471-730
 * (with its globals) unchanged from /root/reference into oracle/_ref/libp3d_ref[_dN].so, and
 * tests/test_oracle_vs_ref.py checks this restatement against it BIT FOR BIT: vector / AABB /
 * camera / quantiser / rand helpers, BVH and grid build + both traversals, the four intersectors
 * (16 000 known answers), single rayTracing() calls, whole frames (float bits, rgb8, Ray::nextId)
 * for every golden case, 24 generated scenes (all accel modes, depths 1-6), the spp > 0 sample
 * loop and the SOFT_SHADOW / FUZZY_REFLECTION branches.  tests/golden/*.npz are reference outputs
 * (tests/golden/make_golden.py).  Not built from the reference, hence pinned only by the scene
 * files themselves: the .p3f loader (RT/scene.cpp:476-675 calls the DevIL skybox loader).
 */
#ifndef P3D_ORACLE_H
#define P3D_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct p3o_scene p3o_scene;

/* primitive type codes shared by every oracle entry point */
enum { P3O_SPHERE = 0, P3O_TRIANGLE = 1, P3O_BOX = 2, P3O_PLANE = 3 };

typedef struct p3o_params {
    int32_t max_depth;    /* RT/main.cpp:34 MAX_DEPTH (compile-time 4 in the reference)          */
    int32_t accel;        /* -1 = value from the file; 0 NONE, 1 GRID, 2 BVH (RT/scene.h:18)     */
    int32_t spp;          /* -1 = value from the file; 0 = Whitted, n = n*n jittered + DOF        */
    uint32_t seed;        /* srand() seed that replaces time(NULL) (RT/main.cpp:747)             */
    int32_t threads;      /* 1 = serial, exactly the reference's order (the oracle proper)       */
    int32_t break_fixed;  /* 1 = use the BVH closest hit, no fall-through (SURVEY Q1 removed)    */
    int32_t y0, y1;       /* row range [y0,y1); y1<=0 means all rows                             */
    int32_t soft_shadow;  /* SOFT_SHADOW of RT/main.cpp:41: 4x4 area-light grid when spp == 0, one
                             jittered light sample per pixel sample otherwise (RT/main.cpp:598-625)  */
    int32_t fuzzy_reflection; /* FUZZY_REFLECTION of RT/main.cpp:43 (RT/main.cpp:651-660)            */
    int32_t skybox;       /* 1 = a miss returns Scene::GetSkyboxColor(ray) (RT/scene.cpp:383-461; never called by
                             the reference itself) from the cube map given to p3o_scene_set_skybox */
} p3o_params;

typedef struct p3o_counters {
    uint64_t rays;            /* what Ray::nextId would reach (RT/ray.h:10)                      */
    uint64_t closest_queries; /* rayTracing() invocations                                        */
    uint64_t shadow_queries;  /* shadow rays constructed                                         */
    uint64_t aabb_tests;      /* AABB::intercepts calls                                          */
    uint64_t sphere_tests, tri_tests, box_tests, plane_tests;
    uint64_t get_object;      /* Scene::getObject calls (SURVEY Q1 evidence)                     */
} p3o_counters;

p3o_scene* p3o_scene_load(const char* path);
void       p3o_scene_free(p3o_scene*);
/* out[0..7] = n_prims, n_lights, n_materials, res_x, res_y, accel, spp, parse_ok */
void       p3o_scene_info(const p3o_scene*, int32_t* out);
void       p3o_scene_set_resolution(p3o_scene*, int32_t w, int32_t h);
/* per-primitive dump in scene order: type[n], data[n*12], material[n] */
void       p3o_scene_prims(const p3o_scene*, int32_t* type, float* data12, int32_t* material);
/* materials: 12 floats each = diff rgb, Kd, spec rgb, Ks, shine, T, ior, 0 */
void       p3o_scene_materials(const p3o_scene*, float* out12);
void       p3o_scene_lights(const p3o_scene*, float* out6);
void       p3o_scene_bg(const p3o_scene*, float* out3);
/* camera derived values: eye3,u3,v3,n3,w,h,plane_dist,aperture,focal_ratio,res_x,res_y (19 floats) */
void       p3o_scene_camera(const p3o_scene*, float* out19);

/* Full-frame render. rgb8 is bottom row first exactly like img_Data (RT/main.cpp:803-805).
 * rgb32f / hit_id / NULL allowed. hit_id = scene index of the primary hit or -1
 * (first sample when spp>0). Returns 0 on success. */
int p3o_render(p3o_scene*, const p3o_params*, uint8_t* rgb8, float* rgb32f,
               int32_t* hit_id, p3o_counters* ctr);

/* The six cube-map faces Scene::LoadSkybox would hold (RT/scene.cpp:333-381: right, left, top, bottom, front, back;
 * rows bottom-up, 3 or 4 bytes per pixel); copied.  p3o_skybox_color = Scene::GetSkyboxColor (RT/scene.cpp:383-461). */
void p3o_scene_set_skybox(p3o_scene*, const uint8_t* const faces[6], const uint32_t* res_x, const uint32_t* res_y,
                          const uint32_t* bytes_per_pixel);
void p3o_skybox_color(const p3o_scene*, const float* dir3, float* rgb3);

/* one rayTracing(ray, 1, 1.0) call on an arbitrary ray (unclamped colour) */
void p3o_trace(p3o_scene*, int accel, int max_depth, int soft_shadow, const float* o3, const float* d3,
               float* rgb3);

/* ---- known-answer entry points (unit level) ---- */
/* prim12: sphere c3,r | triangle p0,p1,p2 | box min3,max3 | plane p0,p1,p2 */
int   p3o_intersect(int type, const float* prim12, const float* o3, const float* d3,
                    float* t_out, float* normal_at_hit3);
int   p3o_aabb_intercepts(const float* min3, const float* max3, const float* o3,
                          const float* d3, float* t_out);
void  p3o_prim_bbox(int type, const float* prim12, float* min3, float* max3);
void  p3o_normalize(float* v3);
void  p3o_primary_ray(const p3o_scene*, float px, float py, float* o3, float* d3);
void  p3o_primary_ray_lens(const p3o_scene*, float lx, float ly, float px, float py,
                           float* o3, float* d3);
uint8_t p3o_u8fromfloat(float x);
void  p3o_rand_floats(uint32_t seed, int32_t n, float* out);

/* reference-BVH restatement, exposed for the oracle-vs-_ref comparison */
int32_t p3o_refbvh_node_count(p3o_scene*);
/* nodes: 8 floats each = min3,max3,(float)leaf,(float)index ; n_objs separately */
void  p3o_refbvh_dump(p3o_scene*, float* nodes8, int32_t* n_objs, int32_t* order);
int   p3o_refbvh_shadow(p3o_scene*, const float* o3, const float* d3);
int   p3o_refbvh_closest(p3o_scene*, const float* o3, const float* d3, int32_t* obj, float* t);
void  p3o_refgrid_dims(p3o_scene*, int32_t* nxyz, int32_t* cell_counts /* nx*ny*nz or NULL */);
int   p3o_refgrid_shadow(p3o_scene*, const float* o3, const float* d3);
int   p3o_refgrid_closest(p3o_scene*, const float* o3, const float* d3, int32_t* obj, float* t);

#ifdef __cplusplus
}
#endif
#endif
