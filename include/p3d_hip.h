/*
 * p3d_hip.h -- C-ABI of the MI355X-native Whitted renderer (libp3d_hip.so).
 *
 * The reference (P3D_RayTracer_Template2, RT/ = /root/reference/P3D_RayTracer_Template2/)
 * has no plugin/FFI surface: its hot path is one function, renderScene() (RT/main.cpp:732),
 * reached through globals.  This header is the boundary a maintainer binds instead of that
 * loop: plain pointers and sizes only, no C++ or torch types.  Each entry point names the
 * reference code it replaces.  INTEGRATION.md shows the reference-side call sites.
 *
 * Conventions
 *   - every function returns 0 on success or a negative p3d_status; p3d_last_error()
 *     describes the failure (the reference prints and exit()s, RT/main.cpp:827; the library
 *     never exits the process);
 *   - the caller owns every host buffer; the library owns device memory behind p3d_scene;
 *   - a p3d_scene is bound to one HIP device and must not be used from two threads at once;
 *   - render calls are asynchronous on the scene's HIP stream; p3d_sync() or a host-memory
 *     download waits for them.
 */
#ifndef P3D_HIP_H
#define P3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P3D_ABI_VERSION 4

typedef enum p3d_status {
    P3D_OK = 0,
    P3D_ERR_ARG = -1,        /* bad argument / inconsistent sizes                       */
    P3D_ERR_HIP = -2,        /* a HIP runtime call failed                               */
    P3D_ERR_NO_DEVICE = -3,  /* no usable gfx950 device                                 */
    P3D_ERR_LIMIT = -4,      /* scene exceeds a kernel limit (LDS stack, depth)         */
    P3D_ERR_STATE = -5,      /* call made in the wrong state                            */
    P3D_ERR_COMM = -6        /* an RCCL call failed                                     */
} p3d_status;

/* primitive kinds, RT/scene.h:67-145 */
enum { P3D_SPHERE = 0, P3D_TRIANGLE = 1, P3D_BOX = 2, P3D_PLANE = 3 };

/* accelerator enum of RT/scene.h:18; selects the SHADOW-RAY semantics:
 * NONE = un-normalised direction, no distance bound; GRID/BVH = normalised, t < |L|.
 * Closest hits are always "nearest, lowest scene index on ties" (SURVEY Q1). */
enum { P3D_ACCEL_NONE = 0, P3D_ACCEL_GRID = 1, P3D_ACCEL_BVH = 2 };

/* Flattened scene in SCENE ORDER (the order Scene::addObject saw, RT/scene.cpp:302).
 * prim_data holds 12 floats per primitive:
 *   sphere   c.x c.y c.z r                    (RT/scene.h:116-131)
 *   triangle P0 P1 P2                         (RT/scene.cpp:10)
 *   box      min max                          (RT/scene.cpp:188)
 *   plane    PN.x PN.y PN.z D                 (unit normal and offset, RT/scene.cpp:95-115)
 * materials hold 12 floats: diffuse rgb, Kd, specular rgb, Ks, shine, T, ior, reflection
 * (RT/scene.h:23-55; reflection == Ks for loader-made materials, RT/scene.h:31).
 * lights hold 6 floats: position xyz, colour rgb (RT/scene.h:57-65). */
typedef struct p3d_scene_desc {
    uint32_t        n_prims;
    const uint32_t* prim_type;
    const float*    prim_data;
    const uint32_t* prim_material;
    uint32_t        n_materials;
    const float*    materials;
    uint32_t        n_lights;
    const float*    lights;
    float           background[3];   /* Scene::GetBackgroundColor, RT/scene.h:155 */
} p3d_scene_desc;

/* BVH construction knobs (replaces BVH::Build, RT/bvh.cpp:28-158; the tree need not match
 * the reference's because its closest-hit result is discarded). */
typedef struct p3d_build_opts {
    uint32_t leaf_max;        /* max primitives per leaf, 1..8; 0 = default (4)          */
    uint32_t sah_bins;        /* 0 = default (16)                                       */
    uint32_t builder;         /* 0 = binned SAH on the host (default); 1 = linear BVH built on the
                                 device (Morton sort + Karras hierarchy + refit): a tree of lower
                                 quality in a fraction of the time, for scene-reload loops and
                                 scenes of millions of primitives. Same images either way.     */
    uint32_t cull_never_hit;  /* 1 = leave out of the BVH every triangle the reference's own test can
                                 never accept: Triangle::intercepts rejects |det| < 1e-3
                                 (RT/scene.cpp:66-67) and |det| <= |d| * |e1 x e2|, so a
                                 triangle with sqrt(2) * |e1 x e2| below that threshold is invisible to
                                 every ray whose direction is at most sqrt(2) long -- all closest-hit
                                 rays (the odd refraction ray of SURVEY Q6 reaches sqrt(2)) and the
                                 normalised shadow rays of GRID / BVH mode. NONE-mode shadow rays are
                                 not normalised, so p3d_render() rejects accel NONE on such a scene.
                                 Same images, same ray counts, fewer box / triangle tests. Default 0:
                                 every primitive is traversed, like in the reference.            */
} p3d_build_opts;

/* The values Camera::Camera derives (RT/camera.h:35-73); PrimaryRay (RT/camera.h:91-127)
 * is evaluated on the device from these. */
typedef struct p3d_camera {
    float   eye[3], u[3], v[3], n[3];
    float   w, h, plane_dist;
    float   aperture;        /* lens aperture in world units, RT/camera.h:65            */
    float   focal_ratio;
    int32_t res_x, res_y;
} p3d_camera;

typedef struct p3d_render_params {
    int32_t  max_depth;      /* MAX_DEPTH, RT/main.cpp:34                                */
    int32_t  accel;          /* P3D_ACCEL_*: shadow-ray semantics, RT/main.cpp:476-510  */
    int32_t  spp;            /* 0 = Whitted one sample at pixel centre; n = n*n samples
                                with thin lens, summed and divided by 16   */
    const float* samples;    /* spp>0: HOST array [res_y][res_x][spp*spp][4] =
                                pixel sample x, y, lens x, lens y in reference RNG order
                                (RT/main.cpp:776-801); uploaded by the call              */
    /* image-space sharding: this device renders the row blocks b with
     * b % world == rank, row_block rows each, into a COMPACT buffer of
     * p3d_local_rows() rows.  world = 1 renders the whole frame. */
    int32_t  row_block;      /* rows per block, multiple of 16; 0 = default (16)         */
    int32_t  rank, world;
    uint32_t flags;          /* P3D_FLAG_*                                              */
    uint32_t features;       /* P3D_FEATURE_*: the reference's distribution-ray-tracing switches */
    uint32_t seed;           /* seed of the device random streams those features draw from       */
} p3d_render_params;

/* SOFT_SHADOW / FUZZY_REFLECTION of RT/main.cpp:41,43 (compile-time false there).
 * SOFT_SHADOW: every light becomes the reference's 0.5 x 0.5 area light -- with spp == 0 its
 * deterministic 4x4 grid of sub-lights of colour/16 (RT/main.cpp:601-618, bit-faithful), with
 * spp > 0 one jittered position per pixel sample in that sample's stratum (RT/main.cpp:620-624).
 * FUZZY_REFLECTION: mirror directions perturbed inside a sphere of radius 0.3 (RT/main.cpp:651-660).
 * The jitter and the fuzz draw from counter-based device random streams keyed by (seed, pixel, sample,
 * tree node): the reference's serial rand() stream is not reproducible in parallel, so these two are
 * statistically, not bitwise, equal to it. They need the wavefront schedule (P3D_FLAG_TREE_KERNEL is
 * rejected). DEPTH_OF_FIELD is implied by spp > 0 as in the reference (RT/main.cpp:943-944);
 * MOTION_BLUR only stamps rays with a time no object reads, so it has no switch. */
#define P3D_FEATURE_SOFT_SHADOW 1u
#define P3D_FEATURE_FUZZY_REFLECTION 2u
/* A ray that hits nothing returns Scene::GetSkyboxColor(ray) (RT/scene.cpp:383-461) from the cube map given to
 * p3d_scene_set_skybox() instead of the background colour.  The reference holds this function and the `env` loader
 * command (RT/scene.cpp:652-658) but never calls the former -- its rayTracing() returns bgColor on a miss
 * (RT/main.cpp:582) -- so this switch is what the function is evidently there for, off by default.
 * The lookup itself is bit-exact against the reference's object code (tests/test_oracle_vs_ref.py).  Like the other
 * features it needs the tile or the wavefront schedule. */
#define P3D_FEATURE_SKYBOX 4u

#define P3D_FLAG_COUNTERS 1u     /* accumulate p3d_counters on the device (slower kernels)  */
/* Kernel schedules: three ways to run the same per-node code, bit-identical frames.  Which one p3d_render() uses
 * when none is forced (p3d_last_schedule() reports it):
 *   - scenes whose flattened records fit 24 KiB are rendered from an LDS copy, and go BY RULE: one-sample frames
 *     (spp == 0) the wavefront schedule, sample loops (spp > 0) the tile schedule;
 *   - every other scene is read from HBM / L2 and the choice is MEASURED per configuration (resolution, depth, accel,
 *     spp, rank / world, flags): the first six frames of a configuration run the wavefront, tree and tile schedules
 *     twice each (first run untimed: code-object load, workspace allocation; a schedule whose workspace does not fit
 *     is skipped), the fastest one stays.  During those frames p3d_render() WAITS on a HIP event for the previous
 *     frame's timing even when p3d_outputs::memory == 1, i.e. it is not asynchronous; a frame issued while the stream
 *     is being captured measures nothing and uses the choice already made (the tile schedule if there is none yet).
 * A schedule whose workspace does not fit the budget falls back tile -> wavefront (in bands) -> tree.
 * One scene handle carries ONE frame at a time per stream: its workspace and the parity words of its queues are
 * ordered by the stream only, so a caller must not issue frames of one handle on two streams concurrently (use one
 * handle per stream, as bench.py does).  At most one of the three forcing flags may be set. */
#define P3D_FLAG_TILE_KERNEL 64u /* ONE launch per frame: persistent 256-thread workgroups draw 16x16-pixel tiles and
                                    run a tile's whole ray tree level by level among themselves (queues in a
                                    private workspace slot, counters in LDS, no global atomics between levels)  */
#define P3D_FLAG_WAVEFRONT 32u   /* one launch per tree level over the whole frame + resolve launches          */
#define P3D_FLAG_DEVICE_SAMPLES 128u /* p3d_render_params::samples is a DEVICE pointer on the scene's device (same
                                    layout): the caller uploaded the sample array once instead of per call    */
#define P3D_FLAG_PROFILE 16u     /* bracket the frame and its dominant kernel (the level-1 /
                                    tree launch) with HIP events for p3d_get_profile()           */
#define P3D_FLAG_PRIVATE_WALK 8u /* scenes read from HBM: every lane walks the BVH for its own ray only.  Without it the
                                    lanes of a wave may SHARE their walks -- idle lanes take over pending subtrees of busy
                                    ones, results merged by minimum over (t, scene id) / OR: same hits, same bits -- and
                                    whether they do is part of the measured schedule choice                              */
#define P3D_FLAG_PACKET_WALK 256u /* wave-wide (packet) BVH walk for trees of up to 64 node pairs: node and primitive
                                    records fetched once per wave, a node visited when any lane's slab test passes.
                                    Same results as the default per-lane walk; measured slower since the leaves
                                    became typed runs (0.137 vs 0.133 ms on BASELINE config 2)                       */
#define P3D_FLAG_NO_LDS_SCENE 4u /* read the scene from HBM/L2 even when it would fit in LDS    */
#define P3D_FLAG_TREE_KERNEL 2u  /* one launch, each lane walks its pixel's whole tree with a post-order frame
                                    stack (LDS, or private memory for scenes read from HBM); also the last
                                    fallback when neither the tile nor the wavefront workspace fits the budget */

/* Work counters in the unit of SURVEY §8d (one ray = one closest-hit or shadow query). */
typedef struct p3d_counters {
    uint64_t closest_queries;
    uint64_t shadow_queries;
    uint64_t box_tests;      /* node AABB slab tests (32 algorithmic bytes each)         */
    uint64_t sphere_tests;   /* 16 B each                                               */
    uint64_t tri_tests;      /* 48 B each                                               */
    uint64_t aabox_tests;    /* 32 B each                                               */
    uint64_t plane_tests;    /* 16 B each                                               */
    uint64_t pixels;
} p3d_counters;

/* Output planes. Pointers may be NULL. memory: 0 = host pointers (the call copies and
 * returns after the frame is complete), 1 = device pointers on the scene's device (the
 * call only enqueues the kernel). A plane holds res_y rows when world == 1 and
 * p3d_local_rows() rows (this rank's compact shard) when world > 1. Layout follows img_Data / colors of RT/main.cpp:70-76:
 * row 0 is the BOTTOM row; rgb8 is 3 bytes per pixel, rgb32f 3 floats per pixel (clamped
 * colour before quantisation), hit_id the scene index of the primary hit or -1. */
typedef struct p3d_outputs {
    uint8_t* rgb8;
    float*   rgb32f;
    int32_t* hit_id;
    int32_t  memory;
} p3d_outputs;

typedef struct p3d_scene p3d_scene;

typedef struct p3d_scene_stats {
    uint32_t n_nodes, n_leaves, max_depth, n_leaf_refs;
    uint32_t n_spheres, n_triangles, n_boxes, n_planes, n_culled;
    uint64_t device_bytes;
    float    sah_cost;
} p3d_scene_stats;

int         p3d_abi_version(void);
const char* p3d_last_error(void);
int         p3d_device_count(int* count);

/* Replaces init_scene()'s accelerator set-up (RT/main.cpp:912-936) and BVH::Build
 * (RT/bvh.cpp:28): flattens nothing (the caller did), builds the BVH on the host, uploads
 * scene + BVH to `device`. opts may be NULL.  The uniform grid of GRID mode (Grid::Build, RT/grid.cpp:30) is
 * built and uploaded by the first p3d_render() with accel == P3D_ACCEL_GRID -- a one-off synchronous cost of that
 * frame; such a frame is refused (P3D_ERR_STATE) while the stream is being captured. */
int p3d_scene_create(const p3d_scene_desc* desc, const p3d_build_opts* opts, int device,
                     p3d_scene** out);
int p3d_scene_destroy(p3d_scene* scene);
/* Replaces Scene::LoadSkybox (RT/scene.cpp:333-381; DevIL image loading stays with the caller): the six faces in the
 * reference's order right, left, top, bottom, front, back, each res_x[i] x res_y[i] pixels of bytes_per_pixel[i] (3 or
 * 4) bytes, rows bottom-up (the reference loads them with a lower-left origin).  HOST pointers; copied to the device. */
int p3d_scene_set_skybox(p3d_scene* scene, const uint8_t* const faces[6], const uint32_t res_x[6], const uint32_t res_y[6],
                         const uint32_t bytes_per_pixel[6]);
int p3d_scene_get_stats(const p3d_scene* scene, p3d_scene_stats* out);

/* Rows of the compact per-rank tile buffer: ceil(n_row_blocks / world) * row_block, the
 * same on every rank so that the gather moves equal-sized buffers (rows past the image are
 * never written). */
int p3d_local_rows(int32_t res_y, int32_t row_block, int32_t world);

/* Replaces renderScene() (RT/main.cpp:732-832) incl. Camera::PrimaryRay, rayTracing(),
 * processLight() and every intercepts() beneath them. */
int p3d_render(p3d_scene* scene, const p3d_camera* cam, const p3d_render_params* params,
               const p3d_outputs* out);
int p3d_sync(p3d_scene* scene);
/* counters of the most recent render made with P3D_FLAG_COUNTERS (waits for it) */
int p3d_get_counters(p3d_scene* scene, p3d_counters* out);

/* Launch tuning that never changes results (0 keeps the current value): xcd_chunk =
 * consecutive 16x4-pixel tiles given to one XCD before moving to the next (1 = round robin,
 * best load balance; larger = more L2 locality per XCD for big scenes); workspace_mib = HBM
 * budget for the wavefront ray queues (default 65536, shared by the concurrent sample passes of a frame; frames that need more run in bands);
 * waves_per_simd = register budget of the ray kernels expressed as resident waves per SIMD:
 * 0 compiler default, 5 / 6 trade spilled registers for latency hiding, -1 keeps. */
int p3d_set_tuning(p3d_scene* scene, int32_t xcd_chunk, int32_t workspace_mib, int32_t waves_per_simd);

/* Elapsed device time of the most recent render made with P3D_FLAG_PROFILE (waits for it):
 * the whole frame (all launches of the call, samples and bands included) and its dominant
 * kernel alone -- wf_primary_kernel, or whitted_tree_kernel with P3D_FLAG_TREE_KERNEL -- for
 * the first sample / band. HIP events on the scene's stream. */
int p3d_get_profile(p3d_scene* scene, float* frame_ms, float* kernel_ms);

/* Kernel schedule the most recent p3d_render() of this scene used: 0 = wavefront (level kernels),
 * 1 = tree (one launch, per-lane stacks), 2 = tile (one launch, per-tile levels). P3D_ERR_STATE before the
 * first render. */
int p3d_last_schedule(p3d_scene* scene, int32_t* schedule);

/* Schedule choice under the caller's load. For scenes read from HBM the library measures its kernel schedules (and whether
 * the lanes of a wave share their BVH walks) on the first frames of a configuration, ONE frame at a time, and keeps the
 * fastest (p3d_last_schedule). A caller that keeps several frames in flight -- n scene handles of the same scene, one
 * stream each -- can have the same candidates measured the way it runs them: every candidate renders `frames` frames
 * (<= 0: 3) on all n handles at once, timed as a batch on the host clock, and all handles adopt the candidate with the
 * shortest time per frame for this configuration (resolution, depth, accel, spp, flags, features). outs[i] is where
 * handle i renders (device memory: the frames are real frames; results never depend on the choice). Synchronous.
 * ms_per_frame (6 floats or NULL): wavefront / tree / tile with shared walks, then with private walks; -1 = not
 * available. best (or NULL): the adopted candidate's index, -1 when this configuration's schedule is set by rule
 * (scenes served from LDS) or by a P3D_FLAG_* of params -- then nothing is measured. Not while capturing a graph. */
int p3d_tune_schedule(p3d_scene** scenes, int32_t n, const p3d_camera* cam, const p3d_render_params* params,
                      const p3d_outputs* outs, int32_t frames, float* ms_per_frame, int32_t* best);

/* Use an existing hipStream_t (e.g. the caller's framework stream); NULL restores the
 * scene's own stream. */
int p3d_set_stream(p3d_scene* scene, void* hip_stream);

/* HIP-event bracket on the scene's stream: begin, enqueue renders, end -> elapsed ms of
 * everything enqueued in between (waits for completion). */
int p3d_timer_begin(p3d_scene* scene);
int p3d_timer_end(p3d_scene* scene, float* elapsed_ms);

/* Rank-0 side of the multi-GPU frame: `gathered` points at rank 0's compact
 * tile buffer (p3d_local_rows() rows); rank r's buffer starts rank_stride_bytes * r further
 * (0 = buffers back to back). Writes the full bottom-up frame. bytes_per_pixel = 3 (rgb8),
 * 12 (rgb32f) or 4 (hit_id). Device pointers, enqueued on the scene's stream. */
int p3d_deinterleave(p3d_scene* scene, const void* gathered, void* frame, int32_t res_x,
                     int32_t res_y, int32_t row_block, int32_t world, int32_t bytes_per_pixel,
                     uint64_t rank_stride_bytes);
/* The same for a batch of n_frames frames in ONE launch: frame f of rank r starts at
 * gathered + r * rank_stride_bytes + f * tile_stride_bytes (0 = one compact tile buffer; rank stride 0 =
 * n_frames tile buffers back to back) and is written to frames + f * frame_stride_bytes (0 = frames
 * back to back). Rows and strides that are multiples of 16 bytes are moved 16 bytes at a time. */
int p3d_deinterleave_frames(p3d_scene* scene, const void* gathered, void* frames, int32_t res_x, int32_t res_y,
                            int32_t row_block, int32_t world, int32_t bytes_per_pixel, uint64_t rank_stride_bytes,
                            int32_t n_frames, uint64_t tile_stride_bytes, uint64_t frame_stride_bytes);

/* ---- the multi-GPU frame (replaces nothing in the reference, which is one CPU thread:
 * it is what main()'s "renderScene(); save image" (RT/main.cpp:966-970) becomes on N GPUs) ----
 * Every rank renders its row blocks (p3d_render with rank / world) into a compact tile buffer of
 * p3d_local_rows() rows; ONE gather per frame moves the tile buffers to rank 0 over RCCL (grouped
 * ncclSend / ncclRecv: each peer writes straight into rank 0's memory over its own xGMI link, no ring),
 * where p3d_deinterleave() restores row order.  Scene + BVH are replicated: one p3d_scene per device. */
typedef struct p3d_comm p3d_comm;
#define P3D_COMM_ID_BYTES 128

/* One process per GPU: rank 0 makes an id (ncclGetUniqueId), the launcher carries its 128 bytes to the
 * other ranks (a torch.distributed / MPI broadcast, a file, ...), then every rank calls p3d_comm_create
 * with the same id (ncclCommInitRank; blocks until all `world` ranks have called). */
int p3d_comm_unique_id(void* id_out /* P3D_COMM_ID_BYTES */);
int p3d_comm_create(const void* id, int rank, int world, int device, p3d_comm** out);
/* One process driving n GPUs: out[r] is rank r on devices[r] (NULL = devices 0..n-1); ncclCommInitAll. */
int p3d_comm_create_all(const int* devices, int n, p3d_comm** out /* [n] */);
int p3d_comm_destroy(p3d_comm* comm);
int p3d_comm_info(const p3d_comm* comm, int* rank, int* world, int* device);

/* The gather, enqueued on `scene`'s stream (scene and comm on the same device): rank r > 0 sends
 * tile_bytes from `tile`; rank 0 receives rank r's bytes at gathered + r * tile_bytes and copies its own
 * tile to gathered + 0 (skipped when tile == gathered).  `gathered` is only read on rank 0.  Device
 * pointers.  Asynchronous: p3d_sync() / stream order as usual.  world == 1 degenerates to the copy. */
int p3d_gather(p3d_comm* comm, p3d_scene* scene, const void* tile, void* gathered, uint64_t tile_bytes);
/* The same for all n ranks of a p3d_comm_create_all() group from ONE thread (their sends and receives
 * must share one RCCL group): comms[r], scenes[r], tiles[r] belong to rank r. */
int p3d_gather_all(p3d_comm* const* comms, p3d_scene* const* scenes, const void* const* tiles, int n,
                   void* gathered, uint64_t tile_bytes);

/* Device memory on a scene's device for callers that have no HIP runtime of their own (the C++ host
 * layer): allocate / free, and copy from / to the host (enqueued on the scene's stream, waits for it). */
int p3d_device_alloc(p3d_scene* scene, uint64_t bytes, void** out);
int p3d_device_free(p3d_scene* scene, void* ptr);
int p3d_upload(p3d_scene* scene, void* device_dst, const void* host_src, uint64_t bytes);
int p3d_download(p3d_scene* scene, void* host_dst, const void* device_src, uint64_t bytes);

/* (Diagnostic build only -- make -C csrc stamps, libp3d_hip_stamps.so: the product library compiles the hooks out and
 * answers P3D_ERR_STATE to a non-NULL buffer.) With a device buffer of (tiles x waves-per-workgroup x 8) uint64 set here, the
 * level-1 kernel writes per-wave 100 MHz timestamps (tile start, after ray generation, after the
 * closest hit, after shading, after the queue append; slot 7 = hardware id). NULL turns it off
 * (default). Never changes results. tools/stamps.py turns them into a per-stage timeline. */
int p3d_debug_set_stamps(p3d_scene* scene, void* device_buffer);
/* Which launch of a frame writes the stamps: 1 (default) = the level-1 launch of the wavefront schedule, or the
 * single launch of the tree schedule (slots 0 = wave start, 4 = wave end); l >= 2 = the wavefront schedule's level-l
 * launch, ONE RECORD PER WAVE of that launch (buffer: waves x 8 uint64; at most 65536 waves are launched): slots
 * 0 start, 1 queue read, 2 closest hit, 3 shading (incl. shadow queries), 4 queue append / pair combine of the wave's
 * first batch, 5 wave done, 6 = number of batches it ran, 7 = hardware id. */
int p3d_debug_set_stamp_level(p3d_scene* scene, int32_t level);

/* Unit-level probe used by the parity tests: intersect n rays with one primitive each using
 * the DEVICE intersectors (Sphere/Triangle/aaBox/Plane::intercepts, RT/scene.cpp:55-283).
 * Host arrays: type[n], prim12[n*12] (plane = PN,D), origin[n*3], dir[n*3] -> hit[n], t[n],
 * normal[n*3] (getNormal(hit point).normalize()). */
int p3d_debug_intersect(int device, uint32_t n, const uint32_t* type, const float* prim12,
                        const float* origin, const float* dir, int32_t* hit, float* t,
                        float* normal);

/* Unit-level probe of the one transcendental of the path: out[i] = the device's restatement of the host C library's
 * powf(x[i], y[i]) (the Blinn-Phong exponent, RT/main.cpp:520; csrc/p3d_powf.h). Host arrays of n floats.
 * The parity tests compare it bit for bit with the box's own libm. */
int p3d_debug_powf(int device, uint32_t n, const float* x, const float* y, float* out);

/* Exhaustive check of the device's 3-instruction reciprocal (csrc/p3d_device_math.h: frcp) against the correctly rounded
 * division 1.0f / x it replaces in normalize() and Triangle::intercepts: the bit patterns first_bits .. first_bits +
 * count - 1 (count <= 2^32: all floats). n_bad = patterns whose results differ (NaN = NaN), first_bad = the lowest. */
int p3d_debug_check_rcp(int device, uint32_t first_bits, uint64_t count, uint64_t* n_bad, uint32_t* first_bad);

#ifdef __cplusplus
}
#endif
#endif /* P3D_HIP_H */
