// p3d_device_types.h -- POD records shared by the host uploader and the HIP kernels.
// Layouts are sized for 16-byte vector loads (global_load_dwordx4); DESIGN.md §"Data layout".
#ifndef P3D_DEVICE_TYPES_H
#define P3D_DEVICE_TYPES_H

#include <stdint.h>

#if defined(__HIPCC__)
#define P3D_HD __host__ __device__
#else
#define P3D_HD
#endif

namespace p3d {

// One BVH2 inner node = both children's boxes + both child references: 64 B, four
// dwordx4 loads, two slab tests per visit (2 x 32 algorithmic bytes).
// child >= 0: inner node index.  child < 0: leaf.  As the builders emit it: ~child = (first_ref << 3) | (count - 1)
// into a reference list; as uploaded (after type_leaves): a leaf reference, see kLeaf* below.
// An absent child has NaN bounds (every slab comparison is false).
struct NodePair {
    float   lo0[3]; float hi0x;
    float   hi0yz[2]; float lo1xy[2];
    float   lo1z; float hi1[3];
    int32_t child0, child1; uint32_t pad0, pad1;
};
static_assert(sizeof(NodePair) == 64, "NodePair must be 64 bytes");

// The same node pair as scenes read from HBM get it: 32 B.  Per child three dwords, one per axis, of 16-bit plane
// codes (lo | hi << 16; plane = q_base + code * q_scale, LaunchParams) and the child reference.  quantise_nodes()
// (scene_flatten.cpp) rounds lo down and hi up and adds one code of margin, so the coded box contains the f32 one.
struct QNode { uint32_t x0, y0, z0; int32_t child0; uint32_t x1, y1, z1; int32_t child1; };
static_assert(sizeof(QNode) == 32, "QNode must be 32 bytes");

// A leaf of the uploaded tree: three typed runs in primitive arrays stored in LEAF ORDER (scene_flatten.h:
// type_leaves).  Leaf 0 is the empty leaf absent children point at.
struct LeafRec { uint32_t tri_first, sph_first, box_first, counts; };         // counts: tris | spheres << 8 | boxes << 16
// Leaf references as uploaded (bit 31 set).  Bits 30..29 say what the other bits are:
//   3  ~reference = index of a LeafRec (the complement form: small indices have both bits set)
//   1  a run of triangles, 2  a run of spheres: bits 28..25 = count - 1, bits 24..0 = first index in the leaf-ordered array
//   0  never emitted (0x80000000 is the traversal's "done" marker)
constexpr uint32_t kLeafKindShift = 29, kLeafCountShift = 25, kLeafFirstMask = (1u << 25) - 1u;
constexpr uint32_t kLeafTris = 1, kLeafSpheres = 2, kLeafIndirect = 3;
static_assert(sizeof(LeafRec) == 16, "LeafRec must be one quad");

// primitive reference: kind in the top 2 bits, index into that kind's array below
constexpr uint32_t kRefKindShift = 30;
constexpr uint32_t kRefIndexMask = (1u << kRefKindShift) - 1u;

struct SphereRec { float cx, cy, cz, r; };                                   // 16 B
// Host form of a triangle.  On the device the first 48 bytes -- what an intersection test reads -- are one array with a
// 48-byte stride (LaunchParams::tri_quads = 3 quads per triangle; 4 = the 64-byte stride of round 2, kept as a comparison switch) and the shading normals another (one quad each, read once per shaded
// hit): 25 % fewer bytes under the leaf walks of scenes read from HBM than the 64-byte record of round 2.
struct TriRec { float p0[3]; uint32_t scene_id; float e1[3]; uint32_t material;
                float e2[3]; uint32_t pad;                                    // 48 B: the test record
                float n[3]; uint32_t pad2; };                                 // +16 B: getNormal().normalize()

struct BoxRec { float mn[3]; uint32_t scene_id; float mx[3]; uint32_t material; };   // 32 B
struct PlaneRec { float nx, ny, nz, d; };                                    // 16 B
struct PrimMeta { uint32_t scene_id, material; };   // spheres and planes keep ids out of line

struct MaterialRec { float diff[3]; float kd; float spec[3]; float ks;
                     float shine, T, ior, refl; };                           // 48 B
struct LightRec { float pos[3]; float pad0; float col[3]; float pad1; };     // 32 B

// ---- wavefront workspace records (one kernel per tree level, DESIGN.md §"Kernels")
// A queued ray of level d >= 2.  link = index of its parent's NodeRec at level d-1, bit 31 set
// for the refraction child (clear for the reflection child).
struct RayRec { float o[3]; float ior; float d[3]; uint32_t link; };         // 32 B
// A tree node that is waiting for children.  Children (or the resolve pass of the level below)
// overwrite refl_ret / refr_ret; both start as zero, which is what the reference adds for a
// child it never traced.  link = parent NodeRec like RayRec::link, or the compact pixel index
// for level-1 nodes.
struct NodeRec { float color[3]; float KR; float refl_ret[3]; uint32_t mat;
                 float refr_ret[3]; uint32_t link; };                        // 48 B
constexpr uint32_t kLinkRefr = 0x80000000u;
constexpr uint32_t kPairEmpty = 0x7FFFFFFFu;      // RayRec::link of the unused half of a sibling pair (LaunchParams::wf_pair_out)
struct DeviceCounters {
    unsigned long long closest_queries, shadow_queries, box_tests, sphere_tests, tri_tests,
        aabox_tests, plane_tests, pixels;
};

constexpr uint32_t kWfShards = 64;           // queue shards of the wavefront schedule (LaunchParams::wf_shards); == the wave size
constexpr uint32_t kFeatSoftJitter = 1u, kFeatFuzzy = 2u, kFeatSky = 4u;   // LaunchParams::features
constexpr uint32_t kShareDwords = 384;      // per-wave LDS of the work-sharing walk (p3d_traverse.h), behind the wave's stack slots

// Everything a render launch needs, passed by value (lands in SGPRs / kernarg segment).
struct LaunchParams {
    // scene: one blob of 16-byte quads holding every per-lane-indexed array (nodes, leaf records,
    // spheres, sphere meta, triangles, boxes, materials; section offsets in quads), so that a
    // small scene can be copied into LDS with one loop; wave-uniform arrays stay separate
    const void*        blob;
    uint32_t           blob_quads;
    const QNode*       qnodes;            // quantised node pairs: what kernels that read the scene from HBM walk
    float              q_scale[3], q_base[3];
    uint32_t           off_nodes, off_leaves, off_spheres, off_sphere_meta, off_tris, off_tri_normals, off_boxes, off_mats;
    uint32_t           tri_quads;         // quads between two triangles' test records: 3 (48-byte records, normals out of line) or 4
    int32_t            wg_waves;          // waves per workgroup of this launch (1, or 4 with an LDS scene)
    const PlaneRec*    planes;
    const PrimMeta*    plane_meta;
    const LightRec*    lights;
    uint32_t n_planes, n_lights, n_materials;
    uint32_t trav_stack_entries, trav_stack_dwords;   // walk-stack slots per lane; dwords of a wave's stack region (incl. the share region)
    uint32_t share_min_idle;          // work-sharing walk (p3d_traverse.h): idle lanes a wave must have before they take over pending subtrees
    float bg[3];
    // camera (RT/camera.h)
    float eye[3], u[3], v[3], n[3];
    float w, h, plane_dist, aperture, focal_ratio;
    int32_t res_x, res_y;
    // frame constants of Camera::PrimaryRay (RT/camera.h:93-95): u*w, v*h, n*(-plane_dist), and
    // per-column / per-row tables of (x+0.5)/res_x - 0.5 and (y+0.5)/res_y - 0.5 (device, cached)
    float uw[3], vh[3], vz[3];
    const float* ray_fx; const float* ray_fy;
    // render
    int32_t max_depth, accel, spp;
    const float* samples;          // device copy of the host sample array or nullptr
    int32_t row_block, rank, world, local_rows;
    int32_t row_block_shift;           // log2(row_block) when it is a power of two (the default 16 is), else -1: a division by a launch parameter is ~20 scalar instructions
    int32_t tiles_x, tiles_y, n_tiles, xcd_chunk, grid_blocks;
    // outputs (device)
    uint8_t* rgb8; float* rgb32f; int32_t* hit_id;
    DeviceCounters* counters;
    // wavefront pass state (set per launch by the host)
    int32_t wf_level;              // tree level this launch traces / resolves (1 = primary rays)
    int32_t wf_sample, wf_nsamples;
    int32_t wf_tile_row0, wf_tile_rows;      // band of 16x4-tile rows handled by this pass
    // The queues are split into wf_shards independent shards (tile t -> shard t % wf_shards; a
    // ray stays in its pixel's shard for its whole tree) so that the per-wave slot allocation
    // atomics spread over wf_shards counters instead of serialising on one word.  Pointers below
    // address shard 0; shard s is at + s * cap entries / + s counters.
    int32_t wf_shards;             // == kWfShards (the kernels use the constant: a modulo by a launch parameter is ~20 scalar instructions)
    uint32_t wf_cap_in, wf_cap_out, wf_ncap_parent, wf_ncap_self;    // entries per shard
    const RayRec* wf_rays_in;  const uint32_t* wf_count_in;      // level wf_level queue
    RayRec* wf_rays_out;       uint32_t* wf_count_out;           // level wf_level + 1 queue
    // Counters without a clearing launch and without host state: the two counter arrays the level-1 launch fills --
    // level-2 rays, level-1 nodes -- exist twice (wf_alt: [set][0 = rays, 1 = nodes][shard]) and the set in use
    // alternates per pass under two device words: the level-1 launch reads A = wf_ctrl[0] (nobody writes it then),
    // copies it to B = wf_ctrl[32] for the later launches of the pass, and clears the OTHER set and every deeper
    // level's counters (wf_clear: all dead since the previous pass ended); the level-2 launch sets A = 1 - B.
    uint32_t* wf_clear; uint32_t wf_clear_words;
    uint32_t* wf_alt; uint32_t* wf_ctrl;
    uint32_t dbg_skip;                // diagnostic builds only (P3D_DEBUG_SKIP): 1 = no shading after the closest hit, 2 = no shadow queries
    NodeRec* wf_nodes_parent;                                     // level wf_level - 1 nodes
    // Last level of a pass ("pair mode"): the launch of level D - 1 queues the two children of a node in an even / odd
    // slot pair (wf_pair_out; a missing child leaves a kPairEmpty slot), and the launch of level D (wf_pair_in) -- whose
    // rays all return at once, RT/main.cpp:632-634 -- combines each pair with its parent's parked record in registers and
    // hands the result to the GRANDPARENT (level D - 2 nodes, or the pixel): no resolve launch for level D - 1.
    int32_t wf_pair_in, wf_pair_out;
    NodeRec* wf_nodes_grand; uint32_t wf_ncap_grand;
    NodeRec* wf_nodes_self;    uint32_t* wf_ncount_self;         // level wf_level nodes
    float* wf_planes; uint64_t wf_plane_stride;                   // [sample][local px][3] clamped sample colours; floats per plane
    int32_t wf_min_width;            // fewest lanes a deeper-level wave may use (64 = never narrow)
    // distribution-ray-tracing features with random draws (p3d_shade.h): feature bits, frame seed, and
    // the random-stream key of every queued ray ([shard][cap] like the ray queues; nullptr when off)
    uint32_t features, seed;
    const uint32_t* wf_rng_in; uint32_t* wf_rng_out;
    // cube map of P3D_FEATURE_SKYBOX (Scene::skybox_img, RT/scene.h:190-195): face i starts at sky + sky_off[i]
    const uint8_t* sky; uint32_t sky_off[6], sky_w[6], sky_h[6], sky_bpp[6];
    unsigned long long* dbg_stamps;   // diagnostic: per (tile, wave) 8 x u64 timestamps, or nullptr
    int32_t dbg_stamp_level;          // which launch of a frame writes them: <= 1 the level-1 / tree / tile launch, l >= 2 the level-l launch
    // ---- uniform grid of GRID mode (accel 1; RT/grid.cpp): cell c holds grid_items[grid_cells[c] .. grid_cells[c+1])
    // = primitive refs (kind << 30 | index, planes kind 3) in scene order; nullptr until a GRID frame is asked for
    const uint32_t* grid_cells; const uint32_t* grid_items;
    int32_t grid_n[3]; float grid_min[3], grid_max[3];
    // ---- tile schedule (wf_tile_kernel): ONE launch per frame.  Persistent 256-thread workgroups draw
    // 16x16-pixel tiles from tw_ctrl[0] and run a tile's whole ray tree level by level among themselves;
    // every queue of a tile lives in the workgroup's private slot of the workspace (slot = blockIdx.x),
    // its counters in LDS.  tw_ctrl[1] counts workgroups that have finished: the last one resets both.
    uint8_t*  tw_base; uint64_t tw_slot_bytes;
    uint32_t* tw_ctrl;
    uint32_t  tw_rays_off, tw_nodes_off, tw_rng_off;     // byte offsets of the three regions inside a slot
    // the k-th tile a workgroup draws is tile_order[k] (nullptr: k); tile_cost[tile] = how long the tile took, in 10 ns
    // ticks, for the next ordering (bvh_device.hip: sort_tiles_by_cost).  Scenes read from HBM only.
    const uint32_t* tile_order; uint32_t* tile_cost;
};

// Every level of a pass for the fused resolve launch (small frames / shards, wf_resolve_fused_kernel): level l's
// parked nodes (shard s at nodes[l] + s * cap[l]) and their per-shard counts; levels top .. 1 are combined.
struct ResolveLevels { NodeRec* nodes[18]; const uint32_t* ncount[18]; uint32_t cap[18]; int32_t top; };

// tile schedule geometry: a tile is 16 x 16 pixels = one 256-thread workgroup = 4 waves of 16 x 4
constexpr uint32_t kTilePx = 256;
constexpr int kMaxTileLevels = 18;        // max_depth <= 16: levels 0..17 index the per-tile counters
// level l (1-based) of a tile's tree holds at most kTilePx << (l-1) rays / nodes; levels are packed back to back:
// ray queues start at level 2, node arrays at level 1
P3D_HD inline uint32_t tile_ray_offset(int l) { return kTilePx * ((1u << (l - 1)) - 2u); }
P3D_HD inline uint32_t tile_node_offset(int l) { return kTilePx * ((1u << (l - 1)) - 1u); }
P3D_HD inline uint64_t tile_ray_entries(int D) { return D >= 2 ? (uint64_t)kTilePx * ((1ull << D) - 2ull) : 0; }
P3D_HD inline uint64_t tile_node_entries(int D) { return D >= 2 ? (uint64_t)kTilePx * ((1ull << (D - 1)) - 1ull) : 0; }

}  // namespace p3d
#endif
