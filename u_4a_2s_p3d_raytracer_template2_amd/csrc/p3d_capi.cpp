// p3d_capi.cpp -- implementation of the C-ABI declared in include/p3d_hip.h.
// Host side only: flattens the caller's scene into the device records, builds the BVH
// (bvh_builder.cpp), owns device memory and enqueues the kernels of p3d_kernels.hip.
#include "p3d_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "bvh_builder.h"
#include "grid_builder.h"
#include "p3d_device_types.h"
#include "scene_flatten.h"
#define P3D_POWF_TABLES_ONLY
#include "p3d_powf.h"

namespace p3d {
size_t tree_kernel_lds_bytes(const LaunchParams& P, bool lds);
size_t wavefront_lds_bytes(const LaunchParams& P, bool lds);
hipError_t launch_tree(const LaunchParams& P, bool count, bool lds, int occ, bool shared, hipStream_t stream);
hipError_t launch_wf_primary(const LaunchParams& P, bool count, bool lds, int walk, int occ, hipStream_t stream);
hipError_t launch_wf_secondary(const LaunchParams& P, bool count, bool lds, int walk, int occ, unsigned waves,
                               hipStream_t stream);
hipError_t launch_wf_resolve(const LaunchParams& P, unsigned blocks, hipStream_t stream);
hipError_t launch_wf_resolve_fused(const LaunchParams& P, const ResolveLevels& R, unsigned shards, hipStream_t stream);
hipError_t launch_clear_words(uint32_t* p, uint32_t n, hipStream_t stream);
hipError_t wf_resident_waves(const LaunchParams& P, bool primary, bool count, bool lds, int walk, int occ, unsigned* waves);
size_t tile_kernel_lds_bytes(const LaunchParams& P, bool lds);
hipError_t tile_kernel_resident_blocks(const LaunchParams& P, bool count, bool lds, int walk, int occ, int* blocks);
hipError_t launch_wf_tile(const LaunchParams& P, bool count, bool lds, int walk, int occ, unsigned blocks, hipStream_t stream);
hipError_t launch_raygen_table(float* fx, float* fy, int res_x, int res_y, hipStream_t stream);
hipError_t prepare_kernels(size_t max_lds);
hipError_t launch_sum_samples(const LaunchParams& P, size_t first_px, size_t n_px, hipStream_t stream);
hipError_t launch_deinterleave(const void* gathered, void* frames, int res_x, int res_y, int row_block,
                               int world, size_t rank_stride, int bpp, int n_frames, size_t in_stride,
                               size_t out_stride, hipStream_t stream);
hipError_t build_lbvh_device(const std::vector<BuildPrim>& prims, const BvhOptions& opt, NodePair* d_nodes,
                             uint32_t* d_refs, BvhStats& stats, hipStream_t stream);
hipError_t sort_tiles_by_cost(const uint32_t* cost, uint32_t* cost_sorted, uint32_t* iota, uint32_t* order, uint32_t n,
                              void* temp, size_t& temp_bytes, hipStream_t stream);
bool kernels_have_stamps();
hipError_t launch_debug_check_rcp(uint32_t first, uint64_t count, unsigned long long* n_bad, uint32_t* first_bad, hipStream_t stream);
hipError_t launch_debug_powf(uint32_t n, const float* x, const float* y, float* out, hipStream_t stream);
hipError_t launch_debug_intersect(uint32_t n, const uint32_t* type, const float* prim12, const float* origin,
                                  const float* dir, int32_t* hit, float* t, float* normal, hipStream_t stream);
}  // namespace p3d

using namespace p3d;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(P3D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));   \
    } while (0)

constexpr size_t kMaxLdsBytes = 160 * 1024;   // gfx950: 160 KiB per CU
constexpr int kMaxDepth = 16;
constexpr size_t kLdsSceneLimit = 24 * 1024;
#ifndef P3D_HBM_STACK_DWORDS
#define P3D_HBM_STACK_DWORDS 64u      // per entry per wave: RefStack 64 (4-byte entries; p3d_traverse.h), SlimStack 96
#endif
constexpr uint32_t kHbmStackDwordsPerEntry = P3D_HBM_STACK_DWORDS;
constexpr int kLanes = 4;                             // concurrent sample passes of one frame
constexpr unsigned kPersistentWavesNarrow = 256 * 16 * 4;   // ... of HBM-resident scenes (narrow waves, 4 per SIMD)

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t upload(const std::vector<T>& h) {
        n = h.size();
        size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
        hipError_t e = hipMalloc((void**)&p, bytes);
        if (e != hipSuccess) return e;
        if (n) e = hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
    size_t bytes() const { return std::max<size_t>(n, 1) * sizeof(T); }
};

struct RawBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct p3d_scene {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    DevBuf<uint32_t> blob;              // powf tables (32 quads) | leaf records | spheres | sphere meta | tris | boxes | materials [| f32 nodes: LDS scenes]
    DevBuf<QNode> qnodes;               // 32-byte node pairs: what kernels that read the scene from HBM walk
    float q_scale[3] = {1, 1, 1}, q_base[3] = {0, 0, 0};
    uint32_t blob_quads = 0;
    uint32_t off_nodes = 0, off_leaves = 0, off_spheres = 0, off_sphere_meta = 0, off_tris = 0, off_tri_normals = 0, off_boxes = 0, off_mats = 0;
    DevBuf<PlaneRec> planes;
    DevBuf<PrimMeta> plane_meta;
    DevBuf<LightRec> lights;
    // GRID mode (accel 1): the reference's uniform grid, built from grid_src on the first GRID frame
    std::vector<GridPrim> grid_src;
    DevBuf<uint32_t> grid_cells, grid_items;
    GridHost grid_info; bool grid_ready = false;
    DevBuf<LightRec> soft_lights;          // 16 sub-lights per light, built on first use (SOFT_SHADOW, spp == 0)
    DevBuf<uint8_t> sky;                   // cube map of P3D_FEATURE_SKYBOX: the six faces back to back
    uint32_t sky_off[6] = {0, 0, 0, 0, 0, 0}, sky_w[6] = {0, 0, 0, 0, 0, 0}, sky_h[6] = {0, 0, 0, 0, 0, 0}, sky_bpp[6] = {0, 0, 0, 0, 0, 0};
    std::vector<LightRec> host_lights;
    size_t lds_scene_limit = kLdsSceneLimit;  // blobs up to this size are rendered from an LDS copy
    bool lds_capable = false;            // ... and then carry the f32 nodes the LDS walk reads
    int last_schedule = -1;
    bool unit_rays_only = false;         // built with cull_never_hit: cannot serve un-normalised (NONE-mode) shadow rays
    uint32_t packet_node_limit = 64;     // trees up to this many node pairs use the wave-wide walk
    float bg[3] = {0, 0, 0};
    uint32_t n_lights = 0, n_materials = 0;
    p3d_scene_stats stats{};
    RawBuf fb_rgb8, fb_rgb32f, fb_hit, samples;
    RawBuf ray_tab; int tab_res_x = 0, tab_res_y = 0;   // cached per-column / per-row ray factors
    // wavefront workspace: ray queues (levels 2..D), parked nodes (levels 1..D-1), counters
    // Wavefront workspaces.  A frame of several sample passes (spp > 0) runs up to kLanes passes at a
    // time, each on its own stream with its own queues: the latency-bound deep levels and launch tails
    // of one pass fill with another pass's work, like independent frames do.  Lane 0 is the scene's stream.
    struct Workspace {
        RawBuf rays[kMaxDepth + 2], nodes[kMaxDepth + 2], counts;   // counts: cleared by a kernel in front of every pass
        RawBuf rng[kMaxDepth + 2];           // random-stream keys of the queued rays (stochastic features)
        size_t held() const {
            size_t b = 0;
            for (auto& q : rays) b += q.cap;
            for (auto& q : nodes) b += q.cap;
            for (auto& q : rng) b += q.cap;
            return b;
        }
        void release() {
            for (auto& b : rays) b.release();
            for (auto& b : nodes) b.release();
            for (auto& b : rng) b.release();
            counts.release();
        }
    } ws[kLanes];
    RawBuf wf_planes;                        // [sample][local px][3] clamped sample colours (spp > 0)
    // tile schedule: (resident workgroups) x (one 16x16 tile's worst-case queues), and the tile counter + exit
    // ticket the kernel re-arms itself (zeroed once, at allocation)
    RawBuf tile_ws, tile_ctrl;
    // "heaviest tile first" for scenes read from HBM: per-tile durations written by the tile kernel, and the order made of
    // them after the first frame of a configuration and every kTileOrderPeriod frames from then on (all on the frame's stream)
    struct TileOrder {
        RawBuf cost, sorted, iota, order, temp;
        size_t temp_bytes = 0;
        int32_t key[9] = {0, 0, 0, -1, -1, 0, 0, 0, 0};
        bool valid = false;          // `order` holds an order for `key`
        int frames = 0;              // frames rendered since it was made
        void release() { cost.release(); sorted.release(); iota.release(); order.release(); temp.release(); }
    } tile_lpt, wave_lpt;            // 16x16 tiles of the tile schedule / 16x4 wave tiles of the tree and wavefront level-1 launches
    bool tile_lpt_enabled = true;
    struct { uint32_t key = 0xFFFFFFFFu; size_t lds = 0; int blocks = 0; } tile_occ[2];   // cached occupancy queries (private / shared walk)
    struct { uint32_t key = 0xFFFFFFFFu; uint32_t stack = 0; unsigned waves = 0, primary_waves = 0; } wf_occ;                 // ... of the deeper-level kernel
    hipStream_t lane_stream[kLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kLanes] = {nullptr, nullptr, nullptr, nullptr};
    // upper limit of the workspace one frame may allocate (wavefront schedule: worst-case level queues of a band
    // of tile rows; tile schedule: one slot per resident workgroup).  64 GiB holds BASELINE config 4's wavefront
    // queues (4096^2, depth 6: 58 GB worst case) in one band: 10.7 -> 9.3 ms against 8 GiB.
    size_t workspace_budget = (size_t)64 << 30;
    // what of that budget this device can actually give: re-read (hipMemGetInfo) whenever the budget or the frame
    // configuration changes, so that several scene handles, or a framework holding most of the HBM, shrink the
    // bands / fall back to another schedule instead of failing in hipMalloc
    size_t budget_avail = 0; int32_t budget_key[4] = {0, 0, 0, -1};
    DeviceCounters* d_counters = nullptr;
    bool counters_valid = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_prof[4] = {nullptr, nullptr, nullptr, nullptr};   // frame begin/end, dominant kernel begin/end
    // Schedule choice: measured, not guessed.  The first frames of a (resolution, depth, accel, spp, shard, flags)
    // configuration run every available schedule twice, the second time bracketed by HIP events; later frames
    // use the fastest one.  All produce identical bits, so the choice is invisible in the output.
    struct SchedulePick {
        int32_t key[8] = {0, 0, 0, -1, -1, 0, 0, 0};
        float ms[6] = {-1.0f, -1.0f, -1.0f, -1.0f, -1.0f, -1.0f};     // wavefront, tree, tile with shared walks; the same with private walks
        int pending = -1, step = 0, best = 2;
    } pick;
    // p3d_tune_schedule(): a candidate forced for the frames it times, the winner to adopt at the next frame, and what the
    // most recent frame's measured choice had to choose from (0: the choice is made by rule or by a flag)
    int tune_force = -1, tune_commit = -1, tune_candidates = 0;
    uint32_t tune_avail = 0;
    hipEvent_t ev_pick[2] = {nullptr, nullptr};
    bool profile_valid = false;
    bool timer_open = false;
    size_t lds_prepared = 0;
    int xcd_chunk = 1;
    int frame_streams = 1;               // bands of a one-sample frame run concurrently on this many streams (experiment knob)
    int resolve_blocks_per_shard = 16;   // a resolve launch is latency-bound: few nodes per thread, many threads
    int fused_resolve_shard_px = 8192;   // frames with at most this many pixels per shard resolve all levels in one launch
    bool pair_mode = true;               // the last level combines sibling rays with their parent (LaunchParams::wf_pair_in)
    uint32_t dbg_skip = 0;               // diagnostic builds only (LaunchParams::dbg_skip)
    unsigned long long* dbg_stamps = nullptr; int dbg_stamp_level = 1;
    int occupancy = 0;     // 0 = compiler default register budget, else 5 / 6 / 8 waves per SIMD
    uint32_t tri_quads = 3;    // 16-byte quads per triangle test record (3; 4 = round 2's 64-byte stride, P3D_TRI_STRIDE=64)
    bool verbose = false;      // P3D_VERBOSE=1: launch geometry on stderr (diagnostic)
    int share_min_idle = 16;   // work-sharing walk of scenes read from HBM: idle lanes before a steal round (0 or > 64: private walks)
};

extern "C" int p3d_internal_set_error(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
// device and stream a scene is bound to (for p3d_comm.cpp)
extern "C" int p3d_internal_scene_binding(p3d_scene* s, int* device, void** stream) {
    if (!s) return fail(P3D_ERR_ARG, "scene is NULL");
    *device = s->device; *stream = (void*)s->stream;
    return P3D_OK;
}

extern "C" {

int p3d_abi_version(void) { return P3D_ABI_VERSION; }
const char* p3d_last_error(void) { return g_err.c_str(); }

int p3d_device_count(int* count) {
    if (!count) return fail(P3D_ERR_ARG, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(P3D_ERR_NO_DEVICE, hipGetErrorString(e)); }
    *count = n;
    return P3D_OK;
}

int p3d_local_rows(int32_t res_y, int32_t row_block, int32_t world) {
    if (row_block <= 0) row_block = 16;
    if (world <= 0) world = 1;
    int nblocks = (res_y + row_block - 1) / row_block;
    return ((nblocks + world - 1) / world) * row_block;
}

int p3d_scene_create(const p3d_scene_desc* d, const p3d_build_opts* opts, int device, p3d_scene** out) {
    if (!d || !out) return fail(P3D_ERR_ARG, "desc/out is NULL");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(P3D_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(P3D_ERR_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    FlatScene F;
    std::string why = flatten_scene(*d, F);
    if (!why.empty()) return fail(why == "too many primitives" ? P3D_ERR_LIMIT : P3D_ERR_ARG, why);
    BvhOptions bo;
    if (opts) {
        if (opts->leaf_max) bo.leaf_max = std::min<uint32_t>(opts->leaf_max, 8);
        if (opts->sah_bins) bo.bins = opts->sah_bins;
    }
    // optional: triangles no ray of length <= sqrt(2) can hit (see p3d_build_opts::cull_never_hit).
    // Bound on the FLOAT determinant the reference computes: |det| <= |d| (|e1 x e2| + 1e-6 |e1| |e2|)
    // (products and sums of RT/scene.cpp:64-65 each round once, 6e-8 relative; 1e-6 covers them all).
    uint32_t n_culled = 0;
    if (opts && opts->cull_never_hit) {
        std::vector<BuildPrim> kept;
        kept.reserve(F.build_prims.size());
        for (const BuildPrim& b : F.build_prims) {
            bool never = false;
            if ((b.ref >> kRefKindShift) == 1u) {
                const TriRec& t = F.tris[b.ref & kRefIndexMask];
                const double e1[3] = {t.e1[0], t.e1[1], t.e1[2]}, e2[3] = {t.e2[0], t.e2[1], t.e2[2]};
                const double cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
                const double cross = std::sqrt(cx * cx + cy * cy + cz * cz);
                const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]);
                const double l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
                never = 1.41422 * (cross + 1e-6 * l1 * l2) < 0.999e-3;
            }
            if (never) n_culled++; else kept.push_back(b);
        }
        F.build_prims.swap(kept);
    }
    std::vector<NodePair> nodes; std::vector<uint32_t> refs; BvhStats bs;
    if (opts && opts->builder > 1) return fail(P3D_ERR_ARG, "builder must be 0 (host SAH) or 1 (device LBVH)");
    // the device builder needs at least two leaves; tiny scenes are built on the host either way
    const bool device_build = opts && opts->builder == 1 && F.build_prims.size() >= 64;
    if (device_build) {
        // built on the device into scratch buffers, read back: the leaves are typed and the primitive arrays put
        // into leaf order on the host (type_leaves) before anything is uploaded for rendering
        nodes.assign((F.build_prims.size() + 1) / 2 - 1, NodePair());
        refs.assign(F.build_prims.size(), 0u);
        NodePair* d_nodes = nullptr; uint32_t* d_refs = nullptr;
        hipError_t be = hipMalloc((void**)&d_nodes, nodes.size() * sizeof(NodePair));
        if (be == hipSuccess) be = hipMalloc((void**)&d_refs, refs.size() * sizeof(uint32_t));
        if (be == hipSuccess) be = build_lbvh_device(F.build_prims, bo, d_nodes, d_refs, bs, nullptr);
        if (be == hipSuccess) be = hipMemcpy(nodes.data(), d_nodes, nodes.size() * sizeof(NodePair), hipMemcpyDeviceToHost);
        if (be == hipSuccess) be = hipMemcpy(refs.data(), d_refs, refs.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
        (void)hipFree(d_nodes); (void)hipFree(d_refs);
        if (be != hipSuccess) return fail(P3D_ERR_HIP, std::string("device BVH build: ") + hipGetErrorString(be));
    } else {
        build_bvh(F.build_prims, bo, nodes, refs, bs);
    }
    TypedLeaves TL;
    // Scenes small enough to be rendered from an LDS copy keep a record per leaf; the others name single-type leaves
    // in the reference itself (p3d_traverse.h: sv_leaf).  Upper bound of the blob with a record per leaf:
    const size_t blob_bound = nodes.size() * (sizeof(NodePair) + 2 * sizeof(LeafRec)) + 16 + F.spheres.size() * (sizeof(SphereRec) + sizeof(PrimMeta)) +
                              F.tris.size() * sizeof(TriRec) + F.boxes.size() * sizeof(BoxRec) + F.materials.size() * sizeof(MaterialRec) + 8 * 16 + P3D_POW_TAB_BYTES;
    const bool small_scene = blob_bound <= kLdsSceneLimit;
    type_leaves(nodes, refs, F, TL, !small_scene);
    if (TL.overflow) return fail(P3D_ERR_LIMIT, "too many mixed-type leaves");
    std::vector<SphereRec>& spheres = F.spheres; std::vector<PrimMeta>& sphere_meta = F.sphere_meta;
    std::vector<TriRec>& tris = F.tris; std::vector<BoxRec>& boxes = F.boxes;
    std::vector<PlaneRec>& planes = F.planes; std::vector<PrimMeta>& plane_meta = F.plane_meta;
    std::vector<MaterialRec>& mats = F.materials; std::vector<LightRec>& lights = F.lights;

    p3d_scene* s = new p3d_scene();
    s->device = device;
    if (const char* e = getenv("P3D_FRAME_STREAMS")) { int v = atoi(e); if (v >= 1 && v <= kLanes) s->frame_streams = v; }
    if (const char* e = getenv("P3D_RESOLVE_BLOCKS")) { int v = atoi(e); if (v >= 1 && v <= 64) s->resolve_blocks_per_shard = v; }   // tuning experiments
    if (const char* e = getenv("P3D_FUSED_RESOLVE_PX")) { int v = atoi(e); if (v >= 0 && v <= (1 << 24)) s->fused_resolve_shard_px = v; }
    s->pair_mode = getenv("P3D_NO_PAIR_MODE") == nullptr;
    s->verbose = getenv("P3D_VERBOSE") != nullptr;
    if (const char* e = getenv("P3D_TILE_LPT")) s->tile_lpt_enabled = atoi(e) != 0;
    if (const char* e = getenv("P3D_OCC")) { int v = atoi(e); if (v == 0 || v == 5 || v == 6) s->occupancy = v; }
    if (const char* e = getenv("P3D_TRI_STRIDE")) s->tri_quads = atoi(e) == 64 ? 4u : 3u;
    if (const char* e = getenv("P3D_SHARE_MIN_IDLE")) { int v = atoi(e); if (v >= 0 && v <= 65) s->share_min_idle = v; }
    if (const char* e = getenv("P3D_DEBUG_SKIP")) s->dbg_skip = (uint32_t)atoi(e);      // read by -DP3D_DEBUG_SKIP builds only
    auto bail = [&](hipError_t e, const char* what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        p3d_scene_destroy(s);
        return fail(P3D_ERR_HIP, msg);
    };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    s->stream = s->own_stream;
    if ((e = hipEventCreate(&s->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&s->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    for (auto& ev : s->ev_prof) if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
    for (auto& ev : s->ev_pick) if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    for (int i = 1; i < kLanes; i++) {
        if ((e = hipStreamCreateWithFlags(&s->lane_stream[i], hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
        if ((e = hipEventCreateWithFlags(&s->ev_join[i], hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    }
    {   // pack the per-lane-indexed arrays into one blob of 16-byte quads
        std::vector<uint32_t> blob;
        auto section = [&](const void* data, size_t bytes) {
            uint32_t off = (uint32_t)(blob.size() / 4);
            size_t dw = (bytes + 15) / 16 * 4;
            size_t at = blob.size();
            blob.resize(at + std::max<size_t>(dw, 4), 0u);
            if (bytes) memcpy(blob.data() + at, data, bytes);
            return off;
        };
        {   // powf's tables lead the blob: kernels that render from an LDS copy of it read them at LDS address 0 (p3d_powf.h)
            static const double log2_tab[16][2] = P3D_POW_LOG2_TAB_INIT;
            static const uint64_t exp2_tab[32] = P3D_POW_EXP2_TAB_INIT;
            static const double coefs[10] = P3D_POW_COEF_INIT;
            static_assert(sizeof log2_tab + sizeof exp2_tab + sizeof coefs == P3D_POW_TAB_BYTES, "powf table layout");
            unsigned char tab[P3D_POW_TAB_BYTES];
            memcpy(tab, log2_tab, sizeof log2_tab);
            memcpy(tab + sizeof log2_tab, exp2_tab, sizeof exp2_tab);
            memcpy(tab + sizeof log2_tab + sizeof exp2_tab, coefs, sizeof coefs);
            if (section(tab, sizeof tab) != 0u) return bail(hipErrorUnknown, "scene blob layout");
        }
        s->off_leaves = section(TL.leaves.data(), TL.leaves.size() * sizeof(LeafRec));
        s->off_spheres = section(spheres.data(), spheres.size() * sizeof(SphereRec));
        s->off_sphere_meta = section(sphere_meta.data(), sphere_meta.size() * sizeof(PrimMeta));
        {   // triangles: 48-byte test records, shading normals out of line (p3d_device_types.h: TriRec)
            const uint32_t tq = s->tri_quads;
            std::vector<uint32_t> test(tris.size() * 4 * tq), nrm(tris.size() * 4);
            for (size_t i = 0; i < tris.size(); i++) {
                memcpy(test.data() + 4 * tq * i, &tris[i], 16 * tq);
                memcpy(nrm.data() + 4 * i, tris[i].n, 12);
            }
            s->off_tris = section(test.data(), test.size() * 4);
            s->off_tri_normals = section(nrm.data(), nrm.size() * 4);
        }
        s->off_boxes = section(boxes.data(), boxes.size() * sizeof(BoxRec));
        s->off_mats = section(mats.data(), mats.size() * sizeof(MaterialRec));
        // the f32 nodes only travel with scenes small enough to be rendered from an LDS copy of the blob
        if (small_scene && blob.size() * 4 + nodes.size() * sizeof(NodePair) <= s->lds_scene_limit) {
            s->off_nodes = section(nodes.data(), nodes.size() * sizeof(NodePair));
            s->lds_capable = true;
        }
        s->blob_quads = (uint32_t)(blob.size() / 4);
        if ((e = s->blob.upload(blob)) != hipSuccess) return bail(e, "upload scene blob");
    }
    {
        QuantisedNodes Q;
        quantise_nodes(nodes, Q);
        if ((e = s->qnodes.upload(Q.nodes)) != hipSuccess) return bail(e, "upload nodes");
        memcpy(s->q_scale, Q.scale, sizeof s->q_scale); memcpy(s->q_base, Q.base, sizeof s->q_base);
    }
    if ((e = s->planes.upload(planes)) != hipSuccess) return bail(e, "upload planes");
    if ((e = s->plane_meta.upload(plane_meta)) != hipSuccess) return bail(e, "upload plane meta");
    if ((e = s->lights.upload(lights)) != hipSuccess) return bail(e, "upload lights");
    s->host_lights = lights;
    grid_prims_from_desc(*d, s->grid_src);
    for (GridPrim& g : s->grid_src) {             // references in the uploaded (leaf-order) numbering
        const uint32_t kind = g.ref >> kRefKindShift, idx = g.ref & kRefIndexMask;
        if (kind == 0u) g.ref = (0u << kRefKindShift) | TL.map_sph[idx];
        else if (kind == 1u) g.ref = (1u << kRefKindShift) | TL.map_tri[idx];
        else if (kind == 2u) g.ref = (2u << kRefKindShift) | TL.map_box[idx];
    }
    if ((e = hipMalloc((void**)&s->d_counters, sizeof(DeviceCounters))) != hipSuccess) return bail(e, "alloc counters");
    if ((e = hipMemset(s->d_counters, 0, sizeof(DeviceCounters))) != hipSuccess) return bail(e, "clear counters");
    memcpy(s->bg, d->background, sizeof s->bg);
    s->n_lights = d->n_lights; s->n_materials = d->n_materials;
    s->stats.n_nodes = bs.n_nodes; s->stats.n_leaves = bs.n_leaves; s->stats.max_depth = bs.max_depth;
    s->stats.n_leaf_refs = bs.n_leaf_refs; s->stats.sah_cost = bs.sah_cost;
    s->stats.n_spheres = (uint32_t)spheres.size(); s->stats.n_triangles = (uint32_t)tris.size();
    s->stats.n_boxes = (uint32_t)boxes.size(); s->stats.n_planes = (uint32_t)planes.size();
    s->stats.n_culled = n_culled;
    s->unit_rays_only = n_culled > 0;
    s->stats.device_bytes = s->blob.bytes() + s->qnodes.bytes() + s->planes.bytes() + s->plane_meta.bytes() + s->lights.bytes();
    *out = s;
    return P3D_OK;
}

int p3d_scene_destroy(p3d_scene* s) {
    if (!s) return P3D_OK;
    (void)hipSetDevice(s->device);
    if (s->own_stream) (void)hipStreamSynchronize(s->own_stream);
    s->grid_cells.release(); s->grid_items.release();
    s->blob.release(); s->qnodes.release(); s->planes.release(); s->plane_meta.release(); s->lights.release(); s->soft_lights.release(); s->sky.release();
    s->fb_rgb8.release(); s->fb_rgb32f.release(); s->fb_hit.release(); s->samples.release(); s->ray_tab.release();
    for (auto& w : s->ws) w.release();
    s->wf_planes.release(); s->tile_ws.release(); s->tile_ctrl.release();
    s->tile_lpt.release(); s->wave_lpt.release();
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    for (int i = 1; i < kLanes; i++) {
        if (s->ev_join[i]) (void)hipEventDestroy(s->ev_join[i]);
        if (s->lane_stream[i]) { (void)hipStreamSynchronize(s->lane_stream[i]); (void)hipStreamDestroy(s->lane_stream[i]); }
    }
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (auto& ev : s->ev_prof) if (ev) (void)hipEventDestroy(ev);
    for (auto& ev : s->ev_pick) if (ev) (void)hipEventDestroy(ev);
    if (s->own_stream) (void)hipStreamDestroy(s->own_stream);
    delete s;
    return P3D_OK;
}

int p3d_scene_set_skybox(p3d_scene* s, const uint8_t* const faces[6], const uint32_t res_x[6], const uint32_t res_y[6],
                         const uint32_t bytes_per_pixel[6]) {
    if (!s || !faces || !res_x || !res_y || !bytes_per_pixel) return fail(P3D_ERR_ARG, "NULL argument");
    std::vector<uint8_t> all;
    uint32_t off[6];
    for (int i = 0; i < 6; i++) {
        if (!faces[i] || res_x[i] == 0 || res_y[i] == 0 || res_x[i] > 16384 || res_y[i] > 16384 || (bytes_per_pixel[i] != 3 && bytes_per_pixel[i] != 4))
            return fail(P3D_ERR_ARG, "skybox faces must be 1..16384 pixels wide and high, 3 or 4 bytes per pixel");
        const size_t bytes = (size_t)res_x[i] * res_y[i] * bytes_per_pixel[i];
        if (all.size() + bytes > 0xFFFFFFF0ull) return fail(P3D_ERR_LIMIT, "skybox too large");
        off[i] = (uint32_t)all.size();
        all.insert(all.end(), faces[i], faces[i] + bytes);
        all.resize((all.size() + 15) / 16 * 16);
    }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));              // (frames in flight may still read the old map)
    s->sky.release();
    HIP_TRY(s->sky.upload(all));
    for (int i = 0; i < 6; i++) { s->sky_off[i] = off[i]; s->sky_w[i] = res_x[i]; s->sky_h[i] = res_y[i]; s->sky_bpp[i] = bytes_per_pixel[i]; }
    return P3D_OK;
}

int p3d_scene_get_stats(const p3d_scene* s, p3d_scene_stats* out) {
    if (!s || !out) return fail(P3D_ERR_ARG, "scene/out is NULL");
    *out = s->stats;
    return P3D_OK;
}

int p3d_set_stream(p3d_scene* s, void* hip_stream) {
    if (!s) return fail(P3D_ERR_ARG, "scene is NULL");
    s->stream = hip_stream ? (hipStream_t)hip_stream : s->own_stream;
    return P3D_OK;
}

int p3d_set_tuning(p3d_scene* s, int32_t xcd_chunk, int32_t workspace_mib, int32_t waves_per_simd) {
    if (!s) return fail(P3D_ERR_ARG, "scene is NULL");
    if (xcd_chunk < 0 || xcd_chunk > (1 << 20)) return fail(P3D_ERR_ARG, "xcd_chunk must be >= 0");
    if (workspace_mib < 0) return fail(P3D_ERR_ARG, "workspace_mib must be >= 0");
    s->pick.key[0] = 0;                                                // tuning changes what the schedules cost: measure again
    if (xcd_chunk) s->xcd_chunk = xcd_chunk;
    if (workspace_mib) { s->workspace_budget = (size_t)workspace_mib << 20; s->budget_key[0] = 0; }
    if (waves_per_simd >= 0) {
        if (waves_per_simd != 0 && waves_per_simd != 5 && waves_per_simd != 6)
            return fail(P3D_ERR_ARG, "waves_per_simd must be 0 (default), 5 or 6");
        s->occupancy = waves_per_simd;
    }
    return P3D_OK;
}

namespace {

// worst-case workspace bytes per pixel for a depth-D tree: level l holds at most 2^(l-1) rays
size_t wavefront_bytes_per_pixel(int D) {
    size_t b = 0;
    for (int l = 2; l <= D; l++) b += ((size_t)1 << (l - 1)) * sizeof(RayRec);
    for (int l = 1; l <= D - 1; l++) b += ((size_t)1 << (l - 1)) * sizeof(NodeRec);
    return b;
}

constexpr int kShards = 64;   // queue shards; spreads the slot-allocation atomics. == the wave size: the deeper-level
                              // kernel holds one shard's count per lane (wf_secondary_kernel)
// counter buffer of a workspace: [level][shard] ray counts, node counts (all cleared by the first launch of a pass),
// then the two alternating level-1 sets and the parity words (LaunchParams::wf_alt)
constexpr size_t kCountWords = (size_t)2 * (kMaxDepth + 2) * kShards;
constexpr size_t kCountBufferWords = kCountWords + 4 * kShards + 64;

// One sample pass over one band of tile rows, level by level (see p3d_kernels.hip).
// shard_px = pixels a shard can own in this band (worst case), so level l holds at most
// shard_px << (l-1) rays / nodes per shard.
int run_wavefront_pass(p3d_scene* s, p3d_scene::Workspace& ws, hipStream_t stream, LaunchParams P, bool count, bool lds,
                       int walk, size_t shard_px, bool profile) {
    const int D = P.max_depth;
    // scenes read from HBM wait on fetches most of the time: a register budget of 6 waves per SIMD measured 3 %
    // faster than the compiler's default there (10^6 primitives 3.19 -> 3.08 ms); LDS scenes keep the default
    // ... and since round 3 the LDS scenes' level kernels run at 6 too: the deeper-level kernel needs 82 VGPRs by default
    // (5 waves per SIMD), 78 under that budget without a spill -- config 2 0.0720 -> 0.0692 ms/frame, config 4 on this
    // schedule 5.30 -> 5.05 ms (profiles/r03_exp10_register_budgets.txt)
    const int occ = s->occupancy ? s->occupancy : 6;
    const size_t n_counts = kCountWords;
    uint32_t* counts = (uint32_t*)ws.counts.p;                              // [level][shard] ray counts, then node counts
    // No clearing launch and nothing about a frame in host state (a captured frame can be replayed any number of
    // times): the level-1 launch zeroes what the previous pass left, under a device-side parity -- see
    // LaunchParams::wf_alt.  The buffer is zeroed once, when it is allocated.
    P.wf_clear = counts; P.wf_clear_words = (uint32_t)n_counts;
    P.wf_alt = counts + kCountWords; P.wf_ctrl = counts + kCountWords + 4 * kShards;
    auto rays = [&](int l) { return (l >= 2 && l <= D) ? (RayRec*)ws.rays[l].p : nullptr; };
    auto nodes = [&](int l) { return (l >= 1 && l <= D - 1) ? (NodeRec*)ws.nodes[l].p : nullptr; };
    auto qcount = [&](int l) { return counts + (size_t)l * kShards; };
    auto ncount = [&](int l) { return counts + (size_t)(kMaxDepth + 2 + l) * kShards; };
    auto cap = [&](int l) { return (uint32_t)(shard_px << (l - 1)); };
    P.wf_shards = kShards;
    P.wf_level = 1;
    P.wf_rays_in = nullptr; P.wf_count_in = nullptr; P.wf_cap_in = 0;
    P.wf_rays_out = rays(2); P.wf_count_out = qcount(2); P.wf_cap_out = cap(2);
    auto rng = [&](int l) { return (P.features && l >= 2 && l <= D) ? (uint32_t*)ws.rng[l].p : nullptr; };
    P.wf_rng_in = nullptr; P.wf_rng_out = rng(2);
    P.wf_nodes_parent = nullptr; P.wf_ncap_parent = 0;
    P.wf_nodes_self = nodes(1); P.wf_ncount_self = ncount(1); P.wf_ncap_self = cap(1);
    {   // persistent grids: as many waves as can be resident (cached occupancy queries)
        const uint32_t okey = (count ? 1u : 0u) | (lds ? 2u : 0u) | ((uint32_t)walk << 2) | (P.features ? 16u : 0u) | ((uint32_t)occ << 5);
        if (s->wf_occ.key != okey || s->wf_occ.stack != P.trav_stack_dwords) {
            HIP_TRY(wf_resident_waves(P, false, count, lds, walk, occ, &s->wf_occ.waves));
            s->wf_occ.key = okey; s->wf_occ.stack = P.trav_stack_dwords;
        }
    }
    const unsigned resident_waves = s->wf_occ.waves;
    // pair mode (LaunchParams::wf_pair_in): the last level combines sibling rays with their parent in registers, so
    // level D - 1 needs no resolve launch
    const bool pair_mode = D >= 2 && s->pair_mode;
    P.wf_pair_in = 0; P.wf_pair_out = (pair_mode && D == 2) ? 1 : 0;
    P.wf_nodes_grand = nullptr; P.wf_ncap_grand = 0;
    if (profile) HIP_TRY(hipEventRecord(s->ev_prof[2], stream));
    HIP_TRY(launch_wf_primary(P, count, lds, walk, occ, stream));
    if (profile) HIP_TRY(hipEventRecord(s->ev_prof[3], stream));
    for (int l = 2; l <= D; l++) {
        P.wf_level = l;
        P.wf_rays_in = rays(l); P.wf_count_in = qcount(l); P.wf_cap_in = cap(l);
        P.wf_rays_out = rays(l + 1); P.wf_count_out = qcount(l + 1); P.wf_cap_out = cap(l + 1);
        P.wf_rng_in = rng(l); P.wf_rng_out = rng(l + 1);
        P.wf_nodes_parent = nodes(l - 1); P.wf_ncap_parent = cap(l - 1);
        P.wf_nodes_self = nodes(l); P.wf_ncount_self = ncount(l); P.wf_ncap_self = cap(l);
        P.wf_pair_out = (pair_mode && l == D - 1) ? 1 : 0;
        P.wf_pair_in = (pair_mode && l == D) ? 1 : 0;
        P.wf_nodes_grand = l >= 3 ? nodes(l - 2) : nullptr; P.wf_ncap_grand = l >= 3 ? cap(l - 2) : 0;
        size_t total = (size_t)cap(l) * kShards;
        // LDS scenes: as many waves as can be resident (the kernel numbers its batches through all shards)
        unsigned waves = (unsigned)std::min<size_t>((total + 63) / 64, lds ? resident_waves : kPersistentWavesNarrow);
        waves = std::max<unsigned>(kShards * 4, (waves / (kShards * 4)) * (kShards * 4));   // whole workgroups per shard
        HIP_TRY(launch_wf_secondary(P, count, lds, walk, occ, waves, stream));
    }
    P.wf_pair_in = P.wf_pair_out = 0;
    const int top = pair_mode ? D - 2 : D - 1;            // highest level that still has to be resolved by a launch
    if (top >= 2 && shard_px <= (size_t)s->fused_resolve_shard_px) {
        // small frame (a rank's share of a tiled frame): all resolve levels in ONE launch, a workgroup per shard
        ResolveLevels R;
        memset(&R, 0, sizeof R);
        for (int l = 1; l <= top; l++) { R.nodes[l] = nodes(l); R.ncount[l] = ncount(l); R.cap[l] = cap(l); }
        R.top = top;
        P.wf_level = 1;
        HIP_TRY(launch_wf_resolve_fused(P, R, kShards, stream));
        return P3D_OK;
    }
    for (int l = top; l >= 1; l--) {
        P.wf_level = l;
        P.wf_nodes_self = nodes(l); P.wf_ncount_self = ncount(l); P.wf_ncap_self = cap(l);
        P.wf_nodes_parent = nodes(l - 1); P.wf_ncap_parent = l > 1 ? cap(l - 1) : 0;
        HIP_TRY(launch_wf_resolve(P, kShards * (unsigned)s->resolve_blocks_per_shard, stream));
    }
    return P3D_OK;
}

}  // namespace

namespace {
constexpr int kTileOrderPeriod = 64;
// Before a launch that draws its tiles through an order: buffers for this configuration, the order to use (nullptr until a
// frame has measured the tiles) and whether this frame's measurements should be sorted into a new order afterwards.
int tile_order_begin(p3d_scene* s, p3d_scene::TileOrder& L, const int32_t (&key)[9], uint32_t n_tiles, const uint32_t** order,
                     uint32_t** cost, bool* sort_after) {
    *order = nullptr; *cost = nullptr; *sort_after = false;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s->stream, &cap);
    if (memcmp(key, L.key, sizeof key) != 0) {
        L.valid = false;
        if (cap != hipStreamCaptureStatusNone) { L.key[3] = -1; return P3D_OK; }      // (allocations: not while a capture is open)
        const size_t nb = (size_t)n_tiles * sizeof(uint32_t);
        HIP_TRY(L.cost.ensure(nb)); HIP_TRY(L.sorted.ensure(nb)); HIP_TRY(L.iota.ensure(nb)); HIP_TRY(L.order.ensure(nb));
        size_t tb = 0;
        HIP_TRY(sort_tiles_by_cost(nullptr, nullptr, nullptr, nullptr, n_tiles, nullptr, tb, s->stream));
        HIP_TRY(L.temp.ensure(tb)); L.temp_bytes = tb;
        HIP_TRY(launch_clear_words((uint32_t*)L.cost.p, n_tiles, s->stream));
        memcpy(L.key, key, sizeof key); L.frames = 0;
    }
    *cost = (uint32_t*)L.cost.p;
    *order = L.valid ? (const uint32_t*)L.order.p : nullptr;
    *sort_after = cap == hipStreamCaptureStatusNone && (!L.valid || L.frames >= kTileOrderPeriod);
    return P3D_OK;
}
int tile_order_end(p3d_scene* s, p3d_scene::TileOrder& L, uint32_t n_tiles, bool sort_after) {
    L.frames++;
    if (sort_after) {            // the order the NEXT frames draw their tiles in, from what this one measured
        HIP_TRY(sort_tiles_by_cost((const uint32_t*)L.cost.p, (uint32_t*)L.sorted.p, (uint32_t*)L.iota.p, (uint32_t*)L.order.p, n_tiles,
                                   L.temp.p, L.temp_bytes, s->stream));
        L.valid = true; L.frames = 0;
    }
    return P3D_OK;
}
}  // namespace

int p3d_render(p3d_scene* s, const p3d_camera* cam, const p3d_render_params* prm, const p3d_outputs* out) {
    if (!s || !cam || !prm || !out) return fail(P3D_ERR_ARG, "NULL argument");
    if (cam->res_x <= 0 || cam->res_y <= 0) return fail(P3D_ERR_ARG, "bad resolution");
    if (prm->max_depth < 1 || prm->max_depth > 16) return fail(P3D_ERR_ARG, "max_depth must be in 1..16");
    if (prm->accel < 0 || prm->accel > 2) return fail(P3D_ERR_ARG, "accel must be 0, 1 or 2");
    if (s->unit_rays_only && prm->accel == P3D_ACCEL_NONE)
        return fail(P3D_ERR_STATE, "scene was built with cull_never_hit: accel NONE (un-normalised shadow rays) is not served");
    if (prm->spp < 0 || prm->spp > 8) return fail(P3D_ERR_ARG, "spp must be in 0..8");
    if (prm->spp > 0 && !prm->samples) return fail(P3D_ERR_ARG, "spp > 0 needs the host sample array");
    int world = prm->world > 0 ? prm->world : 1;
    int rank = prm->rank;
    if (rank < 0 || rank >= world) return fail(P3D_ERR_ARG, "rank outside [0, world)");
    int row_block = prm->row_block > 0 ? prm->row_block : 16;
    if (row_block % 16) return fail(P3D_ERR_ARG, "row_block must be a multiple of 16");
    HIP_TRY(hipSetDevice(s->device));

    LaunchParams P;
    memset(&P, 0, sizeof P);
    P.blob = s->blob.p; P.blob_quads = s->blob_quads;
    P.qnodes = s->qnodes.p; memcpy(P.q_scale, s->q_scale, sizeof P.q_scale); memcpy(P.q_base, s->q_base, sizeof P.q_base);
    P.off_nodes = s->off_nodes; P.off_leaves = s->off_leaves; P.off_spheres = s->off_spheres;
    P.off_sphere_meta = s->off_sphere_meta; P.off_tris = s->off_tris; P.off_tri_normals = s->off_tri_normals; P.tri_quads = s->tri_quads; P.off_boxes = s->off_boxes; P.off_mats = s->off_mats;
    P.planes = s->planes.p; P.plane_meta = s->plane_meta.p; P.lights = s->lights.p;
    // small scenes are rendered from an LDS copy shared by the 4 waves of a 256-thread workgroup
    const bool lds_scene = !(prm->flags & P3D_FLAG_NO_LDS_SCENE) && s->lds_capable;
    P.wg_waves = lds_scene ? 4 : 1;
    // small trees are walked by the whole wave together (packet walk), large ones per lane; GRID mode walks the
    // reference's uniform grid per lane
    // (the per-lane walk is the default everywhere: with typed leaves it is the faster one on BASELINE config 2 too,
    //  0.133 vs 0.137 ms; P3D_FLAG_PACKET_WALK asks for the wave-wide walk, which exists for trees up to 64 node pairs)
    const bool packet = prm->accel != P3D_ACCEL_GRID && (prm->flags & P3D_FLAG_PACKET_WALK) &&
                        s->stats.n_nodes <= s->packet_node_limit;
    // scenes read from HBM: the lanes of a wave can share their walks (p3d_traverse.h: closest_hit_shared).  Whether they do
    // is part of the measured choice below (it pays on some scenes and costs on others); P3D_FLAG_PRIVATE_WALK rules it out.
    const bool can_share = !lds_scene && prm->accel != P3D_ACCEL_GRID && !packet && s->share_min_idle > 0 && s->share_min_idle <= 64 &&
                           !(prm->flags & P3D_FLAG_PRIVATE_WALK);
    bool shared_walk = can_share;
    int walk = prm->accel == P3D_ACCEL_GRID ? 2 : (packet ? 1 : (shared_walk ? 3 : 0));
    if (prm->accel == P3D_ACCEL_GRID) {
        if (s->unit_rays_only) return fail(P3D_ERR_STATE, "scene was built with cull_never_hit: GRID mode walks the reference's grid over ALL primitives; use accel BVH");
        if (!s->grid_ready) {
            // built on the first GRID frame of a scene, with synchronous allocations and copies: not something a
            // stream capture can hold.  Render one GRID frame before capturing.
            hipStreamCaptureStatus gcap = hipStreamCaptureStatusNone;
            (void)hipStreamIsCapturing(s->stream, &gcap);
            if (gcap != hipStreamCaptureStatusNone)
                return fail(P3D_ERR_STATE, "the first GRID-mode frame of a scene builds and uploads the grid: render one before capturing the stream");
            if (!build_grid(s->grid_src, s->grid_info)) return fail(P3D_ERR_LIMIT, "the reference's grid formula asks for more than 2^31 cells");
            std::vector<GridPrim>().swap(s->grid_src);          // (28 B per primitive: only the build needed it)
            HIP_TRY(s->grid_cells.upload(s->grid_info.cell_start));
            HIP_TRY(s->grid_items.upload(s->grid_info.items));
            s->stats.device_bytes += s->grid_cells.bytes() + s->grid_items.bytes();
            std::vector<uint32_t>().swap(s->grid_info.cell_start); std::vector<uint32_t>().swap(s->grid_info.items);
            s->grid_ready = true;
        }
        P.grid_cells = s->grid_cells.p; P.grid_items = s->grid_items.p;
        for (int a = 0; a < 3; a++) { P.grid_n[a] = s->grid_info.n[a]; P.grid_min[a] = s->grid_info.mn[a]; P.grid_max[a] = s->grid_info.mx[a]; }
    }
    P.n_planes = s->stats.n_planes; P.n_lights = s->n_lights; P.n_materials = s->n_materials;
    P.trav_stack_entries = std::max<uint32_t>(s->stats.max_depth + 1, 2);
    // 4-byte slots (node references only, p3d_traverse.h) for both scene placements
#ifndef P3D_LDS_STACK_DWORDS
#define P3D_LDS_STACK_DWORDS 64u      // RefStack: 4-byte slots (WideStack: 128)
#endif
    P.trav_stack_dwords = P.trav_stack_entries * (lds_scene ? P3D_LDS_STACK_DWORDS : kHbmStackDwordsPerEntry);
    if (shared_walk) P.trav_stack_dwords += kShareDwords;      // the wave's share region sits behind its stack slots
    P.share_min_idle = (uint32_t)s->share_min_idle;
    memcpy(P.bg, s->bg, sizeof P.bg);
    memcpy(P.eye, cam->eye, sizeof P.eye); memcpy(P.u, cam->u, sizeof P.u);
    memcpy(P.v, cam->v, sizeof P.v); memcpy(P.n, cam->n, sizeof P.n);
    P.w = cam->w; P.h = cam->h; P.plane_dist = cam->plane_dist; P.aperture = cam->aperture;
    P.focal_ratio = cam->focal_ratio; P.res_x = cam->res_x; P.res_y = cam->res_y;
    for (int a = 0; a < 3; a++) {               // same float products Camera::PrimaryRay forms per call
        P.uw[a] = cam->u[a] * cam->w; P.vh[a] = cam->v[a] * cam->h; P.vz[a] = cam->n[a] * -cam->plane_dist;
    }
    if (s->tab_res_x != cam->res_x || s->tab_res_y != cam->res_y) {
        HIP_TRY(s->ray_tab.ensure(((size_t)cam->res_x + cam->res_y) * sizeof(float)));
        HIP_TRY(launch_raygen_table((float*)s->ray_tab.p, (float*)s->ray_tab.p + cam->res_x, cam->res_x, cam->res_y, s->stream));
        s->tab_res_x = cam->res_x; s->tab_res_y = cam->res_y;
    }
    P.ray_fx = (const float*)s->ray_tab.p; P.ray_fy = P.ray_fx + cam->res_x;
    P.max_depth = prm->max_depth; P.accel = prm->accel; P.spp = prm->spp;
    P.row_block = row_block; P.rank = rank; P.world = world;
    P.row_block_shift = -1;
    for (int sh = 4; sh < 20; sh++) if ((1 << sh) == row_block) P.row_block_shift = sh;
    P.local_rows = p3d_local_rows(cam->res_y, row_block, world);
    const int tile_rows = 4 * P.wg_waves;
    P.tiles_x = (cam->res_x + 15) / 16; P.tiles_y = P.local_rows / tile_rows;
    P.n_tiles = P.tiles_x * P.tiles_y;
    P.xcd_chunk = s->xcd_chunk > 0 ? s->xcd_chunk : 1;
    {   // grid = whole chunks for all 8 XCDs (surplus blocks exit immediately)
        int chunks = (P.n_tiles + P.xcd_chunk - 1) / P.xcd_chunk;
        P.grid_blocks = ((chunks + 7) / 8) * 8 * P.xcd_chunk;
    }
    P.counters = s->d_counters;
    P.dbg_skip = s->dbg_skip;
    P.dbg_stamps = s->dbg_stamps; P.dbg_stamp_level = s->dbg_stamp_level;
    P.wf_min_width = lds_scene ? 64 : 8;

    // distribution-ray-tracing switches (RT/main.cpp:40-45)
    if (prm->features & ~(P3D_FEATURE_SOFT_SHADOW | P3D_FEATURE_FUZZY_REFLECTION | P3D_FEATURE_SKYBOX)) return fail(P3D_ERR_ARG, "unknown feature bit");
    if (prm->features & P3D_FEATURE_SKYBOX) {
        if (!s->sky.p || s->sky_w[0] == 0) return fail(P3D_ERR_STATE, "P3D_FEATURE_SKYBOX needs p3d_scene_set_skybox() first");
        P.features |= kFeatSky;
        P.sky = s->sky.p;
        for (int i = 0; i < 6; i++) { P.sky_off[i] = s->sky_off[i]; P.sky_w[i] = s->sky_w[i]; P.sky_h[i] = s->sky_h[i]; P.sky_bpp[i] = s->sky_bpp[i]; }
    }
    if ((prm->features & P3D_FEATURE_SOFT_SHADOW) && prm->spp == 0 && s->n_lights) {
        // the 4x4 grid of RT/main.cpp:601-618 as 16 sub-lights per light, in the reference's loop order
        // and with its float arithmetic (cur_x / cur_y advance by repeated addition)
        if (!s->soft_lights.p) {
            std::vector<LightRec> sub;
            for (const LightRec& li : s->host_lights) {
                const float shadow = 0.5f, distance = shadow / 4;
                float cur_x = li.pos[0] - distance * shadow * 4;
                float cur_y = li.pos[1] - distance * shadow * 4;
                for (int a = 0; a < 4; a++) {
                    for (int b = 0; b < 4; b++) {
                        LightRec q = li;
                        q.pos[0] = cur_x; q.pos[1] = cur_y;
                        for (int c = 0; c < 3; c++) q.col[c] = li.col[c] / (4 * 4);
                        sub.push_back(q);
                        cur_x += distance;
                    }
                    cur_y += distance;
                    cur_x = li.pos[0] - distance * shadow * 4;
                }
            }
            HIP_TRY(s->soft_lights.upload(sub));
        }
        P.lights = s->soft_lights.p; P.n_lights = s->n_lights * 16;
    }
    if ((prm->features & P3D_FEATURE_SOFT_SHADOW) && prm->spp > 0) P.features |= kFeatSoftJitter;
    if (prm->features & P3D_FEATURE_FUZZY_REFLECTION) P.features |= kFeatFuzzy;
    P.seed = prm->seed;
    const bool stochastic = P.features != 0;
    const bool count = (prm->flags & P3D_FLAG_COUNTERS) != 0;

    // ---- schedule.  Three ways to run the same per-node code, bit-identical frames:
    //   TILE       one launch; a workgroup keeps a 16x16 tile's whole tree to itself (default)
    //   WAVEFRONT  one launch per tree level over the whole frame + resolve launches
    //   TREE       one launch; each lane walks its pixel's whole tree
    // None wins everywhere, so the default is MEASURED per configuration: see SchedulePick below.
    enum { SCHED_WAVEFRONT = 0, SCHED_TREE = 1, SCHED_TILE = 2 };
    const uint32_t forced = prm->flags & (P3D_FLAG_TREE_KERNEL | P3D_FLAG_WAVEFRONT | P3D_FLAG_TILE_KERNEL);
    if (forced & (forced - 1)) return fail(P3D_ERR_ARG, "at most one of P3D_FLAG_TREE_KERNEL / _WAVEFRONT / _TILE_KERNEL");
    if (stochastic && (prm->flags & P3D_FLAG_TREE_KERNEL))
        return fail(P3D_ERR_ARG, "features with random draws (and the skybox) need the tile or the wavefront schedule");

    {   // the workspace budget, capped by what the device has free (+ what this handle already holds)
        const int32_t bkey[4] = {cam->res_x, cam->res_y, prm->max_depth, prm->spp};
        if (memcmp(bkey, s->budget_key, sizeof bkey) != 0) {
            size_t free_b = 0, total_b = 0;
            size_t held = s->tile_ws.cap + s->wf_planes.cap;
            for (auto& w : s->ws) held += w.held();
            s->budget_avail = s->workspace_budget;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
                s->budget_avail = std::min<size_t>(s->workspace_budget, (size_t)((double)(free_b + held) * 0.85));
            memcpy(s->budget_key, bkey, sizeof bkey);
        }
    }
    const size_t budget = s->budget_avail;
    // what the tile schedule needs: its LDS, and one private workspace slot per resident workgroup
    LaunchParams PT = P;                                   // tile geometry: 16x16 tiles, 4 waves per workgroup
    PT.wg_waves = 4;
    PT.tiles_y = P.local_rows / 16; PT.n_tiles = PT.tiles_x * PT.tiles_y;
    const int D = prm->max_depth;
    const size_t slot_rays = (size_t)tile_ray_entries(D) * sizeof(RayRec), slot_nodes = (size_t)tile_node_entries(D) * sizeof(NodeRec);
    const size_t slot_rng = stochastic ? (size_t)tile_ray_entries(D) * sizeof(uint32_t) : 0;
    const size_t slot_bytes = ((slot_rays + slot_nodes + slot_rng + 255) / 256) * 256 + 256;
    int tile_blocks = 0;
    // register budget of the tile kernel: scenes read from HBM run it at 5 waves per SIMD (with the work-sharing walk the
    // default allocation is 129 VGPRs: one over the step to 3 waves per SIMD, i.e. 3 workgroups per CU instead of 4)
    // LDS scenes too: 106 VGPRs by default (4 waves per SIMD), 96 with 3 spilled at 5: config 4 4.96 -> 4.75 ms; at 6 (80
    // VGPRs, 17 spilled) 5.03 (profiles/r03_exp10_register_budgets.txt)
    const int tile_occ = s->occupancy ? s->occupancy : 5;
    bool tile_ok = false;
    auto tile_query = [&]() -> int {                      // what the tile schedule can have with the walk in force
        tile_blocks = 0;
        tile_ok = tile_kernel_lds_bytes(PT, lds_scene) <= kMaxLdsBytes && D <= 16;
        if (!tile_ok) return P3D_OK;
        const uint32_t okey = (count ? 1u : 0u) | (lds_scene ? 2u : 0u) | ((uint32_t)walk << 2) | (stochastic ? 16u : 0u) | ((uint32_t)tile_occ << 5);
        const size_t olds = tile_kernel_lds_bytes(PT, lds_scene);
        auto* slot = &s->tile_occ[walk == 3 ? 1 : 0];
        if (slot->key != okey || slot->lds != olds) {
            HIP_TRY(tile_kernel_resident_blocks(PT, count, lds_scene, walk, tile_occ, &slot->blocks));
            slot->key = okey; slot->lds = olds;
        }
        tile_blocks = std::min(slot->blocks, PT.n_tiles);
        tile_blocks = (int)std::min<size_t>((size_t)tile_blocks, budget / slot_bytes);
        // fewer resident workgroups than a quarter of the CUs: the per-tile worst case of this depth does not fit
        tile_ok = tile_blocks >= std::min(64, PT.n_tiles);
        return P3D_OK;
    };
    auto apply_walk = [&](bool share) -> int {            // private <-> shared walks for this frame (scenes read from HBM)
        if (share == shared_walk) return P3D_OK;
        shared_walk = share;
        walk = shared_walk ? 3 : 0;
        P.trav_stack_dwords = P.trav_stack_entries * kHbmStackDwordsPerEntry + (shared_walk ? kShareDwords : 0u);
        PT.trav_stack_dwords = P.trav_stack_dwords;
        return tile_query();
    };
    { int rc = tile_query(); if (rc) return rc; }
    // wavefront bands: worst-case queues for a band of tile rows must fit the workspace budget
    const size_t tile_row_px = (size_t)P.tiles_x * 64 * P.wg_waves;
    // sample passes of one frame run on up to kLanes streams, each with its share of the budget
    // ... and a one-sample frame may be cut into frame_streams bands of tile rows that run concurrently the same way
    const int frame_streams = (prm->spp == 0 && lds_scene) ? std::max(1, std::min(kLanes, s->frame_streams)) : 1;
    const int lanes = prm->spp > 0 ? std::min(kLanes, prm->spp * prm->spp) : frame_streams;
    size_t wf_bpp = wavefront_bytes_per_pixel(prm->max_depth);
    if (stochastic) for (int l = 2; l <= prm->max_depth; l++) wf_bpp += ((size_t)1 << (l - 1)) * sizeof(uint32_t);
    size_t band_tile_rows = wf_bpp ? budget / lanes / (wf_bpp * tile_row_px) : (size_t)P.tiles_y;
    if (wf_bpp == 0) band_tile_rows = (size_t)P.tiles_y;
    band_tile_rows = std::min<size_t>(band_tile_rows, (size_t)P.tiles_y);
    if (frame_streams > 1) band_tile_rows = std::min<size_t>(band_tile_rows, ((size_t)P.tiles_y + frame_streams - 1) / frame_streams);
    const bool wavefront_ok = band_tile_rows > 0;

    s->tune_candidates = 0;
    int sched = SCHED_TILE;
    int measuring = -1;                 // schedule this frame is timed as, for the pick below
    int measured_cand = -1;             // ... and the pick's candidate (schedule x shared / private walks) it counts for
    if (prm->flags & P3D_FLAG_TREE_KERNEL) sched = SCHED_TREE;
    else if (prm->flags & P3D_FLAG_WAVEFRONT) sched = SCHED_WAVEFRONT;
    else if (prm->flags & P3D_FLAG_TILE_KERNEL) sched = SCHED_TILE;
    else if (lds_scene) {
        // scenes served from LDS: by rule (measured once, on BASELINE configs 2 and 4: a one-sample 1080p frame
        // 0.136 ms wavefront / 0.22 tile / 0.22 tree; 4096^2 x 4 samples 5.4 / 5.1 / 11.1 -- with samples the tile
        // schedule needs no per-sample planes and no summing launch).  A timing-based pick is not used here: these
        // frames are short enough to be run several at a time on separate handles, which falsifies the timings.
        sched = prm->spp > 0 ? SCHED_TILE : SCHED_WAVEFRONT;
    } else {
        // No schedule wins everywhere (one 1080p frame of a 12-primitive scene: wavefront 0.14 ms, tile 0.22;
        // 4096^2 x 4 samples of it: tile 5.1, wavefront 5.4; the dragon: tree = tile 2.3, wavefront 3.6; 1e6
        // random primitives: wavefront 3.7, tile 6.1, tree 9.5), so the library MEASURES: the first frames of a
        // configuration run every available schedule twice -- the first time untimed: code-object load,
        // workspace allocation -- and the fastest one stays.  All produce identical bits.
        // Scenes whose lanes can share their walks measure every schedule both ways (candidates 0-2 shared, 3-5 private).
        constexpr int NS = 3;
        const int NC = can_share ? 2 * NS : NS;
        const bool avail_s[NS] = {wavefront_ok, !stochastic, tile_ok};
        auto cand_share = [&](int c) { return can_share && c < NS; };
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s->stream, &cap);
        p3d_scene::SchedulePick& pk = s->pick;
        const int32_t key[8] = {cam->res_x, cam->res_y, prm->max_depth, prm->accel, prm->spp, rank, world,
                                (int32_t)((prm->flags & (P3D_FLAG_NO_LDS_SCENE | P3D_FLAG_PACKET_WALK | P3D_FLAG_COUNTERS | P3D_FLAG_PRIVATE_WALK)) | (prm->features << 9))};
        int cand = -1;
        s->tune_candidates = NC;
        s->tune_avail = 0;
        for (int k = 0; k < NC; k++) if (avail_s[k % NS]) s->tune_avail |= 1u << k;
        if (s->tune_commit >= 0) {              // p3d_tune_schedule() measured this configuration under the caller's load
            memcpy(pk.key, key, sizeof key);
            pk.best = s->tune_commit < NC ? s->tune_commit : SCHED_TILE;
            pk.step = 2 * NC + 1; pk.pending = -1;
            s->tune_commit = -1;
        }
        if (s->tune_force >= 0) {
            if (s->tune_force < NC) cand = s->tune_force;
        } else if (cap != hipStreamCaptureStatusNone) {
            // events cannot be read while the stream is being captured: use what is known, measure nothing
            if (memcmp(key, pk.key, sizeof key) == 0 && pk.step >= 2 * NC) cand = pk.best;
        } else {
            if (memcmp(key, pk.key, sizeof key) != 0) {
                memcpy(pk.key, key, sizeof key);
                for (float& m : pk.ms) m = -1.0f;
                pk.pending = -1; pk.step = 0; pk.best = SCHED_TILE;
            }
            if (pk.pending >= 0) {              // collect the measurement of the previous frame
                HIP_TRY(hipEventSynchronize(s->ev_pick[1]));
                float ms = 0.0f;
                HIP_TRY(hipEventElapsedTime(&ms, s->ev_pick[0], s->ev_pick[1]));
                if (pk.ms[pk.pending] < 0.0f || ms < pk.ms[pk.pending]) pk.ms[pk.pending] = ms;
                pk.pending = -1;
            }
            while (pk.step < 2 * NC && !avail_s[(pk.step / 2) % NS]) pk.step = (pk.step / 2 + 1) * 2;     // skip what cannot run
            if (pk.step < 2 * NC) {
                cand = pk.step / 2;
                if (pk.step & 1) measuring = cand;
                pk.step++;
            } else {
                if (pk.step == 2 * NC) {
                    pk.best = -1;
                    for (int k = 0; k < NC; k++)
                        if (avail_s[k % NS] && pk.ms[k] >= 0.0f && (pk.best < 0 || pk.ms[k] < pk.ms[pk.best])) pk.best = k;
                    if (pk.best < 0) pk.best = tile_ok ? SCHED_TILE : (wavefront_ok ? SCHED_WAVEFRONT : SCHED_TREE);
                    pk.step++;
                    if (s->verbose)
                        fprintf(stderr, "p3d: measured choice (ms; wavefront / tree / tile, shared walks then private): %.3f %.3f %.3f | %.3f %.3f %.3f -> %d\n",
                                pk.ms[0], pk.ms[1], pk.ms[2], pk.ms[3], pk.ms[4], pk.ms[5], pk.best);
                }
                cand = pk.best;
            }
        }
        if (cand >= 0) {
            sched = cand % NS;
            int rc = apply_walk(cand_share(cand));
            if (rc) return rc;
        }
        measured_cand = measuring;
        if (measuring >= 0) measuring = measuring % NS;
    }
    // a schedule whose workspace does not fit falls back: tile -> wavefront (bands) -> tree
    if (sched == SCHED_TILE && !tile_ok) sched = wavefront_ok ? SCHED_WAVEFRONT : SCHED_TREE;
    if (sched == SCHED_WAVEFRONT && !wavefront_ok) sched = SCHED_TREE;
    if (sched == SCHED_TREE && stochastic) return fail(P3D_ERR_LIMIT, "workspace budget too small for the schedules the features need");
    const bool use_tree = sched == SCHED_TREE, use_tile = sched == SCHED_TILE;
    s->last_schedule = sched;
    size_t lds = use_tree ? tree_kernel_lds_bytes(P, lds_scene) : use_tile ? tile_kernel_lds_bytes(PT, lds_scene) : wavefront_lds_bytes(P, lds_scene);
    if (lds > kMaxLdsBytes) return fail(P3D_ERR_LIMIT, "BVH depth / max_depth need more LDS than a CU has");
    if (lds > 64 * 1024 && !use_tile && !s->lds_prepared) {
        HIP_TRY(prepare_kernels(kMaxLdsBytes));
        s->lds_prepared = kMaxLdsBytes;
    }

    const size_t npx = (size_t)P.local_rows * cam->res_x;
    if (prm->spp > 0) {
        size_t bytes = (size_t)cam->res_y * cam->res_x * prm->spp * prm->spp * 4 * sizeof(float);
        if (prm->flags & P3D_FLAG_DEVICE_SAMPLES) {
            P.samples = prm->samples;                      // already on this device (uploaded once by the caller)
        } else {
            HIP_TRY(s->samples.ensure(bytes));
            HIP_TRY(hipMemcpyAsync(s->samples.p, prm->samples, bytes, hipMemcpyHostToDevice, s->stream));
            P.samples = (const float*)s->samples.p;
        }
    }
    if (out->memory == 1) {
        P.rgb8 = out->rgb8; P.rgb32f = out->rgb32f; P.hit_id = out->hit_id;
    } else {
        if (out->rgb8) { HIP_TRY(s->fb_rgb8.ensure(npx * 3)); P.rgb8 = (uint8_t*)s->fb_rgb8.p; }
        if (out->rgb32f) { HIP_TRY(s->fb_rgb32f.ensure(npx * 12)); P.rgb32f = (float*)s->fb_rgb32f.p; }
        if (out->hit_id) { HIP_TRY(s->fb_hit.ensure(npx * 4)); P.hit_id = (int32_t*)s->fb_hit.p; }
    }
    if (count) {
        HIP_TRY(launch_clear_words((uint32_t*)s->d_counters, (uint32_t)(sizeof(DeviceCounters) / 4), s->stream));
        s->counters_valid = true;
    }
    P.wf_nsamples = prm->spp > 0 ? prm->spp * prm->spp : 1;
    const bool profile = (prm->flags & P3D_FLAG_PROFILE) != 0;
    if (profile) HIP_TRY(hipEventRecord(s->ev_prof[0], s->stream));
    if (use_tile) {
        // (resident workgroups) x (one tile's worst-case queues); the tile counter and the exit ticket are
        // zeroed once here and re-armed by the kernel itself at the end of every launch
        HIP_TRY(s->tile_ws.ensure((size_t)tile_blocks * slot_bytes));
        if (!s->tile_ctrl.p) {
            HIP_TRY(s->tile_ctrl.ensure(256));
            HIP_TRY(launch_clear_words((uint32_t*)s->tile_ctrl.p, 64, s->stream));
        }
        PT.samples = P.samples; PT.rgb8 = P.rgb8; PT.rgb32f = P.rgb32f; PT.hit_id = P.hit_id;
        PT.wf_nsamples = P.wf_nsamples;
        PT.tw_base = (uint8_t*)s->tile_ws.p; PT.tw_slot_bytes = slot_bytes; PT.tw_ctrl = (uint32_t*)s->tile_ctrl.p;
        PT.tw_rays_off = 0; PT.tw_nodes_off = (uint32_t)slot_rays; PT.tw_rng_off = (uint32_t)(slot_rays + slot_nodes);
        if (measuring >= 0) HIP_TRY(hipEventRecord(s->ev_pick[0], s->stream));
        if (profile) HIP_TRY(hipEventRecord(s->ev_prof[2], s->stream));
        // heaviest tile first.  Scenes read from HBM: their tiles differ by orders of magnitude.  Scenes served from LDS run
        // this schedule for sample loops (config 4), where a glass tile is a serial chain 15 x as long as a sky tile and the
        // natural order left the last fifth of the frame to a few workgroups: 4.77 -> 3.81 ms (profiles/r03_exp31_32_config4_tail.txt)
        bool lpt_sort = false;
        const int32_t lkey[9] = {cam->res_x, cam->res_y, prm->max_depth, prm->accel, prm->spp, rank, world, (int32_t)prm->features, SCHED_TILE};
        if (s->tile_lpt_enabled && PT.n_tiles >= 64) {
            int rc = tile_order_begin(s, s->tile_lpt, lkey, (uint32_t)PT.n_tiles, &PT.tile_order, &PT.tile_cost, &lpt_sort);
            if (rc) return rc;
        }
        if (s->verbose)
            fprintf(stderr, "p3d: tile schedule: %d workgroups (occupancy query: %d on the device), %zu B LDS each, %zu B workspace slot, %d tiles\n",
                    tile_blocks, s->tile_occ[walk == 3 ? 1 : 0].blocks, tile_kernel_lds_bytes(PT, lds_scene), slot_bytes, PT.n_tiles);
        HIP_TRY(launch_wf_tile(PT, count, lds_scene, walk, tile_occ, (unsigned)tile_blocks, s->stream));
        if (profile) HIP_TRY(hipEventRecord(s->ev_prof[3], s->stream));
        if (PT.tile_cost) { int rc = tile_order_end(s, s->tile_lpt, (uint32_t)PT.n_tiles, lpt_sort); if (rc) return rc; }
    } else if (use_tree) {
        if (measuring >= 0) HIP_TRY(hipEventRecord(s->ev_pick[0], s->stream));
        P.wf_tile_row0 = 0; P.wf_tile_rows = P.tiles_y;
        bool lpt_sort = false;
        const int32_t lkey[9] = {cam->res_x, cam->res_y, prm->max_depth, prm->accel, prm->spp, rank, world, (int32_t)prm->features, SCHED_TREE};
        if (!lds_scene && s->tile_lpt_enabled && P.n_tiles >= 64 && P.xcd_chunk == 1) {
            int rc = tile_order_begin(s, s->wave_lpt, lkey, (uint32_t)P.n_tiles, &P.tile_order, &P.tile_cost, &lpt_sort);
            if (rc) return rc;
        }
        if (profile) HIP_TRY(hipEventRecord(s->ev_prof[2], s->stream));
        // scenes read from HBM: a register budget of 6 waves per SIMD (dragon with the flat walk loop: 1.285 ms at 5, 1.239 at 6)
        HIP_TRY(launch_tree(P, count, lds_scene, s->occupancy ? s->occupancy : (lds_scene ? 0 : 6), shared_walk, s->stream));
        if (profile) HIP_TRY(hipEventRecord(s->ev_prof[3], s->stream));
        if (P.tile_cost) { int rc = tile_order_end(s, s->wave_lpt, (uint32_t)P.n_tiles, lpt_sort); if (rc) return rc; }
    } else {
        // a shard owns every kShards-th tile of the band
        const size_t band_tiles = band_tile_rows * (size_t)P.tiles_x;
        const size_t shard_px = ((band_tiles + kShards - 1) / kShards) * 64 * P.wg_waves;
        for (int ln = 0; ln < lanes; ln++) {
            p3d_scene::Workspace& w = s->ws[ln];
            for (int l = 2; l <= D; l++) HIP_TRY(w.rays[l].ensure((shard_px << (l - 1)) * kShards * sizeof(RayRec)));
            for (int l = 1; l <= D - 1; l++) HIP_TRY(w.nodes[l].ensure((shard_px << (l - 1)) * kShards * sizeof(NodeRec)));
            if (!w.counts.p) {   // counters + the alternating level-1 sets + the two parity words (64 words apart)
                const size_t words = kCountBufferWords;
                HIP_TRY(w.counts.ensure(words * sizeof(uint32_t)));
                HIP_TRY(launch_clear_words((uint32_t*)w.counts.p, (uint32_t)words, s->stream));   // once; ordered before the lanes' fork
            }
            if (stochastic)
                for (int l = 2; l <= D; l++) HIP_TRY(w.rng[l].ensure((shard_px << (l - 1)) * kShards * sizeof(uint32_t)));
        }
        if (prm->spp > 0) {
            HIP_TRY(s->wf_planes.ensure((size_t)P.wf_nsamples * npx * 12));
            P.wf_planes = (float*)s->wf_planes.p; P.wf_plane_stride = (uint64_t)npx * 3;
        }
        bool lpt_sort = false;
        const int32_t lkey[9] = {cam->res_x, cam->res_y, prm->max_depth, prm->accel, prm->spp, rank, world, (int32_t)prm->features, SCHED_WAVEFRONT};
        if (!lds_scene && s->tile_lpt_enabled && P.n_tiles >= 64 && P.xcd_chunk == 1 && lanes == 1 && band_tile_rows >= (size_t)P.tiles_y) {
            // (whole-frame passes only: a band numbers its tiles from its own first row)
            int rc = tile_order_begin(s, s->wave_lpt, lkey, (uint32_t)P.n_tiles, &P.tile_order, &P.tile_cost, &lpt_sort);
            if (rc) return rc;
        }
        if (measuring >= 0) HIP_TRY(hipEventRecord(s->ev_pick[0], s->stream));   // after the (host-side) allocations
        if (lanes > 1) {                                   // the other lanes start after what the stream holds
            HIP_TRY(hipEventRecord(s->ev_fork, s->stream));
            for (int ln = 1; ln < lanes; ln++) HIP_TRY(hipStreamWaitEvent(s->lane_stream[ln], s->ev_fork, 0));
        }
        int task = 0;
        for (int smp = 0; smp < P.wf_nsamples; smp++) {
            P.wf_sample = smp;
            for (size_t r0 = 0; r0 < (size_t)P.tiles_y; r0 += band_tile_rows, task++) {
                // sample passes go round the lanes; so do the bands of a one-sample frame cut for concurrency
                const int ln = (prm->spp > 0 ? smp : task) % lanes;
                const hipStream_t lane_stream = ln == 0 ? s->stream : s->lane_stream[ln];
                LaunchParams B = P;
                B.wf_tile_row0 = (int32_t)r0;
                B.wf_tile_rows = (int32_t)std::min<size_t>(band_tile_rows, (size_t)P.tiles_y - r0);
                B.n_tiles = B.tiles_x * B.wf_tile_rows;
                int chunks = (B.n_tiles + B.xcd_chunk - 1) / B.xcd_chunk;
                B.grid_blocks = ((chunks + 7) / 8) * 8 * B.xcd_chunk;
                int rc = run_wavefront_pass(s, s->ws[ln], lane_stream, B, count, lds_scene, walk, shard_px,
                                            profile && smp == 0 && r0 == 0);
                if (rc) return rc;
            }
        }
        for (int ln = 1; ln < lanes; ln++) {               // ... and the stream continues after all of them
            HIP_TRY(hipEventRecord(s->ev_join[ln], s->lane_stream[ln]));
            HIP_TRY(hipStreamWaitEvent(s->stream, s->ev_join[ln], 0));
        }
        // a pixel's clamped sample colours are summed in sample order, then divided by 16
        if (prm->spp > 0) HIP_TRY(launch_sum_samples(P, 0, npx, s->stream));
        if (P.tile_cost) { int rc = tile_order_end(s, s->wave_lpt, (uint32_t)P.n_tiles, lpt_sort); if (rc) return rc; }
    }
    if (profile) { HIP_TRY(hipEventRecord(s->ev_prof[1], s->stream)); s->profile_valid = true; }
    if (measuring >= 0) {
        HIP_TRY(hipEventRecord(s->ev_pick[1], s->stream));
        // a frame pushed onto another schedule by the workspace budget says the measured one is not available
        if (sched == measuring) s->pick.pending = measured_cand;
        else s->pick.ms[measured_cand] = 3.0e38f;
    }
    if (out->memory != 1) {
        // host planes hold res_y rows for a whole frame, p3d_local_rows() rows for a shard
        const size_t cpx = (world == 1 ? (size_t)cam->res_y : (size_t)P.local_rows) * cam->res_x;
        if (out->rgb8) HIP_TRY(hipMemcpyAsync(out->rgb8, P.rgb8, cpx * 3, hipMemcpyDeviceToHost, s->stream));
        if (out->rgb32f) HIP_TRY(hipMemcpyAsync(out->rgb32f, P.rgb32f, cpx * 12, hipMemcpyDeviceToHost, s->stream));
        if (out->hit_id) HIP_TRY(hipMemcpyAsync(out->hit_id, P.hit_id, cpx * 4, hipMemcpyDeviceToHost, s->stream));
        HIP_TRY(hipStreamSynchronize(s->stream));
    }
    return P3D_OK;
}

int p3d_sync(p3d_scene* s) {
    if (!s) return fail(P3D_ERR_ARG, "scene is NULL");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return P3D_OK;
}

int p3d_get_counters(p3d_scene* s, p3d_counters* out) {
    if (!s || !out) return fail(P3D_ERR_ARG, "scene/out is NULL");
    if (!s->counters_valid) return fail(P3D_ERR_STATE, "no render with P3D_FLAG_COUNTERS yet");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    DeviceCounters c;
    HIP_TRY(hipMemcpy(&c, s->d_counters, sizeof c, hipMemcpyDeviceToHost));
    out->closest_queries = c.closest_queries; out->shadow_queries = c.shadow_queries;
    out->box_tests = c.box_tests; out->sphere_tests = c.sphere_tests; out->tri_tests = c.tri_tests;
    out->aabox_tests = c.aabox_tests; out->plane_tests = c.plane_tests; out->pixels = c.pixels;
    return P3D_OK;
}

int p3d_debug_set_stamps(p3d_scene* s, void* device_buffer) {
    if (!s) return fail(P3D_ERR_ARG, "scene is NULL");
    if (device_buffer && !kernels_have_stamps())
        return fail(P3D_ERR_STATE, "this build has no stamp hooks: use the diagnostic build (make -C csrc stamps, libp3d_hip_stamps.so)");
    s->dbg_stamps = (unsigned long long*)device_buffer;
    return P3D_OK;
}

int p3d_debug_set_stamp_level(p3d_scene* s, int32_t level) {
    if (!s) return fail(P3D_ERR_ARG, "scene is NULL");
    if (level < 1 || level > kMaxDepth) return fail(P3D_ERR_ARG, "level must be in 1..16");
    s->dbg_stamp_level = level;
    return P3D_OK;
}

int p3d_get_profile(p3d_scene* s, float* frame_ms, float* kernel_ms) {
    if (!s || !frame_ms || !kernel_ms) return fail(P3D_ERR_ARG, "NULL argument");
    if (!s->profile_valid) return fail(P3D_ERR_STATE, "no render with P3D_FLAG_PROFILE yet");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventSynchronize(s->ev_prof[1]));
    HIP_TRY(hipEventElapsedTime(frame_ms, s->ev_prof[0], s->ev_prof[1]));
    HIP_TRY(hipEventElapsedTime(kernel_ms, s->ev_prof[2], s->ev_prof[3]));
    return P3D_OK;
}

int p3d_last_schedule(p3d_scene* s, int32_t* schedule) {
    if (!s || !schedule) return fail(P3D_ERR_ARG, "NULL argument");
    if (s->last_schedule < 0) return fail(P3D_ERR_STATE, "no render yet");
    *schedule = s->last_schedule;
    return P3D_OK;
}

int p3d_timer_begin(p3d_scene* s) {
    if (!s) return fail(P3D_ERR_ARG, "scene is NULL");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventRecord(s->ev0, s->stream));
    s->timer_open = true;
    return P3D_OK;
}

int p3d_timer_end(p3d_scene* s, float* ms) {
    if (!s || !ms) return fail(P3D_ERR_ARG, "scene/ms is NULL");
    if (!s->timer_open) return fail(P3D_ERR_STATE, "p3d_timer_begin was not called");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipEventRecord(s->ev1, s->stream));
    HIP_TRY(hipEventSynchronize(s->ev1));
    HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
    s->timer_open = false;
    return P3D_OK;
}

int p3d_deinterleave_frames(p3d_scene* s, const void* gathered, void* frames, int32_t res_x, int32_t res_y,
                            int32_t row_block, int32_t world, int32_t bpp, uint64_t rank_stride_bytes,
                            int32_t n_frames, uint64_t tile_stride_bytes, uint64_t frame_stride_bytes) {
    if (!s || !gathered || !frames) return fail(P3D_ERR_ARG, "NULL argument");
    if (res_x <= 0 || res_y <= 0 || world <= 0 || n_frames <= 0) return fail(P3D_ERR_ARG, "bad sizes");
    if (row_block <= 0) row_block = 16;
    if (bpp != 3 && bpp != 4 && bpp != 12) return fail(P3D_ERR_ARG, "bytes_per_pixel must be 3, 4 or 12");
    HIP_TRY(hipSetDevice(s->device));
    const size_t tile = (size_t)p3d_local_rows(res_y, row_block, world) * res_x * bpp;
    const size_t in_stride = tile_stride_bytes ? (size_t)tile_stride_bytes : tile;
    const size_t stride = rank_stride_bytes ? (size_t)rank_stride_bytes : in_stride * n_frames;
    const size_t out_stride = frame_stride_bytes ? (size_t)frame_stride_bytes : (size_t)res_y * res_x * bpp;
    HIP_TRY(launch_deinterleave(gathered, frames, res_x, res_y, row_block, world, stride, bpp, n_frames, in_stride,
                                out_stride, s->stream));
    return P3D_OK;
}

int p3d_deinterleave(p3d_scene* s, const void* gathered, void* frame, int32_t res_x, int32_t res_y,
                     int32_t row_block, int32_t world, int32_t bpp, uint64_t rank_stride_bytes) {
    return p3d_deinterleave_frames(s, gathered, frame, res_x, res_y, row_block, world, bpp, rank_stride_bytes, 1, 0, 0);
}

int p3d_device_alloc(p3d_scene* s, uint64_t bytes, void** out) {
    if (!s || !out) return fail(P3D_ERR_ARG, "scene/out is NULL");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMalloc(out, bytes ? (size_t)bytes : 1));
    return P3D_OK;
}
int p3d_device_free(p3d_scene* s, void* ptr) {
    if (!s) return fail(P3D_ERR_ARG, "scene is NULL");
    HIP_TRY(hipSetDevice(s->device));
    if (ptr) HIP_TRY(hipFree(ptr));
    return P3D_OK;
}
int p3d_upload(p3d_scene* s, void* device_dst, const void* host_src, uint64_t bytes) {
    if (!s || !device_dst || !host_src) return fail(P3D_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(device_dst, host_src, (size_t)bytes, hipMemcpyHostToDevice, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return P3D_OK;
}
int p3d_download(p3d_scene* s, void* host_dst, const void* device_src, uint64_t bytes) {
    if (!s || !host_dst || !device_src) return fail(P3D_ERR_ARG, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpyAsync(host_dst, device_src, (size_t)bytes, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return P3D_OK;
}

int p3d_debug_intersect(int device, uint32_t n, const uint32_t* type, const float* prim12, const float* origin,
                        const float* dir, int32_t* hit, float* t, float* normal) {
    if (!type || !prim12 || !origin || !dir || !hit || !t || !normal) return fail(P3D_ERR_ARG, "NULL argument");
    if (n == 0) return P3D_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(P3D_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    uint32_t* d_type = nullptr; float *d_prim = nullptr, *d_o = nullptr, *d_d = nullptr, *d_t = nullptr, *d_n = nullptr;
    int32_t* d_hit = nullptr;
    int rc = P3D_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_type); (void)hipFree(d_prim); (void)hipFree(d_o); (void)hipFree(d_d);
        (void)hipFree(d_t); (void)hipFree(d_n); (void)hipFree(d_hit);
    };
#define DBG_TRY(expr)                                                                                  \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) { rc = fail(P3D_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return rc; } } while (0)
    DBG_TRY(hipMalloc((void**)&d_type, n * 4)); DBG_TRY(hipMalloc((void**)&d_prim, (size_t)n * 48));
    DBG_TRY(hipMalloc((void**)&d_o, (size_t)n * 12)); DBG_TRY(hipMalloc((void**)&d_d, (size_t)n * 12));
    DBG_TRY(hipMalloc((void**)&d_t, n * 4)); DBG_TRY(hipMalloc((void**)&d_n, (size_t)n * 12));
    DBG_TRY(hipMalloc((void**)&d_hit, n * 4));
    DBG_TRY(hipMemcpy(d_type, type, n * 4, hipMemcpyHostToDevice));
    DBG_TRY(hipMemcpy(d_prim, prim12, (size_t)n * 48, hipMemcpyHostToDevice));
    DBG_TRY(hipMemcpy(d_o, origin, (size_t)n * 12, hipMemcpyHostToDevice));
    DBG_TRY(hipMemcpy(d_d, dir, (size_t)n * 12, hipMemcpyHostToDevice));
    DBG_TRY(launch_debug_intersect(n, d_type, d_prim, d_o, d_d, d_hit, d_t, d_n, nullptr));
    DBG_TRY(hipDeviceSynchronize());
    DBG_TRY(hipMemcpy(hit, d_hit, n * 4, hipMemcpyDeviceToHost));
    DBG_TRY(hipMemcpy(t, d_t, n * 4, hipMemcpyDeviceToHost));
    DBG_TRY(hipMemcpy(normal, d_n, (size_t)n * 12, hipMemcpyDeviceToHost));
#undef DBG_TRY
    cleanup();
    return rc;
}

int p3d_tune_schedule(p3d_scene** scenes, int32_t n, const p3d_camera* cam, const p3d_render_params* prm, const p3d_outputs* outs,
                      int32_t frames, float* ms_per_frame, int32_t* best) {
    if (!scenes || n <= 0 || !cam || !prm || !outs) return fail(P3D_ERR_ARG, "NULL argument");
    for (int i = 0; i < n; i++) if (!scenes[i]) return fail(P3D_ERR_ARG, "scene is NULL");
    if (frames <= 0) frames = 3;
    if (best) *best = -1;
    if (ms_per_frame) for (int k = 0; k < 6; k++) ms_per_frame[k] = -1.0f;
    auto all = [&](int force) -> int {          // one frame on every handle, then wait for all of them
        for (int i = 0; i < n; i++) {
            scenes[i]->tune_force = force;
            int rc = p3d_render(scenes[i], cam, prm, &outs[i]);
            scenes[i]->tune_force = -1;
            if (rc) return rc;
        }
        return P3D_OK;
    };
    auto wait = [&]() -> int { for (int i = 0; i < n; i++) { int rc = p3d_sync(scenes[i]); if (rc) return rc; } return P3D_OK; };
    int rc = all(0);                            // also tells what there is to choose from
    if (rc == P3D_OK) rc = wait();
    if (rc) return rc;
    const int NC = scenes[0]->tune_candidates;
    if (NC == 0) return P3D_OK;                 // the schedule of this configuration is set by rule or by a flag: nothing to tune
    const uint32_t avail = scenes[0]->tune_avail;
    float ms[6] = {-1.0f, -1.0f, -1.0f, -1.0f, -1.0f, -1.0f};
    int win = -1;
    for (int c = 0; c < NC; c++) {
        if (!(avail & (1u << c))) continue;
        if ((rc = all(c)) != P3D_OK || (rc = wait()) != P3D_OK) return rc;                  // untimed: code objects, workspaces
        const auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) if ((rc = all(c)) != P3D_OK) return rc;
        if ((rc = wait()) != P3D_OK) return rc;
        ms[c] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() / (float)(frames * n);
        if (win < 0 || ms[c] < ms[win]) win = c;
    }
    for (int i = 0; i < n; i++) scenes[i]->tune_commit = win;
    if (scenes[0]->verbose)
        fprintf(stderr, "p3d: tuned choice, %d handle(s) in flight (ms per frame; wavefront / tree / tile, shared walks then private): %.3f %.3f %.3f | %.3f %.3f %.3f -> %d\n",
                n, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], win);
    if (ms_per_frame) memcpy(ms_per_frame, ms, sizeof ms);
    if (best) *best = win;
    return P3D_OK;
}

int p3d_debug_check_rcp(int device, uint32_t first_bits, uint64_t count, uint64_t* n_bad, uint32_t* first_bad) {
    if (!n_bad || !first_bad) return fail(P3D_ERR_ARG, "NULL argument");
    *n_bad = 0; *first_bad = 0xFFFFFFFFu;
    if (count == 0) return P3D_OK;
    if (count > (1ull << 32)) return fail(P3D_ERR_ARG, "count exceeds the 2^32 bit patterns");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(P3D_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    unsigned long long* d = nullptr;                      // [0] mismatches, [1] (low word) first mismatching pattern
    HIP_TRY(hipMalloc((void**)&d, 16));
    const unsigned long long init[2] = {0ull, 0xFFFFFFFFull};
    hipError_t e = hipMemcpy(d, init, 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_debug_check_rcp(first_bits, count, d, (uint32_t*)(d + 1), nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long out[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpy(out, d, 16, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(P3D_ERR_HIP, std::string("p3d_debug_check_rcp: ") + hipGetErrorString(e));
    *n_bad = out[0]; *first_bad = (uint32_t)out[1];
    return P3D_OK;
}

int p3d_debug_powf(int device, uint32_t n, const float* x, const float* y, float* out) {
    if (!x || !y || !out) return fail(P3D_ERR_ARG, "NULL argument");
    if (n == 0) return P3D_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(P3D_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipSetDevice(device));
    float* d = nullptr;                                   // x | y | out
    HIP_TRY(hipMalloc((void**)&d, (size_t)n * 12));
    hipError_t e = hipMemcpy(d, x, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, y, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_debug_powf(n, d, d + n, d + 2 * (size_t)n, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d + 2 * (size_t)n, (size_t)n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(P3D_ERR_HIP, std::string("p3d_debug_powf: ") + hipGetErrorString(e));
    return P3D_OK;
}

}  // extern "C"
