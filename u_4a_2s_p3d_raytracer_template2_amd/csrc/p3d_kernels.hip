// p3d_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Whitted hot path.
//
// Two schedules of the same per-node code (p3d_shade.h), bit-identical results:
//
//  * WAVEFRONT (default).  The reference's recursion rayTracing() (RT/main.cpp:530-721) is
//    unrolled by tree level: one launch per level traces one ray per lane; nodes that spawn
//    children are parked as 48-byte NodeRec, their child rays are compacted into the next
//    level's queue with wave ballot + mbcnt prefix sums (one atomic per wave), and resolve
//    launches walk the levels back up combining children into parents in the reference's
//    post-order arithmetic (the last level combines sibling pairs with their parent itself:
//    combine_pair).  Every launch is short and uniform: no lane waits for a
//    neighbour's deeper tree, there is no per-lane recursion stack, and the only LDS use is the
//    BVH traversal stack.
//  * TREE (P3D_FLAG_TREE_KERNEL, and the fallback when the worst-case queues would not fit):
//    one launch, each lane walks its pixel's whole tree with an explicit post-order frame
//    stack in LDS.
//
// A wave owns a 16x4 pixel tile; the blockIdx -> tile map hands chunks of tiles to XCDs.
// Numerics: compiled with -ffp-contract=off, no fast-math; see p3d_device_math.h.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "p3d_device_types.h"
#include "p3d_shade.h"

namespace p3d {

// ------------------------------------------------------------------ common helpers
__device__ __forceinline__ uint32_t lane_rank(uint64_t mask) {          // # set bits below this lane
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

template <bool COUNT>
__device__ __forceinline__ void flush_counters(const LaunchParams& P, const Ctr& ctr, uint32_t pixels) {
    if (COUNT) {
        DeviceCounters* c = P.counters;
        atomicAdd(&c->closest_queries, (unsigned long long)ctr.closest);
        atomicAdd(&c->shadow_queries, (unsigned long long)ctr.shadow);
        atomicAdd(&c->box_tests, (unsigned long long)ctr.box);
        atomicAdd(&c->sphere_tests, (unsigned long long)ctr.sph);
        atomicAdd(&c->tri_tests, (unsigned long long)ctr.tri);
        atomicAdd(&c->aabox_tests, (unsigned long long)ctr.aab);
        atomicAdd(&c->plane_tests, (unsigned long long)ctr.pln);
        atomicAdd(&c->pixels, (unsigned long long)pixels);
    }
}

// img_Data / colors of RT/main.cpp:803-815 for compact pixel p
__device__ __forceinline__ void write_pixel(const LaunchParams& P, size_t p, V3 color) {
    if (P.rgb8) {
        P.rgb8[3 * p] = (uint8_t)u8fromfloat(color.x);
        P.rgb8[3 * p + 1] = (uint8_t)u8fromfloat(color.y);
        P.rgb8[3 * p + 2] = (uint8_t)u8fromfloat(color.z);
    }
    if (P.rgb32f) { P.rgb32f[3 * p] = color.x; P.rgb32f[3 * p + 1] = color.y; P.rgb32f[3 * p + 2] = color.z; }
}

// One finished primary-ray tree: "rayTracing(...).clamp()", summed over samples and divided
// by 4*4 in the anti-aliased path (RT/main.cpp:774,797-800).
// In the anti-aliased path every sample pass writes its clamped colours to its own plane (so that the
// passes of a frame can run concurrently) and sum_samples_kernel adds the planes in sample order.
__device__ __forceinline__ void sink_sample(const LaunchParams& P, size_t p, V3 ret) {
    V3 c = clampc(ret);
    if (P.wf_nsamples <= 1 && P.spp == 0) { write_pixel(P, p, c); return; }
    float* a = P.wf_planes + (size_t)P.wf_sample * P.wf_plane_stride + 3 * p;
    a[0] = c.x; a[1] = c.y; a[2] = c.z;
}

// row of the compact local buffer -> image row (this rank's row blocks are every world-th one)
__device__ __forceinline__ int image_row(const LaunchParams& P, int row) {
    const int blk = P.row_block_shift >= 0 ? (row >> P.row_block_shift) : row / P.row_block;
    return (blk * P.world + P.rank) * P.row_block + (row - blk * P.row_block);
}

// "color += rayTracing(...).clamp()" over the samples in order, then "color / (4 * 4)"
// (RT/main.cpp:797-800), for the rows [row0, row0 + rows) of the compact buffer
__global__ __launch_bounds__(256) void sum_samples_kernel(const LaunchParams P, size_t first_px, size_t n_px) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = first_px + i;
        // rows past the image (the compact buffer is padded to whole row blocks) are never written:
        // a caller's device plane only holds res_y rows when world == 1 (include/p3d_hip.h)
        const int row = (int)(p / (size_t)P.res_x);
        if (image_row(P, row) >= P.res_y) continue;
        V3 acc = mk(0.0f, 0.0f, 0.0f);
        for (int smp = 0; smp < P.wf_nsamples; smp++) {
            const float* a = P.wf_planes + (size_t)smp * P.wf_plane_stride + 3 * p;
            acc = add(acc, mk(a[0], a[1], a[2]));
        }
        write_pixel(P, p, mk(fdiv(acc.x, 16.0f), fdiv(acc.y, 16.0f), fdiv(acc.z, 16.0f)));
    }
}

// tile -> pixel.  Returns false for lanes outside the image.
template <bool ORDERED = false>
__device__ __forceinline__ bool tile_pixel(const LaunchParams& P, int& x, int& y, int& row, int* tile_out = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    x = 0; y = 0; row = 0;
    int tx, ty, tile;
    if (!ORDERED && gridDim.y > 1) {
        // 2-D launch (identity tile map, xcd_chunk == 1): blockIdx.x / .y ARE the tile's column and row -- no division by
        // launch parameters on the scalar unit (three of them cost this kernel ~60 of its ~350 scalar instructions per wave)
        tx = blockIdx.x; ty = blockIdx.y;
        tile = ty * P.tiles_x + tx;
        if (tile_out) *tile_out = tile;
        ty += P.wf_tile_row0;
    } else {
        // XCD-aware tile map.  Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8
        // share one, each XCD has its own L2), so XCD k is given CHUNKS of xcd_chunk consecutive
        // tiles: chunk c goes to XCD c % 8.  xcd_chunk = 1 is the identity map (best load balance),
        // larger chunks trade balance for L2 locality on scenes whose BVH does not fit one L2.
        const int bid = blockIdx.x;
        const int j = bid >> 3;
        tile = P.xcd_chunk == 1 ? bid : ((j / P.xcd_chunk) * 8 + (bid & 7)) * P.xcd_chunk + (j % P.xcd_chunk);
        if (tile_out) *tile_out = tile;
        if (tile >= P.n_tiles) return false;
        if (ORDERED && P.tile_order) {            // heaviest first (scenes read from HBM, once a frame has measured the tiles)
            tile = (int)P.tile_order[tile];
            if (tile_out) *tile_out = tile;
        }
        tx = tile % P.tiles_x; ty = P.wf_tile_row0 + tile / P.tiles_x;
    }
    x = tx * 16 + (lane & 15);
    row = ty * (P.wg_waves * 4) + (lane >> 4) + wave * 4;         // row in the compact local buffer
    y = image_row(P, row);
    return x < P.res_x && y < P.res_y;
}

__device__ __forceinline__ Ray camera_ray(const LaunchParams& P, int x, int y, int sample) {
    if (P.spp == 0) return primary_ray_tab(P, x, y);                                  // RT/main.cpp:756-772
    const int ns = P.spp * P.spp;                                                     // RT/main.cpp:776-795
    const float4 sm = reinterpret_cast<const float4*>(P.samples)[((size_t)y * P.res_x + x) * ns + sample];
    return primary_ray_lens(P, sm.z, sm.w, sm.x, sm.y);
}

// ------------------------------------------------------------------ WAVEFRONT schedule
// One shard's slice of the level queues (see LaunchParams::wf_shards).
struct Shard {
    const RayRec* rays_in; uint32_t count_in;
    RayRec* rays_out; uint32_t* count_out;
    NodeRec* nodes_parent; NodeRec* nodes_self; uint32_t* ncount_self;
    NodeRec* nodes_grand;                          // level wf_level - 2 (pair mode)
    const uint32_t* rng_in; uint32_t* rng_out;     // random-stream keys of the queued rays, or nullptr
};
// the counter arrays of this launch under pass parity `par` (see LaunchParams::wf_alt)
__device__ __forceinline__ const uint32_t* count_in_array(const LaunchParams& P, uint32_t par) {
    return P.wf_level == 2 ? P.wf_alt + (size_t)(par * 2u) * kWfShards : P.wf_count_in;
}
__device__ __forceinline__ uint32_t* count_out_array(const LaunchParams& P, uint32_t par) {
    return P.wf_level == 1 ? P.wf_alt + (size_t)(par * 2u) * kWfShards : P.wf_count_out;
}
__device__ __forceinline__ uint32_t* ncount_self_array(const LaunchParams& P, uint32_t par) {
    return P.wf_level == 1 ? P.wf_alt + (size_t)(par * 2u + 1u) * kWfShards : P.wf_ncount_self;
}
__device__ __forceinline__ Shard shard_of(const LaunchParams& P, uint32_t s, uint32_t par) {
    Shard h;
    h.rays_in = P.wf_rays_in ? P.wf_rays_in + (size_t)s * P.wf_cap_in : nullptr;
    h.count_in = P.wf_rays_in ? count_in_array(P, par)[s] : 0u;
    h.rays_out = P.wf_rays_out ? P.wf_rays_out + (size_t)s * P.wf_cap_out : nullptr;
    h.count_out = count_out_array(P, par) + s;
    h.nodes_parent = P.wf_nodes_parent ? P.wf_nodes_parent + (size_t)s * P.wf_ncap_parent : nullptr;
    h.nodes_self = P.wf_nodes_self ? P.wf_nodes_self + (size_t)s * P.wf_ncap_self : nullptr;
    h.nodes_grand = P.wf_nodes_grand ? P.wf_nodes_grand + (size_t)s * P.wf_ncap_grand : nullptr;
    h.ncount_self = ncount_self_array(P, par) + s;
    h.rng_in = P.wf_rng_in ? P.wf_rng_in + (size_t)s * P.wf_cap_in : nullptr;
    h.rng_out = P.wf_rng_out ? P.wf_rng_out + (size_t)s * P.wf_cap_out : nullptr;
    return h;
}

// Hand a finished node's return value to whoever waits for it.
__device__ __forceinline__ void deliver(const LaunchParams& P, const Shard& sh, int level, uint32_t link, V3 ret) {
    if (level == 1) { sink_sample(P, (size_t)link, ret); return; }
    NodeRec* parent = sh.nodes_parent + (link & ~kLinkRefr);
    float* dst = (link & kLinkRefr) ? parent->refr_ret : parent->refl_ret;
    dst[0] = ret.x; dst[1] = ret.y; dst[2] = ret.z;
}

// Park a node with children and queue its child rays.  Must be reached by ALL lanes of the
// wave together (converged): slots are handed out with ballot + mbcnt prefix sums and one
// atomic per counter per wave.
__device__ __forceinline__ void emit(const LaunchParams& P, const Shard& sh, int level, bool valid, uint32_t link,
                                     float ior_1, const NodeOut& o) {
    const int lane = threadIdx.x & 63;
    if (valid && o.terminal) deliver(P, sh, level, link, o.ret);
    const bool parks = valid && !o.terminal;
    const uint64_t m_node = __ballot(parks);
    if (m_node == 0) return;                                   // wave-uniform
    const uint64_t m_refl = __ballot(parks && o.has_refl);
    const uint64_t m_refr = __ballot(parks && o.has_refr);
    const uint32_t n_refl = (uint32_t)__popcll(m_refl), n_refr = (uint32_t)__popcll(m_refr);
    const bool pairs = P.wf_pair_out != 0;                     // sibling pairs in even / odd slots for the last level
    uint32_t node_base = 0, ray_base = 0;
    if (lane == (int)__builtin_ctzll(m_node)) {
        node_base = atomicAdd(sh.ncount_self, (uint32_t)__popcll(m_node));
        ray_base = atomicAdd(sh.count_out, pairs ? 2u * (uint32_t)__popcll(m_node) : n_refl + n_refr);
    }
    node_base = __shfl(node_base, (int)__builtin_ctzll(m_node));
    ray_base = __shfl(ray_base, (int)__builtin_ctzll(m_node));
    if (!parks) return;
    const uint32_t my_node = node_base + lane_rank(m_node);
    float4* nd = reinterpret_cast<float4*>(sh.nodes_self + my_node);
    nd[0] = make_float4(o.color.x, o.color.y, o.color.z, o.KR);
    nd[1] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(o.mat));
    nd[2] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(link));
    if (pairs) {
        const uint32_t slot = ray_base + 2u * lane_rank(m_node);
        float4* rq = reinterpret_cast<float4*>(sh.rays_out + slot);
        const float4 z = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kPairEmpty));
        rq[0] = o.has_refl ? make_float4(o.refl.o.x, o.refl.o.y, o.refl.o.z, ior_1) : z;
        rq[1] = o.has_refl ? make_float4(o.refl.d.x, o.refl.d.y, o.refl.d.z, __uint_as_float(my_node)) : z;
        rq[2] = o.has_refr ? make_float4(o.refr.o.x, o.refr.o.y, o.refr.o.z, o.newIor) : z;
        rq[3] = o.has_refr ? make_float4(o.refr.d.x, o.refr.d.y, o.refr.d.z, __uint_as_float(my_node | kLinkRefr)) : z;
        if (sh.rng_out) { sh.rng_out[slot] = o.rng_refl; sh.rng_out[slot + 1] = o.rng_refr; }
        return;
    }
    if (o.has_refl) {                                           // reflection child keeps ior_1
        const uint32_t slot = ray_base + lane_rank(m_refl);
        float4* rq = reinterpret_cast<float4*>(sh.rays_out + slot);
        rq[0] = make_float4(o.refl.o.x, o.refl.o.y, o.refl.o.z, ior_1);
        rq[1] = make_float4(o.refl.d.x, o.refl.d.y, o.refl.d.z, __uint_as_float(my_node));
        if (sh.rng_out) sh.rng_out[slot] = o.rng_refl;
    }
    if (o.has_refr) {
        const uint32_t slot = ray_base + n_refl + lane_rank(m_refr);
        float4* rq = reinterpret_cast<float4*>(sh.rays_out + slot);
        rq[0] = make_float4(o.refr.o.x, o.refr.o.y, o.refr.o.z, o.newIor);
        rq[1] = make_float4(o.refr.d.x, o.refr.d.y, o.refr.d.z, __uint_as_float(my_node | kLinkRefr));
        if (sh.rng_out) sh.rng_out[slot] = o.rng_refr;
    }
}

// Scene view of this launch.  LDS variant: the workgroup first copies the blob into LDS (all
// threads, then one barrier -- the only barrier of the launch; call before any early exit).
template <bool LDS> struct View;
template <> struct View<false> {
    typedef GlobalScene type;
    static __device__ __forceinline__ GlobalScene make(const LaunchParams& P) {
        GlobalScene g; g.q = reinterpret_cast<const float4*>(P.blob);
        g.o = SceneOffsets{P.off_nodes, P.off_leaves, P.off_spheres, P.off_sphere_meta, P.off_tris, P.off_tri_normals, P.off_boxes, P.off_mats, P.tri_quads};
        g.qn = reinterpret_cast<const uint4*>(P.qnodes);
        for (int a = 0; a < 3; a++) { g.qs[a] = P.q_scale[a]; g.qb[a] = P.q_base[a]; }
        return g;
    }
    static __device__ __forceinline__ GlobalScene make_shading(const LaunchParams& P) { return make(P); }
    static __device__ __forceinline__ uint32_t scene_dwords(const LaunchParams&) { return 0; }
};
template <> struct View<true> {
    typedef LdsScene type;
    static __device__ __forceinline__ LdsScene make(const LaunchParams& P) {
        const float4* src = reinterpret_cast<const float4*>(P.blob);
        float4* dst = reinterpret_cast<float4*>(p3d_lds);
        for (uint32_t i = threadIdx.x; i < P.blob_quads; i += blockDim.x) dst[i] = src[i];
        __syncthreads();
        LdsScene l;
        l.o = SceneOffsets{P.off_nodes, P.off_leaves, P.off_spheres, P.off_sphere_meta, P.off_tris, P.off_tri_normals, P.off_boxes, P.off_mats, P.tri_quads};
        return l;
    }
    static __device__ __forceinline__ LdsScene make_shading(const LaunchParams& P) { return make(P); }
    static __device__ __forceinline__ uint32_t scene_dwords(const LaunchParams& P) { return P.blob_quads * 4; }
};

// Last level in pair mode (LaunchParams::wf_pair_in): every ray of the level has returned (o.ret), lanes 2k / 2k+1 hold
// the reflection / refraction child of one level-(D-1) node.  The even lane combines them with the node's parked record
// -- "color += reflection_color * KR * specColor + refraction_color * (1 - KR)", RT/main.cpp:719, a never-traced child
// adds zero -- and hands the result one level further up.  Must be reached by all lanes of the wave together.
__device__ __forceinline__ void combine_pair(const LaunchParams& P, const Shard& sh, bool valid, uint32_t link, const NodeOut& o) {
    const int lane = threadIdx.x & 63;
    const V3 mine = valid ? o.ret : mk(0.0f, 0.0f, 0.0f);
    const V3 other = mk(__shfl_xor(mine.x, 1), __shfl_xor(mine.y, 1), __shfl_xor(mine.z, 1));
    const uint32_t other_link = (uint32_t)__shfl_xor((int)link, 1);
    const bool other_valid = __shfl_xor(valid ? 1 : 0, 1) != 0;
    if ((lane & 1) != 0 || !(valid || other_valid)) return;
    const uint32_t parent = (valid ? link : other_link) & ~kLinkRefr;
    const GlobalScene gv = View<false>::make(P);
    const float4* nd = reinterpret_cast<const float4*>(sh.nodes_parent + parent);
    const float4 a = nd[0], b = nd[1], c = nd[2];
    const Mtl M = load_material(gv, __float_as_uint(b.w));
    const V3 ret = combine_node(mk(a.x, a.y, a.z), a.w, M.spec, mine, other);
    const uint32_t up = __float_as_uint(c.w);
    if (P.wf_level == 2) { sink_sample(P, (size_t)up, ret); return; }
    NodeRec* g = sh.nodes_grand + (up & ~kLinkRefr);
    float* dst = (up & kLinkRefr) ? g->refr_ret : g->refl_ret;
    dst[0] = ret.x; dst[1] = ret.y; dst[2] = ret.z;
}

// this wave's traversal stack: after the (optional) scene copy, one region per wave
template <bool LDS>
__device__ __forceinline__ TravCtx wave_stack(const LaunchParams& P, uint32_t extra_dwords_per_wave, uint32_t** wave_base = nullptr) {
    const uint32_t wave = threadIdx.x >> 6;
    uint32_t* base = p3d_lds + View<LDS>::scene_dwords(P) + wave * (P.trav_stack_dwords + extra_dwords_per_wave);
    if (wave_base) *wave_base = base;
    TravCtx tc;
    tc.lane.region = base; tc.lane.lane = threadIdx.x & 63; tc.lane.slots = P.trav_stack_entries;
    tc.wave.base = reinterpret_cast<int32_t*>(base);     // the two walks never run in the same launch
    tc.share = base + P.trav_stack_dwords - kShareDwords; // work-sharing walk: the last kShareDwords of the wave's region (host: p3d_render)
    return tc;
}

// diagnostic stamps (only when a stamp buffer was set with p3d_debug_set_stamps): slot k of the
// record of (tile, wave) gets the 100 MHz real-time counter; slot 7 the hardware id registers
__device__ __forceinline__ void stamp_record(unsigned long long* r, int k) {
    r[k] = __builtin_amdgcn_s_memrealtime();
    if (k == 0) r[7] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
                       ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);
}
// ALL stamp hooks exist only in builds with -DP3D_STAMPS (`make stamps`, libp3d_hip_stamps.so; the timeline tools select it):
// each hook is a scalar compare and branch per wave on kernels that are bound by scalar issue (the level-1 kernel of
// config 2: 5 hooks = 20 of its 290 scalar instructions per wave).
#ifdef P3D_STAMPS
constexpr bool kStamps = true;
#else
constexpr bool kStamps = false;
#endif
__device__ __forceinline__ void stamp(const LaunchParams& P, int tile, int k) {
    if (kStamps && P.dbg_stamps && P.dbg_stamp_level <= 1 && (threadIdx.x & 63) == 0)
        stamp_record(P.dbg_stamps + ((size_t)tile * P.wg_waves + (threadIdx.x >> 6)) * 8, k);
}
// the same for the deeper-level kernel (p3d_debug_set_stamp_level(l), l >= 2): one record per WAVE of the launch, written
// for the wave's first batch -- 0 start, 1 queue read, 2 closest hit, 3 shading, 4 emit / pair combine -- then 5 = the wave
// is done and 6 = the number of batches it ran
// (only in builds with -DP3D_STAMPS -- `make stamps`, libp3d_hip_stamps.so, tools/wave_timeline.py: the checks cost the
//  deeper-level kernel 25 spilled scalar registers, and that kernel is bound by instruction issue)
__device__ __forceinline__ bool stamps_on(const LaunchParams& P) {
#ifdef P3D_STAMPS
    return P.dbg_stamps && P.dbg_stamp_level == P.wf_level && (threadIdx.x & 63) == 0;
#else
    return false;
#endif
}
__device__ __forceinline__ void stamp_wave(const LaunchParams& P, uint32_t wave_id, int k, bool first = true) {
    if (first && stamps_on(P)) stamp_record(P.dbg_stamps + (size_t)wave_id * 8, k);
}

// level 1: camera rays of one sample pass over a band of tiles
// OCC = requested waves per SIMD (amdgpu_waves_per_eu): caps the VGPR allocation so that more
// waves hide each other's latency, at the price of a few spilled registers.  Selected at run
// time by p3d_set_tuning(); never changes results.
#ifndef P3D_OCC_FLOOR
#define P3D_OCC_FLOOR 1          // build-time experiment knob: minimum waves per SIMD of every ray kernel
#endif
#define P3D_OCC(OCC) __attribute__((amdgpu_waves_per_eu(((OCC) > P3D_OCC_FLOOR ? (OCC) : P3D_OCC_FLOOR), 8)))

// One workgroup per 16 x (4 x wg_waves) tile, dispatched by the hardware.  (A persistent variant -- resident-sized
// grid, scene copied once per workgroup, every wave drawing 16x4 tiles from device counters with the next number
// prefetched -- was measured and dropped: 520 us with one counter (a word saturates at ~88 returning atomics per
// microsecond), 110 us with 64 counters on separate lines, against 50 us for this plain grid.)
// (Measured and dropped in round 3, profiles/r03_exp03_sharing_wg_levers.txt / r03_exp04_tiles_lpt_bound.txt: 8- and 16-wave
//  workgroups -- level 1 of config 2 46 -> 51 -> 54 us -- and 2-4 tiles per workgroup one after the other: the waves of
//  this launch start at ~830 per microsecond whatever the workgroup shape, and the loop's registers cost occupancy.)
template <bool COUNT, bool LDS, int WALK, int OCC, bool STOCH = false>
__global__ __launch_bounds__(LDS ? 256 : 64) P3D_OCC(OCC) void wf_primary_kernel(const LaunchParams P) {
    const uint32_t par = P.wf_ctrl[0] & 1u;                     // this pass's counter set (LaunchParams::wf_alt)
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (threadIdx.x == 0) P.wf_ctrl[32] = par;
        for (uint32_t i = threadIdx.x; i < P.wf_clear_words; i += blockDim.x) P.wf_clear[i] = 0u;
        uint32_t* other = P.wf_alt + (size_t)((1u - par) * 2u) * kWfShards;
        for (uint32_t i = threadIdx.x; i < 2u * kWfShards; i += blockDim.x) other[i] = 0u;
    }
    const typename View<LDS>::type sv = View<LDS>::make_shading(P);
    int x, y, row, tile;
    const bool valid = tile_pixel<!LDS>(P, x, y, row, &tile);
    if (__ballot(valid) == 0) return;
    const Shard sh = shard_of(P, (uint32_t)tile % kWfShards, par);
    const TravCtx tc = wave_stack<LDS>(P, 0);
    Ctr ctr = {0, 0, 0, 0, 0, 0, 0};
    const size_t p = (size_t)row * P.res_x + x;
    const unsigned long long t_tile = (!LDS && P.tile_cost) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    stamp(P, tile, 0);
    Ray ray; ray.o = mk(0.0f, 0.0f, 0.0f); ray.d = mk(1.0f, 0.0f, 0.0f);
    if (valid) ray = camera_ray(P, x, y, P.wf_sample);
    stamp(P, tile, 1);
    const Hit h = find_closest<COUNT, WALK>(P, sv, ray, valid, tc, ctr);
    stamp(P, tile, 2);
    if (valid && P.hit_id && P.wf_sample == 0) P.hit_id[p] = (h.ref == 0xFFFFFFFFu) ? -1 : (int32_t)h.sid;
    // the random stream of a pixel sample is keyed by the pixel's place in the FULL frame, so a frame
    // sharded over several GPUs draws the same numbers as on one
    const uint32_t rng = STOCH ? rng_mix(rng_mix(P.seed, (uint32_t)(y * P.res_x + x)), (uint32_t)P.wf_sample) : 0u;
    const NodeOut o = shade_hit<COUNT, WALK, typename View<LDS>::type, STOCH>(P, sv, ray, h, valid, 1, 1.0f, tc, ctr, rng);
    stamp(P, tile, 3);
    emit(P, sh, 1, valid, (uint32_t)p, 1.0f, o);
    stamp(P, tile, 4);
    if (!LDS && P.tile_cost && threadIdx.x == 0) P.tile_cost[tile] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_tile);
    if (valid) flush_counters<COUNT>(P, ctr, P.wf_sample == 0 ? 1u : 0u);
}

// Lanes a wave of a deeper level uses: a short queue is spread over ALL the shard's waves with
// fewer rays each instead of filling 64-wide waves.  Deeper levels hold few, incoherent rays and
// are bound by the latency of one wave's ray step, not by issue slots: narrow waves diverge less
// (the step lasts as long as the slowest of its lanes) and more of them are in flight to hide
// each other's memory latency.  Scenes served from LDS keep full waves (wf_min_width = 64):
// there the extra waves only cost issue slots (measured: 0.14 -> 0.20 ms on config 2).
__device__ __forceinline__ uint32_t wave_width(uint32_t count, uint32_t waves_per_shard, uint32_t min_width) {
    // narrow only while the rays still fit the waves that can be RESIDENT at once (a quarter of the
    // launched ones): beyond that, narrower waves just spend issue slots on idle lanes
    const unsigned long long resident = waves_per_shard >= 4 ? waves_per_shard / 4 : 1;
    uint32_t width = 64;
    while (width > min_width && resident * (width >> 1) >= count) width >>= 1;
    return width;
}

// level >= 2: one queued ray per lane, persistent waves striding over the queue
template <bool COUNT, bool LDS, int WALK, int OCC, bool STOCH = false>
__global__ __launch_bounds__(LDS ? 256 : 64) P3D_OCC(OCC) void wf_secondary_kernel(const LaunchParams P) {
    constexpr uint32_t S = kWfShards;
    const uint32_t wave_id = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t par = P.wf_ctrl[32] & 1u;                    // set by the level-1 launch of this pass
    if (P.wf_level == 2 && blockIdx.x == 0 && threadIdx.x == 0) P.wf_ctrl[0] = 1u - par;   // the next pass takes the other set
    const uint32_t* counts_in = count_in_array(P, par);
    if (LDS) {
        // Scenes served from LDS: full 64-ray batches, numbered THROUGH all shards (the host launches S == 64
        // shards, one count per lane: batches per shard, wave-wide prefix sum), batch b goes to wave b % n_waves.
        // A deeper level of a 1080p frame is ~1.15 batches per resident-at-4-per-SIMD wave, and a launch lasts as
        // long as its busiest wave: with the grid sized to what can be RESIDENT (host: occupancy query) and every
        // wave owning at most one batch whichever shard it is in, the level costs one ray step instead of two.
        const uint32_t c = (uint32_t)lane < S ? counts_in[lane] : 0u;
        const uint32_t nb = (c + 63u) >> 6;
        uint32_t incl = nb;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        const uint32_t total = __shfl(incl, 63);
        if (((blockIdx.x * blockDim.x) >> 6) >= total) return;      // workgroup-uniform, before the scene copy's barrier
        const typename View<LDS>::type sv = View<LDS>::make_shading(P);
        const TravCtx tc = wave_stack<LDS>(P, 0);
        Ctr ctr = {0, 0, 0, 0, 0, 0, 0};
        uint32_t n_batches = 0;
        for (uint32_t b = wave_id; b < total; b += n_waves, n_batches++) {
            const bool st1 = n_batches == 0;
            stamp_wave(P, wave_id, 0, st1);
            const int s = (int)__builtin_ctzll(__ballot(incl > b));  // the shard batch b belongs to
            const uint32_t first = __shfl(incl - nb, s);             // batches in the shards before it
            const Shard sh = shard_of(P, (uint32_t)s, par);
            const uint32_t i = (b - first) * 64u + lane;
            const bool valid = i < sh.count_in;
            uint32_t link = 0, rng = 0; float ior_1 = 1.0f;
            Ray ray; ray.o = mk(0.0f, 0.0f, 0.0f); ray.d = mk(1.0f, 0.0f, 0.0f);
            if (valid) {
                const float4* rq = reinterpret_cast<const float4*>(sh.rays_in + i);
                float4 a = rq[0], bq = rq[1];
                ray.o = mk(a.x, a.y, a.z); ray.d = mk(bq.x, bq.y, bq.z);
                ior_1 = a.w; link = __float_as_uint(bq.w);
                if (STOCH) rng = sh.rng_in[i];
            }
            const bool live = valid && link != kPairEmpty;           // (the unused half of a sibling pair)
            if (stamps_on(P) && st1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp_wave(P, wave_id, 1); }
            const Hit h = find_closest<COUNT, WALK>(P, sv, ray, live, tc, ctr);
            stamp_wave(P, wave_id, 2, st1);
            const NodeOut o = shade_hit<COUNT, WALK, typename View<LDS>::type, STOCH>(P, sv, ray, h, live, P.wf_level, ior_1, tc,
                                                                                    ctr, rng);
            stamp_wave(P, wave_id, 3, st1);
            if (P.wf_pair_in) combine_pair(P, sh, live, link, o);
            else emit(P, sh, P.wf_level, live, link, ior_1, o);
            stamp_wave(P, wave_id, 4, st1);
        }
        if (stamps_on(P) && n_batches) { stamp_wave(P, wave_id, 5); P.dbg_stamps[(size_t)wave_id * 8 + 6] = n_batches; }
        flush_counters<COUNT>(P, ctr, 0u);
        return;
    }
    // Scenes read from HBM: wave g works on shard g % S; the (gridwaves / S) waves of a shard stride over its
    // queue, with narrow waves when the queue is short (wave_width)
    const uint32_t per_shard = n_waves / S;
    {   // workgroup-uniform early exit (before the scene copy's barrier): nothing queued for any of
        // this workgroup's waves
        const uint32_t w0 = (blockIdx.x * blockDim.x) >> 6, nw = blockDim.x >> 6;
        bool any = false;
        for (uint32_t w = w0; w < w0 + nw; w++) {
            const uint32_t c = counts_in[w % S];
            if ((w / S) * wave_width(c, per_shard, (uint32_t)P.wf_min_width) < c) any = true;
        }
        if (!any) return;
    }
    const typename View<LDS>::type sv = View<LDS>::make_shading(P);
    const Shard sh = shard_of(P, wave_id % S, par);
    const TravCtx tc = wave_stack<LDS>(P, 0);
    Ctr ctr = {0, 0, 0, 0, 0, 0, 0};
    const uint32_t width = wave_width(sh.count_in, per_shard, (uint32_t)P.wf_min_width);
    uint32_t n_batches = 0;
    for (uint32_t base = (wave_id / S) * width; base < sh.count_in; base += per_shard * width, n_batches++) {
        const bool st1 = n_batches == 0;
        stamp_wave(P, wave_id, 0, st1);
        const uint32_t i = base + lane;
        const bool valid = (uint32_t)lane < width && i < sh.count_in;
        uint32_t link = 0, rng = 0; float ior_1 = 1.0f;
        Ray ray; ray.o = mk(0.0f, 0.0f, 0.0f); ray.d = mk(1.0f, 0.0f, 0.0f);
        if (valid) {
            const float4* rq = reinterpret_cast<const float4*>(sh.rays_in + i);
            float4 a = rq[0], b = rq[1];
            ray.o = mk(a.x, a.y, a.z); ray.d = mk(b.x, b.y, b.z);
            ior_1 = a.w; link = __float_as_uint(b.w);
            if (STOCH) rng = sh.rng_in[i];
        }
        const bool live = valid && link != kPairEmpty;
        const Hit h = find_closest<COUNT, WALK>(P, sv, ray, live, tc, ctr);
        stamp_wave(P, wave_id, 2, st1);
        const NodeOut o = shade_hit<COUNT, WALK, typename View<LDS>::type, STOCH>(P, sv, ray, h, live, P.wf_level, ior_1, tc,
                                                                                ctr, rng);
        stamp_wave(P, wave_id, 3, st1);
        if (P.wf_pair_in) combine_pair(P, sh, live, link, o);
        else emit(P, sh, P.wf_level, live, link, ior_1, o);
        stamp_wave(P, wave_id, 4, st1);
    }
    if (stamps_on(P) && n_batches) { stamp_wave(P, wave_id, 5); P.dbg_stamps[(size_t)wave_id * 8 + 6] = n_batches; }
    flush_counters<COUNT>(P, ctr, 0u);
}

// walk one level back up: node = color + (refl_ret*KR*spec + refr_ret*(1-KR)), RT/main.cpp:719
__global__ __launch_bounds__(256) void wf_resolve_kernel(const LaunchParams P) {
    constexpr uint32_t S = kWfShards;
    const Shard sh = shard_of(P, blockIdx.x % S, P.wf_ctrl[32] & 1u);
    const uint32_t count = *sh.ncount_self;
    const uint32_t per_shard = gridDim.x / S;
    const GlobalScene gv = View<false>::make(P);
    for (uint32_t i = (blockIdx.x / S) * blockDim.x + threadIdx.x; i < count; i += per_shard * blockDim.x) {
        const float4* nd = reinterpret_cast<const float4*>(sh.nodes_self + i);
        float4 a = nd[0], b = nd[1], c = nd[2];
        Mtl M = load_material(gv, __float_as_uint(b.w));
        V3 ret = combine_node(mk(a.x, a.y, a.z), a.w, M.spec, mk(b.x, b.y, b.z), mk(c.x, c.y, c.z));
        deliver(P, sh, P.wf_level, __float_as_uint(c.w), ret);
    }
}

// The same for ALL levels in one launch, one 1024-thread workgroup per shard: a ray never leaves its pixel's shard, so a
// shard's levels only depend on each other and a __syncthreads() between levels replaces the launch boundary.  For
// small frames -- a rank's share of a frame tiled over several GPUs -- where a level holds a few nodes per thread and a
// frame is bound by the number of launches (tools/shard_probe.py); large frames keep one wide launch per level.
__global__ __launch_bounds__(1024) void wf_resolve_fused_kernel(const LaunchParams P, const ResolveLevels R) {
    const uint32_t s = blockIdx.x, par = P.wf_ctrl[32] & 1u;
    const GlobalScene gv = View<false>::make(P);
    for (int l = R.top; l >= 1; l--) {
        Shard sh;
        sh.rays_in = nullptr; sh.count_in = 0; sh.rays_out = nullptr; sh.count_out = nullptr; sh.rng_in = nullptr; sh.rng_out = nullptr;
        sh.nodes_self = R.nodes[l] + (size_t)s * R.cap[l];
        sh.nodes_parent = l > 1 ? R.nodes[l - 1] + (size_t)s * R.cap[l - 1] : nullptr;
        sh.ncount_self = nullptr;
        const uint32_t count = l == 1 ? (P.wf_alt + (size_t)(par * 2u + 1u) * kWfShards)[s] : R.ncount[l][s];
        for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
            const float4* nd = reinterpret_cast<const float4*>(sh.nodes_self + i);
            float4 a = nd[0], b = nd[1], c = nd[2];
            Mtl M = load_material(gv, __float_as_uint(b.w));
            V3 ret = combine_node(mk(a.x, a.y, a.z), a.w, M.spec, mk(b.x, b.y, b.z), mk(c.x, c.y, c.z));
            deliver(P, sh, l, __float_as_uint(c.w), ret);
        }
        __syncthreads();             // the workgroup's own stores are visible to it after the barrier
    }
}

// ------------------------------------------------------------------ TILE schedule
// ONE launch per frame.  The wavefront schedule above spends six of its seven launches on the deeper
// levels of a 1080p frame, each lasting as long as one or two incoherent ray steps of a wave whatever
// the number of rays (launch ramp + quantisation: 1.15 batches per wave rounds up to 2), and hands rays
// from level to level through device-scope atomics and HBM-sized worst-case queues.  A ray never leaves
// its pixel's tile, so here a 256-thread workgroup keeps a 16x16 tile's whole tree to itself: level 1
// traces the camera rays, child rays are compacted (ballot + mbcnt, one LDS atomic per counter per wave)
// into the workgroup's PRIVATE slot of the workspace, the four waves share each deeper level's batches,
// and the resolve passes walk the levels back up -- all between __syncthreads(), no global atomics, no
// cross-CU hand-off (a workgroup's own stores are visible to it after the barrier).  Workgroups are
// persistent and draw tiles from one counter, so cheap sky tiles flow past expensive glass tiles and the
// workspace is (resident workgroups) x (one tile's worst case): ~200 MB at depth 4 instead of gigabytes.
// Samples of the anti-aliased path run back to back inside the tile: "color += rayTracing().clamp()" in
// sample order is an LDS accumulator, there are no per-sample planes and no summing launch.
// Same shade_hit()/combine_node() as the other schedules: bit-identical frames.
struct TileLds {
    uint32_t n_rays[kMaxTileLevels];      // rays queued for level l (2..D)
    uint32_t n_nodes[kMaxTileLevels];     // nodes parked at level l (1..D-1)
    uint32_t tile, pad;
    float acc[3 * kTilePx];               // sum of the clamped sample colours of each pixel (spp > 0)
};

// LDS scenes: the first kTileLdsRays queued rays of a level stay in LDS -- two buffers, written and read alternately
// by consecutive levels (a level's own barrier separates them); only what does not fit goes through the workgroup's
// slot in HBM.  A 16x16 tile of config 4 queues ~750 rays per sample pass, ~300 of them for its largest level: the
// ray half of the queue traffic (11.6 GB per frame in round 2, profiles/r02_config4_pmc.json) stays on the CU.
constexpr uint32_t kTileLdsRays = 256;
constexpr uint32_t kTileLdsRayDwords = 2u * kTileLdsRays * 8u;        // 16 KB

struct TileCtx {
    uint32_t lds_off;                     // dword offset of the TileLds inside p3d_lds
    uint32_t lq_off;                      // dword offset of the LDS ray buffers (0 = none: scenes read from HBM)
    // ray slot `i` of level `l`: in LDS below kTileLdsRays (when the kernel has the buffers), else in the HBM slot
    __device__ __forceinline__ float4* ray_slot(int l, uint32_t i) const {
        if (lq_off != 0u && i < kTileLdsRays)
            return reinterpret_cast<float4*>(p3d_lds + lq_off) + ((uint32_t)(l & 1) * kTileLdsRays + i) * 2u;
        return reinterpret_cast<float4*>(rays + tile_ray_offset(l) + i);
    }
    RayRec* rays; NodeRec* nodes; uint32_t* keys;     // this workgroup's slot
    int tx, ty;                           // tile coordinates
    __device__ __forceinline__ TileLds* lds() const { return reinterpret_cast<TileLds*>(p3d_lds + lds_off); }
};

// compact pixel index of tile-local pixel `link` (= the thread id that traced it)
__device__ __forceinline__ size_t tile_pixel_index(const LaunchParams& P, const TileCtx& X, uint32_t link) {
    const int x = X.tx * 16 + (int)(link & 15u);
    const int row = X.ty * 16 + (int)(link >> 6) * 4 + (int)((link >> 4) & 3u);
    return (size_t)row * P.res_x + x;
}

__device__ __forceinline__ void tile_deliver(const LaunchParams& P, const TileCtx& X, int level, uint32_t link, V3 ret) {
    if (level == 1) {                                            // "rayTracing(...).clamp()" of a pixel sample
        const V3 c = clampc(ret);
        if (P.spp == 0) { write_pixel(P, tile_pixel_index(P, X, link), c); return; }
        float* a = X.lds()->acc + 3 * link;                      // color += ... in sample order (RT/main.cpp:797)
        a[0] = a[0] + c.x; a[1] = a[1] + c.y; a[2] = a[2] + c.z;
        return;
    }
    NodeRec* parent = X.nodes + tile_node_offset(level - 1) + (link & ~kLinkRefr);
    float* dst = (link & kLinkRefr) ? parent->refr_ret : parent->refl_ret;
    dst[0] = ret.x; dst[1] = ret.y; dst[2] = ret.z;
}

// like emit(): must be reached by all lanes of the wave together
__device__ __forceinline__ void tile_emit(const LaunchParams& P, const TileCtx& X, int level, bool valid, uint32_t link,
                                          float ior_1, const NodeOut& o) {
    const int lane = threadIdx.x & 63;
    if (valid && o.terminal) tile_deliver(P, X, level, link, o.ret);
    const bool parks = valid && !o.terminal;
    const uint64_t m_node = __ballot(parks);
    if (m_node == 0) return;                                   // wave-uniform
    const uint64_t m_refl = __ballot(parks && o.has_refl);
    const uint64_t m_refr = __ballot(parks && o.has_refr);
    const uint32_t n_refl = (uint32_t)__popcll(m_refl), n_refr = (uint32_t)__popcll(m_refr);
    uint32_t node_base = 0, ray_base = 0;
    const int first = (int)__builtin_ctzll(m_node);
    if (lane == first) {
        TileLds* T = X.lds();
        node_base = atomicAdd(&T->n_nodes[level], (uint32_t)__popcll(m_node));
        ray_base = atomicAdd(&T->n_rays[level + 1], n_refl + n_refr);
    }
    node_base = __shfl(node_base, first);
    ray_base = __shfl(ray_base, first);
    if (!parks) return;
    const uint32_t my_node = node_base + lane_rank(m_node);
    float4* nd = reinterpret_cast<float4*>(X.nodes + tile_node_offset(level) + my_node);
    nd[0] = make_float4(o.color.x, o.color.y, o.color.z, o.KR);
    nd[1] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(o.mat));
    nd[2] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(link));
    if (o.has_refl) {                                           // reflection child keeps ior_1
        const uint32_t slot = ray_base + lane_rank(m_refl);
        float4* rq = X.ray_slot(level + 1, slot);
        rq[0] = make_float4(o.refl.o.x, o.refl.o.y, o.refl.o.z, ior_1);
        rq[1] = make_float4(o.refl.d.x, o.refl.d.y, o.refl.d.z, __uint_as_float(my_node));
        if (X.keys) X.keys[tile_ray_offset(level + 1) + slot] = o.rng_refl;
    }
    if (o.has_refr) {
        const uint32_t slot = ray_base + n_refl + lane_rank(m_refr);
        float4* rq = X.ray_slot(level + 1, slot);
        rq[0] = make_float4(o.refr.o.x, o.refr.o.y, o.refr.o.z, o.newIor);
        rq[1] = make_float4(o.refr.d.x, o.refr.d.y, o.refr.d.z, __uint_as_float(my_node | kLinkRefr));
        if (X.keys) X.keys[tile_ray_offset(level + 1) + slot] = o.rng_refr;
    }
}

template <bool COUNT, bool LDS, int WALK, int OCC, bool STOCH = false>
__global__ __launch_bounds__(256) P3D_OCC(OCC) void wf_tile_kernel(const LaunchParams P) {
    const typename View<LDS>::type sv = View<LDS>::make_shading(P);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const TravCtx tc = wave_stack<LDS>(P, 0);
    TileCtx X;
    X.lds_off = View<LDS>::scene_dwords(P) + 4u * P.trav_stack_dwords;
    X.lq_off = LDS ? X.lds_off + (uint32_t)((sizeof(TileLds) + 15) / 16) * 4u : 0u;
    uint8_t* slot = P.tw_base + (size_t)blockIdx.x * P.tw_slot_bytes;
    X.rays = reinterpret_cast<RayRec*>(slot + P.tw_rays_off);
    X.nodes = reinterpret_cast<NodeRec*>(slot + P.tw_nodes_off);
    X.keys = STOCH ? reinterpret_cast<uint32_t*>(slot + P.tw_rng_off) : nullptr;
    TileLds* T = X.lds();
    Ctr ctr = {0, 0, 0, 0, 0, 0, 0};
    const int D = P.max_depth, ns = P.spp > 0 ? P.spp * P.spp : 1;
    uint32_t my_pixels = 0;
    int prev_tile = -1;
    unsigned long long t_tile = 0;
    for (;;) {
        if (tid == 0) T->tile = atomicAdd(&P.tw_ctrl[0], 1u);
        __syncthreads();
        if ((int)T->tile >= P.n_tiles) {                         // workgroup-uniform
            if (P.tile_cost && tid == 0 && prev_tile >= 0) P.tile_cost[prev_tile] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_tile);
            break;
        }
        const int tile = P.tile_order ? (int)P.tile_order[T->tile] : (int)T->tile;     // (heaviest first, once measured)
        if (tid == 0 && (P.tile_cost || (kStamps && P.dbg_stamps))) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (P.tile_cost && prev_tile >= 0) P.tile_cost[prev_tile] = (uint32_t)(now - t_tile);
            if (kStamps && P.dbg_stamps) {                       // diagnostic: one record per TILE (0 start, 1 end, 2 workgroup)
                if (prev_tile >= 0) P.dbg_stamps[(size_t)prev_tile * 8 + 1] = now;
                P.dbg_stamps[(size_t)tile * 8] = now;
                P.dbg_stamps[(size_t)tile * 8 + 2] = blockIdx.x;
            }
            t_tile = now;
        }
        prev_tile = tile;
        X.tx = tile % P.tiles_x; X.ty = tile / P.tiles_x;
        const int x = X.tx * 16 + (lane & 15);
        const int row = X.ty * 16 + wave * 4 + (lane >> 4);      // row in the compact local buffer
        const int y = image_row(P, row);
        const bool inside = x < P.res_x && y < P.res_y;
        const size_t p = (size_t)row * P.res_x + x;
        if (P.spp > 0) { T->acc[3 * tid] = 0.0f; T->acc[3 * tid + 1] = 0.0f; T->acc[3 * tid + 2] = 0.0f; }
        for (int smp = 0; smp < ns; smp++) {
            if (tid < 2 * kMaxTileLevels) T->n_rays[tid] = 0u;   // n_rays and n_nodes are contiguous
            __syncthreads();
            // ---- trace: level 1 = this sample's camera rays (one per thread), level l >= 2 = the queue the
            // level above filled, shared by the four waves.  ONE call site for both, so that the per-node code
            // is instantiated once per kernel.
            for (int l = 1; l <= D; l++) {
                const uint32_t n = l == 1 ? kTilePx : T->n_rays[l];
                // Whole rounds of 4 x 64 rays, then the remainder.  With the work-sharing walk (scenes read from HBM) the
                // remainder is SPREAD over the four waves -- 40 queued rays are 10 per wave with 54 helper lanes each, not
                // one wave of 40 beside three idle ones: the levels of a heavy tile are short queues of long walks.
                constexpr bool kSpread = WALK == WALK_SHARED && !LDS;
                const uint32_t full = (kSpread && l > 1) ? (n / 256u) * 256u : n, rem = n - full;     // (not spread: the plain loop)
                const uint32_t per = (rem + 3u) / 4u;
                for (uint32_t base = (uint32_t)wave * 64u; base < full || (base == full + (uint32_t)wave * 64u && rem > 0u); base += 256u) {
                    const bool tail = base >= full;
                    const uint32_t i = tail ? full + (uint32_t)wave * per + lane : base + lane;
                    const bool in_batch = tail ? ((uint32_t)lane < per && i < n) : true;
                    bool valid; uint32_t link = (uint32_t)tid, rng = 0; float ior_1 = 1.0f;
                    Ray ray; ray.o = mk(0.0f, 0.0f, 0.0f); ray.d = mk(1.0f, 0.0f, 0.0f);
                    if (l == 1) {
                        valid = inside;
                        if (inside) ray = camera_ray(P, x, y, smp);
                        if (STOCH) rng = rng_mix(rng_mix(P.seed, (uint32_t)(y * P.res_x + x)), (uint32_t)smp);
                    } else {
                        valid = in_batch && i < n;
                        if (valid) {
                            const float4* rq = X.ray_slot(l, i);
                            const float4 a = rq[0], b = rq[1];
                            ray.o = mk(a.x, a.y, a.z); ray.d = mk(b.x, b.y, b.z);
                            ior_1 = a.w; link = __float_as_uint(b.w);
                            if (STOCH) rng = X.keys[tile_ray_offset(l) + i];
                        }
                    }
                    const Hit h = find_closest<COUNT, WALK>(P, sv, ray, valid, tc, ctr);
                    if (l == 1 && smp == 0 && inside && P.hit_id) P.hit_id[p] = (h.ref == 0xFFFFFFFFu) ? -1 : (int32_t)h.sid;
                    const NodeOut o = shade_hit<COUNT, WALK, typename View<LDS>::type, STOCH>(P, sv, ray, h, valid, l, ior_1, tc,
                                                                                              ctr, rng, smp);
                    tile_emit(P, X, l, valid, link, ior_1, o);
                }
                __syncthreads();
            }
            for (int l = D - 1; l >= 1; l--) {                   // ---- resolve: RT/main.cpp:719, deepest level first
                const uint32_t n = T->n_nodes[l];
                for (uint32_t i = (uint32_t)tid; i < n; i += 256u) {
                    const float4* nd = reinterpret_cast<const float4*>(X.nodes + tile_node_offset(l) + i);
                    const float4 a = nd[0], b = nd[1], c = nd[2];
                    const Mtl M = load_material(sv, __float_as_uint(b.w));
                    const V3 ret = combine_node(mk(a.x, a.y, a.z), a.w, M.spec, mk(b.x, b.y, b.z), mk(c.x, c.y, c.z));
                    tile_deliver(P, X, l, __float_as_uint(c.w), ret);
                }
                __syncthreads();
            }
        }
        if (P.spp > 0 && inside) {                               // "color / (4 * 4)", RT/main.cpp:800
            const float* a = T->acc + 3 * tid;
            write_pixel(P, p, mk(fdiv(a[0], 16.0f), fdiv(a[1], 16.0f), fdiv(a[2], 16.0f)));
        }
        my_pixels += inside ? 1u : 0u;
    }
    if (kStamps && P.dbg_stamps && tid == 0 && prev_tile >= 0) P.dbg_stamps[(size_t)prev_tile * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    flush_counters<COUNT>(P, ctr, my_pixels);
    // the last workgroup out re-arms the tile counter for the next launch on this workspace (the frame
    // is self-contained on the device: safe to capture into a HIP graph and replay)
    if (tid == 0 && atomicAdd(&P.tw_ctrl[1], 1u) == gridDim.x - 1u) {
        atomicExch(&P.tw_ctrl[0], 0u);
        atomicExch(&P.tw_ctrl[1], 0u);
    }
}


// ------------------------------------------------------------------ TREE schedule
// shade-stack frame: 12 dwords.  STRIDE 64: in LDS, [field][lane] (scenes rendered from an LDS copy, and trees deeper
// than 8 levels).  STRIDE 1: in the lane's PRIVATE memory (scratch) -- scenes read from HBM are bound by how many waves
// are resident to hide fetch latency, frames are touched once per tree node, and 9 KB of LDS per wave for three of them
// held the dragon to 10 waves per CU: 1.81 -> 1.55 ms with the frames in scratch (16-25 waves per CU).
template <int STRIDE>
struct Frames {
    uint32_t* base;   // this lane's field-0 of frame-0; field stride STRIDE, frame stride 12 * STRIDE
    __device__ __forceinline__ uint32_t& f(int frame, int field) { return base[(frame * 12 + field) * STRIDE]; }
    __device__ __forceinline__ void put3(int frame, int field, V3 v) {
        f(frame, field) = __float_as_uint(v.x); f(frame, field + 1) = __float_as_uint(v.y);
        f(frame, field + 2) = __float_as_uint(v.z);
    }
    __device__ __forceinline__ V3 get3(int frame, int field) {
        return mk(__uint_as_float(f(frame, field)), __uint_as_float(f(frame, field + 1)),
                  __uint_as_float(f(frame, field + 2)));
    }
};
// frame fields.  While the reflection child runs, A/RD/IOR hold the parked refraction ray;
// afterwards A holds reflection_color * KR * specColor.
enum { FR_C = 0, FR_KR = 3, FR_META = 4, FR_A = 5, FR_RD = 8, FR_IOR = 11 };
#define FR_HAS_REFR 0x40000000u
#define FR_WAIT_REFR 0x80000000u

// One primary ray's whole tree: rayTracing(ray, 1, 1.0) of RT/main.cpp:530-721, iterative.
template <bool COUNT, bool GRID, class SV, class FR>
__device__ __forceinline__ V3 trace_tree(const LaunchParams& P, const SV& sv, Ray ray, const TravCtx& tc, FR fr,
                                         int32_t& primary_hit, Ctr& ctr) {
    int fsp = 0;              // frames on the stack == depth - 1
    float ior_1 = 1.0f;
    bool first = true;
    V3 ret = mk(0.0f, 0.0f, 0.0f);
    const V3 zero = mk(0.0f, 0.0f, 0.0f);
    for (;;) {
        Hit h = find_closest<COUNT, GRID ? WALK_GRID : WALK_LANE>(P, sv, ray, true, tc, ctr);
        if (first) { primary_hit = (h.ref == 0xFFFFFFFFu) ? -1 : (int32_t)h.sid; first = false; }
        NodeOut o = shade_hit<COUNT, GRID ? WALK_GRID : WALK_LANE>(P, sv, ray, h, true, fsp + 1, ior_1, tc, ctr);
        if (!o.terminal) {
            fr.put3(fsp, FR_C, o.color);
            fr.f(fsp, FR_KR) = __float_as_uint(o.KR);
            if (o.has_refl) {
                fr.f(fsp, FR_META) = o.mat | (o.has_refr ? FR_HAS_REFR : 0u);
                fr.put3(fsp, FR_A, o.refr.o);
                fr.put3(fsp, FR_RD, o.refr.d);
                fr.f(fsp, FR_IOR) = __float_as_uint(o.newIor);
                ray = o.refl;                                    // ior_1 unchanged
            } else {
                Mtl M = load_material(sv, o.mat);
                fr.f(fsp, FR_META) = o.mat | FR_WAIT_REFR;
                fr.put3(fsp, FR_A, cmul(mul(zero, o.KR), M.spec));
                ray = o.refr; ior_1 = o.newIor;
            }
            fsp++;
            continue;
        }
        ret = o.ret;
        // ---- return path: combine into parents (RT/main.cpp:719)
        bool resumed = false;
        while (fsp > 0) {
            int k = fsp - 1;
            uint32_t meta = fr.f(k, FR_META);
            V3 C = fr.get3(k, FR_C);
            float KR = __uint_as_float(fr.f(k, FR_KR));
            if (!(meta & FR_WAIT_REFR)) {
                Mtl M = load_material(sv, meta & 0x3FFFFFFFu);
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
                ret = add(C, add(A, mul(zero, 1.0f - KR)));
            } else {
                V3 A = fr.get3(k, FR_A);
                ret = add(C, add(A, mul(ret, 1.0f - KR)));
            }
            fsp--;
        }
        if (!resumed) return ret;
    }
}

// The same trees with the wave's lanes in ONE loop (work-sharing walk, WALK_SHARED): every iteration all 64 lanes reach
// find_closest() / shade_hit() together -- a lane whose pixel is finished (or outside the image) comes along as a helper
// of the others' walks -- and a lane that finishes a sample's tree starts its next sample at once.
template <bool COUNT, class SV, class FR>
__device__ __forceinline__ void trace_trees_shared(const LaunchParams& P, const SV& sv, int x, int y, bool valid, const TravCtx& tc, FR fr,
                                                   V3& color, int32_t& hid, Ctr& ctr) {
    const int ns = P.spp > 0 ? P.spp * P.spp : 1;
    const V3 zero = mk(0.0f, 0.0f, 0.0f);
    int smp = 0, fsp = 0;
    float ior_1 = 1.0f;
    bool alive = valid, first = true;
    V3 acc = zero;
    Ray ray; ray.o = zero; ray.d = mk(1.0f, 0.0f, 0.0f);
    if (alive) ray = camera_ray(P, x, y, 0);
    while (__ballot(alive) != 0) {
        const Hit h = find_closest<COUNT, WALK_SHARED>(P, sv, ray, alive, tc, ctr);
        if (alive && first) { if (smp == 0) hid = (h.ref == 0xFFFFFFFFu) ? -1 : (int32_t)h.sid; first = false; }
        const NodeOut o = shade_hit<COUNT, WALK_SHARED>(P, sv, ray, h, alive, fsp + 1, ior_1, tc, ctr);
        if (!alive) continue;
        if (!o.terminal) {
            fr.put3(fsp, FR_C, o.color);
            fr.f(fsp, FR_KR) = __float_as_uint(o.KR);
            if (o.has_refl) {
                fr.f(fsp, FR_META) = o.mat | (o.has_refr ? FR_HAS_REFR : 0u);
                fr.put3(fsp, FR_A, o.refr.o);
                fr.put3(fsp, FR_RD, o.refr.d);
                fr.f(fsp, FR_IOR) = __float_as_uint(o.newIor);
                ray = o.refl;                                    // ior_1 unchanged
            } else {
                Mtl M = load_material(sv, o.mat);
                fr.f(fsp, FR_META) = o.mat | FR_WAIT_REFR;
                fr.put3(fsp, FR_A, cmul(mul(zero, o.KR), M.spec));
                ray = o.refr; ior_1 = o.newIor;
            }
            fsp++;
            continue;
        }
        V3 ret = o.ret;
        bool resumed = false;                                    // ---- return path: combine into parents (RT/main.cpp:719)
        while (fsp > 0) {
            const int k = fsp - 1;
            const uint32_t meta = fr.f(k, FR_META);
            const V3 C = fr.get3(k, FR_C);
            const float KR = __uint_as_float(fr.f(k, FR_KR));
            if (!(meta & FR_WAIT_REFR)) {
                Mtl M = load_material(sv, meta & 0x3FFFFFFFu);
                const V3 A = cmul(mul(ret, KR), M.spec);
                if (meta & FR_HAS_REFR) {
                    ray.o = fr.get3(k, FR_A);
                    ray.d = fr.get3(k, FR_RD);
                    ior_1 = __uint_as_float(fr.f(k, FR_IOR));
                    fr.put3(k, FR_A, A);
                    fr.f(k, FR_META) = meta | FR_WAIT_REFR;
                    resumed = true;
                    break;
                }
                ret = add(C, add(A, mul(zero, 1.0f - KR)));
            } else {
                const V3 A = fr.get3(k, FR_A);
                ret = add(C, add(A, mul(ret, 1.0f - KR)));
            }
            fsp--;
        }
        if (resumed) continue;
        const V3 c = clampc(ret);                                // "rayTracing(...).clamp()" of this sample
        if (P.spp == 0) { color = c; alive = false; continue; }
        acc = add(acc, c);                                       // RT/main.cpp:797, in sample order
        smp++;
        if (smp < ns) { ray = camera_ray(P, x, y, smp); fsp = 0; ior_1 = 1.0f; }
        else { color = mk(fdiv(acc.x, 16.0f), fdiv(acc.y, 16.0f), fdiv(acc.z, 16.0f)); alive = false; }
    }
}

// PRIV = dwords of private memory for the frames (12 per level below the first), 0 = frames in LDS
template <bool COUNT, bool LDS, int OCC, bool GRID = false, int PRIV = 0, bool SHARED = false>
__global__ __launch_bounds__(LDS ? 256 : 64) P3D_OCC(OCC) void whitted_tree_kernel(const LaunchParams P) {
    const typename View<LDS>::type sv = View<LDS>::make_shading(P);
    const int lane = threadIdx.x & 63;
    int x, y, row, tile;
    const bool in_image = tile_pixel<!LDS>(P, x, y, row, &tile);
    if (SHARED ? __ballot(in_image) == 0 : !in_image) return;   // no barriers below: early exit is safe
    const unsigned long long t_tile = (!LDS && P.tile_cost) ? __builtin_amdgcn_s_memrealtime() : 0ull;
    stamp(P, tile, 0);
    uint32_t priv[PRIV > 0 ? PRIV : 1];
    uint32_t* wbase;
    const uint32_t frame_dwords = PRIV > 0 ? 0u : (uint32_t)(P.max_depth > 1 ? (P.max_depth - 1) : 1) * 12 * 64;
    const TravCtx st = wave_stack<LDS>(P, frame_dwords, &wbase);
    Frames<(PRIV > 0) ? 1 : 64> fr;
    if constexpr (PRIV > 0) fr.base = priv; else fr.base = wbase + P.trav_stack_dwords + lane;

    Ctr ctr = {0, 0, 0, 0, 0, 0, 0};
    V3 color = mk(0.0f, 0.0f, 0.0f);
    int32_t hid = -1;
    if constexpr (SHARED) {
        trace_trees_shared<COUNT>(P, sv, x, y, in_image, st, fr, color, hid, ctr);
        if (in_image) {
            const size_t p = (size_t)row * P.res_x + x;
            write_pixel(P, p, color);
            if (P.hit_id) P.hit_id[p] = hid;
            flush_counters<COUNT>(P, ctr, 1u);
        }
        stamp(P, tile, 4);
        if (!LDS && P.tile_cost && threadIdx.x == 0) P.tile_cost[tile] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_tile);
        return;
    }
    if (P.spp == 0) {                                    // RT/main.cpp:756-775
        color = clampc(trace_tree<COUNT, GRID>(P, sv, camera_ray(P, x, y, 0), st, fr, hid, ctr));
    } else {                                             // RT/main.cpp:776-801
        const int ns = P.spp * P.spp;
        for (int s = 0; s < ns; s++) {
            int32_t h2 = -1;
            V3 c = clampc(trace_tree<COUNT, GRID>(P, sv, camera_ray(P, x, y, s), st, fr, h2, ctr));
            color = add(color, c);
            if (s == 0) hid = h2;
        }
        color = mk(fdiv(color.x, 16.0f), fdiv(color.y, 16.0f), fdiv(color.z, 16.0f));
    }
    const size_t p = (size_t)row * P.res_x + x;
    write_pixel(P, p, color);
    if (P.hit_id) P.hit_id[p] = hid;
    flush_counters<COUNT>(P, ctr, 1u);
    stamp(P, tile, 4);              // (diagnostic; the wave has reconverged here: its slowest lane is done)
    if (!LDS && P.tile_cost && threadIdx.x == 0) P.tile_cost[tile] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - t_tile);
}

// per-column / per-row factors of the pixel-centre camera rays (one launch per resolution)
__global__ void raygen_table_kernel(float* fx, float* fy, int res_x, int res_y) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < res_x) fx[i] = fdiv((float)i + 0.5f, (float)res_x) - 0.5f;      // pixel.x / res_x - 0.5f
    if (i < res_y) fy[i] = fdiv((float)i + 0.5f, (float)res_y) - 0.5f;
}

// ------------------------------------------------------------------ rank-0 de-interleave
// T = uint4 when rows, strides and pointers are 16-byte multiples (1920 x 3 B rows are), else uint8_t.
// blockIdx.y = frame of a batch (frame f of rank r starts at r * rank_stride + f * in_stride).
template <class T>
__global__ void deinterleave_kernel(const uint8_t* __restrict__ gathered, uint8_t* __restrict__ frames,
                                    size_t row_bytes, int res_y, int row_block, int world, size_t rank_stride,
                                    size_t in_stride, size_t out_stride) {
    const size_t row_units = row_bytes / sizeof(T);
    const size_t total = row_units * res_y;
    const uint8_t* src = gathered + (size_t)blockIdx.y * in_stride;
    uint8_t* dst = frames + (size_t)blockIdx.y * out_stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        size_t y = i / row_units, off = (i - y * row_units) * sizeof(T);
        int blk = (int)(y / row_block);
        int rank = blk % world, lblk = blk / world;
        size_t lrow = (size_t)lblk * row_block + (y - (size_t)blk * row_block);
        *reinterpret_cast<T*>(dst + y * row_bytes + off) =
            *reinterpret_cast<const T*>(src + (size_t)rank * rank_stride + lrow * row_bytes + off);
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
        if (h) {     // the host-side statement of this lives in scene_flatten.cpp; the probe keeps the device arithmetic
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
static size_t scene_lds_bytes(const LaunchParams& P, bool lds) { return lds ? (size_t)P.blob_quads * 16 : 0; }
// frames of the tree kernel: private memory for scenes read from HBM up to depth 8, LDS otherwise
static int tree_private_dwords(const LaunchParams& P, bool lds) {
    if (lds || P.accel == 1) return 0;
    return P.max_depth <= 4 ? 36 : (P.max_depth <= 8 ? 84 : 0);
}
size_t tree_kernel_lds_bytes(const LaunchParams& P, bool lds) {
    if (tree_private_dwords(P, lds)) return scene_lds_bytes(P, lds) + (size_t)P.trav_stack_dwords * 4 * P.wg_waves;
    int frames = P.max_depth > 1 ? (P.max_depth - 1) : 1;
    size_t wave_dwords = (size_t)P.trav_stack_dwords + (size_t)frames * 12 * 64;
    return scene_lds_bytes(P, lds) + wave_dwords * 4 * P.wg_waves;
}
size_t wavefront_lds_bytes(const LaunchParams& P, bool lds) {
    return scene_lds_bytes(P, lds) + (size_t)P.trav_stack_dwords * 4 * P.wg_waves;
}

// Kernel variants are picked through function pointers: {count} x {scene in LDS} x {walk: lane / packet / grid}
// x {register budget: only for the timed non-grid builds} x {random draws}.  Counting, grid and stochastic
// builds always use the default register budget.
template <class F> static const void* fn_ptr(F f) { return reinterpret_cast<const void*>(f); }
#define P3D_DEFINE_SELECTOR(NAME, KERNEL)                                                                         \
    static const void* NAME(bool count, bool lds, int walk, int occ, bool stoch) {                                \
        if (walk == WALK_GRID || count || stoch) occ = 1;                                                         \
        if (walk == WALK_SHARED && !lds) {                                                                        \
            if (stoch) return count ? fn_ptr(KERNEL<true, false, 3, 1, true>) : fn_ptr(KERNEL<false, false, 3, 1, true>);                                                                  \
            if (count) return fn_ptr(KERNEL<true, false, 3, 1, false>);                                           \
            return occ == 5 ? fn_ptr(KERNEL<false, false, 3, 5, false>) : occ == 6 ? fn_ptr(KERNEL<false, false, 3, 6, false>) : fn_ptr(KERNEL<false, false, 3, 1, false>);               \
        }                                                                                                         \
        if (walk == WALK_SHARED) walk = WALK_LANE;                                                                \
        if (stoch) {                                                                                              \
            if (count) return lds ? (walk == 2 ? fn_ptr(KERNEL<true, true, 2, 1, true>) : walk == 1 ? fn_ptr(KERNEL<true, true, 1, 1, true>) : fn_ptr(KERNEL<true, true, 0, 1, true>))       \
                                  : (walk == 2 ? fn_ptr(KERNEL<true, false, 2, 1, true>) : walk == 1 ? fn_ptr(KERNEL<true, false, 1, 1, true>) : fn_ptr(KERNEL<true, false, 0, 1, true>));   \
            return lds ? (walk == 2 ? fn_ptr(KERNEL<false, true, 2, 1, true>) : walk == 1 ? fn_ptr(KERNEL<false, true, 1, 1, true>) : fn_ptr(KERNEL<false, true, 0, 1, true>))              \
                       : (walk == 2 ? fn_ptr(KERNEL<false, false, 2, 1, true>) : walk == 1 ? fn_ptr(KERNEL<false, false, 1, 1, true>) : fn_ptr(KERNEL<false, false, 0, 1, true>));          \
        }                                                                                                         \
        if (count) return lds ? (walk == 2 ? fn_ptr(KERNEL<true, true, 2, 1, false>) : walk == 1 ? fn_ptr(KERNEL<true, true, 1, 1, false>) : fn_ptr(KERNEL<true, true, 0, 1, false>))      \
                              : (walk == 2 ? fn_ptr(KERNEL<true, false, 2, 1, false>) : walk == 1 ? fn_ptr(KERNEL<true, false, 1, 1, false>) : fn_ptr(KERNEL<true, false, 0, 1, false>));  \
        if (walk == 2) return lds ? fn_ptr(KERNEL<false, true, 2, 1, false>) : fn_ptr(KERNEL<false, false, 2, 1, false>);                                                                  \
        if (occ == 5) return lds ? (walk == 1 ? fn_ptr(KERNEL<false, true, 1, 5, false>) : fn_ptr(KERNEL<false, true, 0, 5, false>))                                                        \
                                 : (walk == 1 ? fn_ptr(KERNEL<false, false, 1, 5, false>) : fn_ptr(KERNEL<false, false, 0, 5, false>));                                                     \
        if (occ == 6) return lds ? (walk == 1 ? fn_ptr(KERNEL<false, true, 1, 6, false>) : fn_ptr(KERNEL<false, true, 0, 6, false>))                                                        \
                                 : (walk == 1 ? fn_ptr(KERNEL<false, false, 1, 6, false>) : fn_ptr(KERNEL<false, false, 0, 6, false>));                                                     \
        return lds ? (walk == 1 ? fn_ptr(KERNEL<false, true, 1, 1, false>) : fn_ptr(KERNEL<false, true, 0, 1, false>))                                                                      \
                   : (walk == 1 ? fn_ptr(KERNEL<false, false, 1, 1, false>) : fn_ptr(KERNEL<false, false, 0, 1, false>));                                                                   \
    }
P3D_DEFINE_SELECTOR(wf_primary_fn, wf_primary_kernel)
P3D_DEFINE_SELECTOR(wf_secondary_fn, wf_secondary_kernel)
P3D_DEFINE_SELECTOR(wf_tile_fn, wf_tile_kernel)
#undef P3D_DEFINE_SELECTOR

static const void* tree_fn(bool count, bool lds, int occ, bool grid, int priv = 0, bool shared = false) {
    if (shared && !lds && !grid && priv == 36) {
        if (count) return fn_ptr(whitted_tree_kernel<true, false, 1, false, 36, true>);
        return occ == 5 ? fn_ptr(whitted_tree_kernel<false, false, 5, false, 36, true>) : occ == 6 ? fn_ptr(whitted_tree_kernel<false, false, 6, false, 36, true>)
                                                                                                 : fn_ptr(whitted_tree_kernel<false, false, 1, false, 36, true>);
    }
    if (shared && !lds && !grid && priv == 84) {
        if (count) return fn_ptr(whitted_tree_kernel<true, false, 1, false, 84, true>);
        return occ == 5 ? fn_ptr(whitted_tree_kernel<false, false, 5, false, 84, true>) : occ == 6 ? fn_ptr(whitted_tree_kernel<false, false, 6, false, 84, true>)
                                                                                                 : fn_ptr(whitted_tree_kernel<false, false, 1, false, 84, true>);
    }
    if (shared && !lds && !grid && priv == 0) {
        if (count) return fn_ptr(whitted_tree_kernel<true, false, 1, false, 0, true>);
        return fn_ptr(whitted_tree_kernel<false, false, 1, false, 0, true>);
    }
    if (grid) return count ? (lds ? fn_ptr(whitted_tree_kernel<true, true, 1, true>) : fn_ptr(whitted_tree_kernel<true, false, 1, true>))
                           : (lds ? fn_ptr(whitted_tree_kernel<false, true, 1, true>) : fn_ptr(whitted_tree_kernel<false, false, 1, true>));
    if (!lds && priv == 36) {
        if (count) return fn_ptr(whitted_tree_kernel<true, false, 1, false, 36>);
        return occ == 5 ? fn_ptr(whitted_tree_kernel<false, false, 5, false, 36>) : occ == 6 ? fn_ptr(whitted_tree_kernel<false, false, 6, false, 36>)
                                                                                           : fn_ptr(whitted_tree_kernel<false, false, 1, false, 36>);
    }
    if (!lds && priv == 84) {
        if (count) return fn_ptr(whitted_tree_kernel<true, false, 1, false, 84>);
        return occ == 5 ? fn_ptr(whitted_tree_kernel<false, false, 5, false, 84>) : occ == 6 ? fn_ptr(whitted_tree_kernel<false, false, 6, false, 84>)
                                                                                           : fn_ptr(whitted_tree_kernel<false, false, 1, false, 84>);
    }
    if (count) return lds ? fn_ptr(whitted_tree_kernel<true, true, 1>) : fn_ptr(whitted_tree_kernel<true, false, 1>);
    if (occ == 5) return lds ? fn_ptr(whitted_tree_kernel<false, true, 5>) : fn_ptr(whitted_tree_kernel<false, false, 5>);
    if (occ == 6) return lds ? fn_ptr(whitted_tree_kernel<false, true, 6>) : fn_ptr(whitted_tree_kernel<false, false, 6>);
    return lds ? fn_ptr(whitted_tree_kernel<false, true, 1>) : fn_ptr(whitted_tree_kernel<false, false, 1>);
}
static hipError_t launch_by_pointer(const void* fn, const LaunchParams& P, dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    LaunchParams Pc = P;
    void* args[] = {&Pc};
    return hipLaunchKernel(fn, grid, block, args, shmem, stream);
}

hipError_t launch_tree(const LaunchParams& P, bool count, bool lds, int occ, bool shared, hipStream_t stream) {
    return launch_by_pointer(tree_fn(count, lds, occ, P.accel == 1, tree_private_dwords(P, lds), shared), P, dim3((unsigned)P.grid_blocks),
                             dim3(64 * P.wg_waves), tree_kernel_lds_bytes(P, lds), stream);
}
hipError_t launch_wf_primary(const LaunchParams& P, bool count, bool lds, int walk, int occ, hipStream_t stream) {
    // identity tile map and no learned order: a 2-D grid, blockIdx = (tile column, tile row) -- see tile_pixel()
    // (LDS scenes only: the kernels of scenes read from HBM number their tiles through the learned order)
    const bool grid2d = lds && P.xcd_chunk == 1 && P.wf_tile_rows > 1 && P.tiles_x * P.wf_tile_rows == P.n_tiles;
    const dim3 grid = grid2d ? dim3((unsigned)P.tiles_x, (unsigned)P.wf_tile_rows) : dim3((unsigned)P.grid_blocks);
    return launch_by_pointer(wf_primary_fn(count, lds, walk, occ, P.features != 0), P, grid,
                             dim3(64 * P.wg_waves), wavefront_lds_bytes(P, lds), stream);
}
hipError_t launch_wf_secondary(const LaunchParams& P, bool count, bool lds, int walk, int occ, unsigned waves,
                               hipStream_t stream) {
    return launch_by_pointer(wf_secondary_fn(count, lds, walk, occ, P.features != 0), P, dim3((waves + P.wg_waves - 1) / P.wg_waves),
                             dim3(64 * P.wg_waves), wavefront_lds_bytes(P, lds), stream);
}
// waves of the deeper-level kernel that can be resident on the device at once (LDS-scene variants: 256-thread workgroups)
hipError_t wf_resident_waves(const LaunchParams& P, bool primary, bool count, bool lds, int walk, int occ, unsigned* waves) {
    const void* fn = primary ? wf_primary_fn(count, lds, walk, occ, P.features != 0) : wf_secondary_fn(count, lds, walk, occ, P.features != 0);
    int per_cu = 0, dev = 0, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * P.wg_waves, wavefront_lds_bytes(P, lds));
    if (e != hipSuccess) return e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    *waves = (unsigned)((per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1) * P.wg_waves);
    return hipSuccess;
}
// tile schedule: 256-thread workgroups
size_t tile_kernel_lds_bytes(const LaunchParams& P, bool lds) {
    return scene_lds_bytes(P, lds) + (size_t)P.trav_stack_dwords * 4 * 4 + (sizeof(TileLds) + 15) / 16 * 16 + (lds ? (size_t)kTileLdsRayDwords * 4 : 0);
}
// workgroups of this variant that can be resident on the whole device (persistent grid size)
hipError_t tile_kernel_resident_blocks(const LaunchParams& P, bool count, bool lds, int walk, int occ, int* blocks) {
    const void* fn = wf_tile_fn(count, lds, walk, occ, P.features != 0);
    const size_t shmem = tile_kernel_lds_bytes(P, lds);
    if (shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    int per_cu = 0, dev = 0, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, shmem);
    if (e != hipSuccess) return e;
    if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    *blocks = (per_cu > 0 ? per_cu : 1) * (cus > 0 ? cus : 1);
    return hipSuccess;
}
hipError_t launch_wf_tile(const LaunchParams& P, bool count, bool lds, int walk, int occ, unsigned blocks, hipStream_t stream) {
    return launch_by_pointer(wf_tile_fn(count, lds, walk, occ, P.features != 0), P, dim3(blocks), dim3(256),
                             tile_kernel_lds_bytes(P, lds), stream);
}


hipError_t launch_wf_resolve_fused(const LaunchParams& P, const ResolveLevels& R, unsigned shards, hipStream_t stream) {
    hipLaunchKernelGGL(wf_resolve_fused_kernel, dim3(shards), dim3(1024), 0, stream, P, R);
    return hipGetLastError();
}
hipError_t launch_wf_resolve(const LaunchParams& P, unsigned blocks, hipStream_t stream) {
    hipLaunchKernelGGL(wf_resolve_kernel, dim3(blocks), dim3(256), 0, stream, P);
    return hipGetLastError();
}

__global__ void clear_words_kernel(uint32_t* p, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0u;
}
hipError_t launch_clear_words(uint32_t* p, uint32_t n, hipStream_t stream) {
    hipLaunchKernelGGL(clear_words_kernel, dim3((n + 255) / 256 < 64 ? (n + 255) / 256 : 64), dim3(256), 0, stream, p, n);
    return hipGetLastError();
}

hipError_t launch_sum_samples(const LaunchParams& P, size_t first_px, size_t n_px, hipStream_t stream) {
    const unsigned blocks = (unsigned)std::min<size_t>((n_px + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(sum_samples_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, stream, P, first_px, n_px);
    return hipGetLastError();
}

hipError_t prepare_kernels(size_t max_lds) {
    // only the tree kernel without an LDS scene copy can need more than the 64 KiB default
    const void* fns[] = {tree_fn(true, false, 1, false), tree_fn(false, false, 1, false), tree_fn(false, false, 5, false),
                         tree_fn(false, false, 6, false), tree_fn(true, false, 1, true), tree_fn(false, false, 1, true),
                         tree_fn(true, false, 1, false, 0, true), tree_fn(false, false, 1, false, 0, true)};
    for (const void* f : fns) {
        hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_raygen_table(float* fx, float* fy, int res_x, int res_y, hipStream_t stream) {
    int n = res_x > res_y ? res_x : res_y;
    hipLaunchKernelGGL(raygen_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, fx, fy, res_x, res_y);
    return hipGetLastError();
}
hipError_t launch_deinterleave(const void* gathered, void* frames, int res_x, int res_y, int row_block,
                               int world, size_t rank_stride, int bpp, int n_frames, size_t in_stride,
                               size_t out_stride, hipStream_t stream) {
    const size_t row_bytes = (size_t)res_x * bpp;
    const bool wide = row_bytes % 16 == 0 && rank_stride % 16 == 0 && in_stride % 16 == 0 && out_stride % 16 == 0 &&
                      (uintptr_t)gathered % 16 == 0 && (uintptr_t)frames % 16 == 0;
    const size_t units = (wide ? row_bytes / 16 : row_bytes) * (size_t)res_y;
    const dim3 grid((unsigned)std::min<size_t>((units + 255) / 256, 2048), (unsigned)n_frames);
    if (wide)
        hipLaunchKernelGGL(deinterleave_kernel<uint4>, grid, dim3(256), 0, stream, (const uint8_t*)gathered, (uint8_t*)frames,
                           row_bytes, res_y, row_block, world, rank_stride, in_stride, out_stride);
    else
        hipLaunchKernelGGL(deinterleave_kernel<uint8_t>, grid, dim3(256), 0, stream, (const uint8_t*)gathered, (uint8_t*)frames,
                           row_bytes, res_y, row_block, world, rank_stride, in_stride, out_stride);
    return hipGetLastError();
}

// every bit pattern in [first, first + count): frcp(x) against the division it replaces (NaNs compare equal to NaNs)
__global__ void debug_check_rcp_kernel(uint32_t first, uint64_t count, unsigned long long* n_bad, uint32_t* first_bad) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = first + (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float a = frcp(x), b = 1.0f / x;
        const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
        if (!same) { atomicAdd(n_bad, 1ull); atomicMin(first_bad, bits); }
    }
}
bool kernels_have_stamps() { return kStamps; }
hipError_t launch_debug_check_rcp(uint32_t first, uint64_t count, unsigned long long* n_bad, uint32_t* first_bad, hipStream_t stream) {
    hipLaunchKernelGGL(debug_check_rcp_kernel, dim3(256 * 16), dim3(256), 0, stream, first, count, n_bad, first_bad);
    return hipGetLastError();
}

__global__ void debug_powf_kernel(uint32_t n, const float* x, const float* y, float* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = p3d_powf(x[i], y[i]);
}
hipError_t launch_debug_powf(uint32_t n, const float* x, const float* y, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(debug_powf_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, x, y, out);
    return hipGetLastError();
}

hipError_t launch_debug_intersect(uint32_t n, const uint32_t* type, const float* prim12, const float* origin,
                                  const float* dir, int32_t* hit, float* t, float* normal, hipStream_t stream) {
    hipLaunchKernelGGL(debug_intersect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, type, prim12,
                       origin, dir, hit, t, normal);
    return hipGetLastError();
}

}  // namespace p3d
