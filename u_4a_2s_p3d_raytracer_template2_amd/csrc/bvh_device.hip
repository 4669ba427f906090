// bvh_device.hip -- BVH construction on the GPU: a linear BVH over
// Morton-sorted primitives, written straight into the scene blob in the NodePair / leaf-reference
// format the traversal kernels walk.  Replaces BVH::Build (RT/bvh.cpp:28-158) for the "load another
// scene" loop (RT/main.cpp:963-976) and for scenes whose host SAH build takes seconds (1e6 primitives:
// 1.6 s on the host).  Like the host builder's tree it only has to be CONSERVATIVE: the reference
// discards its own BVH's closest hit, so any tree over the padded primitive bounds gives
// the same image.  Its quality is below binned SAH (measured: 1e6 random primitives build in 0.08 s
// instead of 1.5 s incl. flattening and upload, and trace at 6.9 ms instead of 4.3 ms per 1080p frame);
// p3d_build_opts::builder selects it.
//
// Steps (one launch each, all on the caller's stream):
//   1. bounds of the primitive centroids                         (float min/max as ordered ints)
//   2. 63-bit Morton key per primitive, 21 bits per axis
//   3. radix sort of (key, primitive) pairs                      (hipcub::DeviceRadixSort)
//   4. leaves = runs of kLeafPrims consecutive sorted primitives: leaf reference list, leaf boxes
//   5. Karras' parallel hierarchy over the leaf keys              (HPG 2012; ties broken by index)
//   6. bottom-up box refit with one atomic counter per inner node (second arriver continues)
//   7. NodePair records (a node holds its CHILDREN's boxes), tree depth, SAH cost
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "bvh_builder.h"
#include "p3d_device_types.h"

namespace p3d {

namespace {

constexpr uint32_t kLeafPrims = 2;

struct Box6 { float lo[3], hi[3]; };

__device__ __forceinline__ uint32_t ordered(float f) {          // monotone float -> uint map
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unordered(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__global__ void lbvh_bounds_kernel(const BuildPrim* prims, uint32_t n, uint32_t* bounds /* lo3, hi3 as ordered */) {
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const BuildPrim p = prims[i];
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * (p.lo[a] + p.hi[a]);
            lo[a] = fminf(lo[a], c); hi[a] = fmaxf(hi[a], c);
        }
    }
    for (int a = 0; a < 3; a++) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; a++) {
            atomicMin(bounds + a, ordered(lo[a]));
            atomicMax(bounds + 3 + a, ordered(hi[a]));
        }
    }
}

__device__ __forceinline__ uint64_t spread21(uint32_t v) {      // 21 bits -> every third bit
    uint64_t x = v & 0x1FFFFFu;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}

__global__ void lbvh_morton_kernel(const BuildPrim* prims, uint32_t n, const uint32_t* bounds, uint64_t* keys,
                                   uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BuildPrim p = prims[i];
    uint64_t key = 0;
    for (int a = 0; a < 3; a++) {
        const float lo = unordered(bounds[a]), hi = unordered(bounds[3 + a]);
        const float ext = fmaxf(hi - lo, 1e-30f);
        const float c = 0.5f * (p.lo[a] + p.hi[a]);
        float f = (c - lo) / ext * 2097152.0f;
        f = fminf(fmaxf(f, 0.0f), 2097151.0f);
        key |= spread21((uint32_t)f) << (2 - a);
    }
    keys[i] = key; vals[i] = i;
}

__global__ void lbvh_leaves_kernel(const BuildPrim* prims, uint32_t n, uint32_t n_leaves, const uint64_t* keys,
                                   const uint32_t* order, uint32_t* refs, uint64_t* leaf_keys, Box6* leaf_box) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    const uint32_t first = j * kLeafPrims, last = min(first + kLeafPrims, n);
    Box6 b;
    for (int a = 0; a < 3; a++) { b.lo[a] = 3.4e38f; b.hi[a] = -3.4e38f; }
    for (uint32_t k = first; k < last; k++) {
        const BuildPrim p = prims[order[k]];
        refs[k] = p.ref;
        for (int a = 0; a < 3; a++) { b.lo[a] = fminf(b.lo[a], p.lo[a]); b.hi[a] = fmaxf(b.hi[a], p.hi[a]); }
    }
    leaf_keys[j] = keys[first];
    leaf_box[j] = b;
}

// common-prefix length of leaf keys i and j (ties broken by the index, so all keys are distinct)
__device__ __forceinline__ int delta(const uint64_t* k, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t x = k[i] ^ k[j];
    if (x == 0) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)x);
}

// children: >= 0 inner node, < 0 leaf ~index.  parent_* give each node its parent inner node.
__global__ void lbvh_hierarchy_kernel(const uint64_t* k, int n_leaves, int2* children, int* parent_inner, int* parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves - 1) return;
    const int d = (delta(k, n_leaves, i, i + 1) - delta(k, n_leaves, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(k, n_leaves, i, i - d);
    int lmax = 2;
    while (delta(k, n_leaves, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(k, n_leaves, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(k, n_leaves, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (delta(k, n_leaves, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int lo = min(i, j), hi = max(i, j);
    int2 c;
    if (lo == gamma) { c.x = ~gamma; parent_leaf[gamma] = i; } else { c.x = gamma; parent_inner[gamma] = i; }
    if (hi == gamma + 1) { c.y = ~(gamma + 1); parent_leaf[gamma + 1] = i; } else { c.y = gamma + 1; parent_inner[gamma + 1] = i; }
    children[i] = c;
    if (i == 0) parent_inner[0] = -1;
}

__device__ __forceinline__ Box6 child_box(int c, const Box6* leaf_box, const Box6* node_box) {
    return c < 0 ? leaf_box[~c] : node_box[c];
}
// the same while the refit is running: an inner node's box was written by another workgroup a moment
// ago, so it is read past this CU's L1 (a cached line may predate the write)
__device__ __forceinline__ Box6 child_box_fresh(int c, const Box6* leaf_box, const Box6* node_box) {
    if (c < 0) return leaf_box[~c];
    Box6 b;
    const float* src = reinterpret_cast<const float*>(node_box + c);
    for (int x = 0; x < 3; x++) {
        b.lo[x] = __hip_atomic_load(src + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        b.hi[x] = __hip_atomic_load(src + 3 + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return b;
}

__global__ void lbvh_refit_kernel(int n_leaves, const int2* children, const int* parent_inner, const int* parent_leaf,
                                  const Box6* leaf_box, Box6* node_box, uint32_t* arrived, uint32_t* max_depth) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_leaves) return;
    int node = parent_leaf[j];
    uint32_t depth = 1;                      // levels of inner nodes above this leaf
    // boxes: the second thread to reach a node finds both children complete
    bool fitting = true;
    while (node >= 0) {
        if (fitting) {
            __threadfence();
            if (atomicAdd(arrived + node, 1u) == 0u) fitting = false;
            else {
                const int2 c = children[node];
                const Box6 a = child_box_fresh(c.x, leaf_box, node_box), b = child_box_fresh(c.y, leaf_box, node_box);
                Box6 u;
                for (int x = 0; x < 3; x++) { u.lo[x] = fminf(a.lo[x], b.lo[x]); u.hi[x] = fmaxf(a.hi[x], b.hi[x]); }
                node_box[node] = u;
            }
        }
        node = parent_inner[node];
        if (node >= 0) depth++;
    }
    atomicMax(max_depth, depth);
}

__device__ __forceinline__ float half_area(const Box6& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

__global__ void lbvh_emit_kernel(int n_leaves, uint32_t n_prims, const int2* children, const Box6* leaf_box,
                                 const Box6* node_box, NodePair* nodes, float cost_traverse, float cost_intersect,
                                 float* sah) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float cost = 0.0f;
    if (i < n_leaves - 1) {
        const int2 c = children[i];
        const Box6 a = child_box(c.x, leaf_box, node_box), b = child_box(c.y, leaf_box, node_box);
        auto leaf_code = [&](int leaf) {
            const uint32_t first = (uint32_t)leaf * kLeafPrims;
            const uint32_t cnt = min(kLeafPrims, n_prims - first);
            return (int32_t)~((first << 3) | (cnt - 1u));
        };
        NodePair nd;
        nd.lo0[0] = a.lo[0]; nd.lo0[1] = a.lo[1]; nd.lo0[2] = a.lo[2]; nd.hi0x = a.hi[0];
        nd.hi0yz[0] = a.hi[1]; nd.hi0yz[1] = a.hi[2]; nd.lo1xy[0] = b.lo[0]; nd.lo1xy[1] = b.lo[1];
        nd.lo1z = b.lo[2]; nd.hi1[0] = b.hi[0]; nd.hi1[1] = b.hi[1]; nd.hi1[2] = b.hi[2];
        nd.child0 = c.x < 0 ? leaf_code(~c.x) : c.x;
        nd.child1 = c.y < 0 ? leaf_code(~c.y) : c.y;
        nd.pad0 = 0; nd.pad1 = 0;
        nodes[i] = nd;
        // SAH cost in the host builder's units: node visits + primitive tests, area-weighted
        const float root = fmaxf(half_area(node_box[0]), 1e-30f);
        cost = cost_traverse * half_area(node_box[i]) / root;
        if (c.x < 0) cost += cost_intersect * (float)((~nd.child0 & 7) + 1) * half_area(a) / root;
        if (c.y < 0) cost += cost_intersect * (float)((~nd.child1 & 7) + 1) * half_area(b) / root;
    }
    for (int off = 32; off > 0; off >>= 1) cost += __shfl_xor(cost, off);
    if ((threadIdx.x & 63) == 0 && cost != 0.0f) atomicAdd(sah, cost);
}

}  // namespace

// Builds the tree of `prims` (host array, padded bounds) into device memory: nodes[n_leaves - 1],
// refs[n].  n >= 2 * kLeafPrims.  Synchronous (returns when the tree is complete).
hipError_t build_lbvh_device(const std::vector<BuildPrim>& prims, const BvhOptions& opt, NodePair* d_nodes,
                             uint32_t* d_refs, BvhStats& stats, hipStream_t stream) {
    const uint32_t n = (uint32_t)prims.size();
    const uint32_t L = (n + kLeafPrims - 1) / kLeafPrims;
    hipError_t e;
    char* pool = nullptr;
    // one scratch allocation, carved up
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; };
    const size_t o_prims = carve((size_t)n * sizeof(BuildPrim));
    const size_t o_keys = carve((size_t)n * 8), o_keys2 = carve((size_t)n * 8);
    const size_t o_vals = carve((size_t)n * 4), o_vals2 = carve((size_t)n * 4);
    const size_t o_lkeys = carve((size_t)L * 8), o_lbox = carve((size_t)L * sizeof(Box6)), o_nbox = carve((size_t)L * sizeof(Box6));
    const size_t o_child = carve((size_t)L * 8), o_pin = carve((size_t)L * 4), o_pleaf = carve((size_t)L * 4);
    const size_t o_arr = carve((size_t)L * 4), o_misc = carve(64);
    size_t sort_bytes = 0;
    e = hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr,
                                           (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)n, 0, 63, stream);
    if (e != hipSuccess) return e;
    const size_t o_sort = carve(sort_bytes);
    if ((e = hipMalloc((void**)&pool, off)) != hipSuccess) return e;
    auto done = [&](hipError_t rc) { (void)hipFree(pool); return rc; };
    BuildPrim* d_prims = (BuildPrim*)(pool + o_prims);
    uint64_t *keys = (uint64_t*)(pool + o_keys), *keys2 = (uint64_t*)(pool + o_keys2), *lkeys = (uint64_t*)(pool + o_lkeys);
    uint32_t *vals = (uint32_t*)(pool + o_vals), *vals2 = (uint32_t*)(pool + o_vals2);
    Box6 *lbox = (Box6*)(pool + o_lbox), *nbox = (Box6*)(pool + o_nbox);
    int2* child = (int2*)(pool + o_child);
    int *pin = (int*)(pool + o_pin), *pleaf = (int*)(pool + o_pleaf);
    uint32_t *arrived = (uint32_t*)(pool + o_arr), *misc = (uint32_t*)(pool + o_misc);   // misc: bounds[6], depth, sah
    if ((e = hipMemcpyAsync(d_prims, prims.data(), (size_t)n * sizeof(BuildPrim), hipMemcpyHostToDevice, stream)) != hipSuccess) return done(e);
    const uint32_t init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    if ((e = hipMemcpyAsync(misc, init, sizeof init, hipMemcpyHostToDevice, stream)) != hipSuccess) return done(e);
    if ((e = hipMemsetAsync(arrived, 0, (size_t)L * 4, stream)) != hipSuccess) return done(e);
    const unsigned T = 256;
    hipLaunchKernelGGL(lbvh_bounds_kernel, dim3(std::min<unsigned>((n + T - 1) / T, 2048)), dim3(T), 0, stream, d_prims, n, misc);
    hipLaunchKernelGGL(lbvh_morton_kernel, dim3((n + T - 1) / T), dim3(T), 0, stream, d_prims, n, misc, keys, vals);
    e = hipcub::DeviceRadixSort::SortPairs(pool + o_sort, sort_bytes, keys, keys2, vals, vals2, (int)n, 0, 63, stream);
    if (e != hipSuccess) return done(e);
    hipLaunchKernelGGL(lbvh_leaves_kernel, dim3((L + T - 1) / T), dim3(T), 0, stream, d_prims, n, L, keys2, vals2, d_refs, lkeys, lbox);
    hipLaunchKernelGGL(lbvh_hierarchy_kernel, dim3((L + T - 1) / T), dim3(T), 0, stream, lkeys, (int)L, child, pin, pleaf);
    hipLaunchKernelGGL(lbvh_refit_kernel, dim3((L + T - 1) / T), dim3(T), 0, stream, (int)L, child, pin, pleaf, lbox, nbox, arrived, misc + 6);
    hipLaunchKernelGGL(lbvh_emit_kernel, dim3((L + T - 1) / T), dim3(T), 0, stream, (int)L, n, child, lbox, nbox, d_nodes,
                       opt.cost_traverse, opt.cost_intersect, (float*)(misc + 7));
    if ((e = hipGetLastError()) != hipSuccess) return done(e);
    uint32_t back[8];
    if ((e = hipMemcpyAsync(back, misc, sizeof back, hipMemcpyDeviceToHost, stream)) != hipSuccess) return done(e);
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return done(e);
    stats = BvhStats();
    stats.n_nodes = L - 1; stats.n_leaves = L; stats.n_leaf_refs = n;
    stats.max_depth = back[6] + 1;            // + the leaf level, like the host builder's count
    memcpy(&stats.sah_cost, &back[7], 4);
    return done(hipSuccess);
}

// ------------------------------------------------------------------ tile order of the persistent tile schedule
// "Heaviest first": tiles sorted by the duration the previous frames measured for them (p3d_kernels.hip: wf_tile_kernel
// writes LaunchParams::tile_cost).  Persistent workgroups that draw tiles in that order finish together; drawn row by row,
// the expensive tiles of a frame are started late and the launch ends with a few workgroups still on them (dragon: 1.29 ms
// against 0.76 for the same tile durations list-scheduled heaviest first, profiles/r03_tile_timeline.txt).
__global__ void iota_kernel(uint32_t* v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
// temp_bytes == 0: only reports the temporary storage the sort needs
hipError_t sort_tiles_by_cost(const uint32_t* cost, uint32_t* cost_sorted, uint32_t* iota, uint32_t* order, uint32_t n,
                              void* temp, size_t& temp_bytes, hipStream_t stream) {
    if (temp == nullptr)
        return hipcub::DeviceRadixSort::SortPairsDescending(nullptr, temp_bytes, cost, cost_sorted, iota, order, (int)n, 0, 32, stream);
    hipLaunchKernelGGL(iota_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, iota, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return hipcub::DeviceRadixSort::SortPairsDescending(temp, temp_bytes, cost, cost_sorted, iota, order, (int)n, 0, 32, stream);
}

}  // namespace p3d
