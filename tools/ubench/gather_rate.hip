// gather_rate.hip -- how fast a CU serves per-lane global_load_dwordx4 by address pattern (gfx950 microbenchmark).
// Patterns: 0 = all lanes one address, 1 = lanes consecutive (1 KiB per instruction), 2 = every lane a random 16-B quad,
// 3 = every lane the 4 quads of a random 64-B record (the BVH node fetch), 4 = like 3 with lanes in groups of 8 sharing a record.
// Table sizes: 2 MiB (L2 of one XCD holds it), 64 MiB (Infinity Cache).   build: hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7FEB352Du; a ^= a >> 15; a *= 0x846CA68Bu; a ^= a >> 16; return a; }

template <int PATTERN>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ table, uint32_t n_records, int iters, float* out) {
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float4 acc = make_float4(0, 0, 0, 0);
    uint32_t h = mix(wave * 977u + 1u);
    for (int i = 0; i < iters; i++) {
        h = mix(h + (uint32_t)i);
        uint32_t rec;
        if (PATTERN == 0) rec = h % n_records;
        else if (PATTERN == 1) rec = (h % (n_records - 64)) + lane / 4;
        else if (PATTERN == 4) rec = mix(h ^ ((lane >> 3) * 0x9E3779B9u)) % n_records;
        else rec = mix(h ^ (lane * 0x9E3779B9u)) % n_records;
        const float4* p = table + (size_t)rec * 4;
        if (PATTERN == 1) p += lane & 3;
        if (PATTERN == 3 || PATTERN == 4) {
            float4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc.x += a.x + b.y + c.z + d.w;
        } else if (PATTERN == 2) {
            float4 a = p[h & 3];
            acc.x += a.x;
        } else {
            float4 a = p[0];
            acc.x += a.x;
        }
    }
    if (acc.x == 123.456f) out[0] = acc.x;
}

template <int PATTERN>
static void run(const char* name, const float4* table, uint32_t n_records, int waves_per_cu, float* out) {
    const int cus = 256, iters = 2000;
    dim3 grid(cus * waves_per_cu / 4), block(256);
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<PATTERN>, grid, block, 0, 0, table, n_records, 100, out);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<PATTERN>, grid, block, 0, 0, table, n_records, iters, out);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)cus * waves_per_cu * iters * ((PATTERN == 3 || PATTERN == 4) ? 4 : 1);
    printf("%-44s table %6.1f MiB  %2d waves/CU: %8.3f ms  %7.2f ns per wave-load per CU  (%.1f cycles @2.4GHz)  %.2f TB/s\n", name,
           n_records * 64.0 / 1048576, waves_per_cu, ms, ms * 1e6 / (instr / cus), ms * 1e6 / (instr / cus) * 2.4, instr * 1024 / ms / 1e9);
}

int main() {
    for (uint32_t mib : {2u, 64u}) {
        const uint32_t n_records = mib * 1048576u / 64u;
        float4* table; float* out;
        CHECK(hipMalloc(&table, (size_t)n_records * 64)); CHECK(hipMalloc(&out, 64));
        CHECK(hipMemset(table, 0, (size_t)n_records * 64));
        for (int w : {8, 24}) {
            run<0>("one address for the wave", table, n_records, w, out);
            run<1>("consecutive lanes (1 KiB)", table, n_records, w, out);
            run<2>("random quad per lane", table, n_records, w, out);
            run<3>("4 quads of a random 64-B record per lane", table, n_records, w, out);
            run<4>("4 quads of a record, 8 lanes per record", table, n_records, w, out);
        }
        CHECK(hipFree(table)); CHECK(hipFree(out));
    }
    return 0;
}
