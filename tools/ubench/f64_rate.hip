// f64_rate.hip -- micro-benchmark (tool, not product): issue rate of the FP64 VALU ops the libm powf restatement is made
// of, beside FP32 FMA, on gfx950 at 1..8 waves per SIMD; and the cost of one powf call, device library against
// csrc/p3d_powf.h.  Prints s_memtime ticks (100 MHz) per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iu_4a_2s_p3d_raytracer_template2_amd/csrc tools/ubench/f64_rate.hip -o tools/ubench/f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "p3d_powf.h"

#define REP 64
#define ITERS 128

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, float seed) {
    double d0 = seed + threadIdx.x, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, m = 1.0000001, c = 1e-9;
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, fm = 1.0001f, fc = 1e-9f;
    float px = 0.3f + 1e-4f * threadIdx.x, py = 20.0f, acc = 0.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (KIND == 0) {
                asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fm), "v"(fc));
            } else if (KIND == 1) {
                asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             "v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(m), "v"(c));
            } else if (KIND == 2) {
                asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                             "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(m));
            } else if (KIND == 3) {
                asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                             "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(c));
            } else if (KIND == 4) {   // dependent chain
                asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                             "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
                             : "+v"(d0) : "v"(m), "v"(c));
            } else if (KIND == 5) {   // conversions: 4 x (f32 -> f64 -> f32)
                asm volatile("v_cvt_f64_f32 %4, %0\n v_cvt_f32_f64 %0, %4\n v_cvt_f64_f32 %5, %1\n v_cvt_f32_f64 %1, %5\n"
                             "v_cvt_f64_f32 %6, %2\n v_cvt_f32_f64 %2, %6\n v_cvt_f64_f32 %7, %3\n v_cvt_f32_f64 %3, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            } else if (KIND == 6) {   // dependent f32 fma chain
                asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                             : "+v"(a0) : "v"(fm), "v"(fc));
            } else if (KIND == 7) {   // 8 calls of the device library's powf (each feeds the next base)
                for (int q = 0; q < 8; q++) { float v = powf(px, py); acc += v; px = 0.2f + 0.5f * px + v; }
            } else if (KIND == 8) {   // 8 calls of the restatement, tables in constant memory
                for (int q = 0; q < 8; q++) { float v = p3d::p3d_powf_nonneg(px, py, p3d::PowTabConst()); acc += v; px = 0.2f + 0.5f * px + v; }
            } else if (KIND == 9) {   // the same, tables in LDS
                for (int q = 0; q < 8; q++) { float v = p3d::p3d_powf_nonneg(px, py, p3d::PowTabLds()); acc += v; px = 0.2f + 0.5f * px + v; }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)(d0 + d1 + d2 + d3) + acc;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}


template <int KIND>
static void run(const char* name, int insts_per_rep8) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 3 * 256 * 512 * 4); hipMalloc(&cyc, 3 * 256 * 8 * 8);
    for (int waves_per_simd : {1, 2, 4, 6}) {
        int threads = waves_per_simd == 1 ? 256 : 512;                     // blocks of 1 or 2 waves per SIMD ...
        int blocks = 256 * (waves_per_simd == 1 ? 1 : waves_per_simd / 2); // ... 1, 1, 2 or 3 of them per CU
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 1024, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        double per_wave = s / h.size();
        double n_inst = (double)ITERS * (REP / 8) * insts_per_rep8;
        printf("%-44s waves/SIMD %d: %.3f ticks per wave-op (per wave), %.3f per SIMD\n", name, waves_per_simd, per_wave / n_inst,
               per_wave / n_inst / waves_per_simd);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32 x4 indep", 8);
    run<6>("v_fma_f32 dependent", 8);
    run<1>("v_fma_f64 x4 indep", 8);
    run<2>("v_mul_f64 x4 indep", 8);
    run<3>("v_add_f64 x4 indep", 8);
    run<4>("v_fma_f64 dependent", 8);
    run<5>("v_cvt_f64_f32 + v_cvt_f32_f64 x4", 8);
    run<7>("powf, device library (per call)", 8);
    run<8>("powf restated, constant tables (per call)", 8);
    run<9>("powf restated, LDS tables (per call; garbage)", 8);
    return 0;
}
