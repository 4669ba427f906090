// valu_rate.hip -- micro-benchmark (tool, not product): issue rate of scalar vs packed FP32 VALU ops and
// of the IEEE divide / sqrt expansions on gfx950, at 1..8 waves per SIMD.  Prints cycles per
// wave-instruction per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
#define ITERS 256

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, unsigned long long* cyc, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float m = 1.0001f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (KIND == 0) {          // 8 independent v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                             "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (KIND == 1) {   // 4 independent v_pk_mul_f32 (same flops as 8 v_mul)
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m};
                asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm));
                a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
            } else if (KIND == 2) {   // dependent chain of v_mul_f32
                asm volatile("v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                             "v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1\n"
                             : "+v"(a0) : "v"(m));
            } else if (KIND == 3) {   // dependent chain of v_pk_mul_f32
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 p0 = {a0, a1}, mm = {m, m};
                asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n"
                             "v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %0, %0, %1\n"
                             : "+v"(p0) : "v"(mm));
                a0 = p0.x; a1 = p0.y;
            } else if (KIND == 4) {   // 8 IEEE divides (compiler expansion), independent
                a0 = m / a0; a1 = m / a1; a2 = m / a2; a3 = m / a3; a4 = m / a4; a5 = m / a5; a6 = m / a6; a7 = m / a7;
            } else if (KIND == 5) {   // 8 IEEE sqrt
                a0 = __builtin_sqrtf(a0 + 2.f); a1 = __builtin_sqrtf(a1 + 2.f); a2 = __builtin_sqrtf(a2 + 2.f); a3 = __builtin_sqrtf(a3 + 2.f);
                a4 = __builtin_sqrtf(a4 + 2.f); a5 = __builtin_sqrtf(a5 + 2.f); a6 = __builtin_sqrtf(a6 + 2.f); a7 = __builtin_sqrtf(a7 + 2.f);
            } else if (KIND == 6) {   // 8 v_mov_b32
                asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                             "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 8) {   // 8 independent s_add_u32 / s_mul_i32 (scalar ALU)
                unsigned s0 = it, s1 = it + 1, s2 = it + 2, s3 = it + 3;
                asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                             "s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
                a0 += (float)(s0 + s1 + s2 + s3) * 1e-30f;
            } else if (KIND == 9) {   // 4 x (v_cmp_lt_f32 vcc ; v_cndmask vcc): the compare/select pairs of the hit tests
                asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %4, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %4, vcc\n"
                             "v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %4, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");
            } else if (KIND == 10) {  // 4 x (v_cmp -> s_and_saveexec -> restore): a divergent `if` without a branch
                asm volatile("v_cmp_lt_f32 vcc, %0, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %0, %4\n s_mov_b64 exec, s[20:21]\n"
                             "v_cmp_lt_f32 vcc, %1, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %1, %1, %4\n s_mov_b64 exec, s[20:21]\n"
                             "v_cmp_lt_f32 vcc, %2, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %2, %2, %4\n s_mov_b64 exec, s[20:21]\n"
                             "v_cmp_lt_f32 vcc, %3, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %3, %3, %4\n s_mov_b64 exec, s[20:21]\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc", "s20", "s21");
            } else if (KIND == 11) {  // 8 v_readfirstlane_b32 (VALU -> SGPR)
                unsigned t0, t1, t2, t3;
                asm volatile("v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %6\n v_readfirstlane_b32 %3, %7\n"
                             "v_readfirstlane_b32 %0, %4\n v_readfirstlane_b32 %1, %5\n v_readfirstlane_b32 %2, %6\n v_readfirstlane_b32 %3, %7\n"
                             : "=s"(t0), "=s"(t1), "=s"(t2), "=s"(t3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
                a4 += (float)(t0 ^ t1 ^ t2 ^ t3) * 1e-30f;
            } else if (KIND == 12) {  // LDS round trip: dependent ds_read_b32 chain (address from the previous read)
                extern __shared__ unsigned lds[];
                unsigned idx = threadIdx.x & 255;
                lds[idx] = idx;
                for (int q = 0; q < 8; q++) idx = lds[idx & 255];
                a5 += (float)idx * 1e-30f;
            } else if (KIND == 13) {  // 8 v_mul_f32 + 4 s_add_u32 interleaved in ONE wave (the 2:1 mix of the ray kernels)
                unsigned s0 = it, s1 = it + 1;
                asm volatile("v_mul_f32 %0, %0, %10\n v_mul_f32 %1, %1, %10\n s_add_u32 %8, %8, 3\n v_mul_f32 %2, %2, %10\n v_mul_f32 %3, %3, %10\n s_add_u32 %9, %9, 5\n"
                             "v_mul_f32 %4, %4, %10\n v_mul_f32 %5, %5, %10\n s_add_u32 %8, %8, 3\n v_mul_f32 %6, %6, %10\n v_mul_f32 %7, %7, %10\n s_add_u32 %9, %9, 5\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(m) : "scc");
                a0 += (float)(s0 + s1) * 1e-30f;
            } else if (KIND == 14) {  // even waves: 8 v_mul_f32, odd waves: 8 s_add_u32 -- do vector and scalar issue of DIFFERENT waves overlap?
                if (((threadIdx.x >> 6) & 1) == 0) {
                    asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                                 "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
                } else {
                    unsigned s0 = it, s1 = it + 1, s2 = it + 2, s3 = it + 3;
                    asm volatile("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                                 "s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n s_add_u32 %2, %2, 7\n s_add_u32 %3, %3, 9\n"
                                 : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
                    a0 += (float)(s0 + s1 + s2 + s3) * 1e-30f;
                }
            } else if (KIND == 7) {   // 8 v_cndmask
                asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m) : "vcc");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int insts_per_rep8) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4 * 8); hipMalloc(&cyc, 256 * 16 * 8 * 8);
    for (int waves_per_simd : {1, 2, 4}) {
        int threads = 64 * 4 * waves_per_simd;    // one block per CU
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 1024, 0, out, cyc, 1.0f);
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(threads), 1024, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * threads / 64);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        double per_wave = s / h.size();
        double n_inst = (double)ITERS * (REP / 8) * insts_per_rep8;
        // s_memtime ticks at 100 MHz?  report both raw ticks per instruction and per-SIMD rate
        printf("%-28s waves/SIMD %d: %.2f ticks per wave-instr (per wave), %.2f ticks per wave-instr per SIMD\n", name,
               waves_per_simd, per_wave / n_inst, per_wave / n_inst / waves_per_simd);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_mul_f32 x8 indep", 8);
    run<1>("v_pk_mul_f32 x4 indep", 4);
    run<2>("v_mul_f32 dependent", 8);
    run<3>("v_pk_mul_f32 dependent", 8);
    run<4>("IEEE fdiv x8 (expansion)", 8);
    run<5>("IEEE sqrt x8 (expansion)", 8);
    run<6>("v_mov_b32 x8", 8);
    run<7>("v_cndmask x8 (vcc stale)", 8);
    run<8>("s_add_u32 x8 (+2 valu)", 8);
    run<9>("v_cmp+v_cndmask x4 pairs", 8);
    run<10>("cmp/saveexec/add/restore x4", 16);
    run<11>("v_readfirstlane x8 (+2 valu)", 8);
    run<12>("dependent ds_read_b32 x8", 8);
    run<13>("8 v_mul + 4 s_add, one wave", 12);
    run<14>("v_mul waves beside s_add waves", 8);
    return 0;
}
