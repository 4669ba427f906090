// p3d_powf.h -- the host C library's powf(), restated for the device.
//
// The reference's Blinn-Phong term calls powf(max(0, H.N), shine) (RT/main.cpp:520) and its result reaches the
// frame unrounded, so the one float operation of the hot path that is NOT an IEEE-754 basic operation decides whether
// rgb32f is equal or merely close to the reference's.  The device library's powf is a different (1-2 ulp) algorithm.
// This is the algorithm of the C library the reference links on the build image and on the GPU box -- glibc 2.35,
// sysdeps/ieee754/flt-32/e_powf.c (Szabolcs Nagy's table-driven powf of ARM optimized-routines), in the variant
// x86-64 selects at load time on a host with FMA (__powf_fma): log2(x) from a 16-entry table and a degree-5
// polynomial in double, y*log2(x) in double, 2^s from a 32-entry table and a cubic, one final rounding to float.
// glibc is not part of /root/reference and not vendored there; the constants below are the published tables of that
// routine (POWF_LOG2_TABLE_BITS 4, EXP2F_TABLE_BITS 5), read out of the image's libm.so.6 and checked value by
// value; which multiply-adds are fused was read from the same object code.  The final rounding makes the result
// the correctly rounded float in all but about 1 case in 10^5 -- and in those the device agrees with the host
// because every double operation is the same IEEE operation in the same order.
// tests/test_gpu_powf.py checks it bit for bit against the box's own libm over random and edge-case arguments.
#ifndef P3D_POWF_H
#define P3D_POWF_H

#include <stdint.h>
#if defined(P3D_POWF_TABLES_ONLY)
// host code that only wants the table initialisers (p3d_capi.cpp)
#elif defined(P3D_POWF_HOST_CHECK)
// tests/test_powf_port.py compiles this header with g++ -mfma to run the same expressions against libm on the CPU
#include <string.h>
#define __device__
#define __forceinline__ inline
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline long long __double_as_longlong(double d) { long long u; memcpy(&u, &d, 8); return u; }
static inline double __longlong_as_double(long long u) { double d; memcpy(&d, &u, 8); return d; }
#else
#include <hip/hip_runtime.h>
#endif

namespace p3d {

// {1/c, log2(c)} for the 16 sub-intervals of [0x1.66p-1, 0x1.66p0)
#define P3D_POW_LOG2_TAB_INIT { \
    {0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2}, {0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2}, \
    {0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2}, {0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2}, \
    {0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2}, {0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3}, \
    {0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3}, {0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4}, \
    {0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5}, {0x1p+0, 0x0p+0}, \
    {0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4},  {0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3}, \
    {0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2}, \
    {0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2},  {0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2}, }
// bits of 2^(i/32) with i << 47 subtracted from them, so that adding k << 47 carries the integer part into the exponent
#define P3D_POW_EXP2_TAB_INIT { \
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, \
    0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, \
    0x3feedea64c123422ull, 0x3feece086061892dull, 0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, \
    0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, \
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, 0x3feee89f995ad3adull, \
    0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, \
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull, }
// the polynomial coefficients, in the order the LDS copy holds them (pairs that one 16-byte read fetches together):
// log2: A0 A1 | A2 A3 | A4, exp2: SHIFT | C0 C1 | C2, pad
#define P3D_POW_COEF_INIT { 0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1, \
    0x1.71547652ab82bp+0, 0x1.8p+47, 0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1, 0.0 }
#define P3D_POW_TAB_BYTES 592u              /* 16 x {double, double} + 32 x u64 + 10 doubles: the first 37 quads of a scene blob */
#ifndef P3D_POWF_TABLES_ONLY
__device__ static const double kPowLog2Tab[16][2] = P3D_POW_LOG2_TAB_INIT;
__device__ static const uint64_t kPowExp2Tab[32] = P3D_POW_EXP2_TAB_INIT;

// Where a kernel reads the two tables: constant memory, or -- kernels that render from an LDS copy of the scene blob --
// the head of that copy (p3d_capi.cpp puts the tables first in every blob): two LDS reads instead of two trips to L1/L2.
struct PowTabConst {
    __device__ __forceinline__ void log2_entry(int i, double& invc, double& logc) const { invc = kPowLog2Tab[i][0]; logc = kPowLog2Tab[i][1]; }
    __device__ __forceinline__ uint64_t exp2_entry(uint32_t j) const { return kPowExp2Tab[j]; }
    __device__ __forceinline__ void log2_coefs(double& a0, double& a1, double& a2, double& a3, double& a4) const {
        const double c[10] = P3D_POW_COEF_INIT;
        a0 = c[0]; a1 = c[1]; a2 = c[2]; a3 = c[3]; a4 = c[4];
    }
    __device__ __forceinline__ void exp2_coefs(double& shift, double& c0, double& c1, double& c2) const {
        const double c[10] = P3D_POW_COEF_INIT;
        shift = c[5]; c0 = c[6]; c1 = c[7]; c2 = c[8];
    }
};
#ifndef P3D_POWF_HOST_CHECK
extern __shared__ __attribute__((aligned(16))) uint32_t p3d_lds[];
struct PowTabLds {
    __device__ __forceinline__ void log2_entry(int i, double& invc, double& logc) const {
        const double2 e = reinterpret_cast<const double2*>(p3d_lds)[i]; invc = e.x; logc = e.y;
    }
    __device__ __forceinline__ uint64_t exp2_entry(uint32_t j) const { return reinterpret_cast<const uint64_t*>(p3d_lds)[32 + j]; }
    // The coefficients come from LDS as well (uniform address: one broadcast read per pair).  As literals they need
    // scalar register pairs, which the compiler hoists out of the light loop: ten more live SGPRs pushed SGPR spill code
    // into the traversal loops of the level kernels (+37 vector / +21 scalar instructions per wave, config 2 -7 %).
    __device__ __forceinline__ void log2_coefs(double& a0, double& a1, double& a2, double& a3, double& a4) const {
        const double2* c = reinterpret_cast<const double2*>(p3d_lds) + 32;
        const double2 p0 = c[0], p1 = c[1], p2 = c[2];
        a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x;
    }
    __device__ __forceinline__ void exp2_coefs(double& shift, double& c0, double& c1, double& c2) const {
        const double2* c = reinterpret_cast<const double2*>(p3d_lds) + 32;
        const double2 p2 = c[2], p3 = c[3], p4 = c[4];
        shift = p2.y; c0 = p3.x; c1 = p3.y; c2 = p4.x;
    }
};

// Kernels that read the scene from HBM read the same 592 bytes at the head of the blob through the vector memory path
// (L1 hits after the first wave): VGPR temporaries again, no LDS -- 592 bytes of LDS per one-wave workgroup cost the 10^6-
// primitive scene a resident wave per CU (2.16 against 1.97 ms per frame in flight).  `zero` is a VGPR holding 0 that the
// compiler cannot see through: with a provably uniform address it would use scalar loads, i.e. SGPR pairs, again.
struct PowTabGlobal {
    const double2* base; uint32_t zero;
    __device__ __forceinline__ explicit PowTabGlobal(const void* blob) : base(reinterpret_cast<const double2*>(blob)) {
        zero = 0u; asm volatile("" : "+v"(zero));
    }
    __device__ __forceinline__ void log2_entry(int i, double& invc, double& logc) const { const double2 e = base[(uint32_t)i + zero]; invc = e.x; logc = e.y; }
    __device__ __forceinline__ uint64_t exp2_entry(uint32_t j) const { return reinterpret_cast<const uint64_t*>(base)[32u + j + zero]; }
    __device__ __forceinline__ void log2_coefs(double& a0, double& a1, double& a2, double& a3, double& a4) const {
        const double2* c = base + 32u + zero;
        const double2 p0 = c[0], p1 = c[1], p2 = c[2];
        a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x;
    }
    __device__ __forceinline__ void exp2_coefs(double& shift, double& c0, double& c1, double& c2) const {
        const double2* c = base + 32u + zero;
        const double2 p2 = c[2], p3 = c[3], p4 = c[4];
        shift = p2.y; c0 = p3.x; c1 = p3.y; c2 = p4.x;
    }
};
#endif

// 0: y is not an integer, 1: odd integer, 2: even integer (decides the sign and validity of (x < 0)^y)
__device__ __forceinline__ int pow_int_kind(uint32_t iy) {
    const int e = (int)((iy >> 23) & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    const uint32_t unit = 1u << (0x7f + 23 - e);
    if (iy & (unit - 1u)) return 0;
    return (iy & unit) ? 1 : 2;
}
__device__ __forceinline__ bool pow_zero_inf_nan(uint32_t i) { return 2u * i - 1u >= 2u * 0x7f800000u - 1u; }

__device__ __forceinline__ bool pow_signaling(uint32_t i) { return 2u * (i ^ 0x00400000u) > 2u * 0x7fc00000u; }

// log2 of the positive normal float whose bits are ix: k + log2(c) + log2(z / c), z in [0x1.66p-1, 0x1.66p0), r = z / c - 1
template <class TAB>
__device__ __forceinline__ double pow_log2(uint32_t ix, const TAB& tab) {
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int)top >> 23;
    double invc, logc;
    tab.log2_entry(i, invc, logc);
    const double z = (double)__uint_as_float(iz);
    double A0, A1, A2, A3, A4;
    tab.log2_coefs(A0, A1, A2, A3, A4);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    const double q1 = __builtin_fma(A0, r, A1);
    const double p = __builtin_fma(A2, r, A3);
    const double r4 = r2 * r2;
    double q = __builtin_fma(A4, r, y0);
    q = __builtin_fma(p, r2, q);
    return __builtin_fma(q1, r4, q);
}
// 2^s rounded to float: 2^(k/32) * 2^r with |r| <= 1/64; sign_bias (0 or 1 << 16) lands on the sign bit
template <class TAB>
__device__ __forceinline__ float pow_exp2(double s, uint64_t sign_bias, const TAB& tab) {
    double kShift, C0, C1, C2;
    tab.exp2_coefs(kShift, C0, C1, C2);
    double kd = s + kShift;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= kShift;
    const double r = s - kd;
    const uint64_t t = tab.exp2_entry((uint32_t)ki & 31u) + ((ki + sign_bias) << 47);
    const double scale = __longlong_as_double((long long)t);
    const double z = __builtin_fma(C0, r, C1);
    const double r2 = r * r;
    double e = __builtin_fma(C2, r, 1.0);
    e = __builtin_fma(z, r2, e);
    return (float)(e * scale);
}

// Every case of the routine, written with its branches: the arguments shading never produces on a healthy scene
// (negative, infinite or NaN bases; zero, infinite or NaN exponents) come here, and so does the debug probe's full sweep.
template <class TAB>
__device__ inline float powf_any(float x, float y, const TAB& tab) {
    uint32_t ix = __float_as_uint(x);
    const uint32_t iy = __float_as_uint(y);
    uint64_t sign_bias = 0;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || pow_zero_inf_nan(iy)) {
        // x is zero, subnormal, negative, infinite or NaN, or y is zero, infinite or NaN
        if (pow_zero_inf_nan(iy)) {
            if (2u * iy == 0u) return pow_signaling(ix) ? x + y : 1.0f;      // x^0 (quiet NaN included)
            if (ix == 0x3f800000u) return pow_signaling(iy) ? x + y : 1.0f;  // 1^y
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;                    // (-1)^(+-inf)
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (pow_zero_inf_nan(ix)) {
            float x2 = x * x;
            bool neg = false;
            if ((ix & 0x80000000u) && pow_int_kind(iy) == 1) { x2 = -x2; neg = true; }
            if (2u * ix == 0u && (iy & 0x80000000u)) return neg ? -__builtin_inff() : __builtin_inff();   // 0^(y < 0)
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {                                              // finite x < 0
            const int kind = pow_int_kind(iy);
            if (kind == 0) return __builtin_nanf("");
            if (kind == 1) sign_bias = 1ull << 16;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {                                              // subnormal: scale into the normal range
            ix = __float_as_uint(x * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    const double ylogx = (double)y * pow_log2(ix, tab);
    if (((uint64_t)__double_as_longlong(ylogx) >> 47 & 0xffffu) >= 0x80bfu) {    // |y log2 x| >= 126
        if (ylogx > 0x1.fffffffd1d571p+6) return sign_bias ? -__builtin_inff() : __builtin_inff();
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
        if (ylogx < -149.0) return sign_bias ? -0x1p-149f : 0x1p-149f;           // 0x1.4p-75f * 0x1.4p-75f, rounded to nearest
    }
    return pow_exp2(ylogx, sign_bias, tab);
}

// powf(x, y) for a base whose sign bit is clear and which is not NaN: what the shading calls.  Its base is
// max(0, H.N) = (0 < v) ? v : 0 -- +0 on every lane that faces away (and for a NaN v), else positive -- and its exponent is
// whatever float the material holds.  Positive normal bases with finite non-zero exponents run straight-line code and
// (+0)^y is a select; everything else -- subnormal or infinite bases, zero, infinite or NaN exponents, results beyond
// the float range -- sits behind wave-level branches no wave of a healthy scene takes.  The level-1 kernel issues
// scalar instructions at a third of the vector rate from ONE scalar unit per CU, so the shape of this function is
// chosen for its scalar count: masks are combined once, constants that need scalar registers are shared.
template <class TAB>
__device__ __forceinline__ float p3d_powf_nonneg(float x, float y, const TAB& tab) {
    const uint32_t ix = __float_as_uint(x), iy = __float_as_uint(y);
    const bool y_ok = ((iy & 0x7fffffffu) - 1u) < 0x7f7fffffu;                   // finite and not zero
    const bool hot = y_ok && (ix - 0x00800000u) < 0x7f000000u;                   // positive normal base
    const bool fine = hot || (y_ok && ix == 0u);
    // bypassed lanes compute finite or NaN garbage (the table indices are masked into range)
    const double ylogx = (double)y * pow_log2(ix, tab);
    float r = pow_exp2(ylogx, 0, tab);
    const uint32_t hi = (uint32_t)((uint64_t)__double_as_longlong(ylogx) >> 32);
    const bool big = hot && ((hi >> 15) & 0xffffu) >= 0x80bfu;                   // |y log2 x| >= 126
#ifndef P3D_POWF_HOST_CHECK
    if (__ballot(big) != 0) {
        asm volatile("");                        // keeps this a branch: the three f64 compares stay off the common path
#else
    {
#endif
        if (big) {
            r = ylogx < -149.0 ? 0x1p-149f : r;
            r = ylogx <= -150.0 ? 0.0f : r;
            r = ylogx > 0x1.fffffffd1d571p+6 ? __builtin_inff() : r;
        }
    }
    r = ix == 0u ? ((iy & 0x80000000u) ? __builtin_inff() : 0.0f) : r;           // (+0)^y
#ifndef P3D_POWF_HOST_CHECK
    if (__ballot(!fine) != 0) {
        asm volatile("");
#else
    {
#endif
        if (!fine) r = powf_any(x, y, tab);
    }
    return r;
}

// any arguments (the debug probe)
__device__ __forceinline__ float p3d_powf(float x, float y) {
    return (__float_as_uint(x) <= 0x7f800000u) ? p3d_powf_nonneg(x, y, PowTabConst()) : powf_any(x, y, PowTabConst());
}
#endif  // P3D_POWF_TABLES_ONLY

}  // namespace p3d
#endif
