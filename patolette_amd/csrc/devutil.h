// devutil.h -- device-side helpers shared by the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pamd {

constexpr int kWave = 64;            // CDNA wavefront

// --------------------------------------------------------------------------------------------
// Order-independent ("binned") floating-point accumulation.
//
// Every reduction that feeds a reported float or a discrete decision splits each addend v
// (|v| <= 2^E, at most 2^P addends) into two parts that lie on fixed grids:
//     v0 = v rounded to a multiple of g0 = 2^(E-B),   v1 = (v - v0) rounded to a multiple of
//     g1 = g0 * 2^-B,   B = 51 - P.
// Sums of up to 2^P such parts never leave the 53-bit significand, so they are EXACT in any
// order: per-thread partials, wave shuffles, LDS atomics and global f64 atomics all give the
// same bits regardless of launch geometry or scheduling.  The result (sum0 + sum1, one
// rounding) is the sum of the addends each rounded to ~2B bits below the bound -- at least as
// accurate as the sequential f64 sums of the reference, and run-to-run deterministic.
// --------------------------------------------------------------------------------------------
struct BinK {            // magic constants: M0 = 1.5 * 2^52 * g0, M1 = 1.5 * 2^52 * g1
    double M0, M1;
};

__host__ __device__ inline BinK make_bink(int E, int P) {
    int B = 51 - P;
    if (B > 40) B = 40;
    if (B < 8) B = 8;
    BinK k;
    k.M0 = ldexp(1.5, 52 + E - B);
    k.M1 = ldexp(1.5, 52 + E - 2 * B);
    return k;
}

__device__ __forceinline__ void bin_add(double v, const BinK k, double &a0, double &a1) {
    double v0 = (v + k.M0) - k.M0;
    double r = v - v0;
    double v1 = (r + k.M1) - k.M1;
    a0 += v0;
    a1 += v1;
}
__device__ __forceinline__ void bin_split(double v, const BinK k, double &v0, double &v1) {
    v0 = (v + k.M0) - k.M0;
    double r = v - v0;
    v1 = (r + k.M1) - k.M1;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// Sum over the wavefront by DPP moves (VALU only, no LDS round trips): the total lands in lane 63.  For exact addends.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(const double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int tlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int thi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(thi, tlo);
}
__device__ __forceinline__ double wave_sum_dpp63(double v) {
    v = dpp_add_f64<0x111, 0xf>(v);          // row_shr:1
    v = dpp_add_f64<0x112, 0xf>(v);          // row_shr:2
    v = dpp_add_f64<0x114, 0xf>(v);          // row_shr:4
    v = dpp_add_f64<0x118, 0xf>(v);          // row_shr:8  -> lane 15 of every row holds the row's sum
    v = dpp_add_f64<0x142, 0xa>(v);          // row_bcast:15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xc>(v);          // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// inclusive prefix sum over the 64 lanes through DPP row shifts and row broadcasts (a __shfl_up loop is six LDS round trips)
__device__ __forceinline__ unsigned wave_scan_incl_u32(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);      // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);      // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);      // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);      // row_shr:8  -> inclusive within each row of 16
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);      // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);      // row_bcast:31 into rows 2 and 3
    return v;
}

// Block-wide sum of NV doubles per thread (exact addends -> any order); result valid in thread 0.
// smem must hold NV * (blockDim.x / 64) doubles.
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *smem) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++) smem[wid * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < nw; w++) {
#pragma unroll
            for (int i = 0; i < NV; i++) v[i] += smem[w * NV + i];
        }
    }
    __syncthreads();
}

// Monotone map double -> uint64 so that min / max can use integer atomics (exact, order-free).
__host__ __device__ inline unsigned long long f64_key(double d) {
    union { double f; unsigned long long u; } c;
    c.f = d;
    return (c.u & 0x8000000000000000ULL) ? ~c.u : (c.u | 0x8000000000000000ULL);
}
__host__ __device__ inline double key_f64(unsigned long long k) {
    union { double f; unsigned long long u; } c;
    c.u = (k & 0x8000000000000000ULL) ? (k & 0x7FFFFFFFFFFFFFFFULL) : ~k;
    return c.f;
}

// splitmix64 -> u(seed, i) in [0,1)   
__host__ __device__ inline unsigned long long splitmix64(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ULL;
    unsigned long long z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ inline double u01(unsigned long long seed, unsigned long long i) {
    return (double)(splitmix64((seed << 32) + i) >> 11) * 0x1.0p-53;
}

}  // namespace pamd
