// color_device.h -- device-side scalar colour conversions (f64), expression-for-expression the
// reference's lib/src/color/*.c (cited per function).  Compiled with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/patolette_amd.h"
#include "pow_tables.h"

namespace pamd {

constexpr int PAMD_COPY = 100;       // internal: no conversion (colour space sRGB)

constexpr int kStatSlots = 32;       // same-address atomics serialise in L2: blocks spread over slots
struct ConvertStats {                // filled by k_convert / k_weight_stats (ordered-key min/max per slot)
    unsigned long long minkey[kStatSlots][3], maxkey[kStatSlots][3], wmaxkey[kStatSlots];
    double sum[kStatSlots][3][2];    // binned column sums of the converted image (the root mean of the global quantiser)
    double mom[kStatSlots][6][2];    // binned raw second moments xx, yx, zx, yy, zy, zz (the root covariance)
    unsigned int nonfinite_f32;      // some converted value is NaN / Inf once cast to float (faiss Clustering.cpp:295-304 scans for that)
};

// pow(x, y) for the conversions: x >= 0 (anything else returns what libm returns: NaN for x < 0 with a non-integer y),
// any finite y.  The conversions spend nine pow() per pixel (ICtCp) and the library routine costs ~250 instructions;
// this one ~70: log2 by a 128-entry table + degree-8 series, the product y*log2(x) and the argument of exp2 carried
// as double-double, exp2 by a 64-entry table + degree-7 series (tables: tools/gen_pow_tables.py).  Error <= 0.51 ulp:
// 99.8 % of results are bit-identical to glibc's correctly rounded pow, the rest are its neighbours.
// The two tables (4 KB) are read at data-dependent places in the middle of every pow: from global memory that is an L1 round
// trip on the critical path, nine times in a row per ICtCp pixel.  A translation unit that defines PAMD_POW_TABLES_IN_LDS before
// including this header keeps a copy in LDS: each of its kernels calls pow_tables_to_lds() (+ a barrier) before its first pow.
#ifdef PAMD_POW_TABLES_IN_LDS
__shared__ double s_pow_log[128][3];
__shared__ double s_pow_exp[64][2];
__device__ __forceinline__ void pow_tables_to_lds() {
    for (int i = threadIdx.x; i < 128 * 3; i += blockDim.x) (&s_pow_log[0][0])[i] = (&powtab::kLog[0][0])[i];
    for (int i = threadIdx.x; i < 64 * 2; i += blockDim.x) (&s_pow_exp[0][0])[i] = (&powtab::kExp[0][0])[i];
}
#define PAMD_POW_LOG s_pow_log
#define PAMD_POW_EXP s_pow_exp
#else
#define PAMD_POW_LOG kLog
#define PAMD_POW_EXP kExp
#endif
// One Horner step z * acc + c with the coefficient in scalar registers: written with __builtin_fma the compiler emits
// v_mov_b64 (coefficient into the accumulator) + v_fmac_f64 -- two vector instructions per step, thirteen steps per pow, nine
// pow per ICtCp pixel -- where one v_fma_f64 reading the constant from an SGPR pair does the same arithmetic.
__device__ __forceinline__ double horner_sc(const double z, const double acc, const double c) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(z), "v"(acc), "s"(c));
    return r;
}
__device__ __forceinline__ double pamd_pow(double x, double y) {
    using namespace powtab;
    if (!(x > 0.0)) return x == 0.0 ? (y > 0 ? 0.0 : (y == 0 ? 1.0 : INFINITY)) : NAN;
    if (isinf(x)) return y > 0 ? INFINITY : (y == 0 ? 1.0 : 0.0);
    const double m = __builtin_amdgcn_frexp_mant(x) * 2.0;                 // [1, 2)
    const int e = __builtin_amdgcn_frexp_exp(x) - 1;
    const int i = (int)((m - 1.0) * 128.0) & 127;                           // (already in 0..127: the mask tells the compiler, whose full
                                                                             // 32-bit multiply for the table offset runs at a quarter of the rate)
    const double r = PAMD_POW_LOG[i][0], Thi = PAMD_POW_LOG[i][1], Tlo = PAMD_POW_LOG[i][2];
    const double ph = m * r, pl = __builtin_fma(m, r, -ph);                  // m*r exactly = ph + pl
    const double zh = ph - 1.0, zl = pl;                                     // z = m*r - 1 exactly = zh + zl, |z| < 2^-8
    const double a = zh * kInvLn2Hi, ae = __builtin_fma(zh, kInvLn2Hi, -a);
    const double s = (double)e + Thi;                                        // exact: Thi is a multiple of 2^-42
    const double Lh = s + a, bb = Lh - s, err = (s - (Lh - bb)) + (a - bb);  // two-sum
    const double z = zh;
    const double poly = (z * z) * horner_sc(z, horner_sc(z, horner_sc(z, horner_sc(z, horner_sc(z, horner_sc(z, kL8, kL7), kL6), kL5), kL4), kL3), kL2);
    const double Ll = ((((err + ae) + zh * kInvLn2Lo) + zl * kInvLn2Hi) + Tlo) + (poly + (2 * kL2) * (zh * zl));
    const double Ph = y * Lh, Pl = __builtin_fma(y, Lh, -Ph) + y * Ll;       // y * log2(x) = Ph + Pl
    if (Ph > 1100.0) return INFINITY;
    if (Ph < -1200.0) return 0.0;
    const double kd = __builtin_rint(Ph * 64.0);
    const double f = (Ph - kd * 0.015625) + Pl;                              // the subtraction is exact
    const int k = (int)kd;                                                   // |kd| <= 1200 * 64: one conversion (a 64-bit one costs six)
    const int j = k & 63, n = k >> 6;
    const double q = f * horner_sc(f, horner_sc(f, horner_sc(f, horner_sc(f, horner_sc(f, horner_sc(f, kE7, kE6), kE5), kE4), kE3), kE2), kE1);
    const double Th = PAMD_POW_EXP[j][0], Tl = PAMD_POW_EXP[j][1];
    return ldexp(Th + __builtin_fma(Th, q, Tl), n);
}

namespace dc {
// eotf.c:13-18
__device__ constexpr double Lp = 10000, m1 = 0.1593017578125, m2 = 78.84375;
__device__ constexpr double c1 = 0.8359375, c2 = 18.8515625, c3 = 18.6875;

// a / D for a constant D, correctly rounded in three instructions instead of the ~20 of the generic IEEE sequence:
// y = RN(1/D) (folded at compile time), q = RN(a y), r = a - q D (exact, one FMA), result RN(q + r y) = RN(a / D)
// (Markstein's theorem; checked against exact rational arithmetic for these divisors).  The residual must not underflow, so
// operands below 1e-200 take the generic division.
__device__ __forceinline__ double div_const(const double a, const double D) {
    const double y = 1.0 / D;
    if (__builtin_expect(fabs(a) < 1e-200 && a != 0.0, 0)) return a / D;
    const double q = a * y;
    const double r = __builtin_fma(-q, D, a);
    return __builtin_fma(r, y, q);
}

__device__ __forceinline__ double eotf(double v) {                       // eotf.c:29-42
    double m1d = 1 / m1, m2d = 1 / m2;
    double V_p = pamd_pow(v, m2d);
    double n = fmax(0.0, V_p - c1);
    double L = pamd_pow((n / (c2 - c3 * V_p)), m1d);
    return Lp * L;
}
__device__ __forceinline__ double eotf_inv(double v) {                   // eotf.c:44-57
    double y_ = pamd_pow(div_const(v, Lp), m1);
    return pamd_pow((c1 + c2 * y_) / (1 + c3 * y_), m2);
}
__device__ __forceinline__ double gamma_decode(double c) {               // sRGB.c:70-89
    double r = (c <= 0.0404500) ? div_const(c, 12.92) : pamd_pow(div_const(c + 0.055, 1.055), 2.4);
    return fmin(fmax(r, 0.0), 1.0);
}
__device__ __forceinline__ double gamma_encode(double c) {               // sRGB.c:91-110
    double r = (c <= 0.0031308) ? c * 12.92 : 1.055 * pamd_pow(c, 1.0 / 2.4) - 0.055;
    return fmin(fmax(r, 0.0), 1.0);
}
__device__ __forceinline__ void linear_to_xyz(double R, double G, double B, double &x, double &y, double &z) {   // xyz.c:27-39
    x = R * 0.4124564 + G * 0.3575761 + B * 0.1804375;
    y = R * 0.2126729 + G * 0.7151522 + B * 0.0721750;
    z = R * 0.0193339 + G * 0.1191920 + B * 0.9503041;
}
__device__ __forceinline__ void srgb_to_xyz(const double c[3], double &x, double &y, double &z) {   // xyz.c:14-40
    linear_to_xyz(gamma_decode(c[0]), gamma_decode(c[1]), gamma_decode(c[2]), x, y, z);
}
__device__ __forceinline__ void xyz_to_rec2020(double x, double y, double z, double o[3]) {          // rec2020.c:80-102
    o[0] = x * 1.71666343 + y * -0.35567332 + z * -0.25336809;
    o[1] = x * -0.66667384 + y * 1.61645574 + z * 0.0157683;
    o[2] = x * 0.01764248 + y * -0.04277698 + z * 0.94224328;
}
__device__ __forceinline__ void srgb_to_rec2020(double c[3]) {          // rec2020.c:104-126
    double x, y, z;
    srgb_to_xyz(c, x, y, z);
    xyz_to_rec2020(x, y, z, c);
}
__device__ __forceinline__ void rec2020_to_ictcp(double c[3]) {         // ICtCp.c:41-79 (Ct halved)
    double r = c[0], g = c[1], b = c[2];
    double L = (r * 1688 + g * 2146 + b * 262) / 4096;
    double M = (r * 683 + g * 2951 + b * 462) / 4096;
    double S = (r * 99 + g * 309 + b * 3688) / 4096;
    double L_ = eotf_inv(L), M_ = eotf_inv(M), S_ = eotf_inv(S);
    c[0] = L_ * 0.5 + M_ * 0.5;
    c[1] = (L_ * 6610 - M_ * 13613 + S_ * 7003) / 4096;
    c[2] = (L_ * 17933 - M_ * 17390 - S_ * 543) / 4096;
    c[1] *= 0.5;
}
__device__ __forceinline__ void ictcp_to_rec2020(double c[3]) {         // rec2020.c:32-69 (Ct doubled)
    double I = c[0], Ct = c[1] * 2, Cp = c[2];
    double L_ = I + 0.00860904 * Ct + 0.11102963 * Cp;
    double M_ = I - 0.00860904 * Ct - 0.11102963 * Cp;
    double S_ = I + 0.56003134 * Ct - 0.32062717 * Cp;
    double L = eotf(L_), M = eotf(M_), S = eotf(S_);
    c[0] = L * 3.43660669 - M * 2.50645212 + S * 0.06984542;
    c[1] = -L * 0.79132956 + M * 1.98360045 - S * 0.1922709;
    c[2] = -L * 0.0259499 - M * 0.09891371 + S * 1.12486361;
}
// CIELuv.c:19-25
__device__ constexpr double rwx = 0.95047, rwy = 1.0, rwz = 1.08883;
__device__ constexpr double kE = 216.0 / 24389.0, kK = 24389.0 / 27.0, kKE = 8.0;

__device__ __forceinline__ void linear_to_cieluv(double c[3]);
__device__ __forceinline__ void srgb_to_cieluv(double c[3]) {           // CIELuv.c:166-197 + :54-89
    c[0] = gamma_decode(c[0]); c[1] = gamma_decode(c[1]); c[2] = gamma_decode(c[2]);
    linear_to_cieluv(c);
}
__device__ __forceinline__ void linear_to_cieluv(double c[3]) {         // the part after the companding
    double r = c[0], g = c[1], b = c[2];
    double x = r * 0.4124564 + g * 0.3575761 + b * 0.1804375;
    double y = r * 0.2126729 + g * 0.7151522 + b * 0.0721750;
    double z = r * 0.0193339 + g * 0.1191920 + b * 0.9503041;
    double den = x + 15.0 * y + 3.0 * z;
    double up = (den > 0.0) ? ((4.0 * x) / (x + 15.0 * y + 3.0 * z)) : 0.0;
    double vp = (den > 0.0) ? ((9.0 * y) / (x + 15.0 * y + 3.0 * z)) : 0.0;
    double urp = (4.0 * rwx) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double vrp = (9.0 * rwy) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double yr = y / rwy;
    double L_ = (yr > kE) ? (116.0 * pamd_pow(yr, 1.0 / 3.0) - 16.0) : (kK * yr);
    c[0] = L_;
    c[1] = 13.0 * L_ * (up - urp);
    c[2] = 13.0 * L_ * (vp - vrp);
}
__device__ __forceinline__ void cieluv_to_rec2020(double c[3]) {        // CIELuv.c:100-164 + rec2020.c:150-173
    double L = c[0], u = c[1], v = c[2];
    double y_ = (L > kKE) ? pamd_pow((L + 16.0) / 116.0, 3.0) : (L / kK);
    double u0 = (4.0 * rwx) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double v0 = (9.0 * rwy) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double a, a_den = u + 13.0 * L * u0;
    if (a_den == 0.0) a = 0; else a = (((52.0 * L) / a_den) - 1.0) / 3.0;
    double b = -5.0 * y_;
    double cc = -1.0 / 3.0;
    double d, d_den = v + 13.0 * L * v0;
    if (d_den == 0.0) d = 0; else d = y_ * (((39.0 * L) / d_den) - 5.0);
    double x_, x_den = a - cc;
    if (x_den == 0.0) x_ = 0; else x_ = (d - b) / x_den;
    double z_ = x_ * a + b;
    xyz_to_rec2020(x_, y_, z_, c);
}
__device__ __forceinline__ void rec2020_to_srgb(double c[3]) {          // sRGB.c:32-59 + xyz.c:42-64
    double r2 = c[0], g2 = c[1], b2 = c[2];
    double x = r2 * 0.63695351 + g2 * 0.14461919 + b2 * 0.16885585;
    double y = r2 * 0.26269834 + g2 * 0.67800877 + b2 * 0.0592929;
    double z = g2 * 0.02807314 + b2 * 1.06082723;
    double r = x * 3.2404542 - y * 1.5371385 - z * 0.4985314;
    double g = -x * 0.9692660 + y * 1.8760108 + z * 0.0415560;
    double b = x * 0.0556434 - y * 0.2040259 + z * 1.0572252;
    c[0] = gamma_encode(r); c[1] = gamma_encode(g); c[2] = gamma_encode(b);
}
}  // namespace dc

// pixel sources: planar f64 sRGB (the reference ABI) or interleaved 8-bit sRGB (v/255.0 in f64, what every
// caller of the reference computes by hand, README.md:156-158)
struct SrcF64 {
    const double *p; size_t n;
    __device__ __forceinline__ void load(size_t i, double c[3]) const { c[0] = p[i]; c[1] = p[n + i]; c[2] = p[2 * n + i]; }
};
struct SrcF64Rows {                  // (N,3) row-major f64: what numpy hands over without a transpose
    const double *p;
    __device__ __forceinline__ void load(size_t i, double c[3]) const { c[0] = p[3 * i]; c[1] = p[3 * i + 1]; c[2] = p[3 * i + 2]; }
};
struct SrcU8 {
    const unsigned char *p; int ch;
    __device__ __forceinline__ void load_bytes(size_t i, unsigned &r, unsigned &g, unsigned &b) const {
        const unsigned char *q = p + i * (size_t)ch;
        r = q[0]; g = q[1]; b = q[2];
    }
    __device__ __forceinline__ void load(size_t i, double c[3]) const {
        const unsigned char *q = p + i * (size_t)ch;
        c[0] = (double)q[0] / 255.0; c[1] = (double)q[1] / 255.0; c[2] = (double)q[2] / 255.0;
    }
};

// conversions out of sRGB whose companding (sRGB.c:70-89) has already been applied: 8-bit sources look the 256
// possible values up instead of evaluating three pamd_pow() per pixel; same expressions, same results
template <int WHICH>
__device__ __forceinline__ void dev_convert_linear(double c[3]) {
    if constexpr (WHICH == PAMD_SRGB_TO_ICTCP) {
        double x, y, z;
        dc::linear_to_xyz(c[0], c[1], c[2], x, y, z);
        dc::xyz_to_rec2020(x, y, z, c);
        dc::rec2020_to_ictcp(c);
    } else if constexpr (WHICH == PAMD_SRGB_TO_CIELUV) {
        dc::linear_to_cieluv(c);
    }
}

template <int WHICH>
__device__ __forceinline__ void dev_convert(double c[3]) {
    if constexpr (WHICH == PAMD_SRGB_TO_ICTCP) { dc::srgb_to_rec2020(c); dc::rec2020_to_ictcp(c); }
    else if constexpr (WHICH == PAMD_SRGB_TO_CIELUV) { dc::srgb_to_cieluv(c); }
    else if constexpr (WHICH == PAMD_ICTCP_TO_REC2020) { dc::ictcp_to_rec2020(c); }
    else if constexpr (WHICH == PAMD_CIELUV_TO_REC2020) { dc::cieluv_to_rec2020(c); }
    else if constexpr (WHICH == PAMD_SRGB_TO_REC2020) { dc::srgb_to_rec2020(c); }
    else if constexpr (WHICH == PAMD_REC2020_TO_SRGB) { dc::rec2020_to_srgb(c); }
    else if constexpr (WHICH == PAMD_CIELUV_TO_ICTCP) {                 // patolette.c:305-314 fused
        dc::cieluv_to_rec2020(c); dc::rec2020_to_srgb(c); dc::srgb_to_rec2020(c); dc::rec2020_to_ictcp(c);
    }
}

}  // namespace pamd
