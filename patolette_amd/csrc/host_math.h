// host_math.h -- host-side scalar pieces of the pipeline (product code, C++):
//   * 3x3 symmetric eigen-solve following LAPACK dsyev('V','L') for n = 3 -- what the reference
//     calls at lib/src/math/eigen.c:83-140; the eigenvector SIGN LAPACK returns defines bucket
//     direction, hence left/right children and palette order (hard part 1);
//   * scalar colour conversions for the K palette rows (lib/src/color/*.c) -- run on the host
//     with libm so the final sRGB palette follows the reference's own arithmetic;
//   * the global-quantiser DP over 512 bucket moments (lib/src/quantize/global.c:99-298,
//     cells.c:141-328) -- O(12*512^2) scalar work on 45 KB of data, not worth a launch.
// Compiled with -ffp-contract=off; FMAs appear only where written.  The eigen-solver is __host__ __device__: the split loop's
// device-side control kernel (pipeline.hip k_lq_control) runs the very same code, one problem per lane (IEEE f64 add / mul /
// fma / div / sqrt are correctly rounded on gfx950 as on the host, so the eigenvector signs are the same bits).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#if defined(__HIPCC__)
#define PAMD_HD __host__ __device__
#else
#define PAMD_HD
#endif

namespace pamd {
namespace hm {

// --------------------------------------------------------------------------------------------
// LAPACK dsyev('V','L'), n = 3:  dsytd2('L') -> dorgtr('L') -> dsteqr('V')
// --------------------------------------------------------------------------------------------
constexpr double kEps = 0x1.0p-53;                        // dlamch('E')
constexpr double kSafmin = 2.2250738585072014e-308;       // dlamch('S')

PAMD_HD inline double sgn(double a, double b) { return std::copysign(std::fabs(a), b); }
PAMD_HD inline double lapy2(double x, double y) {
    double xa = std::fabs(x), ya = std::fabs(y);
    double w = xa > ya ? xa : ya, z = xa > ya ? ya : xa;
    if (z == 0.0 || w > 1.79769313486231571e308) return w;
    double q = z / w;
    return w * std::sqrt(1.0 + q * q);
}
PAMD_HD inline void lartg(double f, double g, double &c, double &s, double &r) {   // LAPACK >= 3.10
    const double safmax = 1.0 / kSafmin;
    const double rtmin = std::sqrt(kSafmin), rtmax = std::sqrt(safmax / 2);
    double f1 = std::fabs(f), g1 = std::fabs(g);
    if (g == 0.0) { c = 1.0; s = 0.0; r = f; }
    else if (f == 0.0) { c = 0.0; s = sgn(1.0, g); r = g1; }
    else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
        double d = std::sqrt(f * f + g * g);
        c = f1 / d; r = sgn(d, f); s = g / r;
    } else {
        double u = std::fmin(safmax, std::fmax(kSafmin, std::fmax(f1, g1)));
        double fs = f / u, gs = g / u;
        double d = std::sqrt(fs * fs + gs * gs);
        c = std::fabs(fs) / d; r = sgn(d, f); s = gs / r; r = r * u;
    }
}
PAMD_HD inline void laev2(double a, double b, double c, double &rt1, double &rt2, double &cs1, double &sn1) {
    double sm = a + c, df = a - c, adf = std::fabs(df), tb = b + b, ab = std::fabs(tb);
    double acmx, acmn, rt;
    int sgn1, sgn2;
    if (std::fabs(a) > std::fabs(c)) { acmx = a; acmn = c; } else { acmx = c; acmn = a; }
    if (adf > ab) { double q = ab / adf; rt = adf * std::sqrt(1.0 + q * q); }
    else if (adf < ab) { double q = adf / ab; rt = ab * std::sqrt(1.0 + q * q); }
    else rt = ab * std::sqrt(2.0);
    if (sm < 0.0) { rt1 = 0.5 * (sm - rt); sgn1 = -1; rt2 = (acmx / rt1) * acmn - (b / rt1) * b; }
    else if (sm > 0.0) { rt1 = 0.5 * (sm + rt); sgn1 = 1; rt2 = (acmx / rt1) * acmn - (b / rt1) * b; }
    else { rt1 = 0.5 * rt; rt2 = -0.5 * rt; sgn1 = 1; }
    double cs;
    if (df >= 0.0) { cs = df + rt; sgn2 = 1; } else { cs = df - rt; sgn2 = -1; }
    if (std::fabs(cs) > ab) { double ct = -tb / cs; sn1 = 1.0 / std::sqrt(1.0 + ct * ct); cs1 = ct * sn1; }
    else if (ab == 0.0) { cs1 = 1.0; sn1 = 0.0; }
    else { double tn = -cs / tb; cs1 = 1.0 / std::sqrt(1.0 + tn * tn); sn1 = tn * cs1; }
    if (sgn1 == sgn2) { double tn = cs1; cs1 = -sn1; sn1 = tn; }
}
// dlasr('R','V',F|B) on a 3-row column-major block starting at column pointer z
PAMD_HD inline void lasr(bool forward, int mm, const double *c, const double *s, double *z) {
    for (int t = 0; t < mm - 1; t++) {
        int j = forward ? t : (mm - 2 - t);
        double ct = c[j], st = s[j];
        if (ct != 1.0 || st != 0.0) {
            for (int i = 0; i < 3; i++) {
                double temp = z[(j + 1) * 3 + i];
                z[(j + 1) * 3 + i] = ct * temp - st * z[j * 3 + i];
                z[j * 3 + i] = st * temp + ct * z[j * 3 + i];
            }
        }
    }
}

PAMD_HD inline int steqr3(double *dd, double *ee, double *z) {
    const int n = 3;
    const double eps2 = kEps * kEps, safmax = 1.0 / kSafmin;
    const double ssfmax = std::sqrt(safmax) / 3.0, ssfmin = std::sqrt(kSafmin) / eps2;
    const int nmaxit = n * 30;
    int jtot = 0, l1 = 1;
    double wc[2], ws[2];
    auto D = [&](int i) -> double & { return dd[i - 1]; };
    auto E = [&](int i) -> double & { return ee[i - 1]; };
    auto Z = [&](int i) -> double * { return z + (i - 1) * 3; };
    while (l1 <= n) {
        if (l1 > 1) E(l1 - 1) = 0.0;
        int m = n;
        for (int mm = l1; mm <= n - 1; mm++) {
            double tst = std::fabs(E(mm));
            if (tst == 0.0) { m = mm; break; }
            if (tst <= (std::sqrt(std::fabs(D(mm))) * std::sqrt(std::fabs(D(mm + 1)))) * kEps) { E(mm) = 0.0; m = mm; break; }
        }
        int l = l1, lsv = l, lend = m, lendsv = lend;
        l1 = m + 1;
        if (lend == l) continue;
        double anorm = 0.0;
        for (int i = l; i <= lend; i++) anorm = std::fabs(D(i)) > anorm ? std::fabs(D(i)) : anorm;
        for (int i = l; i <= lend - 1; i++) anorm = std::fabs(E(i)) > anorm ? std::fabs(E(i)) : anorm;
        if (anorm == 0.0) continue;
        int iscale = 0;
        if (anorm > ssfmax) { iscale = 1; for (int i = l; i <= lend; i++) D(i) *= ssfmax / anorm; for (int i = l; i < lend; i++) E(i) *= ssfmax / anorm; }
        else if (anorm < ssfmin) { iscale = 2; for (int i = l; i <= lend; i++) D(i) *= ssfmin / anorm; for (int i = l; i < lend; i++) E(i) *= ssfmin / anorm; }
        if (std::fabs(D(lend)) < std::fabs(D(l))) { lend = lsv; l = lendsv; }
        if (lend > l) {                                   // QL
            for (;;) {
                m = lend;
                if (l != lend)
                    for (int mm = l; mm <= lend - 1; mm++) {
                        double tst = std::fabs(E(mm)) * std::fabs(E(mm));
                        if (tst <= (eps2 * std::fabs(D(mm))) * std::fabs(D(mm + 1)) + kSafmin) { m = mm; break; }
                    }
                if (m < lend) E(m) = 0.0;
                double p = D(l);
                if (m == l) { l++; if (l <= lend) continue; break; }
                if (m == l + 1) {
                    double rt1, rt2, c, s;
                    laev2(D(l), E(l), D(l + 1), rt1, rt2, c, s);
                    wc[0] = c; ws[0] = s;
                    lasr(false, 2, wc, ws, Z(l));
                    D(l) = rt1; D(l + 1) = rt2; E(l) = 0.0;
                    l += 2; if (l <= lend) continue; break;
                }
                if (jtot == nmaxit) break;
                jtot++;
                double g = (D(l + 1) - p) / (2.0 * E(l));
                double r = lapy2(g, 1.0);
                g = D(m) - p + (E(l) / (g + sgn(r, g)));
                double s = 1.0, c = 1.0;
                p = 0.0;
                double tc[2], ts[2];
                for (int i = m - 1; i >= l; i--) {
                    double f = s * E(i), b = c * E(i);
                    lartg(g, f, c, s, r);
                    if (i != m - 1) E(i + 1) = r;
                    g = D(i + 1) - p;
                    r = (D(i) - g) * s + 2.0 * c * b;
                    p = s * r;
                    D(i + 1) = g + p;
                    g = c * r - b;
                    tc[i - l] = c; ts[i - l] = -s;
                }
                lasr(false, m - l + 1, tc, ts, Z(l));
                D(l) = D(l) - p;
                E(l) = g;
            }
        } else {                                          // QR
            for (;;) {
                m = lend;
                if (l != lend)
                    for (int mm = l; mm >= lend + 1; mm--) {
                        double tst = std::fabs(E(mm - 1)) * std::fabs(E(mm - 1));
                        if (tst <= (eps2 * std::fabs(D(mm))) * std::fabs(D(mm - 1)) + kSafmin) { m = mm; break; }
                    }
                if (m > lend) E(m - 1) = 0.0;
                double p = D(l);
                if (m == l) { l--; if (l >= lend) continue; break; }
                if (m == l - 1) {
                    double rt1, rt2, c, s;
                    laev2(D(l - 1), E(l - 1), D(l), rt1, rt2, c, s);
                    wc[0] = c; ws[0] = s;
                    lasr(true, 2, wc, ws, Z(l - 1));
                    D(l - 1) = rt1; D(l) = rt2; E(l - 1) = 0.0;
                    l -= 2; if (l >= lend) continue; break;
                }
                if (jtot == nmaxit) break;
                jtot++;
                double g = (D(l - 1) - p) / (2.0 * E(l - 1));
                double r = lapy2(g, 1.0);
                g = D(m) - p + (E(l - 1) / (g + sgn(r, g)));
                double s = 1.0, c = 1.0;
                p = 0.0;
                double tc[2], ts[2];
                for (int i = m; i <= l - 1; i++) {
                    double f = s * E(i), b = c * E(i);
                    lartg(g, f, c, s, r);
                    if (i != m) E(i - 1) = r;
                    g = D(i) - p;
                    r = (D(i + 1) - g) * s + 2.0 * c * b;
                    p = s * r;
                    D(i) = g + p;
                    g = c * r - b;
                    tc[i - m] = c; ts[i - m] = s;
                }
                lasr(true, l - m + 1, tc, ts, Z(m));
                D(l) = D(l) - p;
                E(l - 1) = g;
            }
        }
        if (iscale == 1) { for (int i = lsv; i <= lendsv; i++) D(i) *= anorm / ssfmax; for (int i = lsv; i < lendsv; i++) E(i) *= anorm / ssfmax; }
        else if (iscale == 2) { for (int i = lsv; i <= lendsv; i++) D(i) *= anorm / ssfmin; for (int i = lsv; i < lendsv; i++) E(i) *= anorm / ssfmin; }
        if (jtot >= nmaxit) { int info = 0; for (int i = 1; i < n; i++) if (E(i) != 0.0) info++; return info; }
    }
    for (int ii = 2; ii <= n; ii++) {                     // selection sort, ascending
        int i = ii - 1, k = i;
        double p = D(i);
        for (int j = ii; j <= n; j++) if (D(j) < p) { k = j; p = D(j); }
        if (k != i) {
            D(k) = D(i); D(i) = p;
            for (int r = 0; r < 3; r++) { const double t_ = Z(i)[r]; Z(i)[r] = Z(k)[r]; Z(k)[r] = t_; }
        }
    }
    return 0;
}

// a: column-major 3x3, lower triangle read; out: w ascending, a = eigenvectors (columns).
// The level-1/2 BLAS steps inside dsytd2/dorg2r are written with the FMAs OpenBLAS' x86-64
// kernels contract them to (validated against OpenBLAS 0.3.28 dsyev, tests/golden/eigen_*).
PAMD_HD inline int eigen_sym3(double a[9], double w[3]) {
    double a11 = a[0], a21 = a[1], a31 = a[2], a22 = a[4], a32 = a[5], a33 = a[8];
    double d[3], e[2], tau = 0.0, v2 = 0.0;
    {
        double alpha = a21, x = a31;
        double xnorm = std::fabs(x);
        if (xnorm != 0.0) {                                // dlarfg
            double beta = -sgn(lapy2(alpha, xnorm), alpha);
            const double sfm = kSafmin / kEps, rsfm = 1.0 / sfm;
            int knt = 0;
            if (std::fabs(beta) < sfm) {
                do { knt++; x *= rsfm; beta *= rsfm; alpha *= rsfm; } while (std::fabs(beta) < sfm && knt < 20);
                xnorm = std::fabs(x);
                beta = -sgn(lapy2(alpha, xnorm), alpha);
            }
            tau = (beta - alpha) / beta;
            x = x * (1.0 / (alpha - beta));
            for (int j = 0; j < knt; j++) beta *= sfm;
            alpha = beta;
            v2 = x;
        }
        e[0] = alpha;
        if (tau != 0.0) {
            double t1 = tau * 1.0;                         // dsymv('L')
            double y1 = std::fma(t1, a22, 0.0), y2 = std::fma(t1, a32, 0.0);
            double t2 = std::fma(a32, v2, 0.0);
            y1 = std::fma(tau, t2, y1);
            t1 = tau * v2;
            y2 = std::fma(t1, a33, y2);
            double dot = std::fma(y2, v2, y1 * 1.0);       // ddot
            double al = -0.5 * tau * dot;
            double w1 = std::fma(al, 1.0, y1), w2 = std::fma(al, v2, y2);   // daxpy
            double temp1 = -1.0 * w1, temp2 = -1.0 * 1.0;  // dsyr2('L')
            a22 = std::fma(w1, temp2, std::fma(1.0, temp1, a22));
            a32 = std::fma(w2, temp2, std::fma(v2, temp1, a32));
            temp1 = -1.0 * w2; temp2 = -1.0 * v2;
            a33 = std::fma(w2, temp2, std::fma(v2, temp1, a33));
        }
        d[0] = a11;
    }
    e[1] = a32; d[1] = a22; d[2] = a33;
    double z[9];
    z[0] = 1.0; z[1] = 0.0; z[2] = 0.0; z[3] = 0.0; z[6] = 0.0;
    if (tau != 0.0) {                                      // dorg2r on the trailing 2x2
        double temp = -tau * v2;
        z[7] = 0.0 + 1.0 * temp;
        z[8] = std::fma(v2, temp, 1.0);
    } else { z[7] = 0.0; z[8] = 1.0; }
    z[5] = -tau * v2;
    z[4] = 1.0 - tau;
    int info = steqr3(d, e, z);
    w[0] = d[0]; w[1] = d[1]; w[2] = d[2];
    for (int i = 0; i < 9; i++) a[i] = z[i];
    return info;
}

// Principal axis of a covariance given as its 6 lower-triangle entries (xx,xy,xz,yy,yz,zz).
PAMD_HD inline bool principal_axis(const double c6[6], double axis[3]) {
    double a[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    double w[3];
    if (eigen_sym3(a, w) != 0) return false;
    axis[0] = a[6]; axis[1] = a[7]; axis[2] = a[8];
    return true;
}

// An UPPER bound of the largest eigenvalue of a symmetric positive semi-definite 3x3 (xx,yx,zx,yy,zy,zz) without the eigen-solve:
// with m = trace / 3 and q = sqrt(|A - m I|_F^2 / 6) the eigenvalues are m + 2 q cos(phi + 2 pi k / 3), so lambda_max <= m + 2 q
// (exact for an isotropic and for a rank-one matrix); a few Newton steps on the characteristic polynomial, which is convex and
// increasing above its largest root, come down towards it FROM ABOVE and are taken only while p(x) is clearly positive
// (1e-10 x^3: far beyond the rounding of the evaluation), so every iterate stays an upper bound.  Used where only a bound is
// needed at once and the exact solve can run beside other work (the device-driven split loop).  NaN in -> NaN out.
PAMD_HD inline double lambda_max_bound(const double c[6]) {
    const double tr = c[0] + c[3] + c[5], m = tr / 3.0;
    const double d0 = c[0] - m, d1 = c[3] - m, d2 = c[5] - m;
    const double fro = d0 * d0 + d1 * d1 + d2 * d2 + 2.0 * (c[1] * c[1] + c[2] * c[2] + c[4] * c[4]);
    double x = m + 2.0 * std::sqrt(fro / 6.0);
    x = x * (1.0 + 1e-12);
    const double c2 = tr;
    const double c1 = (c[0] * c[3] - c[1] * c[1]) + (c[0] * c[5] - c[2] * c[2]) + (c[3] * c[5] - c[4] * c[4]);
    const double c0 = c[0] * (c[3] * c[5] - c[4] * c[4]) - c[1] * (c[1] * c[5] - c[4] * c[2]) + c[2] * (c[1] * c[4] - c[3] * c[2]);
    for (int it = 0; it < 4; it++) {
        const double px = ((x - c2) * x + c1) * x - c0, dpx = (3.0 * x - 2.0 * c2) * x + c1;
        if (!(px > 1e-10 * x * x * x) || !(dpx > 0.0)) break;
        const double xn = x - px / dpx;
        if (!(xn > 0.0) || !(xn < x)) break;
        x = xn;
    }
    return x;
}

// --------------------------------------------------------------------------------------------
// Scalar colour conversions for palette rows (reference: lib/src/color/*.c)
// --------------------------------------------------------------------------------------------
namespace color {
constexpr double Lp = 10000, m1 = 0.1593017578125, m2 = 78.84375, c1 = 0.8359375, c2 = 18.8515625, c3 = 18.6875;
inline double eotf(double v) {                             // eotf.c:29-42
    double V_p = std::pow(v, 1 / m2);
    double n = std::fmax(0, V_p - c1);
    return Lp * std::pow(n / (c2 - c3 * V_p), 1 / m1);
}
inline double eotf_inv(double v) {                         // eotf.c:44-57
    double y_ = std::pow(v / Lp, m1);
    return std::pow((c1 + c2 * y_) / (1 + c3 * y_), m2);
}
inline double gamma_decode(double c) {                     // sRGB.c:70-89
    double r = (c <= 0.0404500) ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4);
    return std::fmin(std::fmax(r, 0.0), 1.0);
}
inline double gamma_encode(double c) {                     // sRGB.c:91-110
    double r = (c <= 0.0031308) ? c * 12.92 : 1.055 * std::pow(c, 1.0 / 2.4) - 0.055;
    return std::fmin(std::fmax(r, 0.0), 1.0);
}
inline void xyz_to_rec2020(double x, double y, double z, double o[3]) {     // rec2020.c:80-102
    o[0] = x * 1.71666343 + y * -0.35567332 + z * -0.25336809;
    o[1] = x * -0.66667384 + y * 1.61645574 + z * 0.0157683;
    o[2] = x * 0.01764248 + y * -0.04277698 + z * 0.94224328;
}
inline void srgb_to_xyz(const double c[3], double &x, double &y, double &z) {   // xyz.c:14-40
    double R = gamma_decode(c[0]), G = gamma_decode(c[1]), B = gamma_decode(c[2]);
    x = R * 0.4124564 + G * 0.3575761 + B * 0.1804375;
    y = R * 0.2126729 + G * 0.7151522 + B * 0.0721750;
    z = R * 0.0193339 + G * 0.1191920 + B * 0.9503041;
}
inline void srgb_to_rec2020(double c[3]) { double x, y, z; srgb_to_xyz(c, x, y, z); xyz_to_rec2020(x, y, z, c); }
inline void rec2020_to_ictcp(double c[3]) {                // ICtCp.c:41-79
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
inline void srgb_to_ictcp(double c[3]) { srgb_to_rec2020(c); rec2020_to_ictcp(c); }
inline void ictcp_to_rec2020(double c[3]) {                // rec2020.c:32-69
    double I = c[0], Ct = c[1] * 2, Cp = c[2];
    double L_ = I + 0.00860904 * Ct + 0.11102963 * Cp;
    double M_ = I - 0.00860904 * Ct - 0.11102963 * Cp;
    double S_ = I + 0.56003134 * Ct - 0.32062717 * Cp;
    double L = eotf(L_), M = eotf(M_), S = eotf(S_);
    c[0] = L * 3.43660669 - M * 2.50645212 + S * 0.06984542;
    c[1] = -L * 0.79132956 + M * 1.98360045 - S * 0.1922709;
    c[2] = -L * 0.0259499 - M * 0.09891371 + S * 1.12486361;
}
constexpr double rwx = 0.95047, rwy = 1.0, rwz = 1.08883;
constexpr double kK = 24389.0 / 27.0, kKE = 8.0;
inline void cieluv_to_rec2020(double c[3]) {               // CIELuv.c:100-164 + rec2020.c:150-173
    double L = c[0], u = c[1], v = c[2];
    double y_ = (L > kKE) ? std::pow((L + 16.0) / 116.0, 3.0) : (L / kK);
    double u0 = (4.0 * rwx) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double v0 = (9.0 * rwy) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double a, a_den = u + 13.0 * L * u0;
    if (!a_den) a = 0; else a = (((52.0 * L) / a_den) - 1.0) / 3.0;
    double b = -5.0 * y_;
    double cc = -1.0 / 3.0;
    double d, d_den = v + 13.0 * L * v0;
    if (!d_den) d = 0; else d = y_ * (((39.0 * L) / d_den) - 5.0);
    double x_, x_den = a - cc;
    if (!x_den) x_ = 0; else x_ = (d - b) / x_den;
    double z_ = x_ * a + b;
    xyz_to_rec2020(x_, y_, z_, c);
}
inline void rec2020_to_srgb(double c[3]) {                 // sRGB.c:32-59
    double r2 = c[0], g2 = c[1], b2 = c[2];
    double x = r2 * 0.63695351 + g2 * 0.14461919 + b2 * 0.16885585;      // xyz.c:42-64
    double y = r2 * 0.26269834 + g2 * 0.67800877 + b2 * 0.0592929;
    double z = g2 * 0.02807314 + b2 * 1.06082723;
    double r = x * 3.2404542 - y * 1.5371385 - z * 0.4985314;
    double g = -x * 0.9692660 + y * 1.8760108 + z * 0.0415560;
    double b = x * 0.0556434 - y * 0.2040259 + z * 1.0572252;
    c[0] = gamma_encode(r); c[1] = gamma_encode(g); c[2] = gamma_encode(b);
}
}  // namespace color

// --------------------------------------------------------------------------------------------
// Global principal quantiser on the 512-bucket moment table (reference: cells.c, global.c)
// --------------------------------------------------------------------------------------------
struct CellMoments {                                        // inclusive prefix sums, 1-based (cells.c:53-139)
    uint64_t w0[513];
    double w1[3][513];
    double w2[513];
    double wrs[6][513];                                     // (r,s) r<=s: 00,01,11,02,12,22 -> index s*(s+1)/2 + r
    static int rs(int r, int s) { return s * (s + 1) / 2 + r; }
    double distortion(size_t a, size_t b) const {           // cells.c:141-182
        if (w0[a] == w0[b]) return 0;
        double q0 = w1[0][b] - w1[0][a], q1 = w1[1][b] - w1[1][a], q2 = w1[2][b] - w1[2][a];
        return w2[b] - w2[a] - (q0 * q0 + q1 * q1 + q2 * q2) / (double)(w0[b] - w0[a]);
    }
    double vcov(size_t a, size_t b, int r, int s) const {   // cells.c:184-223
        if (w0[a] == w0[b]) return 0;
        double cnt = (double)(w0[b] - w0[a]);
        return (wrs[rs(r, s)][b] - wrs[rs(r, s)][a]) / cnt - (w1[r][b] - w1[r][a]) * (w1[s][b] - w1[s][a]) / (cnt * cnt);
    }
    bool axis(size_t a, size_t b, double ax[3]) const {     // cells.c:225-278
        double m[9] = {0};
        for (int s = 0; s < 3; s++) for (int r = 0; r <= s; r++) m[s * 3 + r] = vcov(a, b, r, s);
        m[0 * 3 + 2] = m[2 * 3 + 0]; m[0 * 3 + 1] = m[1 * 3 + 0]; m[1 * 3 + 2] = m[2 * 3 + 1];
        double w[3];
        if (eigen_sym3(m, w) != 0) return false;
        ax[0] = m[6]; ax[1] = m[7]; ax[2] = m[8];
        return true;
    }
    static double norm3(const double a[3]) { double s = 0; for (int i = 0; i < 3; i++) s += std::pow(a[i], 2); return std::sqrt(s); }
    double bias(size_t a, size_t b, const double ax[3]) const {   // cells.c:280-328
        double ca[3];
        if (!axis(a, b, ca)) return -1;
        double norms = norm3(ax) * norm3(ca);
        if (norms < 1e-16) return 0;
        double dot = (ca[0] * ax[0] + ca[1] * ax[1] + ca[2] * ax[2]);
        return std::fmin(1, std::fabs(dot / norms));
    }
};

inline bool gq_should_terminate(const std::vector<size_t> &q, const double ax[3], const CellMoments &c, bool &error) {  // global.c:99-187
    double distortion = 0;
    for (size_t j = 0; j + 1 < q.size(); j++) distortion += c.distortion(q[j], q[j + 1]);
    if (distortion < 1e-16) return true;
    double bias = 0;
    for (size_t i = 0; i + 1 < q.size(); i++) {
        double cd = c.distortion(q[i], q[i + 1]);
        double cb = c.bias(q[i], q[i + 1], ax);
        if (cb < 0) { error = true; return true; }
        if (cb < 0.9) continue;
        bias += (cd / distortion) * cb;
    }
    return bias < 0.1;
}

// global.c:189-298: returns the cut vector [0 = q0, ..., qk = 512]; empty on error.  The O(k * 512^2) dynamic
// programme itself (global.c:232-280) runs on the device (quant.hip k_gq_dp) for every k up to max_k; `cut[k][n]` is
// the reference's L[k][n].  Here: the bias termination test per k (global.c:99-187) and the backtrack.
inline std::vector<size_t> gq_principal_quantizer(size_t palette_size, const CellMoments &c, const int (*cut)[513]) {
    const size_t N = 512, max_k = 12;
    double ax[3];
    if (!c.axis(0, N, ax)) return {};
    bool error = false;
    size_t kmax = palette_size < max_k ? palette_size : max_k;
    std::vector<size_t> result = {0, N};
    for (size_t k = 2; k <= kmax; k++) {
        if (gq_should_terminate(result, ax, c, error)) break;
        result.assign(k + 1, 0);
        size_t t = N;
        for (size_t j = k - 1; j >= 1; j--) { t = (size_t)cut[j + 1][t]; result[j] = t; }
        result[0] = 0; result[k] = N;
    }
    return result;
}

// --------------------------------------------------------------------------------------------
// std::mt19937 + the prefix of faiss' rand_perm (faiss/utils/random.cpp:35-51,184-194)
// --------------------------------------------------------------------------------------------
struct MT19937 {
    uint32_t mt[624]; int idx;
    explicit MT19937(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    static inline uint32_t twist(uint32_t a, uint32_t b, uint32_t c) {
        const uint32_t y = (a & 0x80000000U) | (b & 0x7fffffffU);
        return c ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
    }
    uint32_t next() {
        if (idx >= 624) {                                       // word i sees the old i + 1 and i + 397, or the new i - 227
            int i = 0;
            for (; i < 624 - 397; i++) mt[i] = twist(mt[i], mt[i + 1], mt[i + 397]);
            for (; i < 623; i++) mt[i] = twist(mt[i], mt[i + 1], mt[i + 397 - 624]);
            mt[623] = twist(mt[623], mt[0], mt[396]);
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680U; y ^= (y << 15) & 0xefc60000U; y ^= (y >> 18);
        return y;
    }
};

// First `take` entries of rand_perm(n, seed): a forward Fisher-Yates prefix is final after step i (later steps swap positions
// >= their own index), so only `take` draws are needed.  Two passes: (1) all the draws j_i = i + mt() % (n - i); (2) the swaps,
// with positions below `take` in a flat array (read once each, in order) and the rest -- 98 % of the targets of a 262 144-entry
// prefix over 16.8 M -- in an open-addressing table of (position, value) pairs whose slots are PREFETCHED 24 steps ahead (the
// targets are known from pass 1; the table is 4-8 MB of random accesses, i.e. a cache miss each).  8.7x the one-pass form with two
// 64-bit tables allocated per call (25 -> 2.9 ms on a 2.1 GHz Xeon); same list, entry for entry (tests: faiss goldens).
struct PermScratch {
    struct Slot { uint32_t key, val; };
    std::vector<Slot> tab;
    std::vector<uint32_t> scr;
};
inline void rand_perm_prefix(size_t n, size_t take, uint32_t seed, int32_t *out, PermScratch &ws) {
    MT19937 rng(seed);
    if (take > n) take = n;
    if (n >= 0xFFFFFFFFull) {                                   // beyond 32-bit positions (the ABI stops at 40000^2 = 1.6e9): plain form
        size_t cap = 1;
        while (cap < 4 * take + 16) cap <<= 1;
        std::vector<uint64_t> keys(cap, UINT64_MAX), vals(cap);
        auto slot = [&](uint64_t k) {
            size_t s = (size_t)((k * 0x9E3779B97F4A7C15ULL) >> 20) & (cap - 1);
            while (keys[s] != UINT64_MAX && keys[s] != k) s = (s + 1) & (cap - 1);
            return s;
        };
        for (size_t i = 0; i < take; i++) {
            uint64_t j = i;
            if (i + 1 < n) j = i + (uint64_t)(rng.next() % (uint64_t)(n - i));
            size_t si = slot(i);
            uint64_t vi = keys[si] == UINT64_MAX ? (uint64_t)i : vals[si];
            size_t sj = slot(j);
            uint64_t vj = keys[sj] == UINT64_MAX ? j : vals[sj];
            out[i] = (int32_t)vj;
            keys[sj] = j; vals[sj] = vi;
        }
        return;
    }
    size_t cap = 1024;
    while (cap < 2 * take) cap <<= 1;
    if (ws.tab.size() < cap) ws.tab.resize(cap);
    std::memset(ws.tab.data(), 0xFF, cap * sizeof(PermScratch::Slot));
    if (ws.scr.size() < 2 * take) ws.scr.resize(2 * take);
    PermScratch::Slot *tab = ws.tab.data();
    uint32_t *js = ws.scr.data(), *low = ws.scr.data() + take;
    const uint32_t mask = (uint32_t)cap - 1, tk = (uint32_t)take, n32 = (uint32_t)n;
    for (uint32_t i = 0; i < tk; i++) {                         // random.cpp:184-194: i2 = i + rand_int(n - i), no draw for the last position
        js[i] = ((size_t)i + 1 < n) ? i + rng.next() % (n32 - i) : i;
        low[i] = i;
    }
    auto home = [&](uint32_t k) { return ((k * 0x9E3779B1u) >> 7) & mask; };
    constexpr uint32_t PF = 24;
    for (uint32_t i = 0; i < tk; i++) {
        if (i + PF < tk) __builtin_prefetch(&tab[home(js[i + PF])], 1, 1);
        const uint32_t j = js[i], vi = low[i];                  // position i is read here for the last time: later targets are >= their own step
        uint32_t vj;
        if (j < tk) { vj = low[j]; low[j] = vi; }
        else {
            uint32_t s = home(j);
            while (tab[s].key != 0xFFFFFFFFu && tab[s].key != j) s = (s + 1) & mask;
            vj = tab[s].key == 0xFFFFFFFFu ? j : tab[s].val;
            tab[s].key = j; tab[s].val = vi;
        }
        out[i] = (int32_t)vj;
    }
}
inline void rand_perm_prefix(size_t n, size_t take, uint32_t seed, int32_t *out) { PermScratch ws; rand_perm_prefix(n, take, seed, out, ws); }

}  // namespace hm
}  // namespace pamd
