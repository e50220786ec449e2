/*
 * patolette_oracle.c -- CPU restatement of big-nacho/patolette's quantisation path.
 *
 * TEST INFRASTRUCTURE ONLY (see patolette_oracle.h).  Plain C, no BLAS / LAPACK / FLANN /
 * faiss.  Build with -O2 -ffp-contract=off: the reference's C stages are compiled for
 * baseline x86-64 (no FMA contraction possible); the FMAs that the AVX2 faiss build
 * contains are written out explicitly with fmaf() below.
 *
 * Pinning status (details in oracle/README.md):
 *   - colour conversions: pinned against oracle/_ref/libref_color.so (the reference's own
 *     lib/src/color + lib/src/array sources compiled as they lie) -- bit-exact.
 *   - 3x3 eigen-solve: the arithmetic lives in third-party LAPACK (OpenBLAS, version
 *     unpinned by the reference).  Pinned against the LAPACK that is in this image
 *     (scipy's OpenBLAS 0.3.28 dsyev) through tests/golden/eigen_*.npz.
 *   - KMeans: pinned against oracle/_ref/libref_faiss.so when that build is available.
 *   - GQ / LQ / NN-map / dither orchestration: the reference sources need <cblas.h> and
 *     <flann/flann.h>, absent from this image, so they cannot be built without stand-ins:
 *     PARITY UNPINNED by a reference build; restated line by line from the cited source.
 */
#include "patolette_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define ORC_DELTA 1e-16                 /* lib/include/math/misc.h:5 */
#define SQ(x) ((x) * (x))
#define BUCKETS 512                     /* quantize/global.c:22, local.c:15 */

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- test knobs (tests/tie_prover.py, tests/test_tie_prover.py) -----------------------------
 * g_sum_reversed: the reference's sequential f64 sums (matrix2D.c:200-233, pca.c:84-97, cluster.c:111-152,
 * local.c:118-134, cells.c:82-112) taken back to front instead of front to back -- another member of the set of
 * results that differ from the reference's only by the rounding of its sums (what an OpenBLAS dgemv kernel chosen
 * for another CPU, or the HIP path's order-free sums, produce).  g_fault: a deliberately WRONG decision rule, to
 * show that the prover tells a tie from a bug: 1 = the cut one bucket past the arg-max (local.c:171),
 * 2 = the greedy step takes the SECOND best cluster (local.c:277-307), 3 = the cut at the LAST maximum of the
 * objective instead of the first (a tie-set member: must NOT turn the prover red on its own). */
static int g_sum_reversed = 0, g_fault = 0;
void orc_set_sum_reversed(int on) { g_sum_reversed = on != 0; }
void orc_set_fault(int which) { g_fault = which; }
#define AT(ii, n) (g_sum_reversed ? (n) - 1 - (ii) : (ii))

static orc_SplitTraceHeader g_trace_hdr;
static orc_SplitRecord *g_trace = NULL;
static size_t g_trace_len = 0, g_trace_cap = 0;
static void trace_reset(void) { g_trace_len = 0; memset(&g_trace_hdr, 0, sizeof g_trace_hdr); }
static orc_SplitRecord *trace_push(void) {
    if (g_trace_len == g_trace_cap) {
        g_trace_cap = g_trace_cap ? 2 * g_trace_cap : 256;
        g_trace = (orc_SplitRecord *)realloc(g_trace, g_trace_cap * sizeof *g_trace);
    }
    orc_SplitRecord *r = &g_trace[g_trace_len++];
    memset(r, 0, sizeof *r);
    return r;
}
size_t orc_last_split_trace(orc_SplitTraceHeader *hdr, orc_SplitRecord *recs, size_t capacity) {
    if (hdr) *hdr = g_trace_hdr;
    if (recs) memcpy(recs, g_trace, sizeof *recs * (g_trace_len < capacity ? g_trace_len : capacity));
    return g_trace_len;
}

/* ======================================================================================
 * Synthetic inputs
 * ==================================================================================== */
uint64_t orc_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static double u01(uint64_t seed, uint64_t i) {
    return (double)(orc_splitmix64((seed << 32) + i) >> 11) * 0x1.0p-53;
}
void orc_fill_uniform(double *out, size_t n, uint64_t seed) {
    for (size_t i = 0; i < n; i++) out[i] = u01(seed, i);
}
void orc_fill_image(double *planar, size_t n, uint64_t s) {
    for (int p = 0; p < 3; p++) orc_fill_uniform(planar + (size_t)p * n, n, 1000 * s + (uint64_t)p);
}
void orc_fill_weights(double *w, size_t n, uint64_t s) {
    for (size_t i = 0; i < n; i++) w[i] = 1.0 + 3.0 * u01(1000 * s + 7, i);
}

/* ======================================================================================
 * Colour conversions -- lib/src/color/{eotf,sRGB,xyz,rec2020,ICtCp,CIELuv}.c
 * ==================================================================================== */
/* eotf.c:13-18 */
static const double Lp = 10000, m1 = 0.1593017578125, m2 = 78.84375;
static const double c1 = 0.8359375, c2 = 18.8515625, c3 = 18.6875;

static double eotf_ST2084(double v) {                       /* eotf.c:29-42 */
    double m1d = 1 / m1, m2d = 1 / m2;
    double V_p = pow(v, m2d);
    double n = fmax(0, V_p - c1);
    double L = pow((n / (c2 - c3 * V_p)), m1d);
    return Lp * L;
}
static double eotf_inverse_ST2084(double v) {               /* eotf.c:44-57 */
    double y_ = pow(v / Lp, m1);
    return pow((c1 + c2 * y_) / (1 + c3 * y_), m2);
}
static double gamma_decode(double c) {                      /* sRGB.c:70-89 */
    double r = (c <= 0.0404500) ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4);
    return fmin(fmax(r, 0.0), 1.0);
}
static double gamma_encode(double c) {                      /* sRGB.c:91-110 */
    double r = (c <= 0.0031308) ? c * 12.92 : 1.055 * pow(c, 1.0 / 2.4) - 0.055;
    return fmin(fmax(r, 0.0), 1.0);
}
static void srgb_to_xyz(double r, double g, double b, double *x, double *y, double *z) { /* xyz.c:14-40 */
    double R = gamma_decode(r), G = gamma_decode(g), B = gamma_decode(b);
    *x = R * 0.4124564 + G * 0.3575761 + B * 0.1804375;
    *y = R * 0.2126729 + G * 0.7151522 + B * 0.0721750;
    *z = R * 0.0193339 + G * 0.1191920 + B * 0.9503041;
}
static void rec2020_to_xyz(double r, double g, double b, double *x, double *y, double *z) { /* xyz.c:42-64 */
    *x = r * 0.63695351 + g * 0.14461919 + b * 0.16885585;
    *y = r * 0.26269834 + g * 0.67800877 + b * 0.0592929;
    *z = g * 0.02807314 + b * 1.06082723;
}
static void xyz_to_rec2020(double x, double y, double z, double *r, double *g, double *b) { /* rec2020.c:80-102 */
    *r = x * 1.71666343 + y * -0.35567332 + z * -0.25336809;
    *g = x * -0.66667384 + y * 1.61645574 + z * 0.0157683;
    *b = x * 0.01764248 + y * -0.04277698 + z * 0.94224328;
}
static void rec2020_to_ictcp(double r, double g, double b, double *I, double *Ct, double *Cp) { /* ICtCp.c:41-79 */
    double L = (r * 1688 + g * 2146 + b * 262) / 4096;
    double M = (r * 683 + g * 2951 + b * 462) / 4096;
    double S = (r * 99 + g * 309 + b * 3688) / 4096;
    double L_ = eotf_inverse_ST2084(L), M_ = eotf_inverse_ST2084(M), S_ = eotf_inverse_ST2084(S);
    *I = L_ * 0.5 + M_ * 0.5;
    *Ct = (L_ * 6610 - M_ * 13613 + S_ * 7003) / 4096;
    *Cp = (L_ * 17933 - M_ * 17390 - S_ * 543) / 4096;
    *Ct *= 0.5;
}
static void ictcp_to_rec2020(double I, double Ct, double Cp, double *r, double *g, double *b) { /* rec2020.c:32-69 */
    Ct *= 2;
    double L_ = I + 0.00860904 * Ct + 0.11102963 * Cp;
    double M_ = I - 0.00860904 * Ct - 0.11102963 * Cp;
    double S_ = I + 0.56003134 * Ct - 0.32062717 * Cp;
    double L = eotf_ST2084(L_), M = eotf_ST2084(M_), S = eotf_ST2084(S_);
    *r = L * 3.43660669 - M * 2.50645212 + S * 0.06984542;
    *g = -L * 0.79132956 + M * 1.98360045 - S * 0.1922709;
    *b = -L * 0.0259499 - M * 0.09891371 + S * 1.12486361;
}
/* CIELuv.c:19-25 */
static const double rwx = 0.95047, rwy = 1.0, rwz = 1.08883;
static const double kE = 216.0 / 24389.0, kK = 24389.0 / 27.0, kKE = 8.0;

static void xyz_to_cieluv(double x, double y, double z, double *L, double *u, double *v) { /* CIELuv.c:54-89 */
    double den = x + 15.0 * y + 3.0 * z;
    double up = (den > 0.0) ? ((4.0 * x) / (x + 15.0 * y + 3.0 * z)) : 0.0;
    double vp = (den > 0.0) ? ((9.0 * y) / (x + 15.0 * y + 3.0 * z)) : 0.0;
    double urp = (4.0 * rwx) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double vrp = (9.0 * rwy) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double yr = y / rwy;
    double L_ = (yr > kE) ? (116.0 * pow(yr, 1.0 / 3.0) - 16.0) : (kK * yr);
    *L = L_;
    *u = 13.0 * L_ * (up - urp);
    *v = 13.0 * L_ * (vp - vrp);
}
static void cieluv_to_xyz(double L, double u, double v, double *x, double *y, double *z) { /* CIELuv.c:100-164 */
    double y_ = (L > kKE) ? pow((L + 16.0) / 116.0, 3.0) : (L / kK);
    double u0 = (4.0 * rwx) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double v0 = (9.0 * rwy) / (rwx + 15.0 * rwy + 3.0 * rwz);
    double a, a_den = u + 13.0 * L * u0;
    if (!a_den) a = 0; else a = (((52.0 * L) / a_den) - 1.0) / 3.0;
    double b = -5.0 * y_;
    double c = -1.0 / 3.0;
    double d, d_den = v + 13.0 * L * v0;
    if (!d_den) d = 0; else d = y_ * (((39.0 * L) / d_den) - 5.0);
    double x_, x_den = a - c;
    if (!x_den) x_ = 0; else x_ = (d - b) / x_den;
    double z_ = x_ * a + b;
    *x = x_; *y = y_; *z = z_;
}

#define PLANES(m, n) double *p0 = (m), *p1 = (m) + (n), *p2 = (m) + 2 * (n)

void orc_srgb_to_ictcp(double *m, size_t n) {               /* ICtCp.c:120-146 -> :81-109 */
    PLANES(m, n);
    for (size_t i = 0; i < n; i++) {
        double x, y, z, r, g, b;
        srgb_to_xyz(p0[i], p1[i], p2[i], &x, &y, &z);       /* rec2020.c:104-126 */
        xyz_to_rec2020(x, y, z, &r, &g, &b);
        rec2020_to_ictcp(r, g, b, &p0[i], &p1[i], &p2[i]);
    }
}
void orc_srgb_to_cieluv(double *m, size_t n) {              /* CIELuv.c:166-197 */
    PLANES(m, n);
    for (size_t i = 0; i < n; i++) {
        double r = gamma_decode(p0[i]), g = gamma_decode(p1[i]), b = gamma_decode(p2[i]);
        double x = r * 0.4124564 + g * 0.3575761 + b * 0.1804375;
        double y = r * 0.2126729 + g * 0.7151522 + b * 0.0721750;
        double z = r * 0.0193339 + g * 0.1191920 + b * 0.9503041;
        xyz_to_cieluv(x, y, z, &p0[i], &p1[i], &p2[i]);
    }
}
void orc_ictcp_to_rec2020(double *m, size_t n) {            /* rec2020.c:128-148 */
    PLANES(m, n);
    for (size_t i = 0; i < n; i++) ictcp_to_rec2020(p0[i], p1[i], p2[i], &p0[i], &p1[i], &p2[i]);
}
void orc_cieluv_to_rec2020(double *m, size_t n) {           /* rec2020.c:150-173 */
    PLANES(m, n);
    for (size_t i = 0; i < n; i++) {
        double x, y, z;
        cieluv_to_xyz(p0[i], p1[i], p2[i], &x, &y, &z);
        xyz_to_rec2020(x, y, z, &p0[i], &p1[i], &p2[i]);
    }
}
void orc_srgb_to_rec2020(double *m, size_t n) {             /* rec2020.c:175-195 */
    PLANES(m, n);
    for (size_t i = 0; i < n; i++) {
        double x, y, z;
        srgb_to_xyz(p0[i], p1[i], p2[i], &x, &y, &z);
        xyz_to_rec2020(x, y, z, &p0[i], &p1[i], &p2[i]);
    }
}
void orc_rec2020_to_srgb(double *m, size_t n) {             /* sRGB.c:112-132 -> :32-59 */
    PLANES(m, n);
    for (size_t i = 0; i < n; i++) {
        double x, y, z;
        rec2020_to_xyz(p0[i], p1[i], p2[i], &x, &y, &z);
        double r = x * 3.2404542 - y * 1.5371385 - z * 0.4985314;
        double g = -x * 0.9692660 + y * 1.8760108 + z * 0.0415560;
        double b = x * 0.0556434 - y * 0.2040259 + z * 1.0572252;
        p0[i] = gamma_encode(r); p1[i] = gamma_encode(g); p2[i] = gamma_encode(b);
    }
}

/* ======================================================================================
 * 3x3 symmetric eigen-solve: LAPACK dsyev('V','L') specialised to n = 3
 * (math/eigen.c:83-140 only *calls* dsyev_; the algorithm is third-party LAPACK >= 3.10 as
 *  bundled by OpenBLAS 0.3.28: dsytd2('L') -> dorgtr('L') -> dsteqr('V'), SURVEY.md App. B.)
 * ==================================================================================== */
static const double LA_EPS = 0x1.0p-53;                     /* dlamch('E') */
static const double LA_SAFMIN = 2.2250738585072014e-308;    /* dlamch('S') */

static double la_sign(double a, double b) { return copysign(fabs(a), b); }
static double dlapy2(double x, double y) {
    double xa = fabs(x), ya = fabs(y);
    double w = xa > ya ? xa : ya, z = xa > ya ? ya : xa;
    if (z == 0.0 || w > 1.79769313486231571e308) return w;
    return w * sqrt(1.0 + (z / w) * (z / w));
}
/* dlartg, LAPACK >= 3.10 (la_lartg.f90) */
static void dlartg(double f, double g, double *c, double *s, double *r) {
    const double safmin = LA_SAFMIN, safmax = 1.0 / LA_SAFMIN;
    const double rtmin = sqrt(safmin), rtmax = sqrt(safmax / 2);
    double f1 = fabs(f), g1 = fabs(g);
    if (g == 0.0) { *c = 1.0; *s = 0.0; *r = f; }
    else if (f == 0.0) { *c = 0.0; *s = la_sign(1.0, g); *r = g1; }
    else if (f1 > rtmin && f1 < rtmax && g1 > rtmin && g1 < rtmax) {
        double d = sqrt(f * f + g * g);
        *c = f1 / d; *r = la_sign(d, f); *s = g / *r;
    } else {
        double u = fmin(safmax, fmax(safmin, fmax(f1, g1)));
        double fs = f / u, gs = g / u;
        double d = sqrt(fs * fs + gs * gs);
        *c = fabs(fs) / d; *r = la_sign(d, f); *s = gs / *r; *r = *r * u;
    }
}
static void dlaev2(double a, double b, double c, double *rt1, double *rt2, double *cs1, double *sn1) {
    double sm = a + c, df = a - c, adf = fabs(df), tb = b + b, ab = fabs(tb);
    double acmx, acmn, rt;
    int sgn1, sgn2;
    if (fabs(a) > fabs(c)) { acmx = a; acmn = c; } else { acmx = c; acmn = a; }
    if (adf > ab) rt = adf * sqrt(1.0 + (ab / adf) * (ab / adf));
    else if (adf < ab) rt = ab * sqrt(1.0 + (adf / ab) * (adf / ab));
    else rt = ab * sqrt(2.0);
    if (sm < 0.0) { *rt1 = 0.5 * (sm - rt); sgn1 = -1; *rt2 = (acmx / *rt1) * acmn - (b / *rt1) * b; }
    else if (sm > 0.0) { *rt1 = 0.5 * (sm + rt); sgn1 = 1; *rt2 = (acmx / *rt1) * acmn - (b / *rt1) * b; }
    else { *rt1 = 0.5 * rt; *rt2 = -0.5 * rt; sgn1 = 1; }
    double cs;
    if (df >= 0.0) { cs = df + rt; sgn2 = 1; } else { cs = df - rt; sgn2 = -1; }
    double acs = fabs(cs);
    if (acs > ab) { double ct = -tb / cs; *sn1 = 1.0 / sqrt(1.0 + ct * ct); *cs1 = ct * *sn1; }
    else if (ab == 0.0) { *cs1 = 1.0; *sn1 = 0.0; }
    else { double tn = -cs / tb; *cs1 = 1.0 / sqrt(1.0 + tn * tn); *sn1 = tn * *cs1; }
    if (sgn1 == sgn2) { double tn = *cs1; *cs1 = -*sn1; *sn1 = tn; }
}
/* dlasr('R','V',direct): rotate columns (j, j+1), j = 0..mm-2, of the 3-row matrix z (col-major) */
static void dlasr_rv(int forward, int mm, const double *c, const double *s, double *z) {
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
/* dsteqr('V') for n = 3.  d[3], e[2], z 3x3 col-major.  Returns info. */
static int dsteqr3(double *d, double *e, double *z) {
    const int n = 3, maxit = 30;
    const double eps = LA_EPS, eps2 = eps * eps, safmin = LA_SAFMIN, safmax = 1.0 / safmin;
    const double ssfmax = sqrt(safmax) / 3.0, ssfmin = sqrt(safmin) / eps2;
    const int nmaxit = n * maxit;
    int jtot = 0, l1 = 1, nm1 = n - 1;
    double work[4];                                          /* work[0..1] = c, work[2..3] = s (1-based: WORK(I), WORK(N-1+I)) */
#define D(i) d[(i) - 1]
#define E(i) e[(i) - 1]
#define WC(i) work[(i) - 1]
#define WS(i) work[n - 1 + (i) - 1]
#define ZCOL(i) (z + ((i) - 1) * 3)
    for (;;) {
        if (l1 > n) break;
        if (l1 > 1) E(l1 - 1) = 0.0;
        int m;
        if (l1 <= nm1) {
            for (m = l1; m <= nm1; m++) {
                double tst = fabs(E(m));
                if (tst == 0.0) goto L30;
                if (tst <= (sqrt(fabs(D(m))) * sqrt(fabs(D(m + 1)))) * eps) { E(m) = 0.0; goto L30; }
            }
        }
        m = n;
    L30:;
        int l = l1, lsv = l, lend = m, lendsv = lend;
        l1 = m + 1;
        if (lend == l) continue;
        /* scale submatrix */
        double anorm = 0.0;
        for (int i = l; i <= lend; i++) { if (fabs(D(i)) > anorm) anorm = fabs(D(i)); }
        for (int i = l; i <= lend - 1; i++) { if (fabs(E(i)) > anorm) anorm = fabs(E(i)); }
        int iscale = 0;
        if (anorm == 0.0) continue;
        if (anorm > ssfmax) {
            iscale = 1;
            for (int i = l; i <= lend; i++) D(i) = D(i) * (ssfmax / anorm);       /* dlascl; mid-range never hits this */
            for (int i = l; i <= lend - 1; i++) E(i) = E(i) * (ssfmax / anorm);
        } else if (anorm < ssfmin) {
            iscale = 2;
            for (int i = l; i <= lend; i++) D(i) = D(i) * (ssfmin / anorm);
            for (int i = l; i <= lend - 1; i++) E(i) = E(i) * (ssfmin / anorm);
        }
        if (fabs(D(lend)) < fabs(D(l))) { lend = lsv; l = lendsv; }
        if (lend > l) {
            /* QL iteration */
            for (;;) {
                if (l != lend) {
                    for (m = l; m <= lend - 1; m++) {
                        double tst = fabs(E(m)) * fabs(E(m));
                        if (tst <= (eps2 * fabs(D(m))) * fabs(D(m + 1)) + safmin) goto L60;
                    }
                }
                m = lend;
            L60:
                if (m < lend) E(m) = 0.0;
                double p = D(l);
                if (m == l) {                                /* eigenvalue found */
                    D(l) = p; l = l + 1;
                    if (l <= lend) continue;
                    break;
                }
                if (m == l + 1) {
                    double rt1, rt2, c, s;
                    dlaev2(D(l), E(l), D(l + 1), &rt1, &rt2, &c, &s);
                    WC(l) = c; WS(l) = s;
                    dlasr_rv(0, 2, &WC(l), &WS(l), ZCOL(l));
                    D(l) = rt1; D(l + 1) = rt2; E(l) = 0.0;
                    l = l + 2;
                    if (l <= lend) continue;
                    break;
                }
                if (jtot == nmaxit) break;
                jtot++;
                double g = (D(l + 1) - p) / (2.0 * E(l));
                double r = dlapy2(g, 1.0);
                g = D(m) - p + (E(l) / (g + la_sign(r, g)));
                double s = 1.0, c = 1.0;
                p = 0.0;
                for (int i = m - 1; i >= l; i--) {
                    double f = s * E(i), b = c * E(i);
                    dlartg(g, f, &c, &s, &r);
                    if (i != m - 1) E(i + 1) = r;
                    g = D(i + 1) - p;
                    r = (D(i) - g) * s + 2.0 * c * b;
                    p = s * r;
                    D(i + 1) = g + p;
                    g = c * r - b;
                    WC(i) = c; WS(i) = -s;
                }
                dlasr_rv(0, m - l + 1, &WC(l), &WS(l), ZCOL(l));
                D(l) = D(l) - p;
                E(l) = g;
            }
        } else {
            /* QR iteration */
            for (;;) {
                if (l != lend) {
                    for (m = l; m >= lend + 1; m--) {
                        double tst = fabs(E(m - 1)) * fabs(E(m - 1));
                        if (tst <= (eps2 * fabs(D(m))) * fabs(D(m - 1)) + safmin) goto L110;
                    }
                }
                m = lend;
            L110:
                if (m > lend) E(m - 1) = 0.0;
                double p = D(l);
                if (m == l) {
                    D(l) = p; l = l - 1;
                    if (l >= lend) continue;
                    break;
                }
                if (m == l - 1) {
                    double rt1, rt2, c, s;
                    dlaev2(D(l - 1), E(l - 1), D(l), &rt1, &rt2, &c, &s);
                    WC(m) = c; WS(m) = s;
                    dlasr_rv(1, 2, &WC(m), &WS(m), ZCOL(l - 1));
                    D(l - 1) = rt1; D(l) = rt2; E(l - 1) = 0.0;
                    l = l - 2;
                    if (l >= lend) continue;
                    break;
                }
                if (jtot == nmaxit) break;
                jtot++;
                double g = (D(l - 1) - p) / (2.0 * E(l - 1));
                double r = dlapy2(g, 1.0);
                g = D(m) - p + (E(l - 1) / (g + la_sign(r, g)));
                double s = 1.0, c = 1.0;
                p = 0.0;
                for (int i = m; i <= l - 1; i++) {
                    double f = s * E(i), b = c * E(i);
                    dlartg(g, f, &c, &s, &r);
                    if (i != m) E(i - 1) = r;
                    g = D(i) - p;
                    r = (D(i + 1) - g) * s + 2.0 * c * b;
                    p = s * r;
                    D(i) = g + p;
                    g = c * r - b;
                    WC(i) = c; WS(i) = s;
                }
                dlasr_rv(1, l - m + 1, &WC(m), &WS(m), ZCOL(m));
                D(l) = D(l) - p;
                E(l - 1) = g;
            }
        }
        /* undo scaling */
        if (iscale == 1) {
            for (int i = lsv; i <= lendsv; i++) D(i) = D(i) * (anorm / ssfmax);
            for (int i = lsv; i <= lendsv - 1; i++) E(i) = E(i) * (anorm / ssfmax);
        } else if (iscale == 2) {
            for (int i = lsv; i <= lendsv; i++) D(i) = D(i) * (anorm / ssfmin);
            for (int i = lsv; i <= lendsv - 1; i++) E(i) = E(i) * (anorm / ssfmin);
        }
        if (jtot >= nmaxit) {
            int info = 0;
            for (int i = 1; i <= n - 1; i++) if (E(i) != 0.0) info++;
            return info;
        }
    }
    /* selection sort, ascending, swapping columns */
    for (int ii = 2; ii <= n; ii++) {
        int i = ii - 1, k = i;
        double p = D(i);
        for (int j = ii; j <= n; j++) if (D(j) < p) { k = j; p = D(j); }
        if (k != i) {
            D(k) = D(i); D(i) = p;
            for (int r = 0; r < 3; r++) { double t = ZCOL(i)[r]; ZCOL(i)[r] = ZCOL(k)[r]; ZCOL(k)[r] = t; }
        }
    }
    return 0;
#undef D
#undef E
#undef WC
#undef WS
#undef ZCOL
}

/* OpenBLAS level-1/2 kernels (dsymv, ddot, daxpy, dsyr2, dger) contract a*b+c into FMAs on every
 * x86-64 target with FMA3; bit k enables the FMA form of one kernel.  31 = all, which matches
 * scipy-OpenBLAS-0.3.28 dsyev bit for bit most often (tests/test_oracle_eigen.py). */
static int g_eig_fma = 31;
void orc_set_eigen_fma(int m) { g_eig_fma = m; }

int orc_eigen_sym3(double a[9], double w[3]) {
    /* lower triangle, column-major: A(i,j) = a[j*3+i] */
    double a11 = a[0], a21 = a[1], a31 = a[2], a22 = a[4], a32 = a[5], a33 = a[8];
    double d[3], e[2], tau = 0.0, v2 = 0.0;
    /* dsytd2('L'), i = 1: dlarfg(2, alpha = A21, x = [A31]) */
    {
        double alpha = a21, x = a31;
        double xnorm = fabs(x);                              /* dnrm2 of one element */
        if (xnorm == 0.0) {
            tau = 0.0;
        } else {
            double beta = -la_sign(dlapy2(alpha, xnorm), alpha);
            const double sfm = LA_SAFMIN / LA_EPS, rsfm = 1.0 / sfm;
            int knt = 0;
            if (fabs(beta) < sfm) {
                do { knt++; x *= rsfm; beta *= rsfm; alpha *= rsfm; } while (fabs(beta) < sfm && knt < 20);
                xnorm = fabs(x);
                beta = -la_sign(dlapy2(alpha, xnorm), alpha);
            }
            tau = (beta - alpha) / beta;
            x = x * (1.0 / (alpha - beta));                  /* dscal by the reciprocal */
            for (int j = 0; j < knt; j++) beta *= sfm;
            alpha = beta;
            v2 = x;
        }
        e[0] = alpha;
        if (tau != 0.0) {
            /* dsymv('L', 2, tau, B, v=(1,v2)) -> y ; B = [[a22,.],[a32,a33]] */
            double y1, y2, dot, al, w1, w2;
            if (g_eig_fma & 1) {
                double t1 = tau * 1.0;
                y1 = fma(t1, a22, 0.0); y2 = fma(t1, a32, 0.0);
                double t2 = fma(a32, v2, 0.0);
                y1 = fma(tau, t2, y1);
                t1 = tau * v2;
                y2 = fma(t1, a33, y2);
            } else {
                double t1 = tau * 1.0;
                y1 = t1 * a22; y2 = t1 * a32;
                double t2 = a32 * v2;
                y1 = y1 + tau * t2;
                t1 = tau * v2;
                y2 = y2 + t1 * a33;
            }
            /* alpha = -half*taui*ddot(y, v) ; daxpy */
            if (g_eig_fma & 2) dot = fma(y2, v2, y1 * 1.0); else dot = y1 * 1.0 + y2 * v2;
            al = -0.5 * tau * dot;
            if (g_eig_fma & 4) { w1 = fma(al, 1.0, y1); w2 = fma(al, v2, y2); } else { w1 = y1 + al * 1.0; w2 = y2 + al * v2; }
            /* dsyr2('L', 2, -1, v, w, B) */
            if (g_eig_fma & 8) {
                double temp1 = -1.0 * w1, temp2 = -1.0 * 1.0;
                a22 = fma(w1, temp2, fma(1.0, temp1, a22));
                a32 = fma(w2, temp2, fma(v2, temp1, a32));
                temp1 = -1.0 * w2; temp2 = -1.0 * v2;
                a33 = fma(w2, temp2, fma(v2, temp1, a33));
            } else {
                double temp1 = -1.0 * w1, temp2 = -1.0 * 1.0;
                a22 = a22 + 1.0 * temp1 + w1 * temp2;
                a32 = a32 + v2 * temp1 + w2 * temp2;
                temp1 = -1.0 * w2; temp2 = -1.0 * v2;
                a33 = a33 + v2 * temp1 + w2 * temp2;
            }
        }
        d[0] = a11;
    }
    /* i = 2: dlarfg(1, ...) -> tau = 0 */
    e[1] = a32; d[1] = a22; d[2] = a33;
    /* dorgtr('L') -> dorg2r on the trailing 2x2 */
    double z[9];
    z[0] = 1.0; z[1] = 0.0; z[2] = 0.0;
    z[3] = 0.0; z[6] = 0.0;
    if (tau != 0.0) {
        double wv = v2;                                      /* v^T C with C = (0,1) */
        double temp = -tau * wv;                             /* dger: alpha*y(j) */
        z[7] = 0.0 + 1.0 * temp;                             /* Q(2,3) */
        z[8] = (g_eig_fma & 16) ? fma(v2, temp, 1.0) : 1.0 + v2 * temp;   /* Q(3,3) */
    } else {
        z[7] = 0.0; z[8] = 1.0;
    }
    z[5] = -tau * v2;                                        /* dscal(-tau) on v2 -> Q(3,2) */
    z[4] = 1.0 - tau;                                        /* Q(2,2) */
    int info = dsteqr3(d, e, z);
    w[0] = d[0]; w[1] = d[1]; w[2] = d[2];
    memcpy(a, z, sizeof z);
    return info;
}

/* ======================================================================================
 * PCA -- math/pca.c, array/matrix2D.c
 * ==================================================================================== */
/* matrix2D.c:200-233: weighted column mean, sequential sums */
static void vector_mean(const double *c, const double *w, size_t n, double mean[3]) {
    for (int j = 0; j < 3; j++) {
        const double *col = c + (size_t)j * n;
        double acc = 0;
        for (size_t ii = 0; ii < n; ii++) {
            size_t i = AT(ii, n);
            double wi = w == NULL ? 1 : w[i];
            double v = col[i] * wi;
            acc += v;
        }
        mean[j] = acc;
    }
    double s;
    if (w == NULL) s = 1 / (double)n;
    else { double ws = 0; for (size_t ii = 0; ii < n; ii++) ws += w[AT(ii, n)]; s = 1 / ws; }
    for (int j = 0; j < 3; j++) mean[j] *= s;
}

/* pca.c:62-101 + :122-149.  vcov_out (optional) receives the covariance before the solve. */
int orc_pca_axis(const double *c, const double *w, size_t n, double axis[3], double vcov_out[9]) {
    double mean[3];
    vector_mean(c, w, n, mean);
    double *cen = (double *)malloc(sizeof(double) * 3 * (n ? n : 1));
    for (int j = 0; j < 3; j++)
        for (size_t i = 0; i < n; i++) cen[(size_t)j * n + i] = c[(size_t)j * n + i] - mean[j];
    double w_sum;
    if (w == NULL) w_sum = (double)n;
    else { w_sum = 0; for (size_t ii = 0; ii < n; ii++) w_sum += w[AT(ii, n)]; }
    double vcov[9];
    for (int j = 0; j < 3; j++) {
        for (int k = 0; k < 3; k++) {
            double value = 0;
            const double *cj = cen + (size_t)j * n, *ck = cen + (size_t)k * n;
            for (size_t ii = 0; ii < n; ii++) {
                size_t i = AT(ii, n);
                double wi = w == NULL ? 1 : w[i];
                value += wi * cj[i] * ck[i];
            }
            vcov[k * 3 + j] = value / w_sum;                 /* index(vcov, j, k) = data[k*3 + j] */
        }
    }
    free(cen);
    if (vcov_out) memcpy(vcov_out, vcov, sizeof vcov);
    double evals[3];
    if (orc_eigen_sym3(vcov, evals) != 0) return -1;
    axis[0] = vcov[6]; axis[1] = vcov[7]; axis[2] = vcov[8];  /* column 2, pca.c:136-138 */
    return 0;
}

/* ======================================================================================
 * Bucket sort along an axis -- quantize/sort.c:12-91
 * ==================================================================================== */
static int g_sort_degenerate;                                /* the last orc_axis_sort took sort.c:61-79's round-robin rule */
void orc_axis_sort(const double *c, size_t n, const double axis[3], size_t bucket_count, size_t *map) {
    g_sort_degenerate = 0;
    const double *p0 = c, *p1 = c + n, *p2 = c + 2 * n;
    double *dots = (double *)malloc(sizeof(double) * (n ? n : 1));
    dots[0] = 0.0;                                           /* n == 0: nothing to sort, keep the reads below defined */
    /* cblas_dgemv(ColMajor, NoTrans): y = A x.  Evaluation order inside OpenBLAS is not
     * specified; SURVEY.md 7(4): three different orders gave identical buckets. */
    for (size_t i = 0; i < n; i++) dots[i] = (p0[i] * axis[0] + p1[i] * axis[1]) + p2[i] * axis[2];
    double min_dot = dots[0], max_dot = dots[0];             /* vector.c:26-46 strict compares */
    for (size_t i = 0; i < n; i++) { if (dots[i] < min_dot) min_dot = dots[i]; }
    for (size_t i = 0; i < n; i++) { if (dots[i] > max_dot) max_dot = dots[i]; }
    if (max_dot - min_dot < ORC_DELTA) {                     /* sort.c:61-79 */
        g_sort_degenerate = 1;
        size_t j = 0;
        for (size_t i = 0; i < n; i++) {
            map[i] = j;
            if (j >= bucket_count - 1) j = 0; else j++;
        }
        free(dots);
        return;
    }
    double s = 1 / (max_dot - min_dot);
    for (size_t i = 0; i < n; i++) {
        double ratio = (dots[i] - min_dot) * s;
        size_t bucket = (size_t)((double)bucket_count * ratio);
        map[i] = bucket < bucket_count - 1 ? bucket : bucket_count - 1;
    }
    free(dots);
}

/* ======================================================================================
 * Cell moments -- quantize/cells.c
 * ==================================================================================== */
typedef struct {
    uint64_t w0[BUCKETS + 1];
    double w1[3][BUCKETS + 1];
    double w2[BUCKETS + 1];
    double wrs[3][3][BUCKETS + 1];                           /* [r][s], r <= s used */
} Cells;

static void cells_preprocess(const double *c, size_t n, const size_t *bucket_map, Cells *k) { /* cells.c:53-139 */
    memset(k, 0, sizeof *k);
    const double *p[3] = {c, c + n, c + 2 * n};
    for (size_t ii = 0; ii < n; ii++) {
        size_t i = AT(ii, n);
        size_t j = bucket_map[i] + 1;
        double cx = p[0][i], cy = p[1][i], cz = p[2][i];
        k->w0[j] += 1;
        k->w1[0][j] += cx; k->w1[1][j] += cy; k->w1[2][j] += cz;
        k->w2[j] += (SQ(cx) + SQ(cy) + SQ(cz));
    }
    for (size_t ii = 0; ii < n; ii++) {
        size_t i = AT(ii, n);
        size_t j = bucket_map[i] + 1;
        for (int s = 0; s < 3; s++)
            for (int r = 0; r <= s; r++) k->wrs[r][s][j] += p[r][i] * p[s][i];
    }
    for (size_t i = 1; i <= BUCKETS; i++) { k->w0[i] += k->w0[i - 1]; k->w2[i] += k->w2[i - 1]; }
    for (size_t i = 1; i <= BUCKETS; i++) for (int r = 0; r < 3; r++) k->w1[r][i] += k->w1[r][i - 1];
    for (size_t i = 1; i <= BUCKETS; i++)
        for (int s = 0; s < 3; s++)
            for (int r = 0; r <= s; r++) k->wrs[r][s][i] += k->wrs[r][s][i - 1];
}
static double cell_distortion(size_t a, size_t b, const Cells *k) {            /* cells.c:141-182 */
    uint64_t w0a = k->w0[a], w0b = k->w0[b];
    if (w0a == w0b) return 0;
    return (k->w2[b] - k->w2[a] -
            (SQ(k->w1[0][b] - k->w1[0][a]) + SQ(k->w1[1][b] - k->w1[1][a]) + SQ(k->w1[2][b] - k->w1[2][a])) /
                (double)(w0b - w0a));
}
static double cell_eval_vcov(size_t a, size_t b, int r, int s, const Cells *k) { /* cells.c:184-223 */
    uint64_t w0a = k->w0[a], w0b = k->w0[b];
    if (w0a == w0b) return 0;
    return ((k->wrs[r][s][b] - k->wrs[r][s][a]) / (double)(w0b - w0a) -
            (k->w1[r][b] - k->w1[r][a]) * (k->w1[s][b] - k->w1[s][a]) / SQ((double)(w0b - w0a)));
}
static int cell_pca_axis(size_t a, size_t b, const Cells *k, double axis[3]) {  /* cells.c:225-278 */
    double v[9];
    memset(v, 0, sizeof v);
    for (int s = 0; s < 3; s++)
        for (int r = 0; r <= s; r++) v[s * 3 + r] = cell_eval_vcov(a, b, r, s, k);  /* index(vcov, r, s) */
    v[0 * 3 + 2] = v[2 * 3 + 0];
    v[0 * 3 + 1] = v[1 * 3 + 0];
    v[1 * 3 + 2] = v[2 * 3 + 1];
    double ev[3];
    if (orc_eigen_sym3(v, ev) != 0) return -1;
    axis[0] = v[6]; axis[1] = v[7]; axis[2] = v[8];
    return 0;
}
static double vec_norm3(const double a[3]) {                                   /* vector.c snorm/norm: pow(x,2) sums */
    double s = 0;
    for (int i = 0; i < 3; i++) s += pow(a[i], 2);
    return sqrt(s);
}
static double cell_bias(size_t a, size_t b, const double axis[3], const Cells *k) { /* cells.c:280-328 */
    double ca[3];
    if (cell_pca_axis(a, b, k, ca) != 0) return -1;
    double norms = vec_norm3(axis) * vec_norm3(ca);
    if (norms < ORC_DELTA) return 0;
    double dot = (ca[0] * axis[0] + ca[1] * axis[1] + ca[2] * axis[2]);
    double cs = dot / norms;
    return fmin(1, fabs(cs));
}

/* ======================================================================================
 * Global principal quantiser -- quantize/global.c
 * ==================================================================================== */
static const size_t GQ_MAX_K = 12;                           /* global.c:19 */
static const double GQ_BIAS_THR = 0.1, GQ_CELL_BIAS_THR = 0.9; /* global.c:20-21 */

static int gq_should_terminate(const size_t *q, size_t qlen, const double axis[3], const Cells *k, int *error) { /* global.c:99-187 */
    double distortion = 0;
    for (size_t j = 0; j + 1 < qlen; j++) distortion += cell_distortion(q[j], q[j + 1], k);
    if (distortion < ORC_DELTA) return 1;
    double bias = 0;
    for (size_t i = 0; i + 1 < qlen; i++) {
        double cd = cell_distortion(q[i], q[i + 1], k);
        double cb = cell_bias(q[i], q[i + 1], axis, k);
        if (cb < 0) { *error = 1; return 1; }
        if (cb < GQ_CELL_BIAS_THR) continue;
        bias += (cd / distortion) * cb;
    }
    return bias < GQ_BIAS_THR;
}

/* global.c:189-298.  Returns quantiser cuts q[0..*qlen-1] (malloc'd) or NULL. */
static size_t *gq_principal_quantizer(size_t palette_size, const Cells *cache, size_t *qlen) {
    int error = 0;
    const size_t N = BUCKETS;
    double axis[3];
    if (cell_pca_axis(0, N, cache, axis) != 0) return NULL;
    double *E = (double *)calloc(N + 1, sizeof(double));
    double *E__ = (double *)calloc(N + 1, sizeof(double));
    /* L is only ever read/written at rows <= min(max_k, K) -- keep those rows (global.c:226-240) */
    size_t kmax = palette_size < GQ_MAX_K ? palette_size : GQ_MAX_K;
    size_t lcols = (palette_size > N ? palette_size : N) + 1;
    double *L = (double *)calloc((kmax + 2) * lcols, sizeof(double));
#define LIDX(k, n) L[(k) * lcols + (n)]
    for (size_t i = 1; i <= N; i++) E[i] = cell_distortion(0, i, cache);
    for (size_t i = 1; i <= palette_size && i <= kmax + 1; i++) LIDX(i, i) = (double)i;
    size_t *result = (size_t *)calloc(kmax + 2, sizeof(size_t));
    size_t rlen = 2;
    result[0] = 0; result[1] = N;                            /* l_chain(L, 1, N) */
    for (size_t k = 2; k <= kmax; k++) {
        if (gq_should_terminate(result, rlen, axis, cache, &error)) break;
        memcpy(E__, E, sizeof(double) * (N + 1));
        for (size_t n = k + 1; n <= N; n++) {
            double cut = (double)(n - 1);
            double e = E__[n - 1];
            for (size_t t = n - 2; t >= k - 1; t--) {
                double c = (E__[t] + cell_distortion(t, n, cache));
                if (c < e) { cut = (double)t; e = c; }
                if (t == 0) break;
            }
            LIDX(k, n) = cut;
            E[n] = e;
        }
        /* l_chain(L, k, N), global.c:72-97 */
        size_t t = N;
        for (size_t j = k - 1; j >= 1; j--) { t = (size_t)LIDX(j + 1, t); result[j] = t; }
        result[0] = 0; result[k] = N;
        rlen = k + 1;
    }
#undef LIDX
    free(E); free(E__); free(L);
    *qlen = rlen;
    return result;
}

/* ======================================================================================
 * Colour clusters -- quantize/cluster.c
 * ==================================================================================== */
typedef struct Cluster {
    size_t *idx; size_t n;
    const double *dataset; const double *dataset_w; size_t N;
    double *colors;    /* gathered planar (n,3), cluster.c:219-238 */
    double *weights;   /* gathered, cluster.c:80-109 */
    int has_center; double center[3];
    double distortion; /* -1 = not computed */
    int has_axis; double axis[3];
    double vcov[9];    /* the matrix handed to the eigen-solver (pca.c:62-101), for the split trace */
} Cluster;

static Cluster *cluster_init(const double *dataset, const double *dw, size_t N, size_t *idx, size_t n) {
    Cluster *c = (Cluster *)calloc(1, sizeof *c);
    c->idx = idx; c->n = n; c->dataset = dataset; c->dataset_w = dw; c->N = N;
    c->distortion = -1.0;
    return c;
}
static void cluster_destroy(Cluster *c) {
    if (!c) return;
    free(c->idx); free(c->colors); free(c->weights); free(c);
}
static const double *cluster_colors(Cluster *c) {
    if (c->colors) return c->colors;
    c->colors = (double *)malloc(sizeof(double) * 3 * (c->n ? c->n : 1));
    for (size_t i = 0; i < c->n; i++) {
        size_t r = c->idx[i];
        for (int j = 0; j < 3; j++) c->colors[(size_t)j * c->n + i] = c->dataset[(size_t)j * c->N + r];
    }
    return c->colors;
}
static const double *cluster_weights(Cluster *c) {
    if (!c->dataset_w) return NULL;
    if (c->weights) return c->weights;
    c->weights = (double *)malloc(sizeof(double) * (c->n ? c->n : 1));
    for (size_t i = 0; i < c->n; i++) c->weights[i] = c->dataset_w[c->idx[i]];
    return c->weights;
}
static const double *cluster_center(Cluster *c) {             /* cluster.c:171-189 */
    if (c->has_center) return c->center;
    vector_mean(cluster_colors(c), cluster_weights(c), c->n, c->center);
    c->has_center = 1;
    return c->center;
}
static double cluster_distortion(Cluster *c) {                /* cluster.c:111-152 */
    if (c->distortion != -1.0) return c->distortion;
    const double *col = cluster_colors(c), *w = cluster_weights(c), *ctr = cluster_center(c);
    double x = ctr[0], y = ctr[1], z = ctr[2], d = 0;
    size_t n = c->n;
    for (size_t ii = 0; ii < n; ii++) {
        size_t i = AT(ii, n);
        double weight = w == NULL ? 1 : w[i];
        double cx = col[i], cy = col[n + i], cz = col[2 * n + i];
        double distance = (SQ(cx - x) + SQ(cy - y) + SQ(cz - z)) * weight;
        d += distance;
    }
    c->distortion = d;
    return d;
}
static const double *cluster_axis(Cluster *c) {               /* cluster.c:191-217 */
    if (c->has_axis) return c->axis;
    if (orc_pca_axis(cluster_colors(c), cluster_weights(c), c->n, c->axis, c->vcov) != 0) return NULL;
    c->has_axis = 1;
    return c->axis;
}

/* global.c:300-377 */
static Cluster **gq_color_clusters(const double *colors, const double *w, size_t N, const size_t *q, size_t qlen,
                                   const size_t *bucket_map, size_t *count_out) {
    size_t count = qlen - 1;
    size_t *sizes = (size_t *)calloc(count, sizeof(size_t));
    size_t lut[BUCKETS];
    for (size_t b = 0; b < BUCKETS; b++) {
        lut[b] = 0;
        for (size_t j = 0; j < count; j++) if (b + 1 <= q[j + 1]) { lut[b] = j; break; }
    }
    for (size_t i = 0; i < N; i++) sizes[lut[bucket_map[i]]] += 1;
    size_t **ids = (size_t **)calloc(count, sizeof(size_t *));
    for (size_t j = 0; j < count; j++) ids[j] = (size_t *)malloc(sizeof(size_t) * (sizes[j] ? sizes[j] : 1));
    size_t *piv = (size_t *)calloc(count, sizeof(size_t));
    for (size_t i = 0; i < N; i++) { size_t j = lut[bucket_map[i]]; ids[j][piv[j]++] = i; }
    Cluster **cl = (Cluster **)calloc(count, sizeof(Cluster *));
    for (size_t j = 0; j < count; j++) cl[j] = cluster_init(colors, w, N, ids[j], sizes[j]);
    free(sizes); free(ids); free(piv);
    *count_out = count;
    return cl;
}

/* global.c:388-443 */
static Cluster **gq_quantize(const double *colors, const double *w, size_t N, size_t K, size_t *count_out) {
    double axis[3], gq_vcov[9];
    if (orc_pca_axis(colors, NULL, N, axis, gq_vcov) != 0) return NULL;   /* UNWEIGHTED, global.c:407 */
    size_t *bucket_map = (size_t *)malloc(sizeof(size_t) * N);
    orc_axis_sort(colors, N, axis, BUCKETS, bucket_map);
    Cells *cache = (Cells *)malloc(sizeof(Cells));
    cells_preprocess(colors, N, bucket_map, cache);
    size_t qlen = 0;
    size_t *q = gq_principal_quantizer(K, cache, &qlen);
    trace_reset();
    for (int j = 0; j < 3; j++) g_trace_hdr.gq_axis[j] = axis[j];
    g_trace_hdr.gq_cov6[0] = gq_vcov[0]; g_trace_hdr.gq_cov6[1] = gq_vcov[1]; g_trace_hdr.gq_cov6[2] = gq_vcov[2];
    g_trace_hdr.gq_cov6[3] = gq_vcov[4]; g_trace_hdr.gq_cov6[4] = gq_vcov[5]; g_trace_hdr.gq_cov6[5] = gq_vcov[8];
    if (q) { g_trace_hdr.n_base = (int32_t)(qlen - 1); for (size_t j = 0; j < qlen && j < 14; j++) g_trace_hdr.gq_cuts[j] = q[j]; }
    Cluster **res = NULL;
    if (q) res = gq_color_clusters(colors, w, N, q, qlen, bucket_map, count_out);
    free(bucket_map); free(cache); free(q);
    return res;
}

/* ======================================================================================
 * Local quantiser -- quantize/local.c
 * ==================================================================================== */
typedef struct { Cluster *left, *right; size_t split; int degenerate; } Pair;
static size_t g_split_evals, g_split_px;

static size_t lq_optimal_bucket(Cluster *c, const size_t *bucket_map) {     /* local.c:102-177 */
    const double *col = cluster_colors(c), *w = cluster_weights(c);
    size_t n = c->n;
    static size_t sizes[BUCKETS];
    static double sums[3][BUCKETS];
    memset(sizes, 0, sizeof sizes); memset(sums, 0, sizeof sums);
    for (size_t ii = 0; ii < n; ii++) {
        size_t i = AT(ii, n);
        size_t b = bucket_map[i];
        double cx = col[i], cy = col[n + i], cz = col[2 * n + i];
        double weight = w == NULL ? 1 : w[i];
        sums[0][b] += cx * weight; sums[1][b] += cy * weight; sums[2][b] += cz * weight;
        sizes[b] += weight;                                  /* size_t += double: truncates per add (local.c:133) */
    }
    for (size_t i = 1; i < BUCKETS; i++) for (int j = 0; j < 3; j++) sums[j][i] += sums[j][i - 1];
    for (size_t i = 1; i < BUCKETS; i++) sizes[i] += sizes[i - 1];
    size_t loc = 0;
    double best = 0;
    for (size_t i = 0; i < BUCKETS; i++) {
        double obj = 0;
        for (int j = 0; j < 3; j++) {
            double csl = sums[j][i], csr = sums[j][BUCKETS - 1] - csl;
            double sl = (double)sizes[i], sr = (double)sizes[BUCKETS - 1] - sl;
            double v = 0;
            if (sl != 0) v += SQ(csl) / sl;
            if (sr != 0) v += SQ(csr) / sr;
            obj += v;
        }
        if (i == 0) { best = obj; loc = 0; }                 /* vector.c:26-46: first max wins */
        else if (obj > best || (g_fault == 3 && obj == best)) { best = obj; loc = i; }
    }
    if (g_fault == 1) {                                      /* test knob: a WRONG cut -- one more occupied bucket goes left */
        size_t b = loc + 1;
        while (b < BUCKETS - 1 && sizes[b] == sizes[loc]) b++;
        if (b < BUCKETS - 1 && sizes[b] < sizes[BUCKETS - 1]) loc = b;
    }
    return loc;
}

static Pair *lq_split_cluster(Cluster *c) {                   /* local.c:179-254 */
    size_t n = c->n;
    if (n <= 1) return NULL;
    const double *col = cluster_colors(c);
    const double *axis = cluster_axis(c);
    if (!axis) return NULL;
    g_split_evals++; g_split_px += n;
    size_t *bucket_map = (size_t *)malloc(sizeof(size_t) * n);
    orc_axis_sort(col, n, axis, BUCKETS, bucket_map);
    const int degenerate = g_sort_degenerate;
    size_t split = lq_optimal_bucket(c, bucket_map);
    size_t ls = 0, rs = 0;
    for (size_t i = 0; i < n; i++) { if (bucket_map[i] <= split) ls++; else rs++; }
    size_t *li = (size_t *)malloc(sizeof(size_t) * (ls ? ls : 1));
    size_t *ri = (size_t *)malloc(sizeof(size_t) * (rs ? rs : 1));
    size_t pl = 0, pr = 0;
    for (size_t i = 0; i < n; i++) {
        if (bucket_map[i] <= split) li[pl++] = c->idx[i]; else ri[pr++] = c->idx[i];
    }
    free(bucket_map);
    Pair *p = (Pair *)malloc(sizeof *p);
    p->left = cluster_init(c->dataset, c->dataset_w, c->N, li, ls);
    p->right = cluster_init(c->dataset, c->dataset_w, c->N, ri, rs);
    p->split = split; p->degenerate = degenerate;
    return p;
}
static double lq_split_benefit(Cluster *c, Pair *ch) {        /* local.c:256-275 */
    if (!ch) return 0;
    double d = cluster_distortion(c), dl = cluster_distortion(ch->left), dr = cluster_distortion(ch->right);
    return d - (dl + dr);
}

/* local.c:318-404.  clusters[0..count) are consumed; returns array of *out_len clusters. */
static Cluster **lq_quantize(Cluster **clusters, size_t count, size_t K, size_t *out_len) {
    if (count >= K) { *out_len = count; g_trace_hdr.n_clusters = (int32_t)count; return clusters; }
    Cluster **result = (Cluster **)calloc(K, sizeof(Cluster *));
    memcpy(result, clusters, sizeof(Cluster *) * count);
    Pair **children = (Pair **)calloc(K, sizeof(Pair *));
    for (size_t i = 0; i < count; i++) children[i] = lq_split_cluster(clusters[i]);
    size_t len = K;
    double *benefits = (double *)malloc(sizeof(double) * K);
    for (size_t i = count; i < K; i++) {
        size_t best = 0;
        for (size_t j = 0; j < i; j++) benefits[j] = children[j] ? lq_split_benefit(result[j], children[j]) : 0;
        double bv = benefits[0];
        for (size_t j = 0; j < i; j++) if (benefits[j] > bv) { bv = benefits[j]; best = j; }
        if (g_fault == 2) {                                   /* test knob: a WRONG greedy step (the second best, if it splits at all) */
            size_t second = best; double sv = -1;
            for (size_t j = 0; j < i; j++) if (j != best && benefits[j] > sv) { sv = benefits[j]; second = j; }
            if (sv >= ORC_DELTA && sv < bv) best = second;
        }
        double benefit = lq_split_benefit(result[best], children[best]);
        if (benefit < ORC_DELTA) { len = i; g_trace_hdr.stopped_early = 1; break; }
        Cluster *left = children[best]->left, *right = children[best]->right;
        Cluster *old = result[best];
        {
            orc_SplitRecord *r = trace_push();
            r->row = (int32_t)best; r->new_row = (int32_t)i; r->split = (int32_t)children[best]->split;
            r->degenerate = children[best]->degenerate;
            r->n = old->n; r->n_left = left->n; r->n_right = right->n;
            const double *ow = cluster_weights(old);
            double sw = 0;
            if (ow) for (size_t t = 0; t < old->n; t++) sw += ow[t]; else sw = (double)old->n;
            r->sw = sw;
            for (int j = 0; j < 3; j++) r->axis[j] = old->axis[j];
            r->cov6[0] = old->vcov[0]; r->cov6[1] = old->vcov[1]; r->cov6[2] = old->vcov[2];
            r->cov6[3] = old->vcov[4]; r->cov6[4] = old->vcov[5]; r->cov6[5] = old->vcov[8];
            r->dist = cluster_distortion(old); r->dist_left = cluster_distortion(left); r->dist_right = cluster_distortion(right);
            r->benefit = benefit;
        }
        free(children[best]); children[best] = NULL;
        result[i] = left;
        result[best] = right;
        children[i] = lq_split_cluster(left);
        children[best] = lq_split_cluster(right);
        cluster_destroy(old);
    }
    free(benefits);
    for (size_t i = 0; i < K; i++) {
        if (children[i]) { cluster_destroy(children[i]->left); cluster_destroy(children[i]->right); free(children[i]); }
    }
    free(children);
    free(clusters);
    *out_len = len;
    g_trace_hdr.n_clusters = (int32_t)len; g_trace_hdr.n_records = (int32_t)g_trace_len;
    return result;
}

int orc_quantize_clusters(const double *colors, const double *weights, size_t n, size_t K,
                          double *centers, size_t *n_clusters, uint32_t *cluster_of,
                          size_t *n_base, size_t *split_evals, size_t *split_px) {
    size_t count = 0;
    g_split_evals = 0; g_split_px = 0;
    Cluster **gq = gq_quantize(colors, weights, n, K, &count);
    if (!gq) return -1;
    if (n_base) *n_base = count;
    size_t len = 0;
    Cluster **cl = lq_quantize(gq, count, K, &len);
    for (size_t i = 0; i < len; i++) {                        /* create.c:11-33 */
        const double *ctr = cluster_center(cl[i]);
        for (int j = 0; j < 3; j++) centers[(size_t)j * K + i] = ctr[j];
        if (cluster_of) for (size_t t = 0; t < cl[i]->n; t++) cluster_of[cl[i]->idx[t]] = (uint32_t)i;
    }
    *n_clusters = len;
    if (split_evals) *split_evals = g_split_evals;
    if (split_px) *split_px = g_split_px;
    for (size_t i = 0; i < len; i++) cluster_destroy(cl[i]);
    free(cl);
    return 0;
}

/* ======================================================================================
 * KMeans refinement -- palette/refine.c + patched faiss 1.10 (AVX2 flavour)
 * ==================================================================================== */
/* std::mt19937 (utils/random.cpp:35 RandomGenerator wraps it) */
typedef struct { uint32_t mt[624]; int idx; } MT;
static void mt_seed(MT *m, uint32_t s) {
    m->mt[0] = s;
    for (int i = 1; i < 624; i++) m->mt[i] = 1812433253U * (m->mt[i - 1] ^ (m->mt[i - 1] >> 30)) + (uint32_t)i;
    m->idx = 624;
}
static uint32_t mt_next(MT *m) {
    if (m->idx >= 624) {
        for (int i = 0; i < 624; i++) {
            uint32_t y = (m->mt[i] & 0x80000000U) | (m->mt[(i + 1) % 624] & 0x7fffffffU);
            uint32_t v = m->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1U) v ^= 0x9908b0dfU;
            m->mt[i] = v;
        }
        m->idx = 0;
    }
    uint32_t y = m->mt[m->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

/* First `take` entries of faiss rand_perm(perm, n, seed) (random.cpp:184-194): a forward
 * Fisher-Yates prefix is final after step i, so only `take` draws are needed; the touched
 * entries of the virtual identity permutation live in an open-addressing map. */
void orc_kmeans_subsample_indices(size_t n, size_t take, int64_t seed, int32_t *out) {
    MT rng; mt_seed(&rng, (uint32_t)seed);
    size_t cap = 1; while (cap < 4 * take + 16) cap <<= 1;
    uint64_t *keys = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    uint64_t *vals = (uint64_t *)malloc(sizeof(uint64_t) * cap);
    for (size_t i = 0; i < cap; i++) keys[i] = UINT64_MAX;
#define SLOT(k, s) do { s = (size_t)(((k) * 0x9E3779B97F4A7C15ULL) >> 20) & (cap - 1); \
        while (keys[s] != UINT64_MAX && keys[s] != (k)) s = (s + 1) & (cap - 1); } while (0)
    for (size_t i = 0; i < take; i++) {
        uint64_t vi, vj, j = i;
        size_t si, sj;
        if (i + 1 < n) j = i + (uint64_t)(mt_next(&rng) % (uint64_t)(n - i));
        SLOT((uint64_t)i, si); vi = keys[si] == UINT64_MAX ? (uint64_t)i : vals[si];
        SLOT(j, sj); vj = keys[sj] == UINT64_MAX ? j : vals[sj];
        out[i] = (int32_t)vj;                                /* perm[i] after swap */
        keys[sj] = j; vals[sj] = vi;                         /* perm[j] = old perm[i] */
        (void)si;
    }
#undef SLOT
    free(keys); free(vals);
}

/* l2_sqr<3> as GCC 11 -O3 -mfma contracts it in the AVX2 faiss build:
 * fma(v2,v2, fma(v0,v0, v1*v1)) */
static float l2sqr3(const float *v) { return fmaf(v[2], v[2], fmaf(v[0], v[0], v[1] * v[1])); }

/* Threads for the three loops the reference's dependencies run on several cores: faiss' search and compute_centroids
 * (OpenMP, distances.cpp:807-823 / Clustering.cpp:152-201) and FLANN's nearest-neighbour search (cores = 0,
 * nearest.c:189-199).  Everything else in the reference is single-threaded (LQ, GQ, the conversions, the dither).  Every
 * item of these loops is independent (compute_centroids splits the CENTROIDS over threads and each thread scans all
 * samples in order), so the results do not depend on the thread count.  Default 1; bench.py's cpu_baseline sets it. */
static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int orc_get_threads(void) { return g_threads; }

/* IndexFlatL2::search k=1 -> exhaustive_L2sqr_fused_cmax<3,6,1> (simdlib_based.cpp:59-277) */
void orc_kmeans_assign(const float *x, size_t nx, const float *cent, size_t k, int64_t *assign, float *dis) {
    float *yn = (float *)malloc(sizeof(float) * (k ? k : 1));
    for (size_t j = 0; j < k; j++) yn[j] = l2sqr3(cent + 3 * j);
    size_t ny_p = (k / 8) * 8;
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (size_t i = 0; i < nx; i++) {
        const float *xi = x + 3 * i;
        float m2x0 = -2 * xi[0], m2x1 = -2 * xi[1], m2x2 = -2 * xi[2];
        float xn = l2sqr3(xi);
        float lane_d[8]; uint32_t lane_i[8];
        for (int l = 0; l < 8; l++) { lane_d[l] = 3.402823466e+38F - xn; lane_i[l] = 0; }
        for (size_t j = 0; j < ny_p; j++) {
            const float *y = cent + 3 * j;
            float dp = m2x0 * y[0];
            dp = fmaf(m2x1, y[1], dp);
            dp = fmaf(m2x2, y[2], dp);
            dp = dp + yn[j];
            int l = (int)(j & 7);
            if (dp < lane_d[l]) { lane_d[l] = dp; lane_i[l] = (uint32_t)j; }
        }
        float cur_d = 3.402823466e+38F; uint32_t cur_i = 0xFFFFFFFFu;
        for (int l = 0; l < 8; l++) {
            float cand = lane_d[l] + xn;
            if (cand < 0) cand = 0;
            if (cur_d > cand) { cur_d = cand; cur_i = lane_i[l]; }
            else if (cur_d == cand && cur_i > lane_i[l]) cur_i = lane_i[l];
        }
        for (size_t j0 = ny_p; j0 < k; j0++) {               /* leftovers, simdlib_based.cpp:201-216 */
            const float *y = cent + 3 * j0;
            float dp = fmaf(xi[2], y[2], fmaf(xi[1], y[1], xi[0] * y[0]));
            float d = xn + yn[j0] - 2 * dp;
            if (d < 0) d = 0;
            if (cur_d > d) { cur_d = d; cur_i = (uint32_t)j0; }
        }
        assign[i] = (int64_t)cur_i; if (dis) dis[i] = cur_d;
    }
    free(yn);
}

/* compute_centroids (Clustering.cpp:135-204): per centroid, samples in order; weighted adds are
 * FMA-contracted in the AVX2 build (c = fma(x, w, c)), unweighted are plain adds. */
void orc_kmeans_update(const float *x, const float *w, size_t nx, const int64_t *assign,
                       float *cent, size_t k, float *hassign) {
    memset(cent, 0, sizeof(float) * 3 * k);
    memset(hassign, 0, sizeof(float) * k);
    /* Clustering.cpp:152-201: thread r of nt owns centroids [k r / nt, k (r + 1) / nt) and scans ALL samples in order */
    /* (every extra thread re-scans all samples: beyond 16 the scan costs more than the sums it shares out) */
    const int nt = g_threads > 16 ? 16 : (g_threads > 1 ? g_threads : 1);
#pragma omp parallel for schedule(static, 1) num_threads(nt) if (nt > 1)
    for (int r = 0; r < nt; r++) {
        const int64_t c0 = (int64_t)((k * (size_t)r) / (size_t)nt), c1 = (int64_t)((k * (size_t)(r + 1)) / (size_t)nt);
        for (size_t i = 0; i < nx; i++) {
            int64_t ci = assign[i];
            if (ci < c0 || ci >= c1) continue;
            float *c = cent + 3 * ci;
            const float *xi = x + 3 * i;
            if (w) {
                float wi = w[i];
                hassign[ci] += wi;
                for (int j = 0; j < 3; j++) c[j] = fmaf(xi[j], wi, c[j]);
            } else {
                hassign[ci] += 1.0f;
                for (int j = 0; j < 3; j++) c[j] += xi[j];
            }
        }
    }
    for (size_t ci = 0; ci < k; ci++) {
        if (hassign[ci] == 0) continue;
        float norm = 1 / hassign[ci];
        for (int j = 0; j < 3; j++) cent[3 * ci + j] *= norm;
    }
}

/* split_clusters (Clustering.cpp:216-263) */
int orc_kmeans_split_clusters(size_t k, size_t n, float *hassign, float *cent) {
    const size_t d = 3;
    size_t nsplit = 0;
    MT rng; mt_seed(&rng, 1234u);
    for (size_t ci = 0; ci < k; ci++) {
        if (hassign[ci] == 0) {
            size_t cj;
            for (cj = 0; 1; cj = (cj + 1) % k) {
                float p = (float)((hassign[cj] - 1.0) / (float)(n - k));
                float r = (float)mt_next(&rng) / 4294967296.0f;  /* mt() / float(mt.max()) */
                if (r < p) break;
            }
            memcpy(cent + ci * d, cent + cj * d, sizeof(float) * d);
            for (size_t j = 0; j < d; j++) {
                if (j % 2 == 0) {
                    cent[ci * d + j] = (float)(cent[ci * d + j] * (1 + (1 / 1024.)));
                    cent[cj * d + j] = (float)(cent[cj * d + j] * (1 - (1 / 1024.)));
                } else {
                    cent[ci * d + j] = (float)(cent[ci * d + j] * (1 - (1 / 1024.)));
                    cent[cj * d + j] = (float)(cent[cj * d + j] * (1 + (1 / 1024.)));
                }
            }
            hassign[ci] = hassign[cj] / 2;
            hassign[cj] -= hassign[ci];
            nsplit++;
        }
    }
    return (int)nsplit;
}

void orc_kmeans_refine(const double *colors, const double *weights, size_t n,
                       double *centers_io, size_t k, int niter, size_t max_samples) {
    /* refine.c:102-163: f64 -> f32 interleaved copies */
    float *cent = (float *)malloc(sizeof(float) * 3 * k);
    for (size_t i = 0; i < k; i++) for (int j = 0; j < 3; j++) cent[3 * i + j] = (float)centers_io[(size_t)j * k + i];
    const size_t min_samples = 256 * 256;                    /* refine.c:21 */
    size_t ms = max_samples > min_samples ? max_samples : min_samples;
    int mppc = (int)(ms / k);                                /* refine.c:87 */
    size_t nx = n;
    float *x = NULL, *w = NULL;
    int ok = (n >= k);                                       /* Clustering.cpp:272-278 throws otherwise -> centres unchanged */
    if (ok) {
        /* NaN / Inf scan (Clustering.cpp:295-304) throws -> centres unchanged */
        for (size_t i = 0; i < 3 * n && ok; i++) if (!isfinite((float)colors[i])) ok = 0;
    }
    if (ok) {
        if (n > k * (size_t)mppc) {                          /* Clustering.cpp:311-319 */
            nx = k * (size_t)mppc;
            int32_t *perm = (int32_t *)malloc(sizeof(int32_t) * nx);
            orc_kmeans_subsample_indices(n, nx, 1234, perm);
            x = (float *)malloc(sizeof(float) * 3 * nx);
            for (size_t i = 0; i < nx; i++) for (int j = 0; j < 3; j++) x[3 * i + j] = (float)colors[(size_t)j * n + (size_t)perm[i]];
            if (weights) { w = (float *)malloc(sizeof(float) * nx); for (size_t i = 0; i < nx; i++) w[i] = (float)weights[perm[i]]; }
            free(perm);
        } else {
            x = (float *)malloc(sizeof(float) * 3 * nx);
            for (size_t i = 0; i < nx; i++) for (int j = 0; j < 3; j++) x[3 * i + j] = (float)colors[(size_t)j * n + i];
            if (weights) { w = (float *)malloc(sizeof(float) * nx); for (size_t i = 0; i < nx; i++) w[i] = (float)weights[i]; }
        }
        if (nx == k) {                                       /* Clustering.cpp:331-352: copy training set (x_in = un-subsampled) */
            for (size_t i = 0; i < k; i++) for (int j = 0; j < 3; j++) cent[3 * i + j] = (float)colors[(size_t)j * n + i];
        } else {
            int64_t *assign = (int64_t *)malloc(sizeof(int64_t) * nx);
            float *hassign = (float *)malloc(sizeof(float) * k);
            for (int it = 0; it < niter; it++) {
                orc_kmeans_assign(x, nx, cent, k, assign, NULL);
                orc_kmeans_update(x, w, nx, assign, cent, k, hassign);
                orc_kmeans_split_clusters(k, nx, hassign, cent);
            }
            free(assign); free(hassign);
        }
    }
    for (size_t i = 0; i < k; i++) for (int j = 0; j < 3; j++) centers_io[(size_t)j * k + i] = (double)cent[3 * i + j];
    free(cent); free(x); free(w);
}

/* ======================================================================================
 * Exact NN palette map -- palette/nearest.c:150-209 (FLANN L2, eps = 0: dist = ((d0^2)+d1^2)+d2^2)
 * ==================================================================================== */
void orc_nn_map(const double *colors, size_t n, const double *palette, size_t k, size_t *map) {
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (size_t i = 0; i < n; i++) {
        double x = colors[i], y = colors[n + i], z = colors[2 * n + i];
        size_t best = 0; double bd = INFINITY;
        for (size_t j = 0; j < k; j++) {
            double d0 = x - palette[j], d1 = y - palette[k + j], d2 = z - palette[2 * k + j];
            double d = (d0 * d0 + d1 * d1) + d2 * d2;
            if (d < bd) { bd = d; best = j; }
        }
        map[i] = best;
    }
}

/* ======================================================================================
 * Riemersma dither -- dither/riemersma.c
 * ==================================================================================== */
enum { DIR_NONE, DIR_UP, DIR_LEFT, DIR_RIGHT, DIR_DOWN };
typedef struct {
    size_t x, y, width, height;
    /* mode 0: dither; mode 1: record order */
    int mode; uint64_t *order; size_t norder; size_t max_visits, visits;
    const double *img; size_t n;                             /* planar (n,3) */
    const double *pal; size_t k; double *pal_w;              /* palette planar (k,3); pal_w interleaved scaled */
    double q[16][3]; double qw[16];
    size_t *map;
} Dither;

static const double R_weight = 0.51254268114958, G_weight = 0.8234075540095561, B_weight = 0.2435159132377184; /* riemersma.c:38-42 */

static void dither_pixel(Dither *s) {                         /* riemersma.c:275-341 */
    double eR = 0, eG = 0, eB = 0;
    for (int i = 0; i < 16; i++) { double w = s->qw[i]; eR += s->q[i][0] * w; eG += s->q[i][1] * w; eB += s->q[i][2] * w; }
    size_t p = s->y * s->width + s->x;
    double R = s->img[p], G = s->img[s->n + p], B = s->img[2 * s->n + p];
    double cR = R + eR, cG = G + eG, cB = B + eB;
    double qx = R_weight * cR, qy = G_weight * cG, qz = B_weight * cB;
    size_t best = 0; double bd = INFINITY;
    for (size_t j = 0; j < s->k; j++) {
        double d0 = qx - s->pal_w[3 * j], d1 = qy - s->pal_w[3 * j + 1], d2 = qz - s->pal_w[3 * j + 2];
        double d = (d0 * d0 + d1 * d1) + d2 * d2;
        if (d < bd) { bd = d; best = j; }
    }
    s->map[p] = best;
    for (int i = 0; i < 15; i++) { s->q[i][0] = s->q[i + 1][0]; s->q[i][1] = s->q[i + 1][1]; s->q[i][2] = s->q[i + 1][2]; }
    s->q[15][0] = R - s->pal[best]; s->q[15][1] = G - s->pal[s->k + best]; s->q[15][2] = B - s->pal[2 * s->k + best];
}
static void d_move(Dither *s, int dir) {                      /* riemersma.c:146-174 */
    if (s->x < s->width && s->y < s->height) {
        if (s->max_visits == 0 || s->visits < s->max_visits) {
            if (s->mode == 0) dither_pixel(s);
            else s->order[s->norder++] = (uint64_t)(s->y * s->width + s->x);
        }
        s->visits++;
    }
    switch (dir) {
        case DIR_LEFT: s->x--; break;
        case DIR_RIGHT: s->x++; break;
        case DIR_UP: s->y--; break;
        case DIR_DOWN: s->y++; break;
        default: break;
    }
}
static void d_traverse(Dither *s, int level, int dir) {       /* riemersma.c:176-257 */
    if (level == 1) {
        switch (dir) {
            case DIR_LEFT: d_move(s, DIR_RIGHT); d_move(s, DIR_DOWN); d_move(s, DIR_LEFT); break;
            case DIR_RIGHT: d_move(s, DIR_LEFT); d_move(s, DIR_UP); d_move(s, DIR_RIGHT); break;
            case DIR_UP: d_move(s, DIR_DOWN); d_move(s, DIR_RIGHT); d_move(s, DIR_UP); break;
            case DIR_DOWN: d_move(s, DIR_UP); d_move(s, DIR_LEFT); d_move(s, DIR_DOWN); break;
            default: break;
        }
    } else {
        switch (dir) {
            case DIR_LEFT:
                d_traverse(s, level - 1, DIR_UP); d_move(s, DIR_RIGHT);
                d_traverse(s, level - 1, DIR_LEFT); d_move(s, DIR_DOWN);
                d_traverse(s, level - 1, DIR_LEFT); d_move(s, DIR_LEFT);
                d_traverse(s, level - 1, DIR_DOWN); break;
            case DIR_RIGHT:
                d_traverse(s, level - 1, DIR_DOWN); d_move(s, DIR_LEFT);
                d_traverse(s, level - 1, DIR_RIGHT); d_move(s, DIR_UP);
                d_traverse(s, level - 1, DIR_RIGHT); d_move(s, DIR_RIGHT);
                d_traverse(s, level - 1, DIR_UP); break;
            case DIR_UP:
                d_traverse(s, level - 1, DIR_LEFT); d_move(s, DIR_DOWN);
                d_traverse(s, level - 1, DIR_UP); d_move(s, DIR_RIGHT);
                d_traverse(s, level - 1, DIR_UP); d_move(s, DIR_UP);
                d_traverse(s, level - 1, DIR_RIGHT); break;
            case DIR_DOWN:
                d_traverse(s, level - 1, DIR_RIGHT); d_move(s, DIR_UP);
                d_traverse(s, level - 1, DIR_DOWN); d_move(s, DIR_LEFT);
                d_traverse(s, level - 1, DIR_DOWN); d_move(s, DIR_DOWN);
                d_traverse(s, level - 1, DIR_LEFT); break;
            default: break;
        }
    }
}
static int d_level(size_t width, size_t height) {             /* riemersma.c:124-144 */
    int level = 0;
    size_t mx = width > height ? width : height, value = mx;
    while (value > 1) { value >>= 1; level++; }
    if (((size_t)1 << level) < mx) level++;
    return level;
}
static void dither_run(const double *colors, size_t width, size_t height, const double *palette, size_t k, size_t *map, size_t max_visits);
void orc_dither_riemersma(const double *colors, size_t width, size_t height,
                          const double *palette, size_t k, size_t *map) {
    dither_run(colors, width, height, palette, k, map, 0);
}
/* test helper: only the first max_visits in-image steps of the walk are dithered (the chain is causal,
 * so they equal the first max_visits steps of the full run) */
void orc_dither_riemersma_prefix(const double *colors, size_t width, size_t height,
                                 const double *palette, size_t k, size_t *map, size_t max_visits) {
    dither_run(colors, width, height, palette, k, map, max_visits);
}
static void dither_run(const double *colors, size_t width, size_t height, const double *palette, size_t k, size_t *map, size_t max_visits) {
    Dither s; memset(&s, 0, sizeof s);
    s.max_visits = max_visits;
    s.width = width; s.height = height; s.img = colors; s.n = width * height; s.pal = palette; s.k = k; s.map = map;
    /* riemersma.c:360-373 */
    double m = exp(log(16.0) / (16.0 - 1)), v = 1;
    for (int i = 0; i < 16; i++) { s.qw[i] = v / 16.0; v *= m; }
    /* palette index data scaled by (float)-cast weights (riemersma.c:419-425, nearest.c:32-61) */
    double fx = (double)(float)R_weight, fy = (double)(float)G_weight, fz = (double)(float)B_weight;
    s.pal_w = (double *)malloc(sizeof(double) * 3 * (k ? k : 1));
    for (size_t j = 0; j < k; j++) { s.pal_w[3 * j] = palette[j] * fx; s.pal_w[3 * j + 1] = palette[k + j] * fy; s.pal_w[3 * j + 2] = palette[2 * k + j] * fz; }
    int level = d_level(width, height);
    if (level > 0) { d_traverse(&s, level, DIR_UP); d_move(&s, DIR_NONE); }
    free(s.pal_w);
}
size_t orc_hilbert_order(size_t width, size_t height, uint64_t *order) {
    Dither s; memset(&s, 0, sizeof s);
    s.width = width; s.height = height; s.mode = 1; s.order = order;
    int level = d_level(width, height);
    if (level > 0) { d_traverse(&s, level, DIR_UP); d_move(&s, DIR_NONE); }
    return s.norder;
}

/* ======================================================================================
 * patolette() -- lib/src/patolette.c:157-343
 * ==================================================================================== */
static const char *orc_messages[6] = {                        /* patolette.c:32-38 */
    "Quantization successful.", "Internal quantization error.", "Image dimensions should be greater than 0.",
    "Palette size should be greater than 0.", "Image dimensions are too big.", NULL};
const char *orc_exit_message(int exit_code) { return orc_messages[-1 * exit_code]; }

static double g_timings[6];
void orc_last_timings(double out[6]) { memcpy(out, g_timings, sizeof g_timings); }

/* patolette.c:246-336: everything behind the local quantiser -- optional KMeans, mapping or dithering with the colour-space
 * routing, palette write-out.  colors: the image in the quantisation space (converted in place further, as the reference
 * does); pal: planar (len,3) cluster centres, consumed. */
static void patolette_tail(size_t width, size_t height, double *colors, const double *weights, size_t K, const orc_Options *opt,
                           double *pal, size_t len, double *palette, size_t *palette_map) {
    const size_t N = width * height;
    double t = now_s();
    if (opt->kmeans_niter > 0) orc_kmeans_refine(colors, weights, N, pal, len, opt->kmeans_niter, opt->kmeans_max_samples);
    g_timings[3] = now_s() - t;

    t = now_s();
    if (!opt->palette_only) {
        if (opt->dither) {
            if (opt->color_space == ORC_CIELuv) { orc_cieluv_to_rec2020(colors, N); orc_cieluv_to_rec2020(pal, len); }
            else if (opt->color_space == ORC_ICtCp) { orc_ictcp_to_rec2020(colors, N); orc_ictcp_to_rec2020(pal, len); }
            else { orc_srgb_to_rec2020(colors, N); orc_srgb_to_rec2020(pal, len); }
            orc_dither_riemersma(colors, width, height, pal, len, palette_map);
            /* the reference also converts `colors` back to sRGB here (patolette.c:297): dead work, skipped */
            orc_rec2020_to_srgb(pal, len);
        } else {
            if (opt->color_space == ORC_CIELuv) {
                orc_cieluv_to_rec2020(colors, N); orc_cieluv_to_rec2020(pal, len);
                orc_rec2020_to_srgb(colors, N); orc_rec2020_to_srgb(pal, len);
                orc_srgb_to_ictcp(colors, N); orc_srgb_to_ictcp(pal, len);
            }
            orc_nn_map(colors, N, pal, len, palette_map);
            orc_ictcp_to_rec2020(pal, len);
            orc_rec2020_to_srgb(pal, len);
        }
    }
    g_timings[4] = now_s() - t;
    for (size_t j = 0; j < K * 3; j++) palette[j] = -1.0;    /* patolette.c:327-336 */
    for (int j = 0; j < 3; j++) for (size_t i = 0; i < len; i++) palette[K * (size_t)j + i] = pal[(size_t)j * len + i];
}

void orc_patolette(size_t width, size_t height, const double *data, const double *weight_data,
                   size_t K, const orc_Options *opt, double *palette, size_t *palette_map, int *exit_code) {
    *exit_code = 0;                                           /* patolette.c:61-95 */
    size_t N = width * height;
    if (N == 0) { *exit_code = -2; return; }
    if (K < 1) { *exit_code = -3; return; }
    if (N > (size_t)40000 * 40000) { *exit_code = -4; return; }   /* reference sets the code; validate then returns */
    double t0 = now_s(), t;
    memset(g_timings, 0, sizeof g_timings);
    double *colors = (double *)malloc(sizeof(double) * 3 * N);
    memcpy(colors, data, sizeof(double) * 3 * N);
    double *weights = NULL;
    if (weight_data) { weights = (double *)malloc(sizeof(double) * N); memcpy(weights, weight_data, sizeof(double) * N); }
    t = now_s();
    if (opt->color_space == ORC_CIELuv) orc_srgb_to_cieluv(colors, N);
    else if (opt->color_space == ORC_ICtCp) orc_srgb_to_ictcp(colors, N);
    g_timings[0] = now_s() - t;

    t = now_s();
    size_t count = 0;
    g_split_evals = 0; g_split_px = 0;
    Cluster **gq = gq_quantize(colors, weights, N, K, &count);
    g_timings[1] = now_s() - t;
    if (!gq) { *exit_code = -1; free(colors); free(weights); return; }
    t = now_s();
    size_t len = 0;
    Cluster **cl = lq_quantize(gq, count, K, &len);
    g_timings[2] = now_s() - t;

    double *pal = (double *)calloc(3 * (len ? len : 1), sizeof(double));   /* planar (len,3) */
    for (size_t i = 0; i < len; i++) { const double *c = cluster_center(cl[i]); for (int j = 0; j < 3; j++) pal[(size_t)j * len + i] = c[j]; }
    patolette_tail(width, height, colors, weights, K, opt, pal, len, palette, palette_map);
    for (size_t i = 0; i < len; i++) cluster_destroy(cl[i]);
    free(cl); free(pal); free(colors); free(weights);
    g_timings[5] = now_s() - t0;
    *exit_code = 0;
}

/* The reference's path from given cluster centres on (tests/tie_prover.py): `data` sRGB as for orc_patolette, `centers` planar
 * (len,3) in the quantisation space -- what PALETTE_create (create.c:11-33) would hand to patolette.c:246.  Lets a test replay
 * KMeans + mapping / dithering + write-out behind a clustering that differs from the oracle's only by proven ties. */
void orc_patolette_from_centers(size_t width, size_t height, const double *data, const double *weight_data, size_t K,
                                const orc_Options *opt, const double *centers, size_t len, double *palette, size_t *palette_map) {
    const size_t N = width * height;
    double *colors = (double *)malloc(sizeof(double) * 3 * N);
    memcpy(colors, data, sizeof(double) * 3 * N);
    if (opt->color_space == ORC_CIELuv) orc_srgb_to_cieluv(colors, N);
    else if (opt->color_space == ORC_ICtCp) orc_srgb_to_ictcp(colors, N);
    double *pal = (double *)malloc(sizeof(double) * 3 * (len ? len : 1));
    memcpy(pal, centers, sizeof(double) * 3 * len);
    patolette_tail(width, height, colors, weight_data, K, opt, pal, len, palette, palette_map);
    free(pal); free(colors);
}

/* ======================================================================================
 * Minimum-barrier-distance scans of the saliency map -- src/patolette/patolette.pyx:54-201
 * (the Cython loops of the Python binding; everything around them is numpy/scipy and is
 * restated in oracle/saliency.py).  All f32, row-major (rows, cols).
 * ==================================================================================== */
static inline float mbd_max(float a, float b) { return a > b ? a : b; }
static inline float mbd_min(float a, float b) { return a < b ? a : b; }

/* One visit (patolette.pyx:75-98 / :124-147): candidate barriers through the already-scanned
 * vertical neighbour (b1) and horizontal neighbour (b2); keep d when it is <= both, else b1
 * when it is the strictly-better-than-d, not-worse-than-b2 one, else b2. */
static inline void mbd_visit(const float *img, float *L, float *U, float *D, size_t at, size_t vert, size_t horz) {
    float ix = img[at], d = D[at];
    float u1 = U[vert], l1 = L[vert], u2 = U[horz], l2 = L[horz];
    float b1 = mbd_max(u1, ix) - mbd_min(l1, ix);
    float b2 = mbd_max(u2, ix) - mbd_min(l2, ix);
    if (d <= b1 && d <= b2) return;
    if (b1 < d && b1 <= b2) { D[at] = b1; U[at] = mbd_max(u1, ix); L[at] = mbd_min(l1, ix); }
    else { D[at] = b2; U[at] = mbd_max(u2, ix); L[at] = mbd_min(l2, ix); }
}

/* forward scan: rows 1..rows-2, cols 1..cols-2, neighbours above / left (patolette.pyx:72-103) */
static void mbd_scan_forward(size_t rows, size_t cols, const float *img, float *L, float *U, float *D) {
    for (size_t x = 1; x + 1 < rows; x++)
        for (size_t y = 1; y + 1 < cols; y++)
            mbd_visit(img, L, U, D, x * cols + y, (x - 1) * cols + y, x * cols + y - 1);
}
/* inverse scan: rows rows-2..2, cols cols-2..2 (the loops stop at `> 1`), neighbours below / right
 * (patolette.pyx:121-152) */
static void mbd_scan_inverse(size_t rows, size_t cols, const float *img, float *L, float *U, float *D) {
    for (size_t x = rows - 2; x > 1; x--)
        for (size_t y = cols - 2; y > 1; y--)
            mbd_visit(img, L, U, D, x * cols + y, (x + 1) * cols + y, x * cols + y + 1);
}

/* mbd(img, iter) (patolette.pyx:156-201): L = U = img, D = +inf inside / 0 on the one-pixel frame;
 * pass p runs the forward scan when p is odd, the inverse scan when p is even.  Returns 0, or -1
 * for images with rows <= 3 or cols <= 3 (the reference returns None). */
int orc_mbd(size_t rows, size_t cols, const float *img, int iters, float *D) {
    if (rows <= 3 || cols <= 3) return -1;
    size_t n = rows * cols;
    float *L = (float *)malloc(sizeof(float) * n), *U = (float *)malloc(sizeof(float) * n);
    memcpy(L, img, sizeof(float) * n);
    memcpy(U, img, sizeof(float) * n);
    for (size_t x = 0; x < rows; x++)
        for (size_t y = 0; y < cols; y++)
            D[x * cols + y] = (x == 0 || y == 0 || x == rows - 1 || y == cols - 1) ? 0.0f : INFINITY;
    for (int p = 0; p < iters; p++) {
        if (p % 2 == 1) mbd_scan_forward(rows, cols, img, L, U, D);
        else mbd_scan_inverse(rows, cols, img, L, U, D);
    }
    free(L); free(U);
    return 0;
}
