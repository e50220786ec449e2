// saliency.hip -- saliency-derived pixel weights on gfx950.
//
// Replaces `get_weights` of the reference's Python binding (src/patolette/patolette.pyx:203-313):
//   mbd raster scans (:54-201, sequential Cython loops)  -> k_mbd_scan: strips of 64 rows, one wavefront
//       per strip, lanes = rows, skewed by one column per row so that a lane finds the state of the row
//       above in its neighbour's registers (DPP wave shift) and its own previous column in its own
//       registers; strips pipeline through per-strip progress flags in HBM.  f32 min/max/sub only:
//       bit-exact.
//   rgb2lab + border statistics + Mahalanobis contrasts + normalisations + centre prior + sigmoid
//       (:207-313, numpy/scipy/skimage) -> streaming kernels; sums are order-independent (devutil.h),
//       maxima are integer-keyed atomics.
// All row-major (rows = height, cols = width), like the reference's reshape (patolette.pyx:410).
#include "saliency.h"

#include <cmath>
#include <type_traits>

#include "devutil.h"

namespace pamd {

namespace {

constexpr int kE_lab = 8;            // |L|, |a|, |b| < 2^8 for anything rgb2lab produces from [0,1]^3
constexpr int kE_cov = 17;           // centred products < 2^17

struct Band { int r0, r1, c0, c1; };             // [r0,r1) x [c0,c1)
struct Bands { Band b[4]; };

// skimage.color.rgb2lab: sRGB companding, then XYZ (D65, 2 degree) and CIELAB
__device__ __forceinline__ double lab_compand(double c) { return (c > 0.04045) ? pamd_pow((c + 0.055) / 1.055, 2.4) : c / 12.92; }
__device__ __forceinline__ void linear_to_lab(const double v[3], double lab[3]) {
    double x = (v[0] * 0.412453 + v[1] * 0.357580) + v[2] * 0.180423;
    double y = (v[0] * 0.212671 + v[1] * 0.715160) + v[2] * 0.072169;
    double z = (v[0] * 0.019334 + v[1] * 0.119193) + v[2] * 0.950227;
    x = x / 0.95047; y = y / 1.0; z = z / 1.08883;
    x = (x > 0.008856) ? cbrt(x) : 7.787 * x + 16.0 / 116.0;
    y = (y > 0.008856) ? cbrt(y) : 7.787 * y + 16.0 / 116.0;
    z = (z > 0.008856) ? cbrt(z) : 7.787 * z + 16.0 / 116.0;
    lab[0] = (116.0 * y) - 16.0;
    lab[1] = 500.0 * (x - y);
    lab[2] = 200.0 * (y - z);
}
__device__ __forceinline__ void rgb_to_lab(const double c[3], double lab[3]) {
    const double v[3] = {lab_compand(c[0]), lab_compand(c[1]), lab_compand(c[2])};
    linear_to_lab(v, lab);
}

// State per pixel as one 16-byte record {img, D, U, L} (row-major): a visit reads and writes one record.
constexpr size_t kMbdPad = 256;      // records of padding on both sides of the state (lanes read past row ends)

// The scans work on a SKEWED copy of the state: record (r, c) lives at [group g][diagonal d][l], g = (r + 63) / 64,
// l = (r + 63) % 64, d = c + l + 128, so the 64 records a strip touches in one step -- rows 64 s + l, columns t - l: one
// anti-diagonal -- are 1 KB of consecutive memory instead of 64 different cache lines (the address unit, not HBM, bounded the
// row-major scan: 110 ns per step).  The backward scan's strips are aligned to the bottom of the image, so one of its steps
// covers at most two such runs.  Cells of the skewed array that correspond to no pixel are only ever read, their values
// discarded.
struct SkewGeom {
    int dn;                          // diagonals per group: cols + 320 (columns -128 .. cols + 128 of every row of the group)
    __host__ __device__ size_t idx(int r, int c) const {
        const int rr = r + 63;
        return ((size_t)(rr >> 6) * (size_t)dn + (size_t)(c + (rr & 63) + 128)) * 64 + (size_t)(rr & 63);
    }
};

// channel mean (patolette.pyx:204), minimum-barrier initial state (:160-170) and CIELAB (:213)
template <class SRC>
__global__ __launch_bounds__(256) void k_sal_prepare(SRC src, size_t n, int rows, int cols, float4 *__restrict__ sk, const SkewGeom geo,
                                                     double *__restrict__ lab) {
    constexpr bool kLut = std::is_same<SRC, SrcU8>::value;
    __shared__ double glut[kLut ? 256 : 1];                // companding of the 256 possible 8-bit values
    if constexpr (kLut) {
        for (int b = threadIdx.x; b < 256; b += blockDim.x) glut[b] = lab_compand((double)b / 255.0);
        __syncthreads();
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double c[3], o[3];
        src.load(i, c);
        const float m = (float)(((c[0] + c[1]) + c[2]) / 3.0);
        const int r = (int)(i / (size_t)cols), q = (int)(i - (size_t)r * cols);
        const bool frame = r == 0 || q == 0 || r == rows - 1 || q == cols - 1;
        sk[geo.idx(r, q)] = make_float4(m, frame ? 0.0f : INFINITY, m, m);      // {img, D, U, L}, straight into the scans' skewed layout
        if constexpr (kLut) {
            unsigned r8, g8, b8;
            src.load_bytes(i, r8, g8, b8);
            const double v[3] = {glut[r8], glut[g8], glut[b8]};
            linear_to_lab(v, o);
        } else {
            rgb_to_lab(c, o);
        }
        lab[i] = o[0]; lab[n + i] = o[1]; lab[2 * n + i] = o[2];
    }
}

__global__ void k_sal_init(SalDev *d) {
    const int t = threadIdx.x;
    for (int i = t; i < kStatSlots * 4 * 3 * 2; i += blockDim.x) (&d->sum[0][0][0][0])[i] = 0.0;
    for (int i = t; i < kStatSlots * 4 * 6 * 2; i += blockDim.x) (&d->cov[0][0][0][0])[i] = 0.0;
    for (int i = t; i < kSalMax * kStatSlots; i += blockDim.x) (&d->maxkey[0][0])[i] = f64_key(-INFINITY);
    if (t == 0) d->singular = 0;
}

// sums of Lab over the four border bands (np.mean(..., axis=(0,1)), patolette.pyx:221-224); blockIdx.y = band
template <bool CENTRED>
__global__ __launch_bounds__(256) void k_sal_band(const double *__restrict__ lab, size_t n, int cols, Bands bands, SalDev *d,
                                                  BinK kb) {
    __shared__ double sm[6 * 2 * 4];
    const Band b = bands.b[blockIdx.y];
    const size_t wd = (size_t)(b.c1 - b.c0), cnt = wd * (size_t)(b.r1 - b.r0);
    constexpr int NV = CENTRED ? 12 : 6;
    double acc[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) acc[k] = 0.0;
    double m[3] = {0, 0, 0};
    if (CENTRED) { m[0] = d->mean[blockIdx.y][0]; m[1] = d->mean[blockIdx.y][1]; m[2] = d->mean[blockIdx.y][2]; }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < cnt; q += stride) {
        const size_t rr = q / wd, cc = q - rr * wd;
        const size_t i = ((size_t)b.r0 + rr) * (size_t)cols + (size_t)b.c0 + cc;
        const double x = lab[i], y = lab[n + i], z = lab[2 * n + i];
        if (!CENTRED) {
            bin_add(x, kb, acc[0], acc[1]); bin_add(y, kb, acc[2], acc[3]); bin_add(z, kb, acc[4], acc[5]);
        } else {
            const double dx = x - m[0], dy = y - m[1], dz = z - m[2];
            bin_add(dx * dx, kb, acc[0], acc[1]); bin_add(dx * dy, kb, acc[2], acc[3]); bin_add(dx * dz, kb, acc[4], acc[5]);
            bin_add(dy * dy, kb, acc[6], acc[7]); bin_add(dy * dz, kb, acc[8], acc[9]); bin_add(dz * dz, kb, acc[10], acc[11]);
        }
    }
    block_sum<NV>(acc, sm);
    if (threadIdx.x == 0) {
        const int slot = blockIdx.x & (kStatSlots - 1);
        double *dst = CENTRED ? &d->cov[slot][blockIdx.y][0][0] : &d->sum[slot][blockIdx.y][0][0];
#pragma unroll
        for (int k = 0; k < NV; k++) atomicAdd(&dst[k], acc[k]);
    }
}

__global__ void k_sal_means(SalDev *d, Bands bands) {
    const int t = threadIdx.x;
    if (t < 12) {
        const int r = t / 3, c = t % 3;
        const Band b = bands.b[r];
        const double cnt = (double)((size_t)(b.r1 - b.r0) * (size_t)(b.c1 - b.c0));
        double p0 = 0, p1 = 0;                                  // exact parts: any order
        for (int sl = 0; sl < kStatSlots; sl++) { p0 += d->sum[sl][r][c][0]; p1 += d->sum[sl][r][c][1]; }
        d->mean[r][c] = (p0 + p1) / cnt;
    }
}

// np.cov (ddof 1: products scaled by 1/(n-1)) and np.linalg.inv (LU with partial pivoting, then the
// identity solved column by column) for the four 3x3 matrices (patolette.pyx:234-244)
__global__ void k_sal_invcov(SalDev *d, Bands bands) {
    const int r = threadIdx.x;
    if (r >= 4) return;
    const Band b = bands.b[r];
    const double cnt = (double)((size_t)(b.r1 - b.r0) * (size_t)(b.c1 - b.c0));
    const double f = 1.0 / (cnt - 1.0);
    double c6[6];
    for (int k = 0; k < 6; k++) {
        double p0 = 0, p1 = 0;
        for (int sl = 0; sl < kStatSlots; sl++) { p0 += d->cov[sl][r][k][0]; p1 += d->cov[sl][r][k][1]; }
        c6[k] = (p0 + p1) * f;
    }
    double a[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    double inv[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    bool sing = false;
    for (int k = 0; k < 3; k++) {
        int p = k;
        double best = fabs(a[k][k]);
        for (int i = k + 1; i < 3; i++) if (fabs(a[i][k]) > best) { best = fabs(a[i][k]); p = i; }
        if (!(best > 0.0)) { sing = true; break; }
        if (p != k) for (int j = 0; j < 3; j++) { double t = a[k][j]; a[k][j] = a[p][j]; a[p][j] = t; t = inv[k][j]; inv[k][j] = inv[p][j]; inv[p][j] = t; }
        for (int i = k + 1; i < 3; i++) {
            const double l = a[i][k] / a[k][k];
            for (int j = k; j < 3; j++) a[i][j] -= l * a[k][j];
            for (int j = 0; j < 3; j++) inv[i][j] -= l * inv[k][j];
        }
    }
    if (!sing) {
        for (int j = 0; j < 3; j++) {                 // back substitution per right-hand side
            for (int i = 2; i >= 0; i--) {
                double t = inv[i][j];
                for (int k = i + 1; k < 3; k++) t -= a[i][k] * inv[k][j];
                inv[i][j] = t / a[i][i];
            }
        }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) d->vi[r][3 * i + j] = inv[i][j];
    } else {
        atomicExch(&d->singular, 1);
        for (int k = 0; k < 9; k++) d->vi[r][k] = 0.0;
    }
}

// scipy cdist 'mahalanobis' against one point: sqrt((x-m)^T VI (x-m)), row-major VI, sequential sums
__device__ __forceinline__ double mahalanobis(const double lab[3], const double *__restrict__ m, const double *__restrict__ vi) {
    const double d0 = lab[0] - m[0], d1 = lab[1] - m[1], d2 = lab[2] - m[2];
    double t[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { double s = 0.0; s += d0 * vi[3 * i + 0]; s += d1 * vi[3 * i + 1]; s += d2 * vi[3 * i + 2]; t[i] = s; }
    double s = 0.0;
    s += d0 * t[0]; s += d1 * t[1]; s += d2 * t[2];
    return sqrt(s);
}

template <int NV>
__device__ __forceinline__ void block_max_publish(double (&v)[NV], const int (&which)[NV], SalDev *d) {
    __shared__ double smx[4][NV];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double a = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a = fmax(a, __shfl_down(a, o, 64));
        if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6][k] = a;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const int k = threadIdx.x;
        double a = smx[0][k];
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) a = fmax(a, smx[w][k]);
        int idx = which[0];
#pragma unroll
        for (int j = 1; j < NV; j++) if (k == j) idx = which[j];
        atomicMax(&d->maxkey[idx][blockIdx.x & (kStatSlots - 1)], f64_key(a));
    }
}

// fold the slots of maxima [first, first+count) ; `as_f32` rounds through float like the reference's `cdef float`
__global__ void k_sal_fold(SalDev *d, int first, int count, int as_f32) {
    const int k = threadIdx.x;
    if (k >= count) return;
    unsigned long long best = d->maxkey[first + k][0];
    for (int s = 1; s < kStatSlots; s++) { const unsigned long long v = d->maxkey[first + k][s]; if (v > best) best = v; }
    double m = key_f64(best);
    if (as_f32) m = (double)(float)m;
    d->mx[first + k] = m;
}

// pass A: maxima of the four contrasts (patolette.pyx:272-275) and of the barrier distance (:289)
__global__ __launch_bounds__(256) void k_sal_pass_a(const double *__restrict__ lab, const float *__restrict__ D, size_t n, SalDev *d) {
    double mx[5] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double c[3] = {lab[i], lab[n + i], lab[2 * n + i]};
#pragma unroll
        for (int r = 0; r < 4; r++) mx[r] = fmax(mx[r], mahalanobis(c, d->mean[r], d->vi[r]));
        mx[4] = fmax(mx[4], (double)D[i]);
    }
    const int which[5] = {0, 1, 2, 3, 4};
    block_max_publish<5>(mx, which, d);
}

// pass B: u_final = (u_left + u_right + u_top + u_bottom) - max(...) of the normalised contrasts (:277-286)
__global__ __launch_bounds__(256) void k_sal_pass_b(const double *__restrict__ lab, size_t n, SalDev *d, double *__restrict__ out) {
    double mx[1] = {-INFINITY};
    const double m0 = d->mx[0], m1 = d->mx[1], m2 = d->mx[2], m3 = d->mx[3];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double c[3] = {lab[i], lab[n + i], lab[2 * n + i]};
        const double ul = mahalanobis(c, d->mean[0], d->vi[0]) / m0;
        const double ur = mahalanobis(c, d->mean[1], d->vi[1]) / m1;
        const double ut = mahalanobis(c, d->mean[2], d->vi[2]) / m2;
        const double ub = mahalanobis(c, d->mean[3], d->vi[3]) / m3;
        const double um = fmax(fmax(fmax(ul, ur), ut), ub);
        const double uf = (((ul + ur) + ut) + ub) - um;
        out[i] = uf;
        mx[0] = fmax(mx[0], uf);
    }
    const int which[1] = {5};
    block_max_publish<1>(mx, which, d);
}

// pass C: sal / sal_max (f32 / f32) + u_final / u_max_final (:291)
__global__ __launch_bounds__(256) void k_sal_pass_c(const float *__restrict__ D, size_t n, SalDev *d, double *__restrict__ s) {
    double mx[1] = {-INFINITY};
    const float dmax = (float)d->mx[4];
    const double ufmax = d->mx[5];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double v = (double)(D[i] / dmax) + s[i] / ufmax;
        s[i] = v;
        mx[0] = fmax(mx[0], v);
    }
    const int which[1] = {6};
    block_max_publish<1>(mx, which, d);
}

// pass D: normalise (:292) and apply the centre prior C (:294-302)
__global__ __launch_bounds__(256) void k_sal_pass_d(size_t n, int cols, double w2, double h2, double diag, SalDev *d, double *__restrict__ s) {
    double mx[1] = {-INFINITY};
    const double m = d->mx[6];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t r = i / (size_t)cols, q = i - r * (size_t)cols;
        const double dx = (double)q - h2, dy = (double)r - w2;
        const double c = 1.0 - sqrt(dx * dx + dy * dy) / diag;
        const double v = (s[i] / m) * c;
        s[i] = v;
        mx[0] = fmax(mx[0], v);
    }
    const int which[1] = {7};
    block_max_publish<1>(mx, which, d);
}

// pass E: normalise, sigmoid (:304-310), weights (:313)
__global__ __launch_bounds__(256) void k_sal_pass_e(size_t n, double npx, double tile2, const SalDev *d, const double *__restrict__ s,
                                                    double *__restrict__ w) {
    const double m = d->mx[7];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double x = s[i] / m;
        const double f = 1.0 / (1.0 + exp(-10.0 * (x - 0.5)));
        w[i] = 1.0 + ((f * f) * npx) / tile2;
    }
}

// ---------------------------------------------------------------------------------------------------
// minimum-barrier raster scan (patolette.pyx:54-152)
// ---------------------------------------------------------------------------------------------------
struct MbdArgs {
    float4 *st;                      // the skewed state (SkewGeom)
    SkewGeom geo;
    int rows, cols;
    unsigned int *progress;
    unsigned int *stalled;           // set when a strip gave up waiting for the one above it
    int hs;                          // chunks per flag handshake with the neighbouring strips
};

__device__ __forceinline__ float wave_shr1(float from_above, float v) {
    // lane i receives lane i-1's v; lane 0 keeps from_above
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, from_above), __builtin_bit_cast(int, v),
                                                                 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
}

__device__ __forceinline__ float wave_shl1(float v) {
    // lane i receives lane i+1's v; lane 63 keeps its own
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v),
                                                                 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
}

constexpr int kChunk = 32;           // columns of the row above a strip fetched / published at a time

// DIR = +1: forward scan (rows 1..rows-2, cols 1..cols-2, neighbours above / left);
// DIR = -1: inverse scan (rows rows-2..2, cols cols-2..2, neighbours below / right).
// Scan space: row sx, column sy count from the first visited row / column in visiting order, so the
// already-visited neighbours are always (sx-1, sy) and (sx, sy-1), and sx = -1 / sy = -1 is the frame.
// One wavefront per strip of 64 scan rows; lane r visits column t - r at step t, so the visit of the row above
// is one step old in the neighbouring lane (wave shift) and the previous column is the lane's own last result.
// Strip s+1 reads the last row of strip s from HBM, kChunk columns at a time, behind a progress flag.
// v_max_f32 / v_min_f32 without the canonicalising self-max the compiler adds for fmaxf / fminf (no NaNs here)
__device__ __forceinline__ float hw_max(float x, float y) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ float hw_min(float x, float y) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }

template <int DIR>
__global__ __launch_bounds__(64) void k_mbd_scan(MbdArgs a) {
    const int lane = threadIdx.x, strip = blockIdx.x;
    const int R = DIR > 0 ? a.rows - 2 : a.rows - 3;
    const int Cn = DIR > 0 ? a.cols - 2 : a.cols - 3;
    const long rs = DIR > 0 ? (long)a.cols : -(long)a.cols;
    const long base = DIR > 0 ? (long)a.cols + 1 : (long)(a.rows - 2) * a.cols + (a.cols - 2);
    const int sx = strip * 64 + lane;
    const bool rowok = sx < R;
    const int rl = min(63, R - 1 - strip * 64);             // last valid lane of this strip
    // The lane's row and its column at step 0 (one column behind per lane); the record of step t is p[SD * t]: one diagonal
    // further, 64 records on.  Lanes past the last row and columns outside [0, Cn) land in cells that belong to no pixel (or to
    // the frame) and are only ever read.
    constexpr long SD = (long)DIR * 64;
    (void)rs; (void)base;
    const int row_l = DIR > 0 ? 1 + sx : a.rows - 2 - sx, col0_l = DIR > 0 ? 1 - lane : a.cols - 2 + lane;
    float4 *p = a.st + a.geo.idx(row_l, col0_l);
    const float4 *above = a.st + a.geo.idx(DIR > 0 ? 64 * strip : a.rows - 1 - 64 * strip, DIR > 0 ? 1 : a.cols - 2);   // the row before the strip's first
    const int T = Cn + rl;                                   // steps until the last valid lane is done

    float4 ring[kChunk];
    float outU = 0.f, outL = 0.f;                            // this lane's latest result = "previous column" of the next visit
    { const float4 f = p[SD * (lane - 1)]; outU = f.x; outL = f.x; }    // frame column: U = L = img, never modified
    float bu = 0.f, bl = 0.f;                                // row above the strip: lane j holds column chunk0 + j
    const int last = rowok ? lane + Cn - 1 : -1;             // this lane visits at steps lane .. last

    // One visit (patolette.pyx:84-119), without branches: b1 / b2 are the barriers through the neighbour above and the one
    // before; the pixel keeps its state when d <= min(b1, b2), else takes the smaller barrier (the first on a tie) and is
    // written back.  bu / bl hold the row above the strip, one column per lane, and rotate down one lane per step so that
    // lane 0 -- whose "lane above" is that row -- always finds its column in its own register (no readlane round trip
    // through the scalar unit).  The loop-carried chain is shift -> max/min -> sub -> min -> compare -> two selects.
    // steady: every valid lane visits; full (with steady): every lane is a valid row -- the record is then stored unconditionally
    // (unchanged pixels rewrite their own values): no branch around the store, eight instructions less of a step's ~40, and the
    // lone wavefront's issue rate is what a pass waits for
    auto visit = [&](int t, float4 *q, const float4 cur, const bool steady, const bool full) {
        const float u1 = wave_shr1(bu, outU), l1 = wave_shr1(bl, outL);
        bu = wave_shl1(bu); bl = wave_shl1(bl);
        const float ix = cur.x, d = cur.y;
        const float hu1 = hw_max(u1, ix), hl1 = hw_min(l1, ix);
        const float hu2 = hw_max(outU, ix), hl2 = hw_min(outL, ix);
        const float b1 = hu1 - hl1, b2 = hu2 - hl2;
        const float m = hw_min(b1, b2);
        const bool keep = d <= m;
        const bool t1 = b1 <= b2;                               // given !keep this is (b1 < d) && (b1 <= b2)
        const float nu = keep ? cur.z : (t1 ? hu1 : hu2), nl = keep ? cur.w : (t1 ? hl1 : hl2);
        if (steady && full) {
            outU = nu; outL = nl;
            *q = make_float4(ix, keep ? d : m, nu, nl);
        } else if (steady) {
            outU = nu; outL = nl;
            if (!keep && rowok) *q = make_float4(ix, m, nu, nl);
        } else {
            const bool active = t >= lane && t <= last;
            outU = active ? nu : outU; outL = active ? nl : outL;
            if (active && !keep) *q = make_float4(ix, m, nu, nl);
        }
    };
    // The handshake with the strip above (a spin on its progress flag and an acquire fence) and with the strip below (a release
    // fence and a flag store) can be made once per a.hs chunks; the strip then lags its neighbour by two windows of hs * kChunk
    // steps instead of three chunks.  Measured at 4096 x 4096: a handshake costs ~0.7 us, a step ~90 ns, so hs = 2 trades
    // 2 100 more steps for half the handshakes and comes out even (1.37 ms per pass either way); hs = 3, 4 are slower.
    auto chunk_begin = [&](int t0, bool handshake) -> bool {             // true: gave up waiting
        // steps t0 .. t0+kChunk-1 of lane 0 consume scan columns t0 .. t0+kChunk-1 of the row above the strip
        if (t0 >= Cn) return false;
        if (strip > 0 && handshake) {
            const unsigned int need = (unsigned int)min(Cn, t0 + a.hs * kChunk);
            // bounded: the strip above always makes progress when all strips are resident (a few hundred single-wave
            // blocks), but a spin must never be able to hang the device -- past ~2 s the pass gives up and reports it
            unsigned spins = 0;
            while (__hip_atomic_load(&a.progress[strip - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 24)) { if (lane == 0) atomicExch(a.stalled, 1u); return true; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        const int col = t0 + lane;
        if (lane < kChunk && col < Cn) {
            const float4 f = above[SD * col];
            bu = f.z; bl = f.w;
        }
        return false;
    };
    auto chunk_end = [&](int t) {
        // columns of the strip's last row that are complete and stored after step t
        const int done = t - rl + 1;
        if (done <= 0) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) __hip_atomic_store(&a.progress[strip], (unsigned int)min(Cn, done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // Blocks of kChunk steps.  The records of the next block are requested during the first half of the current one,
    // so they have landed when the flag handshake at the block boundary drains the memory counter.
#pragma unroll
    for (int j = 0; j < kChunk; j++) ring[j] = p[SD * j];
    for (int t0 = 0, ci = 0; t0 < T; t0 += kChunk, ci = ci + 1 == a.hs ? 0 : ci + 1) {
        if (chunk_begin(t0, ci == 0)) {                                           // let the strips below give up quickly too
            if (lane == 0) __hip_atomic_store(&a.progress[strip], 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        float4 next[kChunk];
        float4 *q = p + SD * t0;
        // every valid lane visits at every step of the block: after the ramp-up (t0 >= 63) and before lane 0 runs out of columns
        const bool steady = t0 >= 64 && t0 + kChunk <= Cn;
        if (steady && rl == 63) {
#pragma unroll
            for (int j = 0; j < kChunk; j++) {
                if (j < kChunk / 2) { next[2 * j] = q[SD * (kChunk + 2 * j)]; next[2 * j + 1] = q[SD * (kChunk + 2 * j + 1)]; }
                visit(t0 + j, q + SD * j, ring[j], true, true);
            }
        } else if (steady) {
#pragma unroll
            for (int j = 0; j < kChunk; j++) {
                if (j < kChunk / 2) { next[2 * j] = q[SD * (kChunk + 2 * j)]; next[2 * j + 1] = q[SD * (kChunk + 2 * j + 1)]; }
                visit(t0 + j, q + SD * j, ring[j], true, false);
            }
        } else {
#pragma unroll
            for (int j = 0; j < kChunk; j++) {
                if (j < kChunk / 2) { next[2 * j] = q[SD * (kChunk + 2 * j)]; next[2 * j + 1] = q[SD * (kChunk + 2 * j + 1)]; }
                visit(t0 + j, q + SD * j, ring[j], false, false);
            }
        }
        if (ci + 1 == a.hs) chunk_end(t0 + kChunk - 1);
#pragma unroll
        for (int j = 0; j < kChunk; j++) ring[j] = next[j];
    }
    chunk_end(T + kChunk);
}

int stream_grid(size_t n) {
    size_t b = ceil_div(n, 256);
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

int log2_ceil(size_t v) { int p = 0; while (((size_t)1 << p) < v) p++; return p; }

}  // namespace

__global__ __launch_bounds__(256) void k_mbd_init(const float *__restrict__ img, size_t n, int rows, int cols, float4 *__restrict__ st) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / (size_t)cols), q = (int)(i - (size_t)r * cols);
        const float m = img[i];
        st[i] = make_float4(m, (r == 0 || q == 0 || r == rows - 1 || q == cols - 1) ? 0.0f : INFINITY, m, m);
    }
}

// row-major state -> skewed copy: one thread per skewed cell (coalesced writes; the reads are one line per lane)
__global__ __launch_bounds__(256) void k_mbd_skew(const float4 *__restrict__ st, int rows, int cols, SkewGeom geo, int groups, float4 *__restrict__ sk) {
    const size_t cells = (size_t)groups * geo.dn * 64, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
        const int l = (int)(i & 63);
        const size_t gd = i >> 6;
        const int g = (int)(gd / (size_t)geo.dn), d = (int)(gd - (size_t)g * geo.dn);
        const int r = 64 * g + l - 63, c = d - l - 128;
        if (r >= 0 && r < rows && c >= 0 && c < cols) sk[i] = st[(size_t)r * cols + c];
    }
}
// and back: one thread per pixel (coalesced writes), the barrier distance alone -- nothing else of the state is read after the scans
__global__ __launch_bounds__(256) void k_mbd_unskew(const float4 *__restrict__ sk, size_t n, int cols, SkewGeom geo, float *__restrict__ D) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / (size_t)cols), c = (int)(i - (size_t)r * cols);
        D[i] = reinterpret_cast<const float *>(sk + geo.idx(r, c))[1];
    }
}

// pass p of mbd(img, iters) is the forward scan when p is odd, the inverse scan when p is even (patolette.pyx:180-199)
static SkewGeom mbd_geometry(SalWork &w, int rows, int cols) {
    const SkewGeom geo{cols + 320};
    const int groups = ((rows + 126) >> 6) + 1;              // rows -63 .. rows + 62 (the lanes of the last strips run past the image)
    w.skew.reserve((size_t)groups * geo.dn * 64 + 64);
    return geo;
}
// st_rowmajor: the state in row-major order to start from (mbd_device), or nullptr when w.skew holds it already (k_sal_prepare);
// d_out: the barrier distance, row-major
static void run_mbd_scans(SalWork &w, int rows, int cols, int iters, hipStream_t s, const float4 *st_rowmajor, float *d_out) {
    const int strips_f = (int)ceil_div((size_t)rows - 2, 64), strips_i = (int)ceil_div((size_t)rows - 3, 64);
    w.progress.reserve((size_t)strips_f + 1);                // [strips] progress + [1] stall flag
    static const int hs = getenv("PAMD_MBD_HS") ? std::max(1, atoi(getenv("PAMD_MBD_HS"))) : 1;
    const SkewGeom geo = mbd_geometry(w, rows, cols);
    const int groups = ((rows + 126) >> 6) + 1;
    const size_t cells = (size_t)groups * geo.dn * 64;
    if (st_rowmajor) {
        KTIME("k_mbd_skew", s, 32.0 * rows * cols);
        hipLaunchKernelGGL(k_mbd_skew, stream_grid(cells), 256, 0, s, st_rowmajor, rows, cols, geo, groups, w.skew.p);
    }
    MbdArgs ma{w.skew.p, geo, rows, cols, w.progress.p, w.progress.p + strips_f, hs};
    w.d_stall = w.progress.p + strips_f;
    HIP_CHECK(hipMemsetAsync(w.d_stall, 0, sizeof(unsigned int), s));
    for (int pass = 0; pass < iters; pass++) {
        HIP_CHECK(hipMemsetAsync(w.progress.p, 0, sizeof(unsigned int) * (size_t)strips_f, s));
        KTIME("k_mbd_scan", s, 32.0 * rows * cols);
        if (pass % 2 == 1) hipLaunchKernelGGL(k_mbd_scan<1>, strips_f, 64, 0, s, ma);
        else hipLaunchKernelGGL(k_mbd_scan<-1>, strips_i, 64, 0, s, ma);
    }
    {
        KTIME("k_mbd_skew", s, 8.0 * rows * cols);
        hipLaunchKernelGGL(k_mbd_unskew, stream_grid((size_t)rows * cols), 256, 0, s, w.skew.p, (size_t)rows * cols, cols, geo, d_out);
    }
    HIP_CHECK(hipGetLastError());
}

int mbd_device(SalWork &w, const float *h_img, size_t rows, size_t cols, int iters, float *h_out, hipStream_t s) {
    if (rows <= 3 || cols <= 3) return kSalBadShape;
    const size_t n = rows * cols;
    w.st.reserve(n + 2 * kMbdPad); w.tmp.reserve(n);
    HIP_CHECK(hipMemcpyAsync(w.tmp.p, h_img, n * sizeof(float), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_mbd_init, stream_grid(n), 256, 0, s, w.tmp.p, n, (int)rows, (int)cols, w.st.p + kMbdPad);
    run_mbd_scans(w, (int)rows, (int)cols, iters, s, w.st.p + kMbdPad, w.tmp.p);     // (the input image in w.tmp has been consumed by k_mbd_init)
    HIP_CHECK(hipMemcpyAsync(h_out, w.tmp.p, n * sizeof(float), hipMemcpyDeviceToHost, s));
    unsigned int stalled = 0;
    HIP_CHECK(hipMemcpyAsync(&stalled, w.d_stall, sizeof stalled, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (stalled) throw HipError("patolette_amd: the raster scan gave up waiting for a strip (device oversubscribed?)");
    return kSalOk;
}

int saliency_weights(SalWork &w, const double *d_f64, const unsigned char *d_u8, int channels, size_t width, size_t height,
                     double tile_size, double *d_weights, hipStream_t s) {
    const size_t n = width * height;
    const int rows = (int)height, cols = (int)width;
    // shapes get_weights cannot process (patolette.pyx:157-158 mbd -> None; :211 empty border band; :228-232 reshape)
    const int bt = (int)std::floor(0.1 * std::sqrt((double)(height * width)));
    if (rows <= 3 || cols <= 3 || bt < 1 || rows < bt + 1 || cols < bt + 1) return kSalBadShape;

    w.tmp.reserve(n);                                         // the barrier distance after the scans
    w.lab.reserve(3 * n); w.s.reserve(n);
    const SkewGeom geo = mbd_geometry(w, rows, cols);        // the barrier state is made in the scans' skewed layout at once (k_sal_prepare)
    w.dev.reserve(1);
    w.host.reserve(1);
    const int g = stream_grid(n);

    hipLaunchKernelGGL(k_sal_init, 1, 256, 0, s, w.dev.p);
    {
        KTIME("k_sal_prepare", s, (d_u8 ? (double)channels : 24.0) * n + 40.0 * n);
        if (d_u8) hipLaunchKernelGGL((k_sal_prepare<SrcU8>), g, 256, 0, s, SrcU8{d_u8, channels}, n, rows, cols, w.skew.p, geo, w.lab.p);
        else if (channels < 0) hipLaunchKernelGGL((k_sal_prepare<SrcF64Rows>), g, 256, 0, s, SrcF64Rows{d_f64}, n, rows, cols, w.skew.p, geo, w.lab.p);
        else hipLaunchKernelGGL((k_sal_prepare<SrcF64>), g, 256, 0, s, SrcF64{d_f64, n}, n, rows, cols, w.skew.p, geo, w.lab.p);
    }
    // border bands in the reference's order and naming (patolette.pyx:215-219): "left" = first bt rows, "right" = bt rows
    // ending one short of the last, "top" = first bt columns, "bottom" = bt columns ending one short of the last
    Bands bands;
    bands.b[0] = Band{0, bt, 0, cols};
    bands.b[1] = Band{rows - bt - 1, rows - 1, 0, cols};
    bands.b[2] = Band{0, rows, 0, bt};
    bands.b[3] = Band{0, rows, cols - bt - 1, cols - 1};
    size_t maxcnt = 0;
    for (int r = 0; r < 4; r++) maxcnt = std::max(maxcnt, (size_t)(bands.b[r].r1 - bands.b[r].r0) * (size_t)(bands.b[r].c1 - bands.b[r].c0));
    const int P = log2_ceil(maxcnt);
    const dim3 bg((unsigned)std::min<size_t>(ceil_div(maxcnt, 256), 1024), 4);
    {
        KTIME("k_sal_band", s, 24.0 * maxcnt * 4);
        hipLaunchKernelGGL(k_sal_band<false>, bg, 256, 0, s, w.lab.p, n, cols, bands, w.dev.p, make_bink(kE_lab, P));
    }
    hipLaunchKernelGGL(k_sal_means, 1, 64, 0, s, w.dev.p, bands);
    {
        KTIME("k_sal_band", s, 24.0 * maxcnt * 4);
        hipLaunchKernelGGL(k_sal_band<true>, bg, 256, 0, s, w.lab.p, n, cols, bands, w.dev.p, make_bink(kE_cov, P));
    }
    hipLaunchKernelGGL(k_sal_invcov, 1, 64, 0, s, w.dev.p, bands);

    run_mbd_scans(w, rows, cols, 3, s, nullptr, w.tmp.p);    // mbd(img_mean, 3), patolette.pyx:205
    {
        KTIME("k_sal_pass_a", s, 28.0 * n);
        hipLaunchKernelGGL(k_sal_pass_a, g, 256, 0, s, w.lab.p, (const float *)w.tmp.p, n, w.dev.p);
    }
    hipLaunchKernelGGL(k_sal_fold, 1, 64, 0, s, w.dev.p, 0, 5, 1);
    {
        KTIME("k_sal_pass_b", s, 32.0 * n);
        hipLaunchKernelGGL(k_sal_pass_b, g, 256, 0, s, w.lab.p, n, w.dev.p, w.s.p);
    }
    hipLaunchKernelGGL(k_sal_fold, 1, 64, 0, s, w.dev.p, 5, 1, 1);
    {
        KTIME("k_sal_pass_c", s, 20.0 * n);
        hipLaunchKernelGGL(k_sal_pass_c, g, 256, 0, s, (const float *)w.tmp.p, n, w.dev.p, w.s.p);
    }
    hipLaunchKernelGGL(k_sal_fold, 1, 64, 0, s, w.dev.p, 6, 1, 0);
    const double w2 = (double)rows / 2.0, h2 = (double)cols / 2.0;
    const double diag = std::sqrt(w2 * w2 + h2 * h2);
    {
        KTIME("k_sal_pass_d", s, 16.0 * n);
        hipLaunchKernelGGL(k_sal_pass_d, g, 256, 0, s, n, cols, w2, h2, diag, w.dev.p, w.s.p);
    }
    hipLaunchKernelGGL(k_sal_fold, 1, 64, 0, s, w.dev.p, 7, 1, 0);
    {
        KTIME("k_sal_pass_e", s, 16.0 * n);
        hipLaunchKernelGGL(k_sal_pass_e, g, 256, 0, s, n, (double)(height * width), tile_size * tile_size, w.dev.p, w.s.p, d_weights);
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(w.host.p, w.dev.p, sizeof(SalDev), hipMemcpyDeviceToHost, s));
    unsigned int stalled = 0;
    HIP_CHECK(hipMemcpyAsync(&stalled, w.d_stall, sizeof stalled, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (stalled) throw HipError("patolette_amd: the raster scan gave up waiting for a strip (device oversubscribed?)");
    if (w.host.p->singular) return kSalSingular;
    return kSalOk;
}

}  // namespace pamd
