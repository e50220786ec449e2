// map.hip -- final palette mapping on gfx950: exact nearest-palette assignment and the
// Riemersma (Hilbert-curve) error-diffusion dither.
//
// k_nn_map replaces patolette__PALETTE_fill_palette_map_nearest (lib/src/palette/nearest.c:150-209,
// FLANN exact 1-NN, eps = 0): f64 distance ((d0^2)+d1^2)+d2^2 over all K entries, strict '<' in
// ascending index order (lowest index on ties).  Algorithmic traffic 24 B/px read + 1 (K<=256) or
// 4 B/px written; brute force at K = 256 is f64-VALU-bound, the palette comes
// through the scalar cache (wave-uniform index -> s_load).
//
// k_dither replaces patolette__DITHER_riemersma (lib/src/dither/riemersma.c:437-459): the chain
// of W*H steps is serial by construction (each choice feeds the 16-entry error queue), so one
// wavefront walks one image: per 64-step batch the lanes decode 64 curve positions and prefetch
// their pixels in parallel, then the steps run in order -- error sum in reference order, K/64
// palette entries per lane, wave arg-min with the lowest-index tie rule.  Latency-bound; batch
// parallelism (one wavefront per image) is where throughput comes from.
#include "map.h"

#include <cmath>

#include "devutil.h"

namespace pamd {

template <typename OutT>
__global__ __launch_bounds__(256) void k_nn_map(const double *__restrict__ c, size_t N, size_t n, const double *__restrict__ pal, int k,
                                                OutT *__restrict__ out) {
    // c: planar with plane stride N (n <= N pixels mapped); pal: planar (k,3)
    const double *palx = pal, *paly = pal + k, *palz = pal + 2 * k;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double x = c[i], y = c[N + i], z = c[2 * N + i];
        double bd = INFINITY; int best = 0;
        for (int j = 0; j < k; j++) {
            const double d0 = x - palx[j], d1 = y - paly[j], d2 = z - palz[j];
            const double d = (d0 * d0 + d1 * d1) + d2 * d2;
            if (d < bd) { bd = d; best = j; }
        }
        out[i] = (OutT)best;
    }
}

void launch_nn_map(const double *d_colors, size_t plane_stride, size_t n, const double *d_pal, int k, void *d_out, int elem_bytes, hipStream_t s) {
    size_t g = ceil_div(n, 256);
    if (g > 256 * 32) g = 256 * 32;
    if (g < 1) g = 1;
    KTIME("k_nn_map", s, (24.0 + elem_bytes) * n);
    if (elem_bytes == 1) hipLaunchKernelGGL(k_nn_map<unsigned char>, (int)g, 256, 0, s, d_colors, plane_stride, n, d_pal, k, (unsigned char *)d_out);
    else if (elem_bytes == 4) hipLaunchKernelGGL(k_nn_map<unsigned int>, (int)g, 256, 0, s, d_colors, plane_stride, n, d_pal, k, (unsigned int *)d_out);
    else if (elem_bytes == 8) hipLaunchKernelGGL(k_nn_map<unsigned long long>, (int)g, 256, 0, s, d_colors, plane_stride, n, d_pal, k, (unsigned long long *)d_out);
    else throw HipError("patolette_amd: map element size must be 1, 4 or 8");
    HIP_CHECK(hipGetLastError());
}

// --------------------------------------------------------------------------------------------
// Riemersma dither
// --------------------------------------------------------------------------------------------
// Hilbert index -> (x, y) on the 2^L square; visiting order identical to traverse_level(L, UP)
// from (0,0) (riemersma.c:176-257), checked against the oracle's recorded walk.
__device__ __forceinline__ void hilbert_d2xy(int L, unsigned long long d, unsigned &xo, unsigned &yo) {
    unsigned x = 0, y = 0;
    unsigned long long t = d;
    for (int lv = 0; lv < L; lv++) {
        const unsigned s = 1u << lv;
        const unsigned rx = 1u & (unsigned)(t >> 1);
        const unsigned ry = 1u & ((unsigned)t ^ rx);
        if (ry == 0) {
            if (rx == 1) { x = s - 1 - x; y = s - 1 - y; }
            unsigned tmp = x; x = y; y = tmp;
        }
        x += s * rx; y += s * ry;
        t >>= 2;
    }
    xo = x; yo = y;
}

struct DitherWeights { double w[16]; };

constexpr double kRw = 0.51254268114958, kGw = 0.8234075540095561, kBw = 0.2435159132377184;   // riemersma.c:38-42

template <typename OutT>
__global__ __launch_bounds__(64) void k_dither(const double *__restrict__ img, size_t plane_stride, unsigned width, unsigned height,
                                               const double *__restrict__ pal /* planar (k,3), linear Rec2020 */, int k,
                                               OutT *__restrict__ out, DitherWeights wts) {
    extern __shared__ double lds[];
    double *pwx = lds, *pwy = lds + k, *pwz = lds + 2 * k;          // palette scaled by (float)-cast weights (riemersma.c:419-425)
    double *prx = lds + 3 * k, *pry = lds + 4 * k, *prz = lds + 5 * k;  // raw palette
    const int lane = threadIdx.x;
    const double fx = (double)(float)kRw, fy = (double)(float)kGw, fz = (double)(float)kBw;
    for (int j = lane; j < k; j += 64) {
        const double a = pal[j], b = pal[k + j], c = pal[2 * k + j];
        prx[j] = a; pry[j] = b; prz[j] = c;
        pwx[j] = a * fx; pwy[j] = b * fy; pwz[j] = c * fz;
    }
    __syncthreads();
    // level = ceil(log2(max(w,h)))  (riemersma.c:124-144); level 0 visits nothing
    const unsigned mx = width > height ? width : height;
    int L = 0;
    while ((1u << L) < mx) L++;
    if (L == 0) return;
    // queue weights w_i = m^i / 16, m = exp(ln 16 / 15)  (riemersma.c:360-373)
    // (computed on the host with libm like the reference; passed in so no device exp/log ulp leaks in)
    double qw[16];
#pragma unroll
    for (int i = 0; i < 16; i++) qw[i] = wts.w[i];
    double qr[16], qg[16], qb[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { qr[i] = 0; qg[i] = 0; qb[i] = 0; }

    const unsigned long long total = 1ULL << (2 * L);
    const double *pr = img, *pg = img + plane_stride, *pb = img + 2 * plane_stride;
    unsigned long long d0 = 0;
    while (d0 < total) {
        // skip whole out-of-image sub-squares: a 4^j-aligned run of 4^j steps stays inside one aligned 2^j square
        if (L >= 3) {
            bool skipped = false;
            for (int j = L; j >= 3; j--) {
                const unsigned long long span = 1ULL << (2 * j);
                if ((d0 & (span - 1)) != 0) continue;
                unsigned x0, y0;
                hilbert_d2xy(L, d0, x0, y0);
                x0 &= ~((1u << j) - 1u); y0 &= ~((1u << j) - 1u);
                if (x0 >= width || y0 >= height) { d0 += span; skipped = true; break; }
            }
            if (skipped) continue;
        }
        // the lanes decode 64 (or 4^L if smaller) consecutive steps and prefetch their pixels
        const unsigned long long dl = d0 + (unsigned long long)lane;
        unsigned x = 0, y = 0;
        bool inb = false;
        double R = 0, G = 0, B = 0;
        if (dl < total) {
            hilbert_d2xy(L, dl, x, y);
            inb = x < width && y < height;
            if (inb) { const size_t p = (size_t)y * width + x; R = pr[p]; G = pg[p]; B = pb[p]; }
        }
        const unsigned long long mask = __ballot(inb);
        int myidx = 0;
        for (int t = 0; t < 64; t++) {
            if (!((mask >> t) & 1ULL)) continue;                       // wave-uniform
            const double pR = __shfl(R, t, 64), pG = __shfl(G, t, 64), pB = __shfl(B, t, 64);
            double eR = 0, eG = 0, eB = 0;                            // riemersma.c:286-296
#pragma unroll
            for (int i = 0; i < 16; i++) { eR += qr[i] * qw[i]; eG += qg[i] * qw[i]; eB += qb[i] * qw[i]; }
            const double cR = pR + eR, cG = pG + eG, cB = pB + eB;
            const double qx = kRw * cR, qy = kGw * cG, qz = kBw * cB;
            double bd = INFINITY; int bi = 0x7fffffff;
            for (int j = lane; j < k; j += 64) {
                const double e0 = qx - pwx[j], e1 = qy - pwy[j], e2 = qz - pwz[j];
                const double dd = (e0 * e0 + e1 * e1) + e2 * e2;
                if (dd < bd) { bd = dd; bi = j; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double od = __shfl_xor(bd, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
            }
            if (lane == t) myidx = bi;
#pragma unroll
            for (int i = 0; i < 15; i++) { qr[i] = qr[i + 1]; qg[i] = qg[i + 1]; qb[i] = qb[i + 1]; }
            qr[15] = pR - prx[bi]; qg[15] = pG - pry[bi]; qb[15] = pB - prz[bi];   // original pixel - chosen colour
        }
        if (inb) out[(size_t)y * width + x] = (OutT)myidx;
        d0 += 64;
    }
}

void launch_dither(const double *d_img, size_t plane_stride, size_t width, size_t height, const double *d_pal, int k,
                   void *d_out, int elem_bytes, hipStream_t s) {
    size_t lds = (size_t)6 * k * sizeof(double);
    if (lds > 150 * 1024) throw HipError("patolette_amd: palette too large for the dither kernel (K <= 3200)");
    DitherWeights wts;
    {
        const double m = std::exp(std::log(16.0) / (16.0 - 1));
        double v = 1;
        for (int i = 0; i < 16; i++) { wts.w[i] = v / 16.0; v *= m; }
    }
    KTIME("k_dither", s, (24.0 + elem_bytes) * width * height);
    if (elem_bytes == 1) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_dither<unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_dither<unsigned char>, 1, 64, lds, s, d_img, plane_stride, (unsigned)width, (unsigned)height, d_pal, k, (unsigned char *)d_out, wts);
    } else if (elem_bytes == 4) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_dither<unsigned int>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_dither<unsigned int>, 1, 64, lds, s, d_img, plane_stride, (unsigned)width, (unsigned)height, d_pal, k, (unsigned int *)d_out, wts);
    } else if (elem_bytes == 8) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_dither<unsigned long long>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(k_dither<unsigned long long>, 1, 64, lds, s, d_img, plane_stride, (unsigned)width, (unsigned)height, d_pal, k, (unsigned long long *)d_out, wts);
    } else throw HipError("patolette_amd: map element size must be 1, 4 or 8");
    HIP_CHECK(hipGetLastError());
}

}  // namespace pamd
