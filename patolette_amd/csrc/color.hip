// color.hip -- per-pixel colour conversions on planar (N,3) f64 matrices (gfx950).
//
// Replaces lib/src/color/{ICtCp,CIELuv,rec2020,sRGB,xyz,eotf}.c of the reference.  The
// reference walks the matrix once per hop (sRGB->XYZ->Rec2020->ICtCp is one fused loop, but
// the CIELuv no-dither path is three full sweeps, patolette.c:305-314); here every chain is
// one kernel: 3 coalesced 8-byte loads per pixel, the whole chain in registers, 3 stores.
// Arithmetic follows the reference expression by expression in f64 with contraction off;
// only pow() differs (ocml vs glibc, last-ulp).  Bound: 48 B/px of HBM traffic; ~9 f64 pow
// per pixel make the ICtCp chain VALU-heavy, still under the HBM time at 16 B/lane loads.
#define PAMD_POW_TABLES_IN_LDS
#include "color_device.h"
#include "common.h"
#include "devutil.h"

#include <type_traits>

namespace pamd {

// sumk.M0 != 0: also accumulate the column sums of the output (order-independent binned parts; the bound behind sumk
// is a property of the colour space, so it is known before the pass)
template <int WHICH, class SRC>
__global__ __launch_bounds__(256) void k_convert(SRC src, double *__restrict__ dst, size_t n, ConvertStats *stats, BinK sumk, BinK momk,
                                                 size_t begin, size_t end) {
    // pixels [begin, end) of the n-pixel image (n = plane stride): the host entry converts an image chunk by chunk behind its upload
    // per-plane min / max of the OUTPUT (bounds for the binned accumulators downstream)
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    double acc[6] = {0, 0, 0, 0, 0, 0};
    bool bad = false;                                      // (float)value would be NaN or Inf: at or beyond FLT_MAX + half an ulp
    const bool do_sum = stats != nullptr && sumk.M0 != 0.0;
    // momk.M0 != 0: the raw second moments of the output too (xx, yx, zx, yy, zy, zz) -- with the column sums they give the root's
    // centred covariance without another sweep (pipeline.hip quantize_clusters)
    const bool do_mom = do_sum && momk.M0 != 0.0;
    double acc2[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    constexpr bool kLut = std::is_same<SRC, SrcU8>::value && (WHICH == PAMD_SRGB_TO_ICTCP || WHICH == PAMD_SRGB_TO_CIELUV);
    __shared__ double glut[kLut ? 256 : 1];                // companding of the 256 possible 8-bit values (sRGB.c:70-89)
    pow_tables_to_lds();
    __syncthreads();
    if constexpr (kLut) {
        for (int b = threadIdx.x; b < 256; b += blockDim.x) glut[b] = dc::gamma_decode((double)b / 255.0);
        __syncthreads();
    }
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    auto finish = [&](const size_t i, double (&c)[3]) {
        dst[i] = c[0]; dst[n + i] = c[1]; dst[2 * n + i] = c[2];
#pragma unroll
        for (int p = 0; p < 3; p++) { mn[p] = fmin(mn[p], c[p]); mx[p] = fmax(mx[p], c[p]); bad |= !(fabs(c[p]) < 0x1.ffffffp127); }
        if (do_sum) {
#pragma unroll
            for (int p = 0; p < 3; p++) bin_add(c[p], sumk, acc[2 * p], acc[2 * p + 1]);
        }
        if (do_mom) {
            bin_add(c[0] * c[0], momk, acc2[0], acc2[1]); bin_add(c[1] * c[0], momk, acc2[2], acc2[3]); bin_add(c[2] * c[0], momk, acc2[4], acc2[5]);
            bin_add(c[1] * c[1], momk, acc2[6], acc2[7]); bin_add(c[2] * c[1], momk, acc2[8], acc2[9]); bin_add(c[2] * c[2], momk, acc2[10], acc2[11]);
        }
    };
    // the pow chains are VALU work (nine pow per pixel for ICtCp) during which nothing was in flight: the NEXT pixel's
    // components are requested before this pixel's chain starts
    size_t i = begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double nxt[3] = {0, 0, 0};
    unsigned nr = 0, ng = 0, nb = 0;
    if (i < end) { if constexpr (kLut) src.load_bytes(i, nr, ng, nb); else src.load(i, nxt); }
    for (; i < end; i += stride) {
        double c[3];
        if constexpr (kLut) { c[0] = glut[nr]; c[1] = glut[ng]; c[2] = glut[nb]; }
        else { c[0] = nxt[0]; c[1] = nxt[1]; c[2] = nxt[2]; }
        if (i + stride < end) { if constexpr (kLut) src.load_bytes(i + stride, nr, ng, nb); else src.load(i + stride, nxt); }
        if constexpr (kLut) dev_convert_linear<WHICH>(c); else dev_convert<WHICH>(c);
        finish(i, c);
    }
    if (do_sum) {                                          // block-uniform
        __shared__ double ssum[6 * 4];
        block_sum<6>(acc, ssum);
        if (threadIdx.x == 0) {
            const int slot = blockIdx.x & (kStatSlots - 1);
#pragma unroll
            for (int q = 0; q < 6; q++) unsafeAtomicAdd(&stats->sum[slot][q >> 1][q & 1], acc[q]);
        }
    }
    if (do_mom) {                                          // block-uniform
        __shared__ double smom[12 * 4];
        block_sum<12>(acc2, smom);
        if (threadIdx.x == 0) {
            const int slot = blockIdx.x & (kStatSlots - 1);
#pragma unroll
            for (int q = 0; q < 12; q++) unsafeAtomicAdd(&stats->mom[slot][q >> 1][q & 1], acc2[q]);
        }
    }
    if (stats) {
        if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&stats->nonfinite_f32, 1u);
        __shared__ double smn[4][3], smx[4][3];
#pragma unroll
        for (int p = 0; p < 3; p++) {
            double a = mn[p], b = mx[p];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { a = fmin(a, __shfl_down(a, o, 64)); b = fmax(b, __shfl_down(b, o, 64)); }
            if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6][p] = a; smx[threadIdx.x >> 6][p] = b; }
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int p = threadIdx.x;
            double a = smn[0][p], b = smx[0][p];
            for (int w = 1; w < 4; w++) { a = fmin(a, smn[w][p]); b = fmax(b, smx[w][p]); }
            if (a <= b) {   // skip blocks that saw no pixel (and NaNs)
                const int slot = blockIdx.x & (kStatSlots - 1);
                atomicMin(&stats->minkey[slot][p], f64_key(a));
                atomicMax(&stats->maxkey[slot][p], f64_key(b));
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_weight_stats(const double *__restrict__ w, size_t n, ConvertStats *stats) {
    double mx = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) mx = fmax(mx, fabs(w[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_down(mx, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(&stats->wmaxkey[blockIdx.x & (kStatSlots - 1)], f64_key(mx));
}

__global__ void k_init_stats(ConvertStats *s) {
    if (threadIdx.x < kStatSlots) {
        for (int p = 0; p < 3; p++) { s->minkey[threadIdx.x][p] = ~0ULL; s->maxkey[threadIdx.x][p] = 0ULL; s->sum[threadIdx.x][p][0] = 0.0; s->sum[threadIdx.x][p][1] = 0.0; }
        for (int q = 0; q < 6; q++) { s->mom[threadIdx.x][q][0] = 0.0; s->mom[threadIdx.x][q][1] = 0.0; }
        s->wmaxkey[threadIdx.x] = f64_key(0.0);
    }
    if (threadIdx.x == 0) s->nonfinite_f32 = 0u;
}

__global__ __launch_bounds__(256) void k_fill_uniform(double *out, size_t n, unsigned long long seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = u01(seed, i);
}
__global__ __launch_bounds__(256) void k_fill_weights(double *out, size_t n, unsigned long long seed) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = 1.0 + 3.0 * u01(seed, i);
}

static int stream_grid(size_t n) {
    size_t b = ceil_div(n, 256);
    if (b > 256 * 16) b = 256 * 16;      // 256 CUs x 16 blocks, grid-stride the rest
    if (b < 1) b = 1;
    return (int)b;
}

// begin / end: the pixels to convert (the whole image by default); init_stats: clear the statistics first (the first chunk)
void launch_convert(int which, const double *src_p, double *dst, size_t n, ConvertStats *stats, hipStream_t s, BinK sumk, BinK momk,
                    size_t begin, size_t end, bool init_stats) {
    if (end > n) end = n;
    if (stats && init_stats) hipLaunchKernelGGL(k_init_stats, 1, 64, 0, s, stats);
    int g = stream_grid(end - begin);
    KTIME("k_convert", s, 48.0 * (end - begin));
    const SrcF64 src{src_p, n};
    switch (which) {
        case PAMD_SRGB_TO_ICTCP: hipLaunchKernelGGL((k_convert<PAMD_SRGB_TO_ICTCP, SrcF64>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_SRGB_TO_CIELUV: hipLaunchKernelGGL((k_convert<PAMD_SRGB_TO_CIELUV, SrcF64>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_ICTCP_TO_REC2020: hipLaunchKernelGGL((k_convert<PAMD_ICTCP_TO_REC2020, SrcF64>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_CIELUV_TO_REC2020: hipLaunchKernelGGL((k_convert<PAMD_CIELUV_TO_REC2020, SrcF64>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_SRGB_TO_REC2020: hipLaunchKernelGGL((k_convert<PAMD_SRGB_TO_REC2020, SrcF64>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_REC2020_TO_SRGB: hipLaunchKernelGGL((k_convert<PAMD_REC2020_TO_SRGB, SrcF64>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_CIELUV_TO_ICTCP: hipLaunchKernelGGL((k_convert<PAMD_CIELUV_TO_ICTCP, SrcF64>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_COPY: hipLaunchKernelGGL((k_convert<PAMD_COPY, SrcF64>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        default: throw HipError("patolette_amd: unknown conversion");
    }
    HIP_CHECK(hipGetLastError());
}

void launch_convert_rows(int which, const double *rows, double *dst, size_t n, ConvertStats *stats, hipStream_t s, BinK sumk, BinK momk,
                         size_t begin, size_t end, bool init_stats) {
    if (end > n) end = n;
    if (stats && init_stats) hipLaunchKernelGGL(k_init_stats, 1, 64, 0, s, stats);
    int g = stream_grid(end - begin);
    KTIME("k_convert", s, 48.0 * (end - begin));
    const SrcF64Rows src{rows};
    switch (which) {
        case PAMD_SRGB_TO_ICTCP: hipLaunchKernelGGL((k_convert<PAMD_SRGB_TO_ICTCP, SrcF64Rows>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_SRGB_TO_CIELUV: hipLaunchKernelGGL((k_convert<PAMD_SRGB_TO_CIELUV, SrcF64Rows>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_COPY: hipLaunchKernelGGL((k_convert<PAMD_COPY, SrcF64Rows>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        default: throw HipError("patolette_amd: unknown conversion");
    }
    HIP_CHECK(hipGetLastError());
}

void launch_convert_u8(int which, const unsigned char *pixels, int channels, double *dst, size_t n, ConvertStats *stats,
                       hipStream_t s, BinK sumk, BinK momk, size_t begin, size_t end, bool init_stats) {
    if (end > n) end = n;
    if (stats && init_stats) hipLaunchKernelGGL(k_init_stats, 1, 64, 0, s, stats);
    int g = stream_grid(end - begin);
    KTIME("k_convert_u8", s, (24.0 + channels) * (end - begin));
    const SrcU8 src{pixels, channels};
    switch (which) {
        case PAMD_SRGB_TO_ICTCP: hipLaunchKernelGGL((k_convert<PAMD_SRGB_TO_ICTCP, SrcU8>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_SRGB_TO_CIELUV: hipLaunchKernelGGL((k_convert<PAMD_SRGB_TO_CIELUV, SrcU8>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        case PAMD_COPY: hipLaunchKernelGGL((k_convert<PAMD_COPY, SrcU8>), g, 256, 0, s, src, dst, n, stats, sumk, momk, begin, end); break;
        default: throw HipError("patolette_amd: unknown conversion");
    }
    HIP_CHECK(hipGetLastError());
}

// quantized image = palette_u8[map], interleaved RGB (README.md:184-191); 4 pixels (12 bytes) per thread
template <typename MapT>
__global__ __launch_bounds__(256) void k_reconstruct(const MapT *__restrict__ map, size_t n, const unsigned char *__restrict__ pal_u8,
                                                     int k, unsigned char *__restrict__ out) {
    extern __shared__ unsigned char spal[];
    for (int i = threadIdx.x; i < 3 * k; i += blockDim.x) spal[i] = pal_u8[i];
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t groups = n / 4;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        unsigned int w[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned int m = (unsigned int)map[4 * g + j];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const int b = 3 * j + c;
                w[b >> 2] |= (unsigned int)spal[3 * m + c] << (8 * (b & 3));
            }
        }
        unsigned int *o = reinterpret_cast<unsigned int *>(out + 12 * g);       // out is hipMalloc'd: 12*g is 4-byte aligned
        o[0] = w[0]; o[1] = w[1]; o[2] = w[2];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = groups * 4 + threadIdx.x;
        const unsigned int m = (unsigned int)map[i];
        for (int c = 0; c < 3; c++) out[3 * i + c] = spal[3 * m + c];
    }
}

void launch_reconstruct(const void *map, int map_elem, size_t n, const unsigned char *pal_u8, int k, unsigned char *out, hipStream_t s) {
    KTIME("k_reconstruct", s, (3.0 + map_elem) * n);
    const int g = stream_grid(ceil_div(n, 4));
    if (map_elem == 1) hipLaunchKernelGGL(k_reconstruct<unsigned char>, g, 256, 3 * k, s, (const unsigned char *)map, n, pal_u8, k, out);
    else hipLaunchKernelGGL(k_reconstruct<unsigned int>, g, 256, 3 * k, s, (const unsigned int *)map, n, pal_u8, k, out);
    HIP_CHECK(hipGetLastError());
}

__global__ __launch_bounds__(256) void k_pow(const double *__restrict__ x, double y, double *__restrict__ out, size_t n) {
    pow_tables_to_lds();
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = pamd_pow(x[i], y);
}
void launch_pow(const double *x, double y, double *out, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(k_pow, stream_grid(n), 256, 0, s, x, y, out, n);
    HIP_CHECK(hipGetLastError());
}

void launch_weight_stats(const double *w, size_t n, ConvertStats *stats, hipStream_t s) {
    KTIME("k_weight_stats", s, 8.0 * n);
    hipLaunchKernelGGL(k_weight_stats, stream_grid(n), 256, 0, s, w, n, stats);
    HIP_CHECK(hipGetLastError());
}

void launch_fill_image(double *d, size_t n, uint64_t seed, hipStream_t s) {
    for (int p = 0; p < 3; p++)
        hipLaunchKernelGGL(k_fill_uniform, stream_grid(n), 256, 0, s, d + (size_t)p * n, n, 1000ULL * seed + (unsigned long long)p);
    HIP_CHECK(hipGetLastError());
}
void launch_fill_weights(double *d, size_t n, uint64_t seed, hipStream_t s) {
    hipLaunchKernelGGL(k_fill_weights, stream_grid(n), 256, 0, s, d, n, 1000ULL * seed + 7ULL);
    HIP_CHECK(hipGetLastError());
}

}  // namespace pamd
