// kmeans.hip -- KMeans palette refinement on gfx950, replacing the reference's bundled faiss
// (lib/src/palette/refine.c:56-221 -> lib/faiss/faiss/Clustering.cpp:267-554).
//
// Bit-level contract (pinned against oracle/_ref/libref_faiss.so through the oracle):
//   assign   = IndexFlatL2 top-1 as the AVX2 fused kernel computes it
//              (utils/distances_fused/simdlib_based.cpp:59-221): t_j = fma(-2x2,y2, fma(-2x1,y1,
//              (-2x0)*y0)) + |y_j|^2, eight strict-'<' trackers over j = lane (mod 8), then
//              d = max(0, t + |x|^2), smaller d wins, equal d -> smaller index; scalar leftovers
//              for k % 8 != 0.  Norms: fma(v2,v2, fma(v0,v0, v1*v1)).
//   update   = compute_centroids (Clustering.cpp:135-204): per centroid, in SAMPLE ORDER,
//              c = fma(x, w, c) (weighted) or c += x; then c *= 1/h.  Sample order is kept by a
//              stable counting sort of the samples by assignment; one wavefront then replays the
//              sequential f32 chain of one centroid from coalesced 1-KiB loads.
//   split    = split_clusters (Clustering.cpp:216-263) with std::mt19937(1234), on one lane.
// f32 arithmetic, contraction off, FMAs only where written.  assign is 24 B/sample of HBM
// traffic and VALU-bound by brute force at k = 256; update is a latency chain.
#include "kmeans.h"

namespace pamd {

constexpr int kChunk = 2048;          // samples per wavefront in the counting sort

// ---- sample extraction: f64 planar -> f32 SoA (refine.c:127-163), optional subsample gather ----
template <bool W>
__global__ __launch_bounds__(256) void k_km_gather(const double *__restrict__ planar, size_t N, const int *__restrict__ perm,
                                                   size_t nx, KmSamples s) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += stride) {
        size_t p = perm ? (size_t)perm[i] : i;
        s.x[i] = (float)planar[p]; s.y[i] = (float)planar[N + p]; s.z[i] = (float)planar[2 * N + p];
        if constexpr (W) s.w[i] = (float)planar[3 * N + p];
    }
}

__global__ void k_km_prep(const float *__restrict__ cent, int k, float4 *c4) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < k) {
        float v0 = cent[3 * j], v1 = cent[3 * j + 1], v2 = cent[3 * j + 2];
        float n = __builtin_fmaf(v2, v2, __builtin_fmaf(v0, v0, v1 * v1));
        c4[j] = make_float4(v0, v1, v2, n);
    }
}

__global__ __launch_bounds__(256) void k_km_assign(KmSamples s, size_t nx, const float4 *__restrict__ c4, int k, int *__restrict__ assign) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nx) return;
    const float x0 = s.x[i], x1 = s.y[i], x2 = s.z[i];
    const float m0 = -2 * x0, m1 = -2 * x1, m2 = -2 * x2;
    const float xn = __builtin_fmaf(x2, x2, __builtin_fmaf(x0, x0, x1 * x1));
    float ld[8]; unsigned li[8];
#pragma unroll
    for (int l = 0; l < 8; l++) { ld[l] = 3.402823466e+38F - xn; li[l] = 0u; }
    const int ny_p = (k / 8) * 8;
    for (int j = 0; j < ny_p; j += 8) {
#pragma unroll
        for (int l = 0; l < 8; l++) {
            const float4 y = c4[j + l];                    // wave-uniform address -> scalar load
            float dp = m0 * y.x;
            dp = __builtin_fmaf(m1, y.y, dp);
            dp = __builtin_fmaf(m2, y.z, dp);
            dp = dp + y.w;
            if (dp < ld[l]) { ld[l] = dp; li[l] = (unsigned)(j + l); }
        }
    }
    float cur_d = 3.402823466e+38F; unsigned cur_i = 0xFFFFFFFFu;
#pragma unroll
    for (int l = 0; l < 8; l++) {
        float cand = ld[l] + xn;
        if (cand < 0) cand = 0;
        if (cur_d > cand) { cur_d = cand; cur_i = li[l]; }
        else if (cur_d == cand && cur_i > li[l]) cur_i = li[l];
    }
    for (int j0 = ny_p; j0 < k; j0++) {                     // simdlib_based.cpp:201-216
        const float4 y = c4[j0];
        float dp = __builtin_fmaf(x2, y.z, __builtin_fmaf(x1, y.y, x0 * y.x));
        float d = xn + y.w - 2 * dp;
        if (d < 0) d = 0;
        if (cur_d > d) { cur_d = d; cur_i = (unsigned)j0; }
    }
    assign[i] = (int)cur_i;
}

// lanes holding the same key (within `valid`) -- nbits ballots
__device__ __forceinline__ unsigned long long match_mask(int key, int nbits, unsigned long long valid) {
    unsigned long long m = valid;
    for (int b = 0; b < nbits; b++) {
        const bool bit = (key >> b) & 1;
        unsigned long long bm = __ballot(bit);
        m &= bit ? bm : ~bm;
    }
    return m;
}

// ---- stable counting sort by assignment: one wavefront owns a chunk of consecutive samples ----
__global__ __launch_bounds__(256) void k_km_count(const int *__restrict__ assign, size_t nx, int k, int nbits, int nchunks,
                                                  unsigned int *table /* [k][nchunks] */) {
    extern __shared__ unsigned int lds_u[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned int *cnt = lds_u + (size_t)wid * k;
    const int chunk = blockIdx.x * 4 + wid;
    for (int j = lane; j < k; j += 64) cnt[j] = 0u;
    if (chunk >= nchunks) return;
    const size_t lo = (size_t)chunk * kChunk;
    const size_t hi = lo + kChunk < nx ? lo + kChunk : nx;
    for (size_t base = lo; base < hi; base += 64) {
        const size_t i = base + lane;
        const bool v = i < hi;
        const int a = v ? assign[i] : 0;
        const unsigned long long valid = __ballot(v);
        const unsigned long long m = match_mask(a, nbits, valid);
        if (v && (m & ((1ULL << lane) - 1ULL)) == 0ULL) cnt[a] += (unsigned)__popcll(m);   // group leader; distinct addresses
    }
    for (int j = lane; j < k; j += 64) table[(size_t)j * nchunks + chunk] = cnt[j];
}

// per centroid: exclusive scan of its row of chunk counts (in place) + row total
__global__ __launch_bounds__(256) void k_km_rowscan(unsigned int *table, int nchunks, unsigned int *rowtot) {
    __shared__ unsigned int sw[4];
    __shared__ unsigned int carry;
    unsigned int *row = table + (size_t)blockIdx.x * nchunks;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int c0 = 0; c0 < nchunks; c0 += 256) {
        int c = c0 + threadIdx.x;
        unsigned v = c < nchunks ? row[c] : 0u, inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) sw[wid] = inc;
        __syncthreads();
        unsigned pre = 0;
        for (int w = 0; w < wid; w++) pre += sw[w];
        const unsigned cr = carry;
        if (c < nchunks) row[c] = cr + pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = cr + pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) rowtot[blockIdx.x] = carry;
}

__global__ __launch_bounds__(256) void k_km_base(const unsigned int *__restrict__ rowtot, int k, unsigned long long *rowbase /* k+1 */) {
    __shared__ unsigned long long su[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0ULL;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int c0 = 0; c0 < k; c0 += 256) {
        int c = c0 + threadIdx.x;
        unsigned long long v = c < k ? (unsigned long long)rowtot[c] : 0ULL, inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned long long t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) su[wid] = inc;
        __syncthreads();
        unsigned long long pre = 0;
        for (int w = 0; w < wid; w++) pre += su[w];
        const unsigned long long cr = carry;
        if (c < k) rowbase[c] = cr + pre + inc - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = cr + pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) rowbase[k] = carry;
}

template <bool W>
__global__ __launch_bounds__(256) void k_km_scatter(KmSamples s, const int *__restrict__ assign, size_t nx, int k, int nbits, int nchunks,
                                                    const unsigned int *__restrict__ table, const unsigned long long *__restrict__ rowbase,
                                                    float4 *sorted) {
    extern __shared__ unsigned int lds_u[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned int *cnt = lds_u + (size_t)wid * k;
    const int chunk = blockIdx.x * 4 + wid;
    for (int j = lane; j < k; j += 64) cnt[j] = 0u;
    if (chunk >= nchunks) return;
    const size_t lo = (size_t)chunk * kChunk;
    const size_t hi = lo + kChunk < nx ? lo + kChunk : nx;
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    for (size_t base = lo; base < hi; base += 64) {
        const size_t i = base + lane;
        const bool v = i < hi;
        const int a = v ? assign[i] : 0;
        const unsigned long long valid = __ballot(v);
        const unsigned long long m = match_mask(a, nbits, valid);
        unsigned run = 0;
        if (v) run = cnt[a];                                       // read before the leader bumps it
        const unsigned r = (unsigned)__popcll(m & lt);
        if (v) {
            const size_t dst = (size_t)rowbase[a] + table[(size_t)a * nchunks + chunk] + run + r;
            float w = 1.0f;
            if constexpr (W) w = s.w[i];
            sorted[dst] = make_float4(s.x[i], s.y[i], s.z[i], w);
        }
        if (v && r == 0) cnt[a] = run + (unsigned)__popcll(m);     // leader
    }
}

// ---- centroid update: one wavefront replays one centroid's sequential f32 chain ----
template <bool W>
__global__ __launch_bounds__(64) void k_km_update(const float4 *__restrict__ sorted, const unsigned long long *__restrict__ rowbase,
                                                  float *cent, float *hassign) {
    const int kidx = blockIdx.x, lane = threadIdx.x;
    const size_t lo = (size_t)rowbase[kidx], hi = (size_t)rowbase[kidx + 1];
    float c0 = 0.f, c1 = 0.f, c2 = 0.f, h = 0.f;
    float4 cur = make_float4(0, 0, 0, 0);
    if (lo + lane < hi) cur = sorted[lo + lane];
    for (size_t base = lo; base < hi; base += 64) {
        float4 nxt = make_float4(0, 0, 0, 0);
        if (base + 64 + lane < hi) nxt = sorted[base + 64 + lane];         // prefetch the next 1 KiB
        const int cnt = (int)(hi - base < 64 ? hi - base : 64);
        if (cnt == 64) {
#pragma unroll
            for (int t = 0; t < 64; t++) {
                const float x = __shfl(cur.x, t, 64), y = __shfl(cur.y, t, 64), z = __shfl(cur.z, t, 64);
                if constexpr (W) {
                    const float w = __shfl(cur.w, t, 64);
                    h += w;
                    c0 = __builtin_fmaf(x, w, c0); c1 = __builtin_fmaf(y, w, c1); c2 = __builtin_fmaf(z, w, c2);
                } else {
                    h += 1.0f;
                    c0 += x; c1 += y; c2 += z;
                }
            }
        } else {
            for (int t = 0; t < cnt; t++) {
                const float x = __shfl(cur.x, t, 64), y = __shfl(cur.y, t, 64), z = __shfl(cur.z, t, 64);
                if constexpr (W) {
                    const float w = __shfl(cur.w, t, 64);
                    h += w;
                    c0 = __builtin_fmaf(x, w, c0); c1 = __builtin_fmaf(y, w, c1); c2 = __builtin_fmaf(z, w, c2);
                } else {
                    h += 1.0f;
                    c0 += x; c1 += y; c2 += z;
                }
            }
        }
        cur = nxt;
    }
    if (lane == 0) {
        if (h != 0.f) { const float norm = 1 / h; c0 *= norm; c1 *= norm; c2 *= norm; }
        cent[3 * kidx] = c0; cent[3 * kidx + 1] = c1; cent[3 * kidx + 2] = c2;
        hassign[kidx] = h;
    }
}

// ---- split_clusters (Clustering.cpp:216-263) on one lane, std::mt19937(1234) per call ----
struct DevMT {
    unsigned mt[624]; int idx;
    __device__ void seed(unsigned s) {
        mt[0] = s;
        for (int i = 1; i < 624; i++) mt[i] = 1812433253U * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i;
        idx = 624;
    }
    __device__ unsigned next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; i++) {
                unsigned y = (mt[i] & 0x80000000U) | (mt[(i + 1) % 624] & 0x7fffffffU);
                unsigned v = mt[(i + 397) % 624] ^ (y >> 1);
                if (y & 1U) v ^= 0x9908b0dfU;
                mt[i] = v;
            }
            idx = 0;
        }
        unsigned y = mt[idx++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680U; y ^= (y << 15) & 0xefc60000U; y ^= (y >> 18);
        return y;
    }
};

__global__ void k_km_split(float *cent, float *hassign, int k, unsigned long long n, DevMT *scratch, int *nsplit_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    bool any = false;
    for (int ci = 0; ci < k; ci++) if (hassign[ci] == 0.f) { any = true; break; }
    if (!any) { if (nsplit_out) *nsplit_out = 0; return; }
    DevMT &rng = *scratch;
    rng.seed(1234u);
    int nsplit = 0;
    for (int ci = 0; ci < k; ci++) {
        if (hassign[ci] == 0.f) {
            int cj;
            for (cj = 0; true; cj = (cj + 1) % k) {
                float p = (float)(((double)hassign[cj] - 1.0) / (double)(float)(n - (unsigned long long)k));
                float r = (float)rng.next() / 4294967296.0f;
                if (r < p) break;
            }
            for (int j = 0; j < 3; j++) cent[ci * 3 + j] = cent[cj * 3 + j];
            for (int j = 0; j < 3; j++) {
                if (j % 2 == 0) {
                    cent[ci * 3 + j] = (float)((double)cent[ci * 3 + j] * (1 + (1 / 1024.)));
                    cent[cj * 3 + j] = (float)((double)cent[cj * 3 + j] * (1 - (1 / 1024.)));
                } else {
                    cent[ci * 3 + j] = (float)((double)cent[ci * 3 + j] * (1 - (1 / 1024.)));
                    cent[cj * 3 + j] = (float)((double)cent[cj * 3 + j] * (1 + (1 / 1024.)));
                }
            }
            hassign[ci] = hassign[cj] / 2;
            hassign[cj] -= hassign[ci];
            nsplit++;
        }
    }
    if (nsplit_out) *nsplit_out = nsplit;
}

// --------------------------------------------------------------------------------------------
void KMeansWork::reserve(size_t nx, int k) {
    sx.reserve(nx); sy.reserve(nx); sz.reserve(nx); sw.reserve(nx);
    assign.reserve(nx); sorted.reserve(nx);
    int nchunks = (int)ceil_div(nx, kChunk);
    table.reserve((size_t)k * (size_t)(nchunks > 0 ? nchunks : 1));
    rowtot.reserve(k); rowbase.reserve(k + 1);
    cent.reserve(3 * (size_t)k); hassign.reserve(k); c4.reserve(k);
    perm.reserve(nx);
    if (!mt.p) mt.reserve(1);
}

void kmeans_gather(const double *d_planar, size_t N, bool weighted, const int *d_perm, size_t nx, KMeansWork &w, hipStream_t s) {
    KmSamples ks{w.sx.p, w.sy.p, w.sz.p, w.sw.p};
    size_t g = ceil_div(nx, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    KTIME("k_km_gather", s, (weighted ? 48.0 : 36.0) * nx);
    if (weighted) hipLaunchKernelGGL(k_km_gather<true>, (int)g, 256, 0, s, d_planar, N, d_perm, nx, ks);
    else hipLaunchKernelGGL(k_km_gather<false>, (int)g, 256, 0, s, d_planar, N, d_perm, nx, ks);
    HIP_CHECK(hipGetLastError());
}

void kmeans_iterate(KMeansWork &w, size_t nx, int k, bool weighted, int niter, hipStream_t s) {
    KmSamples ks{w.sx.p, w.sy.p, w.sz.p, w.sw.p};
    int nbits = 0;
    while ((1 << nbits) < k) nbits++;
    const int nchunks = (int)ceil_div(nx, kChunk);
    const int cblocks = (nchunks + 3) / 4;
    const size_t lds = (size_t)4 * k * sizeof(unsigned int);
    static bool attr = false;
    if (!attr) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_count, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKMeansMaxK * 4));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKMeansMaxK * 4));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kKMeansMaxK * 4));
        attr = true;
    }
    for (int it = 0; it < niter; it++) {
        { KTIME("k_km_prep", s, 28.0 * k); hipLaunchKernelGGL(k_km_prep, (k + 255) / 256, 256, 0, s, w.cent.p, k, w.c4.p); }
        { KTIME("k_km_assign", s, 16.0 * nx); hipLaunchKernelGGL(k_km_assign, (int)ceil_div(nx, 256), 256, 0, s, ks, nx, w.c4.p, k, w.assign.p); }
        { KTIME("k_km_count", s, 4.0 * nx); hipLaunchKernelGGL(k_km_count, cblocks, 256, lds, s, w.assign.p, nx, k, nbits, nchunks, w.table.p); }
        { KTIME("k_km_rowscan", s, 8.0 * k * nchunks); hipLaunchKernelGGL(k_km_rowscan, k, 256, 0, s, w.table.p, nchunks, w.rowtot.p); }
        { KTIME("k_km_base", s, 12.0 * k); hipLaunchKernelGGL(k_km_base, 1, 256, 0, s, w.rowtot.p, k, w.rowbase.p); }
        {
            KTIME("k_km_scatter", s, (weighted ? 36.0 : 32.0) * nx);
            if (weighted) hipLaunchKernelGGL(k_km_scatter<true>, cblocks, 256, lds, s, ks, w.assign.p, nx, k, nbits, nchunks, w.table.p, w.rowbase.p, w.sorted.p);
            else hipLaunchKernelGGL(k_km_scatter<false>, cblocks, 256, lds, s, ks, w.assign.p, nx, k, nbits, nchunks, w.table.p, w.rowbase.p, w.sorted.p);
        }
        {
            KTIME("k_km_update", s, 16.0 * nx);
            if (weighted) hipLaunchKernelGGL(k_km_update<true>, k, 64, 0, s, w.sorted.p, w.rowbase.p, w.cent.p, w.hassign.p);
            else hipLaunchKernelGGL(k_km_update<false>, k, 64, 0, s, w.sorted.p, w.rowbase.p, w.cent.p, w.hassign.p);
        }
        { KTIME("k_km_split", s, 16.0 * k); hipLaunchKernelGGL(k_km_split, 1, 64, 0, s, w.cent.p, w.hassign.p, k, (unsigned long long)nx, w.mt.p, (int *)nullptr); }
    }
    HIP_CHECK(hipGetLastError());
}

}  // namespace pamd
