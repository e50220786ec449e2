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
//              sequential f32 chain of one centroid from coalesced 1-KiB loads staged in LDS.
//   split    = split_clusters (Clustering.cpp:216-263) with std::mt19937(1234), on one lane.
// f32 arithmetic, contraction off, FMAs only where written.
//
// One Lloyd iteration = 4 launches on one stream, no host round trip:
//   k_km_assign_count  16 B/sample (12 R + 4 W), VALU-bound by brute force at k = 256
//   k_km_rowscan       per-centroid scan of the chunk-count table
//   k_km_scatter       32-36 B/sample
//   k_km_update        16 B/sample; latency chain per centroid; the last wavefront to finish
//                      (agent-scope release/acquire ticket) handles empty clusters and prepares
//                      (y, |y|^2) for the next iteration.
#include "kmeans.h"

namespace pamd {

typedef __attribute__((address_space(1))) const void gvoid_t;
typedef __attribute__((address_space(3))) void lvoid_t;

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

// one image over several GPUs: this GPU holds pixels [begin, begin + n_local) of the image; samples outside it are left as zero
// bits (the group then SUMs the bit patterns as integers: every element has exactly one non-zero addend)
template <bool W>
__global__ __launch_bounds__(256) void k_km_gather_slice(const double *__restrict__ planar, size_t n_local, const int *__restrict__ perm,
                                                         size_t nx, size_t begin, KmSamples s) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += stride) {
        const size_t g = perm ? (size_t)perm[i] : i;
        const bool mine = g >= begin && g - begin < n_local;
        const size_t p = mine ? g - begin : 0;
        s.x[i] = mine ? (float)planar[p] : 0.0f; s.y[i] = mine ? (float)planar[n_local + p] : 0.0f;
        s.z[i] = mine ? (float)planar[2 * n_local + p] : 0.0f;
        if constexpr (W) s.w[i] = mine ? (float)planar[3 * n_local + p] : 0.0f;
    }
}

__device__ __forceinline__ float4 make_c4(float v0, float v1, float v2) {
    return make_float4(v0, v1, v2, __builtin_fmaf(v2, v2, __builtin_fmaf(v0, v0, v1 * v1)));
}

// The centroid table the assignment kernels read: c4[j] = (y0, y1, y2, |y|^2), and behind it (c4 + k) the same numbers pair by
// pair -- entry p = {y0 of 2p, y0 of 2p+1, y1.., y1.., y2.., y2.., |y|^2.., |y|^2..} -- for the packed-f32 full scan.
__device__ __forceinline__ void km_store_c4(float4 *c4, const int k, const int j, const float v0, const float v1, const float v2) {
    // Write-through (agent-scope) stores: in the update kernels every centroid's block writes its own record, and after
    // split_clusters the last block writes records again that another block -- possibly on another XCD, behind another L2 --
    // wrote moments before.  Plain stores would leave two dirty copies of one line whose write-back order at the end of the
    // kernel decides what the next assignment reads; these go to memory in program order (each block drains its stores before
    // it takes its ticket, the last block writes after the last ticket).
    const float4 r = make_c4(v0, v1, v2);
    float *q = reinterpret_cast<float *>(c4 + j);
    __hip_atomic_store(q, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 2, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 3, r.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float *pp = reinterpret_cast<float *>(c4 + k) + 8 * (j >> 1) + (j & 1);
    __hip_atomic_store(pp, r.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pp + 2, r.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pp + 4, r.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(pp + 6, r.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// flags (kKmFlagWords words, the list path of the default 512^2 samples): [0] = the iterations have reached a fixed point (no sample
// changed its centroid in the last assignment and no cluster was re-seeded: every later iteration reproduces the same centroids bit
// for bit, so the remaining launches return at once); [1 + j] = centroid j is DIRTY -- a sample joined or left it in the last
// assignment, or split_clusters touched it after the last update.  A clean centroid's members are the same samples in the same
// order as one iteration ago: its sequential f32 sums come out the same, and its block skips collecting them.
constexpr int kKmFlagWords = 2 + 256;
__global__ void k_km_prep(const float *__restrict__ cent, int k, float4 *c4, unsigned int *flags) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < k) km_store_c4(c4, k, j, cent[3 * j], cent[3 * j + 1], cent[3 * j + 2]);
    if (flags && j < kKmFlagWords) flags[j] = j == 0 ? 0u : 1u;         // nothing is known about the first assignment: all dirty
}

// top-1 of one sample against all centroids, exactly as the AVX2 fused kernel orders it
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f4_t *scalar_c4_t;      // uniform index -> s_load: the 4 KB table sits in the scalar cache

template <typename TabT>
__device__ __forceinline__ int km_assign_one(const float x0, const float x1, const float x2, const TabT c4, const int k) {
    const float m0 = -2 * x0, m1 = -2 * x1, m2 = -2 * x2;
    const float xn = __builtin_fmaf(x2, x2, __builtin_fmaf(x0, x0, x1 * x1));
    float ld[8]; unsigned li[8];
    int lb[8];                                             // group (j) of the lane's current minimum: index = group + lane
#pragma unroll
    for (int l = 0; l < 8; l++) { ld[l] = 3.402823466e+38F - xn; lb[l] = -l; }
    const int ny_p = (k / 8) * 8;
    for (int j = 0; j < ny_p; j += 8) {
#pragma unroll
        for (int l = 0; l < 8; l++) {
            const auto y = c4[j + l];                      // wave-uniform address: LDS broadcast read or scalar load
            float dp = m0 * y.x;
            dp = __builtin_fmaf(m1, y.y, dp);
            dp = __builtin_fmaf(m2, y.z, dp);
            dp = dp + y.w;
            if (dp < ld[l]) { ld[l] = dp; lb[l] = j; }     // one copy of the (uniform) j serves the eight lanes
        }
    }
#pragma unroll
    for (int l = 0; l < 8; l++) li[l] = (unsigned)(lb[l] + l);
    float cur_d = 3.402823466e+38F; unsigned cur_i = 0xFFFFFFFFu;
#pragma unroll
    for (int l = 0; l < 8; l++) {
        float cand = ld[l] + xn;
        if (cand < 0) cand = 0;
        if (cur_d > cand) { cur_d = cand; cur_i = li[l]; }
        else if (cur_d == cand && cur_i > li[l]) cur_i = li[l];
    }
    for (int j0 = ny_p; j0 < k; j0++) {                     // simdlib_based.cpp:201-216
        const auto y = c4[j0];
        float dp = __builtin_fmaf(x2, y.z, __builtin_fmaf(x1, y.y, x0 * y.x));
        float d = xn + y.w - 2 * dp;
        if (d < 0) d = 0;
        if (cur_d > d) { cur_d = d; cur_i = (unsigned)j0; }
    }
    return (int)cur_i;
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

// ---- assignment + first half of the stable counting sort: one wavefront owns a chunk of consecutive
// samples, assigns them 64 at a time and counts per centroid in wave-private LDS ----
// BIG (k > kKMeansMaxK, no limit): the wave-private counters do not fit LDS; the wavefront counts straight into its column of
// the (zeroed) table with L2 atomics -- the column is its own, the atomics only keep the vector L1 out of the way.
template <bool BIG>
__global__ __launch_bounds__(256) void k_km_assign_count(KmSamples s, size_t nx, const float4 *__restrict__ c4, int k, int nbits,
                                                         int chunk_len, int nchunks, int *__restrict__ assign,
                                                         unsigned int *table /* [k][nchunks] */) {
    extern __shared__ unsigned int lds_u[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // the centroid table (k x 16 bytes, read-only here) is read through the scalar cache: every lane needs the same
    // entry at the same time, so the operands arrive in SGPRs and no LDS traffic sits between the FMAs
    unsigned int *cnt = lds_u + 4 * (size_t)k + (size_t)wid * k;
    if constexpr (!BIG) {
        for (int j = lane; j < k; j += 64) cnt[j] = 0u;
        __syncthreads();
    }
    const int chunk = blockIdx.x * 4 + wid;
    if (chunk >= nchunks) return;
    const size_t lo = (size_t)chunk * chunk_len;
    const size_t hi = lo + chunk_len < nx ? lo + chunk_len : nx;
    for (size_t base = lo; base < hi; base += 64) {
        const size_t i = base + lane;
        const bool v = i < hi;
        int a = 0;
        if (v) { a = km_assign_one(s.x[i], s.y[i], s.z[i], (scalar_c4_t)(unsigned long long)c4, k); assign[i] = a; }
        const unsigned long long valid = __ballot(v);
        const unsigned long long m = match_mask(a, nbits, valid);
        if (v && (m & ((1ULL << lane) - 1ULL)) == 0ULL) {                                    // group leader; distinct addresses
            if constexpr (BIG) (void)atomicAdd(&table[(size_t)a * nchunks + chunk], (unsigned)__popcll(m));
            else cnt[a] += (unsigned)__popcll(m);
        }
    }
    if constexpr (!BIG) for (int j = lane; j < k; j += 64) table[(size_t)j * nchunks + chunk] = cnt[j];
}

// BIG: exclusive prefix of the row totals (where every centroid's segment of `sorted` starts), one block
__global__ __launch_bounds__(256) void k_km_rowbase(const unsigned int *__restrict__ rowtot, int k, unsigned int *rowbase) {
    __shared__ unsigned int wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int per = (k + 255) / 256;
    const int j0 = threadIdx.x * per, j1 = (j0 + per < k) ? j0 + per : k;
    unsigned sum = 0;
    for (int j = j0; j < j1; j++) sum += rowtot[j];
    unsigned inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    unsigned pre = 0;
    for (int w = 0; w < wid; w++) pre += wsum[w];
    unsigned run = pre + inc - sum;
    for (int j = j0; j < j1; j++) { rowbase[j] = run; run += rowtot[j]; }
}

// BIG: second half of the stable counting sort with the cursors in memory: after k_km_rowscan table[j][chunk] = members of
// centroid j in earlier chunks; the wavefront that owns the chunk advances that entry as it places its samples (one L2 atomic
// per group of equal assignments and trip of 64 samples, the old value handed to the group's lanes)
template <bool W>
__global__ __launch_bounds__(256) void k_km_scatter_big(KmSamples s, const int *__restrict__ assign, size_t nx, int nbits, int chunk_len,
                                                        int nchunks, unsigned int *table, const unsigned int *__restrict__ rowbase,
                                                        float4 *sorted) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int chunk = blockIdx.x * 4 + wid;
    if (chunk >= nchunks) return;
    const size_t lo = (size_t)chunk * chunk_len;
    const size_t hi = lo + chunk_len < nx ? lo + chunk_len : nx;
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    for (size_t base = lo; base < hi; base += 64) {
        const size_t i = base + lane;
        const bool v = i < hi;
        const int a = v ? assign[i] : 0;
        const unsigned long long valid = __ballot(v);
        const unsigned long long m = match_mask(a, nbits, valid);
        const unsigned r = (unsigned)__popcll(m & lt);
        unsigned cur = 0;
        if (v && r == 0) cur = atomicAdd(&table[(size_t)a * nchunks + chunk], (unsigned)__popcll(m));
        const int leader = (v && m) ? __ffsll((long long)m) - 1 : lane;
        cur = (unsigned)__shfl((int)cur, leader, 64);
        if (v) {
            float w = 1.0f;
            if constexpr (W) w = s.w[i];
            sorted[(size_t)rowbase[a] + cur + r] = make_float4(s.x[i], s.y[i], s.z[i], w);
        }
    }
}

// per centroid: exclusive scan of its row of chunk counts (in place) + row total
__global__ __launch_bounds__(256) void k_km_rowscan(unsigned int *table, int nchunks, unsigned int *rowtot) {
    // exclusive prefix of one row of the count table; 16 consecutive entries per thread (4 x 16-byte loads), one block
    // scan of the thread totals per 4096 entries
    constexpr int IT = 16;
    __shared__ unsigned int sw[4];
    __shared__ unsigned int carry;
    unsigned int *row = table + (size_t)blockIdx.x * nchunks;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int c0 = 0; c0 < nchunks; c0 += 256 * IT) {
        const int base = c0 + (int)threadIdx.x * IT;
        unsigned v[IT];
        if (base + IT <= nchunks && (((size_t)blockIdx.x * nchunks + base) & 3) == 0) {
#pragma unroll
            for (int q = 0; q < IT / 4; q++) {
                const uint4 u = *reinterpret_cast<const uint4 *>(row + base + 4 * q);
                v[4 * q] = u.x; v[4 * q + 1] = u.y; v[4 * q + 2] = u.z; v[4 * q + 3] = u.w;
            }
        } else {
#pragma unroll
            for (int q = 0; q < IT; q++) v[q] = base + q < nchunks ? row[base + q] : 0u;
        }
        unsigned tot = 0;
#pragma unroll
        for (int q = 0; q < IT; q++) { const unsigned x = v[q]; v[q] = tot; tot += x; }      // exclusive within the thread
        unsigned inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) sw[wid] = inc;
        __syncthreads();
        unsigned pre = 0;
        for (int w = 0; w < wid; w++) pre += sw[w];
        const unsigned cr = carry;
        const unsigned off = cr + pre + inc - tot;
#pragma unroll
        for (int q = 0; q < IT; q++) if (base + q < nchunks) row[base + q] = off + v[q];
        __syncthreads();
        if (threadIdx.x == 255) carry = cr + pre + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) rowtot[blockIdx.x] = carry;
}

// --------------------------------------------------------------------------------------------
// Exact candidate pruning for the assignment when many samples are clustered (kmeans_max_samples large).
// Brute force costs 256 x 8 instructions per sample.  A G^3 grid over the samples' bounding box stores, per cell, the
// centroids that can win anywhere in the cell: mindist^2(cell, c_j) <= min_k maxdist^2(cell, c_k) + M, where M covers
// four times the worst rounding error of the f32 expression the reference evaluates (|computed - true| <=
// 16 * 2^-24 * (|x| + |c|)^2), so every centroid whose COMPUTED distance can equal or beat the winner's is in the list.
// The survivors go through exactly the arithmetic of km_assign_one: the list is ordered by (j mod 8, j), i.e. by the
// SIMD lane the AVX2 kernel would have held them in, each lane's strict-'<' minimum is taken in ascending j, the lanes
// are merged with the (distance, index) rule, then the scalar leftovers (j >= 8*(k/8)) follow.  Centroids not in the
// list lose every comparison they would take part in, so dropping them does not change the result.  Cells with more
// than 15 candidates fall back to the full scan.  Rebuilt every iteration (the centroids move).
// --------------------------------------------------------------------------------------------
struct KmGridDev { float lo[3], hi[3]; };

__device__ __forceinline__ unsigned f32_key(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

constexpr int kKmSlots = 32;
__global__ __launch_bounds__(256) void k_km_bounds(KmSamples s, size_t nx, unsigned int *keys /* [kKmSlots][6] */) {
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nx; i += stride) {
        const float v[3] = {s.x[i], s.y[i], s.z[i]};
#pragma unroll
        for (int a = 0; a < 3; a++) { mn[a] = fminf(mn[a], v[a]); mx[a] = fmaxf(mx[a], v[a]); }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        float lo = mn[a], hi = mx[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_down(lo, o, 64)); hi = fmaxf(hi, __shfl_down(hi, o, 64)); }
        if ((threadIdx.x & 63) == 0 && lo <= hi) {
            unsigned int *k6 = keys + (blockIdx.x & (kKmSlots - 1)) * 6;
            atomicMin(&k6[a], f32_key(lo)); atomicMax(&k6[3 + a], f32_key(hi));
        }
    }
}
__global__ void k_km_bounds_init(unsigned int *keys) {
    const int t = threadIdx.x;
    if (t < kKmSlots * 6) keys[t] = (t % 6) < 3 ? 0xFFFFFFFFu : 0u;
}
__global__ void k_km_bounds_fold(const unsigned int *keys, KmGridDev *g) {
    const int a = threadIdx.x;
    if (a >= 3) return;
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (int sl = 0; sl < kKmSlots; sl++) { lo = min(lo, keys[sl * 6 + a]); hi = max(hi, keys[sl * 6 + 3 + a]); }
    g->lo[a] = key_f32(lo); g->hi[a] = key_f32(hi);
}

__device__ __forceinline__ int km_cell(float x0, float x1, float x2, const KmGridDev &g, int G) {
    int idx[3];
    const float v[3] = {x0, x1, x2};
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double r = (double)g.hi[a] - (double)g.lo[a];
        const double inv = r > 0 ? (double)G / r : 0.0;
        int q = (int)(((double)v[a] - (double)g.lo[a]) * inv);
        idx[a] = q < 0 ? 0 : (q >= G ? G - 1 : q);
    }
    return (idx[2] * G + idx[1]) * G + idx[0];
}

// The rule of one cell: threshold from the box [cl, ch] (already widened) and the list of entries to consider.
struct KmBox { double cl[3], ch[3], cellnorm2; };
__device__ __forceinline__ KmBox km_box(const KmGridDev &g, int G, const int idx[3], int span) {
    KmBox b; b.cellnorm2 = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        // every box is widened by 1e-5 of the range: k_km_assign_mid finds its cell in f32 (two roundings of 2^-24 and
        // a scale factor 1 - 2^-18, together < 3e-4 of a 32^3 cell = 1e-5 of the range); km_cell (f64) needs 1e-9
        const double r = (double)g.hi[a] - (double)g.lo[a], cw = r / G, m = 1e-5 * r + 1e-30;
        b.cl[a] = (double)g.lo[a] + idx[a] * cw - m;
        b.ch[a] = (double)g.lo[a] + (idx[a] + span) * cw + m;
        const double f = fmax(fabs(b.cl[a]), fabs(b.ch[a]));
        b.cellnorm2 += f * f;
    }
    return b;
}
__device__ __forceinline__ double km_maxd2(const KmBox &b, const float4 y) {
    const double p[3] = {(double)y.x, (double)y.y, (double)y.z};
    double mx = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) { const double d = fmax(fabs(p[a] - b.cl[a]), fabs(p[a] - b.ch[a])); mx += d * d; }
    return mx;
}
__device__ __forceinline__ double km_mind2(const KmBox &b, const float4 y) {
    const double p[3] = {(double)y.x, (double)y.y, (double)y.z};
    double mn = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) { const double d = fmax(fmax(b.cl[a] - p[a], p[a] - b.ch[a]), 0.0); mn += d * d; }
    return mn;
}
__device__ __forceinline__ double km_threshold(const KmBox &b, double U, double cn2) {
    const double rr = sqrt(b.cellnorm2) + sqrt(cn2);
    return U * (1.0 + 1e-12) + 4.0 * (16.0 * 0x1.0p-24) * rr * rr + 1e-300;
}

// Coarse pass: one wavefront per block of 4x4x4 cells, lanes across the centroids.  The survivors of the rule on the
// big box are a superset of the survivors on every cell inside (smaller minimum distance, larger upper bound, larger
// margin), so the fine pass tests only those.  The list is written in the order the fine pass needs: by SIMD lane
// (j mod 8) of the reference's kernel, ascending j inside a lane, scalar leftovers last.
constexpr int kKmCoarseMax = 96;
__global__ __launch_bounds__(64) void k_km_lut_coarse(const float4 *__restrict__ c4, int k, const KmGridDev *__restrict__ gp, int G,
                                                      unsigned char *__restrict__ clist /* [cells][2 + kKmCoarseMax]: count, flag, entries */,
                                                      double *__restrict__ cn2_out /* largest squared centroid norm (same from every block) */) {
    const KmGridDev g = *gp;
    const int Gc = G / 4, cell = (int)blockIdx.x, lane = (int)threadIdx.x;
    const int idx[3] = {4 * (cell % Gc), 4 * ((cell / Gc) % Gc), 4 * (cell / (Gc * Gc))};
    const KmBox b = km_box(g, G, idx, 4);
    double U = INFINITY, cn2 = 0;
    for (int j = lane; j < k; j += 64) {
        const float4 y = c4[j];
        U = fmin(U, km_maxd2(b, y));
        cn2 = fmax(cn2, ((double)y.x * y.x + (double)y.y * y.y) + (double)y.z * y.z);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { U = fmin(U, __shfl_xor(U, o, 64)); cn2 = fmax(cn2, __shfl_xor(cn2, o, 64)); }
    const double thr = km_threshold(b, U, cn2);
    if (lane == 0) *cn2_out = cn2;
    unsigned char *out = clist + (size_t)cell * (2 + kKmCoarseMax);
    int cnt = 0;                                                 // wave-uniform
    const int ny_p = (k / 8) * 8;
    auto sweep = [&](int j, bool valid) {
        const bool keep = valid && km_mind2(b, c4[valid ? j : 0]) <= thr;
        const unsigned long long m = __ballot(keep);
        const int pos = cnt + (int)__popcll(m & ((1ULL << lane) - 1ULL));
        if (keep && pos < kKmCoarseMax) out[2 + pos] = (unsigned char)j;
        cnt += (int)__popcll(m);
    };
    for (int l = 0; l < 8; l++)
        for (int q0 = 0; l + 8 * q0 < ny_p; q0 += 64) { const int j = l + 8 * (q0 + lane); sweep(j, j < ny_p); }
    for (int j0 = ny_p; j0 < k; j0 += 64) sweep(j0 + lane, j0 + lane < k);
    if (lane == 0) { out[0] = (unsigned char)(cnt <= kKmCoarseMax ? cnt : 0); out[1] = (unsigned char)(cnt <= kKmCoarseMax ? 0 : 1); }   // flag 1: test all
}

// --------------------------------------------------------------------------------------------
// Many samples (G = 64): like the palette map, the per-sample lookup is served from LDS.  A 32^3 table of four-byte
// entries (up to four candidates in the order the reference visits them: SIMD lane, then index, leftovers last; padded by
// repeating the last one, which changes nothing) is filled by the rule above plus a bisector test against q* = the
// centroid with the smallest maxdist: p is dropped when |x-p|^2 - |x-q*|^2 exceeds the rounding margin on the whole box,
// i.e. its COMPUTED distance is strictly larger than q*'s everywhere in the cell, so it loses every comparison that
// matters (it can never be the winner, and whatever it displaces inside its SIMD lane was no winner either).  Samples of
// cells with more than four survivors are parked (coordinates + index, 16 bytes) in a per-wavefront LDS queue and go
// through the 16-byte records of the G^3 table with full wavefronts.
// --------------------------------------------------------------------------------------------
constexpr int kKmMidSurv = 48;                                 // survivors of the first rule listed per cell of the 32^3 table
constexpr unsigned kKmMidOverflow = 0x00000001u;               // bytes {1, 0, 0, 0}: SIMD lane 1 before lane 0, impossible for a list
__device__ __forceinline__ bool km_mid_is_overflow(unsigned e) { return e == kKmMidOverflow; }

// One wavefront fills the eight cells of the 32^3 table inside one coarse block: lane = (cell << 3) | slice, the eight lanes
// of a cell share the list (slice s takes positions s, s + 8, ...); list order is kept through the ballots.
__device__ __forceinline__ void km_mid_entries(const float4 *__restrict__ c4, const int k, const KmGridDev &g, const int G,
                                               const unsigned char *__restrict__ cand, const double cn2, const int c0, const int c1, const int c2,
                                               unsigned int *__restrict__ mid) {
    const int lane = (int)threadIdx.x, m = lane >> 3, sl = lane & 7;
    const int midx[3] = {2 * c0 + (m & 1), 2 * c1 + ((m >> 1) & 1), 2 * c2 + (m >> 2)};
    const int idx[3] = {2 * midx[0], 2 * midx[1], 2 * midx[2]};
    const KmBox b = km_box(g, G, idx, 2);
    const bool all = cand[1] != 0;
    const int ntest = all ? k : (int)cand[0];
    const int ny_p = (k / 8) * 8;
    auto entry = [&](int t) -> int {
        if (!all) return (int)cand[2 + t];
        if (t >= ny_p) return t;
        const int per = ny_p / 8;
        return (t / per) + 8 * (t % per);
    };
    double U = INFINITY; int qs = 0x7fffffff;
    for (int t = sl; t < ntest; t += 8) { const int j = entry(t); const double mx = km_maxd2(b, c4[j]); if (mx < U) { U = mx; qs = j; } }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        const double u2 = __shfl_xor(U, o, 64); const int q2i = __shfl_xor(qs, o, 64);
        if (u2 < U || (u2 == U && q2i < qs)) { U = u2; qs = q2i; }
    }
    const double thr = km_threshold(b, U, cn2);
    const double margin = thr - U;                              // the absolute rounding margin of the rule (plus 1e-12 U)
    const float4 yq = c4[qs];
    // `far(p, o)`: the computed distance of entry p exceeds that of entry o on the whole box, beyond rounding: |x-p|^2 - |x-o|^2 is
    // linear in x, its minimum over the box is taken term by term
    auto far = [&](const float4 yp, const float4 yo) -> bool {
        const double p[3] = {(double)yp.x, (double)yp.y, (double)yp.z}, o[3] = {(double)yo.x, (double)yo.y, (double)yo.z};
        const double o2 = (o[0] * o[0] + o[1] * o[1]) + o[2] * o[2];
        double f = -o2, scale = o2;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double w = o[a] - p[a];
            f += 2.0 * fmin(b.cl[a] * w, b.ch[a] * w) + p[a] * p[a];
            const double big = fmax(fmax(fabs(b.cl[a]), fabs(b.ch[a])), fabs(p[a]));
            scale += 4.0 * big * big;
        }
        return f > margin + 1e-12 * scale;
    };
    // first rule + the bisector test against q*; the survivors of the cell are listed in LDS (list order kept)
    __shared__ unsigned char sv[8][kKmMidSurv];
    int ns = 0;                                                  // uniform over the eight lanes of a cell
    for (int t0 = 0; t0 < ntest; t0 += 8) {                      // wave-uniform trip count (ntest is)
        const int t = t0 + sl;
        bool keep = false; int j = 0;
        if (t < ntest) {
            j = entry(t);
            const float4 y = c4[j];
            keep = km_mind2(b, y) <= thr && !far(y, yq);
        }
        const unsigned bits = (unsigned)((__ballot(keep) >> (8 * m)) & 0xffULL);
        const int pos = ns + __popc(bits & ((1u << sl) - 1u));
        if (keep && pos < kKmMidSurv) sv[m][pos] = (unsigned char)j;
        ns += __popc(bits);
    }
    __builtin_amdgcn_wave_barrier();
    // second rule, pairwise: a survivor that is `far` from ANOTHER survivor everywhere in the cell can neither win nor tie there
    // (being farther is a strict partial order: whatever is dropped is beaten by something that stays), and whatever it would
    // displace inside its SIMD lane was no winner either.  Cells near a Voronoi vertex keep five or more; most keep <= 4.
    unsigned e = 0; int cnt = 0;
    const int nsv = ns <= kKmMidSurv ? ns : 0;
    int nmax = nsv;                                              // wave-uniform trip count
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
    for (int u0 = 0; u0 < nmax; u0 += 8) {
        const int u = u0 + sl;
        bool keep = false; int j = 0;
        if (u < nsv) {
            j = (int)sv[m][u];
            const float4 y = c4[j];
            keep = true;
            for (int v = 0; v < nsv; v++) { if (v != u && far(y, c4[sv[m][v]])) { keep = false; break; } }
        }
        const unsigned bits = (unsigned)((__ballot(keep) >> (8 * m)) & 0xffULL);
        const int pos = cnt + __popc(bits & ((1u << sl) - 1u));
        if (keep && pos < 4) e |= (unsigned)j << (8 * pos);
        cnt += __popc(bits);
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) e |= (unsigned)__shfl_xor((int)e, o, 64);
    if (cnt > 4 || ns > kKmMidSurv) e = kKmMidOverflow;
    else {
        const unsigned last = (e >> (8 * (cnt - 1))) & 0xffu;
        for (int t = cnt; t < 4; t++) e |= last << (8 * t);
    }
    const int Gm = G / 2;
    if (sl == 0) mid[(midx[2] * Gm + midx[1]) * Gm + midx[0]] = e;
}

// Fine pass: one wavefront per coarse block, lane = one of its 64 cells (uniform candidate list).
__global__ __launch_bounds__(64) void k_km_lut_build(const float4 *__restrict__ c4, int k, const KmGridDev *__restrict__ gp, int G,
                                                     const unsigned char *__restrict__ clist, const double *__restrict__ cn2_in,
                                                     unsigned char *__restrict__ lut, unsigned int *__restrict__ mid) {
    const KmGridDev g = *gp;
    const int Gc = G / 4, cc = (int)blockIdx.x, t64 = (int)threadIdx.x;
    const int idx[3] = {4 * (cc % Gc) + (t64 & 3), 4 * ((cc / Gc) % Gc) + ((t64 >> 2) & 3), 4 * (cc / (Gc * Gc)) + (t64 >> 4)};
    const int cell = (idx[2] * G + idx[1]) * G + idx[0];
    const KmBox b = km_box(g, G, idx, 1);
    const unsigned char *cand = clist + (size_t)cc * (2 + kKmCoarseMax);
    const bool all = cand[1] != 0;
    const int ntest = all ? k : (int)cand[0];
    const int ny_p = (k / 8) * 8;
    // position t of the "all" order: by SIMD lane, ascending inside, leftovers last
    auto entry = [&](int t) -> int {
        if (!all) return (int)cand[2 + t];
        if (t >= ny_p) return t;
        const int per = ny_p / 8;
        return (t / per) + 8 * (t % per);
    };
    double U = INFINITY; int qs = 0;
    const double cn2 = *cn2_in;                                   // over all centroids, from the coarse pass
    for (int t = 0; t < ntest; t++) { const int j = entry(t); const double m = km_maxd2(b, c4[j]); if (m < U) { U = m; qs = j; } }
    const double thr = km_threshold(b, U, cn2);
    const double margin = thr - U;
    const float4 yq = c4[qs];
    const double q[3] = {(double)yq.x, (double)yq.y, (double)yq.z};
    const double q2 = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    unsigned char rec[16];
    int cnt = 0, last = 0;
    for (int t = 0; t < ntest; t++) {
        const int j = entry(t);
        const float4 y = c4[j];
        if (km_mind2(b, y) > thr) continue;
        // second rule (see k_km_lut_mid): its computed distance exceeds q*'s on the whole box
        const double p[3] = {(double)y.x, (double)y.y, (double)y.z};
        double f = -q2, scale = q2;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double w = q[a] - p[a];
            f += 2.0 * fmin(b.cl[a] * w, b.ch[a] * w) + p[a] * p[a];
            const double big = fmax(fmax(fabs(b.cl[a]), fabs(b.ch[a])), fabs(p[a]));
            scale += 4.0 * big * big;
        }
        if (f > margin + 1e-12 * scale) continue;
        if (cnt < 15) rec[1 + cnt] = (unsigned char)j;
        last = j;
        cnt++;
    }
    rec[0] = (unsigned char)(cnt <= 15 ? cnt : 255);
    for (int t = cnt < 15 ? cnt + 1 : 16; t < 16; t++) rec[t] = (unsigned char)last;   // padding repeats the last entry
    uint4 out;
    out.x = rec[0] | (rec[1] << 8) | (rec[2] << 16) | ((unsigned)rec[3] << 24);
    out.y = rec[4] | (rec[5] << 8) | (rec[6] << 16) | ((unsigned)rec[7] << 24);
    out.z = rec[8] | (rec[9] << 8) | (rec[10] << 16) | ((unsigned)rec[11] << 24);
    out.w = rec[12] | (rec[13] << 8) | (rec[14] << 16) | ((unsigned)rec[15] << 24);
    reinterpret_cast<uint4 *>(lut)[cell] = out;
    if (mid != nullptr) km_mid_entries(c4, k, g, G, cand, cn2, cc % Gc, (cc / Gc) % Gc, cc / (Gc * Gc), mid);
}

__device__ __forceinline__ int km_assign_pruned(const float x0, const float x1, const float x2, const float4 *c4, const int k, const uint4 rec) {
    unsigned long long w0 = ((unsigned long long)rec.y << 32) | rec.x, w1 = ((unsigned long long)rec.w << 32) | rec.z;
    const int cnt = (int)(w0 & 0xffULL);
    if (cnt == 255) return km_assign_one(x0, x1, x2, c4, k);
    w0 >>= 8;
    int left = 7;
    const float m0 = -2 * x0, m1 = -2 * x1, m2 = -2 * x2;
    const float xn = __builtin_fmaf(x2, x2, __builtin_fmaf(x0, x0, x1 * x1));
    const int ny_p = (k / 8) * 8;
    float cur_d = 3.402823466e+38F; unsigned cur_i = 0xFFFFFFFFu;
    int curl = -1;
    float trd = 0.f; unsigned tri = 0u;
    auto merge = [&]() {
        float cand = trd + xn;
        if (cand < 0) cand = 0;
        if (cur_d > cand) { cur_d = cand; cur_i = tri; }
        else if (cur_d == cand && cur_i > tri) cur_i = tri;
    };
    for (int t = 0; t < cnt; t++) {
        const int j = (int)(w0 & 0xffULL);
        w0 >>= 8;
        if (--left == 0) { w0 = w1; left = 8; }
        const float4 y = c4[j];
        if (j < ny_p) {
            const int l = j & 7;
            if (l != curl) {
                if (curl >= 0) merge();
                curl = l; trd = 3.402823466e+38F - xn; tri = 0u;
            }
            float dp = m0 * y.x;
            dp = __builtin_fmaf(m1, y.y, dp);
            dp = __builtin_fmaf(m2, y.z, dp);
            dp = dp + y.w;
            if (dp < trd) { trd = dp; tri = (unsigned)j; }
        } else {
            if (curl >= 0) { merge(); curl = -1; }
            float dp = __builtin_fmaf(x2, y.z, __builtin_fmaf(x1, y.y, x0 * y.x));     // simdlib_based.cpp:201-216
            float d = xn + y.w - 2 * dp;
            if (d < 0) d = 0;
            if (cur_d > d) { cur_d = d; cur_i = (unsigned)j; }
        }
    }
    if (curl >= 0) merge();
    return (int)cur_i;
}

__global__ __launch_bounds__(256) void k_km_assign_lut(KmSamples s, size_t nx, const float4 *__restrict__ c4, int k, int nbits,
                                                       int chunk_len, int nchunks, int *__restrict__ assign, unsigned int *table,
                                                       const KmGridDev *__restrict__ gp, int G, const unsigned char *__restrict__ lut) {
    extern __shared__ unsigned int lds_u[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float4 *lc4 = (float4 *)lds_u;
    unsigned int *cnt = lds_u + 4 * (size_t)k + (size_t)wid * k;
    for (int j = threadIdx.x; j < k; j += 256) lc4[j] = c4[j];
    for (int j = lane; j < k; j += 64) cnt[j] = 0u;
    __syncthreads();
    const KmGridDev g = *gp;
    const int chunk = blockIdx.x * 4 + wid;
    if (chunk >= nchunks) return;
    const size_t lo = (size_t)chunk * chunk_len;
    const size_t hi = lo + chunk_len < nx ? lo + chunk_len : nx;
    for (size_t base = lo; base < hi; base += 64) {
        const size_t i = base + lane;
        const bool v = i < hi;
        int a = 0;
        if (v) {
            const float x0 = s.x[i], x1 = s.y[i], x2 = s.z[i];
            const uint4 rec = reinterpret_cast<const uint4 *>(lut)[km_cell(x0, x1, x2, g, G)];
            a = km_assign_pruned(x0, x1, x2, lc4, k, rec);
            assign[i] = a;
        }
        const unsigned long long valid = __ballot(v);
        const unsigned long long m = match_mask(a, nbits, valid);
        if (v && (m & ((1ULL << lane) - 1ULL)) == 0ULL) cnt[a] += (unsigned)__popcll(m);   // group leader; distinct addresses
    }
    for (int j = lane; j < k; j += 64) table[(size_t)j * nchunks + chunk] = cnt[j];
}

// A record of the G^3 table for k % 8 == 0, without branches per candidate: the first four entries unconditionally (the
// record is padded with its last entry), the rest only if some lane of the wavefront holds more.  Same arithmetic and
// merge rule as km_assign_pruned.
__device__ __forceinline__ int km_assign_rec4(const float x0, const float x1, const float x2, const float4 *c4, const int k, const uint4 rec) {
    const int cnt = (int)(rec.x & 0xffu);
    if (cnt == 255) return km_assign_one(x0, x1, x2, c4, k);
    const float m0 = -2 * x0, m1 = -2 * x1, m2 = -2 * x2;
    const float xn = __builtin_fmaf(x2, x2, __builtin_fmaf(x0, x0, x1 * x1));
    auto dot = [&](const unsigned j) {
        const float4 y = c4[j];
        float d = m0 * y.x;
        d = __builtin_fmaf(m1, y.y, d);
        d = __builtin_fmaf(m2, y.z, d);
        return d + y.w;
    };
    unsigned long long key = ~0ULL;
    unsigned prev = (rec.x >> 8) & 0xffu, tri = prev;
    float trd = dot(prev);
    auto merge = [&](const bool doit) {
        float cand = trd + xn;
        cand = cand < 0 ? 0.f : cand;
        const unsigned long long kk = ((unsigned long long)__float_as_uint(cand) << 32) | tri;
        key = (doit && kk < key) ? kk : key;
    };
    auto step = [&](const unsigned j) {
        const bool newlane = ((j ^ prev) & 7u) != 0u;
        merge(newlane);
        const float d = dot(j);
        const bool better = newlane || d < trd;
        trd = better ? d : trd; tri = better ? j : tri;
        prev = j;
    };
    step((rec.x >> 16) & 0xffu); step(rec.x >> 24); step(rec.y & 0xffu);
    if (__any(cnt > 4)) {
        unsigned long long w0 = ((unsigned long long)rec.z << 32) | rec.y, w1 = rec.w;     // bytes 4.. of the record
        w0 = (w0 >> 8) | ((w1 & 0xffULL) << 56); w1 >>= 8;                                 // byte 4 (entry 3) is done
        for (int t = 4; t < 15; t++) {
            if (!__any(t < cnt)) break;
            const unsigned j = (unsigned)(w0 & 0xffULL);
            w0 = (w0 >> 8) | ((w1 & 0xffULL) << 56); w1 >>= 8;
            if (t < cnt) step(j);
        }
    }
    merge(true);
    return (int)(unsigned)key;
}

constexpr int kKmQueue = 48;                                   // parked samples per wavefront (16 B each)

// minimum of two non-negative-or-not doubles given as bit patterns, no canonicalisation (the operands are never NaN here)
__device__ __forceinline__ double km_min_key(const double a, const double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Four candidates given by their LDS addresses (index << 4 into the centroid records at `lb`), branch-free.  The reference
// keeps one strict-'<' tracker per SIMD lane (j & 7) over the raw form dp = fma(-2 x2, y2, fma(-2 x1, y1, (-2 x0) y0)) + |y|^2
// and merges the lane minima by (max(0, dp + |x|^2), index).  With s_t = dp_t + |x|^2 >= 0 for every candidate, the winner of
// that procedure is the candidate with the smallest (s, index) over ALL candidates, unless two candidates of one lane share
// the smallest s with different dp (the tracker then prefers the smaller dp, not the smaller index): s is monotone in dp, so
// a lane's tracker holds a smallest-s member, and among equal s the merge takes the smallest index.  {s bits : address} is
// one 64-bit key (s >= 0: bit patterns order like values; read as a positive double, v_min_f64 is its 64-bit minimum).
// Returns the winner's address; `doubt` is set when some OTHER index reaches the smallest s or the smallest s is negative
// (the reference clamps: ties at 0) -- then the caller runs the exact procedure; on real data a handful per million.
// The order of the four and repeated entries (padding) do not matter.
// `lb` + ad[t] are LDS byte addresses (not generic pointers: a pointer built from the kernel's extern LDS array costs one add of a
// link-time constant per access; the callers keep the centroid records at a known address).
typedef float KmVec4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const KmVec4 LdsF4;
typedef __attribute__((address_space(3))) const unsigned int LdsU32;
__device__ __forceinline__ double km_eval4(const float x0, const float x1, const float x2, const unsigned lb, const unsigned (&ad)[4], bool &doubt) {
    const float m0 = -2 * x0, m1 = -2 * x1, m2 = -2 * x2;
    const float xn = __builtin_fmaf(x2, x2, __builtin_fmaf(x0, x0, x1 * x1));
    int sb[4];
    double key[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const KmVec4 y = *(LdsF4 *)(lb + ad[t]);
        float d = m0 * y.x;
        d = __builtin_fmaf(m1, y.y, d);
        d = __builtin_fmaf(m2, y.z, d);
        d = d + y.w;
        sb[t] = __float_as_int(d + xn);
        key[t] = __hiloint2double(sb[t], (int)ad[t]);
    }
    const double kmin = km_min_key(km_min_key(key[0], key[1]), km_min_key(key[2], key[3]));
    const int smin = __double2hiint(kmin);
    const unsigned amin = (unsigned)__double2loint(kmin);
    unsigned flag = (unsigned)(smin < 0);                                       // no short circuits: no branches
#pragma unroll
    for (int t = 0; t < 4; t++) flag |= (unsigned)(sb[t] == smin) & (unsigned)(ad[t] != amin);
    doubt = flag != 0u;
    return kmin;                                                                // {s bits : address} of the winner
}

// Parked samples of k_km_assign_mid, a (nearly) full wavefront at a time: the 64^3 records hold up to fifteen candidates, but
// the cell of a sample is an eighth of the crowded one and four suffice for most -- the evaluation above on the first four
// entries (records are padded with their last), then on the rest; doubtful samples and cells of more than fifteen take the
// exact procedure (a few per million: one straggler used to drag most drains through it).  Called from ONE
// place (the code is long; several inlined copies cost more in instruction fetch than the drains themselves).
struct KmDrainGrid { float lo[3], inv64[3]; int sane; };
__device__ __forceinline__ void km_mid_drain(const int n, const uint4 *q, const float4 *lc4, unsigned int *cnt, unsigned char *assign_chunk,
                                          const unsigned char *__restrict__ lut, const int k, const KmDrainGrid dg) {
    const int lane = threadIdx.x & 63;
    int a = 0;
    uint4 it = make_uint4(0u, 0u, 0u, 0u), rec = it;
    bool exact = false;
    if (lane < n) {
        it = q[lane];
        const float a0 = __uint_as_float(it.x), a1 = __uint_as_float(it.y), a2 = __uint_as_float(it.z);
        const unsigned ix = __float2uint_rz((a0 - dg.lo[0]) * dg.inv64[0]), iy = __float2uint_rz((a1 - dg.lo[1]) * dg.inv64[1]),
                       iz = __float2uint_rz((a2 - dg.lo[2]) * dg.inv64[2]);
        rec = reinterpret_cast<const uint4 *>(lut)[dg.sane ? ((iz << 12) | (iy << 6) | ix) : 0u];
        const unsigned ad[4] = {(rec.x >> 4) & 0xff0u, (rec.x >> 12) & 0xff0u, (rec.x >> 20) & 0xff0u, (rec.y << 4) & 0xff0u};
        bool doubt;
        double kw = km_eval4(a0, a1, a2, 0u, ad, doubt);
        const unsigned cnt = rec.x & 0xffu;
        // entries five to fifteen the same way, four at a time, while some lane holds that many; groups combine like
        // candidates: the smaller key wins, equal s at different addresses is a doubt (a record's padding repeats its LAST
        // entry, so a group beyond the record's count is all one entry of an earlier group and changes nothing)
        auto group = [&](const unsigned wa, const unsigned wb) {
            const unsigned adg[4] = {(wa >> 4) & 0xff0u, (wa >> 12) & 0xff0u, (wa >> 20) & 0xff0u, (wb << 4) & 0xff0u};
            bool doubt2;
            const double k2 = km_eval4(a0, a1, a2, 0u, adg, doubt2);
            doubt = doubt || doubt2 || (__double2hiint(k2) == __double2hiint(kw) && __double2loint(k2) != __double2loint(kw));
            kw = km_min_key(kw, k2);
        };
        if (__any(cnt > 4u && cnt <= 15u)) {
            if (cnt > 4u) group(rec.y, rec.z);
            if (__any(cnt > 8u && cnt <= 15u)) {
                if (cnt > 8u) { group(rec.z, rec.w); group(rec.w, rec.w >> 24); }
            }
        }
        a = (int)((unsigned)__double2loint(kw) >> 4);
        exact = doubt || cnt > 15u || !dg.sane;                                   // 255: the cell keeps more than fifteen
    }
    if (__any(exact)) {
        if (exact) {
            if (!dg.sane) rec.x = 255u;                                           // full scan
            a = km_assign_rec4(__uint_as_float(it.x), __uint_as_float(it.y), __uint_as_float(it.z), lc4, k, rec);
        }
    }
    if (lane < n) {
        assign_chunk[it.w] = (unsigned char)a;                                    // after (program order) the packed store that left a 0 here
        atomicAdd(&cnt[a], 1u);
    }
}

// LDS (uints): centroid records [0, 1024) | per-wavefront counters [1024, 5120) | 32^3 table [5120, 37888) | parked samples
__global__ __launch_bounds__(1024) void k_km_assign_mid(KmSamples s, size_t nx, const float4 *__restrict__ c4, int k,
                                                        int chunk_len, int nchunks, unsigned char *__restrict__ assign, unsigned int *table,
                                                        const KmGridDev *__restrict__ gp, const unsigned int *__restrict__ mid,
                                                        const unsigned char *__restrict__ lut) {
    // assign: ONE BYTE per sample (k <= 256 here): 13 bytes of traffic per sample instead of 16, and k_km_scatter reads a
    // quarter.  Per trip a lane takes PG groups of four CONSECUTIVE samples (group g = samples 256 g + 4 lane ...): three 16-byte
    // loads in and one packed 4-byte store out per group.
    extern __shared__ unsigned int lds_u[];
    constexpr int Gm = 32, ncell = Gm * Gm * Gm, PG = 1, P = 4 * PG;       // PG = 2 needs more than 128 VGPRs: slower
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: chunk bounds and bases live in SGPRs
    float4 *lc4 = (float4 *)lds_u;                                                // [256] at byte 0: a candidate's address is its index << 4
    unsigned int *cnt = lds_u + 4 * 256 + (size_t)wid * 256;                      // this wavefront's counters
    unsigned int *T = lds_u + 4 * 256 + 16 * 256;                                 // [ncell]
    uint4 *q = (uint4 *)(T + ncell) + wid * kKmQueue;                             // this wavefront's parked samples
    // the hot loop addresses LDS by number: the centroid records at byte 0, the table at byte kTableAt (this kernel has no
    // static LDS, so the extern array starts at 0; anything else is a build that must not run)
    constexpr unsigned kTableAt = (4 * 256 + 16 * 256) * 4;
    if ((unsigned)(size_t)(__attribute__((address_space(3))) unsigned int *)lds_u != 0u) __builtin_trap();
    for (int i = threadIdx.x; i < ncell / 4; i += 1024) ((uint4 *)T)[i] = ((const uint4 *)mid)[i];
    for (int j = threadIdx.x; j < 256; j += 1024) lc4[j] = j < k ? c4[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = lane; j < 256; j += 64) cnt[j] = 0u;
    __syncthreads();
    const KmGridDev g = *gp;
    // Grid cells in f32: t = (x - lo) * inv with inv = G' (1 - 2^-18) / range rounded to f32 (G' = 32 for the table in LDS, 64 for
    // the records).  x - lo >= 0 (lo is the exact minimum), three roundings of 2^-24 and the factor 1 - 2^-18 keep t in
    // [0, G') and within 3e-4 of a cell of the exact position: the boxes both tables were built from are wider than that
    // (km_box).  Ranges f32 cannot handle without overflow or underflow send every sample through the exact full scan.
    float flo[3], finv[3], finv64[3];
    bool sane = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double r = (double)g.hi[a] - (double)g.lo[a];
        flo[a] = g.lo[a];
        finv[a] = r > 0 ? (float)(32.0 * (1.0 - 0x1.0p-18) / r) : 0.f;                // (zeroed below when the range is not sane)
        finv64[a] = r > 0 ? (float)(64.0 * (1.0 - 0x1.0p-18) / r) : 0.f;
        sane = sane && fabsf(g.lo[a]) < 1e18f && fabsf(g.hi[a]) < 1e18f && (r == 0 || r > 1e-30);
    }
    // not sane: every sample parks, and its table entry is never used -- a factor 0 sends (x - lo) * 0 = 0 or NaN to cell 0
    // (v_cvt_u32_f32 of NaN is 0) without a select per sample
#pragma unroll
    for (int a = 0; a < 3; a++) finv[a] = sane ? finv[a] : 0.f;
    KmDrainGrid dg;
#pragma unroll
    for (int a = 0; a < 3; a++) { dg.lo[a] = flo[a]; dg.inv64[a] = finv64[a]; }
    dg.sane = sane ? 1 : 0;
    constexpr unsigned lb = 0u;
    for (int chunk = blockIdx.x * 16 + wid; chunk < nchunks; chunk += gridDim.x * 16) {
        const size_t lo = (size_t)chunk * chunk_len;
        const size_t hi = lo + chunk_len < nx ? lo + chunk_len : nx;
        int qn = 0;                                                               // wave-uniform
        auto drain = [&](const int n) { km_mid_drain(n, q, lc4, cnt, assign + lo, lut, k, dg); };
        // the samples of trip n+1 are requested before trip n is evaluated (a wavefront owns its chunk: nothing else hides
        // the memory latency at four wavefronts per SIMD)
        float n0[P], n1[P], n2[P];
        const unsigned len = (unsigned)(hi > lo ? hi - lo : 0);                   // 32-bit positions inside the chunk: scalar compares
        auto fetch = [&](const unsigned rel) {
            const unsigned left = min(64u * P, len - rel);                        // wave-uniform base + 32-bit lane offsets
            const float *bx = s.x + lo + rel, *by = s.y + lo + rel, *bz = s.z + lo + rel;
            if (left == 64u * P) {                                               // chunks start at multiples of 64 samples: 16-byte aligned
#pragma unroll
                for (int gq = 0; gq < PG; gq++) {
                    const float4 v0 = reinterpret_cast<const float4 *>(bx)[64 * gq + lane], v1 = reinterpret_cast<const float4 *>(by)[64 * gq + lane],
                                 v2 = reinterpret_cast<const float4 *>(bz)[64 * gq + lane];   // (non-temporal loads measured the same)
                    n0[4 * gq] = v0.x; n0[4 * gq + 1] = v0.y; n0[4 * gq + 2] = v0.z; n0[4 * gq + 3] = v0.w;
                    n1[4 * gq] = v1.x; n1[4 * gq + 1] = v1.y; n1[4 * gq + 2] = v1.z; n1[4 * gq + 3] = v1.w;
                    n2[4 * gq] = v2.x; n2[4 * gq + 1] = v2.y; n2[4 * gq + 2] = v2.z; n2[4 * gq + 3] = v2.w;
                }
            } else {
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const unsigned j = min(256u * (unsigned)(p >> 2) + (unsigned)lane * 4u + (unsigned)(p & 3), left - 1u);
                    n0[p] = bx[j]; n1[p] = by[j]; n2[p] = bz[j];
                    asm volatile("" ::: "memory");                               // keeps the two branches apart (merged, the 16-byte loads
                }                                                                 // above become twelve 4-byte ones)
            }
        };
        if (len) fetch(0u);
        // the first trip's samples are waited for HERE: left pending into the loop they make the compiler's wait-count
        // bookkeeping (merged over both ways into the loop) wait for the NEXT trip's loads at the top of every trip
#pragma unroll
        for (int p = 0; p < P; p++) asm volatile("" :: "v"(n0[p]), "v"(n1[p]), "v"(n2[p]));
        for (unsigned rel = 0; rel < len; rel += 64 * P) {
            float x0[P], x1[P], x2[P];
            bool v[P];
            const unsigned left = min(64u * P, len - rel);
            unsigned char *ba = assign + lo + rel;
#pragma unroll
            for (int p = 0; p < P; p++) { x0[p] = n0[p]; x1[p] = n1[p]; x2[p] = n2[p]; v[p] = 256u * (unsigned)(p >> 2) + (unsigned)lane * 4u + (unsigned)(p & 3) < left; }
            unsigned packed[PG];
#pragma unroll
            for (int gq = 0; gq < PG; gq++) packed[gq] = 0u;
            fetch(rel + 64 * P < len ? rel + 64 * P : 0u);                       // after the last trip: the first again, unused (an
                                                                                  // unconditional fetch keeps the registers of the two trips apart)
            unsigned e[P];
#pragma unroll
            for (int p = 0; p < P; p++) {
                const unsigned ix = __float2uint_rz((x0[p] - flo[0]) * finv[0]), iy = __float2uint_rz((x1[p] - flo[1]) * finv[1]),
                               iz = __float2uint_rz((x2[p] - flo[2]) * finv[2]);
                e[p] = *(LdsU32 *)(kTableAt + ((iz << 12) | (iy << 7) | (ix << 2)));
            }
            unsigned ovbits = 0;
#pragma unroll
            for (int p = 0; p < P; p++) {
                // the (up to) four candidates of the cell (k % 8 == 0: no scalar leftovers in the reference's procedure)
                const unsigned ad[4] = {(e[p] & 0xffu) << 4, (e[p] >> 4) & 0xff0u, (e[p] >> 12) & 0xff0u, (e[p] >> 20) & 0xff0u};
                bool doubt;
                const unsigned cur_i = (unsigned)__double2loint(km_eval4(x0[p], x1[p], x2[p], lb, ad, doubt)) >> 4;
                const bool park = (doubt || e[p] == kKmMidOverflow || !sane) && v[p];
                ovbits |= park ? (1u << p) : 0u;
                if (v[p] && !park) { packed[p >> 2] |= cur_i << (8 * (p & 3)); atomicAdd(&cnt[cur_i], 1u); }
            }
            // the next trip's samples are waited for BEFORE this trip's store goes out (they were requested ~400 instructions
            // ago): loads and stores retire through one in-order counter, so a wait placed after the store -- where the
            // values are first needed -- would also wait for the store, once per trip
#pragma unroll
            for (int p = 0; p < P; p++) asm volatile("" :: "v"(n0[p]), "v"(n1[p]), "v"(n2[p]));
            if (left == 64u * P) {                                                // parked samples: 0 for now, the drain writes theirs
#pragma unroll
                for (int gq = 0; gq < PG; gq++) reinterpret_cast<unsigned int *>(ba)[64 * gq + lane] = packed[gq];
            } else {
#pragma unroll
                for (int p = 0; p < P; p++) if (v[p]) ba[256u * (unsigned)(p >> 2) + (unsigned)lane * 4u + (unsigned)(p & 3)] = (unsigned char)(packed[p >> 2] >> (8 * (p & 3)));
            }
            const bool lasttrip = rel + 64 * P >= len;
            if (__ballot(ovbits != 0u) || lasttrip) {
                // park: the samples of this trip are numbered slot-major (all of slot 0, then slot 1, ...); as many as fit go
                // into the queue, a full queue is drained, and so on -- one loop around ONE drain site
                unsigned long long m[P];
                int off[P + 1], rank[P];
                off[0] = 0;
#pragma unroll
                for (int p = 0; p < P; p++) {
                    m[p] = __ballot((ovbits >> p) & 1u);
                    off[p + 1] = off[p] + (int)__popcll(m[p]);
                    rank[p] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m[p] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m[p], 0u));
                }
                const int btot = off[P];
                int done = 0;
                do {                                                              // wave-uniform
                    const int take = min(kKmQueue - qn, btot - done);
#pragma unroll
                    for (int p = 0; p < P; p++) {
                        const int gr = off[p] + rank[p] - done;
                        if (((ovbits >> p) & 1u) && gr >= 0 && gr < take)
                            q[qn + gr] = make_uint4(__float_as_uint(x0[p]), __float_as_uint(x1[p]), __float_as_uint(x2[p]),
                                                    rel + 256u * (unsigned)(p >> 2) + (unsigned)lane * 4u + (unsigned)(p & 3));
                    }
                    qn += take; done += take;
                    if (qn == kKmQueue || (lasttrip && done == btot)) { drain(qn); qn = 0; }
                } while (done < btot);
            }
        }
        for (int j = lane; j < k; j += 64) { table[(size_t)j * nchunks + chunk] = cnt[j]; cnt[j] = 0u; }
    }
}

// second half of the sort: samples -> (x,y,z,w) records grouped by centroid, sample order kept
// PAIR (k <= 256, long chunks): a record is 16 bytes, HBM is written in 32-byte sectors.  Records of one centroid leave a
// wavefront one at a time, four trips apart on average, so both halves of almost every sector used to reach memory separately
// (WRITE_SIZE 1.86x the algorithmic bytes at 67 M samples).  Here a record bound for an EVEN slot waits in LDS (one place per
// centroid and wavefront) until its odd neighbour turns up and the two are stored back to back by one lane; what is left over at
// the end of the chunk -- at most one record per centroid -- goes out alone.  Same records in the same places.
template <bool W, typename AT, bool PAIR = false>
__global__ __launch_bounds__(256) void k_km_scatter(KmSamples s, const AT *__restrict__ assign, size_t nx, int k, int nbits,
                                                    int chunk_len, int nchunks, const unsigned int *__restrict__ table,
                                                    const unsigned int *__restrict__ rowtot, float4 *sorted) {
    extern __shared__ unsigned int lds_u[];
    __shared__ unsigned int wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned int *rowbase = lds_u;                                   // [k] exclusive prefix of rowtot
    unsigned int *cnt = lds_u + k + (size_t)wid * k;
    // PAIR: behind the 5 k counters, per wavefront: first slot of every centroid in this chunk [k], waiting records [k], and
    // this trip's records by lane [64]
    unsigned int *first = lds_u + 5 * k + (size_t)wid * k;
    float4 *pend = reinterpret_cast<float4 *>(lds_u + 9 * k) + (size_t)wid * (k + 64);
    float4 *stage = pend + k;
    {   // block-wide exclusive scan of rowtot[0..k): each thread owns a contiguous run
        const int per = (k + 255) / 256;
        const int j0 = threadIdx.x * per, j1 = (j0 + per < k) ? j0 + per : k;
        unsigned sum = 0;
        for (int j = j0; j < j1; j++) sum += rowtot[j];
        unsigned inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        unsigned pre = 0;
        for (int w = 0; w < wid; w++) pre += wsum[w];
        unsigned run = pre + inc - sum;
        for (int j = j0; j < j1; j++) { rowbase[j] = run; run += rowtot[j]; }
    }
    __syncthreads();
    const int chunk = blockIdx.x * 4 + wid;
    if (chunk >= nchunks) return;
    // next free slot of every centroid for THIS chunk: row base + what earlier chunks hold (one strided read of the
    // count table per chunk instead of one random read per sample)
    // (only when a chunk holds more samples than there are centroids; short chunks read their few entries directly)
    const bool prebase = PAIR || chunk_len >= 2 * k;
    if (prebase) {
        for (int j = lane; j < k; j += 64) {
            const unsigned c0 = rowbase[j] + table[(size_t)j * nchunks + chunk];
            cnt[j] = c0;
            if constexpr (PAIR) first[j] = c0;
        }
    } else { for (int j = lane; j < k; j += 64) cnt[j] = 0u; }
    const size_t lo = (size_t)chunk * chunk_len;
    const size_t hi = lo + chunk_len < nx ? lo + chunk_len : nx;
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    // four groups of 64 samples per trip: their assignments and coordinates are requested together, the ranks (which chain
    // through the per-centroid counters) are then taken group by group
    constexpr int U = 4;
    for (size_t base = lo; base < hi; base += 64 * U) {
        int a[U]; bool v[U]; float4 rec[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 64 + lane;
            v[u] = i < hi;
            const size_t j = v[u] ? i : lo;
            a[u] = (int)assign[j];
            float w = 1.0f;
            if constexpr (W) w = s.w[j];
            rec[u] = make_float4(s.x[j], s.y[j], s.z[j], w);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (base + (size_t)u * 64 >= hi) break;                    // wave-uniform
            const unsigned long long valid = __ballot(v[u]);
            const unsigned long long m = match_mask(v[u] ? a[u] : 0, nbits, valid);
            unsigned run = 0;
            if (v[u]) run = cnt[a[u]];                                 // read before the leader bumps it
            const unsigned r = (unsigned)__popcll(m & lt);
            if constexpr (PAIR) {
                const unsigned dst = run + r;
                if (v[u]) stage[lane] = rec[u];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // the even neighbour of an odd slot: the previous lane of this centroid in this trip, else the record waiting in
                // LDS, else (the chunk's first record of the centroid sits on an odd slot) nobody.  Read first, then a barrier:
                // a later lane of the same centroid may put the NEXT waiting record into the same place below
                float4 other = make_float4(0.f, 0.f, 0.f, 0.f);
                bool have = false;
                if (v[u] && (dst & 1u)) {
                    if (r > 0) { other = stage[63 - __clzll(m & lt)]; have = true; }
                    else if (dst > first[a[u]]) { other = pend[a[u]]; have = true; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (v[u]) {
                    if (dst & 1u) {
                        if (have) sorted[dst - 1] = other;
                        sorted[dst] = rec[u];
                    } else if (((m >> lane) >> 1) == 0ULL) {
                        pend[a[u]] = rec[u];                           // last of its centroid in this trip: wait for the neighbour
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else if (v[u]) {
                const size_t dst = prebase ? (size_t)run + r : (size_t)rowbase[a[u]] + table[(size_t)a[u] * nchunks + chunk] + run + r;
                sorted[dst] = rec[u];
            }
            if (v[u] && r == 0) cnt[a[u]] = run + (unsigned)__popcll(m);   // leader
        }
    }
    if constexpr (PAIR) {
        // records still waiting: the centroid's last slot in this chunk is even
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < k; j += 64) {
            const unsigned c1 = cnt[j];
            if (c1 > first[j] && (c1 & 1u)) sorted[c1 - 1] = pend[j];
        }
    }
}

// The same half of the sort through LDS (k <= 256, one-byte assignments, long chunks): k_km_scatter's stores leave a wavefront one
// 16- or 32-byte piece per lane, every piece its own cache line and its own DRAM page (SQ: 56 % of the wavefront time waiting to
// issue; 1.07 GB written at 1.1 TB/s).  Here a BLOCK takes one chunk, 4096 samples at a time (wavefront w the w-th 1024 of them):
// ranks per wavefront as before (match masks, per-centroid counters in LDS), one prefix over (centroid, wavefront) for the batch,
// the records placed in LDS in sorted order, and the stores made from there -- consecutive threads hold consecutive slots, so the
// sixteen records a centroid gets from a batch on average leave as one 256-byte piece.  Same records in the same places: the sort
// stays stable (wavefront, then trip, then lane = sample order).
template <bool W>
__global__ __launch_bounds__(256) void k_km_scatter_lds(KmSamples s, const unsigned char *__restrict__ assign, size_t nx, int k, int chunk_len,
                                                       int nchunks, const unsigned int *__restrict__ table, const unsigned int *__restrict__ rowtot,
                                                       float4 *sorted) {
    extern __shared__ unsigned int lds_u[];
    __shared__ unsigned int wsum[4];
    constexpr int BW = 1024, U = BW / 64, B = 4 * BW;                                // samples per wavefront and per block in a batch
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned int *cur = lds_u;                                                       // [256] next place of every centroid in `sorted`
    unsigned int *start = lds_u + 256;                                               // [256] first slot of every centroid in this batch
    unsigned int *cntw = lds_u + 512;                                                // [4][256] members per wavefront in this batch -> its first slot
    unsigned char *ca = reinterpret_cast<unsigned char *>(lds_u + 512 + 1024);       // [B] centroid of the record in a slot
    float4 *rec = reinterpret_cast<float4 *>(lds_u + 512 + 1024 + B / 4);            // [B] the batch in sorted order
    unsigned int *cnt = cntw + 256 * wid;
    const int chunk = blockIdx.x;
    {   // block-wide exclusive scan of rowtot[0..k), one centroid per thread, + what earlier chunks hold
        const int j = threadIdx.x;
        const unsigned a0 = j < k ? rowtot[j] : 0u;
        const unsigned inc = wave_scan_incl_u32(a0);
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        unsigned pre = inc - a0;
        for (int w2 = 0; w2 < wid; w2++) pre += wsum[w2];
        cur[j] = j < k ? pre + table[(size_t)j * nchunks + chunk] : 0u;
#pragma unroll
        for (int w2 = 0; w2 < 4; w2++) cntw[256 * w2 + j] = 0u;
    }
    __syncthreads();
    const size_t lo = (size_t)chunk * chunk_len;
    const size_t hi = lo + chunk_len < nx ? lo + chunk_len : nx;
    const unsigned long long lt = (1ULL << lane) - 1ULL;
    for (size_t b0 = lo; b0 < hi; b0 += B) {
        const unsigned nb = (unsigned)(hi - b0 < (size_t)B ? hi - b0 : (size_t)B);
        const unsigned w0 = (unsigned)wid * BW;                                       // this wavefront's first sample of the batch
        unsigned ao[U]; float4 r[U];
#pragma unroll
        for (int u = 0; u < U; u++) {                                                // the whole batch is requested before any of it is used
            const size_t i = b0 + w0 + (size_t)u * 64 + lane;
            const size_t j = i < hi ? i : lo;
            ao[u] = assign[j];
            float w = 1.0f;
            if constexpr (W) w = s.w[j];
            r[u] = make_float4(s.x[j], s.y[j], s.z[j], w);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const bool v = w0 + (unsigned)u * 64u + (unsigned)lane < nb;
            const unsigned long long valid = __ballot(v);
            const int a = (int)ao[u];
            const unsigned long long m = match_mask(v ? a : 0, 8, valid);
            unsigned run = 0;
            if (v) run = cnt[a];                                                     // read before the leader bumps it
            const unsigned rk = (unsigned)__popcll(m & lt);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (v && rk == 0) cnt[a] = run + (unsigned)__popcll(m);                  // leader
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            ao[u] = v ? ((unsigned)a | ((run + rk) << 8)) : 0xFFFFFFFFu;
        }
        __syncthreads();
        {   // slots: centroid-major, then wavefront.  Thread j owns centroid j: its four wavefront counts become their first slots
            const int j = threadIdx.x;
            const unsigned c0 = cntw[j], c1 = cntw[256 + j], c2 = cntw[512 + j], c3 = cntw[768 + j];
            const unsigned tot = c0 + c1 + c2 + c3;
            const unsigned inc = wave_scan_incl_u32(tot);
            if (lane == 63) wsum[wid] = inc;
            __syncthreads();
            unsigned pre = inc - tot;
            for (int w2 = 0; w2 < wid; w2++) pre += wsum[w2];
            start[j] = pre;
            cntw[j] = pre; cntw[256 + j] = pre + c0; cntw[512 + j] = pre + c0 + c1; cntw[768 + j] = pre + c0 + c1 + c2;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (ao[u] != 0xFFFFFFFFu) {
                const unsigned a = ao[u] & 0xffu, slot = cnt[a] + (ao[u] >> 8);
                rec[slot] = r[u];
                ca[slot] = (unsigned char)a;
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < B / 256; u++) {
            const unsigned sidx = (unsigned)u * 256u + threadIdx.x;
            if (sidx < nb) {
                const unsigned c = ca[sidx];
                sorted[(size_t)cur[c] + (sidx - start[c])] = rec[sidx];
            }
        }
        __syncthreads();
        {   // advance the places by what the batch held (the next centroid's first slot, or the batch's size, minus this one's)
            const int j = threadIdx.x;
            const unsigned nxt = j + 1 < 256 ? start[j + 1] : nb;
            __syncthreads();
            cur[j] += nxt - start[j];
#pragma unroll
            for (int w2 = 0; w2 < 4; w2++) cntw[256 * w2 + j] = 0u;
        }
        __syncthreads();
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
};

// split_clusters (Clustering.cpp:216-263): every empty cluster ci, in index order, takes over half of a donor cj found by walking
// cj = 0, 1, ... (cyclically) with one draw r of std::mt19937(1234) per candidate until r < (h[cj] - 1) / (n - k); ci becomes a
// copy of cj, the two are nudged apart by a factor 1 +- 1/1024 per coordinate, and the size is shared.
// Taken by ONE WAVEFRONT (all 64 lanes call it).  An empty cluster costs ~k draws (each candidate is accepted
// with probability ~1/k), a posterised image leaves a hundred clusters empty in every iteration, and one lane gets through a draw
// in ~7 ns: 190 us per iteration.  Here lane t tests candidate cj0 + t against draw number t of the remaining sequence -- the
// first hit is the sequential loop's hit, and exactly the draws up to it are consumed.  The generator's state sits in LDS: its
// start (seed 1234, `seeded`, built once per workspace) is copied in, a block of 624 outputs is re-generated 64 words at a time
// in ascending order, every lane reading its three inputs before any lane writes -- word i sees the old i + 1 and i + 397 (or
// the new i - 227), as in the sequential loop.  s_mt: 624 words, s_h: k floats (the sizes, updated as clusters are split).
// HMEM: s_h is a scratch array in device memory (k beyond what LDS holds): read and written past the vector L1 (agent scope).
template <bool HMEM>
struct KmSizes {
    float *p;
    __device__ __forceinline__ float get(const int j) const {
        if constexpr (HMEM) return __hip_atomic_load(p + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else return p[j];
    }
    __device__ __forceinline__ void put(const int j, const float v) const {
        if constexpr (HMEM) __hip_atomic_store(p + j, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else p[j] = v;
    }
};
template <bool HMEM = false>
__device__ __forceinline__ void km_split_clusters_wave(float *cent, float *hassign, const int k, const unsigned long long n, const DevMT *seeded,
                                                       unsigned *s_mt, float *s_h_, const int lane, unsigned int *flags = nullptr) {
    const KmSizes<HMEM> s_h{s_h_};
    for (int i = lane; i < 624; i += 64) s_mt[i] = seeded->mt[i];
    for (int j = lane; j < k; j += 64) s_h.put(j, hassign[j]);
    if constexpr (HMEM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int idx = 624;                                                          // nothing generated yet (DevMT::seed)
    const float denom = (float)(n - (unsigned long long)k);
    for (int ci = 0; ci < k; ci++) {
        if (s_h.get(ci) != 0.f) continue;                                   // wave-uniform (one LDS word)
        int cj0 = 0, cj = 0;
        for (;;) {
            if (idx >= 624) {
                for (int c0 = 0; c0 < 624; c0 += 64) {
                    const int i = c0 + lane;
                    unsigned v = 0;
                    if (i < 624) {
                        const unsigned y = (s_mt[i] & 0x80000000U) | (s_mt[(i + 1) % 624] & 0x7fffffffU);
                        v = s_mt[(i + 397) % 624] ^ (y >> 1);
                        if (y & 1U) v ^= 0x9908b0dfU;
                    }
                    __builtin_amdgcn_wave_barrier();                       // all reads of this chunk before its writes
                    if (i < 624) s_mt[i] = v;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                idx = 0;
            }
            const int navail = 624 - idx < 64 ? 624 - idx : 64;
            bool hit = false;
            if (lane < navail) {
                unsigned y = s_mt[idx + lane];
                y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680U; y ^= (y << 15) & 0xefc60000U; y ^= (y >> 18);
                const float r = (float)y / 4294967296.0f;
                const float p = (float)(((double)s_h.get((cj0 + lane) % k) - 1.0) / (double)denom);
                hit = r < p;
            }
            const unsigned long long m = __ballot(hit);
            if (m) { const int t = __ffsll((long long)m) - 1; idx += t + 1; cj = (cj0 + t) % k; break; }
            idx += navail; cj0 = (cj0 + navail) % k;
        }
        if (lane == 0) {
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
            const float hcj = s_h.get(cj);
            const float hi = hcj / 2, hj = hcj - hi;
            s_h.put(ci, hi); s_h.put(cj, hj);
            hassign[ci] = hi; hassign[cj] = hj;
            if (flags) { flags[1 + ci] = 1u; flags[1 + cj] = 1u; }          // their next update must not be skipped
            if constexpr (HMEM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
__global__ void k_km_mt_seed(DevMT *mt) { mt->seed(1234u); }

// ---- centroid update: one wavefront replays one centroid's sequential f32 chain; the last
// wavefront to finish handles empty clusters and writes (y, |y|^2) for the next assignment ----
// One wavefront per coordinate: the x, y, z (and weight) sums of a centroid are independent sequential chains, so a
// block of four wavefronts runs them side by side -- per sample one LDS broadcast read and one add on each chain.
// `load(i)`: record i of the centroid's members in sample order, as (x, y, z, w) -- only component C and w are looked at
template <bool W, int C, class Loader>
__device__ __forceinline__ float km_chain_over(const Loader load, const size_t count, float4 (*stage)[64], const int lane, const float acc0) {
    // A dependent f32 add issues every ~8 cycles (3.3 ns) on a lone wavefront; an LDS read takes ~130.  So the 64 samples
    // of block n+1 are staged and their sixteen 128-bit broadcast reads issued BEFORE the 64 adds of block n: two register
    // sets, nothing but the chain itself on the critical path.
    constexpr int D = 8;                                                   // 1-KiB global loads kept in flight
    constexpr bool WX = W && C < 3;                                        // weighted coordinate chain: acc = fma(x, w, acc)
    constexpr int NR = WX ? 32 : 16;
    float acc = acc0;
    const size_t nfull = count / 64;
    const int tail = (int)(count - nfull * 64);
    auto comp = [](const float4 v) { return C == 0 ? v.x : (C == 1 ? v.y : (C == 2 ? v.z : v.w)); };
    float4 ring[D];
#pragma unroll
    for (int d = 0; d < D; d++) {
        ring[d] = make_float4(0, 0, 0, 0);
        if ((size_t)d < nfull) ring[d] = load((size_t)d * 64 + lane);
    }
    float4 tv = make_float4(0, 0, 0, 0);                                   // the partial last block, fetched up front
    if (lane < tail) tv = load(nfull * 64 + lane);
    auto put = [&](const int buf, const float4 v) {                        // this wavefront's coordinate (and weight) of 64 samples
        float *sc = reinterpret_cast<float *>(&stage[buf][0]);
        sc[lane] = comp(v);
        if constexpr (WX) sc[64 + lane] = v.w;
    };
    float4 xa[NR], xb[NR];
    auto issue = [&](const int buf, float4 (&x)[NR]) {
        const float4 *s4 = &stage[buf][0];
#pragma unroll
        for (int q = 0; q < NR; q++) x[q] = s4[q];
    };
    auto consume = [&](const float4 (&x)[NR]) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if constexpr (WX) {
                acc = __builtin_fmaf(x[q].x, x[16 + q].x, acc); acc = __builtin_fmaf(x[q].y, x[16 + q].y, acc);
                acc = __builtin_fmaf(x[q].z, x[16 + q].z, acc); acc = __builtin_fmaf(x[q].w, x[16 + q].w, acc);
            } else {
                acc += x[q].x; acc += x[q].y; acc += x[q].z; acc += x[q].w;
            }
        }
    };
    if (nfull > 0) {
        put(0, ring[0]);
        ring[0] = make_float4(0, 0, 0, 0);
        if ((size_t)D < nfull) ring[0] = load((size_t)D * 64 + lane);
        __builtin_amdgcn_wave_barrier();
        issue(0, xa);
    }
    for (size_t n0 = 0; n0 < nfull; n0 += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const size_t n = n0 + d;
            if (n >= nfull) break;                                         // wave-uniform
            if (n + 1 < nfull) {                                           // stage block n+1 and start its reads
                constexpr int dn = 0;
                (void)dn;
                const int rd = (d + 1) % D;
                put((d + 1) & 1, ring[rd]);
                ring[rd] = make_float4(0, 0, 0, 0);
                if (n + 1 + D < nfull) ring[rd] = load((n + 1 + D) * 64 + lane);
                __builtin_amdgcn_wave_barrier();
                if ((d & 1) == 0) issue(1, xb); else issue(0, xa);
            }
            __builtin_amdgcn_sched_barrier(0);                             // keep the reads above the chain
            if ((d & 1) == 0) consume(xa); else consume(xb);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (tail > 0) {
        __builtin_amdgcn_wave_barrier();
        const int buf = (int)(nfull & 1);
        put(buf, tv);
        __builtin_amdgcn_wave_barrier();
        const float *sc = reinterpret_cast<const float *>(&stage[buf][0]);
        const float4 *s4 = reinterpret_cast<const float4 *>(sc), *w4 = reinterpret_cast<const float4 *>(sc + 64);
        for (int q = 0; 4 * q < tail; q++) {
            const float4 x = s4[q];
            float4 w = make_float4(0, 0, 0, 0);
            if constexpr (WX) w = w4[q];
            const float xs[4] = {x.x, x.y, x.z, x.w}, ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                if (4 * q + e < tail) {
                    if constexpr (WX) acc = __builtin_fmaf(xs[e], ws[e], acc);
                    else acc += xs[e];
                }
            }
        }
    }
    return acc;
}

template <bool W, int C>
__device__ __forceinline__ float km_chain(const float4 *__restrict__ sorted, size_t lo, size_t hi, float4 (*stage)[64], int lane) {
    return km_chain_over<W, C>([=](const size_t i) { return sorted[lo + i]; }, hi - lo, stage, lane, 0.f);
}

// Long chains (many samples per centroid) without waiting a dependent-add latency per sample -- and still bit-exact.
// While the accumulator s keeps its sign and binade [2^E, 2^(E+1)), every value it takes is a multiple of u = 2^(E-23), and
//     RN(s + x) = s + u * round_to_nearest(x / u)        unless x / u falls exactly half-way between two integers
// (then the parity of s decides).  So the sum over a block of samples is the INTEGER sum of rn(x_i / u) -- any order, any
// grouping -- provided (a) no addend is an exact tie and (b) no partial sum can leave the binade, which holds whatever the
// order if  M0 + (sum of the negative integers) > 2^23  and  M0 + (sum of the positive ones) < 2^24  (M0 = |s| / u).  A block
// of 64 x K samples is therefore reduced by the whole wavefront at once; when (a) or (b) fails (early in the chain, at the
// ~20 binade changes, at a tie every few hundred thousand samples) the same block is replayed in order, 64 adds at a time,
// from the registers it was loaded into.  For the weighted coordinate sums the addend is the exact product x * w
// (48 significant bits in f64), as in fma(x, w, s).
// sum over the wavefront of an f64 (exact integers here), result in every lane: quad butterflies and row mirrors through DPP,
// the four rows through readlane -- no LDS round trips (ds_bpermute) on the per-block critical path
template <int CTRL>
__device__ __forceinline__ double dpp_f64(const double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double readlane_f64(const double v, const int l) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_f64<0xB1>(v);                                   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);                                   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);                                  // row_half_mirror
    v += dpp_f64<0x140>(v);                                  // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// The exact integer form of the chain over quarters [q0, q0 + nq) of a 1024-sample block in LDS (blk: this chain's component, wblk: the
// weights; 256 samples per quarter, four per lane): true, and acc advanced, if the accumulator keeps its binade over them whatever the
// order and no addend is an exact tie.
__device__ __forceinline__ bool km_block_exact(const float *blk, const float *wblk, const bool wx, const int q0, const int nq, float &acc, const int lane) {
    const float aa = fabsf(acc);
    if (!(aa >= 1e-30f && aa < 1e30f)) return false;
    int ex;
    (void)frexpf(aa, &ex);                                     // aa in [2^(ex-1), 2^ex)
    const double sgn = acc < 0.f ? -1.0 : 1.0;
    const double scale = ldexp(sgn, 24 - ex);                  // +-1/u with u = 2^(ex-1-23)
    const double M0 = (double)aa * fabs(scale);                // integer in [2^23, 2^24)
    double tot = 0.0, mag = 0.0, dev = 0.0;                    // sum r, sum |r|, max |t - r|
    for (int q = q0; q < q0 + nq; q++) {
        const float4 xv = *reinterpret_cast<const float4 *>(blk + q * 256 + lane * 4);
        float4 wv = make_float4(1.f, 1.f, 1.f, 1.f);
        if (wx) wv = *reinterpret_cast<const float4 *>(wblk + q * 256 + lane * 4);
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, ws[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const double x = (double)xs[e] * (double)ws[e];   // exact product (x itself for w = 1)
            const double t = x * scale;
            const double r = rint(t);
            dev = fmax(dev, fabs(t - r));
            tot += r; mag += fabs(r);
        }
    }
    tot = wave_sum_dpp(tot); mag = wave_sum_dpp(mag);
    const double pp = 0.5 * (tot + mag), nn = 0.5 * (tot - mag);
    if (!__any(dev >= 0.5) && M0 + nn >= 8388609.0 && M0 + pp <= 16777215.0) {
        acc = (float)(ldexp(M0 + tot, ex - 24) * sgn);
        return true;
    }
    return false;
}
// samples [from, to) of the block in order, 64 adds at a time, straight from the shared arrays
__device__ __forceinline__ void km_block_replay(const float *blk, const float *wblk, const bool wx, const unsigned from, const unsigned to, float &acc) {
    for (unsigned base = from; base < to; base += 64) {
        const int cnt = (int)(to - base < 64 ? to - base : 64);
        const float4 *s4 = reinterpret_cast<const float4 *>(blk + base), *w4 = reinterpret_cast<const float4 *>(wblk + base);
        if (cnt == 64) {
            float4 xv[16], wv[16];
#pragma unroll
            for (int q = 0; q < 16; q++) { xv[q] = s4[q]; if (wx) wv[q] = w4[q]; }
            if (wx) {
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    acc = __builtin_fmaf(xv[q].x, wv[q].x, acc); acc = __builtin_fmaf(xv[q].y, wv[q].y, acc);
                    acc = __builtin_fmaf(xv[q].z, wv[q].z, acc); acc = __builtin_fmaf(xv[q].w, wv[q].w, acc);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; q++) { acc += xv[q].x; acc += xv[q].y; acc += xv[q].z; acc += xv[q].w; }
            }
        } else {
            for (int q = 0; 4 * q < cnt; q++) {
                const float4 x = s4[q];
                float4 w = make_float4(0, 0, 0, 0);
                if (wx) w = w4[q];
                const float xs[4] = {x.x, x.y, x.z, x.w}, ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    if (4 * q + e < cnt) {
                        if (wx) acc = __builtin_fmaf(xs[e], ws[e], acc);
                        else acc += xs[e];
                    }
                }
            }
        }
    }
}
// One block of up to 1024 samples (len of them real, the rest of the 1024 zero): the whole block at once; a block that fails (the
// accumulator changes binade inside it, an exact tie) is taken quarter by quarter -- the quarters before and after a binade change hold
// again on their own scale -- and only a failing QUARTER is replayed in order (quarters = false: the whole block is, as before round 4:
// k_km_update 1082 -> 864 us at 67 M samples, 4.65 -> 3.42 ms over 32 iterations with a 30 % dominant colour)
__device__ __forceinline__ void km_block_step(const float *blk, const float *wblk, const bool wx, const unsigned len, const bool quarters, float &acc, const int lane) {
    if (km_block_exact(blk, wblk, wx, 0, 4, acc, lane)) return;
    if (quarters && len == 1024u) {
        for (int q = 0; q < 4; q++)
            if (!km_block_exact(blk, wblk, wx, q, 1, acc, lane)) km_block_replay(blk, wblk, wx, (unsigned)q * 256u, (unsigned)(q + 1) * 256u, acc);
        return;
    }
    km_block_replay(blk, wblk, wx, 0u, len, acc);
}

// All four chains of a centroid run this way at once and share the loads: per super-step of 4096 samples wavefront w fetches
// block w (1024 records of 16 bytes) and publishes its four components to LDS ([component][sample], 64 KB); every wavefront
// then runs ITS chain over the four blocks from there -- 4 bytes per sample read per chain instead of 16 from memory, and the
// in-order replays read the same buffer.  Two barriers per super-step; the next super-step's records are in flight meanwhile.
template <bool W>
__device__ __forceinline__ float km_chain_coop(const float4 *__restrict__ sorted, size_t lo, size_t hi, float *coop, int lane, int wid,
                                               const bool km_quarters = true) {
    constexpr int K = 16;
    constexpr size_t BS = (size_t)64 * K, SS = 4 * BS;
    const bool wx = W && wid < 3;                                          // weighted coordinate chain: acc = fma(x, w, acc)
    const bool chain = W || wid < 3;                                       // unweighted: the count is analytic, wavefront 3 only loads
    const float *mine = coop + (size_t)wid * SS, *wts = coop + 3 * SS;
    float acc = 0.f;
    float4 reg[K];
    auto fetch = [&](const size_t p0) {                                    // block `wid` of the super-step starting at p0
#pragma unroll
        for (int j = 0; j < K; j++) {
            const size_t i = p0 + (size_t)wid * BS + (size_t)j * 64 + lane;
            reg[j] = make_float4(0, 0, 0, 0);
            if (i < hi) reg[j] = sorted[i];
        }
    };
    fetch(lo);
    for (size_t p0 = lo; p0 < hi; p0 += SS) {
#pragma unroll
        for (int j = 0; j < K; j++) {
            const int o = wid * (int)BS + j * 64 + lane;
            coop[o] = reg[j].x; coop[SS + o] = reg[j].y; coop[2 * SS + o] = reg[j].z; coop[3 * SS + o] = reg[j].w;
        }
        if (p0 + SS < hi) fetch(p0 + SS);                                  // in flight while this super-step is processed
        __syncthreads();
        if (chain) {
            for (int d = 0; d < 4; d++) {
                const size_t pos = p0 + (size_t)d * BS;
                if (pos >= hi) break;                                      // wave-uniform
                const size_t bend = pos + BS < hi ? pos + BS : hi;
                const float *blk = mine + d * (int)BS, *wblk = wts + d * (int)BS;
                km_block_step(blk, wblk, wx, (unsigned)(bend - pos), km_quarters, acc, lane);
            }
        }
        __syncthreads();                                                   // the buffer is rewritten by the next super-step
    }
    return acc;
}

// The same per-block step over n samples that already sit in LDS, one array per component, zero-padded to a multiple of 1024
// (k_km_update_lists: the pieces of a long member list): exact integer sum of a block of 1024 while the accumulator keeps
// its binade, the block replayed in order otherwise.
template <bool W>
__device__ __forceinline__ float km_coop_lds(const float *mine, const float *wts, const unsigned n, const bool wx, float acc, const int lane) {
    constexpr unsigned BS = 1024;
    for (unsigned pos = 0; pos < n; pos += BS)
        km_block_step(mine + pos, wts + pos, wx, pos + BS < n ? BS : n - pos, true, acc, lane);
    return acc;
}

// Tail of a centroid's update block: res[0..3] = the four chain results; scales the centroid, publishes it, and the LAST block
// to arrive handles empty clusters and writes (y, |y|^2) for the next assignment.  All 256 threads of the block call it.
// HS: capacity of the LDS copy of the cluster sizes split_clusters works on (256 for the list path, kKMeansMaxK for the sorted
// path); 0 = any k, the copy lives in device memory (hs_mem, k floats)
template <bool W, int HS>
__device__ __forceinline__ void km_update_finish(const int kidx, const int k, const unsigned long long nx, const size_t cnt, const float *res,
                                                 int *s_last, float *cent, float *hassign, float4 *c4, unsigned int *ticket, DevMT *mt,
                                                 float *hs_mem = nullptr, unsigned int *flags = nullptr, const bool clean = false) {
    // clean (flags only): this centroid's members did not change -- its centroid and size stand as they are, nothing to publish
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float c0 = res[0], c1 = res[1], c2 = res[2], h = res[3];
    if constexpr (!W) {
        // the reference's h += 1.0f per sample is exact below 2^24 and sticks there (16777216 + 1 rounds back)
        h = cnt < (size_t)16777216 ? (float)cnt : 16777216.0f;
    }
    if (threadIdx.x == 0) {
        if (!clean) {
        if (h != 0.f) { const float norm = 1 / h; c0 *= norm; c1 *= norm; c2 *= norm; }
        // publish with write-through (sc1) stores + drained counter: no per-wave L2 write-back fence
        // (256 release fences, each flushing the XCD's dirty lines, cost more than the chains themselves)
        __hip_atomic_store(&cent[3 * kidx], c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&cent[3 * kidx + 1], c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&cent[3 * kidx + 2], c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&hassign[kidx], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // its record of the next assignment's table too (read by the NEXT kernel): the last block rebuilds the table only after
        // split_clusters has changed centroids -- before, one wavefront re-read all k centroids in the tail of every launch
        km_store_c4(c4, k, kidx, c0, c1, c2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = (tk == (unsigned)k - 1u) ? 1 : 0;
    }
    __syncthreads();
    if (!*s_last || wid != 0) return;                                     // terminated wavefronts do not count at later barriers
    // last block: every centroid was published with sc1 stores; read the sizes back with sc1 (L1-bypassing) loads, and with
    // them the dirty flags (every block read its own before it took its ticket): all requests of a batch go out before the
    // first answer is looked at
    bool mine_empty = false, mine_dirty = false;
    for (int cb = 0; cb < k; cb += 256) {
        float hv[4]; unsigned fv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ci = cb + 64 * q + lane;
            hv[q] = 1.f; fv[q] = 0u;
            if (ci < k) {
                hv[q] = __hip_atomic_load(&hassign[ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (flags) fv[q] = __hip_atomic_load(&flags[1 + ci], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ci = cb + 64 * q + lane;
            mine_empty = mine_empty || hv[q] == 0.f;
            mine_dirty = mine_dirty || fv[q] != 0u;
            if (flags && ci < k) __hip_atomic_store(&flags[1 + ci], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // cleared for the next assignment
        }
    }
    const bool any = __ballot(mine_empty) != 0ULL;
    if (lane == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next iteration's launch
    if (flags) {
        // no sample moved and no cluster is empty: the centroids are a fixed point of the iteration
        const bool any_dirty = __ballot(mine_dirty) != 0ULL;
        if (!any_dirty && !any && lane == 0) __hip_atomic_store(&flags[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!any) return;                                                     // wave-uniform
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the cleared flags are out before split_clusters raises its own
        __shared__ unsigned s_mt[624];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                // the split code uses plain accesses
        if constexpr (HS > 0) {
            __shared__ float s_hs[HS];
            km_split_clusters_wave<false>(cent, hassign, k, nx, mt, s_mt, s_hs, lane, flags);
        } else km_split_clusters_wave<true>(cent, hassign, k, nx, mt, s_mt, hs_mem, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (int j = lane; j < k; j += 64) {                                  // split_clusters moved centroids: the whole table again
        const float v0 = __hip_atomic_load(&cent[3 * j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float v1 = __hip_atomic_load(&cent[3 * j + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float v2 = __hip_atomic_load(&cent[3 * j + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        km_store_c4(c4, k, j, v0, v1, v2);
    }
}

template <bool W, bool BIGK = false>
__global__ __launch_bounds__(256) void k_km_update(const float4 *__restrict__ sorted, const unsigned int *__restrict__ rowtot, int k,
                                                  unsigned long long nx, float *cent, float *hassign, float4 *c4,
                                                  unsigned int *ticket, DevMT *mt, unsigned long long long_min, float *hs_mem, const int quarters) {
    __shared__ float4 stage[4][2][64];
    extern __shared__ __attribute__((aligned(16))) float coop[];          // km_chain_coop: [4 components][4096 samples] when launched with it
    __shared__ float res[4];
    __shared__ int s_last;
    const int kidx = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // segment of this centroid = prefix of the row totals
    unsigned pre = 0;
    for (int j = lane; j < kidx; j += 64) pre += rowtot[j];
    pre = wave_sum_u32(pre);
    pre = __shfl(pre, 0, 64);
    const size_t lo = pre, hi = lo + rowtot[kidx];
    float acc = 0.f;
    if ((size_t)(hi - lo) >= long_min) {                                  // block-uniform (long_min is huge without the LDS buffer)
        acc = km_chain_coop<W>(sorted, lo, hi, coop, lane, wid, quarters != 0);
    } else {
        if (wid == 0) acc = km_chain<W, 0>(sorted, lo, hi, stage[0], lane);
        else if (wid == 1) acc = km_chain<W, 1>(sorted, lo, hi, stage[1], lane);
        else if (wid == 2) acc = km_chain<W, 2>(sorted, lo, hi, stage[2], lane);
        else if (W) acc = km_chain<W, 3>(sorted, lo, hi, stage[3], lane);
    }
    if (lane == 0) res[wid] = acc;
    __syncthreads();
    km_update_finish<W, BIGK ? 0 : kKMeansMaxK>(kidx, k, nx, hi - lo, res, &s_last, cent, hassign, c4, ticket, mt, hs_mem);
}

// --------------------------------------------------------------------------------------------
// Few samples (the default: 512^2 of them, ~1000 per centroid): the global stable counting sort costs more than the sums
// it feeds (three launches and a k x chunks table per iteration).  Here the assignment kernel sorts each block of 1024
// samples LOCALLY (stable counting sort in LDS: sample numbers grouped by centroid, sample order kept inside a group, plus
// the 257 group offsets of the block), and ONE block per centroid then collects its members block by block -- two offsets
// and ~4 sample numbers per block of samples instead of the block's 1024 assignments -- into a list in LDS and replays the
// centroid's sequential f32 chains from there, each wavefront gathering just its own coordinate.  Same sums in the same
// order as k_km_update; two launches per iteration, nothing else moves.
// --------------------------------------------------------------------------------------------
constexpr int kKmDirectCap = 4096;                                     // listed member records between two chain replays (64 KB of LDS)
constexpr int kKmSortBlock = 1024;                                     // samples sorted together by one block of k_km_assign_sort

// inclusive prefix over the threads of a block of up to 1024 (wsum: 16 words of LDS); *total = the block's sum
__device__ __forceinline__ unsigned block_scan_incl_u32(const unsigned v, unsigned *wsum, unsigned *total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);
    const unsigned inc = wave_scan_incl_u32(v);
    __syncthreads();                                                   // wsum may still be read from a previous call
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    unsigned pre = 0, all = 0;
    for (int w = 0; w < nw; w++) { const unsigned t = wsum[w]; all += t; pre += w < wid ? t : 0u; }
    *total = all;
    return pre + inc;
}

// km_assign_one with two centroids per instruction: v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 round each half like the scalar
// instruction, so centroids 2p and 2p+1 (SIMD lanes l, l+1 of the reference's kernel) take four packed instructions instead of
// eight; the pairwise table comes through the scalar cache like the plain one.
typedef float f2_t __attribute__((ext_vector_type(2)));
typedef float f8_t __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(4))) f8_t *scalar_c8_t;
__device__ __forceinline__ int km_assign_one_pk(const float x0, const float x1, const float x2, const scalar_c4_t c4, const scalar_c8_t c8, const int k) {
    const f2_t m0 = {-2 * x0, -2 * x0}, m1 = {-2 * x1, -2 * x1}, m2 = {-2 * x2, -2 * x2};
    const float xn = __builtin_fmaf(x2, x2, __builtin_fmaf(x0, x0, x1 * x1));
    float ld[8]; int lb[8];
#pragma unroll
    for (int l = 0; l < 8; l++) { ld[l] = 3.402823466e+38F - xn; lb[l] = -l; }
    const int ny_p = (k / 8) * 8;
    for (int j = 0; j < ny_p; j += 8) {
#pragma unroll
        for (int l = 0; l < 8; l += 2) {
            const f8_t y = c8[(j + l) >> 1];                    // wave-uniform address: scalar load
            const f2_t y0 = {y[0], y[1]}, y1 = {y[2], y[3]}, y2 = {y[4], y[5]}, yw = {y[6], y[7]};
            f2_t dp = m0 * y0;
            dp = __builtin_elementwise_fma(m1, y1, dp);
            dp = __builtin_elementwise_fma(m2, y2, dp);
            dp = dp + yw;
            if (dp[0] < ld[l]) { ld[l] = dp[0]; lb[l] = j; }
            if (dp[1] < ld[l + 1]) { ld[l + 1] = dp[1]; lb[l + 1] = j; }
        }
    }
    float cur_d = 3.402823466e+38F; unsigned cur_i = 0xFFFFFFFFu;
#pragma unroll
    for (int l = 0; l < 8; l++) {
        const unsigned li = (unsigned)(lb[l] + l);
        float cand = ld[l] + xn;
        if (cand < 0) cand = 0;
        if (cur_d > cand) { cur_d = cand; cur_i = li; }
        else if (cur_d == cand && cur_i > li) cur_i = li;
    }
    for (int j0 = ny_p; j0 < k; j0++) {                     // simdlib_based.cpp:201-216
        const auto y = c4[j0];
        float dp = __builtin_fmaf(x2, y.z, __builtin_fmaf(x1, y.y, x0 * y.x));
        float d = xn + y.w - 2 * dp;
        if (d < 0) d = 0;
        if (cur_d > d) { cur_d = d; cur_i = (unsigned)j0; }
    }
    return (int)cur_i;
}

// The same top-1 with HALF the instructions (the scan is what k_km_assign_sort spends its time on: sixteen wavefronts per CU at
// ~1600 VALU instructions each, four cycles per instruction on a 16-lane SIMD = 13 us of an 18 us launch).  The reference's
// result is a function of the eight lane trackers' MINIMA only, so pass 1 keeps nothing but the minima -- v_min3_f32 folds the
// two candidates a tracker meets in sixteen centroids into it in one instruction: 2.5 instructions per centroid instead of
// 6.25 -- and remembers, per tracker, the last QUARTER of the table in which its minimum went down (compare + select per
// tracker and quarter).  The merge of the trackers needs an index only from the tracker(s) that attain the smallest clamped
// distance -- one, unless two trackers tie bit for bit -- and that index is the FIRST entry of the remembered quarter whose
// value equals the minimum (strict '<' kept the first): pass 2 re-evaluates that tracker's entries of that quarter (8 of 256)
// with the same four operations, from a copy of the table in LDS (per-lane addresses).  Bit-identical to km_assign_one.
__device__ __forceinline__ float km_min3(const float a, const float b, const float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// NS samples per lane share every scalar load of the table.  What the scan waits for besides its arithmetic is the scalar cache:
// scalar loads return out of order, so a wavefront waits for ALL it has issued before it computes (no overlap inside a
// wavefront), and a round trip of eight 32-byte loads cost ~1000 cycles with sixteen wavefronts per CU asking.  The table unit of
// sixteen centroids is therefore fetched as four 64-byte loads: 19 300 -> 15 100 cycles for the scan (s_memtime inside the kernel,
// tools/diag/km_trace.py), of which 16 x 640 are the SIMD's own issue time.
template <int NS>
__device__ __forceinline__ void km_assign_min3(const float (&x0)[NS], const float (&x1)[NS], const float (&x2)[NS], const scalar_c4_t c4,
                                               const scalar_c8_t c8, const float4 *c4s, const int k, int (&out)[NS]) {
    f2_t m0[NS], m1[NS], m2[NS];
    float xn[NS], ld[NS][8]; int lc[NS][8];
#pragma unroll
    for (int e = 0; e < NS; e++) {
        m0[e] = f2_t{-2 * x0[e], -2 * x0[e]}; m1[e] = f2_t{-2 * x1[e], -2 * x1[e]}; m2[e] = f2_t{-2 * x2[e], -2 * x2[e]};
        xn[e] = __builtin_fmaf(x2[e], x2[e], __builtin_fmaf(x0[e], x0[e], x1[e] * x1[e]));
#pragma unroll
        for (int l = 0; l < 8; l++) { ld[e][l] = 3.402823466e+38F - xn[e]; lc[e][l] = -1; }
    }
    const int ny_p = (k / 8) * 8;
    const int U = ny_p >> 4;                                  // units of sixteen centroids; a trailing group of eight is "quarter" 4
    const int CUN = (U + 3) >> 2;                             // units per quarter
    auto pair_dp = [&](const f8_t y, const int e) {
        const f2_t y0 = {y[0], y[1]}, y1 = {y[2], y[3]}, y2 = {y[4], y[5]}, yw = {y[6], y[7]};
        f2_t dp = m0[e] * y0;
        dp = __builtin_elementwise_fma(m1[e], y1, dp);
        dp = __builtin_elementwise_fma(m2[e], y2, dp);
        return dp + yw;
    };
    for (int c = 0; c < 4; c++) {
        const int u0 = c * CUN, u1 = u0 + CUN < U ? u0 + CUN : U;
        if (u0 >= u1) break;                                  // wave-uniform
        float snap[NS][8];
#pragma unroll
        for (int e = 0; e < NS; e++)
#pragma unroll
            for (int l = 0; l < 8; l++) snap[e][l] = ld[e][l];
        for (int u = u0; u < u1; u++) {
            const int p0 = u << 3;                            // pair index of centroid 16 u
            // the unit's 256 bytes as four 64-byte scalar loads (a scalar load is a round trip the wavefront cannot overlap with
            // its own arithmetic -- they return out of order, so it waits for all of them: fewer, wider requests)
            typedef float f16_t __attribute__((ext_vector_type(16)));
            typedef const __attribute__((address_space(4))) f16_t *scalar_c16_t;
            const scalar_c16_t c16 = (scalar_c16_t)c8;
            const f16_t q0 = c16[p0 >> 1], q1 = c16[(p0 >> 1) + 1], q2 = c16[(p0 >> 1) + 2], q3 = c16[(p0 >> 1) + 3];
            auto half = [](const f16_t q, const int h) { return h ? f8_t{q[8], q[9], q[10], q[11], q[12], q[13], q[14], q[15]} : f8_t{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]}; };
#pragma unroll
            for (int l = 0; l < 8; l += 2) {
                const f8_t ya = half(l < 4 ? q0 : q1, (l >> 1) & 1), yb = half(l < 4 ? q2 : q3, (l >> 1) & 1);
#pragma unroll
                for (int e = 0; e < NS; e++) {
                    const f2_t a = pair_dp(ya, e), b = pair_dp(yb, e);
                    ld[e][l] = km_min3(ld[e][l], a[0], b[0]);
                    ld[e][l + 1] = km_min3(ld[e][l + 1], a[1], b[1]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < NS; e++)
#pragma unroll
            for (int l = 0; l < 8; l++) lc[e][l] = ld[e][l] < snap[e][l] ? c : lc[e][l];
    }
    if (ny_p & 8) {                                           // the trailing group of eight
        const int p0 = U << 3;
#pragma unroll
        for (int l = 0; l < 8; l += 2) {
            const f8_t ya = c8[p0 + (l >> 1)];
#pragma unroll
            for (int e = 0; e < NS; e++) {
                const f2_t a = pair_dp(ya, e);
                if (a[0] < ld[e][l]) { ld[e][l] = a[0]; lc[e][l] = 4; }
                if (a[1] < ld[e][l + 1]) { ld[e][l + 1] = a[1]; lc[e][l + 1] = 4; }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NS; e++) {
        // merge of the lane trackers (simdlib_based.cpp:178-199): smallest clamped distance, smallest index among those attaining it
        float cand[8];
        float cur_d = 3.402823466e+38F;
#pragma unroll
        for (int l = 0; l < 8; l++) { float cd = ld[e][l] + xn[e]; if (cd < 0) cd = 0; cand[l] = cd; cur_d = cd < cur_d ? cd : cur_d; }
        unsigned tm = 0u;
#pragma unroll
        for (int l = 0; l < 8; l++) tm |= cand[l] == cur_d ? (1u << l) : 0u;
        unsigned cur_i = 0xFFFFFFFFu;
        while (__any(tm != 0u)) {                             // one trip unless a lane has two trackers tied
            if (tm) {
                const int l = __ffs((int)tm) - 1;
                tm &= tm - 1u;
                float target = ld[e][0]; int c = lc[e][0];
#pragma unroll
                for (int q = 1; q < 8; q++) { target = l == q ? ld[e][q] : target; c = l == q ? lc[e][q] : c; }
                unsigned idx = 0u;                            // a tracker that never went down holds index 0 (lb = -l)
                if (c == 4) idx = (unsigned)((U << 4) + l);
                else if (c >= 0) {
                    const int u0 = c * CUN, u1 = u0 + CUN < U ? u0 + CUN : U;
                    const int base = (u0 << 4) + l, lim = u1 << 4;
                    for (int q = 2 * CUN - 1; q >= 0; q--) {  // descending: the first entry that attains the minimum wins
                        const int j = base + 8 * q;
                        if (j < lim) {
                            const float4 y = c4s[j];
                            float dp = (-2 * x0[e]) * y.x;
                            dp = __builtin_fmaf(-2 * x1[e], y.y, dp);
                            dp = __builtin_fmaf(-2 * x2[e], y.z, dp);
                            dp = dp + y.w;
                            if (dp == target) idx = (unsigned)j;
                        }
                    }
                }
                cur_i = idx < cur_i ? idx : cur_i;
            }
        }
        for (int j0 = ny_p; j0 < k; j0++) {                   // simdlib_based.cpp:201-216
            const auto y = c4[j0];
            float dp = __builtin_fmaf(x2[e], y.z, __builtin_fmaf(x1[e], y.y, x0[e] * y.x));
            float d = xn[e] + y.w - 2 * dp;
            if (d < 0) d = 0;
            if (cur_d > d) { cur_d = d; cur_i = (unsigned)j0; }
        }
        out[e] = cur_i < (unsigned)k ? (int)cur_i : 0;
    }
}

#ifdef PAMD_KM_TRACE
// diagnostic build only (make TRACE=1): per-block phase timestamps of the two kernels of the default KMeans iteration
__device__ unsigned long long g_km_trace[2][256][32];
#define KM_TRACE(kern, slot) do { if ((threadIdx.x & 63) == 0) g_km_trace[kern][blockIdx.x & 255][slot] = wall_clock64(); } while (0)
#define KM_TRACE0(kern, slot) do { if (threadIdx.x == 0) g_km_trace[kern][blockIdx.x & 255][slot] = wall_clock64(); } while (0)
#else
#define KM_TRACE(kern, slot) do { } while (0)
#define KM_TRACE0(kern, slot) do { } while (0)
#endif

// NS = samples per lane: 1 = sixteen wavefronts and the packed full scan (km_assign_one_pk, the round-3 form); 2 = eight wavefronts,
// each taking two of the sixteen runs, with the min3 scan
template <int NS>
__global__ __launch_bounds__(1024 / NS) void k_km_assign_sort(KmSamples s, size_t nx, const float4 *__restrict__ c4, int k, const bool weighted,
                                                              float4 *__restrict__ sorted_rec, unsigned short *__restrict__ offs,
                                                              unsigned char *__restrict__ prev, unsigned int *flags, const bool min3) {
    if (flags[0]) return;                                              // fixed point reached (k_km_prep's comment): nothing changes any more
    __shared__ unsigned char cnt[16][256];                             // members of (step, centroid): 64 at most
    __shared__ unsigned short stepbase[16][256];                       // start of centroid j's group + its members in earlier steps
    __shared__ unsigned int wsum[16];
    __shared__ float4 c4s[256];                                        // the centroid records once more, for per-lane reads (km_assign_min3)
    constexpr int NT = 1024 / NS;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    KM_TRACE0(0, 0);
    for (int i = threadIdx.x; i < 16 * 256 / 4; i += NT) reinterpret_cast<unsigned int *>(&cnt[0][0])[i] = 0u;
    if ((int)threadIdx.x < k) c4s[threadIdx.x] = c4[threadIdx.x];
    __syncthreads();
    KM_TRACE0(0, 1);
    const scalar_c4_t t4 = (scalar_c4_t)(unsigned long long)c4;
    const scalar_c8_t t8 = (scalar_c8_t)(unsigned long long)(c4 + k);
    const size_t blk0 = (size_t)blockIdx.x * kKmSortBlock;
    size_t i[NS]; bool v[NS]; float4 rec[NS]; int step[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) {
        step[e] = wv * NS + e;                                         // the block's run of 64 consecutive samples this lane's e-th sample is in
        i[e] = blk0 + (size_t)step[e] * 64 + lane;
        v[e] = i[e] < nx;
        rec[e] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (v[e]) { rec[e].x = s.x[i[e]]; rec[e].y = s.y[i[e]]; rec[e].z = s.z[i[e]]; if (weighted) rec[e].w = s.w[i[e]]; }
    }
#ifdef PAMD_KM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KM_TRACE0(0, 2);
    if (threadIdx.x == 0) g_km_trace[0][blockIdx.x & 255][24] = clock64();
#endif
    int a[NS];
    if (NS == 1 && !min3) a[0] = v[0] ? km_assign_one_pk(rec[0].x, rec[0].y, rec[0].z, t4, t8, k) : 0;
    else {
        float x0[NS], x1[NS], x2[NS];
#pragma unroll
        for (int e = 0; e < NS; e++) { x0[e] = rec[e].x; x1[e] = rec[e].y; x2[e] = rec[e].z; }
        km_assign_min3<NS>(x0, x1, x2, t4, t8, c4s, k, a);
    }
    unsigned rk[NS];
#pragma unroll
    for (int e = 0; e < NS; e++) {
        if (!v[e]) a[e] = 0;
        if (v[e]) {                                                    // a sample that changed sides marks both centroids dirty
            const int old = prev[i[e]];
            if (old != a[e]) { prev[i[e]] = (unsigned char)a[e]; flags[1 + old] = 1u; flags[1 + a[e]] = 1u; }
        }
        const unsigned long long m = match_mask(a[e], 8, __ballot(v[e]));
        rk[e] = (unsigned)__popcll(m & ((1ULL << lane) - 1ULL));       // rank among the step's samples of the same centroid
        if (v[e] && rk[e] == 0u) cnt[step[e]][a[e]] = (unsigned char)__popcll(m);   // group leader
    }
#ifdef PAMD_KM_TRACE
    if (lane == 0) g_km_trace[0][blockIdx.x & 255][8 + wv] = wall_clock64();
    if (threadIdx.x == 0) g_km_trace[0][blockIdx.x & 255][25] = clock64();
#endif
    __syncthreads();
    KM_TRACE0(0, 3);
    // thread j < 256: centroid j's counts over the sixteen steps -> exclusive prefix; then the groups' starts by a block scan
    unsigned tot = 0;
    unsigned short pre[16];
    if (threadIdx.x < 256) {
#pragma unroll
        for (int st = 0; st < 16; st++) { pre[st] = (unsigned short)tot; tot += cnt[st][threadIdx.x]; }
    }
    unsigned total = 0;
    const unsigned start = block_scan_incl_u32(tot, wsum, &total) - tot;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int st = 0; st < 16; st++) stepbase[st][threadIdx.x] = (unsigned short)(start + pre[st]);
        unsigned short *ob = offs + (size_t)blockIdx.x * 257;
        ob[threadIdx.x] = (unsigned short)start;
        if (threadIdx.x == 255) ob[256] = (unsigned short)total;
    }
    __syncthreads();
    KM_TRACE0(0, 4);
#pragma unroll
    for (int e = 0; e < NS; e++)
        if (v[e]) sorted_rec[blk0 + (unsigned)stepbase[step[e]][a[e]] + rk[e]] = rec[e];   // the sample itself: the update reads its members in runs
#ifdef PAMD_KM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    KM_TRACE0(0, 5);
#endif
}

template <bool W>
__global__ __launch_bounds__(256) void k_km_update_lists(const float4 *__restrict__ sorted_rec, const unsigned short *__restrict__ offs,
                                                        unsigned long long nx, int k, float *cent, float *hassign, float4 *c4,
                                                        unsigned int *ticket, DevMT *mt, unsigned int *flags) {
    if (flags[0]) return;                                                 // fixed point reached: the centroids stand
    __shared__ float4 stage[4][2][64];
    extern __shared__ __attribute__((aligned(16))) float members[];       // [4][kKmDirectCap]: the listed members in sample order, one array per component
    __shared__ float res[4];
    __shared__ int s_last;
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int n_long, long_src[256], long_lo[256], long_nb[256];   // the long runs of the current pass
    const int kidx = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float *const mx = members, *const my = members + kKmDirectCap, *const mz = members + 2 * kKmDirectCap, *const mw = members + 3 * kKmDirectCap;
    unsigned int *const srcidx = reinterpret_cast<unsigned int *>(members + 4 * kKmDirectCap);   // [kKmDirectCap]: where a long run's records come from
    float acc = 0.f;
    size_t total = 0;
    unsigned fill = 0;                                                    // block-uniform
    KM_TRACE0(1, 0);
    bool long_list = false;                                               // a full piece has gone by: this centroid holds thousands of samples
    auto replay = [&]() {                                                 // the chains over the fill listed members, continuing `acc`
        const unsigned n = fill;
        if (n == (unsigned)kKmDirectCap || long_list) {
            // Long lists (a dominant colour: tens of thousands of members where noise gives ~1000) take the block-parallel exact
            // form of the chain, 1024 members at a time (km_chain_coop's step): ~0.7 us per block instead of 4.2 in sequence
            const unsigned padded = (n + 1023u) & ~1023u;
            for (unsigned i = n + threadIdx.x; i < padded; i += 256) { mx[i] = 0.f; my[i] = 0.f; mz[i] = 0.f; mw[i] = 0.f; }
            __syncthreads();
            if ((W || wid < 3) && n) acc = km_coop_lds<W>(members + (size_t)wid * kKmDirectCap, mw, n, W && wid < 3, acc, lane);
            long_list = true;
            return;
        }
        auto rec = [&](const size_t i) { return make_float4(mx[i], my[i], mz[i], mw[i]); };
        if (wid == 0) acc = km_chain_over<W, 0>(rec, n, stage[0], lane, acc);
        else if (wid == 1) acc = km_chain_over<W, 1>(rec, n, stage[1], lane, acc);
        else if (wid == 2) acc = km_chain_over<W, 2>(rec, n, stage[2], lane, acc);
        else if (W) acc = km_chain_over<W, 3>(rec, n, stage[3], lane, acc);
    };
    const bool clean = flags[1 + kidx] == 0u;                             // block-uniform; read before this block takes its ticket
    const unsigned long long nblocks = clean ? 0ULL : (nx + kKmSortBlock - 1) / kKmSortBlock;
    for (unsigned long long c0 = 0; c0 < nblocks; c0 += 256) {            // 256 blocks of samples at a time: one per thread
        const unsigned long long b = c0 + threadIdx.x;
        unsigned o0 = 0, nb = 0;
        if (b < nblocks) { o0 = offs[b * 257 + kidx]; nb = (unsigned)offs[b * 257 + kidx + 1] - o0; }
        unsigned chunk_total = 0;
        const unsigned excl = block_scan_incl_u32(nb, wsum, &chunk_total) - nb;
        // A run of more than kLongRun records (a dominant colour: hundreds of a block's 1024 samples in one centroid) is not
        // copied by its own thread, four records at a time while the others idle, but by the whole block, 256 records at a time
        constexpr unsigned kLongRun = 32;
        const bool is_long = nb > kLongRun;
        if (threadIdx.x == 0) n_long = 0u;
        __syncthreads();
        if (is_long) { const unsigned e = atomicAdd(&n_long, 1u); long_src[e] = (unsigned)(b * kKmSortBlock) + o0; long_lo[e] = excl; long_nb[e] = nb; }
        __syncthreads();
        const unsigned nl = n_long;
        unsigned done = 0;
        while (done < chunk_total) {                                      // block-uniform; one trip unless the list fills up
            const unsigned take = min((unsigned)kKmDirectCap - fill, chunk_total - done);
            if (!is_long) {
                const unsigned lo = max(excl, done), hi = min(excl + nb, done + take);
                const float4 *src = sorted_rec + b * kKmSortBlock + o0;  // this block of samples' members: one contiguous run
                for (unsigned q = lo; q < hi; q += 4) {                   // four independent 16-byte loads per trip
                    float4 r4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) r4[u] = q + u < hi ? src[q + u - excl] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 4; u++) if (q + u < hi) { const unsigned i = fill + q + u - done; mx[i] = r4[u].x; my[i] = r4[u].y; mz[i] = r4[u].z; mw[i] = r4[u].w; }
                }
            }
            if (nl) {                                                     // block-uniform
                // the long runs' records: first WHERE each comes from (LDS only), then all of them in one flat sweep with eight
                // independent loads per thread in flight -- run by run, every run would cost a memory round trip of its own
                for (unsigned i = threadIdx.x; i < take; i += 256) srcidx[fill + i] = ~0u;
                __syncthreads();
                for (unsigned e = 0; e < nl; e++) {
                    const unsigned ex0 = long_lo[e], lo = max(ex0, done), hi = min(ex0 + long_nb[e], done + take);
                    for (unsigned q = lo + threadIdx.x; q < hi; q += 256) srcidx[fill + q - done] = long_src[e] + (q - ex0);
                }
                __syncthreads();
                for (unsigned i0 = threadIdx.x; i0 < take; i0 += 256 * 8) {
                    unsigned ix[8]; float4 r[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const unsigned i = i0 + 256u * u; ix[u] = i < take ? srcidx[fill + i] : ~0u; }
#pragma unroll
                    for (int u = 0; u < 8; u++) r[u] = ix[u] != ~0u ? sorted_rec[ix[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int u = 0; u < 8; u++) if (ix[u] != ~0u) { const unsigned i = fill + i0 + 256u * u; mx[i] = r[u].x; my[i] = r[u].y; mz[i] = r[u].z; mw[i] = r[u].w; }
                }
            }
            fill += take; done += take;
            __syncthreads();
            if (fill == (unsigned)kKmDirectCap) { replay(); fill = 0; __syncthreads(); }
        }
        total += chunk_total;
    }
    __syncthreads();
    KM_TRACE0(1, 1);
    replay();
#ifdef PAMD_KM_TRACE
    if (lane == 0) g_km_trace[1][blockIdx.x & 255][8 + wid] = wall_clock64();
#endif
    if (lane == 0) res[wid] = acc;
    __syncthreads();
    KM_TRACE0(1, 2);
    km_update_finish<W, 256>(kidx, k, nx, total, res, &s_last, cent, hassign, c4, ticket, mt, nullptr, flags, clean);   // this path runs for k <= 256 only
    KM_TRACE0(1, 3);
}

// --------------------------------------------------------------------------------------------
// Order-free update (option, kmeans.h KmSums): per-centroid sums as 64-bit fixed-point integers.  An addend v (|v| <= 2^E, at
// most 2^P of them) is rounded to the nearest multiple of the quantum 2^Q, Q = E + P - 62 (35 bits below the bound at 2^27
// samples), by the magic-number add, and summed as an integer: LDS atomics per block, one global 64-bit atomic per centroid,
// quantity and block at the end -- any order gives the same bits.  13 (17) bytes per sample.
// --------------------------------------------------------------------------------------------
struct KmFix { double magic_x, quantum_x, magic_w, quantum_w; };
__device__ __forceinline__ long long km_fix_quant(const double v, const double magic) {       // v / quantum to nearest-even
    return __double_as_longlong(v + magic) - __double_as_longlong(magic);
}
template <bool W, typename AT>
__global__ __launch_bounds__(256) void k_km_accum(KmSamples s, const AT *__restrict__ assign, size_t nx, int k, KmFix f,
                                                  unsigned long long *gsum /* [k][4] */) {
    extern __shared__ unsigned long long acc64[];                                               // [k][4]
    for (int i = threadIdx.x; i < 4 * k; i += 256) acc64[i] = 0ULL;
    __syncthreads();
    auto add = [&](const unsigned a, const float x, const float y, const float z, const float w) {
        const double wd = W ? (double)w : 1.0;                                                  // f32 * f32: exact in f64
        unsigned long long *r = acc64 + 4 * (size_t)a;
        atomicAdd(r, (unsigned long long)km_fix_quant((double)x * wd, f.magic_x));
        atomicAdd(r + 1, (unsigned long long)km_fix_quant((double)y * wd, f.magic_x));
        atomicAdd(r + 2, (unsigned long long)km_fix_quant((double)z * wd, f.magic_x));
        atomicAdd(r + 3, W ? (unsigned long long)km_fix_quant(wd, f.magic_w) : 1ULL);
    };
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; base < nx; base += stride) {
        if (base + 4 <= nx) {                                                                   // the arrays are 16-byte aligned
            const float4 vx = *reinterpret_cast<const float4 *>(s.x + base), vy = *reinterpret_cast<const float4 *>(s.y + base),
                         vz = *reinterpret_cast<const float4 *>(s.z + base);
            float4 vw = make_float4(1.f, 1.f, 1.f, 1.f);
            if constexpr (W) vw = *reinterpret_cast<const float4 *>(s.w + base);
            unsigned a0, a1, a2, a3;
            if constexpr (sizeof(AT) == 1) {
                const unsigned pk = *reinterpret_cast<const unsigned int *>(assign + base);
                a0 = pk & 0xffu; a1 = (pk >> 8) & 0xffu; a2 = (pk >> 16) & 0xffu; a3 = pk >> 24;
            } else {
                const int4 pk = *reinterpret_cast<const int4 *>(assign + base);
                a0 = (unsigned)pk.x; a1 = (unsigned)pk.y; a2 = (unsigned)pk.z; a3 = (unsigned)pk.w;
            }
            add(a0, vx.x, vy.x, vz.x, vw.x); add(a1, vx.y, vy.y, vz.y, vw.y);
            add(a2, vx.z, vy.z, vz.z, vw.z); add(a3, vx.w, vy.w, vz.w, vw.w);
        } else {
            for (size_t i = base; i < nx; i++) add((unsigned)assign[i], s.x[i], s.y[i], s.z[i], W ? s.w[i] : 1.f);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * k; i += 256) { const unsigned long long v = acc64[i]; if (v) atomicAdd(&gsum[i], v); }
}

// One block: sums -> centroids (c = f32(sum) * (1 / h), h = the count or the weight sum as a float), empty clusters (Clustering.cpp:216-263), the (y, |y|^2) records of the next assignment; zeroes the sums.
template <bool W>
__global__ __launch_bounds__(256) void k_km_update_sums(unsigned long long *gsum, int k, unsigned long long nx, KmFix f, float *cent,
                                                        float *hassign, float4 *c4, DevMT *mt) {
    __shared__ int s_empty;
    if (threadIdx.x == 0) s_empty = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        const double sx = (double)(long long)gsum[4 * j] * f.quantum_x, sy = (double)(long long)gsum[4 * j + 1] * f.quantum_x,
                     sz = (double)(long long)gsum[4 * j + 2] * f.quantum_x;
        float h;
        if constexpr (W) h = (float)((double)(long long)gsum[4 * j + 3] * f.quantum_w);
        else h = (float)gsum[4 * j + 3];                                                        // (the reference's count sticks at 2^24)
        float c0 = (float)sx, c1 = (float)sy, c2 = (float)sz;
        if (h != 0.f) { const float norm = 1 / h; c0 *= norm; c1 *= norm; c2 *= norm; }
        else s_empty = 1;
        cent[3 * j] = c0; cent[3 * j + 1] = c1; cent[3 * j + 2] = c2; hassign[j] = h;
        gsum[4 * j] = 0ULL; gsum[4 * j + 1] = 0ULL; gsum[4 * j + 2] = 0ULL; gsum[4 * j + 3] = 0ULL;
    }
    __threadfence();
    __syncthreads();
    if (s_empty && threadIdx.x < 64) {
        __shared__ unsigned s_mt[624];
        __shared__ float s_hs[kKMeansMaxK];
        km_split_clusters_wave(cent, hassign, k, nx, mt, s_mt, s_hs, (int)threadIdx.x);
        __threadfence();
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) km_store_c4(c4, k, j, cent[3 * j], cent[3 * j + 1], cent[3 * j + 2]);
}

// --------------------------------------------------------------------------------------------
static int chunk_len_for(size_t nx) {
    size_t c = ceil_div(nx, 4096);                 // at most ~4096 chunks (one wavefront each)
    c = ceil_div(c, 64) * 64;
    if (c < 64) c = 64;
    return (int)c;
}

void KMeansWork::reserve(size_t nx, int k) {
    sx.reserve(nx); sy.reserve(nx); sz.reserve(nx); sw.reserve(nx);
    assign.reserve(nx); sorted.reserve(nx);
    const int nchunks = (int)ceil_div(nx, (size_t)chunk_len_for(nx));
    table.reserve(std::max((size_t)k * (size_t)(nchunks > 0 ? nchunks : 1),
                           (ceil_div(nx, (size_t)1024) * 257 + 1) / 2 + 1));        // also the u16 group offsets of k_km_assign_sort
    rowtot.reserve(k);
    if (k > kKMeansMaxK) { rowbase.reserve(k); hs.reserve(k); }
    cent.reserve(3 * (size_t)k); hassign.reserve(k); c4.reserve(2 * (size_t)k + 2);        // + the pairwise copy (km_store_c4)
    perm.reserve(nx);
    if (!mt.p) { mt.reserve(1); mt_seeded = false; }
    if (!ticket.p) { ticket.reserve(1); HIP_CHECK(hipMemset(ticket.p, 0, sizeof(unsigned int))); }
    flags.reserve(2 + 256);
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

void kmeans_gather_slice(const double *d_planar, size_t n_local, bool weighted, const int *d_perm, size_t nx, size_t begin,
                         KMeansWork &w, hipStream_t s) {
    KmSamples ks{w.sx.p, w.sy.p, w.sz.p, w.sw.p};
    size_t g = ceil_div(nx, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    KTIME("k_km_gather", s, (weighted ? 48.0 : 36.0) * nx);
    if (weighted) hipLaunchKernelGGL(k_km_gather_slice<true>, (int)g, 256, 0, s, d_planar, n_local, d_perm, nx, begin, ks);
    else hipLaunchKernelGGL(k_km_gather_slice<false>, (int)g, 256, 0, s, d_planar, n_local, d_perm, nx, begin, ks);
    HIP_CHECK(hipGetLastError());
}

void kmeans_iterate(KMeansWork &w, size_t nx, int k, bool weighted, int niter, hipStream_t s, size_t expect_longest, const KmSums *sums) {
    KmSamples ks{w.sx.p, w.sy.p, w.sz.p, w.sw.p};
    KmFix fix{};
    int ablocks = 1;
    if (sums && k > kKMeansMaxK) sums = nullptr;     // the order-free option keeps its sums in LDS (32 bytes per centroid): beyond that the exact update
    if (sums) {
        int ex = 0, ew = 0, P = 1;
        (void)frexp(std::max(sums->bound_x, 1e-300) * (weighted ? std::max(sums->bound_w, 1e-300) : 1.0), &ex);   // |w x| <= 2^ex
        (void)frexp(std::max(sums->bound_w, 1e-300), &ew);
        while (((size_t)1 << P) < nx) P++;
        P = std::max(P + 1, 12);                                    // |addend| / quantum = 2^(62 - P) must stay below 2^51
        fix.quantum_x = ldexp(1.0, ex + P - 62); fix.magic_x = ldexp(1.5, 52 + ex + P - 62);
        fix.quantum_w = ldexp(1.0, ew + P - 62); fix.magic_w = ldexp(1.5, 52 + ew + P - 62);
        w.fsum.reserve(4 * (size_t)k);
        HIP_CHECK(hipMemsetAsync(w.fsum.p, 0, 4 * (size_t)k * sizeof(unsigned long long), s));
        ablocks = (int)std::min<size_t>(std::max<size_t>(ceil_div(nx, (size_t)4096), 1), 2048);
        static PerDeviceOnce attr5;
        if (attr5.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void *)k_km_accum<true, int>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * kKMeansMaxK));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_km_accum<false, int>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * kKMeansMaxK));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_km_accum<true, unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * kKMeansMaxK));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_km_accum<false, unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * kKMeansMaxK));
        }
    }
    // the order-free update after any of the assignment kernels below (a8: one byte per sample, else an int)
    auto update_sums = [&](const bool a8) {
        {
            KTIME("k_km_accum", s, ((weighted ? 16.0 : 12.0) + (a8 ? 1.0 : 4.0)) * nx);
            const size_t lds = (size_t)32 * k;
            if (a8) {
                if (weighted) hipLaunchKernelGGL((k_km_accum<true, unsigned char>), ablocks, 256, lds, s, ks, (const unsigned char *)w.assign.p, nx, k, fix, w.fsum.p);
                else hipLaunchKernelGGL((k_km_accum<false, unsigned char>), ablocks, 256, lds, s, ks, (const unsigned char *)w.assign.p, nx, k, fix, w.fsum.p);
            } else {
                if (weighted) hipLaunchKernelGGL((k_km_accum<true, int>), ablocks, 256, lds, s, ks, (const int *)w.assign.p, nx, k, fix, w.fsum.p);
                else hipLaunchKernelGGL((k_km_accum<false, int>), ablocks, 256, lds, s, ks, (const int *)w.assign.p, nx, k, fix, w.fsum.p);
            }
        }
        KTIME("k_km_update", s, 32.0 * k);
        if (weighted) hipLaunchKernelGGL(k_km_update_sums<true>, 1, 256, 0, s, w.fsum.p, k, (unsigned long long)nx, fix, w.cent.p, w.hassign.p, w.c4.p, w.mt.p);
        else hipLaunchKernelGGL(k_km_update_sums<false>, 1, 256, 0, s, w.fsum.p, k, (unsigned long long)nx, fix, w.cent.p, w.hassign.p, w.c4.p, w.mt.p);
    };
    int nbits = 0;
    while ((1 << nbits) < k) nbits++;
    const int chunk_len = chunk_len_for(nx);
    const int nchunks = (int)ceil_div(nx, (size_t)chunk_len);
    const int cblocks = (nchunks + 3) / 4;
    const bool bigk = k > kKMeansMaxK;                                // no LDS-resident tables: counters and cursors in memory
    const size_t lds_cnt = bigk ? 0 : (size_t)8 * k * sizeof(unsigned int);     // k float4 + 4 x k counters
    const size_t lds_sct = (size_t)5 * k * sizeof(unsigned int);
    const size_t lds_sct_pair = (size_t)9 * k * sizeof(unsigned int) + (size_t)4 * (k + 64) * sizeof(float4);
    static PerDeviceOnce attr;
    if (attr.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_assign_count<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kKMeansMaxK * 4));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter<true, int>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * kKMeansMaxK * 4));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter<false, int>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * kKMeansMaxK * 4));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter<true, unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * kKMeansMaxK * 4));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter<false, unsigned char>, hipFuncAttributeMaxDynamicSharedMemorySize, 5 * kKMeansMaxK * 4));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_update<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 4));
        HIP_CHECK(hipFuncSetAttribute((const void *)k_km_update<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 4));
    }
    if (!w.mt_seeded) { hipLaunchKernelGGL(k_km_mt_seed, 1, 1, 0, s, w.mt.p); w.mt_seeded = true; }   // the start of std::mt19937(1234), kept
    { KTIME("k_km_prep", s, 28.0 * k); hipLaunchKernelGGL(k_km_prep, (std::max(k, kKmFlagWords) + 255) / 256, 256, 0, s, w.cent.p, k, w.c4.p, w.flags.p); }
    // many samples: exact candidate pruning (the grid is rebuilt per iteration, ~0.1 ms, against ~1 ms of full scans per
    // 16 M samples); few samples (the default 512^2): the full scan is cheaper than building the grid
    const size_t lut_min = getenv("PAMD_KM_LUT_MIN") ? (size_t)atoll(getenv("PAMD_KM_LUT_MIN")) : ((size_t)1 << 21);
    const bool use_lut = nx >= lut_min && k >= 16 && k <= 256;
    const size_t g64_min = getenv("PAMD_KM_G64_MIN") ? (size_t)atoll(getenv("PAMD_KM_G64_MIN")) : ((size_t)1 << 23);
    const int G = nx >= g64_min ? 64 : 32;
    if (use_lut) {
        static PerDeviceOnce attr2;
        if (attr2.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void *)k_km_assign_lut, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kKMeansMaxK * 4));
        }
        w.lut.reserve((size_t)G * G * G * 16); w.bkeys.reserve(kKmSlots * 6); w.grid.reserve(64 + sizeof(double));
        w.clist.reserve((size_t)(G * G * G / 64) * (2 + kKmCoarseMax));
        hipLaunchKernelGGL(k_km_bounds_init, 1, 256, 0, s, w.bkeys.p);
        { KTIME("k_km_bounds", s, 12.0 * nx); hipLaunchKernelGGL(k_km_bounds, (int)std::min<size_t>(ceil_div(nx, 256), 2048), 256, 0, s, ks, nx, w.bkeys.p); }
        hipLaunchKernelGGL(k_km_bounds_fold, 1, 64, 0, s, w.bkeys.p, (KmGridDev *)w.grid.p);
    }
    // few samples: block-local sorts in the assignment kernel, one block per centroid collects its members (k_km_update_lists)
    const size_t direct_max = getenv("PAMD_KM_DIRECT_MAX") ? (size_t)atoll(getenv("PAMD_KM_DIRECT_MAX")) : ((size_t)1 << 19);
    // (a cluster of up to 16 384 samples: its list is summed block-parallel in pieces of 4096 by k_km_update_lists -- a scene with a
    // few flat regions refines in 2.3 ms instead of 3.2; beyond that the global sort + k_km_update's long-chain form wins: a colour
    // covering 10 % / 30 % of the image 4.5 / 6.9 ms against 8.2 / 18.7, measured with the threshold swept over 4096 .. 131 072)
    const size_t list_longest = getenv("PAMD_KM_LIST_LONGEST") ? (size_t)atoll(getenv("PAMD_KM_LIST_LONGEST")) : (size_t)16384;
    const bool use_direct = !sums && !use_lut && k <= 256 && nx <= direct_max && nx < ((size_t)1 << 32) && expect_longest < list_longest;
    if (use_direct) {
        // k_km_assign_sort keeps each sample's previous centroid as ONE BYTE in w.assign (k <= 256 on this path, above) and marks
        // flags[1 + old] / flags[1 + new] when it changes.  k_km_prep has just set every flag (nothing is known about the first
        // assignment), so the first iteration's `old` only has to be a valid byte: zeroed here rather than left as found.
        HIP_CHECK(hipMemsetAsync(w.assign.p, 0, nx, s));
        static PerDeviceOnce attr4;
        if (attr4.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void *)k_km_update_lists<true>, hipFuncAttributeMaxDynamicSharedMemorySize, kKmDirectCap * 20));
            HIP_CHECK(hipFuncSetAttribute((const void *)k_km_update_lists<false>, hipFuncAttributeMaxDynamicSharedMemorySize, kKmDirectCap * 20));
        }
    }
    static const bool mid_enabled = !(getenv("PAMD_KM_MID") && atoi(getenv("PAMD_KM_MID")) == 0);
    const bool use_mid = use_lut && G == 64 && mid_enabled && k % 8 == 0;    // four-candidate table in LDS
    if (use_mid) w.mid.reserve(32 * 32 * 32);
    // clusters of at least this many samples take the block-parallel exact chain (km_chain_coop)
    const unsigned long long long_min = getenv("PAMD_KM_LONG_MIN") ? (unsigned long long)atoll(getenv("PAMD_KM_LONG_MIN")) : 8192ULL;
    static const int km_quarters = (getenv("PAMD_KM_QUARTERS") && atoi(getenv("PAMD_KM_QUARTERS")) == 0) ? 0 : 1;   // A/B of the quarter-wise retry
    for (int it = 0; it < niter; it++) {
        if (use_direct) {
            unsigned short *offs = (unsigned short *)w.table.p;              // 257 x blocks x 2 bytes
            const int sblocks = (int)ceil_div(nx, (size_t)kKmSortBlock);
            {
                KTIME("k_km_assign", s, (weighted ? 32.0 : 28.0) * nx);
                // 1 (default): sixteen wavefronts, one sample per lane, min3 scan; 2: eight wavefronts with two samples per lane (measured
                // slower: 23 000 against 15 100 cycles for the scan -- a wavefront waits for its scalar loads, and with half the
                // wavefronts per SIMD fewer of those waits overlap); 3: the round-3 scan (km_assign_one_pk: 23 500 cycles)
                static const int per_lane = getenv("PAMD_KM_SCAN") ? atoi(getenv("PAMD_KM_SCAN")) : 1;
                if (per_lane == 2) hipLaunchKernelGGL(k_km_assign_sort<2>, sblocks, 512, 0, s, ks, nx, w.c4.p, k, weighted, w.sorted.p, offs, (unsigned char *)w.assign.p, w.flags.p, true);
                else hipLaunchKernelGGL(k_km_assign_sort<1>, sblocks, 1024, 0, s, ks, nx, w.c4.p, k, weighted, w.sorted.p, offs, (unsigned char *)w.assign.p, w.flags.p, per_lane == 1);
            }
            {
                KTIME("k_km_update", s, 16.0 * nx);
                if (weighted) hipLaunchKernelGGL(k_km_update_lists<true>, k, 256, kKmDirectCap * 20, s, (const float4 *)w.sorted.p, (const unsigned short *)offs,
                                                 (unsigned long long)nx, k, w.cent.p, w.hassign.p, w.c4.p, w.ticket.p, w.mt.p, w.flags.p);
                else hipLaunchKernelGGL(k_km_update_lists<false>, k, 256, kKmDirectCap * 20, s, (const float4 *)w.sorted.p, (const unsigned short *)offs,
                                        (unsigned long long)nx, k, w.cent.p, w.hassign.p, w.c4.p, w.ticket.p, w.mt.p, w.flags.p);
            }
            continue;
        }
        if (use_lut) {
            {
                KTIME("k_km_lut_build", s, 16.0 * G * G * G);
                const int ncoarse = G * G * G / 64;
                double *cn2 = (double *)(w.grid.p + 64);                  // a scalar next to the bounding box
                hipLaunchKernelGGL(k_km_lut_coarse, ncoarse, 64, 0, s, w.c4.p, k, (const KmGridDev *)w.grid.p, G, w.clist.p, cn2);
                hipLaunchKernelGGL(k_km_lut_build, ncoarse, 64, 0, s, w.c4.p, k, (const KmGridDev *)w.grid.p, G, (const unsigned char *)w.clist.p,
                                   (const double *)cn2, w.lut.p, use_mid ? w.mid.p : (unsigned int *)nullptr);
            }
            if (use_mid) {
                const int nmid = 32 * 32 * 32;
                const size_t lds_mid = ((size_t)nmid + 4 * 256 + 16 * 256) * 4 + (size_t)16 * kKmQueue * 16;
                static PerDeviceOnce attr3;
                if (attr3.first()) {
                    HIP_CHECK(hipFuncSetAttribute((const void *)k_km_assign_mid, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_mid));
                }
                const int mblocks = std::min(num_cus(), (nchunks + 15) / 16);
                KTIME("k_km_assign", s, 16.0 * nx);
                hipLaunchKernelGGL(k_km_assign_mid, mblocks, 1024, lds_mid, s, ks, nx, w.c4.p, k, chunk_len, nchunks, (unsigned char *)w.assign.p, w.table.p,
                                   (const KmGridDev *)w.grid.p, (const unsigned int *)w.mid.p, (const unsigned char *)w.lut.p);
            } else {
            KTIME("k_km_assign", s, 16.0 * nx);
            hipLaunchKernelGGL(k_km_assign_lut, cblocks, 256, lds_cnt, s, ks, nx, w.c4.p, k, nbits, chunk_len, nchunks, w.assign.p, w.table.p,
                               (const KmGridDev *)w.grid.p, G, w.lut.p);
            }
        } else {
            KTIME("k_km_assign", s, 16.0 * nx);
            if (bigk) {
                HIP_CHECK(hipMemsetAsync(w.table.p, 0, (size_t)k * nchunks * sizeof(unsigned int), s));
                hipLaunchKernelGGL(k_km_assign_count<true>, cblocks, 256, 0, s, ks, nx, w.c4.p, k, nbits, chunk_len, nchunks, w.assign.p, w.table.p);
            } else hipLaunchKernelGGL(k_km_assign_count<false>, cblocks, 256, lds_cnt, s, ks, nx, w.c4.p, k, nbits, chunk_len, nchunks, w.assign.p, w.table.p);
        }
        if (sums) { update_sums(use_lut && use_mid); continue; }
        if (bigk) {
            // palettes beyond kKMeansMaxK entries (the reference has no limit, refine.c:77-89): the same stable counting sort and the
            // same chains, with the sort's per-centroid cursors in memory instead of LDS
            { KTIME("k_km_rowscan", s, 8.0 * k * nchunks); hipLaunchKernelGGL(k_km_rowscan, k, 256, 0, s, w.table.p, nchunks, w.rowtot.p); }
            hipLaunchKernelGGL(k_km_rowbase, 1, 256, 0, s, (const unsigned int *)w.rowtot.p, k, w.rowbase.p);
            {
                KTIME("k_km_scatter", s, (weighted ? 36.0 : 32.0) * nx);
                if (weighted) hipLaunchKernelGGL(k_km_scatter_big<true>, cblocks, 256, 0, s, ks, (const int *)w.assign.p, nx, nbits, chunk_len, nchunks, w.table.p, (const unsigned int *)w.rowbase.p, w.sorted.p);
                else hipLaunchKernelGGL(k_km_scatter_big<false>, cblocks, 256, 0, s, ks, (const int *)w.assign.p, nx, nbits, chunk_len, nchunks, w.table.p, (const unsigned int *)w.rowbase.p, w.sorted.p);
            }
            {
                KTIME("k_km_update", s, 16.0 * nx);
                const bool coop_on = nx >= long_min;
                const size_t lds_up = coop_on ? (size_t)4 * 4096 * sizeof(float) : 0;
                const unsigned long long lm = coop_on ? long_min : ~0ULL;
                static PerDeviceOnce attr6;
                if (attr6.first()) {
                    HIP_CHECK(hipFuncSetAttribute((const void *)k_km_update<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 4));
                    HIP_CHECK(hipFuncSetAttribute((const void *)k_km_update<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 4));
                }
                if (weighted) hipLaunchKernelGGL((k_km_update<true, true>), k, 256, lds_up, s, w.sorted.p, w.rowtot.p, k, (unsigned long long)nx, w.cent.p, w.hassign.p, w.c4.p, w.ticket.p, w.mt.p, lm, w.hs.p, km_quarters);
                else hipLaunchKernelGGL((k_km_update<false, true>), k, 256, lds_up, s, w.sorted.p, w.rowtot.p, k, (unsigned long long)nx, w.cent.p, w.hassign.p, w.c4.p, w.ticket.p, w.mt.p, lm, w.hs.p, km_quarters);
            }
            continue;
        }
        { KTIME("k_km_rowscan", s, 8.0 * k * nchunks); hipLaunchKernelGGL(k_km_rowscan, k, 256, 0, s, w.table.p, nchunks, w.rowtot.p); }
        {
            KTIME("k_km_scatter", s, (weighted ? 36.0 : 32.0) * nx - (use_mid ? 3.0 : 0.0) * nx);
            const unsigned char *a8 = (const unsigned char *)w.assign.p;               // k_km_assign_mid leaves one byte per sample
            static const bool pair_on = !(getenv("PAMD_KM_PAIR") && atoi(getenv("PAMD_KM_PAIR")) == 0);
            static const bool lds_sort_on = !(getenv("PAMD_KM_LDS_SORT") && atoi(getenv("PAMD_KM_LDS_SORT")) == 0);
            if (use_mid && lds_sort_on && chunk_len >= 4096 && k <= 256) {
                constexpr size_t lds_ls = (size_t)(512 + 1024 + 4096 / 4 + 4 * 4096) * sizeof(unsigned int);
                static PerDeviceOnce attr5;
                if (attr5.first()) {
                    HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter_lds<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ls));
                    HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter_lds<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ls));
                }
                if (weighted) hipLaunchKernelGGL(k_km_scatter_lds<true>, nchunks, 256, lds_ls, s, ks, a8, nx, k, chunk_len, nchunks, w.table.p, w.rowtot.p, w.sorted.p);
                else hipLaunchKernelGGL(k_km_scatter_lds<false>, nchunks, 256, lds_ls, s, ks, a8, nx, k, chunk_len, nchunks, w.table.p, w.rowtot.p, w.sorted.p);
            } else if (use_mid && pair_on && chunk_len >= 2 * k) {
                static PerDeviceOnce attr3;
                if (attr3.first()) {
                    HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter<true, unsigned char, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 256 * 4 + 4 * (256 + 64) * 16));
                    HIP_CHECK(hipFuncSetAttribute((const void *)k_km_scatter<false, unsigned char, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 9 * 256 * 4 + 4 * (256 + 64) * 16));
                }
                if (weighted) hipLaunchKernelGGL((k_km_scatter<true, unsigned char, true>), cblocks, 256, lds_sct_pair, s, ks, a8, nx, k, nbits, chunk_len, nchunks, w.table.p, w.rowtot.p, w.sorted.p);
                else hipLaunchKernelGGL((k_km_scatter<false, unsigned char, true>), cblocks, 256, lds_sct_pair, s, ks, a8, nx, k, nbits, chunk_len, nchunks, w.table.p, w.rowtot.p, w.sorted.p);
            } else if (use_mid) {
                if (weighted) hipLaunchKernelGGL((k_km_scatter<true, unsigned char>), cblocks, 256, lds_sct, s, ks, a8, nx, k, nbits, chunk_len, nchunks, w.table.p, w.rowtot.p, w.sorted.p);
                else hipLaunchKernelGGL((k_km_scatter<false, unsigned char>), cblocks, 256, lds_sct, s, ks, a8, nx, k, nbits, chunk_len, nchunks, w.table.p, w.rowtot.p, w.sorted.p);
            } else {
                if (weighted) hipLaunchKernelGGL((k_km_scatter<true, int>), cblocks, 256, lds_sct, s, ks, (const int *)w.assign.p, nx, k, nbits, chunk_len, nchunks, w.table.p, w.rowtot.p, w.sorted.p);
                else hipLaunchKernelGGL((k_km_scatter<false, int>), cblocks, 256, lds_sct, s, ks, (const int *)w.assign.p, nx, k, nbits, chunk_len, nchunks, w.table.p, w.rowtot.p, w.sorted.p);
            }
        }
        {
            KTIME("k_km_update", s, 16.0 * nx);
            // the shared 64 KB buffer of the block-parallel chain is only asked for when clusters of that length can exist
            const bool coop_on = nx >= long_min;
            const size_t lds_up = coop_on ? (size_t)4 * 4096 * sizeof(float) : 0;
            const unsigned long long lm = coop_on ? long_min : ~0ULL;
            if (weighted) hipLaunchKernelGGL((k_km_update<true, false>), k, 256, lds_up, s, w.sorted.p, w.rowtot.p, k, (unsigned long long)nx, w.cent.p, w.hassign.p, w.c4.p, w.ticket.p, w.mt.p, lm, (float *)nullptr, km_quarters);
            else hipLaunchKernelGGL((k_km_update<false, false>), k, 256, lds_up, s, w.sorted.p, w.rowtot.p, k, (unsigned long long)nx, w.cent.p, w.hassign.p, w.c4.p, w.ticket.p, w.mt.p, lm, (float *)nullptr, km_quarters);
        }
    }
    HIP_CHECK(hipGetLastError());
}

}  // namespace pamd

#ifdef PAMD_KM_TRACE
extern "C" int patolette_amd_debug_km_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pamd::g_km_trace), sizeof(pamd::g_km_trace)) == hipSuccess ? 0 : -1;
}
#endif
