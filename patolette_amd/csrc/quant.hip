// quant.hip -- gfx950 kernels of the weighted-PCA cluster-split loop.
//
// Replaces the O(N) inner loops of lib/src/quantize/{global,local,cluster,sort,cells}.c and
// lib/src/math/pca.c.  The reference gathers each cluster's pixels through an index array and
// sweeps them ~12 times per split; here the pixels of every tree node are kept PHYSICALLY
// contiguous (planar x|y|z|w, two ping-pong buffers) and a split is four streaming passes:
//
//   k_minmax   projection extrema on the node's principal axis          24 B/px read
//   k_hist     512-bucket moments of the projection (LDS histogram)     24(+8) B/px read, 2 B/px write
//   k_cut      prefix + objective + arg-max per node (1 block/node)     histogram only
//   k_count / k_scan / k_scatter   stable partition into the children   2 + 24(+8)+2 B/px read, 24(+8) B/px write
//   k_cov      children's centred covariance + distortion               24(+8) B/px read
//
// All of them are HBM-bound (inner dimension 3: no MFMA).  Floating-point sums use the
// order-independent binned accumulation of devutil.h, so LDS / global f64 atomics are exact
// and the results are bit-reproducible for any launch geometry.
#include "quant.h"

#include <atomic>

#include <type_traits>

namespace pamd {

// --------------------------------------------------------------------------------------------
// small block-level primitives (512- or 256-thread blocks)
// --------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T block_scan_incl(T v, T *smem) {          // smem: >= 16 entries
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        T t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    __syncthreads();
    if (lane == 63) smem[wid] = v;
    __syncthreads();
    if (wid == 0) {
        T s = lane < nw ? smem[lane] : T(0);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {
            T t = __shfl_up(s, o, 64);
            if (lane >= o) s += t;
        }
        if (lane < nw) smem[lane] = s;
    }
    __syncthreads();
    if (wid > 0) v += smem[wid - 1];
    return v;
}
// the same for N values per thread at once: three barriers in all instead of three per value.  smem: >= N * 16 entries
template <typename T, int N>
__device__ __forceinline__ void block_scan_incl_vec(T (&v)[N], T *smem) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
        for (int i = 0; i < N; i++) { T t = __shfl_up(v[i], o, 64); if (lane >= o) v[i] += t; }
    }
    __syncthreads();
    if (lane == 63) {
#pragma unroll
        for (int i = 0; i < N; i++) smem[i * 16 + wid] = v[i];
    }
    __syncthreads();
    if (wid == 0 && lane < 16 * N) {                       // lanes [16 i, 16 i + 16) scan value i's wave totals
        const int i = lane >> 4, e = lane & 15;
        T s = (i < N && e < nw) ? smem[i * 16 + e] : T(0);
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { T t = __shfl_up(s, o, 16); if (e >= o) s += t; }
        if (i < N && e < nw) smem[i * 16 + e] = s;
    }
    __syncthreads();
    if (wid > 0) {
#pragma unroll
        for (int i = 0; i < N; i++) v[i] += smem[i * 16 + wid - 1];
    }
}

// --------------------------------------------------------------------------------------------
// root mean: sum of each plane (GQ's PCA is UNWEIGHTED, global.c:407 -> pca.c:151-168)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sum3(const double *__restrict__ c, size_t N, BinK k, double *out6) {
    __shared__ double sm[6 * 4];
    double a[6] = {0, 0, 0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) {
        bin_add(c[i], k, a[0], a[1]);
        bin_add(c[N + i], k, a[2], a[3]);
        bin_add(c[2 * N + i], k, a[4], a[5]);
    }
    block_sum<6>(a, sm);
    if (threadIdx.x == 0)                                   // same-address atomics serialise in L2: spread over slots
        for (int q = 0; q < 6; q++) unsafeAtomicAdd(&out6[(blockIdx.x & (kSum3Slots - 1)) * 6 + q], a[q]);
}

// --------------------------------------------------------------------------------------------
// projection on the node axis: sort.c:43-59 (dgemv, min, max)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ double project(double x, double y, double z, const double a0, const double a1, const double a2) {
    return (x * a0 + y * a1) + z * a2;
}

__device__ __forceinline__ int tile_node(const int *t0, int nr, int t) {   // largest r with t0[r] <= t
    int lo = 0, hi = nr;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t0[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}
template <bool LISTS>
__global__ __launch_bounds__(256) void k_minmax(QuantBuffers qb, const Tile *__restrict__ tiles, NodeDev *nodes, const int from_end,
                                                const RoundDyn *__restrict__ dyn, const RoundLists L) {
    // tiles are taken from the END: this sweep follows the partition kernel, whose most recently written lines are the last
    // tiles' (what is still in flight towards HBM is read last; measured on a replica: 80 -> 73 us behind a copy of 403 MB)
    const unsigned nt = dyn ? (unsigned)dyn->ntA : gridDim.x;   // dyn: the launch is an upper bound, the control kernel knows the extent
    if (blockIdx.x >= nt) return;
    const unsigned tix = from_end ? nt - 1u - blockIdx.x : blockIdx.x;
    Tile t;
    if constexpr (LISTS) {
        // a device-driven round: this sweep is the round's first kernel and sets the round up as it goes -- block b makes tile b of
        // this tiling (for the sweeps behind it too), its share of the partition's tiles and of the cleared bucket tables.  (The
        // round's nodes need no reset: a node's outputs are cleared when k_cut / k_gq_control / k_put_nodes make it, and nothing
        // touches them until its own round.)
        const int nr = dyn->nr, ntP = dyn->ntP;
        {
            const int r = tile_node(L.tA0, nr, (int)tix);
            const int id = L.round_ids[r];
            const unsigned long long o = (unsigned long long)((int)tix - L.tA0[r]) * kTileA, n = nodes[id].n;
            t = Tile{nodes[id].begin + o, (unsigned)(n - o < (unsigned long long)kTileA ? n - o : kTileA), (unsigned)id};
            if (threadIdx.x == 0) L.tilesA[tix] = t;
        }
        for (int tp = (int)blockIdx.x + (int)threadIdx.x * (int)nt; tp < ntP; tp += (int)nt * 256) {
            const int r = tile_node(L.tP0, nr, tp);
            const int id = L.round_ids[r];
            const unsigned long long o = (unsigned long long)(tp - L.tP0[r]) * kTileP, n = nodes[id].n;
            L.tilesP[tp] = Tile{nodes[id].begin + o, (unsigned)(n - o < (unsigned long long)kTileP ? n - o : kTileP), (unsigned)id};
        }
        const size_t nh = L.lqs * (size_t)nr, nb = (size_t)nr * kBuckets;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += (size_t)nt * 256) L.hist[i] = 0.0;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nb; i += (size_t)nt * 256) { L.hsize[i] = 0ULL; L.hcount[i] = 0u; }
    } else {
        t = tiles[tix];
    }
    NodeDev &nd = nodes[t.node];
    const double a0 = nd.axis[0], a1 = nd.axis[1], a2 = nd.axis[2];
    const double *px = qb.buf[nd.buf], *py = px + qb.N, *pz = py + qb.N;
    double mn = INFINITY, mx = -INFINITY;
    unsigned i = threadIdx.x;
    for (; i + 3 * 256 < t.count; i += 4 * 256) {           // twelve loads in flight per lane
        const size_t p = t.start + i;
        double x[4], y[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { x[u] = px[p + u * 256]; y[u] = py[p + u * 256]; z[u] = pz[p + u * 256]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            double d = project(x[u], y[u], z[u], a0, a1, a2);
            mn = fmin(mn, d); mx = fmax(mx, d);
        }
    }
    for (; i < t.count; i += 256) {
        size_t p = t.start + i;
        double d = project(px[p], py[p], pz[p], a0, a1, a2);
        mn = fmin(mn, d); mx = fmax(mx, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_down(mn, o, 64)); mx = fmax(mx, __shfl_down(mx, o, 64)); }
    __shared__ double smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { mn = fmin(mn, smn[w]); mx = fmax(mx, smx[w]); }
        if (mn <= mx) {
            atomicMin(&nd.minkey[blockIdx.x & (kSlots - 1)], f64_key(mn));
            atomicMax(&nd.maxkey[blockIdx.x & (kSlots - 1)], f64_key(mx));
        }
    }
}

// --------------------------------------------------------------------------------------------
// bucket moments: sort.c:61-87 (bucket id) + local.c:118-134 (LQ) / cells.c:82-112 (GQ)
// --------------------------------------------------------------------------------------------
constexpr int kHistCombineMin = 12;    // lanes of a wavefront sharing a bucket from which their addends are summed before the LDS atomics
template <bool W, bool GQ>
__global__ __launch_bounds__(512) void k_hist(QuantBuffers qb, const Tile *__restrict__ tiles, int ntiles, NodeDev *nodes,
                                              double *hist, unsigned long long *hsize, unsigned int *hcount, const int from_end,
                                              const RoundDyn *__restrict__ dyn) {
    constexpr int NQ = GQ ? (W ? 14 : 10) : (W ? 4 : 3);
    int nblk = (int)gridDim.x;
    if (dyn) {                                             // the launch is an upper bound: as many blocks work as a launch sized by the host would have had
        ntiles = dyn->ntA;
        nblk = min(ntiles, nblk);
        if ((int)blockIdx.x >= nblk) return;
    }
    constexpr int NQS = GQ ? kNQ_GQ : kNQ_LQ;            // slot stride in quantities
    extern __shared__ double lds[];
    double *h = lds;                                       // [NQ][2][512]
    unsigned long long *siz = (unsigned long long *)(h + NQ * 2 * kBuckets);     // 64-bit: weights may be large (local.c:133 sums in size_t)
    unsigned int *cnt = (unsigned int *)(siz + kBuckets);
    for (int i = threadIdx.x; i < NQ * 2 * kBuckets; i += blockDim.x) h[i] = 0.0;
    for (int i = threadIdx.x; i < kBuckets; i += blockDim.x) { cnt[i] = 0u; siz[i] = 0ULL; }
    __syncthreads();

    // a block walks consecutive tiles (the tiles of one node are consecutive) and flushes its LDS histogram to HBM
    // only when the node changes: far fewer global f64 atomics than one flush per tile
    // from_end: the blocks take their runs of tiles from the end of the list -- each sweep of a split round starts where the
    // previous one stopped, on the lines that are still in the caches (k_minmax)
    const int per = (ntiles + nblk - 1) / nblk;
    const int bid = (from_end & 1) ? nblk - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int tfirst = bid * per, tlast = min(ntiles, tfirst + per);
    for (int ti = tfirst; ti < tlast; ti++) {
        const Tile t = tiles[ti];
        NodeDev &nd = nodes[t.node];
        const double a0 = nd.axis[0], a1 = nd.axis[1], a2 = nd.axis[2];
        double mn, mx;
        node_minmax(nd, mn, mx);
        const bool degenerate = (mx - mn < kDelta);
        const double sc = 1 / (mx - mn);
        const BinK klin = nd.klin, kquad = nd.kquad;
        const double *px = qb.buf[nd.buf], *py = px + qb.N, *pz = py + qb.N, *pw = pz + qb.N;
        const unsigned long long gslot0 = nd.gslot0;
        if (threadIdx.x == 0) nd.degenerate = degenerate ? 1 : 0;

        // two pixels per trip: both sets of loads are issued before either is consumed
        auto one = [&](const size_t p, const double x, const double y, const double z, const double w) {
            unsigned b;
            if (degenerate) {
                b = (unsigned)((p - nd.begin + gslot0) % kBuckets);   // the node-wide slot (sort.c:61-79)
            } else {
                double ratio = (project(x, y, z, a0, a1, a2) - mn) * sc;
                unsigned long long bq = (unsigned long long)((double)kBuckets * ratio);
                b = bq < (unsigned long long)(kBuckets - 1) ? (unsigned)bq : (unsigned)(kBuckets - 1);
            }
            qb.bkt[p] = (unsigned short)b;
            // the pixel's addends, both binned parts of every quantity
            double pv[2 * NQ];
#define HPART(q, val, K) bin_split((val), (K), pv[2 * (q)], pv[2 * (q) + 1])
            if constexpr (!GQ) {
                HPART(0, x * w, klin); HPART(1, y * w, klin); HPART(2, z * w, klin);
                if constexpr (W) HPART(3, w, klin);
            } else {
                HPART(0, x, klin); HPART(1, y, klin); HPART(2, z, klin);
                HPART(3, (x * x + y * y) + z * z, kquad);
                HPART(4, x * x, kquad); HPART(5, x * y, kquad); HPART(6, y * y, kquad);
                HPART(7, x * z, kquad); HPART(8, y * z, kquad); HPART(9, z * z, kquad);
                if constexpr (W) { HPART(10, x * w, klin); HPART(11, y * w, klin); HPART(12, z * w, klin); HPART(13, w, klin); }
            }
#undef HPART
            unsigned long long wi = 0ULL;
            if constexpr (W && !GQ) wi = (unsigned long long)w;                  // size_t += double truncates (local.c:133)
            // Neighbouring pixels of a flat or smooth region fall into the SAME bucket: 64 lanes adding to one LDS address
            // serialise (a smooth 4096^2 image took 3.2x, a posterised one 5.4x as long as noise).  When the wavefront is complete
            // and at least a third of it shares the first lane's bucket, that group's addends are summed over the wavefront first
            // (exact: they lie on the grids) and added once; the other lanes add their own.
            bool direct = true;
            if (__ballot(true) == ~0ULL) {
                // up to three groups: the bucket of the first lane still holding its addends, if enough lanes share it
                unsigned long long left = ~0ULL;
#pragma unroll 1
                for (int g = 0; g < 3 && left != 0ULL; g++) {
                    const int first = __builtin_ctzll(left);                       // wave-uniform
                    const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)b, first);
                    const bool mine = direct && b == b0;
                    const unsigned long long m = __ballot(mine);
                    if (__popcll(m) < kHistCombineMin) break;
#pragma unroll
                    for (int k2 = 0; k2 < 2 * NQ; k2++) {
                        const double sum = wave_sum_dpp63(mine ? pv[k2] : 0.0);
                        if ((threadIdx.x & 63) == 63) unsafeAtomicAdd(&h[k2 * kBuckets + b0], sum);
                    }
                    if constexpr (W && !GQ) {
                        unsigned long long sw2 = mine ? wi : 0ULL;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) sw2 += __shfl_down(sw2, o, 64);
                        if ((threadIdx.x & 63) == 0) atomicAdd(&siz[b0], sw2);
                    }
                    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt[b0], (unsigned)__popcll(m));
                    if (mine) direct = false;
                    left &= ~m;
                }
            }
#ifdef PAMD_KM_TRACE
            if (from_end & 2) { if (pv[0] == 1.2345e300) h[b] = pv[1] + pv[2] + pv[3] + pv[4] + pv[5]; direct = false; }   // diagnostic: the kernel without its LDS atomics
#endif
            if (direct) {
                atomicAdd(&cnt[b], 1u);
#pragma unroll
                for (int k2 = 0; k2 < 2 * NQ; k2++) unsafeAtomicAdd(&h[k2 * kBuckets + b], pv[k2]);
                if constexpr (W && !GQ) atomicAdd(&siz[b], wi);
            }
        };
        unsigned i = threadIdx.x;
        for (; i + 512 < t.count; i += 1024) {
            const size_t p0 = t.start + i, p1 = p0 + 512;
            const double x0 = px[p0], y0 = py[p0], z0 = pz[p0], x1 = px[p1], y1 = py[p1], z1 = pz[p1];
            double w0 = 1.0, w1 = 1.0;
            if constexpr (W) { w0 = pw[p0]; w1 = pw[p1]; }
            one(p0, x0, y0, z0, w0);
            one(p1, x1, y1, z1, w1);
        }
        for (; i < t.count; i += 512) {
            const size_t p = t.start + i;
            double w = 1.0;
            if constexpr (W) w = pw[p];
            one(p, px[p], py[p], pz[p], w);
        }
        const bool flush = (ti + 1 == tlast) || tiles[ti + 1].node != t.node;           // block-uniform
        if (!flush) continue;
        __syncthreads();
        const size_t slot = (size_t)nd.slot;
        double *gh = hist + slot * (size_t)(NQS * 2 * kBuckets);
        for (int i = threadIdx.x; i < NQ * 2 * kBuckets; i += blockDim.x) {
            double v = h[i];
            if (v != 0.0) { unsafeAtomicAdd(&gh[i], v); h[i] = 0.0; }
        }
        for (int b = threadIdx.x; b < kBuckets; b += blockDim.x) {
            unsigned c = cnt[b];
            if (c) { atomicAdd(&hcount[slot * kBuckets + b], c); cnt[b] = 0u; }
            if constexpr (W && !GQ) { const unsigned long long s = siz[b]; if (s) { atomicAdd(&hsize[slot * kBuckets + b], s); siz[b] = 0ULL; } }
        }
        __syncthreads();
    }
}

// ---- the same histogram with block-local sums in 64-bit fixed point: used for the global quantiser's sweep over images of
// 2^18 pixels and more (launch_hist).  Smaller nodes keep the two-part f64 sums above, which hold 2B >= 66 bits below the bound
// there: on flat or posterised content the cell distortion w2 - |w1|^2 / w0 (cells.c:141-182) is a difference of equal numbers
// and must come out below 1e-16 (global.c:115), which 45 bits cannot deliver.  The local quantiser's histogram (3-4 quantities)
// gained 5 % from it -- it is not bound by its seven LDS atomics -- and stays as it is.
// Block-local accumulation in 64-bit FIXED POINT: one LDS atomic per quantity and pixel instead of two f64 ones (seven LDS
// atomics per pixel made k_hist_lq stall half of its time, twenty-one k_hist_gq).  An addend v (|v| < 2^E, the node's bound)
// is rounded to the nearest multiple of the quantum 2^(E-F), F = min(45, 2B) -- F <= 2B keeps it on the node's fine grid
// (devutil.h), so the rounding is a property of the pixel alone, whatever block sees it -- and added as an integer: exact
// in any order.  A block flushes at least every 2^17 pixels, so |sum| < 2^(F+17) <= 2^62.  The flush turns the integer into
// the two grid parts of the global table (hi a multiple of 2^(E-B), lo the rest; both exact in f64) and adds them with the
// f64 atomics as before: the table's two parts depend on how the pixels were grouped, their SUM -- all any reader forms --
// does not.  45 bits below the bound (54 before at 16 M pixels): the sum of a 4096^2 image's addends is still ~10x closer
// to the exact sum than the reference's sequential f64 sum.
struct FixK { double magic; int s; double ghi, glo; };      // magic = 1.5 * 2^(52+E-F); s = F - B; ghi = 2^(E-B), glo = 2^(E-F)
__device__ __forceinline__ FixK make_fixk(const BinK k) {
    const int e0 = (int)((__double_as_longlong(k.M0) >> 52) & 0x7ff) - 1023 - 52;      // E - B
    const int e1 = (int)((__double_as_longlong(k.M1) >> 52) & 0x7ff) - 1023 - 52;      // E - 2B
    const int B = e0 - e1, E = e0 + B;
    const int F = 2 * B < 45 ? 2 * B : 45;
    FixK f;
    f.magic = ldexp(1.5, 52 + E - F); f.s = F - B; f.ghi = ldexp(1.0, e0); f.glo = ldexp(1.0, E - F);
    return f;
}
__device__ __forceinline__ long long fix_quant(const double v, const double magic) {     // v / quantum, rounded to nearest even
    return __double_as_longlong(v + magic) - __double_as_longlong(magic);                 // same binade: the mantissas subtract
}
template <bool W, bool GQ>
__global__ __launch_bounds__(512) void k_hist_fix(QuantBuffers qb, const Tile *__restrict__ tiles, int ntiles, NodeDev *nodes,
                                              double *hist, unsigned long long *hsize, unsigned int *hcount, const int from_end) {
    constexpr int NQ = GQ ? (W ? 14 : 10) : (W ? 4 : 3);
    constexpr int NQS = GQ ? kNQ_GQ : kNQ_LQ;            // slot stride in quantities
    extern __shared__ unsigned long long lds64[];
    unsigned long long *h = lds64;                         // [NQ][512] fixed-point sums (two's complement)
    unsigned long long *siz = h + NQ * kBuckets;           // 64-bit: weights may be large (local.c:133 sums in size_t)
    unsigned int *cnt = (unsigned int *)(siz + kBuckets);
    for (int i = threadIdx.x; i < NQ * kBuckets; i += blockDim.x) h[i] = 0ULL;
    for (int i = threadIdx.x; i < kBuckets; i += blockDim.x) { cnt[i] = 0u; siz[i] = 0ULL; }
    __syncthreads();

    // a block walks consecutive tiles (the tiles of one node are consecutive) and flushes its LDS histogram to HBM
    // only when the node changes: far fewer global f64 atomics than one flush per tile
    // from_end: the blocks take their runs of tiles from the end of the list -- each sweep of a split round starts where the
    // previous one stopped, on the lines that are still in the caches (k_minmax)
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int bid = from_end ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int tfirst = bid * per, tlast = min(ntiles, tfirst + per);
    int since_flush = 0;                                    // tiles accumulated in LDS (kTileA = 2^13 pixels each)
    for (int ti = tfirst; ti < tlast; ti++) {
        const Tile t = tiles[ti];
        NodeDev &nd = nodes[t.node];
        const double a0 = nd.axis[0], a1 = nd.axis[1], a2 = nd.axis[2];
        double mn, mx;
        node_minmax(nd, mn, mx);
        const bool degenerate = (mx - mn < kDelta);
        const double sc = 1 / (mx - mn);
        const FixK flin = make_fixk(nd.klin), fquad = make_fixk(nd.kquad);
        const double *px = qb.buf[nd.buf], *py = px + qb.N, *pz = py + qb.N, *pw = pz + qb.N;
        const unsigned long long gslot0 = nd.gslot0;
        if (threadIdx.x == 0) nd.degenerate = degenerate ? 1 : 0;

        // two pixels per trip: both sets of loads are issued before either is consumed
        auto one = [&](const size_t p, const double x, const double y, const double z, const double w) {
            unsigned b;
            if (degenerate) {
                b = (unsigned)((p - nd.begin + gslot0) % kBuckets);   // the node-wide slot (sort.c:61-79)
            } else {
                double ratio = (project(x, y, z, a0, a1, a2) - mn) * sc;
                unsigned long long bq = (unsigned long long)((double)kBuckets * ratio);
                b = bq < (unsigned long long)(kBuckets - 1) ? (unsigned)bq : (unsigned)(kBuckets - 1);
            }
            qb.bkt[p] = (unsigned short)b;
            // the pixel's addends: value and the magic constant of its grid
            double pv[NQ], pm[NQ];
#define HPART(q, val, K) pv[q] = (val); pm[q] = (K).magic
            if constexpr (!GQ) {
                HPART(0, x * w, flin); HPART(1, y * w, flin); HPART(2, z * w, flin);
                if constexpr (W) { HPART(3, w, flin); }
            } else {
                HPART(0, x, flin); HPART(1, y, flin); HPART(2, z, flin);
                HPART(3, (x * x + y * y) + z * z, fquad);
                HPART(4, x * x, fquad); HPART(5, x * y, fquad); HPART(6, y * y, fquad);
                HPART(7, x * z, fquad); HPART(8, y * z, fquad); HPART(9, z * z, fquad);
                if constexpr (W) { HPART(10, x * w, flin); HPART(11, y * w, flin); HPART(12, z * w, flin); HPART(13, w, flin); }
            }
#undef HPART
            unsigned long long wi = 0ULL;
            if constexpr (W && !GQ) wi = (unsigned long long)w;                  // size_t += double truncates (local.c:133)
            // Neighbouring pixels of a flat or smooth region fall into the SAME bucket: 64 lanes adding to one LDS address
            // serialise (a smooth 4096^2 image took 3.2x, a posterised one 5.4x as long as noise).  When the wavefront is complete
            // and at least a third of it shares the first lane's bucket, that group's addends are summed over the wavefront first
            // (exact: the quantised values are multiples of the quantum, 64 of them stay below 2^51 quanta) and added once; the
            // other lanes add their own.
            bool direct = true;
            if (__ballot(true) == ~0ULL) {
                // up to three groups: the bucket of the first lane still holding its addends, if enough lanes share it
                unsigned long long left = ~0ULL;
#pragma unroll 1
                for (int g = 0; g < 3 && left != 0ULL; g++) {
                    const int first = __builtin_ctzll(left);                       // wave-uniform
                    const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)b, first);
                    const bool mine = direct && b == b0;
                    const unsigned long long m = __ballot(mine);
                    if (__popcll(m) < kHistCombineMin) break;
#pragma unroll
                    for (int k2 = 0; k2 < NQ; k2++) {
                        const double vq = (pv[k2] + pm[k2]) - pm[k2];              // the addend on its quantum
                        const double sum = wave_sum_dpp63(mine ? vq : 0.0);
                        if ((threadIdx.x & 63) == 63) atomicAdd(&h[k2 * kBuckets + b0], (unsigned long long)fix_quant(sum, pm[k2]));
                    }
                    if constexpr (W && !GQ) {
                        unsigned long long sw2 = mine ? wi : 0ULL;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) sw2 += __shfl_down(sw2, o, 64);
                        if ((threadIdx.x & 63) == 0) atomicAdd(&siz[b0], sw2);
                    }
                    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt[b0], (unsigned)__popcll(m));
                    if (mine) direct = false;
                    left &= ~m;
                }
            }
            if (direct) {
                atomicAdd(&cnt[b], 1u);
#pragma unroll
                for (int k2 = 0; k2 < NQ; k2++) atomicAdd(&h[k2 * kBuckets + b], (unsigned long long)fix_quant(pv[k2], pm[k2]));
                if constexpr (W && !GQ) atomicAdd(&siz[b], wi);
            }
        };
        unsigned i = threadIdx.x;
        for (; i + 512 < t.count; i += 1024) {
            const size_t p0 = t.start + i, p1 = p0 + 512;
            const double x0 = px[p0], y0 = py[p0], z0 = pz[p0], x1 = px[p1], y1 = py[p1], z1 = pz[p1];
            double w0 = 1.0, w1 = 1.0;
            if constexpr (W) { w0 = pw[p0]; w1 = pw[p1]; }
            one(p0, x0, y0, z0, w0);
            one(p1, x1, y1, z1, w1);
        }
        for (; i < t.count; i += 512) {
            const size_t p = t.start + i;
            double w = 1.0;
            if constexpr (W) w = pw[p];
            one(p, px[p], py[p], pz[p], w);
        }
        since_flush++;
        const bool flush = (ti + 1 == tlast) || tiles[ti + 1].node != t.node || since_flush >= 16;   // block-uniform
        if (!flush) continue;
        since_flush = 0;
        __syncthreads();
        const size_t slot = (size_t)nd.slot;
        double *gh = hist + slot * (size_t)(NQS * 2 * kBuckets);
        for (int i2 = threadIdx.x; i2 < NQ * kBuckets; i2 += blockDim.x) {
            const long long S = (long long)h[i2];
            if (S != 0) {
                const int q = i2 / kBuckets, bb = i2 % kBuckets;
                const bool quad = GQ && q >= 3 && q <= 9;
                const FixK &f = quad ? fquad : flin;
                // S quanta = Sh * 2^s + Sl: hi = Sh on the coarse grid, lo = Sl on the quantum (|Sh| < 2^51: B + log2(pixels) <= 51)
                const long long Sh = (S + (1LL << (f.s - 1))) >> f.s, Sl = S - (Sh << f.s);
                if (Sh != 0) unsafeAtomicAdd(&gh[(q * 2 + 0) * kBuckets + bb], (double)Sh * f.ghi);
                if (Sl != 0) unsafeAtomicAdd(&gh[(q * 2 + 1) * kBuckets + bb], (double)Sl * f.glo);
                h[i2] = 0ULL;
            }
        }
        for (int b = threadIdx.x; b < kBuckets; b += blockDim.x) {
            unsigned c = cnt[b];
            if (c) { atomicAdd(&hcount[slot * kBuckets + b], c); cnt[b] = 0u; }
            if constexpr (W && !GQ) { const unsigned long long s2 = siz[b]; if (s2) { atomicAdd(&hsize[slot * kBuckets + b], s2); siz[b] = 0ULL; } }
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------
// optimal cut: local.c:136-176 (prefix, objective, first arg-max) + the children's records
// --------------------------------------------------------------------------------------------
template <bool W>
__global__ __launch_bounds__(512) void k_cut(NodeDev *nodes, const int *__restrict__ round_nodes, const double *__restrict__ hist,
                                             const unsigned long long *__restrict__ hsize, const unsigned int *__restrict__ hcount,
                                             unsigned char *lut, const int fault, const RoundDyn *__restrict__ dyn) {
    if (dyn && (int)blockIdx.x >= dyn->nr) return;
    __shared__ double sd[4 * 16];
    __shared__ unsigned long long su[2 * 16];
    __shared__ double best_v[8];
    __shared__ int best_i[8];
    __shared__ double tot[8];
    __shared__ unsigned long long tots[2];
    __shared__ int s_split;
    const int b = threadIdx.x;
    NodeDev &nd = nodes[round_nodes[blockIdx.x]];
    const size_t slot = (size_t)nd.slot;
    const double *gh = hist + slot * (size_t)(kNQ_LQ * 2 * kBuckets);
    constexpr int NQ = W ? 4 : 3;
    double p[NQ][2];
    {
        // all prefixes of the table in three block scans (the parts are exact in any order): first parts, second parts, sizes
        double v0[NQ], v1[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) { v0[q] = gh[(q * 2 + 0) * kBuckets + b]; v1[q] = gh[(q * 2 + 1) * kBuckets + b]; }
        block_scan_incl_vec<double, NQ>(v0, sd);
        block_scan_incl_vec<double, NQ>(v1, sd);
#pragma unroll
        for (int q = 0; q < NQ; q++) { p[q][0] = v0[q]; p[q][1] = v1[q]; }
    }
    unsigned long long cs[2] = {(unsigned long long)hcount[slot * kBuckets + b], W ? hsize[slot * kBuckets + b] : 0ULL};
    block_scan_incl_vec<unsigned long long, 2>(cs, su);
    const unsigned long long cnt = cs[0], siz = W ? cs[1] : cs[0];
    __syncthreads();
    if (b == kBuckets - 1) {
#pragma unroll
        for (int q = 0; q < NQ; q++) { tot[q * 2] = p[q][0]; tot[q * 2 + 1] = p[q][1]; }
        tots[0] = cnt; tots[1] = siz;
    }
    __syncthreads();
    // objective_b = sum_j csl^2/sl + csr^2/sr  (local.c:150-168)
    double obj = 0;
    {
        const double sl = (double)siz, sr = (double)tots[1] - sl;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double csl = p[j][0] + p[j][1];
            double csr = (tot[j * 2] + tot[j * 2 + 1]) - csl;
            double v = 0;
            if (sl != 0) v += (csl * csl) / sl;
            if (sr != 0) v += (csr * csr) / sr;
            obj += v;
        }
    }
    // first maximum (vector.c:26-46: strict '>' scanning upwards)
    double bv = obj; int bi = b;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double ov = __shfl_down(bv, o, 64); int oi = __shfl_down(bi, o, 64);
        if (ov > bv || (ov == bv && (fault == 3 ? oi > bi : oi < bi))) { bv = ov; bi = oi; }
    }
    if ((b & 63) == 0) { best_v[b >> 6] = bv; best_i[b >> 6] = bi; }
    __syncthreads();
    if (b == 0) {
        for (int w = 1; w < 8; w++)
            if (best_v[w] > bv || (best_v[w] == bv && (fault == 3 ? best_i[w] > bi : best_i[w] < bi))) { bv = best_v[w]; bi = best_i[w]; }
        if (fault == 1) {                                       // patolette_amd_debug_fault(1): a WRONG cut -- one more occupied bucket goes left
            unsigned long long upto = 0;
            for (int i = 0; i <= bi; i++) upto += hcount[slot * kBuckets + i];
            int b2 = bi + 1;
            while (b2 < kBuckets - 1 && hcount[slot * kBuckets + b2] == 0u) b2++;
            if (b2 < kBuckets - 1 && upto + hcount[slot * kBuckets + b2] < tots[0]) bi = b2;
        }
        s_split = bi;
    }
    __syncthreads();
    const int split = s_split;
    lut[slot * kBuckets + b] = b > split ? 1 : 0;
    if (b == split) {
        nd.split = split;
        const unsigned long long nL = cnt, nR = tots[0] - cnt;
        nd.cbegin[0] = nd.begin; nd.cbegin[1] = nd.begin + nL; nd.cbegin[2] = nd.begin + nd.n;
        for (int side = 0; side < 2; side++) {
            NodeDev &ch = nodes[nd.child0 + side];
            ch.begin = side == 0 ? nd.begin : nd.begin + nL;
            ch.n = side == 0 ? nL : nR;
            ch.gn = ch.n;                                                   // the table holds the counts of every GPU sharing the image
            ch.buf = 1 - nd.buf;
            ch.slot = -1; ch.child0 = -1; ch.nchild = 0;
            ch.klin = nd.klin; ch.kquad = nd.kquad;
            ch.psplit = split | (nd.degenerate ? 0x10000 : 0);
            double s0[NQ], s1[NQ];
#pragma unroll
            for (int q = 0; q < NQ; q++) {
                s0[q] = side == 0 ? p[q][0] : tot[q * 2] - p[q][0];        // exact: all parts lie on the bin grids
                s1[q] = side == 0 ? p[q][1] : tot[q * 2 + 1] - p[q][1];
            }
            double sw = W ? (s0[NQ - 1] + s1[NQ - 1]) : (double)ch.n;
            ch.sw = sw;
            const double inv = 1 / sw;                                     // matrix2D.c:229-231: mean *= 1/sum(w)
            for (int j = 0; j < 3; j++) ch.mean[j] = (s0[j] + s1[j]) * inv;
            ch.axis[0] = ch.axis[1] = ch.axis[2] = 0;
        }
    }
    // the children's accumulators and outputs start empty: one store per thread instead of ~500 from the thread above
    for (int i = b; i < 2 * kNodeResetElems; i += kBuckets)
        node_reset_element(nodes[nd.child0 + i / kNodeResetElems], i % kNodeResetElems);
}

// --------------------------------------------------------------------------------------------
// stable partition of every split node's segment into its children (local.c:210-243,
// global.c:300-363): count -> scan -> scatter.  Order inside a child = order inside the
// parent, as the reference's index lists have it.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count(QuantBuffers qb, const Tile *__restrict__ tiles, const NodeDev *__restrict__ nodes,
                                               const unsigned char *__restrict__ lut, unsigned int *tilecnt, const int from_end,
                                               const RoundDyn *__restrict__ dyn) {
    __shared__ unsigned int c[kMaxChildren];
    __shared__ unsigned char sl[kBuckets];
    const unsigned nt = dyn ? (unsigned)dyn->ntP : gridDim.x;
    if (blockIdx.x >= nt) return;
    const unsigned tix = from_end ? nt - 1u - blockIdx.x : blockIdx.x;
    const Tile t = tiles[tix];
    const NodeDev &nd = nodes[t.node];
    const unsigned char *l = lut + (size_t)nd.slot * kBuckets;
    if (threadIdx.x < kMaxChildren) c[threadIdx.x] = 0;
    const int nch = nd.nchild;
    typedef unsigned short us8 __attribute__((ext_vector_type(8), aligned(2)));
    const int split = nd.split;
    if (nch == 2 && split >= 0) {
        // a local-quantiser round (k_cut's table is "bucket > split"): 8 buckets per thread in one 16-byte load, no table
        const unsigned i0 = threadIdx.x * 8u;
        unsigned right = 0;
        if (i0 + 8u <= t.count) {
            const us8 v = *reinterpret_cast<const us8 *>(qb.bkt + t.start + i0);
#pragma unroll
            for (int j = 0; j < 8; j++) right += (int)v[j] > split ? 1u : 0u;
        } else {
            for (unsigned i = i0; i < t.count && i < i0 + 8u; i++) right += (int)qb.bkt[t.start + i] > split ? 1u : 0u;
        }
        __syncthreads();                                         // c[] is zeroed
        right = wave_sum_u32(right);
        if ((threadIdx.x & 63) == 0 && right) atomicAdd(&c[1], right);
        __syncthreads();
        if (threadIdx.x == 0) c[0] = t.count - c[1];
        __syncthreads();
    } else if (nch == 2) {
        // two base clusters out of the global quantiser: the child comes from an LDS copy of the bucket -> child table
        for (int b = threadIdx.x; b < kBuckets; b += blockDim.x) sl[b] = l[b];
        __syncthreads();
        const unsigned i0 = threadIdx.x * 8u;
        unsigned right = 0;
        if (i0 + 8u <= t.count) {
            const us8 v = *reinterpret_cast<const us8 *>(qb.bkt + t.start + i0);
#pragma unroll
            for (int j = 0; j < 8; j++) right += sl[v[j]];
        } else {
            for (unsigned i = i0; i < t.count && i < i0 + 8u; i++) right += sl[qb.bkt[t.start + i]];
        }
        right = wave_sum_u32(right);
        if ((threadIdx.x & 63) == 0 && right) atomicAdd(&c[1], right);
        __syncthreads();
        if (threadIdx.x == 0) c[0] = t.count - c[1];
        __syncthreads();
    } else {
        __syncthreads();
        for (unsigned i = threadIdx.x; i < ((t.count + 255u) & ~255u); i += blockDim.x) {
            int child = i < t.count ? (int)l[qb.bkt[t.start + i]] : 255;
            for (int k = 0; k < nch; k++) {
                unsigned long long m = __ballot(child == k);
                if ((threadIdx.x & 63) == 0 && m) atomicAdd(&c[k], (unsigned)__popcll(m));
            }
        }
        __syncthreads();
    }
    if (threadIdx.x < kMaxChildren) tilecnt[(size_t)tix * kMaxChildren + threadIdx.x] = c[threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_scan(const int *__restrict__ round_nodes, const int *__restrict__ node_tile0, NodeDev *nodes,
                                              const unsigned int *__restrict__ tilecnt, unsigned long long *tileoff,
                                              const RoundDyn *__restrict__ dyn) {
    if (dyn && (int)blockIdx.x >= dyn->nr) return;
    // Exclusive prefix of the per-tile child counts of one node (one block per node).  Every thread owns a run of
    // consecutive tiles: it sums them (eight independent loads in flight at a time), ONE block scan per child orders the
    // thread totals, and the thread writes the offsets of its tiles -- two block scans for a binary split instead of one
    // per 1024 tiles with a global load on the carry chain.
    __shared__ unsigned long long su[16];
    __shared__ unsigned long long total;
    NodeDev &nd = nodes[round_nodes[blockIdx.x]];
    const int t0 = node_tile0[blockIdx.x], t1 = node_tile0[blockIdx.x + 1];
    const int C = (t1 - t0 + (int)blockDim.x - 1) / (int)blockDim.x;
    const int a = t0 + (int)threadIdx.x * C, b = min(t1, a + C);
    unsigned long long base = nd.begin;
    const int nch = nd.nchild;
    if (nch == 2) {
        // a binary split (every round of the local quantiser): both children in one pass -- their counts sit side by side
        // in a tile's record (one 8-byte load), one vector scan, one 16-byte store of the two offsets per tile
        __shared__ unsigned long long sv[2 * 16];
        __shared__ unsigned long long tot2[2];
        unsigned long long sum[2] = {0, 0};
        for (int g = a; g < b; g += 8) {
            uint2 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = g + q < b ? *reinterpret_cast<const uint2 *>(tilecnt + (size_t)(g + q) * kMaxChildren) : make_uint2(0u, 0u);
#pragma unroll
            for (int q = 0; q < 8; q++) { sum[0] += v[q].x; sum[1] += v[q].y; }
        }
        unsigned long long inc[2] = {sum[0], sum[1]};
        block_scan_incl_vec<unsigned long long, 2>(inc, sv);
        if (threadIdx.x == blockDim.x - 1) { tot2[0] = inc[0]; tot2[1] = inc[1]; }
        __syncthreads();
        const unsigned long long base1 = base + tot2[0];
        unsigned long long run0 = base + inc[0] - sum[0], run1 = base1 + inc[1] - sum[1];
        for (int g = a; g < b; g += 8) {
            uint2 v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = g + q < b ? *reinterpret_cast<const uint2 *>(tilecnt + (size_t)(g + q) * kMaxChildren) : make_uint2(0u, 0u);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (g + q < b) *reinterpret_cast<ulonglong2 *>(tileoff + (size_t)(g + q) * kMaxChildren) = make_ulonglong2(run0, run1);
                run0 += v[q].x; run1 += v[q].y;
            }
        }
        if (threadIdx.x == 0) { nd.cbegin[0] = base; nd.cbegin[1] = base1; nd.cbegin[2] = base1 + tot2[1]; }
        return;
    }
    for (int k = 0; k < nch; k++) {
        unsigned long long sum = 0;
        for (int g = a; g < b; g += 8) {
            unsigned v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = g + q < b ? tilecnt[(size_t)(g + q) * kMaxChildren + k] : 0u;
#pragma unroll
            for (int q = 0; q < 8; q++) sum += v[q];
        }
        const unsigned long long inc = block_scan_incl<unsigned long long>(sum, su);
        if (threadIdx.x == blockDim.x - 1) total = inc;
        unsigned long long run = base + inc - sum;
        for (int g = a; g < b; g += 8) {
            unsigned v[8];
#pragma unroll
            for (int q = 0; q < 8; q++) v[q] = g + q < b ? tilecnt[(size_t)(g + q) * kMaxChildren + k] : 0u;
#pragma unroll
            for (int q = 0; q < 8; q++) {
                if (g + q < b) tileoff[(size_t)(g + q) * kMaxChildren + k] = run;
                run += v[q];
            }
        }
        if (threadIdx.x == 0) nd.cbegin[k] = base;
        __syncthreads();
        base += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) nd.cbegin[nch] = base;
}

// INV (with COV): every product is split onto the exact grids before it is added, so the children's moments no longer depend
// on where the tile boundaries fall -- the same bits for any tiling, hence for any way of dealing an image out over several
// GPUs (patolette_amd_slice).  The default adds each thread's products in plain f64 first (a fixed order for a fixed tiling:
// deterministic, but the roundings move with the tile boundaries) and splits the 14 partials once per run of tiles.
template <bool W, bool COV, bool INV = false>
__global__ __launch_bounds__(256, INV ? 4 : 5) void k_scatter(QuantBuffers qb, const Tile *__restrict__ tiles, int ntiles, NodeDev *nodes,
                                                    const unsigned char *__restrict__ lut, const unsigned long long *__restrict__ tileoff,
                                                    const int from_end) {
    constexpr int R = kTileP / 256;
    __shared__ unsigned long long off[R][4][kMaxChildren];
    __shared__ double sm[28 * 4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long ltmask = (1ULL << lane) - 1ULL;
    // A block walks consecutive tiles (the tiles of one node are consecutive).  COV: each thread sums its products per
    // child and quantity in plain f64 -- a fixed set of pixels in a fixed order for a given tiling, so deterministic --
    // and the 14 per-thread partials are split onto the exact grids, reduced over the block and added atomically only
    // when the node changes or the block is done.
    if ((from_end & 4) && nodes[0].nchild == 2) return;      // gated launch (launch_partition): the binary kernel has this partition
    const int per = (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int bid = (from_end & 1) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int tfirst = bid * per, tlast = min(ntiles, tfirst + per);
    constexpr int NP = INV ? 14 : 7;
    double pl[NP], pr[NP];
    if constexpr (COV) {
#pragma unroll
        for (int i = 0; i < NP; i++) { pl[i] = 0; pr[i] = 0; }
    }
    for (int ti = tfirst; ti < tlast; ti++) {
        const Tile t = tiles[ti];
        const NodeDev &nd = nodes[t.node];
        const unsigned char *l = lut + (size_t)nd.slot * kBuckets;
        const int nch = nd.nchild, split = nd.split;
        unsigned cr[R];                                      // child (low byte) | rank within the wave << 8
#pragma unroll
        for (int r = 0; r < R; r++) {
            unsigned i = r * 256 + threadIdx.x;
            int ch = 255;
            if (i < t.count) {
                const unsigned short b = qb.bkt[t.start + i];
                if constexpr (COV) ch = (int)b > split ? 1 : 0;              // binary split: k_cut's table is b > split
                else ch = (int)l[b];
            }
            unsigned rk = 0;
            for (int k = 0; k < nch; k++) {
                unsigned long long m = __ballot(ch == k);
                if (ch == k) rk = (unsigned)__popcll(m & ltmask);
                if (lane == 0) off[r][wid][k] = (unsigned long long)__popcll(m);
            }
            cr[r] = (unsigned)ch | (rk << 8);
        }
        __syncthreads();
        if constexpr (COV) {
            // binary split: the 2 x (R x 4) per-wave counts are turned into offsets by one wavefront (lane = child * 32 + entry,
            // 32-lane prefix) instead of two threads walking 32 LDS entries each while the block waits
            static_assert(R * 4 == 32, "one 32-lane group per child");
            if (threadIdx.x < 64) {
                const int k = lane >> 5, e = lane & 31;
                const unsigned long long c = off[e >> 2][e & 3][k];
                unsigned long long inc = c;
#pragma unroll
                for (int o = 1; o < 32; o <<= 1) { const unsigned long long t2 = __shfl_up(inc, o, 32); if (e >= o) inc += t2; }
                off[e >> 2][e & 3][k] = tileoff[(size_t)ti * kMaxChildren + k] + inc - c;
            }
        } else if ((int)threadIdx.x < nch) {
            unsigned long long run = tileoff[(size_t)ti * kMaxChildren + threadIdx.x];
            for (int r = 0; r < R; r++)
                for (int w = 0; w < 4; w++) { unsigned long long c = off[r][w][threadIdx.x]; off[r][w][threadIdx.x] = run; run += c; }
        }
        __syncthreads();
        const double *sx = qb.buf[nd.buf], *sy = sx + qb.N, *sz = sy + qb.N, *sw = sz + qb.N;
        double *dx = qb.buf[1 - nd.buf], *dy = dx + qb.N, *dz = dy + qb.N, *dw = dz + qb.N;
        // COV (binary splits only): the children's centred moments are accumulated while their pixels pass through
        // registers -- pca.c:62-101 / cluster.c:111-152 about the child means k_cut already wrote
        double m0[2], m1[2], m2[2];
        BinK kinv{0.0, 0.0};
        if constexpr (COV) {
            const NodeDev &c0 = nodes[nd.child0], &c1 = nodes[nd.child0 + 1];
            m0[0] = c0.mean[0]; m1[0] = c0.mean[1]; m2[0] = c0.mean[2];
            m0[1] = c1.mean[0]; m1[1] = c1.mean[1]; m2[1] = c1.mean[2];
            if constexpr (INV) kinv = nd.kquad;
        }
#pragma unroll
        for (int r = 0; r < R; r++) {
            const int ch = (int)(cr[r] & 255u);
            if (ch != 255) {
                const size_t src = t.start + r * 256 + threadIdx.x;
                const size_t dst = off[r][wid][ch] + (cr[r] >> 8);
                const double x = sx[src], y = sy[src], z = sz[src];
                double w = 1.0;
                if constexpr (W) w = sw[src];
                dx[dst] = x; dy[dst] = y; dz[dst] = z;
                if constexpr (W) dw[dst] = w;
                if constexpr (COV) {
                    const bool right = ch != 0;
                    const double rf = right ? 1.0 : 0.0, lf = right ? 0.0 : 1.0;      // q * {0,1} exact, x + (+-0) = x
                    const double ex = x - (right ? m0[1] : m0[0]), ey = y - (right ? m1[1] : m1[0]), ez = z - (right ? m2[1] : m2[0]);
                    const double wx = w * ex, wy = w * ey, wz = w * ez;
                    const double q[7] = {wx * ex, wy * ex, wz * ex, wy * ey, wz * ey, wz * ez, ((ex * ex + ey * ey) + ez * ez) * w};
                    if constexpr (INV) {
#pragma unroll
                        for (int i = 0; i < 7; i++) {
                            double v0, v1;
                            bin_split(q[i], kinv, v0, v1);
                            pl[2 * i] = __builtin_fma(v0, lf, pl[2 * i]); pl[2 * i + 1] = __builtin_fma(v1, lf, pl[2 * i + 1]);
                            pr[2 * i] = __builtin_fma(v0, rf, pr[2 * i]); pr[2 * i + 1] = __builtin_fma(v1, rf, pr[2 * i + 1]);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 7; i++) { pl[i] = __builtin_fma(q[i], lf, pl[i]); pr[i] = __builtin_fma(q[i], rf, pr[i]); }
                    }
                }
            }
        }
        if constexpr (COV) {
            const bool flush = (ti + 1 == tlast) || tiles[ti + 1].node != t.node;       // block-uniform
            if (flush) {
                const BinK kq = nd.kquad;
                double a[28];
                if constexpr (INV) {
#pragma unroll
                    for (int i = 0; i < 14; i++) { a[i] = pl[i]; a[14 + i] = pr[i]; pl[i] = 0; pr[i] = 0; }
                } else {
#pragma unroll
                    for (int i = 0; i < 7; i++) {
                        bin_split(pl[i], kq, a[2 * i], a[2 * i + 1]);
                        bin_split(pr[i], kq, a[14 + 2 * i], a[14 + 2 * i + 1]);
                        pl[i] = 0; pr[i] = 0;
                    }
                }
                block_sum<28>(a, sm);
                if (threadIdx.x == 0) {
                    for (int side = 0; side < 2; side++) {
                        NodeDev &ch = nodes[nd.child0 + side];
                        for (int i = 0; i < 7; i++) {
                            if (a[14 * side + 2 * i] != 0.0) unsafeAtomicAdd(&ch.acc[blockIdx.x & (kSlots - 1)][i][0], a[14 * side + 2 * i]);
                            if (a[14 * side + 2 * i + 1] != 0.0) unsafeAtomicAdd(&ch.acc[blockIdx.x & (kSlots - 1)][i][1], a[14 * side + 2 * i + 1]);
                        }
                    }
                }
            }
        }
        __syncthreads();                                     // `off` is rewritten by the next tile
    }
}

// The binary splits of the local quantiser (every sweep of k_scatter<., true, .> above), software-pipelined.  That kernel
// makes one memory round trip after another -- eight bucket loads, then eight times (three pixel loads, wait, three stores),
// the stores counted in the same in-order vmcnt as the loads -- and spends 81 % of its wavefront time parked with ~30 KB in
// flight per CU, which is what 4.4 TB/s needs at ~2 us of latency and no more.  Here a wavefront issues ALL of a tile's
// pixel loads (24, 32 with weights) first, then the NEXT tile's bucket loads; the offsets are worked out while the pixels
// are on their way, and the next tile's ranks at the end of the trip, behind this tile's stores.  Nothing is waited for out
// of issue order (gfx9's vmcnt retires loads and stores in the order they were issued: a wait for a late load is a wait
// for everything before it), and what a trip hands to the next are ALU results (ranks, scalar offsets), never a load's
// destination: a register copy of one, moved by the allocator behind the next trip's loads, would wait for this trip's stores.
// Same pixels per thread, same order of the per-thread sums, same flush: bit-identical moments and layout.
template <bool W, bool INV>
__global__ __launch_bounds__(256, (W || INV) ? 2 : 3) void k_scatter_bin(QuantBuffers qb, const Tile *__restrict__ tiles, int ntiles, NodeDev *nodes,
                                                                        const NodeDev *__restrict__ nodes_ro,
                                                                        const unsigned long long *__restrict__ tileoff, const int from_end,
                                                                        const RoundDyn *__restrict__ dyn) {
    constexpr int R = kTileP / 256;
    static_assert(R * 4 == 32, "one 32-lane group per child");
    __shared__ unsigned int wcnt[2][2][R * 4];               // [tile parity][child][r * 4 + wavefront]: members per (round, wavefront)
    __shared__ double sm[28 * 4];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned long long ltmask = (1ULL << lane) - 1ULL;
    int nblk = (int)gridDim.x;
    if ((from_end & 2) && nodes_ro[0].nchild != 2) return;   // gated launch (launch_partition): the root's children are not two, the general kernel has them
    if (dyn) {                                               // the launch is an upper bound (see k_hist)
        ntiles = dyn->ntP;
        nblk = min(ntiles, nblk);
        if ((int)blockIdx.x >= nblk) return;
    }
    const int per = (ntiles + nblk - 1) / nblk;
    const int bid = (from_end & 1) ? nblk - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int tfirst = bid * per, tlast = min(ntiles, tfirst + per);
    if (tfirst >= tlast) return;
    constexpr int NP = INV ? 14 : 7;
    double pl[NP], pr[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) { pl[i] = 0; pr[i] = 0; }

    // bucket ids of a tile, one per round (clamped inside the tile: no lane is switched off, nothing branches), and where each
    // child's part of the tile starts (lane l asks for child l >> 5)
    auto fetch_buckets = [&](const Tile &tt, const int ti, unsigned (&b)[R], unsigned long long &toff) {
        const unsigned last = tt.count - 1u;
#pragma unroll
        for (int r = 0; r < R; r++) b[r] = qb.bkt[tt.start + min((unsigned)(r * 256) + threadIdx.x, last)];
        toff = tileoff[(size_t)ti * kMaxChildren + (lane >> 5)];
    };
    // ranks: child of each pixel (k_cut's table is bucket > split), position among the wavefront's pixels of that child;
    // the members per (round, wavefront) go to wcnt[par]
    auto classify = [&](const Tile &tt, const unsigned (&b)[R], const int split, const int par, unsigned (&cr)[R]) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            const bool valid = (unsigned)(r * 256) + threadIdx.x < tt.count;
            const bool right = (int)b[r] > split;
            const unsigned long long mR = __ballot(valid && right), mL = __ballot(valid && !right);
            const unsigned rk = (unsigned)__popcll((right ? mR : mL) & ltmask);
            cr[r] = (valid ? (right ? 1u : 0u) : 2u) | (rk << 8);
            if (lane == 0) { wcnt[par][0][r * 4 + wid] = (unsigned)__popcll(mL); wcnt[par][1][r * 4 + wid] = (unsigned)__popcll(mR); }
        }
    };
    auto lane64 = [](const unsigned long long v, const int l) {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32) |
               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    };
    Tile t = tiles[tfirst];
    unsigned cr[R];                                          // child (low byte, 2 = no pixel) | rank << 8, of the current tile
    unsigned long long baseL, baseR;                         // where the children's parts of the current tile start (wave-uniform)
    {
        unsigned b0[R];
        unsigned long long toff;
        fetch_buckets(t, tfirst, b0, toff);
        classify(t, b0, nodes_ro[t.node].split, 0, cr);
        baseL = lane64(toff, 0); baseR = lane64(toff, 32);
    }
    for (int ti = tfirst; ti < tlast; ti++) {
        const int par = (ti - tfirst) & 1;
        const NodeDev &nd = nodes_ro[t.node];
        const int child0 = nd.child0;
        const double *sx = qb.buf[nd.buf], *sy = sx + qb.N, *sz = sy + qb.N, *sw = sz + qb.N;
        double *dx = qb.buf[1 - nd.buf], *dy = dx + qb.N, *dz = dy + qb.N, *dw = dz + qb.N;
        // 1. every pixel of the tile this thread moves: 3 (4) loads per round, all in flight at once
        double x[R], y[R], z[R], w[R];
        {
            const unsigned last = t.count - 1u;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const size_t src = t.start + min((unsigned)(r * 256) + threadIdx.x, last);
                x[r] = sx[src]; y[r] = sy[src]; z[r] = sz[src];
                w[r] = 1.0;
                if constexpr (W) w[r] = sw[src];
            }
        }
        // 2. behind them, the next tile's buckets and offsets (straight-line: after the last tile the same tile is fetched
        //    once more, so that the waits below count exactly)
        const bool more = ti + 1 < tlast;                    // block-uniform
        const int tnx = more ? ti + 1 : ti;
        const Tile tn = tiles[tnx];
        unsigned bkn[R];
        unsigned long long toffn;
        fetch_buckets(tn, tnx, bkn, toffn);
        // 3. offsets: every wavefront scans the 2 x 32 counts itself, lane = child * 32 + (round * 4 + wavefront)
        __syncthreads();                                     // the only barrier of a tile: wcnt[par] is complete
        const unsigned c = wcnt[par][lane >> 5][lane & 31];
        const unsigned incl = wave_scan_incl_u32(c);
        const unsigned tot0 = (unsigned)__builtin_amdgcn_readlane((int)incl, 31);
        const unsigned excl = incl - c - (lane >= 32 ? tot0 : 0u);
        // the children's means (k_cut wrote them before this launch)
        const NodeDev &c0 = nodes_ro[child0], &c1 = nodes_ro[child0 + 1];
        const double m0L = c0.mean[0], m1L = c0.mean[1], m2L = c0.mean[2], m0R = c1.mean[0], m1R = c1.mean[1], m2R = c1.mean[2];
        BinK kinv{0.0, 0.0};
        if constexpr (INV) kinv = nd.kquad;
        // 4. stores + the children's centred moments (pca.c:62-101 / cluster.c:111-152), the pixels in the order they always had.
        // A lane without a pixel (the last tile of a node) is not switched off: it stores to the slack behind the planes and
        // adds with a zero factor, so nothing branches around the stores and the waits count every one of them (a skipped
        // branch makes the compiler wait as if the stores behind it had never been issued: the late rounds would then wait
        // for EARLIER STORES to land)
        const size_t slack = (size_t)(W ? 4 : 3) * qb.N + (size_t)lane;                  // 64 doubles behind the last plane (gq_prepare)
#pragma unroll
        for (int r = 0; r < R; r++) {
            const unsigned eL = (unsigned)__builtin_amdgcn_readlane((int)excl, r * 4 + wid);
            const unsigned eR = (unsigned)__builtin_amdgcn_readlane((int)excl, 32 + r * 4 + wid);
            const unsigned ch = cr[r] & 255u;
            const bool right = ch == 1u, none = ch == 2u;
            const size_t dst = (right ? baseR + eR : baseL + eL) + (cr[r] >> 8);
            dx[none ? slack : dst] = x[r]; dy[none ? slack - qb.N : dst] = y[r]; dz[none ? slack - 2 * qb.N : dst] = z[r];
            if constexpr (W) dw[none ? slack - 3 * qb.N : dst] = w[r];
            const double rf = right ? 1.0 : 0.0, lf = ch == 0u ? 1.0 : 0.0;          // q * {0,1} exact, x + (+-0) = x; no pixel: both 0
            const double ex = x[r] - (right ? m0R : m0L), ey = y[r] - (right ? m1R : m1L), ez = z[r] - (right ? m2R : m2L);
            const double wx = w[r] * ex, wy = w[r] * ey, wz = w[r] * ez;
            const double q[7] = {wx * ex, wy * ex, wz * ex, wy * ey, wz * ey, wz * ez, ((ex * ex + ey * ey) + ez * ez) * w[r]};
            if constexpr (INV) {
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    double v0, v1;
                    bin_split(q[i], kinv, v0, v1);
                    pl[2 * i] = __builtin_fma(v0, lf, pl[2 * i]); pl[2 * i + 1] = __builtin_fma(v1, lf, pl[2 * i + 1]);
                    pr[2 * i] = __builtin_fma(v0, rf, pr[2 * i]); pr[2 * i + 1] = __builtin_fma(v1, rf, pr[2 * i + 1]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 7; i++) { pl[i] = __builtin_fma(q[i], lf, pl[i]); pr[i] = __builtin_fma(q[i], rf, pr[i]); }
            }
        }
        const bool flush = !more || tn.node != t.node;                                  // block-uniform
        if (flush) {
            const BinK kq = nd.kquad;
            double a[28];
            if constexpr (INV) {
#pragma unroll
                for (int i = 0; i < 14; i++) { a[i] = pl[i]; a[14 + i] = pr[i]; pl[i] = 0; pr[i] = 0; }
            } else {
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    bin_split(pl[i], kq, a[2 * i], a[2 * i + 1]);
                    bin_split(pr[i], kq, a[14 + 2 * i], a[14 + 2 * i + 1]);
                    pl[i] = 0; pr[i] = 0;
                }
            }
            block_sum<28>(a, sm);
            if (threadIdx.x == 0) {
                for (int side = 0; side < 2; side++) {
                    NodeDev &chn = nodes[child0 + side];
                    for (int i = 0; i < 7; i++) {
                        if (a[14 * side + 2 * i] != 0.0) unsafeAtomicAdd(&chn.acc[blockIdx.x & (kSlots - 1)][i][0], a[14 * side + 2 * i]);
                        if (a[14 * side + 2 * i + 1] != 0.0) unsafeAtomicAdd(&chn.acc[blockIdx.x & (kSlots - 1)][i][1], a[14 * side + 2 * i + 1]);
                    }
                }
            }
        }
        // 5. the next tile's ranks and offsets, from the loads of step 2 (they sit BEFORE this tile's stores in the queue).
        // wcnt[par ^ 1]: the trip after the next writes wcnt[par] again only behind the next trip's barrier, which every
        // wavefront reaches after its reads of wcnt[par] above
        if (more) {
            classify(tn, bkn, nodes_ro[tn.node].split, par ^ 1, cr);
            baseL = lane64(toffn, 0); baseR = lane64(toffn, 32);
        }
        t = tn;
    }
}

// --------------------------------------------------------------------------------------------
// centred second moments + distortion: pca.c:62-101 (vcov, product order (w*c_j)*c_k, lower
// triangle) and cluster.c:111-152 (distortion = sum ((dx^2+dy^2)+dz^2)*w)
// --------------------------------------------------------------------------------------------
template <bool W>
__device__ __forceinline__ void cov_accumulate(const double *px, const double *py, const double *pz, const double *pw,
                                               size_t lo, size_t hi, const double m0, const double m1, const double m2,
                                               const BinK kq, double (&a)[14]) {
    for (size_t p = lo + threadIdx.x; p < hi; p += blockDim.x) {
        double dx = px[p] - m0, dy = py[p] - m1, dz = pz[p] - m2;
        double w = 1.0;
        if constexpr (W) w = pw[p];
        double wx = w * dx, wy = w * dy, wz = w * dz;
        bin_add(wx * dx, kq, a[0], a[1]);
        bin_add(wy * dx, kq, a[2], a[3]);
        bin_add(wz * dx, kq, a[4], a[5]);
        bin_add(wy * dy, kq, a[6], a[7]);
        bin_add(wz * dy, kq, a[8], a[9]);
        bin_add(wz * dz, kq, a[10], a[11]);
        bin_add(((dx * dx + dy * dy) + dz * dz) * w, kq, a[12], a[13]);
    }
}

// tiles cover the PARENT's range in the destination buffer; each child accumulates about its own mean
template <bool W>
__global__ __launch_bounds__(256) void k_cov_children(QuantBuffers qb, const Tile *__restrict__ tiles, NodeDev *nodes, const int from_end) {
    __shared__ double sm[14 * 4];
    if ((from_end & 4) && nodes[0].nchild == 2) return;      // gated launch: the binary partition took the children's moments along
    const Tile t = tiles[(from_end & 1) ? gridDim.x - 1u - blockIdx.x : blockIdx.x];
    const NodeDev &nd = nodes[t.node];
    const double *px = qb.buf[1 - nd.buf], *py = px + qb.N, *pz = py + qb.N, *pw = pz + qb.N;
    const size_t t_lo = t.start, t_hi = t.start + t.count;
    for (int k = 0; k < nd.nchild; k++) {
        size_t lo = nd.cbegin[k] > t_lo ? nd.cbegin[k] : t_lo;
        size_t hi = nd.cbegin[k + 1] < t_hi ? nd.cbegin[k + 1] : t_hi;
        if (lo >= hi) continue;                                  // block-uniform
        NodeDev &ch = nodes[nd.child0 + k];
        double a[14];
#pragma unroll
        for (int i = 0; i < 14; i++) a[i] = 0;
        cov_accumulate<W>(px, py, pz, pw, lo, hi, ch.mean[0], ch.mean[1], ch.mean[2], nd.kquad, a);
        block_sum<14>(a, sm);
        if (threadIdx.x == 0)
            for (int q = 0; q < 7; q++) { unsafeAtomicAdd(&ch.acc[blockIdx.x & (kSlots - 1)][q][0], a[2 * q]); unsafeAtomicAdd(&ch.acc[blockIdx.x & (kSlots - 1)][q][1], a[2 * q + 1]); }
    }
}

// tiles cover the node's own segment (root: `planar` = the converted image itself)
template <bool W>
__global__ __launch_bounds__(256) void k_cov_nodes(QuantBuffers qb, const double *planar, const Tile *__restrict__ tiles, NodeDev *nodes,
                                                   const int from_end) {
    __shared__ double sm[14 * 4];
    const Tile t = tiles[from_end ? gridDim.x - 1u - blockIdx.x : blockIdx.x];
    NodeDev &nd = nodes[t.node];
    const double *px = planar ? planar : qb.buf[nd.buf], *py = px + qb.N, *pz = py + qb.N, *pw = pz + qb.N;
    double a[14];
#pragma unroll
    for (int i = 0; i < 14; i++) a[i] = 0;
    cov_accumulate<W>(px, py, pz, pw, t.start, t.start + t.count, nd.mean[0], nd.mean[1], nd.mean[2], nd.kquad, a);
    block_sum<14>(a, sm);
    if (threadIdx.x == 0)
        for (int q = 0; q < 7; q++) { unsafeAtomicAdd(&nd.acc[blockIdx.x & (kSlots - 1)][q][0], a[2 * q]); unsafeAtomicAdd(&nd.acc[blockIdx.x & (kSlots - 1)][q][1], a[2 * q + 1]); }
}

// --------------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------------
size_t hist_slot_doubles() { return (size_t)kNQ_GQ * 2 * kBuckets; }

void launch_sum3(const double *planar, size_t N, BinK k, double *d_out6, hipStream_t s) {
    HIP_CHECK(hipMemsetAsync(d_out6, 0, kSum3Slots * 6 * sizeof(double), s));
    size_t g = ceil_div(N, 256 * 16);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    KTIME("k_sum3", s, 24.0 * N);
    hipLaunchKernelGGL(k_sum3, (int)g, 256, 0, s, planar, N, k, d_out6);
    HIP_CHECK(hipGetLastError());
}

void launch_minmax(const QuantBuffers &qb, const Tile *d_tiles, int ntiles, size_t px, NodeDev *d_nodes, hipStream_t s, bool from_end,
                   const RoundDyn *dyn, const double *px_src, const RoundLists *lists) {
    if (!ntiles) return;
    KTIME_DYN("k_minmax", s, 24.0, px, px_src);
    if (dyn && lists) hipLaunchKernelGGL(k_minmax<true>, ntiles, 256, 0, s, qb, d_tiles, d_nodes, from_end ? 1 : 0, dyn, *lists);
    else hipLaunchKernelGGL(k_minmax<false>, ntiles, 256, 0, s, qb, d_tiles, d_nodes, from_end ? 1 : 0, dyn, RoundLists{});
    HIP_CHECK(hipGetLastError());
}

template <bool W>
static void launch_hist_fix(const QuantBuffers &qb, const Tile *d_tiles, int ntiles, size_t px, NodeDev *d_nodes, double *d_hist,
                            unsigned long long *d_hsize, unsigned int *d_hcount, hipStream_t s, bool from_end) {
    constexpr int NQ = W ? 14 : 10;
    const size_t lds = (size_t)NQ * kBuckets * sizeof(unsigned long long) + kBuckets * (sizeof(unsigned int) + sizeof(unsigned long long));
    static PerDeviceOnce attr_set;
    if (attr_set.first()) HIP_CHECK(hipFuncSetAttribute((const void *)k_hist_fix<W, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KTIME("k_hist_gq", s, (W ? 34.0 : 26.0) * px);
    const int g = std::min(ntiles, 256 * (W ? 2 : 3));       // resident blocks per CU by LDS footprint (46 / 63 KB)
    hipLaunchKernelGGL((k_hist_fix<W, true>), g, 512, lds, s, qb, d_tiles, ntiles, d_nodes, d_hist, d_hsize, d_hcount, from_end ? 1 : 0);
    HIP_CHECK(hipGetLastError());
}

template <bool W, bool GQ>
static void launch_hist_t(const QuantBuffers &qb, const Tile *d_tiles, int ntiles, size_t px, NodeDev *d_nodes, double *d_hist,
                          unsigned long long *d_hsize, unsigned int *d_hcount, hipStream_t s, bool from_end,
                          const RoundDyn *dyn = nullptr, const double *px_src = nullptr) {
    constexpr int NQ = GQ ? (W ? 14 : 10) : (W ? 4 : 3);
    size_t lds = (size_t)NQ * 2 * kBuckets * sizeof(double) + kBuckets * (sizeof(unsigned int) + sizeof(unsigned long long));
    static PerDeviceOnce attr_set;
    if (attr_set.first()) HIP_CHECK(hipFuncSetAttribute((const void *)k_hist<W, GQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KTIME_DYN(GQ ? "k_hist_gq" : "k_hist_lq", s, (W ? 34.0 : 26.0), px, px_src);
    // resident blocks per CU by LDS footprint (4 for the local quantiser's 29 KB, 1 for the global quantiser's 82+ KB);
    // each block walks its run of tiles and flushes once per node run
    const int g = std::min(ntiles, 256 * (GQ ? 1 : 4));
    int fe = from_end ? 1 : 0;
#ifdef PAMD_KM_TRACE
    if (getenv("PAMD_HIST_NOATOMIC") && atoi(getenv("PAMD_HIST_NOATOMIC"))) fe |= 2;     // diagnostic: time the kernel without its LDS atomics (wrong results)
#endif
    hipLaunchKernelGGL((k_hist<W, GQ>), g, 512, lds, s, qb, d_tiles, ntiles, d_nodes, d_hist, d_hsize, d_hcount, fe, dyn);
    HIP_CHECK(hipGetLastError());
}

// --------------------------------------------------------------------------------------------
// Global quantiser DP on the device (global.c:232-280): E_k[n] = min_t E_{k-1}[t] + distortion(t, n).
// The 512-bucket moment table already lives in HBM; the reference's O(k * 512^2) double loop is ~0.3 ms per k on one
// host core (up to 12 k), here one launch per k with a block per n and the t-range across the block.  Same f64
// expressions in the same order (cells.c:141-182), same precedence among equal minima (the sequential loop runs t
// downwards with a strict '<', starting from t = n-1), so the cut table is bit-identical.  All kmax steps are run;
// the host applies the bias termination test (global.c:99-187) to the downloaded table step by step.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_gq_prefix(const double *__restrict__ hist, const unsigned int *__restrict__ hcount, GqDpDev *g) {
    // Five inclusive prefixes, each built sequentially as cells.c:114-136 does (0 = w0, 1..3 = w1[r], 4 = w2).  The 512-step
    // chains run out of LDS: the table is staged with coalesced loads first (a chain that loads from global memory inside
    // its loop pays a memory round trip per step: 35 us instead of 6).
    __shared__ double sh[4][kBuckets + 1];
    __shared__ unsigned long long sc[kBuckets + 1];
    const int b = threadIdx.x;                               // one bucket per thread
#pragma unroll
    for (int hq = 0; hq < 4; hq++)                           // GQ quantities 0..2 = sum c, 3 = sum |c|^2
        sh[hq][b + 1] = hist[(size_t)(hq * 2 + 0) * kBuckets + b] + hist[(size_t)(hq * 2 + 1) * kBuckets + b];
    sc[b + 1] = hcount[b];
    if (b < 4) sh[b][0] = 0;
    if (b == 4) sc[0] = 0;
    __syncthreads();
    if (b == 0) { unsigned long long a = 0; for (int i = 1; i <= kBuckets; i++) { a += sc[i]; sc[i] = a; } }
    else if ((b & 63) == 0 && b <= 256) {                    // four more wavefronts, one chain each
        double *t = sh[(b >> 6) - 1];
        double a = 0;
        for (int i = 1; i <= kBuckets; i++) { a = t[i] + a; t[i] = a; }    // the host mirrors: table[i] += table[i-1]
    }
    __syncthreads();
    for (int i = b; i <= kBuckets; i += 512) {
        g->w0[i] = sc[i];
        g->w1[0][i] = sh[0][i]; g->w1[1][i] = sh[1][i]; g->w1[2][i] = sh[2][i]; g->w2[i] = sh[3][i];
    }
}
__device__ __forceinline__ double gq_distortion(const GqDpDev *g, int a, int b) {      // cells.c:141-182
    if (g->w0[a] == g->w0[b]) return 0;
    const double q0 = g->w1[0][b] - g->w1[0][a], q1 = g->w1[1][b] - g->w1[1][a], q2 = g->w1[2][b] - g->w1[2][a];
    return g->w2[b] - g->w2[a] - (q0 * q0 + q1 * q1 + q2 * q2) / (double)(g->w0[b] - g->w0[a]);
}
__global__ void k_gq_init(GqDpDev *g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= kBuckets) g->E[0][i] = i >= 1 ? gq_distortion(g, 0, i) : 0.0;
}
// step k: reads E[(k & 1)], writes E[(k & 1) ^ 1] and cut[k][n]
__global__ __launch_bounds__(256) void k_gq_dp(GqDpDev *g, int k) {
    const int n = (int)blockIdx.x;                              // 0..512
    const double *Ep = g->E[k & 1];
    double *En = g->E[(k & 1) ^ 1];
    if (n < k + 1) { if (threadIdx.x == 0) En[n] = Ep[n]; return; }
    // candidates t = n-2 .. k-1 across the block; the sequential loop keeps the first strict minimum in descending t,
    // i.e. the largest t among equal minima, and t = n-1 (value Ep[n-1], no distortion term) precedes them all
    double best = INFINITY; int bt = -1;
    for (int t = n - 2 - (int)threadIdx.x; t >= k - 1; t -= (int)blockDim.x) {
        const double v = Ep[t] + gq_distortion(g, t, n);
        if (v < best) { best = v; bt = t; }                     // descending t within the thread: first minimum = largest t
    }
    __shared__ double sv[256];
    __shared__ int st[256];
    sv[threadIdx.x] = best; st[threadIdx.x] = bt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double v2 = sv[threadIdx.x + o]; const int t2 = st[threadIdx.x + o];
            const double v1 = sv[threadIdx.x]; const int t1 = st[threadIdx.x];
            if (t2 >= 0 && (t1 < 0 || v2 < v1 || (v2 == v1 && t2 > t1))) { sv[threadIdx.x] = v2; st[threadIdx.x] = t2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double e = Ep[n - 1]; int cut = n - 1;
        if (st[0] >= 0 && sv[0] < e) { e = sv[0]; cut = st[0]; }
        En[n] = e;
        g->cut[k][n] = cut;
    }
}

void launch_gq_dp(const double *d_hist, const unsigned int *d_hcount, int kmax, GqDpDev *d_g, hipStream_t s) {
    KTIME("k_gq_dp", s, (double)kmax * 513 * 513 * 8);
    hipLaunchKernelGGL(k_gq_prefix, 1, 512, 0, s, d_hist, d_hcount, d_g);
    hipLaunchKernelGGL(k_gq_init, 3, 256, 0, s, d_g);
    for (int k = 2; k <= kmax; k++) hipLaunchKernelGGL(k_gq_dp, kBuckets + 1, 256, 0, s, d_g, k);
    HIP_CHECK(hipGetLastError());
}

void launch_hist(const QuantBuffers &qb, bool gq, const Tile *d_tiles, int ntiles, size_t px, NodeDev *d_nodes,
                 double *d_hist, unsigned long long *d_hsize, unsigned int *d_hcount, hipStream_t s, bool from_end, bool fixed_point,
                 const RoundDyn *dyn, const double *px_src) {
    if (!ntiles) return;
    if (gq && fixed_point) {
        if (qb.weighted) launch_hist_fix<true>(qb, d_tiles, ntiles, px, d_nodes, d_hist, d_hsize, d_hcount, s, from_end);
        else launch_hist_fix<false>(qb, d_tiles, ntiles, px, d_nodes, d_hist, d_hsize, d_hcount, s, from_end);
        return;
    }
    if (qb.weighted) { if (gq) launch_hist_t<true, true>(qb, d_tiles, ntiles, px, d_nodes, d_hist, d_hsize, d_hcount, s, from_end);
                       else launch_hist_t<true, false>(qb, d_tiles, ntiles, px, d_nodes, d_hist, d_hsize, d_hcount, s, from_end, dyn, px_src); }
    else { if (gq) launch_hist_t<false, true>(qb, d_tiles, ntiles, px, d_nodes, d_hist, d_hsize, d_hcount, s, from_end);
           else launch_hist_t<false, false>(qb, d_tiles, ntiles, px, d_nodes, d_hist, d_hsize, d_hcount, s, from_end, dyn, px_src); }
}

// patolette_amd_debug_fault: a deliberately wrong decision rule, for the tests that show the tie prover tells a tie from a bug
std::atomic<int> g_debug_fault{0};
void launch_cut(bool weighted, NodeDev *d_nodes, const int *d_round_nodes, int nround, const double *d_hist,
                const unsigned long long *d_hsize, const unsigned int *d_hcount, unsigned char *d_lut, hipStream_t s, const RoundDyn *dyn, const double *nr_src) {
    if (!nround) return;
    KTIME_DYN("k_cut", s, (double)kNQ_LQ * 2 * kBuckets * 8, nround, nr_src);
    const int fault = g_debug_fault.load(std::memory_order_relaxed);
    if (weighted) hipLaunchKernelGGL(k_cut<true>, nround, 512, 0, s, d_nodes, d_round_nodes, d_hist, d_hsize, d_hcount, d_lut, fault, dyn);
    else hipLaunchKernelGGL(k_cut<false>, nround, 512, 0, s, d_nodes, d_round_nodes, d_hist, d_hsize, d_hcount, d_lut, fault, dyn);
    HIP_CHECK(hipGetLastError());
}

void launch_partition(const QuantBuffers &qb, const Tile *d_ptiles, int nptiles, size_t px, const int *d_round_nodes,
                      const int *d_node_tile0, int nround, NodeDev *d_nodes, const unsigned char *d_lut,
                      unsigned int *d_tilecnt, unsigned long long *d_tileoff, bool fuse_cov, hipStream_t s, bool invariant, bool from_end,
                      const RoundDyn *dyn, const double *px_src, bool gated) {
    // gated (the root's partition after k_gq_control, which alone knows how many base clusters there are): both forms are launched,
    // the binary one with the children's moments fused (fuse_cov) and the general one; each looks at the root's child count and one
    // of them returns at once
    const int fe = (from_end ? 1 : 0) | (gated ? 2 : 0);
    if (!nround) return;
    if (nptiles) { KTIME_DYN("k_count", s, 2.0, px, px_src); hipLaunchKernelGGL(k_count, nptiles, 256, 0, s, qb, d_ptiles, d_nodes, d_lut, d_tilecnt, fe & 1, dyn); }
    // (a GPU holding none of the round's pixels still needs the children's -- empty -- segments: k_scan runs regardless)
    { KTIME_DYN("k_scan", s, 12.0 * kMaxChildren / (double)kTileP, dyn ? px : (size_t)nptiles * kTileP, px_src);
      hipLaunchKernelGGL(k_scan, nround, 1024, 0, s, d_round_nodes, d_node_tile0, d_nodes, d_tilecnt, d_tileoff, dyn); }
    if (nptiles) {
        KTIME_DYN(fuse_cov ? "k_scatter_cov" : "k_scatter", s, (qb.weighted ? 66.0 : 50.0), px, px_src);
        if (fuse_cov) {
            static const bool v1 = getenv("PAMD_SCATTER_V1") && atoi(getenv("PAMD_SCATTER_V1")) != 0;   // the round-trip-per-round kernel (A/B)
            if (!v1 || dyn || gated) {
                const int gb = std::min(nptiles, 256 * ((invariant || qb.weighted) ? 2 : 3));
                if (invariant) {
                    if (qb.weighted) hipLaunchKernelGGL((k_scatter_bin<true, true>), gb, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, (const NodeDev *)d_nodes, d_tileoff, fe, dyn);
                    else hipLaunchKernelGGL((k_scatter_bin<false, true>), gb, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, (const NodeDev *)d_nodes, d_tileoff, fe, dyn);
                } else if (qb.weighted) hipLaunchKernelGGL((k_scatter_bin<true, false>), gb, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, (const NodeDev *)d_nodes, d_tileoff, fe, dyn);
                else hipLaunchKernelGGL((k_scatter_bin<false, false>), gb, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, (const NodeDev *)d_nodes, d_tileoff, fe, dyn);
                HIP_CHECK(hipGetLastError());
                if (gated) {
                    const int fg = (from_end ? 1 : 0) | 4;
                    if (qb.weighted) hipLaunchKernelGGL((k_scatter<true, false>), nptiles, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, d_lut, d_tileoff, fg);
                    else hipLaunchKernelGGL((k_scatter<false, false>), nptiles, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, d_lut, d_tileoff, fg);
                    HIP_CHECK(hipGetLastError());
                }
                return;
            }
            const int g = std::min(nptiles, 256 * (invariant ? 4 : 5));   // resident blocks per CU, each loops over its run of tiles
            if (invariant) {
                if (qb.weighted) hipLaunchKernelGGL((k_scatter<true, true, true>), g, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, d_lut, d_tileoff, fe);
                else hipLaunchKernelGGL((k_scatter<false, true, true>), g, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, d_lut, d_tileoff, fe);
            } else if (qb.weighted) hipLaunchKernelGGL((k_scatter<true, true>), g, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, d_lut, d_tileoff, fe);
            else hipLaunchKernelGGL((k_scatter<false, true>), g, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, d_lut, d_tileoff, fe);
        } else {
            if (qb.weighted) hipLaunchKernelGGL((k_scatter<true, false>), nptiles, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, d_lut, d_tileoff, fe);
            else hipLaunchKernelGGL((k_scatter<false, false>), nptiles, 256, 0, s, qb, d_ptiles, nptiles, d_nodes, d_lut, d_tileoff, fe);
        }
    }
    HIP_CHECK(hipGetLastError());
}

void launch_cov_children(const QuantBuffers &qb, const Tile *d_tiles, int ntiles, size_t px, NodeDev *d_nodes, hipStream_t s, bool from_end, bool gated) {
    if (!ntiles) return;
    KTIME("k_cov", s, (qb.weighted ? 32.0 : 24.0) * px);
    const int fe = (from_end ? 1 : 0) | (gated ? 4 : 0);
    if (qb.weighted) hipLaunchKernelGGL(k_cov_children<true>, ntiles, 256, 0, s, qb, d_tiles, d_nodes, fe);
    else hipLaunchKernelGGL(k_cov_children<false>, ntiles, 256, 0, s, qb, d_tiles, d_nodes, fe);
    HIP_CHECK(hipGetLastError());
}

void launch_cov_nodes(const QuantBuffers &qb, const double *planar_override, const Tile *d_tiles, int ntiles, size_t px,
                      NodeDev *d_nodes, hipStream_t s, bool from_end) {
    if (!ntiles) return;
    KTIME("k_cov", s, ((qb.weighted && !planar_override) ? 32.0 : 24.0) * px);
    // the root PCA is unweighted even when weights exist (global.c:407)
    if (qb.weighted && !planar_override) hipLaunchKernelGGL(k_cov_nodes<true>, ntiles, 256, 0, s, qb, planar_override, d_tiles, d_nodes, from_end ? 1 : 0);
    else hipLaunchKernelGGL(k_cov_nodes<false>, ntiles, 256, 0, s, qb, planar_override, d_tiles, d_nodes, from_end ? 1 : 0);
    HIP_CHECK(hipGetLastError());
}

}  // namespace pamd
