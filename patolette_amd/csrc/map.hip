// map.hip -- final palette mapping on gfx950: exact nearest-palette assignment and the
// Riemersma (Hilbert-curve) error-diffusion dither.
//
// k_nn_map replaces patolette__PALETTE_fill_palette_map_nearest (lib/src/palette/nearest.c:150-209,
// FLANN exact 1-NN, eps = 0): f64 distance ((d0^2)+d1^2)+d2^2 over all K entries, strict '<' in
// ascending index order (lowest index on ties).  Algorithmic traffic 24 B/px read + 1 (K<=256) or
// 4 B/px written; brute force at K = 256 is f64-VALU-bound, the palette comes
// through the scalar cache (wave-uniform index -> s_load).
//
// k_dither replaces patolette__DITHER_riemersma (lib/src/dither/riemersma.c:437-459): a chain of
// W*H steps, each choice feeding the 16-entry error queue.  One wavefront walks one chain: per
// 64-step batch the lanes decode 64 curve positions and prefetch their pixels in parallel, then
// the steps run in order -- error sum in reference order, K/64 palette entries per lane, wave
// arg-min with the lowest-index tie rule.  The chain's state is a function of its last sixteen
// choices, so the curve is cut into runs that are walked side by side from speculative warm-ups,
// verified at every boundary and repaired where the speculation missed: ~2000 runs of one
// wavefront each (DitherSeg), or ~130 000 runs of one LANE each on large images (DitherLanes).
#include "map.h"
#include "dither_slots.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <type_traits>

#define PAMD_POW_TABLES_IN_LDS
#include "color_device.h"
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

// --------------------------------------------------------------------------------------------
// Exact pruning for the NN map.  Brute force is 256 x 8 f64 flops per 25 algorithmic bytes
// (~7x over the f64 ridge): to be HBM-bound the kernel may only evaluate a handful of candidates
// per pixel, and to stay bit-exact the pruning must be provably lossless.
//
// A G^3 grid covers the pixels' bounding box.  For a cell C, U(C) = min_k maxdist^2(C, p_k) bounds the
// nearest distance of every x in C from above, so any entry j with mindist^2(C, p_j) > U(C) is strictly
// farther than the winner for all x in C and can be dropped.  Cell boxes are widened by 1e-9 of the
// range (cell index arithmetic rounds) and the test carries a 1e-12 relative margin, 9 orders above
// the f64 rounding of the distance expression, so the arg-min over the COMPUTED distances --
// including exact ties, kept in ascending index order -- is unchanged.  Cells whose list overflows
// fall back to the full scan.  (32^3 grid -> ~3 candidates per pixel.)
// --------------------------------------------------------------------------------------------
constexpr int kLutMax = 30;                  // candidates kept per cell: 15 in the primary record, 15 in the secondary

struct NNGrid {
    double lo[3], cw[3], inv[3];             // cell (i,j,k) spans lo + i*cw .. lo + (i+1)*cw
    int G;
};

// Coarse pass of the LUT build: a (G/4)^3 grid whose cells each cover 4x4x4 cells of the final grid.  The entries that
// survive the pruning rule on a coarse cell are a superset of the survivors on every cell inside it (a larger box has a
// smaller minimum distance to any entry and a larger upper bound), so the fine pass only tests those -- typically
// 20-40 of the 256 entries.  A coarse cell keeps up to kCoarseMax entries; more than that marks it "all".
constexpr int kCoarseMax = 96;
template <typename CandT>
__global__ __launch_bounds__(64) void k_nn_lut_coarse(const double *__restrict__ pal, int k, NNGrid g, CandT *__restrict__ clist /* [cells][1 + kCoarseMax] */,
                                                       float4 *__restrict__ rec32 /* nullptr, or [257] for k_nn_map_mid */) {
    // one wavefront per coarse cell, lanes across the palette entries
    const int Gc = g.G / 4;
    const int cell = (int)blockIdx.x, lane = (int)threadIdx.x;
    if (rec32 != nullptr && cell == 0) {
        // the f32 records of k_nn_map_mid's first pass and its margin (see the table's comment): entry 256 = {M, 0, 0, 0}
        double wmax = 0.0;
        for (int j = lane; j < 256; j += 64) {
            float4 r = make_float4(0.f, 0.f, 0.f, 3.0e38f);
            if (j < k) {
                const double a = pal[j] - g.lo[0], b = pal[k + j] - g.lo[1], c = pal[2 * k + j] - g.lo[2];
                const double w = (a * a + b * b) + c * c;
                wmax = fmax(wmax, w);
                r = make_float4((float)(-2.0 * a), (float)(-2.0 * b), (float)(-2.0 * c), (float)w);
            }
            rec32[j] = r;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wmax = fmax(wmax, __shfl_xor(wmax, o, 64));
        if (lane == 0) {
            double r2 = 0.0;
            for (int a = 0; a < 3; a++) { const double r = g.cw[a] * g.G; r2 += r * r; }
            const double M = 2.5 * 1.001 * 0x1.0p-24 * (9.0 * wmax + 5.0 * r2);
            float Mf = (float)M;
            if ((double)Mf < M) Mf = __uint_as_float(__float_as_uint(Mf) + 1u);       // rounded up
            if (!(M < 3.0e38)) Mf = INFINITY;                                           // everything ambiguous: the exact loop decides
            rec32[256] = make_float4(Mf, 0.f, 0.f, 0.f);
        }
    }
    const int idx[3] = {cell % Gc, (cell / Gc) % Gc, cell / (Gc * Gc)};
    double cl[3], ch[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double m = 1e-9 * (g.cw[a] * g.G) + 1e-300;
        cl[a] = g.lo[a] + (4 * idx[a]) * g.cw[a] - m;
        ch[a] = g.lo[a] + (4 * idx[a] + 4) * g.cw[a] + m;
    }
    const double *px = pal, *py = pal + k, *pz = pal + 2 * k;
    double U = INFINITY;
    for (int j = lane; j < k; j += 64) {
        const double p[3] = {px[j], py[j], pz[j]};
        double mx = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) { const double d = fmax(fabs(p[a] - cl[a]), fabs(p[a] - ch[a])); mx += d * d; }
        U = fmin(U, mx);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) U = fmin(U, __shfl_xor(U, o, 64));
    const double thr = U * (1.0 + 1e-12) + 1e-300;
    CandT *out = clist + (size_t)cell * (1 + kCoarseMax);
    int cnt = 0;                                                 // wave-uniform
    for (int j0 = 0; j0 < k; j0 += 64) {
        const int j = j0 + lane;
        bool keep = false;
        if (j < k) {
            const double p[3] = {px[j], py[j], pz[j]};
            double mn = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) { const double d = fmax(fmax(cl[a] - p[a], p[a] - ch[a]), 0.0); mn += d * d; }
            keep = mn <= thr;
        }
        const unsigned long long m = __ballot(keep);
        const int pos = cnt + (int)__popcll(m & ((1ULL << lane) - 1ULL));     // ascending index order is kept
        if (keep && pos < kCoarseMax) out[1 + pos] = (CandT)j;
        cnt += (int)__popcll(m);
    }
    if (lane == 0) out[0] = (CandT)(cnt <= kCoarseMax ? cnt : (CandT)~(CandT)0);      // all ones: test every entry
}

// --------------------------------------------------------------------------------------------
// Large images: a (G/2)^3 table of FOUR-BYTE entries lives in LDS (32^3 x 4 B = 128 KB of the CU's 160 KB), so the
// per-pixel lookup never leaves the CU -- on unsorted input the 16-byte records of the G^3 table are one random L2 line
// per pixel and that gather, not HBM, bounded the kernel.  An entry holds FOUR DISTINCT candidates in ascending order.
// The rule that fills it is the G^3 rule plus a bisector test against q* = the entry with the smallest maxdist: p is
// dropped when |x-p|^2 - |x-q*|^2 > 0 on the whole (widened) box -- linear in x, so its minimum sits in a corner -- with
// a 1e-12 relative margin; a dropped entry is strictly farther than q* everywhere in the cell, so it can neither win nor
// tie.  A cell with fewer than four survivors is filled up with palette entries FAR from it (more candidates never change
// an exact arg-min; distinct ones let the map kernel read "second smallest" off a 4-input min / max network).  About 90 %
// of the pixels of a noise image resolve there.  Cells with more than four survivors carry a marker (byte 0 > byte 1,
// impossible for an ascending list); their pixels are queued per wavefront in LDS and taken through the G^3 records.
// The boxes of THIS table are widened by 1e-5 of the range: the map kernel finds its cell in f32 (below).
//
// Round 5: the first pass of k_nn_map_mid is f32.  Behind the table sit 256 records {-2 (p - lo), |p - lo|^2} rounded to
// f32 (lo = the pixel box's corner: shifted coordinates keep the rounding relative to the box, not to the origin), and the
// pixel's four candidates cost t_j = fma(xf0, q0, fma(xf1, q1, fma(xf2, q2, w))) = |x - p_j|^2 - |x - lo|^2 up to
// E = 2^-24 (9 max_j |p_j - lo|^2 + 5 |hi - lo|^2) (1 + 1e-3): three instructions instead of eight f64 ones, one
// ds_read_b128 instead of three ds_read_b64.  The smallest t names the winner of the f64 expression whenever the second
// smallest lies more than M = 2.5 E above it (2 E would do; the reference's own f64 rounding is 1e-9 of that); otherwise
// -- about one pixel in a thousand -- the pixel is parked like an overflow pixel and decided by the exact f64 loop.
// --------------------------------------------------------------------------------------------
constexpr int kMidSurv = 48;                                   // survivors of the first rule listed per cell of the (G/2)^3 table
constexpr unsigned kMidOverflow = 0x00000001u;                 // bytes {1, 0, 0, 0}

// One wavefront fills the eight cells of the (G/2)^3 table that lie in one coarse block: lane = (cell << 3) | slice, the
// eight lanes of a cell share the list (slice s takes entries s, s + 8, ...); list order is kept through the ballots.
__device__ __forceinline__ void nn_mid_entries(const double *__restrict__ pal, const int k, const NNGrid &g, const unsigned char *__restrict__ cand,
                                               const int cidx0, const int cidx1, const int cidx2, unsigned int *__restrict__ mid) {
    const int lane = (int)threadIdx.x, m = lane >> 3, sl = lane & 7;
    // the seven palette entries farthest from the centre of this block of cells (wave-uniform): what short lists are filled up with
    int far7[7];
    {
        const double *qx = pal, *qy = pal + k, *qz = pal + 2 * k;
        const double ctr[3] = {g.lo[0] + (4 * cidx0 + 2) * g.cw[0], g.lo[1] + (4 * cidx1 + 2) * g.cw[1], g.lo[2] + (4 * cidx2 + 2) * g.cw[2]};
        double dv[4]; bool taken[4] = {false, false, false, false};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = lane + 64 * q;
            dv[q] = -1.0;
            if (j < k) { const double a = qx[j] - ctr[0], b = qy[j] - ctr[1], c = qz[j] - ctr[2]; dv[q] = (a * a + b * b) + c * c; }
        }
#pragma unroll
        for (int f = 0; f < 7; f++) {
            double bv = -2.0; int bj = 0x7fffffff;
#pragma unroll
            for (int q = 0; q < 4; q++) if (!taken[q] && dv[q] > bv) { bv = dv[q]; bj = lane + 64 * q; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double v2 = __shfl_xor(bv, o, 64); const int j2 = __shfl_xor(bj, o, 64);
                if (v2 > bv || (v2 == bv && j2 < bj)) { bv = v2; bj = j2; }
            }
            far7[f] = bj;
#pragma unroll
            for (int q = 0; q < 4; q++) taken[q] = taken[q] || (lane + 64 * q == bj);
        }
    }
    const bool all = cand[0] == 0xff;
    const int ntest = all ? k : (int)cand[0];
    const int idx[3] = {2 * cidx0 + (m & 1), 2 * cidx1 + ((m >> 1) & 1), 2 * cidx2 + (m >> 2)};
    double cl[3], ch[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        // the map kernel's cell index is (unsigned)(fl32(x - lo) * fl32(32 / range * (1 - 2^-18))): off by at most 32 (2^-18 + 3 * 2^-24)
        // = 1.3e-4 of a cell = 4e-6 of the range
        const double mg = 1e-5 * (g.cw[a] * g.G) + 1e-300;
        cl[a] = g.lo[a] + (2 * idx[a]) * g.cw[a] - mg;
        ch[a] = g.lo[a] + (2 * idx[a] + 2) * g.cw[a] + mg;
    }
    const double *px = pal, *py = pal + k, *pz = pal + 2 * k;
    double U = INFINITY; int qs = 0x7fffffff;
    for (int t = sl; t < ntest; t += 8) {
        const int j = all ? t : (int)cand[1 + t];
        const double p[3] = {px[j], py[j], pz[j]};
        double mx = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) { const double d = fmax(fabs(p[a] - cl[a]), fabs(p[a] - ch[a])); mx += d * d; }
        if (mx < U) { U = mx; qs = j; }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {                           // (min maxdist, lowest index) over the eight slices of the cell
        const double u2 = __shfl_xor(U, o, 64); const int q2i = __shfl_xor(qs, o, 64);
        if (u2 < U || (u2 == U && q2i < qs)) { U = u2; qs = q2i; }
    }
    const double thr = U * (1.0 + 1e-12) + 1e-300;
    // `far(p, o)`: |x-p|^2 - |x-o|^2 > 0 on the whole (widened) box, with a 1e-12 relative margin: linear in x, so the minimum over the
    // box is taken term by term; p is then strictly farther than o everywhere in the cell and can neither win nor tie
    auto far = [&](const int jp, const int jo) -> bool {
        const double p[3] = {px[jp], py[jp], pz[jp]}, o[3] = {px[jo], py[jo], pz[jo]};
        const double o2 = (o[0] * o[0] + o[1] * o[1]) + o[2] * o[2];
        double f = -o2, scale = o2;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double w = o[a] - p[a];
            f += 2.0 * fmin(cl[a] * w, ch[a] * w) + p[a] * p[a];
            const double big = fmax(fmax(fabs(cl[a]), fabs(ch[a])), fabs(p[a]));
            scale += 4.0 * big * big;
        }
        return f > 1e-12 * scale + 1e-300;
    };
    // first rule + the bisector test against q*; the survivors of the cell are listed in LDS (ascending index)
    __shared__ unsigned char sv[8][kMidSurv];
    int ns = 0;                                                  // uniform over the eight lanes of a cell
    for (int t0 = 0; t0 < ntest; t0 += 8) {                      // wave-uniform trip count (ntest is)
        const int t = t0 + sl;
        bool keep = false; int j = 0;
        if (t < ntest) {
            j = all ? t : (int)cand[1 + t];
            const double p[3] = {px[j], py[j], pz[j]};
            double mn = 0;
#pragma unroll
            for (int a = 0; a < 3; a++) { const double d = fmax(fmax(cl[a] - p[a], p[a] - ch[a]), 0.0); mn += d * d; }
            keep = mn <= thr && !far(j, qs);
        }
        const unsigned bits = (unsigned)((__ballot(keep) >> (8 * m)) & 0xffULL);
        const int pos = ns + __popc(bits & ((1u << sl) - 1u));
        if (keep && pos < kMidSurv) sv[m][pos] = (unsigned char)j;
        ns += __popc(bits);
    }
    __builtin_amdgcn_wave_barrier();
    // second rule, pairwise among the survivors (being strictly farther is a strict partial order: whatever is dropped is beaten by
    // something that stays).  Cells near a Voronoi vertex keep five or more; most keep <= 4.
    unsigned entry = 0; int cnt = 0;
    const int nsv = ns <= kMidSurv ? ns : 0;
    int nmax = nsv;                                              // wave-uniform trip count
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
    for (int u0 = 0; u0 < nmax; u0 += 8) {
        const int u = u0 + sl;
        bool keep = false; int j = 0;
        if (u < nsv) {
            j = (int)sv[m][u];
            keep = true;
            for (int v = 0; v < nsv; v++) { if (v != u && far(j, (int)sv[m][v])) { keep = false; break; } }
        }
        const unsigned bits = (unsigned)((__ballot(keep) >> (8 * m)) & 0xffULL);
        const int pos = cnt + __popc(bits & ((1u << sl) - 1u));
        if (keep && pos < 4) entry |= (unsigned)j << (8 * pos);
        cnt += __popc(bits);
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) entry |= (unsigned)__shfl_xor((int)entry, o, 64);
    if (cnt > 4 || ns > kMidSurv) entry = kMidOverflow;
    else {
        // fill up with far entries that are not in the list, then put the four in ascending order (distinct: byte 0 < byte 1)
        int c = cnt;
        unsigned b[4] = {entry & 0xffu, (entry >> 8) & 0xffu, (entry >> 16) & 0xffu, entry >> 24};
#pragma unroll
        for (int f = 0; f < 7; f++) {
            const unsigned j = (unsigned)far7[f];
            bool dup = j >= (unsigned)k;
#pragma unroll
            for (int t = 0; t < 4; t++) dup = dup || (t < c && b[t] == j);
            if (!dup && c < 4) {
#pragma unroll
                for (int t = 0; t < 4; t++) if (t == c) b[t] = j;
                c++;
            }
        }
        auto cswap = [](unsigned &u, unsigned &v) { const unsigned lo = u < v ? u : v, hi = u < v ? v : u; u = lo; v = hi; };
        cswap(b[0], b[1]); cswap(b[2], b[3]); cswap(b[0], b[2]); cswap(b[1], b[3]); cswap(b[1], b[2]);
        entry = c == 4 ? (b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24)) : kMidOverflow;     // (c < 4: fewer than four palette entries in all)
    }
    const int Gm = g.G / 2;
    if (sl == 0) mid[(idx[2] * Gm + idx[1]) * Gm + idx[0]] = entry;
}

// Per cell two records of 16 entries: primary = [count, c0..c14], secondary = [c15..c29, unused].
// count = 255 marks overflow (full scan).  One 16-byte (u8) / 32-byte (u16) load serves almost every pixel.
template <typename CandT>
__global__ __launch_bounds__(64) void k_nn_lut_build(const double *__restrict__ pal, int k, NNGrid g, CandT *__restrict__ lut,
                                                      CandT *__restrict__ lut2, const CandT *__restrict__ clist, unsigned int *__restrict__ mid) {
    // one wavefront per coarse cell, lane t = the fine cell (t & 3, (t >> 2) & 3, t >> 4) inside it: the list of entries
    // to test is the same for the whole wavefront (uniform loads)
    const int Gc = g.G / 4;
    const int cc = (int)blockIdx.x;
    const int cidx[3] = {cc % Gc, (cc / Gc) % Gc, cc / (Gc * Gc)};
    const int t64 = (int)threadIdx.x;
    int idx[3] = {4 * cidx[0] + (t64 & 3), 4 * cidx[1] + ((t64 >> 2) & 3), 4 * cidx[2] + (t64 >> 4)};
    const int cell = (idx[2] * g.G + idx[1]) * g.G + idx[0];
    // entries to test: the survivors of the enclosing coarse cell (ascending index), or all of them
    const CandT *cand = clist + (size_t)cc * (1 + kCoarseMax);
    const bool all = cand[0] == (CandT)~(CandT)0;
    const int ntest = all ? k : (int)cand[0];
    double cl[3], ch[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const double m = 1e-9 * (g.cw[a] * g.G) + 1e-300;
        cl[a] = g.lo[a] + idx[a] * g.cw[a] - m;
        ch[a] = g.lo[a] + (idx[a] + 1) * g.cw[a] + m;
    }
    const double *px = pal, *py = pal + k, *pz = pal + 2 * k;
    double U = INFINITY; int qs = 0;                            // the entry that attains the minimum survives the coarse rule
    for (int t = 0; t < ntest; t++) {
        const int j = all ? t : (int)cand[1 + t];
        const double p[3] = {px[j], py[j], pz[j]};
        double mx = 0;
#pragma unroll
        for (int a = 0; a < 3; a++) { const double d = fmax(fabs(p[a] - cl[a]), fabs(p[a] - ch[a])); mx += d * d; }
        if (mx < U) { U = mx; qs = j; }
    }
    const double thr = U * (1.0 + 1e-12) + 1e-300;
    const double q[3] = {px[qs], py[qs], pz[qs]};
    const double q2 = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    CandT *rec = lut + (size_t)cell * 16, *rec2 = lut2 + (size_t)cell * 16;
    int cnt = 0, last = 0;
    for (int t = 0; t < ntest; t++) {
        const int j = all ? t : (int)cand[1 + t];
        const double p[3] = {px[j], py[j], pz[j]};
        // second rule (see k_nn_lut_mid): dropped when strictly farther than q* on the whole box
        double mn = 0, f = -q2, scale = q2;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double d = fmax(fmax(cl[a] - p[a], p[a] - ch[a]), 0.0);
            mn += d * d;
            const double w = q[a] - p[a];
            f += 2.0 * fmin(cl[a] * w, ch[a] * w) + p[a] * p[a];
            const double big = fmax(fmax(fabs(cl[a]), fabs(ch[a])), fabs(p[a]));
            scale += 4.0 * big * big;
        }
        if (mn <= thr && !(f > 1e-12 * scale + 1e-300)) {
            if (cnt < 15) rec[1 + cnt] = (CandT)j;
            else if (cnt < kLutMax) rec2[cnt - 15] = (CandT)j;
            last = j;
            cnt++;
        }
    }
    for (int t = cnt; t < 15; t++) rec[1 + t] = (CandT)last;     // padding repeats the last entry: harmless to evaluate
    rec[0] = (CandT)(cnt <= kLutMax ? cnt : 255);              // 255 = overflow -> full scan
    // the eight cells of the (G/2)^3 table inside this block, from the same list (k_nn_map_mid)
    if constexpr (sizeof(CandT) == 1) {
        if (mid != nullptr) nn_mid_entries(pal, k, g, (const unsigned char *)cand, cidx[0], cidx[1], cidx[2], mid);
    }
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// A cell record consumed front to back by shifting (no dynamic register indexing -> no scratch).
template <typename CandT> struct LutRec;
template <> struct LutRec<unsigned char> {
    unsigned long long w0, w1;
    int left;
    __device__ __forceinline__ void load(const unsigned char *p) { const uint4 v = *(const uint4 *)p; w0 = ((unsigned long long)v.y << 32) | v.x; w1 = ((unsigned long long)v.w << 32) | v.z; left = 8; }
    __device__ __forceinline__ int next() {
        const int r = (int)(w0 & 0xffULL);
        w0 >>= 8;
        if (--left == 0) { w0 = w1; left = 8; }
        return r;
    }
};
template <> struct LutRec<unsigned short> {
    unsigned long long w0, w1, w2, w3;
    int left;
    __device__ __forceinline__ void load(const unsigned short *p) {
        const uint4 a = *(const uint4 *)p, b = *((const uint4 *)p + 1);
        w0 = ((unsigned long long)a.y << 32) | a.x; w1 = ((unsigned long long)a.w << 32) | a.z;
        w2 = ((unsigned long long)b.y << 32) | b.x; w3 = ((unsigned long long)b.w << 32) | b.z;
        left = 4;
    }
    __device__ __forceinline__ int next() {
        const int r = (int)(w0 & 0xffffULL);
        w0 >>= 16;
        if (--left == 0) { w0 = w1; w1 = w2; w2 = w3; left = 4; }
        return r;
    }
};

template <typename CandT>
__device__ __forceinline__ int nn_eval(const double x, const double y, const double z, LutRec<CandT> rec, const size_t cell,
                                       const CandT *__restrict__ lut2, const double4 *sp, const int k) {
    const int cnt = rec.next();                                  // entry 0 = count
    double bd = INFINITY; int best = 0;
    if (cnt != 255) {
        const int n1 = cnt < 15 ? cnt : 15;
        for (int t = 0; t < n1; t++) {
            const int j = rec.next();
            const double4 p = sp[j];
            const double d0 = x - p.x, d1 = y - p.y, d2 = z - p.z;
            const double d = (d0 * d0 + d1 * d1) + d2 * d2;
            if (d < bd) { bd = d; best = j; }                   // ascending j + strict '<' = lowest index on ties
        }
        if (cnt > 15) {
            LutRec<CandT> r2;
            r2.load(lut2 + cell * 16);
            for (int t = 15; t < cnt; t++) {
                const int j = r2.next();
                const double4 p = sp[j];
            const double d0 = x - p.x, d1 = y - p.y, d2 = z - p.z;
                const double d = (d0 * d0 + d1 * d1) + d2 * d2;
                if (d < bd) { bd = d; best = j; }
            }
        }
    } else {
        for (int j = 0; j < k; j++) {
            const double4 p = sp[j];
            const double d0 = x - p.x, d1 = y - p.y, d2 = z - p.z;
            const double d = (d0 * d0 + d1 * d1) + d2 * d2;
            if (d < bd) { bd = d; best = j; }
        }
    }
    return best;
}

__device__ __forceinline__ size_t nn_cell(const double x, const double y, const double z, const int G, const double lo0, const double lo1,
                                          const double lo2, const double in0, const double in1, const double in2) {
    int ix = (int)((x - lo0) * in0), iy = (int)((y - lo1) * in1), iz = (int)((z - lo2) * in2);
    ix = ix < 0 ? 0 : (ix >= G ? G - 1 : ix);
    iy = iy < 0 ? 0 : (iy >= G ? G - 1 : iy);
    iz = iz < 0 ? 0 : (iz >= G ? G - 1 : iz);
    return (size_t)(iz * G + iy) * G + ix;
}

template <typename OutT, typename CandT>
__global__ __launch_bounds__(256) void k_nn_map_lut(const double *__restrict__ c, size_t N, size_t n, const double *__restrict__ pal, int k,
                                                    NNGrid g, const CandT *__restrict__ lut, const CandT *__restrict__ lut2,
                                                    OutT *__restrict__ out) {
    extern __shared__ double4 spal[];                          // [k] {x, y, z, 0}: one ds_read_b128 + one ds_read_b64 per candidate
    for (int j = threadIdx.x; j < k; j += blockDim.x) spal[j] = make_double4(pal[j], pal[k + j], pal[2 * k + j], 0.0);
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int G = g.G;
    const double lo0 = g.lo[0], lo1 = g.lo[1], lo2 = g.lo[2], in0 = g.inv[0], in1 = g.inv[1], in2 = g.inv[2];
    for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 2 * stride) {
        // two independent pixels per trip: both cell records are in flight before either is consumed
        const size_t i1 = i0 + stride;
        const bool ok1 = i1 < n;
        const size_t j1 = ok1 ? i1 : i0;
        const double x0 = c[i0], y0 = c[N + i0], z0 = c[2 * N + i0];
        const double x1 = c[j1], y1 = c[N + j1], z1 = c[2 * N + j1];
        const size_t c0 = nn_cell(x0, y0, z0, G, lo0, lo1, lo2, in0, in1, in2), c1 = nn_cell(x1, y1, z1, G, lo0, lo1, lo2, in0, in1, in2);
        LutRec<CandT> r0, r1;
        r0.load(lut + c0 * 16);
        r1.load(lut + c1 * 16);
        const int b0 = nn_eval<CandT>(x0, y0, z0, r0, c0, lut2, spal, k);
        out[i0] = (OutT)b0;
        const int b1 = nn_eval<CandT>(x1, y1, z1, r1, c1, lut2, spal, k);
        if (ok1) out[i1] = (OutT)b1;
    }
}

// palette as three f64 arrays in LDS (6 KB): leaves room for the overflow queues next to the 128 KB table
struct PalSoA { const double *x, *y, *z; };

// LDS of k_nn_map_mid: table, f32 records, f64 palette, and what is left of the CU's 160 KB for the wavefronts' overflow queues
constexpr size_t kMidLdsFixed = 131072 + 256 * 16 + 3 * 256 * 8;
constexpr int mid_queue(const int waves) { const int q = (int)((163840 - kMidLdsFixed) / ((size_t)waves * 28)); return q > 64 ? 64 : (q & ~3); }   // parked pixels per wavefront (28 B each)
constexpr size_t mid_lds(const int waves) { return kMidLdsFixed + (size_t)waves * mid_queue(waves) * 28; }

// One parked pixel through its record of the G^3 table: the first eight entries unconditionally in two groups of four (a record is
// padded with its last entry, and re-evaluating an entry cannot change a strict-'<' arg-min), the second group only when some lane
// of the wavefront holds more than four; longer lists and overflowed cells (rare) take the general loop.
// (r = the first twelve bytes of the pixel's record lut[cell]; the cell itself is worked out again on the rare paths that need the
// rest of the record)
struct uint3r { unsigned x, y, z; };
__device__ __forceinline__ int nn_drain_one(const double x, const double y, const double z, const uint3r r, const unsigned char *__restrict__ lut,
                                            const unsigned char *__restrict__ lut2, const PalSoA sp, const int k, const int G, const double lo0,
                                            const double lo1, const double lo2, const double in0, const double in1, const double in2) {
    const int cnt = (int)(r.x & 0xffu);
    double bd = INFINITY; int best = 0;
    auto test = [&](const int j) {
        const double d0 = x - sp.x[j], d1 = y - sp.y[j], d2 = z - sp.z[j];
        const double d = (d0 * d0 + d1 * d1) + d2 * d2;
        if (d < bd) { bd = d; best = j; }                       // ascending j + strict '<' = lowest index on ties
    };
    test((int)((r.x >> 8) & 0xffu)); test((int)((r.x >> 16) & 0xffu)); test((int)(r.x >> 24)); test((int)(r.y & 0xffu));
    if (__any(cnt > 4)) {
        test((int)((r.y >> 8) & 0xffu)); test((int)((r.y >> 16) & 0xffu)); test((int)(r.y >> 24)); test((int)(r.z & 0xffu));
        if (__any(cnt > 8)) {
            if (cnt == 255) { bd = INFINITY; best = 0; for (int j = 0; j < k; j++) test(j); }
            else if (cnt > 8) {
                const size_t cell = nn_cell(x, y, z, G, lo0, lo1, lo2, in0, in1, in2);
                LutRec<unsigned char> rec;
                rec.load(lut + cell * 16);
                bd = INFINITY; best = 0;
                (void)rec.next();
                const int n1 = cnt < 15 ? cnt : 15;
                for (int t = 0; t < n1; t++) test(rec.next());
                if (cnt > 15) {
                    LutRec<unsigned char> r2;
                    r2.load(lut2 + cell * 16);
                    for (int t = 15; t < cnt; t++) test(r2.next());
                }
            }
        }
    }
    return best;
}

// 1024 threads = 16 wavefronts per CU around the 128 KB table.  A wavefront streams its own tiles of 64 * P pixels: a lane takes
// P / 2 pairs of CONSECUTIVE pixels (pair g = pixels 128 g + 2 lane, + 1): one 16-byte load per plane and pair, the next tile's
// loads in flight while this one is evaluated, one packed store per pair.
#ifdef PAMD_KM_TRACE
__device__ unsigned long long g_nn_trace[256][2];          // diagnostic build: when each block of k_nn_map_mid started and ended
// ... and what its drains met: [0] drains, [1] pixels drained, [2] drains with a list of more than four, [3] of more than eight,
// [4] pixels with more than eight, [5] pixels parked because the f32 pass could not tell first from second
__device__ unsigned long long g_nn_stats[8];
__device__ unsigned long long g_dl_wave[4096][4];  // ... and per wavefront of the speculative launch: clocks, steps with an exact pass, sum of the longest lists, steps with a crowded cell
__device__ unsigned long long g_dl_stats[8];      // k_dither_lanes (PAMD_NN_STATS): lane-steps, outside the grid, long-list cells, ambiguous; wavefront-steps, with an exact pass, with a full scan, candidate-loop trips
__device__ unsigned g_nn_flags;                             // timing experiments (wrong maps): 1 = drains evaluate nothing, 2 = nothing is parked
#endif
template <typename OutT, int P, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_nn_map_mid(const double *__restrict__ c, size_t N, size_t n, const double *__restrict__ pal, int k,
                                                     NNGrid g, const unsigned int *__restrict__ mid, const float4 *__restrict__ rec32,
                                                     const unsigned char *__restrict__ lut, const unsigned char *__restrict__ lut2, OutT *__restrict__ out) {
    extern __shared__ unsigned char smem_mid[];
    constexpr int Gm = 32, ncell = Gm * Gm * Gm;                                  // g.G == 64
    static_assert(P % 2 == 0, "pairs of pixels");
#ifdef PAMD_KM_TRACE
    if (threadIdx.x == 0) g_nn_trace[blockIdx.x & 255][0] = wall_clock64();
#endif
    unsigned int *T = (unsigned int *)smem_mid;                                  // [ncell]
    float4 *R = (float4 *)(smem_mid + (size_t)ncell * 4);                        // [256] {-2 (p - lo), |p - lo|^2} in f32
    double *spx = (double *)(R + 256), *spy = spx + 256, *spz = spy + 256;       // the f64 palette (the exact loop of the parked pixels)
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // this wavefront's overflow queue: coordinates and pixel index of up to kMidQueue pixels
    constexpr int kMidQueue = mid_queue(WAVES);
    double *qx = spz + 256 + wid * 3 * kMidQueue, *qy = qx + kMidQueue, *qz = qy + kMidQueue;
    unsigned int *qi = (unsigned int *)(spz + 256 + WAVES * 3 * kMidQueue) + wid * kMidQueue;
    const PalSoA sp{spx, spy, spz};

    constexpr size_t tile = (size_t)64 * P;
    const size_t step = (size_t)gridDim.x * WAVES * tile;
    double nx[P], ny[P], nz[P];
    auto fetch = [&](const size_t base) {                                         // wave-uniform base
        const double *cb = c + base;
        if (base + tile <= n && ((N | base) & 1) == 0) {                          // 16-byte aligned pairs in every plane
#pragma unroll
            for (int gq = 0; gq < P / 2; gq++) {
                // streamed once: non-temporal, so the pixels do not push the table's records and the palette out of L2
                typedef double d2_t __attribute__((ext_vector_type(2)));
                // (measured: 339 us instead of 349 for 67 M pixels)
                const d2_t vx = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(cb) + 64 * gq + lane),
                           vy = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(cb + N) + 64 * gq + lane),
                           vz = __builtin_nontemporal_load(reinterpret_cast<const d2_t *>(cb + 2 * N) + 64 * gq + lane);
                nx[2 * gq] = vx.x; nx[2 * gq + 1] = vx.y; ny[2 * gq] = vy.x; ny[2 * gq + 1] = vy.y; nz[2 * gq] = vz.x; nz[2 * gq + 1] = vz.y;
            }
        } else {
            const unsigned last = (unsigned)(n - base - 1);
#pragma unroll
            for (int p = 0; p < P; p++) { const unsigned t = min(128u * (unsigned)(p >> 1) + 2u * (unsigned)lane + (unsigned)(p & 1), last); nx[p] = cb[t]; ny[p] = cb[N + t]; nz[p] = cb[2 * N + t]; }
        }
    };
    size_t base = ((size_t)blockIdx.x * WAVES + wid) * tile;
    if (base < n) fetch(base);                                                    // in flight while the table is copied in
    for (int i = threadIdx.x; i < ncell / 4; i += 64 * WAVES) ((uint4 *)T)[i] = ((const uint4 *)mid)[i];
    for (int j = threadIdx.x; j < 256; j += 64 * WAVES) {
        const bool in = j < k;
        spx[j] = in ? pal[j] : 0.0; spy[j] = in ? pal[k + j] : 0.0; spz[j] = in ? pal[2 * k + j] : 0.0;
        R[j] = rec32[j];
    }
    __syncthreads();
    const int G = g.G;
    const double lo0 = g.lo[0], lo1 = g.lo[1], lo2 = g.lo[2], in0 = g.inv[0], in1 = g.inv[1], in2 = g.inv[2];
    // cell of the 32^3 table from the shifted f32 coordinates: 0 <= fl32(x - lo) * im < 32 (the factor 1 - 2^-18 absorbs the roundings)
    const float im0 = (float)(in0 * 0.5 * (1.0 - 0x1.0p-18)), im1 = (float)(in1 * 0.5 * (1.0 - 0x1.0p-18)), im2 = (float)(in2 * 0.5 * (1.0 - 0x1.0p-18));
    const float Mamb = rec32[256].x;                                              // wave-uniform: a scalar load
    const unsigned long long ltmask = (1ULL << lane) - 1ULL;
    int qn = 0;                                                                   // wave-uniform
    const bool out_even = (reinterpret_cast<size_t>(out) & 1) == 0;               // packed two-byte stores need an even address
    for (; base < n; base += step) {
        double x[P], y[P], z[P];
#pragma unroll
        for (int p = 0; p < P; p++) { x[p] = nx[p]; y[p] = ny[p]; z[p] = nz[p]; }
        const bool lasttile = base + step >= n;
        if (!lasttile) fetch(base + step);                                        // the next tile's pixels are in flight during this one
        const unsigned left = (unsigned)min((size_t)tile, n - base);
        OutT *ob = out + base;
        unsigned e[P];
        float xf[P], yf[P], zf[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            xf[p] = (float)(x[p] - lo0); yf[p] = (float)(y[p] - lo1); zf[p] = (float)(z[p] - lo2);
            const unsigned ix = (unsigned)(xf[p] * im0), iy = (unsigned)(yf[p] * im1), iz = (unsigned)(zf[p] * im2);
            e[p] = T[(((iz << 5) | iy) << 5) | ix];
        }
        unsigned ovbits = 0, bests[P];
#pragma unroll
        for (int p = 0; p < P; p++) {
            const unsigned t = 128u * (unsigned)(p >> 1) + 2u * (unsigned)lane + (unsigned)(p & 1);
            const bool ok = t < left;
            const unsigned j0 = e[p] & 0xffu, j1 = (e[p] >> 8) & 0xffu, j2 = (e[p] >> 16) & 0xffu, j3 = e[p] >> 24;
            const float4 r0 = R[j0], r1 = R[j1], r2 = R[j2], r3 = R[j3];
            // t_j = |x - p_j|^2 - |x - lo|^2 within E; the four candidates are distinct (the table's rule)
            const float t0 = __builtin_fmaf(xf[p], r0.x, __builtin_fmaf(yf[p], r0.y, __builtin_fmaf(zf[p], r0.z, r0.w)));
            const float t1 = __builtin_fmaf(xf[p], r1.x, __builtin_fmaf(yf[p], r1.y, __builtin_fmaf(zf[p], r1.z, r1.w)));
            const float t2 = __builtin_fmaf(xf[p], r2.x, __builtin_fmaf(yf[p], r2.y, __builtin_fmaf(zf[p], r2.z, r2.w)));
            const float t3 = __builtin_fmaf(xf[p], r3.x, __builtin_fmaf(yf[p], r3.y, __builtin_fmaf(zf[p], r3.z, r3.w)));
            const float lo01 = __builtin_fminf(t0, t1), hi01 = __builtin_fmaxf(t0, t1), lo23 = __builtin_fminf(t2, t3), hi23 = __builtin_fmaxf(t2, t3);
            const float m1 = __builtin_fminf(lo01, lo23);                                        // smallest
            const float m2 = __builtin_fminf(__builtin_fmaxf(lo01, lo23), __builtin_fminf(hi01, hi23));   // second smallest
            const bool clear = (m2 - m1) > Mamb;                                                 // (a NaN anywhere: not clear)
            const unsigned best = t0 == m1 ? j0 : (t1 == m1 ? j1 : (t2 == m1 ? j2 : j3));
#ifdef PAMD_KM_TRACE
            const bool ov = ok && (j0 > j1 || !clear) && !(g_nn_flags & 2u);
#else
            const bool ov = ok && (j0 > j1 || !clear);
#endif
#ifdef PAMD_NN_STATS
            { const unsigned long long ma = __ballot(ok && j0 < j1 && !clear); if (lane == 0 && ma) atomicAdd(&g_nn_stats[5], (unsigned long long)__popcll(ma)); }
#endif
            bests[p] = best;
            ovbits |= ov ? (1u << p) : 0u;
        }
        // results: parked pixels get a 0 for now, the drain writes theirs later (same wavefront, program order)
        if (left == (unsigned)tile && sizeof(OutT) == 1 && out_even) {               // (an odd `out`: image i of a batch with odd N)
#pragma unroll
            for (int gq = 0; gq < P / 2; gq++)
                reinterpret_cast<unsigned short *>(ob)[64 * gq + lane] = (unsigned short)((bests[2 * gq] & 0xffu) | (bests[2 * gq + 1] << 8));
        } else {
#pragma unroll
            for (int p = 0; p < P; p++) {
                const unsigned t = 128u * (unsigned)(p >> 1) + 2u * (unsigned)lane + (unsigned)(p & 1);
                if (t < left) ob[t] = (OutT)bests[p];
            }
        }
        if (__ballot(ovbits != 0u) || lasttile) {
            // park: the overflow pixels of this tile are numbered slot-major; as many as fit go into the queue, a full queue is
            // drained, and so on -- one loop around ONE drain site
            unsigned long long m[P];
            int off[P + 1];
            off[0] = 0;
#pragma unroll
            for (int p = 0; p < P; p++) { m[p] = __ballot((ovbits >> p) & 1u); off[p + 1] = off[p] + (int)__popcll(m[p]); }
            const int btot = off[P];
            int done = 0;
            do {                                                                  // wave-uniform
                const int take = min(kMidQueue - qn, btot - done);
#pragma unroll
                for (int p = 0; p < P; p++) {
                    const int gr = off[p] + (int)__popcll(m[p] & ltmask) - done;
                    if (((ovbits >> p) & 1u) && gr >= 0 && gr < take) {
                        qx[qn + gr] = x[p]; qy[qn + gr] = y[p]; qz[qn + gr] = z[p];
                        qi[qn + gr] = (unsigned int)base + 128u * (unsigned)(p >> 1) + 2u * (unsigned)lane + (unsigned)(p & 1);
                    }
                }
                qn += take; done += take;
                if (qn == kMidQueue || (lasttile && done == btot)) {
                    // a parked pixel's provisional result was stored above by ANOTHER lane of this wavefront: the fence orders
                    // the two stores (wavefront scope: no instruction, the wave's stores to one address already leave in order)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#ifdef PAMD_NN_STATS
                    {
                        unsigned c8 = 0;
                        if (lane < qn) c8 = lut[nn_cell(qx[lane], qy[lane], qz[lane], G, lo0, lo1, lo2, in0, in1, in2) * 16];
                        const unsigned long long m4 = __ballot(lane < qn && c8 > 4), m8 = __ballot(lane < qn && c8 > 8);
                        if (lane == 0) {
                            atomicAdd(&g_nn_stats[0], 1ULL); atomicAdd(&g_nn_stats[1], (unsigned long long)qn);
                            if (m4) atomicAdd(&g_nn_stats[2], 1ULL);
                            if (m8) { atomicAdd(&g_nn_stats[3], 1ULL); atomicAdd(&g_nn_stats[4], (unsigned long long)__popcll(m8)); }
                        }
                    }
#endif
#ifdef PAMD_KM_TRACE
                    if (lane < qn && !(g_nn_flags & 1u)) {
#else
                    if (lane < qn) {
#endif
                        const double px = qx[lane], py = qy[lane], pz = qz[lane];
                        const size_t cell = nn_cell(px, py, pz, G, lo0, lo1, lo2, in0, in1, in2);
                        out[qi[lane]] = (OutT)nn_drain_one(px, py, pz, *reinterpret_cast<const uint3r *>(lut + cell * 16), lut, lut2, sp, k, G, lo0, lo1, lo2, in0, in1, in2);
                    }
                    qn = 0;
                }
            } while (done < btot);
        }
    }
#ifdef PAMD_KM_TRACE
    __syncthreads();
    if (threadIdx.x == 0) g_nn_trace[blockIdx.x & 255][1] = wall_clock64();
#endif
}

__global__ __launch_bounds__(256) void k_minmax3(const double *__restrict__ c, size_t N, size_t n, unsigned long long *keys /* min[3], max[3] */) {
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
#pragma unroll
        for (int a = 0; a < 3; a++) { const double v = c[a * N + i]; mn[a] = fmin(mn[a], v); mx[a] = fmax(mx[a], v); }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
        double lo = mn[a], hi = mx[a];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_down(lo, o, 64)); hi = fmax(hi, __shfl_down(hi, o, 64)); }
        if ((threadIdx.x & 63) == 0 && lo <= hi) { atomicMin(&keys[a], f64_key(lo)); atomicMax(&keys[3 + a], f64_key(hi)); }
    }
}

static int stream_blocks(size_t n, int per_cu) {
    size_t g = ceil_div(n, 256);
    if (g > (size_t)256 * per_cu) g = (size_t)256 * per_cu;
    if (g < 1) g = 1;
    return (int)g;
}

template <typename OutT>
static void launch_nn_brute(const double *d_colors, size_t plane_stride, size_t n, const double *d_pal, int k, OutT *out, hipStream_t s) {
    KTIME("k_nn_map", s, (24.0 + sizeof(OutT)) * n);
    hipLaunchKernelGGL(k_nn_map<OutT>, stream_blocks(n, 32), 256, 0, s, d_colors, plane_stride, n, d_pal, k, out);
}

template <typename OutT>
static void launch_nn_lut(const double *d_colors, size_t plane_stride, size_t n, const double *d_pal, int k, OutT *out,
                          const NNGrid &g, NNWork &w, hipStream_t s) {
    const int ncell = g.G * g.G * g.G;
    const size_t lds = (size_t)4 * k * sizeof(double);
    if (k <= 256) {
        w.lut.reserve((size_t)ncell * 32);
        unsigned char *l1 = w.lut.p, *l2 = w.lut.p + (size_t)ncell * 16;
        static const bool mid_enabled = !(getenv("PAMD_NN_MID") && atoi(getenv("PAMD_NN_MID")) == 0);
        const bool use_mid = g.G == 64 && mid_enabled;                      // large images: four-candidate table in LDS
        if (use_mid) w.mid.reserve((size_t)(ncell / 8) + 257 * 4);        // the 32^3 table + the f32 records and margin behind it
        const int ncoarse = ncell / 64;
        w.clist.reserve((size_t)ncoarse * (1 + kCoarseMax) * 2);
        {
            KTIME("k_nn_lut_build", s, 32.0 * ncell);
            hipLaunchKernelGGL(k_nn_lut_coarse<unsigned char>, ncoarse, 64, 0, s, d_pal, k, g, w.clist.p, use_mid ? (float4 *)(w.mid.p + ncell / 8) : (float4 *)nullptr);
            hipLaunchKernelGGL(k_nn_lut_build<unsigned char>, ncoarse, 64, 0, s, d_pal, k, g, l1, l2, (const unsigned char *)w.clist.p, use_mid ? w.mid.p : nullptr);
        }
        if (use_mid) {
            if (n >> 32) throw HipError("patolette_amd: the LDS-table map kernel indexes pixels with 32 bits");
            // one-byte maps: sixteen wavefronts per CU with four pixels per lane; PAMD_NN_WAVES=12 = twelve with six (9 KB of loads in
            // flight per wavefront, 168 registers each): 294 against 281 us on the KMeans palette, level on the CIELuv one (interleaved
            // launches, tools/diag/nn_ab.py).  (Four pixels per lane with 4- or 8-byte map elements spill.)
            const int waves_env = getenv("PAMD_NN_WAVES") ? atoi(getenv("PAMD_NN_WAVES")) : 16;      // (read per call: an A/B alternates it)
            auto go = [&](auto pc, auto wc) {
                constexpr int P = decltype(pc)::value, WAVES = decltype(wc)::value;
                static PerDeviceOnce attr_mid;
                if (attr_mid.first()) HIP_CHECK(hipFuncSetAttribute((const void *)(k_nn_map_mid<OutT, P, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mid_lds(WAVES)));
                const int blocks = (int)std::min<size_t>((size_t)num_cus(), ceil_div(n, (size_t)64 * WAVES * P));
                KTIME("k_nn_map", s, (24.0 + sizeof(OutT)) * n);
                hipLaunchKernelGGL((k_nn_map_mid<OutT, P, WAVES>), blocks, 64 * WAVES, mid_lds(WAVES), s, d_colors, plane_stride, n, d_pal, k, g, (const unsigned int *)w.mid.p,
                                   (const float4 *)(w.mid.p + ncell / 8), (const unsigned char *)l1, (const unsigned char *)l2, out);
            };
            if constexpr (sizeof(OutT) == 1) {
                if (waves_env == 12) go(std::integral_constant<int, 6>{}, std::integral_constant<int, 12>{});
                else go(std::integral_constant<int, 4>{}, std::integral_constant<int, 16>{});
            } else go(std::integral_constant<int, 2>{}, std::integral_constant<int, 16>{});
            return;
        }
        KTIME("k_nn_map", s, (24.0 + sizeof(OutT)) * n);
        hipLaunchKernelGGL((k_nn_map_lut<OutT, unsigned char>), stream_blocks(n, 8), 256, lds, s, d_colors, plane_stride, n, d_pal, k, g, (const unsigned char *)l1, (const unsigned char *)l2, out);
    } else {
        w.lut.reserve((size_t)ncell * 64);
        unsigned short *l16 = (unsigned short *)w.lut.p, *l16b = l16 + (size_t)ncell * 16;
        const int ncoarse = ncell / 64;
        w.clist.reserve((size_t)ncoarse * (1 + kCoarseMax) * 2);
        {
            KTIME("k_nn_lut_build", s, 64.0 * ncell);
            hipLaunchKernelGGL(k_nn_lut_coarse<unsigned short>, ncoarse, 64, 0, s, d_pal, k, g, (unsigned short *)w.clist.p, (float4 *)nullptr);
            hipLaunchKernelGGL(k_nn_lut_build<unsigned short>, ncoarse, 64, 0, s, d_pal, k, g, l16, l16b, (const unsigned short *)w.clist.p, (unsigned int *)nullptr);
        }
        static PerDeviceOnce attr;
        if (attr.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void *)(k_nn_map_lut<OutT, unsigned short>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 4096 * 8));
        }
        KTIME("k_nn_map", s, (24.0 + sizeof(OutT)) * n);
        hipLaunchKernelGGL((k_nn_map_lut<OutT, unsigned short>), stream_blocks(n, 8), 256, lds, s, d_colors, plane_stride, n, d_pal, k, g, (const unsigned short *)l16, (const unsigned short *)l16b, out);
    }
}

// lo/hi: per-plane bounds of the colours (exact min/max), or nullptr to have them computed here
void launch_nn_map(const double *d_colors, size_t plane_stride, size_t n, const double *d_pal, int k, void *d_out, int elem_bytes,
                   const double *lo, const double *hi, NNWork &w, hipStream_t s) {
    if (elem_bytes != 1 && elem_bytes != 4 && elem_bytes != 8) throw HipError("patolette_amd: map element size must be 1, 4 or 8");
    const bool use_lut = n >= 16384 && k >= 8 && k <= 4096;
    if (!use_lut) {
        if (elem_bytes == 1) launch_nn_brute<unsigned char>(d_colors, plane_stride, n, d_pal, k, (unsigned char *)d_out, s);
        else if (elem_bytes == 4) launch_nn_brute<unsigned int>(d_colors, plane_stride, n, d_pal, k, (unsigned int *)d_out, s);
        else launch_nn_brute<unsigned long long>(d_colors, plane_stride, n, d_pal, k, (unsigned long long *)d_out, s);
        HIP_CHECK(hipGetLastError());
        return;
    }
    double blo[3], bhi[3];
    if (lo && hi) { for (int a = 0; a < 3; a++) { blo[a] = lo[a]; bhi[a] = hi[a]; } }
    else {
        w.keys.reserve(6);
        unsigned long long init[6] = {~0ULL, ~0ULL, ~0ULL, 0ULL, 0ULL, 0ULL}, got[6];
        HIP_CHECK(hipMemcpyAsync(w.keys.p, init, sizeof init, hipMemcpyHostToDevice, s));
        { KTIME("k_minmax3", s, 24.0 * n); hipLaunchKernelGGL(k_minmax3, stream_blocks(n, 16), 256, 0, s, d_colors, plane_stride, n, w.keys.p); }
        HIP_CHECK(hipMemcpyAsync(got, w.keys.p, sizeof got, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        for (int a = 0; a < 3; a++) { blo[a] = key_f64(got[a]); bhi[a] = key_f64(got[3 + a]); }
    }
    NNGrid g;
    g.G = n >= ((size_t)1 << 22) ? 64 : 32;
    for (int a = 0; a < 3; a++) {
        double r = bhi[a] - blo[a];
        if (!(r > 0) || !std::isfinite(r)) r = 0;
        g.lo[a] = blo[a];
        g.cw[a] = r / g.G;
        g.inv[a] = r > 0 ? g.G / r : 0.0;
    }
    if (elem_bytes == 1) launch_nn_lut<unsigned char>(d_colors, plane_stride, n, d_pal, k, (unsigned char *)d_out, g, w, s);
    else if (elem_bytes == 4) launch_nn_lut<unsigned int>(d_colors, plane_stride, n, d_pal, k, (unsigned int *)d_out, g, w, s);
    else launch_nn_lut<unsigned long long>(d_colors, plane_stride, n, d_pal, k, (unsigned long long *)d_out, g, w, s);
    HIP_CHECK(hipGetLastError());
}

// --------------------------------------------------------------------------------------------
// Riemersma dither
// --------------------------------------------------------------------------------------------
// Hilbert index -> (x, y) on the 2^L square; visiting order identical to traverse_level(L, UP)
// from (0,0) (riemersma.c:176-257), checked against the oracle's recorded walk.
__host__ __device__ __forceinline__ void hilbert_d2xy(int L, unsigned long long d, unsigned &xo, unsigned &yo) {
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

// ---- wave-level helpers for the serial chain (DPP: a few cycles per step, vs ~100 for ds_bpermute) ----
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// minimum over the 64 lanes, returned in every lane (v must not be NaN)
__device__ __forceinline__ double wave_min_f64(double v) {
    v = fmin(v, dpp_f64<0xB1>(v));            // quad_perm [1,0,3,2]
    v = fmin(v, dpp_f64<0x4E>(v));            // quad_perm [2,3,0,1]
    v = fmin(v, dpp_f64<0x124>(v));           // row_ror:4
    v = fmin(v, dpp_f64<0x128>(v));           // row_ror:8  -> every lane holds its 16-lane row minimum
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return fmin(fmin(r0, r1), fmin(r2, r3));
}

// unsigned minimum over the 64 lanes, wave-uniform
template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    // the minimum taken BY the DPP instruction (v_min_u32 with a permuted first operand): one instruction per step instead of
    // move + minimum; s_nop 1 = the two wait states a DPP read of a just-written VGPR needs (the assembler does not add them)
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0" : "+v"(v));                                   // -> every lane holds its 16-lane row minimum
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16),
                   r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    // three scalar minima (the compiler otherwise moves two of the values back into vector registers for a v_min3_u32)
    unsigned m23, m;
    asm("s_min_u32 %0, %2, %3\n\ts_min_u32 %1, %4, %5\n\ts_min_u32 %0, %0, %1" : "=&s"(m), "=&s"(m23) : "s"(r0), "s"(r1), "s"(r2), "s"(r3));
    return m;
}
// lowest lane holding the minimum of a non-negative double: non-negative doubles order like their bit patterns, so the
// high words decide (one 32-bit DPP minimum) and the low words only among lanes that tie on the high word
__device__ __forceinline__ int wave_argmin_nonneg_f64(double v) {
    const unsigned hi = (unsigned)__double2hiint(v), lo = (unsigned)__double2loint(v);
    const unsigned mh = wave_min_u32(hi);
    unsigned long long c = __ballot(hi == mh);
    if (__popcll(c) != 1) {                                              // several lanes share the high word (wave-uniform)
        const unsigned ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
        c = __ballot(hi == mh && lo == ml);
    }
    return (int)__builtin_ctzll(c);
}

// The dither's per-pixel search with FOUR entries per lane (K <= 256), first stage: bd = the lane's smallest distance (already taken),
// e = which of the lane's four entries attains it (the FIRST one: ascending index with strict '<'), and the 16-lane row minima of
// bd's high words.  One hand-scheduled block instead of two compiler-scheduled ones: a DPP instruction may read a register only two
// wait states after it was written, and so may a select read VCC after a 64-bit compare -- the four row-minimum steps used to
// stand behind an s_nop each, and the three compare / select pairs behind theirs; interleaved, each fills the other's gaps:
// 14 issue slots instead of 21 on a chain whose cost is its instruction count.  Returns the row minima (every lane its row's).
// (The wait states inside these two blocks are counted by hand for the CDNA3 / CDNA4 issue rules -- the hazard recogniser does not
// look inside inline asm -- and the K = 100 / 128 / 130 / 256 dither parity tests are their check after every toolchain change.)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "map.hip: the hand-scheduled DPP blocks of the dither are written for gfx950 (MI355X); see the Makefile's ARCH"
#endif
__device__ __forceinline__ unsigned dither_rows4(const double d0, const double d1, const double d2, const double d3, const double bd, int &e) {
    const unsigned hi = (unsigned)__double2hiint(bd);
    unsigned t; int ee;
    asm volatile("v_cmp_eq_f64_e32 vcc, %[d2], %[bd]\n\t"
                 "s_nop 0\n\t"
                 "v_min_u32_dpp %[t], %[hi], %[hi] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_e64 %[e], 3, 2, vcc\n\t"                   // entry 2 if it attains the minimum, else 3
                 "v_cmp_neq_f64_e32 vcc, %[d1], %[bd]\n\t"
                 "v_min_u32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_cndmask_b32_e32 %[e], 1, %[e], vcc\n\t"               // entry 1 if it does
                 "v_min_u32_dpp %[t], %[t], %[t] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cmp_neq_f64_e32 vcc, %[d0], %[bd]\n\t"
                 "s_nop 0\n\t"
                 "v_min_u32_dpp %[t], %[t], %[t] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_e32 %[e], 0, %[e], vcc"                     // entry 0 if it does
                 : [t] "=&v"(t), [e] "=&v"(ee)
                 : [hi] "v"(hi), [bd] "v"(bd), [d0] "v"(d0), [d1] "v"(d1), [d2] "v"(d2)
                 : "vcc");
    (void)d3;
    e = ee;
    return t;
}

// The same stage with two entries per lane (K <= 128): one compare / select pair fills two of the wait states.
__device__ __forceinline__ unsigned dither_rows2(const double d0, const double bd, int &e) {
    const unsigned hi = (unsigned)__double2hiint(bd);
    unsigned t; int ee;
    asm volatile("v_cmp_neq_f64_e32 vcc, %[d0], %[bd]\n\t"
                 "s_nop 0\n\t"
                 "v_min_u32_dpp %[t], %[hi], %[hi] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32_e64 %[e], 0, 1, vcc\n\t"                   // entry 0 if it attains the minimum, else 1
                 "s_nop 0\n\t"
                 "v_min_u32_dpp %[t], %[t], %[t] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %[t], %[t], %[t] row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_min_u32_dpp %[t], %[t], %[t] row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 0"
                 : [t] "=&v"(t), [e] "=&v"(ee)
                 : [hi] "v"(hi), [bd] "v"(bd), [d0] "v"(d0)
                 : "vcc");
    e = ee;
    return t;
}

// Where the t-th in-image pixel (0-based, in curve order) lies: d = first curve position of the aligned block of 64 positions
// (an 8 x 8 sub-square) that holds it, c = in-image pixels before that block.  t >= width * height: d = 4^L, c = width * height.
// A descent through the curve's quadrants; a quadrant's share of the image is a rectangle, so its count is a product.
__host__ __device__ __forceinline__ void dither_locate(int L, unsigned width, unsigned height, unsigned long long t,
                                              unsigned long long &d_out, unsigned long long &c_out) {
    unsigned long long d0 = 0, c = 0;
    for (int j = L; j > 3; j--) {                                    // children are 2^(j-1) squares of 4^(j-1) curve positions
        const unsigned side = 1u << (j - 1);
        const unsigned long long span = 1ULL << (2 * (j - 1));
        bool found = false;
        for (int q = 0; q < 4; q++) {
            const unsigned long long dq = d0 + (unsigned long long)q * span;
            unsigned x0, y0;
            hilbert_d2xy(L, dq, x0, y0);
            x0 &= ~(side - 1u); y0 &= ~(side - 1u);
            const unsigned long long nx = x0 >= width ? 0u : (width - x0 < side ? width - x0 : side);
            const unsigned long long ny = y0 >= height ? 0u : (height - y0 < side ? height - y0 : side);
            const unsigned long long cnt = nx * ny;
            if (t < c + cnt) { d0 = dq; found = true; break; }
            c += cnt;
        }
        if (!found) { d_out = 1ULL << (2 * L); c_out = c; return; }
    }
    d_out = d0; c_out = c;
}

// Segment-parallel form of the chain (DitherSeg::S > 1).  The reference pushes `original pixel - chosen colour` into its queue
// (riemersma.c:333-340): the state of the chain after any step is a pure function of the last sixteen (pixel, choice) pairs.  A
// chain started from a ZERO queue somewhere on the curve is therefore bit-identical to the true chain from the moment it has made
// the true chain's choice sixteen times in a row (measured with the oracle, tests/test_dither_lockon.py: 16 .. ~800 steps).
//   MODE 0, block b of S: the curve's in-image pixels are cut into S runs of equal length (boundaries moved to the start of the
//       aligned 64-position block, dither_locate).  The wavefront starts `warm` in-image pixels before its run from a zero queue,
//       keeps the choices of the last sixteen warm-up steps in side[b][0..16) and writes the choices of its own run to the map.
//   MODE 1, block b (>= 1): the run is final once the state it was started from is the true one, i.e. once the sixteen choices in
//       side[b] equal what the map holds at the sixteen in-image positions before the run.  If they differ, the wavefront rebuilds the
//       queue from the map and the image (the very subtractions the chain performs), records those choices in side[b], bumps
//       *repairs and walks the run again -- only until sixteen consecutive choices of one group equal the entries already
//       there: from that point the old chain was in the same state, and everything it wrote after is what this one would write.
// The host repeats MODE 1 until a launch repairs nothing: in such a launch nothing was written, every side[b] equalled the map,
// and by induction over b (run 0 starts from the true zero queue) the map is the reference's chain.  Exact by construction.
struct DitherSeg {
    unsigned S;                      // runs (1 = the whole image in one chain, no warm-up, no side records)
    unsigned warm;                   // in-image pixels of warm-up
    unsigned *side;                  // [S][16] choices the run's starting state was built from (0xFFFFFFFF = no record)
    unsigned *repairs;               // [0] bumped by every MODE-1 wavefront that had to walk its run again, [1] the lowest such run
    unsigned b0;                     // MODE 1: block 0 takes run b0 (1 in a verification pass)
    unsigned through;                // MODE 1: the walk does not end with the run but goes on until it meets the old chain (below)
};

// One wavefront walks one chain; every instruction of the chain is issued by that one wavefront (~2.5 ns each), so the count of
// instructions per pixel is what matters.
//
// Error sums, systolic: the reference forms, for every pixel, e = q[0] w[0] + ... + q[15] w[15] in that order over the last sixteen
// error vectors (riemersma.c:286-296) -- thirty dependent instructions per channel if one lane does it.  Here lane (c, d) = 16 c + d
// owns channel c of the pixels whose step number is d mod 16: its sum starts when step d - 16 has produced its error vector
// (term 0, weight w[0]) and takes one term per step, as each later error vector appears, so that after step d - 1 all sixteen
// terms are in, added in the reference's order.  Every step is then ONE multiply and ONE add in all lanes at once (each lane
// with the weight of its own phase), and the finished sum of the current pixel is read from the lane whose turn it is.
// Sixteen steps are unrolled so that phases and lanes are compile-time constants.  (Step numbers are the chain's own: a chain that
// starts mid-curve counts from its first pixel, and drops up to fifteen pixels so that its warm-up is whole groups of sixteen.)
//
// Pixels: blocks of 64 curve positions are decoded and loaded by the 64 lanes in parallel; the in-image ones are appended, in
// order, to a ring in LDS and consumed sixteen at a time (the last group may be short), so edges need no second code path.
//
// Nearest colour: all 64 lanes share the palette (lane L owns entries [L PER, (L + 1) PER), in registers when PER <= 4), per-lane
// best in ascending index with strict '<', then the lowest lane among the wave-wide minima = lowest index on ties.
// GT: the palette tables are in global memory (K > 3200).  A template parameter and not a run-time choice of pointer: the chosen
// colour is read once per pixel ON the serial chain, and through a pointer that may be either space the compiler issues a FLAT load
// (both address paths, vmcnt and lgkmcnt waited for) where the LDS table needs a ds_read_b64.
template <typename OutT, int PER, bool GT, int MODE>
__global__ __launch_bounds__(64) void k_dither(const double *__restrict__ img, size_t plane_stride, unsigned width, unsigned height,
                                               const double *__restrict__ pal /* planar (k,3), linear Rec2020 */, int k,
                                               OutT *out, DitherWeights wts, double *gtab, DitherSeg sg) {
    extern __shared__ double lds[];
    constexpr int kRing = 128;                                       // >= 64 + 15 pending pixels
    // the two palette tables live in LDS; a palette too large for that (K > 3200; the reference takes any K,
    // riemersma.c:437-459) keeps them in the 6 k doubles of global memory at `gtab` -- correct, and as slow as it sounds
    double *rpx = GT ? lds : lds + 6 * k;                            // [3][kRing] channels of the pending pixels
    auto table = [&]() { if constexpr (GT) return gtab; else return (double *)lds; };   // (keeps the LDS address space when !GT)
    auto praw = table();                                             // [3][k] raw palette
    auto pwt = praw + 3 * k;                                         // [3][k] palette scaled by (float)-cast weights (riemersma.c:419-425)
    unsigned int *rpos = reinterpret_cast<unsigned int *>(rpx + 3 * kRing);   // [kRing] their linear pixel numbers
    const int lane = threadIdx.x;
    const double fw[3] = {(double)(float)kRw, (double)(float)kGw, (double)(float)kBw};
    // (GT: every block stores the same values to the shared tables -- a benign race)
    for (int j = lane; j < k; j += 64)
        for (int c = 0; c < 3; c++) { const double a = pal[c * k + j]; praw[c * k + j] = a; pwt[c * k + j] = a * fw[c]; }
    __threadfence_block();                                           // (the tables may be global memory)
    __syncthreads();
    const int per = PER > 0 ? PER : (k + 63) / 64;
    // own entries in registers when they fit (PER <= 4)
    double ex[PER > 0 ? PER : 1], ey[PER > 0 ? PER : 1], ez[PER > 0 ? PER : 1];
    if constexpr (PER > 0) {
#pragma unroll
        for (int m = 0; m < PER; m++) {
            const int j = lane * PER + m;
            ex[m] = j < k ? pwt[j] : 1e300; ey[m] = j < k ? pwt[k + j] : 1e300; ez[m] = j < k ? pwt[2 * k + j] : 1e300;
        }
    }
    const unsigned mxd = width > height ? width : height;
    int L = 0;
    while ((1u << L) < mxd) L++;
    if (L == 0) return;                                              // riemersma.c:452-456
    const int ch = lane < 48 ? lane >> 4 : 2, dph = lane & 15;       // lanes 48..63 shadow channel 2 (never read)
    const double Wc = ch == 0 ? kRw : (ch == 1 ? kGw : kBw);         // query weights are the DOUBLE constants (riemersma.c:305-311)
    double wl[16];                                                   // weight this lane applies at step j (mod 16): w[15 - ((d - j - 1) mod 16)]
#pragma unroll
    for (int j = 0; j < 16; j++) {
        double v = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) v = (15 - ((dph - j - 1) & 15)) == i ? wts.w[i] : v;     // w_i = m^i / 16 (host libm, riemersma.c:360-373)
        wl[j] = v;
    }
    double keep[16];
#pragma unroll
    for (int j = 0; j < 16; j++) keep[j] = dph == j ? 0.0 : 1.0;
    double acc = 0.0;                                                // this lane's partial sum (the queue starts as zeros: 0 + x = x)
    const auto prw = praw + ch * k;

    // nearest palette entry of the query (qx, qy, qz)
    auto nearest = [&](const double qx, const double qy, const double qz) -> int {
        double bd = INFINITY; int bj = 0;
        if constexpr (PER > 0) {
            // the lane's smallest distance first (v_min_f64), the entry that attains it afterwards: the FIRST of the lane's
            // entries equal to the minimum = ascending index with strict '<' (distances are never NaN here)
            double dd[PER];
#pragma unroll
            for (int m = 0; m < PER; m++) {
                const double e0 = qx - ex[m], e1 = qy - ey[m], e2 = qz - ez[m];
                dd[m] = (e0 * e0 + e1 * e1) + e2 * e2;
            }
            bd = dd[0];
#pragma unroll
            for (int m = 1; m < PER; m++) bd = fmin(bd, dd[m]);
            if constexpr (PER == 4 || PER == 2) {
                int e;
                unsigned t;
                if constexpr (PER == 4) t = dither_rows4(dd[0], dd[1], dd[2], dd[3], bd, e);
                else t = dither_rows2(dd[0], bd, e);
                bj = e | (lane * PER);
                const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)t, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)t, 16),
                               r2 = (unsigned)__builtin_amdgcn_readlane((int)t, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)t, 48);
                unsigned m23, mh;
                asm("s_min_u32 %0, %2, %3\n\ts_min_u32 %1, %4, %5\n\ts_min_u32 %0, %0, %1" : "=&s"(mh), "=&s"(m23) : "s"(r0), "s"(r1), "s"(r2), "s"(r3));
                const unsigned hi = (unsigned)__double2hiint(bd), lo = (unsigned)__double2loint(bd);
                unsigned long long c = __ballot(hi == mh);
                if (__popcll(c) != 1) {                              // several lanes share the high word (wave-uniform): the low words decide
                    const unsigned ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
                    c = __ballot(hi == mh && lo == ml);
                }
                return __builtin_amdgcn_readlane(bj, (int)__builtin_ctzll(c));
            }
            bj = PER - 1;
#pragma unroll
            for (int m = PER - 2; m >= 0; m--) bj = dd[m] == bd ? m : bj;
            bj += lane * PER;
        } else {
            for (int m = 0; m < per; m++) {
                const int j = lane * per + m;
                if (j < k) {
                    const double e0 = qx - pwt[j], e1 = qy - pwt[k + j], e2 = qz - pwt[2 * k + j];
                    const double dd = (e0 * e0 + e1 * e1) + e2 * e2;
                    if (dd < bd) { bd = dd; bj = j; }
                }
            }
        }
        return __builtin_amdgcn_readlane(bj, wave_argmin_nonneg_f64(bd));
    };

    // ---- this wavefront's stretch of the curve: [d_begin, d_start) warm-up, [d_start, d_end) its own run ----
    const unsigned long long total = 1ULL << (2 * L);
    const unsigned long long npix = (unsigned long long)width * height;
    const unsigned b = MODE == 0 ? blockIdx.x : blockIdx.x + sg.b0;
    unsigned long long d_begin = 0, d_start = 0, d_end = total;
    unsigned head = 16, count = 16;                                  // ring positions (absolute); head is a multiple of 16
    unsigned *const side = sg.S > 1 ? sg.side + 16u * b : nullptr;
    if (sg.S > 1) {
        unsigned long long c_start = 0, c_begin = 0, cc;
        if (b > 0) dither_locate(L, width, height, npix * b / sg.S, d_start, c_start);
        if (b + 1 < sg.S && !(MODE == 1 && sg.through)) dither_locate(L, width, height, npix * (b + 1) / sg.S, d_end, cc);
        d_begin = d_start;
        if (b > 0) {
            // MODE 0: the warm-up; MODE 1: just the sixteen pixels before the run (their positions end up in the ring)
            const unsigned long long back = MODE == 0 ? sg.warm : 16u;
            dither_locate(L, width, height, c_start > back ? c_start - back : 0, d_begin, c_begin);
            const unsigned cw = (unsigned)(c_start - c_begin);
            if constexpr (MODE == 0) {
                count = 16u - (cw & 15u);                            // drop cw mod 16 pixels: the warm-up ends with a whole group
                if (lane < 16) side[lane] = 0xFFFFFFFFu;             // (stays if the warm-up is shorter than one group)
            } else {
                count = 0; head = 0;
            }
        }
    }

    bool done = false;                                               // MODE 1: the old chain has been met
    unsigned met = 0;                                                // ... for this many groups of sixteen in a row
    const unsigned need = (MODE == 1 && sg.through) ? (unsigned)(npix / sg.S / 16) + 3u : 1u;
    // up to sixteen pending pixels, in order; `limit` < 16 only for the very last group
    auto group = [&](const int limit, const bool warmup) {
        double pcs[16];
#pragma unroll
        for (int j = 0; j < 16; j++) pcs[j] = rpx[ch * kRing + ((head + (unsigned)j) & (kRing - 1))];
        unsigned was = 0, wpos = 0;
        if (MODE == 1 && lane < limit) { wpos = rpos[(head + (unsigned)lane) & (kRing - 1)]; was = (unsigned)out[wpos]; }
        int res = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j < limit) {                                         // wave-uniform
                const double qv = Wc * (pcs[j] + acc);               // meaningful in lanes (c, j): their sums are complete (riemersma.c:298-311)
                const int bi = nearest(readlane_f64(qv, j), readlane_f64(qv, 16 + j), readlane_f64(qv, 32 + j));
                res = lane == j ? bi : res;
                const double err = pcs[j] - prw[bi];                 // original pixel - chosen colour (riemersma.c:333-340)
                // lane (c, j) starts the sum of step + 16 with term 0: acc * keep + t with keep = 0 there, 1 elsewhere -- the product is
                // exact, so the fused form rounds exactly like (keep ? acc : 0) + t, in two instructions instead of four
                acc = __builtin_fma(acc, keep[j], err * wl[j]);
            }
        }
        if constexpr (MODE == 0) {
            if (warmup) { if (lane < 16) side[lane] = (unsigned)res; }                           // the last group's stay
            else if (lane < limit) out[rpos[(head + (unsigned)lane) & (kRing - 1)]] = (OutT)res;
        } else {
            if (lane < limit) out[wpos] = (OutT)res;
            // met: a whole group equals what is there.  A walk THROUGH the runs behind asks for more than a run's length of it: in a
            // flat stretch the runs it crosses hold the same cycle at random phases, one in ten happens to be in step
            if (limit == 16 && __all(lane >= 16 || was == (unsigned)res)) { if (++met >= need) done = true; }
            else met = 0;
        }
        head += 16;
    };

    const double *pr = img, *pg = img + plane_stride, *pb = img + 2 * plane_stride;
    // A block of 64 aligned curve positions is an 8 x 8 sub-square: the lane's place inside it (levels 0..2 of the curve) never
    // changes, and the levels above act on it as ONE signed permutation + offset per block, worked out on scalars.
    unsigned xl = 0, yl = 0;
    hilbert_d2xy(3, (unsigned long long)lane, xl, yl);
    const bool needs_skip = width < (1u << L) || height < (1u << L);
    // decode the block of 64 positions at d0 and append its in-image pixels to the ring; false = an empty aligned square was skipped
    auto load_block = [&](unsigned long long &d0) -> bool {
        if (L >= 3 && needs_skip) {                                  // skip whole out-of-image aligned sub-squares
            for (int j = L; j >= 3; j--) {
                const unsigned long long span = 1ULL << (2 * j);
                if ((d0 & (span - 1)) != 0) continue;
                unsigned x0, y0;
                hilbert_d2xy(L, d0, x0, y0);
                x0 &= ~((1u << j) - 1u); y0 &= ~((1u << j) - 1u);
                if (x0 >= width || y0 >= height) { d0 += span; return false; }
            }
        }
        const unsigned long long dl = d0 + (unsigned long long)lane;
        unsigned x = 0, y = 0;
        bool inb = false;
        if (L >= 3) {
            // levels 3 .. L-1 on (selx ? yl : xl) * sgnx + ox (and the same for y): a flip negates and reflects the offset, a swap
            // exchanges the two triples, then the quadrant offset is added -- hilbert_d2xy's loop body on the coefficients
            int selx = 0, sgnx = 1, ox = 0, sely = 1, sgny = 1, oy = 0;
            unsigned long long t = d0 >> 6;
            for (int lv = 3; lv < L; lv++) {
                const int sidx = 1 << lv;
                const int rx = 1 & (int)(t >> 1);
                const int ry = 1 & ((int)t ^ rx);
                if (ry == 0) {
                    if (rx == 1) { sgnx = -sgnx; ox = sidx - 1 - ox; sgny = -sgny; oy = sidx - 1 - oy; }
                    int tmp = selx; selx = sely; sely = tmp;
                    tmp = sgnx; sgnx = sgny; sgny = tmp;
                    tmp = ox; ox = oy; oy = tmp;
                }
                ox += sidx * rx; oy += sidx * ry;
                t >>= 2;
            }
            x = (unsigned)((int)(selx ? yl : xl) * sgnx + ox);
            y = (unsigned)((int)(sely ? yl : xl) * sgny + oy);
            inb = x < width && y < height;
        } else if (dl < total) {
            hilbert_d2xy(L, dl, x, y);
            inb = x < width && y < height;
        }
        const unsigned long long mask = __ballot(inb);
        if (inb) {                                                   // append to the ring in curve order
            const size_t p = (size_t)y * width + x;
            const unsigned slot = (count + (unsigned)__popcll(mask & ((1ULL << lane) - 1ULL))) & (kRing - 1);
            rpx[slot] = pr[p]; rpx[kRing + slot] = pg[p]; rpx[2 * kRing + slot] = pb[p];
            rpos[slot] = (unsigned)p;
        }
        count += (unsigned)__popcll(mask);
        __builtin_amdgcn_wave_barrier();
        d0 += 64;
        return true;
    };

    unsigned long long d0 = d_begin;
    if constexpr (MODE == 1) {
        // the sixteen in-image pixels before the run: at most 16 + 63 pixels lie between d_begin and d_start
        while (d0 < d_start) load_block(d0);
        const unsigned slot = (count - 16u + (unsigned)dph) & (kRing - 1);
        const unsigned cur = (unsigned)out[rpos[slot]];              // every lane: the choice the map holds for predecessor dph
        const bool same = __all(lane >= 16 || cur == side[lane & 15]);
        if (same) return;                                            // the run was started from the true state
        if (lane < 16) side[lane] = cur;
        if (lane == 0) { atomicAdd(sg.repairs, 1u); atomicMin(sg.repairs + 1, b); }
        // the queue as the chain holds it after those sixteen steps: their error vectors pushed in order into zero sums -- lane
        // (c, d) restarts at step d, so whatever it summed before never reaches a pixel
        const double ev = rpx[ch * kRing + slot] - prw[cur];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double e = __shfl(ev, (lane & 48) | j, 64);
            acc = __builtin_fma(acc, keep[j], e * wl[j]);
        }
        head = count = 16;
        __builtin_amdgcn_wave_barrier();
    }
    while (d0 < d_end && !done) {
        const bool warmup = d0 < d_start;
        if (!load_block(d0)) continue;
        while ((int)(count - head) >= 16 && !done) group(16, warmup);
    }
    if (count != head && !done) group((int)(count - head), false);
}

// --------------------------------------------------------------------------------------------
// Riemersma dither, one LANE per run (round 5)
//
// The run-parallel form above spends a whole wavefront on one chain: 64 lanes share the palette search of one pixel, ~85
// instructions per step.  With runs plentiful (the chain's state is its last sixteen choices: any number of runs can be walked
// side by side and verified afterwards) the other layout pays: every LANE walks its own run, the error queue in its registers,
// the nearest colour through the exact-pruning records of the NN map (k_nn_lut_build over the weighted palette: a handful of
// candidates per query instead of K), ~450 instructions per step of a WAVEFRONT = 64 steps.  A run is a few hundred pixels,
// the warm-up as long again -- twice the steps, a twelfth of the instructions each.
//
// Layout: run b covers the ranks (positions along the curve among the in-image pixels) [t_b, t_b+1), t_b = N b / S.  Lanes of a
// wavefront walk 64 consecutive runs in lock step, so the pixels are stored TRANSPOSED: position p of run b at
// ((b / 64) Lmax + p) 64 + b % 64 -- the 64 lanes' loads of one step are 512 consecutive bytes per plane (first version: every
// lane its own cache line, 320 line requests per step, and the kernel was bound by exactly that).  The choices are stored the
// same way.  The warm-up of run b is the last `warm` pixels of run b - 1: the neighbouring column, no copy (warm <= run length).
//   k_dither_order    rank -> linear pixel number: the only place that knows the Hilbert curve
//   k_dither_streams  pixels into the transposed layout (tiles of 64 runs x 64 positions through LDS)
//   k_dither_lanes<0> run b: zero queue `warm` pixels before t_b, the choices of the last sixteen warm-up steps -> side[b], of the run -> smap
//   k_dither_lane_check  boundary b is good iff side[b] equals the last sixteen choices of run b - 1; the others are listed
//   k_dither_lanes<1> a listed run again from the queue rebuilt out of those choices and pixels (side[b] := what it read), until
//                     sixteen consecutive choices equal what is there (the old chain met: the rest stands)
//   ... check / repair until a check lists nothing: nothing was written since the previous repair, every side[b] equals the
//       map, and by induction over b (run 0 starts from the true zero queue) smap is the reference's chain
//   k_dither_unpermute  out[pixel number of rank t_b + p] = smap(b, p)
// Arithmetic per step exactly as riemersma.c:275-341 orders it: e = ((0 + q[0] w[0]) + q[1] w[1]) + ... per channel, query =
// W (pixel + e) with the double weights, squared distance ((d0 d0 + d1 d1) + d2 d2) to the palette scaled by the float-cast
// weights, ascending index with strict '<', pushed error = pixel - chosen colour.
// --------------------------------------------------------------------------------------------
struct DitherRuns {                      // the cut of the curve and the transposed layout built on it
    unsigned long long N;
    unsigned S, Lmax;
    __host__ __device__ unsigned long long t(const unsigned b) const { return N * b / S; }
    __host__ __device__ size_t idx(const unsigned b, const unsigned p) const { return ((size_t)(b >> 6) * Lmax + p) * 64 + (b & 63u); }
};

__global__ __launch_bounds__(64) void k_dither_order(unsigned width, unsigned height, unsigned parts, unsigned *__restrict__ spos) {
    const unsigned mxd = width > height ? width : height;
    int L = 0;
    while ((1u << L) < mxd) L++;
    const int lane = threadIdx.x;
    const unsigned long long total = 1ULL << (2 * L), npix = (unsigned long long)width * height;
    const unsigned g = blockIdx.x;
    unsigned long long d0 = 0, c = 0, d_end = total, cc;
    if (g > 0) dither_locate(L, width, height, npix * g / parts, d0, c);
    if (g + 1 < parts) dither_locate(L, width, height, npix * (g + 1) / parts, d_end, cc);
    const bool needs_skip = width < (1u << L) || height < (1u << L);
    while (d0 < d_end) {
        if (needs_skip) {                                            // whole out-of-image aligned sub-squares
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
        unsigned x, y;
        hilbert_d2xy(L, d0 + (unsigned long long)lane, x, y);
        const bool inb = x < width && y < height;
        const unsigned long long mask = __ballot(inb);
        if (inb) spos[c + (unsigned long long)__popcll(mask & ((1ULL << lane) - 1ULL))] = y * width + x;
        c += (unsigned long long)__popcll(mask);
        d0 += 64;
    }
}

// tile (pt, w): runs 8 w .. 8 w + 7, positions 256 pt .. 256 pt + 255, the three planes together.  Read side: 256 consecutive ranks of
// a run are a 16 x 16 square of the image when the run starts on a multiple of 256 (power-of-two images: always) -- sixteen whole
// 128-byte lines per plane; with tiles of 16 ranks per run every line was fetched twice over (FETCH_SIZE 8.6 GB for 3.5 GB of
// pixels).  Write side: eight consecutive runs = 64 bytes.
// WHICH: the conversion into linear Rec2020 the pixels still need (patolette.c:268-299; PAMD_COPY = none), done on the way -- the
// same device routine k_convert applies, so the same bits, without a pass of its own over the image
template <int WHICH>
__global__ __launch_bounds__(256) void k_dither_streams(const double *__restrict__ img, size_t plane_stride, const unsigned *__restrict__ spos, DitherRuns R,
                                                       double *__restrict__ sx, double *__restrict__ sy, double *__restrict__ sz) {
    __shared__ unsigned long long t0[9];
    __shared__ double tile[3][8][257];
    const unsigned w = blockIdx.y, p0 = blockIdx.x * 256u, tid = threadIdx.x;
    if (tid < 9) { const unsigned b = 8u * w + tid; t0[tid] = b <= R.S ? R.t(b) : R.N; }
    if constexpr (WHICH != PAMD_COPY) pow_tables_to_lds();
    __syncthreads();
    unsigned pix[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {                                    // run i of the tile: consecutive threads = consecutive ranks
        const unsigned p = p0 + tid;
        const bool ok = 8u * w + i < R.S && t0[i] + p < t0[i + 1];
        pix[i] = ok ? spos[t0[i] + p] : 0xFFFFFFFFu;
    }
    double v[8][3];
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) v[i][c] = pix[i] != 0xFFFFFFFFu ? img[(size_t)c * plane_stride + pix[i]] : 0.0;
    if constexpr (WHICH != PAMD_COPY) {
#pragma unroll
        for (int i = 0; i < 8; i++) if (pix[i] != 0xFFFFFFFFu) dev_convert<WHICH>(v[i]);
    }
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) tile[c][i][tid] = v[i][c];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; i++) {                                    // consecutive threads: the eight runs of one position, then the next position
        const unsigned e = tid + 256u * i, pp = e >> 3, l = e & 7u;
        if (p0 + pp < R.Lmax) {
            const size_t at = R.idx(8u * w + l, p0 + pp);
            sx[at] = tile[0][l][pp]; sy[at] = tile[1][l][pp]; sz[at] = tile[2][l][pp];
        }
    }
}

// tile (pt, w): runs 64 w .. 64 w + 63, positions 64 pt .. 64 pt + 63 of the choices, back to the image's pixel order
template <typename OutT>
__global__ __launch_bounds__(256) void k_dither_unpermute(const unsigned char *__restrict__ smap, const unsigned *__restrict__ spos, DitherRuns R, OutT *__restrict__ out) {
    __shared__ unsigned long long t0[65];
    __shared__ unsigned char sm[64][65];
    const unsigned w = blockIdx.y, p0 = blockIdx.x * 64u, tid = threadIdx.x;
    if (tid < 65) { const unsigned b = 64u * w + tid; t0[tid] = b <= R.S ? R.t(b) : R.N; }
#pragma unroll
    for (int i = 0; i < 16; i++) {                                   // consecutive threads: consecutive runs = consecutive bytes
        const unsigned e = tid + 256u * i, pp = e >> 6, l = e & 63u;
        sm[l][pp] = p0 + pp < R.Lmax ? smap[R.idx(64u * w + l, p0 + pp)] : (unsigned char)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {                                   // consecutive threads: consecutive ranks of one run
        const unsigned e = tid + 256u * i, l = e >> 6, p = p0 + (e & 63u);
        if (64u * w + l < R.S && t0[l] + p < t0[l + 1]) out[spos[t0[l] + p]] = (OutT)sm[l][e & 63u];
    }
}

struct DitherLanes {
    DitherRuns R;
    const double *sx, *sy, *sz;          // pixels, transposed layout
    unsigned char *smap;                 // choices, transposed layout
    unsigned short *side;                // [S][16] the choices a run's starting queue was built from (0xFFFF = none: the run is walked again)
    unsigned *list;                      // runs whose boundary check failed; list[-1] = how many; list[-2] counts the periodic jumps taken (a statistic)
    unsigned char *flag;                 // [S + 1] the same per run: 1 = listed by the last check
    int solo;                            // k_dither_lane_repair: one wavefront alone, from the lowest listed boundary through everything in its way
    unsigned warm;                       // <= the shortest run
    const unsigned char *lut, *lut2;     // 16-byte records of the G^3 grid over the weighted palette, and their continuations (k_nn_lut_build)
    NNGrid g;
    double hi[3];                        // upper corner of the grid
    // the same over two wider grids for the queries outside [lo, hi]: the palette's box with 1.5 extents around it (as many cells as
    // the first grid), then with 4 extents around it (32^3 cells); queries outside that too take all k entries
    struct Wider { NNGrid g; double hi[3]; const unsigned char *lut, *lut2; } wide[2];
    float amb;                           // f32 first pass: first and second must differ by more than this
    float amb_p, amb_x;                  // ... for a query outside the grid: amb_p + amb_x |x - lo|^2
};

__global__ __launch_bounds__(256) void k_dither_lane_check(DitherLanes a) {
    const unsigned b = blockIdx.x * blockDim.x + threadIdx.x + 1u;
    if (b >= a.R.S) return;
    const unsigned lp = (unsigned)(a.R.t(b) - a.R.t(b - 1));        // length of run b - 1
    bool same = true;
    for (unsigned i = 0; i < 16; i++) same = same && a.side[16ull * b + i] == (unsigned short)a.smap[a.R.idx(b - 1, lp - 16 + i)];
    a.flag[b] = same ? 0 : 1;                                       // (flag[0] and flag[S] stay 0)
    if (!same) a.list[atomicAdd(a.list - 1, 1u)] = b;
}

// Every palette entry for the lanes that `want` it (a query outside both grids, a cell with more than thirty candidates), through
// the f32 records of k_dither_lanes' LDS: the WAVEFRONT scans for each such lane in turn -- lane L takes entries L, L + 64,
// L + 128, L + 192 against that lane's query, then the smallest and the second smallest of the 64 pairs (the smallest's owner
// contributes its second) -- and the f64 loop over all k decides where the two lie within the margin.  Called by whole
// wavefronts; not inlined on purpose.
__device__ __attribute__((noinline)) int dither_nearest_all(const double x, const double y, const double z, const float xf, const float yf, const float zf,
                                                             const bool want, const float amb, const int k) {
    extern __shared__ double lds[];
    const double *pwt = lds + 3 * k;
    const float4 *r32 = reinterpret_cast<const float4 *>(lds + 6 * k);
    const int ln = (int)(threadIdx.x & 63u);
    float m1 = INFINITY, m2 = INFINITY; int best = 0;
    unsigned long long todo = __ballot(want);
    while (todo) {
        const int l = (int)__builtin_ctzll(todo);
        todo &= todo - 1ULL;
        const float qx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xf), l));
        const float qy = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, yf), l));
        const float qz = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zf), l));
        float a1 = INFINITY, a2 = INFINITY; int ab = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int j = ln + 64 * q;
            const float4 r = r32[j < k ? j : k - 1];
            float tj = __builtin_fmaf(qx, r.x, __builtin_fmaf(qy, r.y, __builtin_fmaf(qz, r.z, r.w)));
            tj = j < k ? tj : INFINITY;
            a2 = __builtin_fminf(a2, __builtin_fmaxf(a1, tj));
            ab = tj < a1 ? j : ab;
            a1 = __builtin_fminf(a1, tj);
        }
        float g1 = a1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) g1 = __builtin_fminf(g1, __shfl_xor(g1, o, 64));
        const unsigned long long wm = __ballot(a1 == g1);
        const int who = wm ? (int)__builtin_ctzll(wm) : 0;           // (nobody: a NaN query -- the exact loop takes it)
        float g2 = ln == who ? a2 : a1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) g2 = __builtin_fminf(g2, __shfl_xor(g2, o, 64));
        const int gb = __builtin_amdgcn_readlane(ab, who);
        if (ln == l) { m1 = wm ? g1 : NAN; m2 = g2; best = gb; }
    }
    if (want && !((m2 - m1) > amb)) {
        double bd = INFINITY; best = 0;
        for (int j = 0; j < k; j++) {
            const double d0 = x - pwt[j], d1 = y - pwt[k + j], d2 = z - pwt[2 * k + j];
            const double d = (d0 * d0 + d1 * d1) + d2 * d2;
            if (d < bd) { bd = d; best = j; }                       // ascending j + strict '<' = lowest index on ties
        }
    }
    return best;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_dither_lanes(DitherLanes a, const double *__restrict__ pal /* planar (k,3), linear Rec2020 */, int k, DitherWeights wts) {
    extern __shared__ double lds[];
    double *praw = lds, *pwt = lds + 3 * k;                        // [3][k] raw palette; [3][k] scaled by the (float)-cast weights (riemersma.c:419-425)
    float4 *r32 = reinterpret_cast<float4 *>(lds + 6 * k);          // [k] {-2 (p - lo), |p - lo|^2} of the weighted palette in f32
    const double fw[3] = {(double)(float)kRw, (double)(float)kGw, (double)(float)kBw};
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
        double sh[3];
        for (int c = 0; c < 3; c++) { const double v = pal[c * k + j]; praw[c * k + j] = v; pwt[c * k + j] = v * fw[c]; sh[c] = v * fw[c] - a.g.lo[c]; }
        r32[j] = make_float4((float)(-2.0 * sh[0]), (float)(-2.0 * sh[1]), (float)(-2.0 * sh[2]), (float)((sh[0] * sh[0] + sh[1] * sh[1]) + sh[2] * sh[2]));
    }
    __syncthreads();
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    bool active;
    unsigned b = 0;
    if constexpr (MODE == 0) { active = gid < a.R.S; b = active ? gid : 0u; }
    else { const unsigned nl = a.list[-1]; active = gid < nl; b = active ? a.list[gid] : 1u; }
    const unsigned len = (unsigned)(a.R.t(b + 1) - a.R.t(b));       // this run
    const unsigned lp = b > 0 ? (unsigned)(a.R.t(b) - a.R.t(b - 1)) : 0u;   // the one before it
    // steps [0, wu) read the end of run b - 1 (the neighbouring column), steps [wu, wu + len) this run
    const unsigned wu = MODE == 0 ? (b > 0 ? a.warm : 0u) : 0u;
    const size_t own = a.R.idx(b, 0);
    size_t at = wu ? a.R.idx(b - 1, lp - wu) : own;                // where step 0 reads
    if constexpr (MODE == 0) {
        if (active && b > 0 && wu < 16) {                           // a warm-up of fewer than sixteen steps leaves no record to pass the check
            for (int i = 0; i < 16; i++) a.side[16ull * b + i] = 0xFFFFu;
        }
    }
#define PAMD_DL_EACH(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#define PAMD_DL_DECL(S) double q0_##S = 0.0, q1_##S = 0.0, q2_##S = 0.0;
    PAMD_DL_EACH(PAMD_DL_DECL)                                      // the error queue: slot s = the error of the run's step s (mod 16); named scalars: dither_slots.h
#undef PAMD_DL_DECL
    if constexpr (MODE == 1) {
        if (active) {
            // the queue as the chain holds it after the last sixteen pixels of run b - 1: original pixel - chosen colour, oldest first
#define PAMD_DL_INIT(S)                                                                                                   \
            {                                                                                                             \
                const size_t rr = a.R.idx(b - 1, lp - 16 + S);                                                            \
                const unsigned c = a.smap[rr];                                                                            \
                a.side[16ull * b + S] = (unsigned short)c;                                                                \
                q0_##S = a.sx[rr] - praw[c]; q1_##S = a.sy[rr] - praw[k + c]; q2_##S = a.sz[rr] - praw[2 * k + c];        \
            }
            PAMD_DL_EACH(PAMD_DL_INIT)
#undef PAMD_DL_INIT
        }
    }
    const int G = a.g.G;
    const double lo0 = a.g.lo[0], lo1 = a.g.lo[1], lo2 = a.g.lo[2], in0 = a.g.inv[0], in1 = a.g.inv[1], in2 = a.g.inv[2];
    const double hi0 = a.hi[0], hi1 = a.hi[1], hi2 = a.hi[2];
    // Nearest colour.  First pass in f32 over the cell's candidates, as in k_nn_map_mid: t_j = |x - p_j|^2 - |x - lo|^2 from the
    // shifted records {-2 (p - lo), |p - lo|^2} (three fma), smallest and second smallest tracked; the smallest names the winner of
    // the f64 expression whenever the second lies more than a.amb above it (the bound of map.hip's table comment: the query is
    // inside the grid here, the palette's |p - lo|^2 is bounded by the host).  Otherwise -- about one query in a thousand, and
    // every query outside the grid or in a cell whose list runs past one record -- the exact f64 loop decides.
#ifdef PAMD_NN_STATS
    unsigned ws_exact = 0, ws_full = 0, ws_trips = 0, ws_long = 0;  // (lane 0 of every wavefront counts its steps with an exact pass / a full scan)
    const unsigned long long ws_t0 = wall_clock64();
#endif
    // the request: which grid, which cell, and the load of its record -- no branch, so that what follows it in the step is
    // scheduled under the load (lanes that are off or outside both grids read the record of a clamped cell and ignore it)
    struct NearestRec { uint4 rec; size_t cell; const unsigned char *tab2; bool inside, outer; };
    auto nearest_request = [&](const double x, const double y, const double z, const bool on) -> NearestRec {
        NearestRec r;
#ifdef PAMD_KM_TRACE
        r.inside = (g_nn_flags & 16u) ? true : (x >= lo0 && x <= hi0 && y >= lo1 && y <= hi1 && z >= lo2 && z <= hi2);   // (16: timing experiment, wrong map: nobody is outside)
#else
        r.inside = x >= lo0 && x <= hi0 && y >= lo1 && y <= hi1 && z >= lo2 && z <= hi2;   // (a NaN: not inside)
#endif
        const double cx = on ? x : lo0, cy = on ? y : lo1, cz = on ? z : lo2;         // (a lane that is off may hold anything: cell 0)
        r.cell = nn_cell(cx, cy, cz, G, lo0, lo1, lo2, in0, in1, in2);               // (clamped into the grid: a record, whatever the query)
        r.outer = false;
        r.tab2 = a.lut2;
        const unsigned char *tab = a.lut;
        // Error diffusion carries the queries far beyond the palette's hull wherever the image is (the queue's weights sum to 5.6:
        // a colour the palette misses by d is asked for 5.6 d further out): those find their candidates in the outer grid.  A
        // branch of the wavefront: on most content most steps have nobody outside.
        if (__any(on && !r.inside)) {
#pragma unroll
            for (int lv = 0; lv < 2; lv++) {
                const DitherLanes::Wider &wd = a.wide[lv];
                const bool here = on && !r.inside && !r.outer && x >= wd.g.lo[0] && x <= wd.hi[0] && y >= wd.g.lo[1] && y <= wd.hi[1] && z >= wd.g.lo[2] && z <= wd.hi[2];
                if (here) {
                    r.cell = nn_cell(x, y, z, wd.g.G, wd.g.lo[0], wd.g.lo[1], wd.g.lo[2], wd.g.inv[0], wd.g.inv[1], wd.g.inv[2]);
                    tab = wd.lut; r.tab2 = wd.lut2; r.outer = true;
                }
            }
        }
#ifdef PAMD_KM_TRACE
        if (g_nn_flags & 4u) r.cell = ((size_t)(threadIdx.x & 63) * 4099u) % (size_t)32768;   // timing experiment (wrong map): the record does not depend on the query
#endif
        r.rec = *reinterpret_cast<const uint4 *>(tab + r.cell * 16);
        return r;
    };
    auto nearest = [&](const double x, const double y, const double z, const bool on, const NearestRec nr) -> int {
        const bool inside = nr.inside, outer = nr.outer;
        int cnt = 255;
        unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0;          // the cell's candidates, a byte each, the first in s0's low byte
        if (on && (inside || outer)) {
            const uint4 rec = nr.rec;
            const size_t cell = nr.cell;
            const unsigned char *const t2 = nr.tab2;
            cnt = (int)(rec.x & 0xffu);
            const unsigned long long w0 = ((unsigned long long)rec.y << 32) | rec.x, w1 = ((unsigned long long)rec.w << 32) | rec.z;
            s0 = (w0 >> 8) | (w1 << 56); s1 = w1 >> 8;              // fifteen entries; past a short list its last one repeats (k_nn_lut_build)
            if (cnt > 15 && cnt != 255) {                           // a crowded cell: entries 15 .. 29 in the second record
                const uint4 r2 = *reinterpret_cast<const uint4 *>(t2 + cell * 16);
                const unsigned long long x0 = ((unsigned long long)r2.y << 32) | r2.x, x1 = ((unsigned long long)r2.w << 32) | r2.z;
                s1 |= x0 << 56; s2 = (x0 >> 8) | (x1 << 56); s3 = x1 >> 8;
            }
        }
        const bool listed = cnt != 255;                             // else: outside both grids, or a cell with more than thirty candidates: all k entries
        const int n = listed ? cnt : 0;
        const float xf = (float)(x - lo0), yf = (float)(y - lo1), zf = (float)(z - lo2);
        // the margin of the first pass: the bound holds for any query with its own |x - lo|^2 in the place of |hi - lo|^2
        const float amb = inside ? a.amb : __builtin_fmaf(a.amb_x, __builtin_fmaf(xf, xf, __builtin_fmaf(yf, yf, zf * zf)), a.amb_p);
        float m1 = INFINITY, m2 = INFINITY; int best = 0;
        {
            // four candidates a trip: their records are requested from LDS together, then evaluated in list order
            unsigned long long v0 = s0, v1 = s1, v2 = s2, v3 = s3;
            for (int t = 0; __any(t < n); t += 4) {
                const unsigned four = (unsigned)v0;
                v0 = (v0 >> 32) | (v1 << 32); v1 = (v1 >> 32) | (v2 << 32); v2 = (v2 >> 32) | (v3 << 32); v3 >>= 32;
                int jj[4]; float4 r[4];
#pragma unroll
                for (int i = 0; i < 4; i++) { jj[i] = t + i < n ? (int)((four >> (8 * i)) & 0xffu) : 0; r[i] = r32[jj[i]]; }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    float tj = __builtin_fmaf(xf, r[i].x, __builtin_fmaf(yf, r[i].y, __builtin_fmaf(zf, r[i].z, r[i].w)));
                    tj = t + i < n ? tj : INFINITY;
                    m2 = __builtin_fminf(m2, __builtin_fmaxf(m1, tj));
                    best = tj < m1 ? jj[i] : best;
                    m1 = __builtin_fminf(m1, tj);
                }
            }
        }
#ifdef PAMD_KM_TRACE
        const bool exact = (g_nn_flags & 8u) ? false : (on && listed && !((m2 - m1) > amb));    // (8: timing experiment, wrong map)
#else
        const bool exact = on && listed && !((m2 - m1) > amb);      // (a NaN anywhere: exact)
#endif
#ifdef PAMD_NN_STATS
        {
            const unsigned long long m_on = __ballot(on), m_out = __ballot(on && !listed), m_long = __ballot(on && (inside || outer) && cnt > 15), m_amb = __ballot(exact);
            const unsigned long long m_ex = __ballot(exact), m_full = __ballot(on && !listed);
            int nmax = n;
            for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
            if ((threadIdx.x & 63) == 0 && m_on) {
                atomicAdd(&g_dl_stats[0], (unsigned long long)__popcll(m_on)); atomicAdd(&g_dl_stats[1], (unsigned long long)__popcll(m_out));
                atomicAdd(&g_dl_stats[2], (unsigned long long)__popcll(m_long)); atomicAdd(&g_dl_stats[3], (unsigned long long)__popcll(m_amb));
                atomicAdd(&g_dl_stats[4], 1ULL);
                if (m_ex) { atomicAdd(&g_dl_stats[5], 1ULL); ws_exact++; }
                if (m_full) { atomicAdd(&g_dl_stats[6], 1ULL); ws_full++; }
                atomicAdd(&g_dl_stats[7], (unsigned long long)nmax);
                ws_trips += (unsigned)nmax; if (m_out) ws_long++;
            }
        }
#endif
        if (__any(exact)) {
            // The f64 expression decides, among the CONTENDERS only: the true nearest entry (and every entry that ties with it) lies
            // within 2 E of the f32 minimum, so the first pass is run over the list once more and only the entries within the margin of
            // its minimum are measured in f64.  In list order with strict '<': the lowest index on ties.
            if (exact) {
                double bd = INFINITY; best = 0;
                const float lim = m1 + amb;
                unsigned long long v0 = s0, v1 = s1, v2 = s2, v3 = s3;
                for (int t = 0; t < n; t++) {
                    const int j = (int)(v0 & 0xffULL);
                    v0 = (v0 >> 8) | (v1 << 56); v1 = (v1 >> 8) | (v2 << 56); v2 = (v2 >> 8) | (v3 << 56); v3 >>= 8;
                    const float4 r = r32[j];
                    const float tj = __builtin_fmaf(xf, r.x, __builtin_fmaf(yf, r.y, __builtin_fmaf(zf, r.z, r.w)));
                    if (tj <= lim) {
                        const double d0 = x - pwt[j], d1 = y - pwt[k + j], d2 = z - pwt[2 * k + j];
                        const double d = (d0 * d0 + d1 * d1) + d2 * d2;
                        if (d < bd) { bd = d; best = j; }
                    }
                }
            }
        }
        if (__any(on && !listed)) {                                 // (a call: rare, and the step's code stays small)
            const int ba = dither_nearest_all(x, y, z, xf, yf, zf, on && !listed, amb, k);
            best = on && !listed ? ba : best;
        }
        return best;
    };
    const unsigned nsteps = wu + len;
    unsigned st = 0;                                                // step number
    double c0 = 0, c1 = 0, c2 = 0;                                  // the pixel of step st, fetched one step ahead
    int cm = 0;                                                     // MODE 1: and the choice smap holds for it
    bool on = active && nsteps > 0;
    if (on) { c0 = a.sx[at]; c1 = a.sy[at]; c2 = a.sz[at]; if constexpr (MODE == 1) cm = (int)a.smap[at]; }
    int streak = 0;                                                 // MODE 1: consecutive choices equal to what smap holds
    // Sixteen steps a trip: the queue's slots are registers, the step that starts at slot j reads them in the order j, j + 1, ...
    // (dither_slots.h; with j a constant of the unrolled loop the macros' switches fold away).  One copy of the step selected by a
    // scalar branch on the slot was measured 10 % slower (a lone wavefront pays for every taken branch with a refill of its
    // instruction buffer), although it is a quarter of the code.
    // The step is software-pipelined by hand: the first fifteen terms of the NEXT step's sums (they do not depend on this step's
    // choice) are computed between the request for the cell's record and its first use.
    double e0 = 0.0, e1 = 0.0, e2 = 0.0;                            // riemersma.c:282-296: the sums of the step at hand, all but the last term
    PAMD_DL_QUEUE_PARTIAL(0)
    double l0 = q0_15, l1 = q1_15, l2 = q2_15;                      // the error of the step before (slot 15 of the starting queue)
    while (__any(on)) {
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const double p0 = c0, p1 = c1, p2 = c2;
        const int was = cm;
        const size_t cur = at;
        const bool nxt = on && st + 1 < nsteps;
        at = (st + 1 == wu) ? own : at + 64;                        // the run starts where the neighbour's column ends
        if (nxt) { c0 = a.sx[at]; c1 = a.sy[at]; c2 = a.sz[at]; if constexpr (MODE == 1) cm = (int)a.smap[at]; }
        e0 += l0 * wts.w[15]; e1 += l1 * wts.w[15]; e2 += l2 * wts.w[15];
        const double x = kRw * (p0 + e0), y = kGw * (p1 + e1), z = kBw * (p2 + e2);
        const NearestRec nr = nearest_request(x, y, z, on);
        e0 = 0.0; e1 = 0.0; e2 = 0.0;
        const int jn = (j + 1) & 15;
        PAMD_DL_QUEUE_PARTIAL(jn)
        const int bi = nearest(x, y, z, on, nr);
        if (on) {
            l0 = p0 - praw[bi]; l1 = p1 - praw[k + bi]; l2 = p2 - praw[2 * k + bi];       // riemersma.c:333-340
            PAMD_DL_QUEUE_PUSH(j, l0, l1, l2)
            if constexpr (MODE == 0) {
                if (st >= wu) a.smap[cur] = (unsigned char)bi;
                else if (st + 16 >= wu) a.side[16ull * b + (st + 16 - wu)] = (unsigned short)bi;
            } else {
                if (was == bi) streak++;
                else { streak = 0; a.smap[cur] = (unsigned char)bi; }
            }
        }
        st++;
        on = nxt && (MODE == 0 || streak < 16);
      }
    }
#undef PAMD_DL_EACH
#ifdef PAMD_NN_STATS
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&g_nn_stats[6], (unsigned long long)ws_exact); atomicMax(&g_nn_stats[7], (unsigned long long)ws_full);
        const unsigned wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        if (MODE == 0 && wv < 4096u) { g_dl_wave[wv][0] = wall_clock64() - ws_t0; g_dl_wave[wv][1] = ws_exact; g_dl_wave[wv][2] = ws_trips; g_dl_wave[wv][3] = ws_long; }
    }
#endif
}

// A listed run again, by ONE WAVEFRONT (the repair passes of the lane layout).  A pass waits for its slowest run, a lane's step is
// ~2 us of dependent loads and a pass listed a few hundred runs: 64 of them side by side in a wavefront gain nothing, each on a
// wavefront of its own takes k_dither's step (the palette search across the 64 lanes, no table: ~0.45 us).  Pixels and choices
// in the transposed layout (one cache line per lane and load: a few thousand runs, it does not matter here); the queue rebuilt
// from the sixteen choices before the run as k_dither<.., 1> does; stops when a whole group of sixteen equals what is there.
template <int PER>
__global__ __launch_bounds__(64) void k_dither_lane_repair(DitherLanes a, const double *__restrict__ pal /* planar (k,3), linear Rec2020 */, int k, DitherWeights wts) {
    extern __shared__ double lds[];
    constexpr int kRing = 128;                                       // >= 64 + 15 pending pixels
    double *praw = lds, *pwt = lds + 3 * k;                          // [3][k] raw palette; scaled by the (float)-cast weights (riemersma.c:419-425)
    double *rpx = lds + 6 * k;                                       // [3][kRing] channels of the pending pixels
    unsigned int *rpos = reinterpret_cast<unsigned int *>(rpx + 3 * kRing);   // [kRing] their places in the transposed layout
    const int lane = threadIdx.x;
    if (blockIdx.x >= a.list[-1]) return;
    const bool solo = a.solo != 0;                                   // (launched as ONE block then)
    unsigned b = a.list[blockIdx.x];
    if (solo) {                                                      // the lowest listed boundary (the list is in no order)
        b = 0;
        for (unsigned q0 = 1; q0 < a.R.S && !b; q0 += 64) {
            const unsigned q = q0 + (unsigned)threadIdx.x;
            const unsigned long long m = __ballot(q < a.R.S && a.flag[q] != 0);
            if (m) b = q0 + (unsigned)__builtin_ctzll(m);
        }
        if (!b) return;
    }
    const double fw[3] = {(double)(float)kRw, (double)(float)kGw, (double)(float)kBw};
    for (int j = lane; j < k; j += 64)
        for (int c = 0; c < 3; c++) { const double v = pal[c * k + j]; praw[c * k + j] = v; pwt[c * k + j] = v * fw[c]; }
    __syncthreads();
    double ex[PER], ey[PER], ez[PER];                                // the lane's own entries [lane PER, (lane + 1) PER)
#pragma unroll
    for (int m = 0; m < PER; m++) {
        const int j = lane * PER + m;
        ex[m] = j < k ? pwt[j] : 1e300; ey[m] = j < k ? pwt[k + j] : 1e300; ez[m] = j < k ? pwt[2 * k + j] : 1e300;
    }
    const int ch = lane < 48 ? lane >> 4 : 2, dph = lane & 15;       // as k_dither: lane (c, d) sums channel c of the step d (mod 16)
    const double Wc = ch == 0 ? kRw : (ch == 1 ? kGw : kBw);
    double wl[16], keep[16];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        double v = 0;
#pragma unroll
        for (int i = 0; i < 16; i++) v = (15 - ((dph - j - 1) & 15)) == i ? wts.w[i] : v;
        wl[j] = v;
        keep[j] = dph == j ? 0.0 : 1.0;
    }
    double acc = 0.0;
    const double *prw = praw + ch * k;
    auto nearest = [&](const double qx, const double qy, const double qz) -> int {     // k_dither's, PER > 0
        double dd[PER];
#pragma unroll
        for (int m = 0; m < PER; m++) {
            const double e0 = qx - ex[m], e1 = qy - ey[m], e2 = qz - ez[m];
            dd[m] = (e0 * e0 + e1 * e1) + e2 * e2;
        }
        double bd = dd[0];
#pragma unroll
        for (int m = 1; m < PER; m++) bd = fmin(bd, dd[m]);
        int bj;
        if constexpr (PER == 4 || PER == 2) {
            int e;
            unsigned t;
            if constexpr (PER == 4) t = dither_rows4(dd[0], dd[1], dd[2], dd[3], bd, e);
            else t = dither_rows2(dd[0], bd, e);
            bj = e | (lane * PER);
            const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)t, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)t, 16),
                           r2 = (unsigned)__builtin_amdgcn_readlane((int)t, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)t, 48);
            unsigned m23, mh;
            asm("s_min_u32 %0, %2, %3\n\ts_min_u32 %1, %4, %5\n\ts_min_u32 %0, %0, %1" : "=&s"(mh), "=&s"(m23) : "s"(r0), "s"(r1), "s"(r2), "s"(r3));
            const unsigned hi = (unsigned)__double2hiint(bd), lo = (unsigned)__double2loint(bd);
            unsigned long long c = __ballot(hi == mh);
            if (__popcll(c) != 1) {
                const unsigned ml = wave_min_u32(hi == mh ? lo : 0xFFFFFFFFu);
                c = __ballot(hi == mh && lo == ml);
            }
            return __builtin_amdgcn_readlane(bj, (int)__builtin_ctzll(c));
        } else {
            bj = PER - 1;
#pragma unroll
            for (int m = PER - 2; m >= 0; m--) bj = dd[m] == bd ? m : bj;
            bj += lane * PER;
            return __builtin_amdgcn_readlane(bj, wave_argmin_nonneg_f64(bd));
        }
    };
    // solo (the passes have stalled: bands of one colour with many failing boundaries each -- every row's wavefront stops at the next
    // row's head, which started from a tail that has just been rewritten: one row per band and pass): this wavefront alone, from the
    // lowest failing boundary through everything in its way, rows and heads alike, until it meets what is there behind a boundary
    // that holds.  Nobody else writes, so the rules that keep two wavefronts off one run do not apply.
    // A boundary that fails right behind another failing one is not this wavefront's: the first of such a row (its head) walks
    // through all of them, one after the other -- each starts from what its predecessor ends with, nothing to gain side by side.
    if (!solo && a.flag[b - 1]) return;                             // (b >= 1: run 0 is never listed)
    constexpr unsigned kHist = 2048;                                // choices remembered (a power of two)
    unsigned char *hist = reinterpret_cast<unsigned char *>(rpos + kRing);   // [kHist] the choices of steps T - kHist .. T - 1, step s at s mod kHist
    unsigned T = 0;                                                 // steps walked since the history was started
    unsigned flat_from = 0;                                         // steps flat_from .. T - 1 were made on pixels of ONE colour (cflat: this lane's channel of it)
    double cflat = 0.0;
    unsigned head = 16, count = 16;
    unsigned tail = 0;                                              // lanes 0 .. 15: the last sixteen choices made, oldest first
    bool done = false;
    auto group = [&](const int limit) {
        double pcs[16];
#pragma unroll
        for (int j = 0; j < 16; j++) pcs[j] = rpx[ch * kRing + ((head + (unsigned)j) & (kRing - 1))];
        unsigned was = 0, wpos = 0;
        if (lane < limit) { wpos = rpos[(head + (unsigned)lane) & (kRing - 1)]; was = (unsigned)a.smap[wpos]; }
        int res = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (j < limit) {                                         // wave-uniform
                const double qv = Wc * (pcs[j] + acc);
                const int bi = nearest(readlane_f64(qv, j), readlane_f64(qv, 16 + j), readlane_f64(qv, 32 + j));
                res = lane == j ? bi : res;
                const double err = pcs[j] - prw[bi];
                acc = __builtin_fma(acc, keep[j], err * wl[j]);
            }
        }
        if (lane < limit && was != (unsigned)res) a.smap[wpos] = (unsigned char)res;
        if (limit == 16 && __all(lane >= 16 || was == (unsigned)res)) done = true;     // the old chain is met: the rest of the run stands
        // the window of the last sixteen choices moves on by `limit`
        const unsigned moved = (unsigned)__shfl((int)tail, (lane + limit) & 63, 64), fresh = (unsigned)__shfl(res, (lane - (16 - limit)) & 63, 64);
        tail = lane < 16 - limit ? moved : fresh;
        // the history, and how long the pixels have been of one colour
        if (lane < limit) hist[(T + (unsigned)lane) & (kHist - 1)] = (unsigned char)res;
        const double mine = rpx[ch * kRing + ((head + (unsigned)dph) & (kRing - 1))];
        const bool same = T > flat_from && __all(lane >= 48 || dph >= limit || mine == cflat);
        if (!same) {                                                 // the stretch starts again at this group's last pixel
            cflat = rpx[ch * kRing + ((head + (unsigned)limit - 1u) & (kRing - 1))];
            flat_from = T + (unsigned)limit - 1u;
        }
        T += (unsigned)limit;
        head += (unsigned)limit;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // the queue as the chain holds it after the sixteen pixels in ring slots head - 16 .. head - 1 with the choices `c16` (lane d
    // of every quarter: the choice of pixel d): their error vectors pushed in order into zero sums -- lane (c, d) restarts at step
    // d, so whatever it summed before never reaches a pixel.
    auto requeue = [&](const unsigned c16) {
        const double ev = rpx[ch * kRing + ((head - 16u + (unsigned)dph) & (kRing - 1))] - prw[c16];
        acc = 0.0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double e = __shfl(ev, (lane & 48) | j, 64);
            acc = __builtin_fma(acc, keep[j], e * wl[j]);
        }
    };
    constexpr unsigned kMaxWalk = 64;                               // runs one wavefront WALKS through in a pass (the rest: next pass)
    bool in_row = true;                                             // still among the failing boundaries this one heads
    bool halt = false;                                              // a jump ended at a boundary this wavefront must not cross
    unsigned r = b, walked = 0;
    unsigned len = 0, base = 0;                                     // run r's length; the run's position that ring slot 16 stands for

    // A FLAT stretch (pixels of one colour that is not a palette entry) makes the chain periodic, out of step with whatever was
    // speculated there: nothing is ever met, and a walk is a pixel per 0.45 us.  But once the last sixteen choices equal the
    // sixteen P steps earlier -- all of it on pixels of the one colour -- the chain's state equals its state P steps earlier, and
    // for as long as the colour lasts choice[s] = choice[s - P].  The walk then WRITES the pattern to the end of the stretch,
    // across run boundaries (side records, row rules as below), and takes its queue up from there.
    auto choice_at = [&](const unsigned step, const unsigned P) -> unsigned {     // (step >= T - kHist)
        return (unsigned)hist[(step < T ? step : T - P + (step - T) % P) & (kHist - 1)];
    };
    auto find_period = [&]() -> unsigned {
        const unsigned span = T - flat_from;                        // >= 48
        const unsigned pmax = span - 16u < kHist - 32u ? span - 16u : kHist - 32u;
        unsigned w4 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) w4 |= (unsigned)hist[(T - 16u + (unsigned)i) & (kHist - 1)] << (8 * i);
        for (unsigned P0 = 1; P0 <= pmax; P0 += 64) {
            const unsigned P = P0 + (unsigned)lane;
            bool ok = P <= pmax;
            unsigned v4 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) v4 |= (unsigned)hist[(T - 16u - P + (unsigned)i) & (kHist - 1)] << (8 * i);
            ok = ok && v4 == w4;
            if (__any(ok)) {
                if (ok) for (unsigned i = 4; i < 16; i++) ok = ok && hist[(T - 16u - P + i) & (kHist - 1)] == hist[(T - 16u + i) & (kHist - 1)];
                const unsigned long long m = __ballot(ok);
                if (m) return P0 + (unsigned)__builtin_ctzll(m);
            }
        }
        return 0u;
    };
    auto jump = [&](const unsigned P) {
        if (lane == 0) atomicAdd(a.list - 2, 1u);                   // patolette_amd__Stats::dither_jumps
        unsigned step = T, p = base + head - 16u;                   // the next step's number; the run's next position
        const unsigned s1 = (head - 1u) & (kRing - 1);
        const double cx = rpx[s1], cy = rpx[kRing + s1], cz = rpx[2 * kRing + s1];
        for (;;) {
            unsigned J = len - p;                                   // positions ahead in run r of the same colour
            bool ended = false;
            for (unsigned q0 = p; q0 < len && !ended; q0 += 64) {
                const unsigned q = q0 + (unsigned)lane;
                bool ne = false;
                if (q < len) { const size_t rr = a.R.idx(r, q); ne = !(a.sx[rr] == cx && a.sy[rr] == cy && a.sz[rr] == cz); }
                const unsigned long long m = __ballot(ne);
                if (m) { J = q0 - p + (unsigned)__builtin_ctzll(m); ended = true; }
            }
            for (unsigned m0 = 0; m0 < J; m0 += 64) {
                const unsigned m = m0 + (unsigned)lane;
                if (m < J) a.smap[a.R.idx(r, p + m)] = (unsigned char)choice_at(step + m, P);
            }
            step += J; p += J;
            if (ended) break;                                       // the colour changes inside run r
            // the end of run r: the rules of the walk below
            const unsigned nr = r + 1;
            if (nr >= a.R.S) { halt = true; break; }
            const bool listed = a.flag[nr] != 0;
            if (listed && !in_row && !solo) { halt = true; break; }
            if (!listed) in_row = false;
            if (lane < 16) a.side[16ull * nr + lane] = (unsigned short)choice_at(step - 16u + (unsigned)lane, P);
            r = nr; p = 0;
            len = (unsigned)(a.R.t(r + 1) - a.R.t(r));
        }
        // the last sixteen choices and pixels (all of the one colour: the stretch was 48 steps and more before the jump)
        tail = choice_at(step - 16u + (unsigned)(lane & 15), P);
        __builtin_amdgcn_wave_barrier();
        if (lane < 16) { rpx[lane] = cx; rpx[kRing + lane] = cy; rpx[2 * kRing + lane] = cz; }
        head = count = 16;
        base = p;
        T = 0; flat_from = 0;                                       // (the history starts again)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        requeue((unsigned)__shfl((int)tail, dph, 64));
    };

    for (;;) {
        len = (unsigned)(a.R.t(r + 1) - a.R.t(r));
        if (r == b) {
            // the sixteen pixels before the run and the choices the map holds for them
            const unsigned lp = (unsigned)(a.R.t(r) - a.R.t(r - 1));
            if (lane < 16) {
                const size_t rr = a.R.idx(r - 1, lp - 16 + lane);
                rpx[lane] = a.sx[rr]; rpx[kRing + lane] = a.sy[rr]; rpx[2 * kRing + lane] = a.sz[rr];
                rpos[lane] = (unsigned)rr;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            tail = (unsigned)a.smap[rpos[dph]];                      // (every lane: the choice of predecessor dph)
        }
        // what run r starts from: recorded, and the queue rebuilt from it (a partial group may have left the lanes' phases anywhere)
        const unsigned c16 = (unsigned)__shfl((int)tail, dph, 64);
        if (lane < 16) a.side[16ull * r + lane] = (unsigned short)tail;
        requeue(c16);
        {   // re-base the ring: the sixteen pixels just used move to slots 0 .. 15
            double m0 = 0, m1 = 0, m2 = 0;
            if (lane < 16) { const unsigned sl = (head - 16u + (unsigned)lane) & (kRing - 1); m0 = rpx[sl]; m1 = rpx[kRing + sl]; m2 = rpx[2 * kRing + sl]; }
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) { rpx[lane] = m0; rpx[kRing + lane] = m1; rpx[2 * kRing + lane] = m2; }
            head = count = 16;
            base = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        done = false;
        for (;;) {
            unsigned pl = base + (count - 16u);                     // the run's next position to fetch
            if (pl < len) {
                const unsigned nld = len - pl < 64u ? len - pl : 64u;
                if ((unsigned)lane < nld) {
                    const size_t rr = a.R.idx(r, pl + (unsigned)lane);
                    const unsigned slot = (count + (unsigned)lane) & (kRing - 1);
                    rpx[slot] = a.sx[rr]; rpx[kRing + slot] = a.sy[rr]; rpx[2 * kRing + slot] = a.sz[rr];
                    rpos[slot] = (unsigned)rr;
                }
                count += nld;
                pl += nld;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            bool jumped = false;
            while ((int)(count - head) >= 16 && !done && !jumped) {
                group(16);
                // every 64 steps of a stretch: has the chain come round?
                if (!done && T - flat_from >= 48u && ((T - flat_from) & 63u) < 16u) {
                    const unsigned P = find_period();
                    if (P) { jump(P); jumped = true; }
                }
            }
            if (done || halt) break;
            if (jumped) continue;                                   // (r, len, base are the jump's)
            if (pl >= len) {
                if (count != head) group((int)(count - head));
                break;
            }
        }
        if (halt) break;
        // run r is final for this pass.  On into the next one?
        const unsigned nr = r + 1;
        if (nr >= a.R.S) break;
        const bool listed = a.flag[nr] != 0;
        if (done) {
            // met what was there: the rest of run r stands.  If the next boundary stands too (or another row starts there), finished;
            // inside this row the next run was started from something else: on, from the end of run r as the map has it
            if (!(listed && (in_row || solo))) break;
            if (lane < 16) {
                const size_t rr = a.R.idx(r, len - 16 + lane);
                const unsigned sl = (head - 16u + (unsigned)lane) & (kRing - 1);
                rpx[sl] = a.sx[rr]; rpx[kRing + sl] = a.sy[rr]; rpx[2 * kRing + sl] = a.sz[rr];
                rpos[sl] = (unsigned)rr;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            tail = (unsigned)a.smap[rpos[(head - 16u + (unsigned)dph) & (kRing - 1)]];
            T = 0; flat_from = 0;                                   // (steps were skipped: the history starts again)
        } else if (listed && !in_row && !solo) break;               // the head of another row: its own wavefront's (checked again next pass)
        if (!listed) in_row = false;
        if (++walked >= (solo ? 65536u : kMaxWalk)) break;
        r = nr;
    }
}

struct DitherConfig { int segments = 0, warm = -1, lanes = -1; };    // 0 / -1 = chosen by launch_dither
// process-wide knobs, set from any thread while batch workers read them: atomics (a reader takes ONE snapshot per call)
static std::atomic<int> g_dither_segments{0}, g_dither_warm{-1}, g_dither_lanes{-1};
static std::atomic<int> g_dither_solo_cap{4096};                     // patolette_amd_debug_dither_solo_cap: solo passes before the lane layout gives up
static DitherConfig dither_cfg_snapshot() { return DitherConfig{g_dither_segments.load(), g_dither_warm.load(), g_dither_lanes.load()}; }
void dither_config(int segments, int warm) { g_dither_segments = segments; g_dither_warm = warm; }
void dither_layout(int lanes) { g_dither_lanes = lanes; }
int dither_solo_cap(int cap) { return g_dither_solo_cap.exchange(cap < 0 ? 4096 : cap); }
static std::atomic<int> g_dither_stall_passes{2};                    // passes without progress before one wavefront goes alone (tests: 0 = at once)
int dither_stall_passes(int n) { return g_dither_stall_passes.exchange(n < 0 ? 2 : n); }

static int current_device() { int d = -1; (void)hipGetDevice(&d); return d; }
static std::atomic<bool> g_dither_order_cache{true};
void dither_order_cache(bool on) { g_dither_order_cache = on; }

// One lane per run (k_dither_lanes): K in [8, 256], images of 2^16 pixels and more.  h_pal: the palette on the host, planar (k,3).
// Returns false when the verification stalls (a long flat stretch whose colour is not a palette entry: launch_dither_waves' comment):
// the caller then takes the wavefront layout, which can walk one run through its successors.
static bool launch_dither_lanes(const double *d_img, size_t plane_stride, int which, size_t width, size_t height, const double *d_pal, const double *h_pal,
                                int k, void *d_out, int elem_bytes, NNWork &w, const DitherConfig &cfg, const DitherWeights &wts, hipStream_t s) {
    const size_t npix = width * height;
    DitherLanes a{};
    // runs: eight wavefronts of 64 per CU (two per SIMD), none shorter than 256 pixels unless asked for
    size_t S = cfg.segments > 0 ? (size_t)cfg.segments : std::min<size_t>((size_t)num_cus() * 8 * 64, npix / 256);
    S = std::max<size_t>(1, std::min(std::min(S, npix / 64), (size_t)8 * 65535));    // (a tile row per 8 runs: grid.y)
    a.R.N = npix; a.R.S = (unsigned)S; a.R.Lmax = (unsigned)ceil_div(npix, S);
    // warm-up: 256 pixels.  With a repair pass that costs a tenth of a millisecond (one wavefront per listed run) the 2-6 % of the
    // boundaries that 256 steps do not settle are cheaper than 256 more steps for every run (8192^2: noise 3.65 -> 3.28 ms,
    // scene 6.0 -> 5.4, gradients + 2 % noise 5.36 -> 5.27; 384 lies between)
    a.warm = (unsigned)std::min<size_t>(cfg.warm >= 0 ? (size_t)cfg.warm : 256, npix / S);     // (the warm-up of a run is the end of its predecessor)
    const size_t nw = ceil_div(S, 64), cells = nw * a.R.Lmax * 64;
    // the grid of the exact-pruning records: the weighted palette's bounding box, half its extent wider on every side -- error
    // diffusion pushes queries beyond the palette's hull; what still falls outside takes the full scan
    std::vector<double> wp(3 * (size_t)k);
    const double fw[3] = {(double)(float)kRw, (double)(float)kGw, (double)(float)kBw};
    NNGrid g, gw[2];
    g.G = npix >= ((size_t)1 << 20) ? 64 : 32;
#ifdef PAMD_KM_TRACE
    if (const char *e = getenv("PAMD_DITHER_GRID")) g.G = atoi(e) == 32 ? 32 : 64;      // diagnostic build: the records' grid
#endif
    for (int c = 0; c < 3; c++) {
        double lo = INFINITY, hi = -INFINITY;
        for (int j = 0; j < k; j++) { const double v = h_pal[(size_t)c * k + j] * fw[c]; wp[(size_t)c * k + j] = v; lo = std::min(lo, v); hi = std::max(hi, v); }
        double r = hi - lo;
        if (!(r > 0) || !std::isfinite(r)) r = 0;
        const double m = 0.5 * r + 1e-3;
        g.lo[c] = lo - m;
        const double Rg = r + 2 * m;
        g.cw[c] = Rg / g.G;
        g.inv[c] = g.G / Rg;
        a.hi[c] = g.lo[c] + Rg;
        // the wider grids: 1.5 extents around the palette's box with the first grid's number of cells, 4 extents with 32 across
        for (int lv = 0; lv < 2; lv++) {
            gw[lv].G = lv == 0 ? g.G : 32;
            const double mo = (lv == 0 ? 1.5 : 4.0) * r + (lv == 0 ? 5e-3 : 1e-2), Ro = r + 2 * mo;
            gw[lv].lo[c] = lo - mo;
            gw[lv].cw[c] = Ro / gw[lv].G;
            gw[lv].inv[c] = gw[lv].G / Ro;
            a.wide[lv].hi[c] = gw[lv].lo[c] + Ro;
        }
    }
    {
        // margin of the f32 first pass (the table comment above k_nn_map_mid): 2.5 E, E = 2^-24 (9 max |p - lo|^2 + 5 |hi - lo|^2) (1 + 1e-3)
        double wmax = 0.0, r2 = 0.0;
        for (int j = 0; j < k; j++) {
            double d = 0.0;
            for (int c = 0; c < 3; c++) { const double t = wp[(size_t)c * k + j] - g.lo[c]; d += t * t; }
            wmax = std::max(wmax, d);
        }
        for (int c = 0; c < 3; c++) { const double t = a.hi[c] - g.lo[c]; r2 += t * t; }
        const double M = 2.5 * 1.001 * 0x1.0p-24 * (9.0 * wmax + 5.0 * r2);
        a.amb = (float)M;
        if ((double)a.amb < M) a.amb = std::nextafter(a.amb, INFINITY);
        if (!(M < 3.0e38)) a.amb = INFINITY;
        // outside the grid the query's own |x - lo|^2 stands where |hi - lo|^2 did (evaluated in f32: 1e-6 relative, inside the 1e-3)
        const double Mp = 2.5 * 1.002 * 0x1.0p-24 * 9.0 * wmax, Mx = 2.5 * 1.002 * 0x1.0p-24 * 5.0;
        a.amb_p = std::nextafter((float)Mp, INFINITY); a.amb_x = std::nextafter((float)Mx, INFINITY);
        if (!(Mp < 3.0e38)) a.amb_p = INFINITY;
    }
    const int ncell = g.G * g.G * g.G, ncellw[2] = {gw[0].G * gw[0].G * gw[0].G, gw[1].G * gw[1].G * gw[1].G};
    w.dtab.reserve(3 * (size_t)k);
    w.lut.reserve(((size_t)ncell + ncellw[0] + ncellw[1]) * 32);
    w.clist.reserve((size_t)((ncell + ncellw[0] + ncellw[1]) / 64) * (1 + kCoarseMax) * 2);
    w.dsort.reserve(3 * cells);
    if (w.dpos.cap < npix) { w.order_w = 0; w.order_h = 0; }
    w.dpos.reserve(npix);
    w.dsmap.reserve(cells + 32 * S + 128);
    w.dside.reserve(S + 16);
    w.dflag.reserve(S + 64);
    HIP_CHECK(hipMemsetAsync(w.dflag.p, 0, S + 64, s));                 // (the check writes entries 1 .. S - 1; 0 and S stay 0)
    w.hrep.reserve(2);
    HIP_CHECK(hipMemcpyAsync(w.dtab.p, wp.data(), wp.size() * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));                               // (wp is a local: the copy must have left the host)
    // the two record grids depend on the palette alone, the gather on the image alone: the grids are built on a side stream
    // (forked off `s` here, joined before the first launch that reads them), a few hundred single-wavefront blocks under a
    // bandwidth-bound kernel
    if (w.side_stream == nullptr || w.side_dev != current_device()) {
        if (w.ev_fork) { (void)hipEventDestroy(w.ev_fork); w.ev_fork = nullptr; }
        if (w.ev_join) { (void)hipEventDestroy(w.ev_join); w.ev_join = nullptr; }
        if (w.ev_join2) { (void)hipEventDestroy(w.ev_join2); w.ev_join2 = nullptr; }
        if (w.side_stream) { (void)hipStreamDestroy(w.side_stream); w.side_stream = nullptr; }
        if (w.side_stream2) { (void)hipStreamDestroy(w.side_stream2); w.side_stream2 = nullptr; }
        HIP_CHECK(hipStreamCreateWithFlags(&w.side_stream, hipStreamNonBlocking));
        HIP_CHECK(hipStreamCreateWithFlags(&w.side_stream2, hipStreamNonBlocking));
        HIP_CHECK(hipEventCreateWithFlags(&w.ev_fork, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&w.ev_join, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&w.ev_join2, hipEventDisableTiming));
        w.side_dev = current_device();
    }
    hipStream_t sb = w.side_stream, sc = w.side_stream2;            // the first grid on one, the two wider ones on the other
    HIP_CHECK(hipEventRecord(w.ev_fork, s));
    HIP_CHECK(hipStreamWaitEvent(sb, w.ev_fork, 0));
    HIP_CHECK(hipStreamWaitEvent(sc, w.ev_fork, 0));
    unsigned char *l1 = w.lut.p, *l2 = w.lut.p + (size_t)ncell * 16;
    {
        KTIME("k_nn_lut_build", sb, 32.0 * ncell);
        hipLaunchKernelGGL(k_nn_lut_coarse<unsigned char>, ncell / 64, 64, 0, sb, (const double *)w.dtab.p, k, g, w.clist.p, (float4 *)nullptr);
        hipLaunchKernelGGL(k_nn_lut_build<unsigned char>, ncell / 64, 64, 0, sb, (const double *)w.dtab.p, k, g, l1, l2, (const unsigned char *)w.clist.p, (unsigned int *)nullptr);
    }
    {
        unsigned char *tp = w.lut.p + (size_t)ncell * 32, *cp = (unsigned char *)w.clist.p + (size_t)(ncell / 64) * (1 + kCoarseMax);
        for (int lv = 0; lv < 2; lv++) {
            unsigned char *t1 = tp, *t2 = tp + (size_t)ncellw[lv] * 16;
            KTIME("k_nn_lut_build", sc, 32.0 * ncellw[lv]);
            hipLaunchKernelGGL(k_nn_lut_coarse<unsigned char>, ncellw[lv] / 64, 64, 0, sc, (const double *)w.dtab.p, k, gw[lv], cp, (float4 *)nullptr);
            hipLaunchKernelGGL(k_nn_lut_build<unsigned char>, ncellw[lv] / 64, 64, 0, sc, (const double *)w.dtab.p, k, gw[lv], t1, t2, (const unsigned char *)cp, (unsigned int *)nullptr);
            a.wide[lv].g = gw[lv]; a.wide[lv].lut = t1; a.wide[lv].lut2 = t2;
            tp += (size_t)ncellw[lv] * 32;
            cp += (size_t)(ncellw[lv] / 64) * (1 + kCoarseMax);
        }
    }
    HIP_CHECK(hipEventRecord(w.ev_join, sb));
    HIP_CHECK(hipEventRecord(w.ev_join2, sc));
    double *sx = w.dsort.p, *sy = sx + cells, *sz = sy + cells;
    const dim3 tiles((unsigned)ceil_div((size_t)a.R.Lmax, 64), (unsigned)nw), tiles8((unsigned)ceil_div((size_t)a.R.Lmax, 256), (unsigned)ceil_div(S, 8));
    if (!(g_dither_order_cache && w.order_w == width && w.order_h == height && w.order_dev == current_device())) {
        // rank -> pixel number is a function of the image's dimensions alone: kept between calls on images of one size
        KTIME("k_dither_order", s, 4.0 * npix);
        const unsigned parts = (unsigned)std::max<size_t>(1, std::min<size_t>(8192, npix / 16384));
        hipLaunchKernelGGL(k_dither_order, parts, 64, 0, s, (unsigned)width, (unsigned)height, parts, w.dpos.p);
        w.order_w = width; w.order_h = height; w.order_dev = current_device();
    }
    {
        KTIME("k_dither_gather", s, 52.0 * npix);
        const unsigned *sp = (const unsigned *)w.dpos.p;
        switch (which) {
            case PAMD_COPY: hipLaunchKernelGGL(k_dither_streams<PAMD_COPY>, tiles8, 256, 0, s, d_img, plane_stride, sp, a.R, sx, sy, sz); break;
            case PAMD_SRGB_TO_REC2020: hipLaunchKernelGGL(k_dither_streams<PAMD_SRGB_TO_REC2020>, tiles8, 256, 0, s, d_img, plane_stride, sp, a.R, sx, sy, sz); break;
            case PAMD_CIELUV_TO_REC2020: hipLaunchKernelGGL(k_dither_streams<PAMD_CIELUV_TO_REC2020>, tiles8, 256, 0, s, d_img, plane_stride, sp, a.R, sx, sy, sz); break;
            case PAMD_ICTCP_TO_REC2020: hipLaunchKernelGGL(k_dither_streams<PAMD_ICTCP_TO_REC2020>, tiles8, 256, 0, s, d_img, plane_stride, sp, a.R, sx, sy, sz); break;
            default: throw HipError("patolette_amd: the dither takes its pixels as linear Rec2020, sRGB, CIELuv or ICtCp");
        }
    }
    HIP_CHECK(hipStreamWaitEvent(s, w.ev_join, 0));
    HIP_CHECK(hipStreamWaitEvent(s, w.ev_join2, 0));
    a.sx = sx; a.sy = sy; a.sz = sz;
    a.smap = w.dsmap.p; a.side = reinterpret_cast<unsigned short *>(w.dsmap.p + ((cells + 63) & ~(size_t)63));
    a.list = w.dside.p + 2;                                          // [0] jumps taken, [1] failing boundaries, then their list
    a.flag = w.dflag.p;
    a.lut = l1; a.lut2 = l2; a.g = g;
    const size_t lds = (size_t)6 * k * sizeof(double) + (size_t)k * sizeof(float4);
    {
        KTIME("k_dither", s, 25.0 * npix);
        hipLaunchKernelGGL(k_dither_lanes<0>, (unsigned)ceil_div(S, 256), 256, lds, s, a, d_pal, k, wts);
    }
    HIP_CHECK(hipGetLastError());
    w.dither_segments = S; w.dither_repairs = 0; w.dither_rounds = 0; w.dither_through = 0; w.dither_jumps = 0; w.dither_solo = 0;
    unsigned prev_nf = 0xFFFFFFFFu;
    int stalled = 0;
    HIP_CHECK(hipMemsetAsync(w.dside.p, 0, sizeof(unsigned), s));
    for (size_t round = 0; S > 1; round++) {
        HIP_CHECK(hipMemsetAsync(w.dside.p + 1, 0, sizeof(unsigned), s));
        {
            KTIME("k_dither_fix", s, 0.0);
            hipLaunchKernelGGL(k_dither_lane_check, (unsigned)ceil_div(S - 1, 256), 256, 0, s, a);
        }
        HIP_CHECK(hipMemcpyAsync(w.hrep.p, w.dside.p, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        const unsigned nf = w.hrep.p[1];
        w.dither_jumps = w.hrep.p[0];
        w.dither_rounds = round + 1;
        static const bool trace = getenv("PAMD_DITHER_TRACE") != nullptr;      // one line per verification pass on stderr
        if (trace) fprintf(stderr, "patolette_amd: dither, lane layout: pass %zu, %u of %zu boundaries fail\n", round + 1, nf, S - 1);
        if (nf == 0) break;
        stalled = (prev_nf != 0xFFFFFFFFu && nf + std::max(1u, prev_nf / 8) >= prev_nf) ? stalled + 1 : 0;
        prev_nf = nf;
        const bool solo = stalled >= g_dither_stall_passes.load(std::memory_order_relaxed);   // no progress twice in a row: one wavefront alone, then the passes resume
        if (solo) {
            w.dither_solo++;
            if (++w.dither_through > (size_t)g_dither_solo_cap.load(std::memory_order_relaxed)) return false;   // (never seen at 4096; the wavefront layout takes the image then)
            stalled = 0; prev_nf = 0xFFFFFFFFu;
        }
        a.solo = solo ? 1 : 0;
        w.dither_repairs += solo ? 1 : nf;
        KTIME("k_dither_fix", s, 0.0);
        // PAMD_DITHER_REPAIR=lanes: the listed runs 64 to a wavefront again (k_dither_lanes<1>), for comparison
        static const bool repair_lanes = getenv("PAMD_DITHER_REPAIR") && !strcmp(getenv("PAMD_DITHER_REPAIR"), "lanes");
        const size_t lds_r = (size_t)6 * k * sizeof(double) + 3 * 128 * sizeof(double) + 128 * sizeof(unsigned) + 2048;   // + the history of choices
        const unsigned nblk = solo ? 1u : nf;
        if (repair_lanes && !solo) hipLaunchKernelGGL(k_dither_lanes<1>, (unsigned)ceil_div((size_t)nf, 256), 256, lds, s, a, d_pal, k, wts);
        else if (k <= 64) hipLaunchKernelGGL(k_dither_lane_repair<1>, nblk, 64, lds_r, s, a, d_pal, k, wts);
        else if (k <= 128) hipLaunchKernelGGL(k_dither_lane_repair<2>, nblk, 64, lds_r, s, a, d_pal, k, wts);
        else hipLaunchKernelGGL(k_dither_lane_repair<4>, nblk, 64, lds_r, s, a, d_pal, k, wts);
        HIP_CHECK(hipGetLastError());
    }
    {
        KTIME("k_dither_unpermute", s, (5.0 + elem_bytes) * npix);
        if (elem_bytes == 1) hipLaunchKernelGGL(k_dither_unpermute<unsigned char>, tiles, 256, 0, s, (const unsigned char *)a.smap, (const unsigned *)w.dpos.p, a.R, (unsigned char *)d_out);
        else if (elem_bytes == 4) hipLaunchKernelGGL(k_dither_unpermute<unsigned int>, tiles, 256, 0, s, (const unsigned char *)a.smap, (const unsigned *)w.dpos.p, a.R, (unsigned int *)d_out);
        else hipLaunchKernelGGL(k_dither_unpermute<unsigned long long>, tiles, 256, 0, s, (const unsigned char *)a.smap, (const unsigned *)w.dpos.p, a.R, (unsigned long long *)d_out);
    }
    HIP_CHECK(hipGetLastError());
    return true;
}

// the pixels into linear Rec2020 in image order (what the wavefront layout reads), for the fall-back out of the lane layout
template <int WHICH>
__global__ __launch_bounds__(256) void k_dither_convert(const double *__restrict__ img, size_t plane_stride, size_t n, double *__restrict__ dst) {
    pow_tables_to_lds();
    __syncthreads();
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double c[3] = {img[i], img[plane_stride + i], img[2 * plane_stride + i]};
        dev_convert<WHICH>(c);
        dst[i] = c[0]; dst[n + i] = c[1]; dst[2 * n + i] = c[2];
    }
}

template <typename OutT>
static void launch_dither_t(int mode, unsigned blocks, const double *d_img, size_t plane_stride, size_t width, size_t height, const double *d_pal,
                            int k, OutT *out, const DitherWeights &wts, size_t lds, double *gtab, const DitherSeg &sg, hipStream_t s) {
#define PAMD_DITHER1(PER, GT, MODE)                                                                                            \
    do {                                                                                                                       \
        HIP_CHECK(hipFuncSetAttribute((const void *)(k_dither<OutT, PER, GT, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL((k_dither<OutT, PER, GT, MODE>), blocks, 64, lds, s, d_img, plane_stride, (unsigned)width, (unsigned)height, d_pal, k, out, wts, gtab, sg); \
    } while (0)
#define PAMD_DITHER(PER, GT)                                                                                                   \
    do { if (mode == 0) PAMD_DITHER1(PER, GT, 0); else PAMD_DITHER1(PER, GT, 1); } while (0)
    if (gtab) PAMD_DITHER(0, true);
    else if (k <= 64) PAMD_DITHER(1, false);
    else if (k <= 128) PAMD_DITHER(2, false);
    else if (k <= 256) PAMD_DITHER(4, false);
    else PAMD_DITHER(0, false);
#undef PAMD_DITHER
#undef PAMD_DITHER1
}

static DitherConfig dither_settings() {
    DitherConfig cfg = dither_cfg_snapshot();
    if (const char *e = getenv("PAMD_DITHER_SEGMENTS")) cfg.segments = atoi(e);
    if (const char *e = getenv("PAMD_DITHER_WARM")) cfg.warm = atoi(e);
    if (const char *e = getenv("PAMD_DITHER_LANES")) cfg.lanes = atoi(e);
    return cfg;
}
// One lane per run where the pruned search applies (8 <= K <= 256) and the image is large: the lane layout needs ~10^5 runs of a few
// hundred pixels to fill the GPU and pays a gather, an un-permute and a repair round of fixed cost -- 2048^2: 2.3 ms against 1.6 ms
// for one wavefront per run, 4096^2: 2.7 ms, 8192^2: 4.8 against 15.6 ms.  dither_layout(1) asks for it from 65 536 pixels on.
bool dither_lane_layout(size_t width, size_t height, int k) {
    const DitherConfig cfg = dither_settings();
    const size_t npix = width * height;
    if (!(k >= 8 && k <= 256) || cfg.segments == 1 || cfg.lanes == 0) return false;
    return cfg.lanes > 0 ? npix >= 65536 : npix >= ((size_t)1 << 23);
}

static void launch_dither_waves(const double *d_img, size_t plane_stride, size_t width, size_t height, const double *d_pal, int k,
                                void *d_out, int elem_bytes, NNWork &w, hipStream_t s);

void launch_dither(const double *d_img, size_t plane_stride, int which, size_t width, size_t height, const double *d_pal, const double *h_pal, int k,
                   void *d_out, int elem_bytes, NNWork &w, hipStream_t s, int layout) {
    if (width * height >> 32) throw HipError("patolette_amd: the dither kernel numbers pixels with 32 bits");
    if (elem_bytes != 1 && elem_bytes != 4 && elem_bytes != 8) throw HipError("patolette_amd: map element size must be 1, 4 or 8");
    {
        const DitherConfig cfg = dither_settings();
        // layout: what the caller decided when it chose the pixels' form (the knobs may change between its look and this one)
        if (layout >= 0 ? layout != 0 : dither_lane_layout(width, height, k)) {
            DitherWeights wts;
            const double m = std::exp(std::log(16.0) / (16.0 - 1));
            double v = 1;
            for (int i = 0; i < 16; i++) { wts.w[i] = v / 16.0; v *= m; }
            std::vector<double> hp;
            if (!h_pal) {
                hp.resize(3 * (size_t)k);
                HIP_CHECK(hipMemcpyAsync(hp.data(), d_pal, hp.size() * sizeof(double), hipMemcpyDeviceToHost, s));
                HIP_CHECK(hipStreamSynchronize(s));
                h_pal = hp.data();
            }
            if (launch_dither_lanes(d_img, plane_stride, which, width, height, d_pal, h_pal, k, d_out, elem_bytes, w, cfg, wts, s)) return;
            // stalled: the wavefront layout from scratch (it needs the pixels as linear Rec2020 in image order)
            const size_t n = width * height;
            if (which != PAMD_COPY) {
                w.dsort.reserve(3 * n);
                const int gb = stream_blocks(n, 16);
                switch (which) {
                    case PAMD_SRGB_TO_REC2020: hipLaunchKernelGGL(k_dither_convert<PAMD_SRGB_TO_REC2020>, gb, 256, 0, s, d_img, plane_stride, n, w.dsort.p); break;
                    case PAMD_CIELUV_TO_REC2020: hipLaunchKernelGGL(k_dither_convert<PAMD_CIELUV_TO_REC2020>, gb, 256, 0, s, d_img, plane_stride, n, w.dsort.p); break;
                    default: hipLaunchKernelGGL(k_dither_convert<PAMD_ICTCP_TO_REC2020>, gb, 256, 0, s, d_img, plane_stride, n, w.dsort.p); break;
                }
                HIP_CHECK(hipGetLastError());
                d_img = w.dsort.p; plane_stride = n; which = PAMD_COPY;
            }
            const size_t lane_passes = w.dither_rounds;
            launch_dither_waves(d_img, plane_stride, width, height, d_pal, k, d_out, elem_bytes, w, s);
            w.dither_rounds += lane_passes;
            return;
        }
        if (which != PAMD_COPY) throw HipError("patolette_amd: the wavefront-per-run dither takes linear Rec2020 pixels");
    }
    launch_dither_waves(d_img, plane_stride, width, height, d_pal, k, d_out, elem_bytes, w, s);
}

static void launch_dither_waves(const double *d_img, size_t plane_stride, size_t width, size_t height, const double *d_pal, int k,
                                void *d_out, int elem_bytes, NNWork &w, hipStream_t s) {
    size_t lds = ((size_t)6 * k + 3 * 128) * sizeof(double) + 128 * sizeof(unsigned int);      // palette (raw + weighted) + the ring of pending pixels
    double *gtab = nullptr;
    if (lds > 150 * 1024) {                                      // K > 3200: the tables in global memory (workspace kept with the engine)
        w.dtab.reserve((size_t)6 * k);
        gtab = w.dtab.p;
        lds = (size_t)3 * 128 * sizeof(double) + 128 * sizeof(unsigned int);
    }
    DitherWeights wts;
    {
        const double m = std::exp(std::log(16.0) / (16.0 - 1));
        double v = 1;
        for (int i = 0; i < 16; i++) { wts.w[i] = v / 16.0; v *= m; }
    }
    // Runs: two wavefronts per SIMD fill the issue slots of the chip (one chain alone uses ~2/3 of its SIMD's); never shorter than
    // the warm-up -- below that the speculative steps outnumber the useful ones.
    const size_t npix = width * height;
    const DitherConfig cfg = dither_settings();
    DitherSeg sg{};
    sg.warm = cfg.warm >= 0 ? (unsigned)cfg.warm : 1024u;
    size_t S = cfg.segments > 0 ? (size_t)cfg.segments : (size_t)num_cus() * 8;
    if (cfg.segments <= 0) S = std::min(S, npix / std::max<size_t>(sg.warm, 256));
    S = std::min(S, npix / 128);                                 // every run after the first has its sixteen predecessors
    if (std::max(width, height) < 16) S = 1;
    if (S < 1) S = 1;
    sg.S = (unsigned)S;
    w.dither_segments = S; w.dither_repairs = 0; w.dither_rounds = 0; w.dither_through = 0; w.dither_jumps = 0; w.dither_solo = 0;
    if (S > 1) {
        w.dside.reserve(16 * S + 16);
        w.hrep.reserve(2);
        sg.side = w.dside.p;
        sg.repairs = w.dside.p + 16 * S;
        sg.b0 = 1; sg.through = 0;
    }
    auto reset_counters = [&]() {                                // [0] = 0 repairs, [1] = no run yet (atomicMin)
        w.hrep.p[0] = 0u; w.hrep.p[1] = 0xFFFFFFFFu;
        HIP_CHECK(hipMemcpyAsync(sg.repairs, w.hrep.p, 2 * sizeof(unsigned), hipMemcpyHostToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));                      // (the pinned words are read back into below)
    };
    if (S > 1) reset_counters();
    auto launch = [&](int mode, unsigned blocks) {
        if (elem_bytes == 1) launch_dither_t<unsigned char>(mode, blocks, d_img, plane_stride, width, height, d_pal, k, (unsigned char *)d_out, wts, lds, gtab, sg, s);
        else if (elem_bytes == 4) launch_dither_t<unsigned int>(mode, blocks, d_img, plane_stride, width, height, d_pal, k, (unsigned int *)d_out, wts, lds, gtab, sg, s);
        else launch_dither_t<unsigned long long>(mode, blocks, d_img, plane_stride, width, height, d_pal, k, (unsigned long long *)d_out, wts, lds, gtab, sg, s);
        HIP_CHECK(hipGetLastError());
    };
    {
        KTIME("k_dither", s, (24.0 + elem_bytes) * width * height);
        launch(0, (unsigned)S);
    }
    if (S == 1 || std::max(width, height) <= 1) return;
    // verify every boundary, walk the runs whose starting state was not the true one again, until a pass finds nothing to do
    // Where the image is FLAT over more than a warm-up and the flat colour is not a palette entry, the true chain is periodic (period
    // 4 .. 14 measured) and a zero-queue chain settles into the same cycle at another phase: it never meets the true one, and a
    // verification pass then fixes one run -- the lowest failing one, whose predecessor is final -- while its neighbours are rebuilt
    // from tails that are about to change.  Two passes in a row that fix next to nothing are taken as that: the lowest failing run
    // is walked THROUGH the runs behind it (one wavefront, the true chain) until it meets what is there -- the end of the flat
    // stretch --, and the passes resume (runs it overwrote pass their next check in sixteen steps).
    unsigned prev_r = 0xFFFFFFFFu;
    int stalled = 0;
    for (size_t round = 0;; round++) {
        if (round > 2 * S + 8) throw HipError("patolette_amd: the dither's boundary repairs did not settle");
        {
            KTIME("k_dither_fix", s, 0.0);
            launch(1, (unsigned)S - 1);
        }
        HIP_CHECK(hipMemcpyAsync(w.hrep.p, sg.repairs, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
        const unsigned r = w.hrep.p[0], lowest = w.hrep.p[1];
        w.dither_rounds = round + 1;
        if (r == 0) break;
        w.dither_repairs += r;
        stalled = (prev_r != 0xFFFFFFFFu && r + std::max(1u, prev_r / 8) >= prev_r) ? stalled + 1 : 0;
        prev_r = r;
        reset_counters();
        if (stalled >= 2 && (size_t)lowest + 1 < S) {
            // the pass above has rebuilt run `lowest` from its final predecessor: it is final now, and the walk starts behind it
            // (the record of that run is voided so that the kernel does not take its boundary for verified)
            HIP_CHECK(hipMemsetAsync(sg.side + 16 * ((size_t)lowest + 1), 0xFF, 16 * sizeof(unsigned), s));
            sg.b0 = lowest + 1; sg.through = 1;
            { KTIME("k_dither_fix", s, 0.0); launch(1, 1u); }
            sg.b0 = 1; sg.through = 0;
            reset_counters();
            w.dither_through++;
            stalled = 0; prev_r = 0xFFFFFFFFu;
        }
    }
}

}  // namespace pamd

// host-side view of the run boundaries (the same function the kernel calls), for the CPU tests
extern "C" void patolette_amd_debug_dither_locate(size_t width, size_t height, unsigned long long t, unsigned long long *d, unsigned long long *c) {
    const size_t mxd = width > height ? width : height;
    int L = 0;
    while (((size_t)1 << L) < mxd) L++;
    pamd::dither_locate(L, (unsigned)width, (unsigned)height, t, *d, *c);
}

#ifdef PAMD_KM_TRACE
extern "C" int patolette_amd_debug_nn_trace(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pamd::g_nn_trace), sizeof(pamd::g_nn_trace)) == hipSuccess ? 0 : -1;
}
extern "C" int patolette_amd_debug_nn_flags(unsigned flags) {
    return hipMemcpyToSymbol(HIP_SYMBOL(pamd::g_nn_flags), &flags, sizeof flags) == hipSuccess ? 0 : -1;
}
extern "C" int patolette_amd_debug_dither_lane_waves(unsigned long long *out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pamd::g_dl_wave), sizeof(unsigned long long) * 4 * (size_t)(n < 4096 ? n : 4096)) == hipSuccess ? 0 : -1;
}
extern "C" int patolette_amd_debug_dither_lane_stats(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pamd::g_dl_stats), sizeof(pamd::g_dl_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(pamd::g_dl_stats), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
extern "C" int patolette_amd_debug_nn_stats(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(pamd::g_nn_stats), sizeof(pamd::g_nn_stats)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(pamd::g_nn_stats), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
