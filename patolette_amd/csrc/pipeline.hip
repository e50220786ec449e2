// pipeline.hip -- host orchestration of the MI355X-native patolette path and its C ABI.
//
// Stage sequencing follows lib/src/patolette.c:157-343 (convert -> GQ -> LQ -> [KMeans] ->
// NN map | dither -> palette back-conversion and write-out); everything O(N) runs in the HIP
// kernels of color.hip / quant.hip / kmeans.hip / map.hip on one stream.  The host keeps only
// the O(K) control work the reference also does serially: 3x3 eigen-solves, the 512-bucket DP
// of the global quantiser, and the greedy split selection (local.c:347-390) replayed over the
// candidate tree the GPU evaluates in rounds.
#include <algorithm>
#include <functional>
#include <chrono>
#include <cmath>
#include <atomic>
#include <memory>
#include <mutex>
#include <thread>
#include <future>

#include "color_device.h"
#include "common.h"
#include "host_math.h"
#include "kmeans.h"
#include "map.h"
#include "saliency.h"
#include "quant.h"

namespace pamd {

// launchers defined in color.hip
void launch_convert(int which, const double *src, double *dst, size_t n, ConvertStats *stats, hipStream_t s, BinK sumk = BinK{0.0, 0.0},
                    BinK momk = BinK{0.0, 0.0}, size_t begin = 0, size_t end = ~(size_t)0, bool init_stats = true);
void launch_convert_rows(int which, const double *rows, double *dst, size_t n, ConvertStats *stats, hipStream_t s,
                         BinK sumk = BinK{0.0, 0.0}, BinK momk = BinK{0.0, 0.0}, size_t begin = 0, size_t end = ~(size_t)0,
                         bool init_stats = true);
void launch_convert_u8(int which, const unsigned char *pixels, int channels, double *dst, size_t n, ConvertStats *stats,
                       hipStream_t s, BinK sumk = BinK{0.0, 0.0}, BinK momk = BinK{0.0, 0.0}, size_t begin = 0, size_t end = ~(size_t)0,
                       bool init_stats = true);
void launch_reconstruct(const void *map, int map_elem, size_t n, const unsigned char *pal_u8, int k, unsigned char *out,
                        hipStream_t s);
void launch_weight_stats(const double *w, size_t n, ConvertStats *stats, hipStream_t s);
void launch_pow(const double *x, double y, double *out, size_t n, hipStream_t s);
void launch_fill_image(double *d, size_t n, uint64_t seed, hipStream_t s);
void launch_fill_weights(double *d, size_t n, uint64_t seed, hipStream_t s);

// --------------------------------------------------------------------------------------------
// kernel timer
// --------------------------------------------------------------------------------------------
// (never destroyed: HIP objects must not be released after the runtime has shut down)
KernelTimer &ktimer() { static thread_local KernelTimer *t = new KernelTimer; return *t; }
int KernelTimer::id_of(const char *name) {
    for (size_t i = 0; i < names.size(); i++) if (names[i] == name) return (int)i;
    names.emplace_back(name); total_ms.push_back(0); total_bytes.push_back(0); launches.push_back(0);
    return (int)names.size() - 1;
}
hipEvent_t KernelTimer::get_event() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); return e;
}
void KernelTimer::begin(int id, hipStream_t s, double bytes, const double *src) {
    Rec r; r.id = id; r.bytes = bytes; r.src = src; r.a = get_event(); r.b = get_event();
    HIP_CHECK(hipEventRecord(r.a, s));
    pending.push_back(r);
}
void KernelTimer::end(hipStream_t s) { HIP_CHECK(hipEventRecord(pending.back().b, s)); }
void KernelTimer::collect() {
    for (auto &r : pending) {
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && !(r.src && *r.src == 0.0)) {
            total_ms[r.id] += ms; total_bytes[r.id] += r.src ? r.bytes * *r.src : r.bytes; launches[r.id] += 1;
        }
        pool.push_back(r.a); pool.push_back(r.b);
    }
    pending.clear();
}
void KernelTimer::reset() { collect(); names.clear(); total_ms.clear(); total_bytes.clear(); launches.clear(); }

// --------------------------------------------------------------------------------------------
// node table scatter / gather (host mirror <-> device table) in one copy + one tiny kernel
// --------------------------------------------------------------------------------------------
// The root's two tilings (consecutive runs of kTileA / kTileP pixels of node 0), its round list and the cleared bucket tables, made ON
// the device in one launch: before, the host built ~10 000 tile records and sent them up in four copies followed by three fills -- 30 us
// of host work during which the GPU had nothing queued behind the conversion, then seven 4.6-us operations in a row.
__global__ __launch_bounds__(256) void k_gq_prepare(Tile *tA, int ntA, Tile *tP, int ntP, unsigned long long N, int *round_nodes, int *node_tile0,
                                                    double *hist, unsigned long long nhist, unsigned long long *hsize, unsigned int *hcount) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x, t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long i = t0; i < (unsigned long long)ntA; i += stride) {
        const unsigned long long o = i * (unsigned long long)kTileA;
        tA[i] = Tile{o, (unsigned)(N - o < (unsigned long long)kTileA ? N - o : (unsigned long long)kTileA), 0u};
    }
    for (unsigned long long i = t0; i < (unsigned long long)ntP; i += stride) {
        const unsigned long long o = i * (unsigned long long)kTileP;
        tP[i] = Tile{o, (unsigned)(N - o < (unsigned long long)kTileP ? N - o : (unsigned long long)kTileP), 0u};
    }
    for (unsigned long long i = t0; i < nhist; i += stride) hist[i] = 0.0;
    for (unsigned long long i = t0; i < (unsigned long long)kBuckets; i += stride) { hsize[i] = 0ULL; hcount[i] = 0u; }
    if (t0 == 0) { round_nodes[0] = 0; node_tile0[0] = 0; node_tile0[1] = ntP; }
}

__global__ void k_put_nodes(NodeDev *table, const NodeIn *stage, const int *ids, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const NodeIn in = stage[i];
    NodeDev &d = table[ids[i]];
    d.begin = in.begin; d.n = in.n; d.gn = in.gn; d.buf = in.buf; d.slot = in.slot; d.child0 = in.child0; d.nchild = in.nchild;
    for (int j = 0; j < 3; j++) { d.axis[j] = in.axis[j]; d.mean[j] = in.mean[j]; }
    d.sw = in.sw; d.klin = in.klin; d.kquad = in.kquad;
    node_reset_outputs(d);
    d.split = in.split; d.psplit = -1;
}
// one record, passed as a kernel argument: no staging copies (the root's record at the start of every call)
__global__ void k_put_node1(NodeDev *table, const int id, const NodeIn in) {
    NodeDev &d = table[id];
    d.begin = in.begin; d.n = in.n; d.gn = in.gn; d.buf = in.buf; d.slot = in.slot; d.child0 = in.child0; d.nchild = in.nchild;
    for (int j = 0; j < 3; j++) { d.axis[j] = in.axis[j]; d.mean[j] = in.mean[j]; }
    d.sw = in.sw; d.klin = in.klin; d.kquad = in.kquad;
    node_reset_outputs(d);
    d.split = in.split; d.psplit = -1;
}
__global__ void k_get_nodes(const NodeDev *table, NodeOut *stage, const int *ids, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const NodeDev &d = table[ids[i]];
    NodeOut o;
    o.begin = d.begin; o.n = d.n; o.gn = d.gn; o.buf = d.buf; o.degenerate = d.degenerate; o.split = d.split; o.psplit = d.psplit;
    o.sw = d.sw;
    for (int j = 0; j < 3; j++) o.mean[j] = d.mean[j];
    for (int q = 0; q < 7; q++) {                          // slot sums are exact (binned parts), any order
        double s0 = 0, s1 = 0;
        for (int k = 0; k < kSlots; k++) { s0 += d.acc[k][q][0]; s1 += d.acc[k][q][1]; }
        o.acc[q][0] = s0; o.acc[q][1] = s1;
    }
    stage[i] = o;
}

// One launch prepares a split round from the packet the host uploaded in one copy: node records into the table,
// the tile lists of both tilings (a pure function of the nodes' segments), and zeroed histograms.
struct RoundSetup {
    const NodeIn *recs; const int *ids; const int *tA0; const int *tP0;    // device pointers into the packet
    int nr, ntA, ntP;
    Tile *tilesA, *tilesP;
    double *hist; size_t nhist;
    unsigned long long *hsize; unsigned int *hcount; size_t nbk;
};
__device__ __forceinline__ int node_of_tile(const int *t0, int nr, int t) {   // largest r with t0[r] <= t
    int lo = 0, hi = nr;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t0[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}
__global__ __launch_bounds__(256) void k_round_setup(NodeDev *table, RoundSetup a) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < (size_t)a.nr; i += stride) {
        const NodeIn in = a.recs[i];
        NodeDev &d = table[a.ids[i]];
        d.begin = in.begin; d.n = in.n; d.gn = in.gn; d.buf = in.buf; d.slot = in.slot; d.child0 = in.child0; d.nchild = in.nchild;
        for (int j = 0; j < 3; j++) { d.axis[j] = in.axis[j]; d.mean[j] = in.mean[j]; }
        d.sw = in.sw; d.klin = in.klin; d.kquad = in.kquad;
    }
    for (size_t i = tid; i < (size_t)a.nr * kNodeResetElems; i += stride)        // outputs of the round's nodes, one store per thread
        node_reset_element(table[a.ids[i / kNodeResetElems]], (int)(i % kNodeResetElems));
    for (size_t t = tid; t < (size_t)a.ntA; t += stride) {
        const int r = node_of_tile(a.tA0, a.nr, (int)t);
        const unsigned long long o = (unsigned long long)((int)t - a.tA0[r]) * kTileA, n = a.recs[r].n;
        a.tilesA[t] = Tile{a.recs[r].begin + o, (unsigned)(n - o < (unsigned long long)kTileA ? n - o : kTileA), (unsigned)a.ids[r]};
    }
    for (size_t t = tid; t < (size_t)a.ntP; t += stride) {
        const int r = node_of_tile(a.tP0, a.nr, (int)t);
        const unsigned long long o = (unsigned long long)((int)t - a.tP0[r]) * kTileP, n = a.recs[r].n;
        a.tilesP[t] = Tile{a.recs[r].begin + o, (unsigned)(n - o < (unsigned long long)kTileP ? n - o : kTileP), (unsigned)a.ids[r]};
    }
    for (size_t i = tid; i < a.nhist; i += stride) a.hist[i] = 0.0;
    for (size_t i = tid; i < a.nbk; i += stride) { a.hsize[i] = 0ULL; a.hcount[i] = 0u; }
}


// =============================================================================================
// The split loop driven from the DEVICE (local.c:318-404 for K <= 256 on one GPU)
// =============================================================================================
// The host-driven loop further down pays, per round, a stream synchronisation, the host's turn (the children's 3x3 eigen-solves,
// the greedy replay, the next round's packet) and an upload: 30-110 us of idle GPU nine times per image -- a third of a
// 1920x1080 call.  Here the rounds follow each other on the stream without the host looking, and the greedy loop of
// local.c:347-390 is replayed ONCE, afterwards, over the evaluated candidate tree (split_cluster(c) depends only on c's members,
// never on the greedy order).  Which leaves to evaluate is decided on the device by a rule that needs no replay and never
// misses a node the replay will commit:
//   the priority of an evaluated node is p(v) = min(benefit(v), p(parent(v))) (its ancestors are committed before it, so the
//   replay reaches v only after benefits >= p(v)).  If M = K - (base clusters) evaluated nodes have p >= tau, the replay's M
//   commits all have benefit >= tau (those M nodes and their ancestors are always available to it), so a leaf u with
//   min(ub(u), p(parent(u))) < tau -- ub an upper bound of any split's benefit -- is never committed and never blocks the
//   replay's exactness test (its bound is below every committed benefit).  tau = a lower estimate of the M-th largest p.
// Per round: k_lq_children (one wavefront per child: moment slots summed, distortion, the bound), k_lq_select (block 0: benefits and
// priorities of the nodes just split, tau, the next round's node list, tile prefixes and sizes; the other blocks meanwhile solve
// the children's eigen-problems with hm::eigen_sym3, the host's own code, whose ~12 us of dependent divisions and square roots
// would otherwise sit on the critical path), then the sweeps, launched with upper-bound grids and reading their sizes from
// device memory (RoundDyn).  The host synchronises once (as many rounds are enqueued as the previous image of this size needed;
// a surplus round is nine empty launches, a deficit one more synchronisation), downloads 16 bytes per node and replays.
// Same decisions as the host-driven loop: same sums, same solver, same comparisons; only the SET of evaluated nodes differs
// (a few more: the host prunes with a heuristic a replay in lock-step can afford).
// what the global quantiser left when it ran on the device (k_gq_control below): base clusters, failure, the cuts [0 = q0 .. q_kbase = 512]
struct GqOut { int kbase, error, pad[2]; int cuts[16]; unsigned long long ticks[8]; };   // ticks: wall_clock64 (100 MHz) at the kernel's phase boundaries, a diagnostic

constexpr int kLqDevMaxK = 256;                       // palette sizes the device-driven loop takes
constexpr int kLqNodeCap = 16 * kLqDevMaxK + 256;     // candidate-tree nodes; beyond it the call starts over on the host-driven loop
constexpr int kLqRoundCap = 512;                      // nodes evaluated per round (more candidates wait for the next one)
constexpr int kLqMaxRounds = 96;

struct LqRec {
    double val;                                       // the split's benefit once known, else the bound `ub`
    int left;                                         // left child (right = +1) once the node is in a round, else -1
    int kn;                                           // known: never splits (one member, solver failed) or its split is evaluated
};
struct LqCen { double mean[3]; double gn; };          // what PALETTE_create needs of a node (create.c:11-33) + its member count
struct LqHead {                                       // copied to the host at every check
    int done, error, rounds, nnodes, neval, pad;
    unsigned long long split_evals, split_px;
    double tau;
    double round_px[kLqMaxRounds], round_nr[kLqMaxRounds];
};
struct LqCtl {                                        // device memory
    LqHead h;
    int K, M, nleaves, pad;
    int ncids[2];                                     // children awaiting finalisation; [control call & 1]: k_lq_select's block 0 writes the NEXT
    RoundDyn dyn;                                     // call's list while the other blocks still read this one's
    int leaves[kLqNodeCap];
    int round_ids[kLqRoundCap];
    int cids[2][2 * kLqRoundCap];
    int tA0[kLqRoundCap + 1], tP0[kLqRoundCap + 1];
    int nparent[kLqNodeCap];
    double np[kLqNodeCap];                            // priority of the evaluated nodes
    LqRec nrec[kLqNodeCap];                           // -> host: the replay's table
    LqCen ncen[kLqNodeCap];                           // -> host: centres
};

// wave-wide maximum of a double through DPP row shifts and row broadcasts (lanes without a source keep their own value)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_f64(const double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int tlo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int thi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    return fmax(v, __hiloint2double(thi, tlo));
}

// The children the last round produced (first launch of a call: the base clusters), ONE WAVEFRONT EACH: the sixteen slots of the
// moment accumulators summed (exact parts: any order), the distortion, and an upper bound of any split's benefit
// (hm::lambda_max_bound x sum of weights: the scatter between two groups along their mean difference cannot exceed the node's
// scatter along that direction; see leaf_bound).  init_kbase > 0: the call's first launch, which also sets the loop's state up
// (the base clusters are nodes init_first .. init_first + init_kbase - 1).
__global__ __launch_bounds__(256) void k_lq_children(NodeDev *nodes, LqCtl *c, const int call, const int eigen_bound, int init_kbase,
                                                     const int init_first, int init_nnodes, const int init_K, const GqOut *gq) {
    const bool init = init_kbase > 0;
    if (!init && c->h.done) return;
    if (init && gq) { init_kbase = gq->kbase; init_nnodes = init_first + init_kbase; }   // the global quantiser ran on the device (k_gq_control): its count
    const int ncids = init ? init_kbase : c->ncids[call & 1];
    const int wave = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (init && blockIdx.x == 0 && threadIdx.x == 0) {
        c->K = init_K; c->M = init_K - init_kbase; c->nleaves = 0; c->ncids[0] = init_kbase; c->dyn = RoundDyn{0, 0, 0, 0};
        c->h.done = 0; c->h.error = 0; c->h.rounds = 0; c->h.nnodes = init_nnodes; c->h.neval = 0; c->h.split_evals = 0ULL; c->h.split_px = 0ULL;
        c->h.tau = 0.0;
        if (gq && gq->error) { c->h.done = 1; c->h.error = 3; c->ncids[0] = 0; }
        // the records below the base clusters (the root) belong to no candidate: k_lq_select's rank count runs over ALL records below
        // nnodes, and what an earlier call or the allocator left there must not pass for an evaluated node (a stale record with a large
        // priority raised tau by one rank: a leaf the replay needed was dropped -- one call in some hundreds, engines out of the pool)
        for (int i = 0; i < init_first; i++) { c->nrec[i] = LqRec{0.0, -1, 1}; c->np[i] = 0.0; c->nparent[i] = -1; }
    }
    if (init && gq && gq->error) return;
    if (wave >= ncids) return;
    const int id = init ? init_first + wave : c->cids[call & 1][wave];
    if (init && lane == 0) c->cids[0][wave] = id;
    NodeDev &d = nodes[id];
    const double *acc = &d.acc[0][0][0];                            // acc[k][q][p]: 224 consecutive doubles; lane r < 14 sums element r of every slot
    double sum = 0;
    if (lane < 14) {
        double v[kSlots];
#pragma unroll
        for (int k = 0; k < kSlots; k++) v[k] = acc[k * 14 + lane];
#pragma unroll
        for (int k = 0; k < kSlots; k++) sum += v[k];
    }
    const double pair = sum + __shfl_down(sum, 1, 64);             // lanes 0, 2, .., 12: first + second part of quantity q = lane / 2
    double cov[7];
#pragma unroll
    for (int q = 0; q < 7; q++)
        cov[q] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(pair), 2 * q), __builtin_amdgcn_readlane(__double2loint(pair), 2 * q));
    if (lane != 0) return;
    double ub = cov[6];
    const bool single = d.gn <= 1ULL;
    const double sw = d.sw;
    if (single || !(sw > 0)) ub = 0;
    else if (eigen_bound) {
        double c6[6];
        for (int q = 0; q < 6; q++) c6[q] = cov[q] / sw;
        const double b = hm::lambda_max_bound(c6) * sw * (1.0 + 1e-9) + 1e-9 * cov[6];
        if (b == b && b < ub) ub = b;
    }
    for (int q = 0; q < 6; q++) d.cov6[q] = cov[q];
    d.dist = cov[6]; d.ub = ub; d.axis_state = 0;
    c->nrec[id] = LqRec{single ? 0.0 : ub, -1, single ? 1 : 0};
    c->nparent[id] = init ? -1 : c->round_ids[wave >> 1];
    c->ncen[id] = LqCen{{d.mean[0], d.mean[1], d.mean[2]}, (double)d.gn};
    c->leaves[(init ? 0 : c->nleaves) + wave] = id;
}

__device__ __forceinline__ int block_excl_scan_256(const int v, int *sw /* [5] */, int &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int inc = (int)wave_scan_incl_u32((unsigned)v);
    __syncthreads();                                   // sw may still be read from the previous scan
    if (lane == 63) sw[wid] = inc;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wid; w++) base += sw[w];
    total = sw[0] + sw[1] + sw[2] + sw[3];
    return base + inc - v;
}

// order-preserving 64-bit key of a priority (negative or NaN -> 0: below every threshold that matters)
__device__ __forceinline__ unsigned long long lq_key(const double p) { return p > 0.0 ? (unsigned long long)__double_as_longlong(p) : 0ULL; }

__global__ __launch_bounds__(256) void k_lq_select(NodeDev *nodes, LqCtl *c, const int call, const int e_lin, const int e_quad, const double beta) {
    if (c->h.done) return;                                         // (set by an EARLIER launch only: every block of this one sees the same)
    const int tid = threadIdx.x, lane = tid & 63;
    const int cur = call & 1, nxt = cur ^ 1;
    const int ncids = c->ncids[cur];
    if (blockIdx.x > 0) {
        // ---- the other blocks: the children's principal axes (cluster.c:191-217 -> pca.c:122-149 -> dsyev), one wavefront per child,
        // needed by the NEXT round's projection only -- solved here, beside the selection, instead of in front of it
        const int w = ((int)blockIdx.x - 1) * 4 + (tid >> 6);
        if (w >= ncids) return;
        NodeDev &d = nodes[c->cids[cur][w]];
        if (d.gn <= 1ULL || !(d.sw > 0)) { if (lane == 0) d.axis_state = d.gn <= 1ULL ? 0 : -1; return; }
        double c6[6];
        for (int q = 0; q < 6; q++) c6[q] = d.cov6[q] / d.sw;
        double a[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
        double wv[3];
        const int info = hm::eigen_sym3(a, wv);
        if (lane == 0) {
            if (info != 0) d.axis_state = -1;
            else { d.axis[0] = a[6]; d.axis[1] = a[7]; d.axis[2] = a[8]; d.axis_state = 1; }
        }
        return;
    }
    __shared__ int sw[5];
    __shared__ unsigned int hist[256];
    __shared__ unsigned long long s_prefix, s_px;
    __shared__ int s_need, s_neval, s_first;
    __shared__ double s_umax[4];
    const int M = c->M;
    const int nn = c->h.nnodes;
    // ---- A. the nodes of the last round are split now: benefit (local.c:256-275) and priority.  A node whose eigen-solve failed a round
    // ago (never seen on finite input) was swept along a zero axis: it never splits, its children are dropped
    const int nr_prev = c->dyn.nr;
    int evald = 0;
    for (int i = tid; i < nr_prev; i += 256) {
        const int id = c->round_ids[i];
        const NodeDev &d = nodes[id];
        if (d.axis_state < 0) { c->nrec[id] = LqRec{0.0, -1, 1}; c->np[id] = 0.0; continue; }
        const int l = d.child0;
        const double b = d.dist - (nodes[l].dist + nodes[l + 1].dist);
        const int par = c->nparent[id];
        const double pp = par < 0 ? INFINITY : c->np[par];
        c->nrec[id].val = b; c->nrec[id].kn = 1;
        c->np[id] = b < pp ? b : pp;
        evald++;
    }
    if (tid == 0) { s_neval = 0; s_px = 0ULL; }
    __syncthreads();
    if (evald) atomicAdd(&s_neval, evald);
    __syncthreads();
    const int neval = c->h.neval + s_neval;
    // ---- B. tau: a lower estimate of the M-th largest priority among the evaluated nodes (three 8-bit radix passes on the order-preserving
    // key: the lower edge of the 2^-12-wide bin that holds it)
    double tau = 0.0;
    if (neval >= M && M > 0) {
        unsigned long long prefix = 0ULL;
        int need = M;
        for (int pass = 0; pass < 3; pass++) {
            const int shift = 56 - 8 * pass;
            hist[tid] = 0u;
            __syncthreads();
            for (int i = tid; i < nn; i += 256) {
                const LqRec r = c->nrec[i];
                if (r.kn && r.left >= 0) {
                    const unsigned long long k = lq_key(c->np[i]);
                    if (pass == 0 || (k >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(k >> shift) & 255ULL], 1u);
                }
            }
            __syncthreads();
            // the bin that holds the need-th largest key: the largest b > 0 whose bins b .. 255 hold `need` keys or more (else bin 0).
            // Thread t takes bin 255 - t: a block scan gives every suffix at once (one thread walking the 256 bins paid an LDS round
            // trip per bin, three passes a call: ~15 of the launch's 22 us)
            {
                const int mine = (int)hist[255 - tid];
                int total;
                const int excl = block_excl_scan_256(mine, sw, total);
                if (tid == 0) s_first = 256;
                __syncthreads();
                if (excl + mine >= need && tid <= 254) atomicMin(&s_first, tid);
                __syncthreads();
                const int f = s_first;
                if (f < 256) { if (tid == f) { s_need = need - excl; s_prefix = prefix | ((unsigned long long)(255 - f) << shift); } }
                else if (tid == 254) { s_need = need - (excl + mine); s_prefix = prefix; }
            }
            __syncthreads();
            need = s_need; prefix = s_prefix;
        }
        tau = __longlong_as_double((long long)prefix);
    }
    const double thr = fmax(kDelta, tau * (1.0 - 1e-9));
    // ---- C. the leaves to evaluate: the children just made and what earlier rounds left waiting.  A leaf whose priority cannot reach
    // tau never will (tau only grows): dropped for good.  Of the others, those far below the best of them wait (`beta`, the host-driven
    // loop's own rule and value: most of them fall below tau before their turn comes -- 500 evaluations a 256-colour palette without
    // the rule, ~330 with it); the best one is always taken, so every round evaluates something, and the loop ends when no leaf reaches
    // tau -- the set the replay needs is complete either way.  After 48 rounds everything that reaches tau is taken (a bound on the
    // number of rounds whatever the content).
    const int nleaves = c->nleaves + ncids;
    auto leaf_prio = [&](const int id) -> double {
        const LqRec r = c->nrec[id];
        const int par = c->nparent[id];
        if (r.kn || (par >= 0 && nodes[par].axis_state < 0)) return -1.0;
        const double pp = par < 0 ? INFINITY : c->np[par];
        return r.val < pp ? r.val : pp;
    };
    double thr2 = thr;
    if (beta > 0.0 && c->h.rounds < 48) {
        double u = -1.0;
        for (int i = tid; i < nleaves; i += 256) { const double pr = leaf_prio(c->leaves[i]); u = pr > u ? pr : u; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const double v = __shfl_xor(u, o, 64); u = v > u ? v : u; }
        __syncthreads();
        if (lane == 0) s_umax[tid >> 6] = u;
        __syncthreads();
        u = fmax(fmax(s_umax[0], s_umax[1]), fmax(s_umax[2], s_umax[3]));
        if (beta * u > thr2) thr2 = beta * u;
    }
    int nkeep = 0, nr = 0;
    for (int base = 0; base < nleaves; base += 256) {
        const int i = base + tid;
        int id = -1, sel = 0, keep = 0;
        if (i < nleaves) {
            id = c->leaves[i];
            const double pr = leaf_prio(id);
            sel = pr >= thr2 ? 1 : 0;
            keep = (!sel && pr >= thr) ? 1 : 0;                      // waits
        }
        int tsel, tkeep;
        const int psel = block_excl_scan_256(sel, sw, tsel);
        if (nr + tsel > kLqRoundCap && sel && nr + psel >= kLqRoundCap) { sel = 0; keep = 1; }   // the round is full: the rest waits for the next one
        const int pkeep = block_excl_scan_256(keep, sw, tkeep);
        if (sel) c->round_ids[nr + psel] = id;
        __syncthreads();
        if (keep) c->leaves[nkeep + pkeep] = id;
        nr = min(nr + tsel, kLqRoundCap);
        nkeep += tkeep;
        __syncthreads();
    }
    if (nr == 0) {                                                 // nothing left that the replay could commit: the loop is over
        if (tid == 0) { c->h.done = 1; c->h.neval = neval; c->h.tau = tau; c->nleaves = 0; c->ncids[nxt] = 0; c->dyn = RoundDyn{0, 0, 0, 0}; }
        return;
    }
    // ---- D. the round: slots, children ids, tile prefixes of both tilings (what the host's packet carries in the host-driven loop)
    if (nn + 2 * nr > kLqNodeCap) {                                // the candidate tree outgrew the table: the host-driven loop takes the call over
        if (tid == 0) { c->h.done = 1; c->h.error = 2; c->dyn = RoundDyn{0, 0, 0, 0}; }
        return;
    }
    int runA = 0, runP = 0;
    for (int base = 0; base < nr; base += 256) {
        const int r = base + tid;
        int ta = 0, tp = 0;
        unsigned long long n = 0ULL;
        if (r < nr) {
            const int id = c->round_ids[r];
            NodeDev &d = nodes[id];
            d.slot = r; d.child0 = nn + 2 * r; d.nchild = 2;
            int P = 1;                                              // the grids of the exact sums follow the node's own size (make_nodedev)
            while ((1ULL << P) < (d.gn > 1ULL ? d.gn : 2ULL)) P++;
            d.klin = make_bink(e_lin, P); d.kquad = make_bink(e_quad, P);
            c->nrec[id].left = nn + 2 * r;
            c->cids[nxt][2 * r] = nn + 2 * r; c->cids[nxt][2 * r + 1] = nn + 2 * r + 1;
            n = d.n;
            ta = (int)((n + kTileA - 1) / kTileA); tp = (int)((n + kTileP - 1) / kTileP);
        }
        int totA, totP;
        const int pa = block_excl_scan_256(ta, sw, totA);
        const int pp = block_excl_scan_256(tp, sw, totP);
        if (r < nr) { c->tA0[r] = runA + pa; c->tP0[r] = runP + pp; }
        runA += totA; runP += totP;
        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);     // pixels of the round (a statistic)
        if (lane == 0 && n) atomicAdd(&s_px, n);
    }
    __syncthreads();
    if (tid == 0) {
        const int round = c->h.rounds;
        const unsigned long long rpx = s_px;
        c->tA0[nr] = runA; c->tP0[nr] = runP;
        c->nleaves = nkeep; c->ncids[nxt] = 2 * nr;
        c->dyn = RoundDyn{nr, runA, runP, 0};
        c->h.nnodes = nn + 2 * nr; c->h.neval = neval; c->h.tau = tau;
        c->h.split_evals += (unsigned long long)nr; c->h.split_px += rpx; c->h.rounds = round + 1;
        if (round < kLqMaxRounds) { c->h.round_px[round] = (double)rpx; c->h.round_nr[round] = (double)nr; }
    }
}

// What the host's replay needs of the loop's state -- the head, 16 bytes per node of the candidate tree and the nodes' centres --
// stored straight into pinned host memory by one launch (three copy-engine transfers of the full-capacity arrays before)
__global__ __launch_bounds__(256) void k_lq_export(const LqCtl *c, LqHead *head, LqRec *rec, LqCen *cen) {
    const int nn = c->h.nnodes;
    const int t = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int i = t; i < nn; i += stride) { rec[i] = c->nrec[i]; cen[i] = c->ncen[i]; }
    const double *src = reinterpret_cast<const double *>(&c->h);
    double *dst = reinterpret_cast<double *>(head);
    for (int i = t; i < (int)(sizeof(LqHead) / sizeof(double)); i += stride) dst[i] = src[i];
}

// the split trace's records of a device-driven call, made when somebody asks for them (patolette_amd_last_split_trace): one thread
// per commit of the replay
struct LqCommit { int row, node, new_row, left; };
__global__ void k_lq_trace(const NodeDev *nodes, const LqCommit *commits, int n, patolette_amd__SplitRecord *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const LqCommit cm = commits[t];
    const NodeDev &h = nodes[cm.node], &hl = nodes[cm.left], &hr = nodes[cm.left + 1];
    patolette_amd__SplitRecord tr;
    tr.row = cm.row; tr.new_row = cm.new_row;
    tr.split = hl.psplit < 0 ? -1 : (hl.psplit & 0xffff); tr.degenerate = hl.psplit < 0 ? 0 : (hl.psplit >> 16) & 1;
    tr.n = h.gn; tr.n_left = hl.gn; tr.n_right = hr.gn; tr.sw = h.sw;
    for (int j = 0; j < 3; j++) tr.axis[j] = h.axis[j];
    for (int q = 0; q < 6; q++) tr.cov6[q] = h.cov6[q] / h.sw;
    tr.dist = h.dist; tr.dist_left = hl.dist; tr.dist_right = hr.dist; tr.benefit = h.dist - (hl.dist + hr.dist);
    out[t] = tr;
}

// what k_round_setup does from the host's packet, from the control kernel's lists: cleared outputs of the round's nodes, both tile
// lists, zeroed histograms
__global__ __launch_bounds__(256) void k_round_setup_dyn(NodeDev *table, const LqCtl *c, Tile *tilesA, Tile *tilesP, double *hist,
                                                         unsigned long long *hsize, unsigned int *hcount, const size_t lqs) {
    const int nr = c->dyn.nr;
    if (nr == 0) return;
    const int ntA = c->dyn.ntA, ntP = c->dyn.ntP;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < (size_t)nr * kNodeResetElems; i += stride)
        node_reset_element(table[c->round_ids[i / kNodeResetElems]], (int)(i % kNodeResetElems));
    for (size_t t = tid; t < (size_t)ntA; t += stride) {
        const int r = node_of_tile(c->tA0, nr, (int)t);
        const NodeDev &d = table[c->round_ids[r]];
        const unsigned long long o = (unsigned long long)((int)t - c->tA0[r]) * kTileA, n = d.n;
        tilesA[t] = Tile{d.begin + o, (unsigned)(n - o < (unsigned long long)kTileA ? n - o : kTileA), (unsigned)c->round_ids[r]};
    }
    for (size_t t = tid; t < (size_t)ntP; t += stride) {
        const int r = node_of_tile(c->tP0, nr, (int)t);
        const NodeDev &d = table[c->round_ids[r]];
        const unsigned long long o = (unsigned long long)((int)t - c->tP0[r]) * kTileP, n = d.n;
        tilesP[t] = Tile{d.begin + o, (unsigned)(n - o < (unsigned long long)kTileP ? n - o : kTileP), (unsigned)c->round_ids[r]};
    }
    for (size_t i = tid; i < lqs * (size_t)nr; i += stride) hist[i] = 0.0;
    for (size_t i = tid; i < (size_t)nr * kBuckets; i += stride) { hsize[i] = 0ULL; hcount[i] = 0u; }
}

// =============================================================================================
// The global quantiser's decisions on the DEVICE (global.c:189-298, cells.c:114-328; round 6)
// =============================================================================================
// Before: eleven DP launches for every k up to 12, three downloads, the host's prefix sums, bias test and backtrack, an upload of
// the bucket -> cluster table and of the base clusters' records: ~50 us of launches and ~150 us of idle GPU in every call.  Here ONE
// block does what the host did, in the host's own arithmetic (the same expressions in the same order on the same sums; the
// eigen-solver is hm::eigen_sym3, compiled for both sides), and the partition follows on the stream without the host looking:
//   1. the eleven inclusive prefixes of the 512-bucket table, each a sequential chain as cells.c:114-136 builds them (one wavefront
//      per chain, out of LDS);
//   2. the principal axis of all buckets (cells.c:225-278);
//   3. for k = 2 .. 12: the bias termination test on the current cells (global.c:99-187; one wavefront per cell solves its 3x3, one
//      thread adds up in order), then -- only if the search goes on -- the DP (global.c:232-280): a FULL step k-1 (sixteen
//      wavefronts, one entry each at a time) and the single entry cut[k][512] the backtrack starts from.  The reference fills the
//      whole table first; what it reads of it is this.  Same candidates, same strict '<' in descending t (the largest t among equal
//      minima, t = n-1 first), so the cuts are the reference's.  Noise and most photographs stop at two cells: no full step at all;
//   4. the bucket -> base cluster table (global.c:328-335) and the base clusters' node records (count, weight, mean from the exact
//      bucket sums), the root's children.
// patolette_amd__SplitTrace's header and the host-driven split loop read GqOut after their next synchronisation.
struct GqTab {                                          // LDS: inclusive prefixes, 1-based (hm::CellMoments)
    unsigned long long w0[kBuckets + 1];
    double t[10][kBuckets + 9];                         // 0..2 w1[r], 3 w2, 4..9 wrs (00,01,11,02,12,22); + 8: the chains read ahead
    __device__ double distortion(int a, int b) const {  // cells.c:141-182
        if (w0[a] == w0[b]) return 0;
        const double q0 = t[0][b] - t[0][a], q1 = t[1][b] - t[1][a], q2 = t[2][b] - t[2][a];
        return t[3][b] - t[3][a] - (q0 * q0 + q1 * q1 + q2 * q2) / (double)(w0[b] - w0[a]);
    }
    __device__ double vcov(int a, int b, int r, int s) const {   // cells.c:184-223
        if (w0[a] == w0[b]) return 0;
        const double cnt = (double)(w0[b] - w0[a]);
        const int rs = 4 + s * (s + 1) / 2 + r;
        return (t[rs][b] - t[rs][a]) / cnt - (t[r][b] - t[r][a]) * (t[s][b] - t[s][a]) / (cnt * cnt);
    }
    __device__ bool axis(int a, int b, double ax[3]) const {     // cells.c:225-278
        double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < 3; s++) for (int r = 0; r <= s; r++) m[s * 3 + r] = vcov(a, b, r, s);
        m[0 * 3 + 2] = m[2 * 3 + 0]; m[0 * 3 + 1] = m[1 * 3 + 0]; m[1 * 3 + 2] = m[2 * 3 + 1];
        double w[3];
        if (hm::eigen_sym3(m, w) != 0) return false;
        ax[0] = m[6]; ax[1] = m[7]; ax[2] = m[8];
        return true;
    }
    __device__ static double norm3(const double a[3]) { double s = 0; for (int i = 0; i < 3; i++) s += a[i] * a[i]; return sqrt(s); }   // pow(x, 2) is x * x
    __device__ double bias(int a, int b, const double ax[3]) const {   // cells.c:280-328
        double ca[3];
        if (!axis(a, b, ca)) return -1;
        const double norms = norm3(ax) * norm3(ca);
        if (norms < 1e-16) return 0;
        const double dot = (ca[0] * ax[0] + ca[1] * ax[1] + ca[2] * ax[2]);
        return fmin(1.0, fabs(dot / norms));
    }
};

// entry (k, n) of the DP by ONE wavefront (k_gq_dp's block, quant.hip): E_k[n] and L[k][n] from Ep = E_{k-1}
__device__ __forceinline__ void gq_dp_entry(const GqTab &c, const double *Ep, const int k, const int n, const int lane, double &e_out, int &cut_out) {
    double best = INFINITY; int bt = -1;
    for (int t = n - 2 - lane; t >= k - 1; t -= 64) {
        const double v = Ep[t] + c.distortion(t, n);
        if (v < best) { best = v; bt = t; }                          // descending t within the lane: first minimum = largest t
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_down(best, o, 64); const int t2 = __shfl_down(bt, o, 64);
        if (t2 >= 0 && (bt < 0 || v2 < best || (v2 == best && t2 > bt))) { best = v2; bt = t2; }
    }
    double e = Ep[n - 1]; int cut = n - 1;
    if (bt >= 0 && best < e) { e = best; cut = bt; }
    e_out = e; cut_out = cut;                                        // (lane 0 holds the wavefront's result)
}

__global__ __launch_bounds__(1024) void k_gq_control(const double *__restrict__ hist, const unsigned int *__restrict__ hcount, GqDpDev *g,
                                                     const int palette_size, const int weighted, NodeDev *nodes, const int base0,
                                                     unsigned char *lut, GqOut *out, GqOut *out_host) {
    __shared__ GqTab c;
    __shared__ double E[2][kBuckets + 1];
    __shared__ int s_q[16], s_nq, s_stop, s_err, s_top;
    __shared__ double s_ax[3], s_cd[16], s_cb[16];
    __shared__ unsigned char s_lut[kBuckets];
    __shared__ double s_part[kGqMaxK][4][2];
    __shared__ unsigned long long s_cnt[kGqMaxK];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __shared__ unsigned long long s_ticks[8];
    if (tid == 0) { for (int i = 0; i < 8; i++) s_ticks[i] = 0; s_ticks[0] = wall_clock64(); }
    // ---- 1. prefixes
    for (int i = tid; i < 10 * kBuckets; i += 1024) {
        const int hq = i / kBuckets, b = i - hq * kBuckets;
        c.t[hq][b + 1] = hist[(size_t)(hq * 2 + 0) * kBuckets + b] + hist[(size_t)(hq * 2 + 1) * kBuckets + b];
    }
    for (int b = tid; b < kBuckets; b += 1024) c.w0[b + 1] = hcount[b];
    if (tid < 10) c.t[tid][0] = 0;
    if (tid == 10) c.w0[0] = 0;
    __syncthreads();
    if (tid == 0) s_ticks[5] = wall_clock64();
    if (wv < 10) {
        // table[i] += table[i-1], i ascending: ONE sequential chain of 512 additions per table (cells.c:114-136), in registers.  Lane l
        // holds entries 8 l + 1 .. 8 l + 8.  Pass A walks the lanes in order -- every lane adds its eight entries to the running sum
        // it is handed (only lane l's are the chain's; the others' work is discarded) and the sum after lane l's entries goes on to
        // lane l + 1 through a scalar register; pass B lets every lane redo ITS eight additions from the sum it was handed: the same
        // operands in the same order, so the same bits.  (Through LDS, one lane per chain: ten wavefronts' loads and stores queued
        // on the CU's one LDS pipeline, 23 us; k_gq_prefix's chains, with a round trip per step, 16 us for five.)
        double *t = c.t[wv];
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = t[1 + lane * 8 + u];
        double carry = 0, cin = 0;
        for (int l = 0; l < 64; l++) {
            double a = carry;
#pragma unroll
            for (int u = 0; u < 8; u++) a = v[u] + a;
            cin = lane == l ? carry : cin;
            carry = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(a), l), __builtin_amdgcn_readlane(__double2loint(a), l));
        }
        double a = cin;
#pragma unroll
        for (int u = 0; u < 8; u++) { a = v[u] + a; t[1 + lane * 8 + u] = a; }
    }
    if (wv == 10) {                                                  // the counts: integers, any order -- a wave scan, eight buckets per lane
        unsigned long long v[8], run = 0;
#pragma unroll
        for (int u = 0; u < 8; u++) { run += c.w0[1 + lane * 8 + u]; v[u] = run; }
        unsigned long long inc = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        const unsigned long long before = inc - run;
#pragma unroll
        for (int u = 0; u < 8; u++) c.w0[1 + lane * 8 + u] = before + v[u];
    }
    __syncthreads();
    if (tid == 0) s_ticks[1] = wall_clock64();
    // ---- 2. the axis of everything, E_1; L as the reference starts it (zero, L[i][i] = i: global.c:236-238)
    for (int i = tid; i < (kGqMaxK + 1) * (kBuckets + 1); i += 1024) { const int r = i / (kBuckets + 1), n = i - r * (kBuckets + 1); g->cut[r][n] = r == n ? r : 0; }
    for (int i = tid; i <= kBuckets; i += 1024) E[1][i] = i >= 1 ? c.distortion(0, i) : 0.0;
    if (tid == 0) {
        double ax[3] = {0, 0, 0};
        s_err = c.axis(0, kBuckets, ax) ? 0 : 1;
        s_ax[0] = ax[0]; s_ax[1] = ax[1]; s_ax[2] = ax[2];
        s_q[0] = 0; s_q[1] = kBuckets; s_nq = 2; s_stop = 0;
    }
    __syncthreads();
    const int kmax = palette_size < kGqMaxK ? palette_size : kGqMaxK;
    if (tid == 0) s_ticks[2] = wall_clock64();
    // ---- 3. E[k & 1 ^ 1] ... : E[(k - 1) & 1] holds E_{k-1} when iteration k reaches its DP
    for (int k = 2; k <= kmax && !s_err; k++) {
        const int cells = s_nq - 1;
        if (lane == 0 && wv < cells) {
            const double ax[3] = {s_ax[0], s_ax[1], s_ax[2]};
            s_cd[wv] = c.distortion(s_q[wv], s_q[wv + 1]);
            if (cells == 1) {                                        // the one cell is everything: its axis is `ax`, solved a moment ago
                const double norms = GqTab::norm3(ax) * GqTab::norm3(ax);
                s_cb[wv] = norms < 1e-16 ? 0.0 : fmin(1.0, fabs((ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]) / norms));
            } else s_cb[wv] = c.bias(s_q[wv], s_q[wv + 1], ax);
        }
        __syncthreads();
        if (tid == 0) {                                              // gq_should_terminate (global.c:99-187)
            double distortion = 0;
            for (int j = 0; j < cells; j++) distortion += s_cd[j];
            int stop = 0;
            if (distortion < 1e-16) stop = 1;
            else {
                double bias = 0;
                for (int i = 0; i < cells; i++) {
                    const double cd = s_cd[i], cb = s_cb[i];
                    if (cb < 0) { stop = 1; break; }                 // (global.c:174-177 sets a flag its caller never reaches: the loop breaks first)
                    if (cb < 0.9) continue;
                    bias += (cd / distortion) * cb;
                }
                if (!stop) stop = bias < 0.1 ? 1 : 0;
            }
            s_stop = stop;
        }
        __syncthreads();
        if (s_stop) break;
        if (k > 2) {                                                 // the full step k-1: E_{k-1} and L[k-1][.] from E_{k-2}
            const int kk = k - 1;
            const double *Ep = E[(kk - 1) & 1];
            double *En = E[kk & 1];
            for (int n = wv; n <= kBuckets; n += 16) {                   // (sixteen wavefronts: the launch is 1024 threads, launch_bounds above)
                if (n < kk + 1) { if (lane == 0) En[n] = Ep[n]; continue; }
                double e; int cut;
                gq_dp_entry(c, Ep, kk, n, lane, e, cut);
                if (lane == 0) { En[n] = e; g->cut[kk][n] = cut; }
            }
            __syncthreads();
        }
        if (wv == 0) {
            double e; int cut;
            gq_dp_entry(c, E[(k - 1) & 1], k, kBuckets, lane, e, cut);
            if (lane == 0) s_top = cut;
        }
        __syncthreads();
        if (tid == 0) {                                              // the backtrack (global.c:282-296)
            int t = s_top;
            s_q[k - 1] = t;
            for (int j = k - 2; j >= 1; j--) { t = g->cut[j + 1][t]; s_q[j] = t; }
            s_q[0] = 0; s_q[k] = kBuckets; s_nq = k + 1;
        }
        __syncthreads();
    }
    const int kbase = s_nq - 1;
    if (tid == 0) s_ticks[3] = wall_clock64();
    // ---- 4. bucket -> base cluster (global.c:328-335), the clusters' records
    for (int b = tid; b < kBuckets; b += 1024) {
        int j = 0;
        while (j < kbase - 1 && !(b + 1 <= s_q[j + 1])) j++;
        s_lut[b] = (unsigned char)j;
        lut[b] = (unsigned char)j;
    }
    __syncthreads();
    const int nq = weighted ? 4 : 3;
    // the clusters' sums: wavefront (q, part) holds its row of the table, eight buckets per lane, and adds up every cluster's share
    // (exact parts on the bin grids: any order); wavefront 8 the counts
    if (wv < 2 * nq) {
        const int q = wv >> 1, part = wv & 1, qi = weighted ? 10 + q : q;
        const double *h = hist + (size_t)(qi * 2 + part) * kBuckets + lane * 8;
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = h[u];
        for (int j = 0; j < kbase; j++) {
            double a = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) a += s_lut[lane * 8 + u] == j ? v[u] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane == 0) s_part[j][q][part] = a;
        }
    } else if (wv == 8) {
        unsigned long long v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = hcount[lane * 8 + u];
        for (int j = 0; j < kbase; j++) {
            unsigned long long a = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) a += s_lut[lane * 8 + u] == j ? v[u] : 0ULL;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            if (lane == 0) s_cnt[j] = a;
        }
    }
    __syncthreads();
    if (tid < kbase) {
        const int j = tid;
        unsigned long long pos = 0;
        for (int i = 0; i < j; i++) pos += s_cnt[i];
        const unsigned long long cnt = s_cnt[j];
        NodeDev &ch = nodes[base0 + j];
        ch.begin = pos; ch.n = cnt; ch.gn = cnt; ch.buf = 1; ch.slot = -1; ch.child0 = -1; ch.nchild = 0;
        const double sw = weighted ? (s_part[j][3][0] + s_part[j][3][1]) : (double)cnt;
        const double inv = 1 / sw;
        for (int q = 0; q < 3; q++) { ch.axis[q] = 0; ch.mean[q] = (s_part[j][q][0] + s_part[j][q][1]) * inv; }
        ch.sw = sw; ch.klin = nodes[0].klin; ch.kquad = nodes[0].kquad;
        node_reset_outputs(ch);
        ch.split = -1; ch.psplit = -1;
    }
    if (tid == 64) {
        NodeDev &root = nodes[0];
        root.child0 = base0; root.nchild = kbase;
        root.split = kbase == 2 ? s_q[1] - 1 : -1;                   // bucket b is in cluster 0 iff b + 1 <= cuts[1] (global.c:328-335)
        GqOut o;
        o.kbase = kbase; o.error = s_err; o.pad[0] = o.pad[1] = 0;
        for (int j = 0; j < 16; j++) o.cuts[j] = j <= kbase ? s_q[j] : 0;
        for (int j = 0; j < 8; j++) o.ticks[j] = s_ticks[j];
        o.ticks[4] = wall_clock64();
        *out = o;
        *out_host = o;                                               // pinned host memory: read after the caller's next synchronisation
    }
}

// the eigen-solver as the control kernel runs it, for the golden test (patolette_amd_eigen_sym3_device)
__global__ void k_eigen_batch(const double *a9, size_t count, double *w3, double *z9, int *info) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double a[9], w[3];
    for (int j = 0; j < 9; j++) a[j] = a9[i * 9 + j];
    info[i] = hm::eigen_sym3(a, w);
    for (int j = 0; j < 3; j++) w3[i * 3 + j] = w[j];
    for (int j = 0; j < 9; j++) z9[i * 9 + j] = a[j];
}

// ---- one image over several GPUs (patolette_amd_slice): the node table's reductions, packed for ONE collective ----
// The caller's collective is an element-wise SUM.  Extrema and per-rank counts travel in a table with one row per rank
// (zero except the sender's own row): after the SUM every rank holds every rank's row and takes the minimum, the maximum
// and the number of members on lower ranks (the node-wide slot of its first member: sort.c:61-79's round-robin rule).
__global__ void k_shard_pack_keys(const NodeDev *table, const int *ids, int n, int rank, int size, unsigned long long *buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const NodeDev &d = table[ids[i]];
    unsigned long long a = ~0ULL, b = 0ULL;
    for (int k = 0; k < kSlots; k++) { a = d.minkey[k] < a ? d.minkey[k] : a; b = d.maxkey[k] > b ? d.maxkey[k] : b; }
    for (int r = 0; r < size; r++) {
        unsigned long long *row = buf + ((size_t)r * n + i) * 3;
        row[0] = r == rank ? a : 0ULL; row[1] = r == rank ? b : 0ULL; row[2] = r == rank ? d.n : 0ULL;
    }
}
__global__ void k_shard_unpack_keys(NodeDev *table, const int *ids, int n, int rank, int size, const unsigned long long *buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    NodeDev &d = table[ids[i]];
    unsigned long long a = ~0ULL, b = 0ULL, before = 0ULL;
    for (int r = 0; r < size; r++) {
        const unsigned long long *row = buf + ((size_t)r * n + i) * 3;
        a = row[0] < a ? row[0] : a; b = row[1] > b ? row[1] : b;
        if (r < rank) before += row[2];
    }
    for (int k = 0; k < kSlots; k++) { d.minkey[k] = ~0ULL; d.maxkey[k] = 0ULL; }
    d.minkey[0] = a; d.maxkey[0] = b; d.gslot0 = before;
}
// centred moments: slot sums are exact (binned parts), so is their SUM over the ranks
__global__ void k_shard_pack_acc(const NodeDev *table, const int *ids, int n, double *buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const NodeDev &d = table[ids[i]];
    for (int q = 0; q < 7; q++) {
        double s0 = 0, s1 = 0;
        for (int k = 0; k < kSlots; k++) { s0 += d.acc[k][q][0]; s1 += d.acc[k][q][1]; }
        buf[(size_t)i * 14 + 2 * q] = s0; buf[(size_t)i * 14 + 2 * q + 1] = s1;
    }
}
__global__ void k_shard_unpack_acc(NodeDev *table, const int *ids, int n, const double *buf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    NodeDev &d = table[ids[i]];
    for (int q = 0; q < 7; q++) {
        for (int k = 1; k < kSlots; k++) { d.acc[k][q][0] = 0; d.acc[k][q][1] = 0; }
        d.acc[0][q][0] = buf[(size_t)i * 14 + 2 * q]; d.acc[0][q][1] = buf[(size_t)i * 14 + 2 * q + 1];
    }
}
// k_cut / the host sized the children from the GROUP's bucket counts; their segments on this GPU are what k_scan found
__global__ void k_shard_children_local(NodeDev *table, const int *round_nodes, int nround) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nround) return;
    const NodeDev &nd = table[round_nodes[i]];
    for (int k = 0; k < nd.nchild; k++) {
        NodeDev &ch = table[nd.child0 + k];
        ch.begin = nd.cbegin[k]; ch.n = nd.cbegin[k + 1] - nd.cbegin[k];
    }
}

struct Bounds {
    double cmax, range, wmax; int e_lin, e_quad; double lo[3], hi[3];
    bool have_sum = false; double sum[3] = {0, 0, 0};        // column sums of the converted image, when the conversion took them
    bool have_mom = false;                                   // ... and the raw second moments with them: both as exact (hi, lo) pairs
    double sum2[3][2] = {{0, 0}, {0, 0}, {0, 0}}, mom2[6][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
    bool nonfinite = false;                                  // a converted value is NaN / Inf as f32 (KMeans then leaves the centres alone)
};

static int exp_bound(double v) {                 // smallest E with 2^E > v (v > 0)
    if (!(v > 0) || !std::isfinite(v)) return 1;
    return std::ilogb(v) + 1;
}

// --------------------------------------------------------------------------------------------
struct Shard {                       // this GPU's part of an image dealt out over a group (patolette_amd_slice)
    size_t total = 0, begin = 0;     // pixels of the whole image; first pixel of the slice
    patolette_amd__Comm comm{};
};

struct Engine {
    int device = -1;
    hipStream_t stream = nullptr;
    const Shard *shard = nullptr;    // set for the duration of a sliced call
    bool invariant = false;          // tiling-invariant children moments (always on while `shard` is set); a property of the CALLING
                                     // THREAD (tl_invariant), copied in whenever an engine starts serving a thread or a batch
    DevBuf<unsigned char> commbuf;
    PinBuf<unsigned char> h_comm;
    DevBuf<int> shard_ids;
    PinBuf<float> h_cent;            // KMeans centroids on their way to / from the device
    PinBuf<ConvertStats> h_cstats;
    hipEvent_t ev_stats = nullptr;
    hipStream_t stream2 = nullptr;        // run_host: the conversion of chunk i runs here while chunk i + 1 is on its way up
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_join = nullptr;
    size_t prep_N = 0, prep_planes = 0;   // gq_prepare() has staged the root's tiles and cleared the tables for an image of this size
    DevBuf<double> src, wsrc, cvt, bufA, bufB, aux;
    DevBuf<unsigned short> bkt;
    DevBuf<NodeDev> nodes;
    DevBuf<NodeIn> stage_in;
    DevBuf<NodeOut> stage_out;
    DevBuf<int> ids, round_nodes, node_tile0;
    DevBuf<Tile> tilesA, tilesP;
    DevBuf<double> hist, sum6, dpal;
    DevBuf<unsigned long long> hsize, tileoff;
    DevBuf<unsigned int> hcount, tilecnt;
    DevBuf<unsigned char> lut, dmap, src8, pal8, quant8;
    DevBuf<ConvertStats> cstats;
    PinBuf<NodeIn> h_stage_in;
    PinBuf<NodeOut> h_stage_out;
    PinBuf<int> h_ids, h_ids_get;
    PinBuf<Tile> h_tilesA, h_tilesP;
    PinBuf<int> h_round, h_tile0;
    PinBuf<double> h_dbl;
    PinBuf<unsigned char> h_bytes, h_packet, h_mapstage;
    DevBuf<unsigned char> packet;
    KMeansWork km;
    NNWork nn;
    SalWork sal;
    DevBuf<double> wsal;
    DevBuf<GqDpDev> gq;
    PinBuf<GqDpDev> h_gq;
    DevBuf<GqOut> gqout;                  // k_gq_control's result (device copy: the split loop's kernels read it; pinned copy: the host, after its next synchronisation)
    PinBuf<GqOut> h_gqout;
    PinBuf<double> h_pal;                 // the palette on its way to the mapping kernels
    DevBuf<LqCtl> lqctl;                  // the device-driven split loop's state; the host's copies of what the replay needs
    PinBuf<LqHead> h_lqhead;
    PinBuf<LqRec> h_lqrec;
    PinBuf<LqCen> h_lqcen;
    std::vector<LqRec> lq_rec_local;      // the replay's table, copied out of the pinned buffer
    std::vector<LqCommit> lq_commits;     // the replay's commits of the last device-driven call: the split trace is made from them on request
    bool trace_pending = false;
    size_t lq_hint_N = 0, lq_hint_K = 0; int lq_hint_rounds = 0;   // rounds the last image of this size and palette needed
    // KMeans subsample list = a prefix of rand_perm(N, seed 1234) (Clustering.cpp:311-319): a pure function of N, and the list
    // for FEWER samples is a prefix of the list for more.  perm_dev holds the first perm_nx entries for an image of perm_N pixels;
    // h_perm (pinned) the first hperm_nx for hperm_N, written by a helper thread (perm_job) that starts at call entry and is
    // waited for when the KMeans stage needs the list -- by then conversion and both quantisers have run (subsample_start/_list)
    DevBuf<int> perm_dev;
    size_t perm_N = 0, perm_nx = 0;
    PinBuf<int32_t> h_perm;
    size_t hperm_N = 0, hperm_nx = 0;
    hm::PermScratch perm_ws;          // the helper's tables (kept: no page faults after the first call)
    std::future<void> perm_job;
    patolette_amd__Stats stats{};
    double ms_saliency = 0.0;
    std::string last_error;
    std::vector<double> map_palette;             // the palette as the mapping stage used it (planar (len,3)): Rec2020 / ICtCp / sRGB
    // what the quantisers of the last call decided (patolette_amd_last_split_trace): a few hundred small records, always kept
    patolette_amd__SplitTrace trace_hdr{};
    std::vector<patolette_amd__SplitRecord> trace;
    std::vector<double> cluster_centers;         // PALETTE_create's rows, planar (len,3), before any KMeans

    void init() {
        if (stream) return;
        int cnt = 0;
        if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0)
            throw HipError("patolette_amd: no HIP device available (this library has no CPU fallback)");
        if (device < 0) { HIP_CHECK(hipGetDevice(&device)); }
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    }
    void sync() { HIP_CHECK(hipStreamSynchronize(stream)); if (ktimer().enabled) ktimer().collect(); }
    ~Engine() {                                  // buffers free themselves (DevBuf / PinBuf); only called while the runtime is alive
        if (stream) { (void)hipSetDevice(device); (void)hipStreamSynchronize(stream); if (ev_stats) (void)hipEventDestroy(ev_stats);
                      for (hipEvent_t e : {ev_up[0], ev_up[1], ev_join}) if (e) (void)hipEventDestroy(e);
                      if (stream2) (void)hipStreamDestroy(stream2);
                      (void)hipStreamDestroy(stream); }
    }
};

// Engines (HIP stream + ~170 bytes of workspace per pixel of the largest image seen) are pooled per device: a thread
// takes one at its first call and hands it back when it exits (no HIP call at thread or process exit: the runtime may
// already be shutting down), so short-lived caller threads neither leak a workspace each nor pay for a new one.
// patolette_amd_release_workspace() gives the memory back.
namespace {
std::mutex g_pool_mu;
std::vector<Engine *> g_pool;
// patolette_amd_set_invariant_sums: the caller's setting lives with the calling thread, not with a pooled engine (an engine
// outlives the thread it served and is handed to others).  -1 = not set: the environment's default.
thread_local int tl_invariant = -1;
bool invariant_default() {
    static const bool env = getenv("PAMD_INVARIANT_SUMS") && atoi(getenv("PAMD_INVARIANT_SUMS")) != 0;
    return tl_invariant < 0 ? env : tl_invariant != 0;
}
Engine *pool_acquire(int device, bool invariant) {   // device < 0: the current device is unknown -> only an engine bound to no GPU yet
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); i++)
            if (g_pool[i]->device == device || g_pool[i]->device < 0) {
                Engine *e = g_pool[i];
                g_pool.erase(g_pool.begin() + i);
                if (e->device < 0) e->device = device;
                e->invariant = invariant;
                return e;
            }
    }
    Engine *e = new Engine;
    e->device = device;
    e->invariant = invariant;
    return e;
}
void pool_release(Engine *e) { std::lock_guard<std::mutex> lk(g_pool_mu); g_pool.push_back(e); }
struct EngineHolder {
    Engine *e = nullptr;
    ~EngineHolder() { if (e) pool_release(e); }
};
EngineHolder &holder() { static thread_local EngineHolder h; return h; }
}  // namespace

static Engine &engine() {
    EngineHolder &h = holder();
    if (!h.e) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        h.e = pool_acquire(dev, invariant_default());
    }
    return *h.e;
}

// ---- the group's collective (element-wise in-place SUM lent by the caller) ----
static size_t comm_elem(int dtype) { return dtype == 2 ? 4 : 8; }
static void comm_sum_dev(Engine &E, void *d, size_t count, int dtype) {         // d: device memory
    if (!count) return;
    HIP_CHECK(hipStreamSynchronize(E.stream));
    const patolette_amd__Comm &c = E.shard->comm;
    int rc;
    if (c.host_buffers) {
        const size_t bytes = count * comm_elem(dtype);
        E.h_comm.reserve(bytes);
        HIP_CHECK(hipMemcpy(E.h_comm.p, d, bytes, hipMemcpyDeviceToHost));
        rc = c.allreduce_sum(c.ctx, E.h_comm.p, count, dtype);
        HIP_CHECK(hipMemcpy(d, E.h_comm.p, bytes, hipMemcpyHostToDevice));
    } else rc = c.allreduce_sum(c.ctx, d, count, dtype);
    if (rc != 0) throw HipError("patolette_amd: the caller's all-reduce reported a failure");
}
static void comm_sum_host(Engine &E, void *h, size_t count, int dtype) {        // h: host memory
    if (!count) return;
    const patolette_amd__Comm &c = E.shard->comm;
    const size_t bytes = count * comm_elem(dtype);
    int rc;
    if (c.host_buffers) rc = c.allreduce_sum(c.ctx, h, count, dtype);
    else {
        E.commbuf.reserve(bytes);
        HIP_CHECK(hipMemcpy(E.commbuf.p, h, bytes, hipMemcpyHostToDevice));
        rc = c.allreduce_sum(c.ctx, E.commbuf.p, count, dtype);
        HIP_CHECK(hipMemcpy(h, E.commbuf.p, bytes, hipMemcpyDeviceToHost));
    }
    if (rc != 0) throw HipError("patolette_amd: the caller's all-reduce reported a failure");
}
// every rank's row of n values (bit patterns travel as i64: one non-zero addend per element)
static std::vector<unsigned long long> comm_gather_u64(Engine &E, const unsigned long long *mine, size_t n) {
    const patolette_amd__Comm &c = E.shard->comm;
    std::vector<unsigned long long> all((size_t)c.size * n, 0ULL);
    for (size_t i = 0; i < n; i++) all[(size_t)c.rank * n + i] = mine[i];
    comm_sum_host(E, all.data(), all.size(), 1);
    return all;
}
// projection extrema + member counts of the nodes whose ids sit at d_ids (device): folded, exchanged, written back
static void shard_exchange_keys(Engine &E, const int *d_ids, int n) {
    if (!n) return;
    const patolette_amd__Comm &c = E.shard->comm;
    const size_t cnt = (size_t)c.size * n * 3;
    E.commbuf.reserve(cnt * 8);
    hipLaunchKernelGGL(k_shard_pack_keys, (n + 63) / 64, 64, 0, E.stream, E.nodes.p, d_ids, n, c.rank, c.size, (unsigned long long *)E.commbuf.p);
    HIP_CHECK(hipGetLastError());
    comm_sum_dev(E, E.commbuf.p, cnt, 1);
    hipLaunchKernelGGL(k_shard_unpack_keys, (n + 63) / 64, 64, 0, E.stream, E.nodes.p, d_ids, n, c.rank, c.size, (const unsigned long long *)E.commbuf.p);
    HIP_CHECK(hipGetLastError());
}
static void shard_exchange_acc(Engine &E, const int *d_ids, int n) {
    if (!n) return;
    E.commbuf.reserve((size_t)n * 14 * 8);
    hipLaunchKernelGGL(k_shard_pack_acc, (n + 63) / 64, 64, 0, E.stream, E.nodes.p, d_ids, n, (double *)E.commbuf.p);
    HIP_CHECK(hipGetLastError());
    comm_sum_dev(E, E.commbuf.p, (size_t)n * 14, 0);
    hipLaunchKernelGGL(k_shard_unpack_acc, (n + 63) / 64, 64, 0, E.stream, E.nodes.p, d_ids, n, (const double *)E.commbuf.p);
    HIP_CHECK(hipGetLastError());
}
static const int *shard_upload_ids(Engine &E, const std::vector<int> &ids) {
    E.shard_ids.reserve(ids.size());
    HIP_CHECK(hipStreamSynchronize(E.stream));
    HIP_CHECK(hipMemcpy(E.shard_ids.p, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
    return E.shard_ids.p;
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// host mirror of one tree node
struct HNode {
    unsigned long long begin = 0, n = 0;     // segment on this GPU
    unsigned long long gn = 0;               // members over the whole group (= n unless the image is sliced over GPUs)
    int buf = 0;
    double sw = 0, mean[3] = {0, 0, 0}, cov6[6] = {0, 0, 0, 0, 0, 0}, dist = 0;
    int left = -1, right = -1;
    int psplit = -1;                         // the parent's cut as the device took it (bucket | degenerate << 16): split trace
    bool split_done = false, nosplit = false;
    // filled when the moments arrive (leaf_bound): the principal axis (dsyev semantics) and an upper bound of the benefit
    // of ANY split of the node
    int axis_state = 0;                      // 0 not solved yet, 1 axis valid, -1 the solver failed (-> nosplit when evaluated)
    double axis[3] = {0, 0, 0};
    double ub = 0;
};

static void build_tiles(const std::vector<int> &round, const std::vector<HNode> &hn, int tile, std::vector<Tile> &out,
                        std::vector<int> *tile0) {
    out.clear();
    if (tile0) tile0->clear();
    for (size_t r = 0; r < round.size(); r++) {
        const HNode &nd = hn[round[r]];
        if (tile0) tile0->push_back((int)out.size());
        for (unsigned long long o = 0; o < nd.n; o += (unsigned long long)tile) {
            Tile t;
            t.start = nd.begin + o;
            t.count = (unsigned)std::min<unsigned long long>((unsigned long long)tile, nd.n - o);
            t.node = (unsigned)round[r];
            out.push_back(t);
        }
    }
    if (tile0) tile0->push_back((int)out.size());
}

static NodeIn make_nodedev(const HNode &h, const Bounds &b) {
    NodeIn d;
    std::memset(&d, 0, sizeof d);
    d.begin = h.begin; d.n = h.n; d.gn = h.gn; d.buf = h.buf; d.slot = -1; d.child0 = -1; d.nchild = 0; d.split = -1;
    for (int j = 0; j < 3; j++) d.mean[j] = h.mean[j];
    d.sw = h.sw;
    int P = 1;
    while ((1ULL << P) < (h.gn > 1 ? h.gn : 2)) P++;
    d.klin = make_bink(b.e_lin, P);
    d.kquad = make_bink(b.e_quad, P);
    return d;
}

// Host staging buffers are pinned and one per purpose; every round ends with a stream sync (get_nodes), so a
// buffer is never rewritten while an earlier async copy from it is still in flight.
static void upload_tiles(Engine &E, const std::vector<Tile> &t, DevBuf<Tile> &dst, PinBuf<Tile> &stage) {
    if (t.empty()) return;
    dst.reserve(t.size());
    stage.reserve(t.size());
    std::memcpy(stage.p, t.data(), t.size() * sizeof(Tile));
    HIP_CHECK(hipMemcpyAsync(dst.p, stage.p, t.size() * sizeof(Tile), hipMemcpyHostToDevice, E.stream));
}
static void upload_ints(Engine &E, const std::vector<int> &v, DevBuf<int> &dst, PinBuf<int> &stage) {
    if (v.empty()) return;
    dst.reserve(v.size());
    stage.reserve(v.size());
    std::memcpy(stage.p, v.data(), v.size() * sizeof(int));
    HIP_CHECK(hipMemcpyAsync(dst.p, stage.p, v.size() * sizeof(int), hipMemcpyHostToDevice, E.stream));
}

static void put_nodes(Engine &E, const std::vector<int> &ids, const std::vector<NodeIn> &recs) {
    const int n = (int)ids.size();
    if (!n) return;
    if (n == 1) {
        hipLaunchKernelGGL(k_put_node1, 1, 1, 0, E.stream, E.nodes.p, ids[0], recs[0]);
        HIP_CHECK(hipGetLastError());
        return;
    }
    E.stage_in.reserve(n); E.ids.reserve(n); E.h_stage_in.reserve(n); E.h_ids.reserve(n);
    std::memcpy(E.h_stage_in.p, recs.data(), n * sizeof(NodeIn));
    std::memcpy(E.h_ids.p, ids.data(), n * sizeof(int));
    HIP_CHECK(hipMemcpyAsync(E.stage_in.p, E.h_stage_in.p, n * sizeof(NodeIn), hipMemcpyHostToDevice, E.stream));
    HIP_CHECK(hipMemcpyAsync(E.ids.p, E.h_ids.p, n * sizeof(int), hipMemcpyHostToDevice, E.stream));
    hipLaunchKernelGGL(k_put_nodes, (n + 63) / 64, 64, 0, E.stream, E.nodes.p, E.stage_in.p, E.ids.p, n);
    HIP_CHECK(hipGetLastError());
    // no sync: h_stage_in / h_ids belong to put_nodes alone and every put is followed by a stream sync (get_nodes or an
    // explicit one) before the next put rewrites them
}

static void get_nodes(Engine &E, const std::vector<int> &ids, std::vector<NodeOut> &recs) {
    const int n = (int)ids.size();
    recs.resize(n);
    if (!n) return;
    E.stage_out.reserve(n); E.ids.reserve(n); E.h_stage_out.reserve(n); E.h_ids_get.reserve(n);
    std::memcpy(E.h_ids_get.p, ids.data(), n * sizeof(int));
    HIP_CHECK(hipMemcpyAsync(E.ids.p, E.h_ids_get.p, n * sizeof(int), hipMemcpyHostToDevice, E.stream));
    hipLaunchKernelGGL(k_get_nodes, (n + 63) / 64, 64, 0, E.stream, E.nodes.p, E.stage_out.p, E.ids.p, n);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(E.h_stage_out.p, E.stage_out.p, n * sizeof(NodeOut), hipMemcpyDeviceToHost, E.stream));
    E.sync();
    std::memcpy(recs.data(), E.h_stage_out.p, n * sizeof(NodeOut));
}

// as get_nodes, the ids already on the device (part of the round packet)
static void get_nodes_dev(Engine &E, const int *d_ids, int n, std::vector<NodeOut> &recs) {
    recs.resize(n);
    if (!n) return;
    E.h_stage_out.reserve(n);
    // the kernel stores the records straight into pinned host memory (device-visible): no copy engine round trip
    hipLaunchKernelGGL(k_get_nodes, (n + 63) / 64, 64, 0, E.stream, E.nodes.p, E.h_stage_out.p, d_ids, n);
    HIP_CHECK(hipGetLastError());
    E.sync();
    std::memcpy(recs.data(), E.h_stage_out.p, n * sizeof(NodeOut));
}

static void absorb_moments(HNode &h, const NodeOut &d) {
    for (int q = 0; q < 6; q++) h.cov6[q] = d.acc[q][0] + d.acc[q][1];
    h.dist = d.acc[6][0] + d.acc[6][1];
}

// principal axis from the node's covariance sums (pca.c:62-101 divides by sum(w), then dsyev)
static bool node_axis(const HNode &h, double axis[3]) {
    if (h.axis_state != 0) {
        for (int j = 0; j < 3; j++) axis[j] = h.axis[j];
        return h.axis_state > 0;
    }
    double c6[6];
    for (int q = 0; q < 6; q++) c6[q] = h.cov6[q] / h.sw;
    return hm::principal_axis(c6, axis);
}

// Solve the node's 3x3 once, when its moments arrive: the axis its split evaluation will use, and an upper bound of the
// benefit of splitting it.  The benefit (local.c:256-274: distortion minus the children's) is the between-group scatter
// sum_g sw_g |mu_g - mu|^2 of the two children; the mean difference has ONE direction u, and the scatter between groups
// along u cannot exceed the node's total scatter along u, u' S u <= lambda_max(S) = sw * lambda_max(cov).  That holds for
// any two-way partition, so it bounds the benefit of the cut the reference will choose -- about three times tighter than
// the distortion (the trace of S) for a roundish cluster.  The margins cover the roundings of the computed sums (~1e-15
// relative to the distortion) with six orders to spare.
static constexpr bool kUseEigenBound = true;
static bool g_lq_eigen_bound = !(getenv("PAMD_LQ_EIGEN_BOUND") && atoi(getenv("PAMD_LQ_EIGEN_BOUND")) == 0);
static void leaf_bound(HNode &h) {
    h.ub = h.dist;
    h.axis_state = 0;
    if (h.gn <= 1 || !(h.sw > 0)) { h.ub = 0; return; }          // never split (local.c:262-264)
    double c6[6];
    for (int q = 0; q < 6; q++) c6[q] = h.cov6[q] / h.sw;
    double a[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    double w[3];
    if (hm::eigen_sym3(a, w) != 0) { h.axis_state = -1; return; }
    h.axis[0] = a[6]; h.axis[1] = a[7]; h.axis[2] = a[8]; h.axis_state = 1;
    if (kUseEigenBound && g_lq_eigen_bound) {
        const double b = w[2] * h.sw * (1.0 + 1e-9) + 1e-9 * h.dist;
        if (b == b && b < h.ub) h.ub = b;                          // NaN / larger: keep the distortion
    }
}

// --------------------------------------------------------------------------------------------
// GQ + LQ + PALETTE_create on the converted image in E.cvt (planar, stride N, optional w plane)
// returns centres planar (len,3)
// --------------------------------------------------------------------------------------------
// Everything the global quantiser needs that depends on the image SIZE only: workspace, the root's two tilings, cleared bucket
// tables.  run_device() enqueues it right behind the conversion kernel, so these small copies and fills run while the host waits
// for the conversion's bounds instead of between the quantiser's kernels.
static void gq_prepare(Engine &E, size_t N, bool weighted) {
    hipStream_t s = E.stream;
    const size_t planes = weighted ? 4 : 3;
    E.bufA.reserve(planes * N + 64); E.bufB.reserve(planes * N + 64); E.bkt.reserve(N);   // + the slack k_scatter_bin's pixel-less lanes store to
    const int ntA = (int)ceil_div(N, (size_t)kTileA), ntP = (int)ceil_div(N, (size_t)kTileP);
    E.tilesA.reserve(ntA); E.tilesP.reserve(ntP);
    E.h_round.reserve(kBuckets);                                // (receives the bucket counts later: sized once, before any copy uses it)
    E.round_nodes.reserve(1); E.node_tile0.reserve(2);
    E.tilecnt.reserve((size_t)ntP * kMaxChildren); E.tileoff.reserve((size_t)ntP * kMaxChildren);
    const size_t hs = hist_slot_doubles();
    E.hist.reserve(hs); E.hsize.reserve(kBuckets); E.hcount.reserve(kBuckets); E.lut.reserve(kBuckets);
    hipLaunchKernelGGL(k_gq_prepare, 64, 256, 0, s, E.tilesA.p, ntA, E.tilesP.p, ntP, (unsigned long long)N, E.round_nodes.p, E.node_tile0.p,
                       E.hist.p, (unsigned long long)hs, E.hsize.p, E.hcount.p);
    HIP_CHECK(hipGetLastError());
    E.prep_N = N; E.prep_planes = planes;
}



// The greedy loop of local.c:347-390 over the evaluated candidate tree (rec: per node the split's benefit once known, else the bound;
// the left child; known).  Exactly the host-driven loop's steps (quantize_clusters_run below), except that nothing is left to
// evaluate: false if a step is blocked by an undecided node all the same (the device's selection rule forbids it).
struct LqReplay { std::vector<int> result; std::vector<LqCommit> commits; bool stopped_early = false; };
// the same loop over the whole frontier at every step (the reference's shape): what PAMD_LQ_REPLAY_CHECK=1 holds the blocked form below to
static bool lq_replay_plain(const LqRec *rec, int kbase, int first_base, size_t K, LqReplay &out) {
    std::vector<int> &result = out.result;
    result.assign(K, -1);
    for (int j = 0; j < kbase; j++) result[j] = first_base + j;
    size_t count = (size_t)kbase;
    std::vector<double> fval(K, 0.0);
    std::vector<char> fkn(K, 0);
    for (size_t j = 0; j < count; j++) { fval[j] = rec[result[j]].val; fkn[j] = (char)rec[result[j]].kn; }
    const int fault = g_debug_fault.load(std::memory_order_relaxed);
    while (count < K) {
        int best = -1; double bv = 0, mu = -1;
        for (size_t j = 0; j < count; j++) {
            if (fkn[j]) { if (best < 0 || fval[j] > bv) { bv = fval[j]; best = (int)j; } }      // first maximum (vector.c:26-46)
            else if (fval[j] > mu) mu = fval[j];
        }
        if (mu < 0 || (best >= 0 && bv > mu)) {
            if (fault == 2) {                                      // tests only: a WRONG greedy step (the second best known one)
                int second = -1; double sv = -1;
                for (size_t j = 0; j < count; j++) if ((int)j != best && fkn[j] && fval[j] > sv) { sv = fval[j]; second = (int)j; }
                if (second >= 0 && sv >= kDelta && sv < bv) { best = second; bv = sv; }
            }
            if (!(bv >= kDelta)) { out.stopped_early = true; break; }          // benefit < DELTA: stop (local.c:365-370)
            const int id = result[best], l = rec[id].left;
            out.commits.push_back(LqCommit{best, id, (int)count, l});
            result[count] = l; result[best] = l + 1;               // local.c:375-376: palette ORDER
            fval[count] = rec[l].val; fkn[count] = (char)rec[l].kn;
            fval[best] = rec[l + 1].val; fkn[best] = (char)rec[l + 1].kn;
            count++;
            continue;
        }
        if (std::max(best >= 0 ? bv : 0.0, mu) < kDelta) { out.stopped_early = true; break; }
        return false;
    }
    result.resize(count);
    return true;
}

static bool lq_replay(const LqRec *rec, int kbase, int first_base, size_t K, LqReplay &out) {
    std::vector<int> &result = out.result;
    result.assign(K, -1);
    for (int j = 0; j < kbase; j++) result[j] = first_base + j;
    size_t count = (size_t)kbase;
    // The frontier in blocks of sixteen rows, each with its first maximum among the known rows and the maximum among the unknown
    // ones: a step changes two rows, so it rescans two blocks and the blocks' summaries instead of the whole frontier (254 steps over
    // up to 256 rows: 50 us of a 1920x1080 call's 1.6 ms before)
    constexpr size_t B = 16;
    const size_t nb = (K + B - 1) / B;
    std::vector<double> fval(nb * B, 0.0);
    std::vector<char> fkn(nb * B, 0);
    std::vector<double> bbv(nb, 0.0), bmu(nb, -1.0);
    std::vector<int> bbest(nb, -1);
    auto rescan = [&](const size_t b) {
        int best = -1; double bv = 0, mu = -1;
        const size_t lo = b * B, hi = std::min(count, lo + B);
        for (size_t j = lo; j < hi; j++) {
            if (fkn[j]) { if (best < 0 || fval[j] > bv) { bv = fval[j]; best = (int)j; } }      // first maximum (vector.c:26-46)
            else if (fval[j] > mu) mu = fval[j];
        }
        bbest[b] = best; bbv[b] = bv; bmu[b] = mu;
    };
    for (size_t j = 0; j < count; j++) { fval[j] = rec[result[j]].val; fkn[j] = (char)rec[result[j]].kn; }
    for (size_t b = 0; b < nb; b++) rescan(b);
    const int fault = g_debug_fault.load(std::memory_order_relaxed);
    while (count < K) {
        int best = -1; double bv = 0, mu = -1;
        const size_t nbu = (count + B - 1) / B;
        for (size_t b = 0; b < nbu; b++) {                       // ascending blocks, strict '>': the first maximum of all rows
            if (bbest[b] >= 0 && (best < 0 || bbv[b] > bv)) { bv = bbv[b]; best = bbest[b]; }
            if (bmu[b] > mu) mu = bmu[b];
        }
        if (mu < 0 || (best >= 0 && bv > mu)) {
            if (fault == 2) {                                      // tests only: a WRONG greedy step (the second best known one)
                int second = -1; double sv = -1;
                for (size_t j = 0; j < count; j++) if ((int)j != best && fkn[j] && fval[j] > sv) { sv = fval[j]; second = (int)j; }
                if (second >= 0 && sv >= kDelta && sv < bv) { best = second; bv = sv; }
            }
            if (!(bv >= kDelta)) { out.stopped_early = true; break; }          // benefit < DELTA: stop (local.c:365-370)
            const int id = result[best], l = rec[id].left;
            out.commits.push_back(LqCommit{best, id, (int)count, l});
            result[count] = l; result[best] = l + 1;               // local.c:375-376: palette ORDER
            fval[count] = rec[l].val; fkn[count] = (char)rec[l].kn;
            fval[best] = rec[l + 1].val; fkn[best] = (char)rec[l + 1].kn;
            count++;
            rescan((size_t)best / B);
            if ((count - 1) / B != (size_t)best / B) rescan((count - 1) / B);
            continue;
        }
        if (std::max(best >= 0 ? bv : 0.0, mu) < kDelta) { out.stopped_early = true; break; }
        return false;
    }
    result.resize(count);
    return true;
}

// patolette_amd_set_split_loop: 2 (default) the device-driven loop for images below kLqDeviceAutoPixels, 1 wherever it applies,
// 0 the host-driven one everywhere.  Measured, round 6 (profiles/r06_split_loop_ab.txt), per call, device against host-driven:
// first version 1920x1080 1.61 / 1.77 ms, 4096^2 5.81 / 5.80, 8192^2 21.7 / 21.0 (every leaf that could reach tau was evaluated at
// once: ~500 evaluations where the host's lock-step pruning makes ~320); with the global quantiser on the device too and the
// host loop's own waiting rule in k_lq_select (same ~320 evaluations): 1.38 / 1.53, 5.28-5.46 / 5.66-5.87, 21.4-21.5 / 21.2-21.7.
std::atomic<int> g_lq_device{getenv("PAMD_LQ_DEVICE") ? atoi(getenv("PAMD_LQ_DEVICE")) : 2};
// patolette_amd_set_global_quantiser: 1 (default) the global quantiser's decisions on the device (k_gq_control), 0 the host's turn
std::atomic<int> g_gq_device{getenv("PAMD_GQ_DEVICE") ? atoi(getenv("PAMD_GQ_DEVICE")) : 1};
constexpr size_t kLqDeviceAutoPixels = (size_t)40 << 20;

// The local quantiser driven from the device (k_lq_children / k_lq_select above).  In: the base clusters are nodes first_base ..
// first_base + kbase - 1 of the table with their segments, means and moment accumulators complete on the stream (no synchronisation
// yet).  Returns 0 (centres, stats filled in; the trace on request), -2 if the candidate tree outgrew the device's table (the caller
// starts over on the host loop).
static int lq_device_loop(Engine &E, size_t N, size_t K, bool weighted, bool inv_sums, const Bounds &bnd, const QuantBuffers &qlq, int kbase,
                          int first_base, int nnodes, bool snake, std::vector<double> &centers, size_t &len, unsigned long long *max_members) {
    hipStream_t s = E.stream;
    const size_t lqs = (size_t)kNQ_LQ * 2 * kBuckets;
    const int ntA_ub = (int)ceil_div(N, (size_t)kTileA) + kLqRoundCap, ntP_ub = (int)ceil_div(N, (size_t)kTileP) + kLqRoundCap;
    E.lqctl.reserve(1); E.h_lqhead.reserve(1); E.h_lqrec.reserve(kLqNodeCap); E.h_lqcen.reserve(kLqNodeCap);
    E.hist.reserve(std::max(hist_slot_doubles(), lqs * kLqRoundCap)); E.hsize.reserve((size_t)kLqRoundCap * kBuckets);
    E.hcount.reserve((size_t)kLqRoundCap * kBuckets); E.lut.reserve((size_t)kLqRoundCap * kBuckets);
    E.tilesA.reserve(ntA_ub); E.tilesP.reserve(ntP_ub);
    E.tilecnt.reserve((size_t)ntP_ub * kMaxChildren); E.tileoff.reserve((size_t)ntP_ub * kMaxChildren);
    LqCtl *c = E.lqctl.p;
    LqHead *head = E.h_lqhead.p;
    for (int r = 0; r < kLqMaxRounds; r++) { head->round_px[r] = 0.0; head->round_nr[r] = 0.0; }
    const int eigen_bound = (kUseEigenBound && g_lq_eigen_bound) ? 1 : 0;
    const RoundDyn *dyn = &c->dyn;
    const int *d_round = c->round_ids, *d_tP0 = c->tP0;
    int call = 0;                                                  // control calls so far: the parity selects the children list (LqCtl::cids)
    static const double spec_beta_dev = getenv("PAMD_SPEC_BETA_DEV") ? atof(getenv("PAMD_SPEC_BETA_DEV")) : 0.25;   // 0: every leaf that reaches tau is evaluated at once
    // kbase < 0: the global quantiser ran on the device too (k_gq_control): the count of base clusters is in E.gqout, twelve at most
    const GqOut *gq = kbase < 0 ? (const GqOut *)E.gqout.p : nullptr;
    const int kb_ub = kbase < 0 ? kGqMaxK : kbase;
    auto control = [&]() {
        const bool first = call == 0;
        {
            KTIME("k_lq_children", s, 0.0);
            if (first) hipLaunchKernelGGL(k_lq_children, (kb_ub + 3) / 4, 256, 0, s, E.nodes.p, c, call, eigen_bound, kb_ub, first_base, nnodes, (int)K, gq);
            else hipLaunchKernelGGL(k_lq_children, 2 * kLqRoundCap / 4, 256, 0, s, E.nodes.p, c, call, eigen_bound, 0, 0, 0, 0, (const GqOut *)nullptr);
        }
        KTIME("k_lq_select", s, 0.0);
        hipLaunchKernelGGL(k_lq_select, 1 + (first ? (kb_ub + 3) / 4 : 2 * kLqRoundCap / 4), 256, 0, s, E.nodes.p, c, call, bnd.e_lin, bnd.e_quad, spec_beta_dev);
        HIP_CHECK(hipGetLastError());
        call++;
    };
    int enq = 0;                                                   // rounds enqueued so far
    auto round = [&]() {
        if (enq >= kLqMaxRounds) throw HipError("patolette_amd: split loop exceeded its round limit");
        const double *px_src = &head->round_px[enq], *nr_src = &head->round_nr[enq];
        const bool rev = snake && (enq % 2 == 1);                  // the sweeps alternate their direction (see the host loop)
        // the round's set-up (tile lists, cleared bucket tables) rides on its first sweep; PAMD_LQ_SETUP_KERNEL=1: a launch of its own (A/B)
        static const bool setup_kernel = getenv("PAMD_LQ_SETUP_KERNEL") && atoi(getenv("PAMD_LQ_SETUP_KERNEL")) != 0;
        if (setup_kernel) {
            hipLaunchKernelGGL(k_round_setup_dyn, 1024, 256, 0, s, E.nodes.p, (const LqCtl *)c, E.tilesA.p, E.tilesP.p, E.hist.p, E.hsize.p, E.hcount.p, lqs);
            HIP_CHECK(hipGetLastError());
            launch_minmax(qlq, E.tilesA.p, ntA_ub, 0, E.nodes.p, s, rev, dyn, px_src);
        } else {
            const RoundLists rl{c->round_ids, c->tA0, c->tP0, E.tilesA.p, E.tilesP.p, E.hist.p, lqs, E.hsize.p, E.hcount.p};
            launch_minmax(qlq, E.tilesA.p, ntA_ub, 0, E.nodes.p, s, rev, dyn, px_src, &rl);
        }
        launch_hist(qlq, false, E.tilesA.p, ntA_ub, 0, E.nodes.p, E.hist.p, E.hsize.p, E.hcount.p, s, snake && !rev, false, dyn, px_src);
        launch_cut(weighted, E.nodes.p, d_round, kLqRoundCap, E.hist.p, E.hsize.p, E.hcount.p, E.lut.p, s, dyn, nr_src);
        launch_partition(qlq, E.tilesP.p, ntP_ub, 0, d_round, d_tP0, kLqRoundCap, E.nodes.p, E.lut.p, E.tilecnt.p, E.tileoff.p, true, s, inv_sums, rev, dyn, px_src);
        enq++;
        control();
    };
    control();                                                     // the base clusters' moments and bounds, round 1
    int want = (E.lq_hint_N == N && E.lq_hint_K == K && E.lq_hint_rounds > 0) ? E.lq_hint_rounds : 9;
    for (;;) {
        while (enq < want) round();
        hipLaunchKernelGGL(k_lq_export, 16, 256, 0, s, (const LqCtl *)c, head, E.h_lqrec.p, E.h_lqcen.p);
        HIP_CHECK(hipGetLastError());
        E.sync();
        if (head->done) break;
        want = enq + 2;
    }
    if (head->error == 2) return -2;
    if (head->error == 3) return -1;                               // the global quantiser's eigen-solve failed (global.c:214-217)
    if (kbase < 0) kbase = E.h_gqout.p->kbase;
    E.lq_hint_N = N; E.lq_hint_K = K; E.lq_hint_rounds = head->rounds;
    LqReplay rp;
    const double t_rp0 = now_ms();
    // the records as the device left them in pinned memory are cache misses one by one (the replay hops from node to node: 36 us for
    // 254 steps); one sequential copy brings them in (18-28 us)
    E.lq_rec_local.resize((size_t)head->nnodes);
    std::memcpy(E.lq_rec_local.data(), E.h_lqrec.p, (size_t)head->nnodes * sizeof(LqRec));
    const bool replay_ok = lq_replay(E.lq_rec_local.data(), kbase, first_base, K, rp);
    const double t_rp1 = now_ms();
    static const bool replay_check = getenv("PAMD_LQ_REPLAY_CHECK") != nullptr;
    if (replay_check || !replay_ok) {
        LqReplay ref;
        const bool ref_ok = lq_replay_plain(E.lq_rec_local.data(), kbase, first_base, K, ref);
        if (ref_ok != replay_ok || ref.result != rp.result || ref.stopped_early != rp.stopped_early)
            fprintf(stderr, "patolette_amd: split loop: the two replays DISAGREE (plain %d, %zu rows; blocked %d, %zu rows), kbase %d, K %zu, nodes %d, rounds %d\n",
                    (int)ref_ok, ref.result.size(), (int)replay_ok, rp.result.size(), kbase, K, head->nnodes, head->rounds);
        else if (!replay_ok)
            fprintf(stderr, "patolette_amd: split loop: both replays blocked after %zu commits, kbase %d, K %zu, nodes %d, rounds %d, evaluated %d, tau %.17g\n",
                    rp.commits.size(), kbase, K, head->nnodes, head->rounds, head->neval, head->tau);
    }
    if (!replay_ok) throw HipError("patolette_amd: split loop: the replay met an unevaluated node");
    len = rp.result.size();
    centers.assign(3 * len, 0.0);
    unsigned long long mg = 0;
    for (size_t i = 0; i < len; i++) {
        const LqCen &cn = E.h_lqcen.p[rp.result[i]];
        for (int j = 0; j < 3; j++) centers[(size_t)j * len + i] = cn.mean[j];      // create.c:11-33
        mg = std::max(mg, (unsigned long long)cn.gn);
    }
    if (max_members) *max_members = mg;
    E.stats.split_evals = (size_t)head->split_evals; E.stats.split_px = (size_t)head->split_px; E.stats.lq_rounds = (size_t)head->rounds;
    E.lq_commits.swap(rp.commits);
    E.trace.clear();
    E.trace_pending = true;
    E.trace_hdr.stopped_early = rp.stopped_early ? 1 : 0;
    static const bool lq_times = getenv("PAMD_LQ_TIMES") != nullptr;
    if (lq_times) {
        fprintf(stderr, "patolette_amd: device-driven split loop: %d rounds, %d evaluated, %zu commits, tau %.3g; host replay %.1f us\n", head->rounds, head->neval,
                E.lq_commits.size(), head->tau, 1e3 * (t_rp1 - t_rp0));
        if (gq) {
            const unsigned long long *t = E.h_gqout.p->ticks;
            fprintf(stderr, "patolette_amd: k_gq_control us: prefixes %.1f, axis %.1f, search %.1f, records %.1f\n", (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01,
                    (t[3] - t[2]) * 0.01, (t[4] - t[3]) * 0.01);
            fprintf(stderr, "patolette_amd: k_gq_control us: of the prefixes, staging %.1f\n", (t[5] - t[0]) * 0.01);
        }
    }
    return 0;
}

// the split trace of a device-driven call, fetched when asked for (the node table stays on the device until the engine's next call)
static void materialise_trace(Engine &E) {
    if (!E.trace_pending) return;
    E.trace_pending = false;
    const int n = (int)E.lq_commits.size();
    E.trace.assign(n, patolette_amd__SplitRecord{});
    if (!n) return;
    DevBuf<LqCommit> dc; DevBuf<patolette_amd__SplitRecord> dr;
    dc.reserve(n); dr.reserve(n);
    HIP_CHECK(hipMemcpyAsync(dc.p, E.lq_commits.data(), sizeof(LqCommit) * n, hipMemcpyHostToDevice, E.stream));
    hipLaunchKernelGGL(k_lq_trace, (n + 63) / 64, 64, 0, E.stream, (const NodeDev *)E.nodes.p, (const LqCommit *)dc.p, n, dr.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(E.trace.data(), dr.p, sizeof(patolette_amd__SplitRecord) * n, hipMemcpyDeviceToHost, E.stream));
    HIP_CHECK(hipStreamSynchronize(E.stream));
}

static int quantize_clusters_run(Engine &E, size_t N, size_t K, bool weighted, const Bounds &bnd,
                                 std::vector<double> &centers, size_t &len, bool verbose, unsigned long long *max_members, bool allow_device) {
    hipStream_t s = E.stream;
    const size_t planes = weighted ? 4 : 3;
    const Shard *sh = E.shard;                                  // the image is dealt out over a group of GPUs: N is this GPU's part
    const size_t Nt = sh ? sh->total : N;                       // pixels of the whole image
    const bool inv_sums = sh || E.invariant;
    if (E.prep_N != N || E.prep_planes != planes) gq_prepare(E, N, weighted);   // (the stage-level entry points come here unprepared)
    E.prep_N = 0;
    // PAMD_LQ_DEVICE=0: the host-driven split loop everywhere (A/B, and what sliced images, K > 256 and verbose calls always take)
    const int lq_mode = g_lq_device.load(std::memory_order_relaxed);
    const bool dev_eligible = allow_device && (lq_mode == 1 || (lq_mode == 2 && N < kLqDeviceAutoPixels)) && !sh && !verbose && K <= (size_t)kLqDevMaxK;
    E.nodes.reserve(std::max<size_t>(4 * K + 64, dev_eligible ? (size_t)kLqNodeCap : 0));
    std::vector<HNode> hn;
    hn.reserve(4 * K + 64);
    double t0 = now_ms();

    // ---------------- global quantiser (global.c:388-443) ----------------
    // unweighted PCA of all pixels: mean, centred covariance, dsyev
    HNode root; root.begin = 0; root.n = N; root.gn = Nt; root.buf = 0; root.sw = (double)Nt;
    E.h_dbl.reserve(16 * kBuckets * 2 + 64);
    if (bnd.have_sum) {
        const double inv = 1 / (double)Nt;                      // matrix2D.c:229; the sums came with the conversion pass
        for (int j = 0; j < 3; j++) root.mean[j] = bnd.sum[j] * inv;
    } else {
        int rootP = 1; while ((1ULL << rootP) < (Nt > 1 ? Nt : 2)) rootP++;
        E.sum6.reserve(kSum3Slots * 6);
        launch_sum3(E.cvt.p, N, make_bink(bnd.e_lin, rootP), E.sum6.p, s);
        if (sh) comm_sum_dev(E, E.sum6.p, kSum3Slots * 6, 0);
        HIP_CHECK(hipMemcpyAsync(E.h_dbl.p, E.sum6.p, kSum3Slots * 6 * sizeof(double), hipMemcpyDeviceToHost, s));
        E.sync();
        const double inv = 1 / (double)Nt;                      // matrix2D.c:229
        for (int j = 0; j < 3; j++) {
            double p0 = 0, p1 = 0;                              // slot partials are exact multiples of the bin grids
            for (int sl = 0; sl < kSum3Slots; sl++) { p0 += E.h_dbl.p[sl * 6 + 2 * j]; p1 += E.h_dbl.p[sl * 6 + 2 * j + 1]; }
            root.mean[j] = (p0 + p1) * inv;
        }
    }
    hn.push_back(root);                                         // id 0
    QuantBuffers qroot{{E.cvt.p, E.bufA.p}, E.bkt.p, N, weighted};
    QuantBuffers qlq{{E.bufB.p, E.bufA.p}, E.bkt.p, N, weighted};
    std::vector<int> round = {0};
    const int ntA0 = (int)ceil_div(N, (size_t)kTileA), ntP0 = (int)ceil_div(N, (size_t)kTileP);   // the root's tilings (gq_prepare)
    // every sweep over the pixels starts where the previous one stopped (see the split rounds below): the conversion wrote the
    // image front to back, so the root's moments are taken back to front, the extrema front to back, ...
    static const bool snake = !(getenv("PAMD_SWEEP_SNAKE") && atoi(getenv("PAMD_SWEEP_SNAKE")) == 0);
    std::vector<NodeOut> got;
    bool mom_path = bnd.have_mom && !sh;                        // one sweep less: the directions of the following ones flip
    if (mom_path) {
        // The conversion pass took the column sums S1 and the raw second moments S2 as exact pairs: the centred sums
        // S2_jk - S1_j S1_k / n (pca.c:62-101 about the mean, matrix2D.c:200-233) follow in extended precision, then ONE rounding
        // to double -- instead of a sweep over the image and a round trip.  What the shortcut cannot undo: every product x_j x_k was
        // rounded to double before it was binned (relative 2^-53, systematic for a flat image), so the difference is good to
        // ~1e-16 of S2 and no better: an image whose spread is tiny against its mean -- trace(cov) below 1e-6 of trace(S2), i.e.
        // fewer than ~10 digits left --, or a difference that came out negative, takes the centred sweep below instead.
        const long double nn = (long double)Nt;
        long double s1[3], c6[6], s2d[3] = {0, 0, 0};
        for (int j = 0; j < 3; j++) s1[j] = (long double)bnd.sum2[j][0] + (long double)bnd.sum2[j][1];
        static const int ja[6] = {0, 1, 2, 1, 2, 2}, jb[6] = {0, 0, 0, 1, 1, 2};       // xx, yx, zx, yy, zy, zz
        for (int q = 0; q < 6; q++) {
            const long double s2 = (long double)bnd.mom2[q][0] + (long double)bnd.mom2[q][1];
            if (ja[q] == jb[q]) s2d[ja[q]] = s2;
            c6[q] = s2 - s1[ja[q]] * s1[jb[q]] / nn;
        }
        const long double tr = c6[0] + c6[3] + c6[5], tr2 = s2d[0] + s2d[1] + s2d[2];
        static const bool guard = !(getenv("PAMD_ROOT_MOMENTS_GUARD") && atoi(getenv("PAMD_ROOT_MOMENTS_GUARD")) == 0);
        if (guard && (!(tr > 1e-6L * tr2) || c6[0] < 0 || c6[3] < 0 || c6[5] < 0)) mom_path = false;
        else {
            for (int q = 0; q < 6; q++) hn[0].cov6[q] = (double)c6[q];
            hn[0].dist = (double)tr;
        }
    }
    if (!mom_path) {
        NodeIn d = make_nodedev(hn[0], bnd);
        put_nodes(E, {0}, {d});
        launch_cov_nodes(qroot, E.cvt.p, E.tilesA.p, ntA0, N, E.nodes.p, s, snake);
        if (sh) shard_exchange_acc(E, shard_upload_ids(E, {0}), 1);
        get_nodes(E, {0}, got);
        absorb_moments(hn[0], got[0]);
    }
    double axis[3];
    if (!node_axis(hn[0], axis)) return -1;
    E.trace.clear();
    E.trace_pending = false;
    E.trace_hdr = patolette_amd__SplitTrace{};
    for (int j = 0; j < 3; j++) E.trace_hdr.gq_axis[j] = axis[j];
    for (int q = 0; q < 6; q++) E.trace_hdr.gq_cov6[q] = hn[0].cov6[q] / hn[0].sw;

    // projection, 512 buckets, cell moments (sort.c, cells.c:53-139)
    {
        NodeIn d = make_nodedev(hn[0], bnd);
        for (int j = 0; j < 3; j++) d.axis[j] = axis[j];
        d.slot = 0;
        put_nodes(E, {0}, {d});
    }
    const size_t hs = hist_slot_doubles();
    launch_minmax(qroot, E.tilesA.p, ntA0, N, E.nodes.p, s, snake && mom_path);
    if (sh) shard_exchange_keys(E, shard_upload_ids(E, {0}), 1);
    launch_hist(qroot, true, E.tilesA.p, ntA0, N, E.nodes.p, E.hist.p, E.hsize.p, E.hcount.p, s, snake && !mom_path, Nt >= ((size_t)1 << 18));
    if (sh) { comm_sum_dev(E, E.hist.p, hs, 0); comm_sum_dev(E, E.hcount.p, kBuckets, 2); }
    // The quantiser's decisions (global.c:189-298) on the device, the partition behind them without the host looking (k_gq_control):
    // one GPU, palettes of more than twelve colours (so that base clusters < K whatever the image), not verbose.  PAMD_GQ_DEVICE=0:
    // the host's turn everywhere (A/B; what sliced images, small palettes and verbose calls always take)
    const bool gq_dev = g_gq_device.load(std::memory_order_relaxed) != 0 && !sh && !verbose && K > (size_t)kGqMaxK;
    int kbase = 0;
    std::vector<int> base_ids;
    if (gq_dev) {
        E.gq.reserve(1); E.gqout.reserve(1); E.h_gqout.reserve(1);
        *E.h_gqout.p = GqOut{};
        {
            KTIME("k_gq_control", s, (double)hs * 8);
            hipLaunchKernelGGL(k_gq_control, 1, 1024, 0, s, (const double *)E.hist.p, (const unsigned int *)E.hcount.p, E.gq.p, (int)std::min<size_t>(K, (size_t)1 << 30),
                               weighted ? 1 : 0, E.nodes.p, 1, E.lut.p, E.gqout.p, E.h_gqout.p);
            HIP_CHECK(hipGetLastError());
        }
        launch_partition(qroot, E.tilesP.p, ntP0, N, E.round_nodes.p, E.node_tile0.p, 1, E.nodes.p, E.lut.p, E.tilecnt.p, E.tileoff.p, true, s, inv_sums,
                         snake && mom_path, nullptr, nullptr, true);
        launch_cov_children(qroot, E.tilesA.p, ntA0, N, E.nodes.p, s, snake && !mom_path, true);
        auto trace_header = [&]() {                              // (after a synchronisation)
            const GqOut &o = *E.h_gqout.p;
            E.trace_hdr.n_base = o.kbase;
            for (int j = 0; j <= o.kbase && j < 14; j++) E.trace_hdr.gq_cuts[j] = (size_t)o.cuts[j];
        };
        if (dev_eligible) {
            E.stats.ms_gq = now_ms() - t0;
            t0 = now_ms();
            const int rc = lq_device_loop(E, N, K, weighted, inv_sums, bnd, qlq, -1, 1, 0, snake, centers, len, max_members);
            trace_header();
            E.stats.n_base_clusters = (size_t)E.h_gqout.p->kbase;
            if (rc != 0) return rc;
            E.stats.n_clusters = len;
            E.trace_hdr.n_clusters = (int32_t)len; E.trace_hdr.n_records = (int32_t)E.lq_commits.size();
            E.cluster_centers = centers;
            E.stats.ms_lq = now_ms() - t0;
            return 0;
        }
        // the host-driven split loop: the count comes down with the base clusters' records (the first kbase of twelve)
        std::vector<int> all(kGqMaxK);
        for (int j = 0; j < kGqMaxK; j++) all[j] = 1 + j;
        get_nodes(E, all, got);
        if (E.h_gqout.p->error) return -1;
        trace_header();
        kbase = E.h_gqout.p->kbase;
        for (int j = 0; j < kbase; j++) {
            HNode c; c.buf = 1; c.begin = got[j].begin; c.n = got[j].n; c.gn = got[j].gn; c.sw = got[j].sw;
            for (int q = 0; q < 3; q++) c.mean[q] = got[j].mean[q];
            absorb_moments(c, got[j]);
            leaf_bound(c);
            base_ids.push_back((int)hn.size());
            hn.push_back(c);
        }
    } else {
    const int gq_kmax = (int)std::min<size_t>(K, kGqMaxK);
    E.gq.reserve(1); E.h_gq.reserve(1);
    launch_gq_dp(E.hist.p, E.hcount.p, gq_kmax, E.gq.p, s);
    HIP_CHECK(hipMemcpyAsync(E.h_gq.p->cut, E.gq.p->cut, sizeof(E.h_gq.p->cut), hipMemcpyDeviceToHost, s));
    std::vector<double> hh(hs);
    std::vector<unsigned int> hc(kBuckets);
    HIP_CHECK(hipMemcpyAsync(E.h_dbl.p, E.hist.p, hs * sizeof(double), hipMemcpyDeviceToHost, s));
    E.h_round.reserve(kBuckets);
    HIP_CHECK(hipMemcpyAsync(E.h_round.p, E.hcount.p, kBuckets * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    E.sync();
    std::memcpy(hh.data(), E.h_dbl.p, hs * sizeof(double));
    std::memcpy(hc.data(), E.h_round.p, kBuckets * sizeof(unsigned int));
    auto H = [&](int q, int b) { return hh[(size_t)(q * 2 + 0) * kBuckets + b] + hh[(size_t)(q * 2 + 1) * kBuckets + b]; };

    auto cm = std::make_unique<hm::CellMoments>();
    std::memset(cm.get(), 0, sizeof(hm::CellMoments));
    for (int b = 0; b < kBuckets; b++) {
        cm->w0[b + 1] = hc[b];
        for (int r = 0; r < 3; r++) cm->w1[r][b + 1] = H(r, b);
        cm->w2[b + 1] = H(3, b);
        for (int q = 0; q < 6; q++) cm->wrs[q][b + 1] = H(4 + q, b);
    }
    for (int i = 1; i <= kBuckets; i++) {                       // cells.c:114-136 sequential prefix
        cm->w0[i] += cm->w0[i - 1]; cm->w2[i] += cm->w2[i - 1];
        for (int r = 0; r < 3; r++) cm->w1[r][i] += cm->w1[r][i - 1];
        for (int q = 0; q < 6; q++) cm->wrs[q][i] += cm->wrs[q][i - 1];
    }
    std::vector<size_t> cuts = hm::gq_principal_quantizer(K, *cm, E.h_gq.p->cut);
    if (cuts.size() < 2) return -1;
    kbase = (int)cuts.size() - 1;
    E.trace_hdr.n_base = kbase;
    for (int j = 0; j <= kbase && j < 14; j++) E.trace_hdr.gq_cuts[j] = cuts[j];

    // base clusters: bucket b belongs to the first cell j with b+1 <= q[j+1] (global.c:328-335)
    std::vector<unsigned char> lut(kBuckets);
    for (int b = 0; b < kBuckets; b++) {
        int j = 0;
        while (j < kbase - 1 && !((size_t)(b + 1) <= cuts[j + 1])) j++;
        lut[b] = (unsigned char)j;
    }
    // Two base clusters (what noise and most photographs give): the partition is a binary split like the local quantiser's, so
    // it takes the same pipelined kernel and the children's centred moments ride along instead of costing a sweep of their own
    const bool gq_binary = kbase == 2;
    {
        unsigned long long pos = 0;
        for (int j = 0; j < kbase; j++) {
            HNode c; c.buf = 1; c.begin = pos;
            unsigned long long cnt = 0;
            double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};      // exact: parts lie on the bin grids
            for (int b = 0; b < kBuckets; b++) if (lut[b] == j) {
                cnt += hc[b];
                for (int q = 0; q < (weighted ? 4 : 3); q++) {
                    int qi = weighted ? 10 + q : q;
                    s0[q] += hh[(size_t)(qi * 2 + 0) * kBuckets + b];
                    s1[q] += hh[(size_t)(qi * 2 + 1) * kBuckets + b];
                }
            }
            c.n = cnt; c.gn = cnt;                                   // sliced image: the segment on this GPU follows from the partition
            c.sw = weighted ? (s0[3] + s1[3]) : (double)cnt;
            const double inv = 1 / c.sw;
            for (int q = 0; q < 3; q++) c.mean[q] = (s0[q] + s1[q]) * inv;
            pos += cnt;
            base_ids.push_back((int)hn.size());
            hn.push_back(c);
        }
    }
    {
        NodeIn d = make_nodedev(hn[0], bnd);
        for (int j = 0; j < 3; j++) d.axis[j] = axis[j];
        d.slot = 0; d.child0 = base_ids[0]; d.nchild = kbase;
        if (gq_binary) d.split = (int)cuts[1] - 1;               // bucket b is in cluster 0 iff b + 1 <= cuts[1] (global.c:328-335)
        std::vector<int> ids = {0};
        std::vector<NodeIn> recs = {d};
        for (int id : base_ids) { ids.push_back(id); NodeIn c = make_nodedev(hn[id], bnd); c.klin = d.klin; c.kquad = d.kquad; recs.push_back(c); }
        put_nodes(E, ids, recs);
    }
    E.h_bytes.reserve(kBuckets);                                // pinned: no synchronisation before the partition
    std::memcpy(E.h_bytes.p, lut.data(), kBuckets);
    HIP_CHECK(hipMemcpyAsync(E.lut.p, E.h_bytes.p, kBuckets, hipMemcpyHostToDevice, s));
    launch_partition(qroot, E.tilesP.p, ntP0, N, E.round_nodes.p, E.node_tile0.p, 1, E.nodes.p, E.lut.p, E.tilecnt.p, E.tileoff.p, gq_binary, s, inv_sums, snake && mom_path);
    if (sh) { hipLaunchKernelGGL(k_shard_children_local, 1, 64, 0, s, E.nodes.p, E.round_nodes.p, 1); HIP_CHECK(hipGetLastError()); }
    if (!gq_binary) launch_cov_children(qroot, E.tilesA.p, ntA0, N, E.nodes.p, s, snake && !mom_path);
    if (sh) shard_exchange_acc(E, shard_upload_ids(E, base_ids), (int)base_ids.size());
    if (dev_eligible && (size_t)kbase < K) {
        // the split loop runs from the device: no synchronisation here, the control kernel takes the base clusters' moments itself
        E.stats.n_base_clusters = (size_t)kbase;
        E.stats.ms_gq = now_ms() - t0;
        t0 = now_ms();
        const int rc = lq_device_loop(E, N, K, weighted, inv_sums, bnd, qlq, kbase, base_ids[0], (int)hn.size(), snake, centers, len, max_members);
        if (rc != 0) return rc;
        E.stats.n_clusters = len;
        E.trace_hdr.n_clusters = (int32_t)len; E.trace_hdr.n_records = (int32_t)E.lq_commits.size();
        E.cluster_centers = centers;
        E.stats.ms_lq = now_ms() - t0;
        return 0;
    }
    get_nodes(E, base_ids, got);
    for (size_t i = 0; i < base_ids.size(); i++) {
        absorb_moments(hn[base_ids[i]], got[i]);
        if (sh) { hn[base_ids[i]].begin = got[i].begin; hn[base_ids[i]].n = got[i].n; }
        leaf_bound(hn[base_ids[i]]);
    }
    }                                                           // (the host's turn of the global quantiser)
    E.stats.n_base_clusters = (size_t)kbase;
    if (verbose) printf("patolette ======== Base cluster count: %zu\n", (size_t)kbase);     // patolette.c:227-229
    E.stats.ms_gq = now_ms() - t0;
    t0 = now_ms();

    // ---------------- local quantiser (local.c:318-404) ----------------
    std::vector<int> result(base_ids);                          // frontier in the reference's order
    std::vector<int> leaves(base_ids);                          // candidate-tree nodes with moments but no split yet
    static const double spec_beta = getenv("PAMD_SPEC_BETA") ? atof(getenv("PAMD_SPEC_BETA")) : 0.25;   // swept 1/256 .. 1 on six kinds of content: 1/4 evaluates least (noise 514 -> 319 splits, lq -8 %)
    size_t count = result.size();
    E.stats.split_evals = 0; E.stats.split_px = 0; E.stats.lq_rounds = 0;
    auto known = [&](const HNode &h) { return h.nosplit || h.gn <= 1 || h.split_done; };
    auto benefit = [&](const HNode &h) -> double {
        if (h.nosplit || h.gn <= 1) return 0;                    // children == NULL -> 0 (local.c:262-264)
        return h.dist - (hn[h.left].dist + hn[h.right].dist);
    };
    if (count < K) {
        result.resize(K, -1);
        // the frontier as two flat arrays (benefit if the node's split is known, else its bound): the greedy steps between two
        // rounds scan them a few hundred times
        std::vector<double> fval(K, 0.0);
        std::vector<char> fkn(K, 0);
        auto refresh = [&](const size_t j) {
            const HNode &h = hn[result[j]];
            fkn[j] = known(h) ? 1 : 0;
            fval[j] = fkn[j] ? benefit(h) : h.ub;
        };
        for (size_t j = 0; j < count; j++) refresh(j);
        static const bool lq_times = getenv("PAMD_LQ_TIMES") != nullptr;       // diagnostic: where the host's turn between two rounds goes
        double tm_greedy = 0, tm_select = 0, tm_axes = 0, tm_packet = 0, tm_enqueue = 0, tm_wait = 0, tm_children = 0, tm_mark = now_ms();
        auto lap = [&](double &acc) { if (lq_times) { const double t = now_ms(); acc += t - tm_mark; tm_mark = t; } };
        for (;;) {
            if (count >= K) break;
            // one greedy step, exact whenever every undecided node is provably not the arg-max
            int best = -1; double bv = 0; double max_unknown = -1;
            for (size_t j = 0; j < count; j++) {
                if (fkn[j]) {
                    if (best < 0 || fval[j] > bv) { bv = fval[j]; best = (int)j; }
                } else if (fval[j] > max_unknown) max_unknown = fval[j];
            }
            // first maximum among ALL entries = first maximum among the known ones iff every unknown
            // benefit (<= that node's distortion) is strictly below it
            if (max_unknown < 0 || (best >= 0 && bv > max_unknown)) {
                if (g_debug_fault.load(std::memory_order_relaxed) == 2) {   // tests only: a WRONG greedy step (the second best known one)
                    int second = -1; double sv = -1;
                    for (size_t j = 0; j < count; j++) if ((int)j != best && fkn[j] && fval[j] > sv) { sv = fval[j]; second = (int)j; }
                    if (second >= 0 && sv >= kDelta && sv < bv) { best = second; bv = sv; }
                }
                if (!(bv >= kDelta)) { E.trace_hdr.stopped_early = 1; break; }   // benefit < DELTA: stop, keep `count` clusters
                const HNode &h = hn[result[best]];
                const int l = h.left, r = h.right;
                {
                    patolette_amd__SplitRecord tr{};
                    tr.row = best; tr.new_row = (int32_t)count;
                    tr.split = hn[l].psplit < 0 ? -1 : (hn[l].psplit & 0xffff); tr.degenerate = hn[l].psplit < 0 ? 0 : (hn[l].psplit >> 16) & 1;
                    tr.n = h.gn; tr.n_left = hn[l].gn; tr.n_right = hn[r].gn; tr.sw = h.sw;
                    for (int j = 0; j < 3; j++) tr.axis[j] = h.axis[j];
                    for (int q = 0; q < 6; q++) tr.cov6[q] = h.cov6[q] / h.sw;
                    tr.dist = h.dist; tr.dist_left = hn[l].dist; tr.dist_right = hn[r].dist; tr.benefit = bv;
                    E.trace.push_back(tr);
                }
                result[count] = l;                              // local.c:375-376: palette ORDER
                result[best] = r;
                refresh(count); refresh((size_t)best);
                count++;
                if (verbose) { printf("patolette ======== Processed colors: %zu\r", count); fflush(stdout); }   // local.c:386-389
                continue;
            }
            if (std::max(best >= 0 ? bv : 0.0, max_unknown) < kDelta) { E.trace_hdr.stopped_early = 1; break; }   // nothing can reach DELTA
            // blocked: evaluate, in ONE round, the split of every leaf of the candidate tree that could still
            // matter -- undecided frontier nodes and, speculatively, the children of decided ones (a node's
            // split depends only on its members, never on the greedy order).  Leaves whose distortion (an
            // upper bound of their benefit) is far below the current best benefit are left for later; the
            // exactness test above catches them if they ever become relevant.
            const double ref_b = std::max(best >= 0 ? bv : 0.0, max_unknown);
            double thr = std::max(kDelta, spec_beta * ref_b);
            {
                // exact pruning: with R commits left, a leaf whose distortion is below the R-th largest
                // KNOWN frontier benefit can never be chosen (its own and all its descendants' benefits are
                // bounded by that distortion, and R better candidates outlast the remaining commits)
                const size_t R = K - count;
                std::vector<double> kb;
                for (size_t j = 0; j < count; j++) if (fkn[j]) kb.push_back(fval[j]);
                if (kb.size() >= R && R > 0) {
                    std::nth_element(kb.begin(), kb.begin() + (R - 1), kb.end(), std::greater<double>());
                    thr = std::max(thr, kb[R - 1] * (1.0 - 1e-9));
                }
            }
            lap(tm_greedy);
            round.clear();
            {
                std::vector<int> keep;
                for (int id : leaves) {
                    HNode &h = hn[id];
                    if (known(h)) continue;                     // n <= 1 / nosplit: never split
                    if (h.ub >= thr) round.push_back(id); else keep.push_back(id);
                }
                leaves.swap(keep);
            }
            if (round.empty()) {                                // every blocking node is a leaf with dist >= ref_b >= thr
                throw HipError("patolette_amd: split loop blocked without candidates");
            }
            lap(tm_select);
            // axes on the host (dsyev semantics), children ids, device records
            HIP_CHECK(hipStreamSynchronize(s));
            E.nodes.grow(hn.size() + 2 * round.size() + 2, hn.size());
            std::vector<int> todo;
            std::vector<NodeIn> recs;
            std::vector<int> ids;
            for (int id : round) {
                double ax[3];
                if (!node_axis(hn[id], ax)) { hn[id].nosplit = true; continue; }
                NodeIn d = make_nodedev(hn[id], bnd);
                for (int j = 0; j < 3; j++) d.axis[j] = ax[j];
                d.slot = (int)todo.size();
                d.child0 = (int)hn.size(); d.nchild = 2;
                hn[id].left = (int)hn.size(); hn[id].right = (int)hn.size() + 1;
                hn.push_back(HNode()); hn.push_back(HNode());
                todo.push_back(id); ids.push_back(id); recs.push_back(d);
            }
            for (size_t j = 0; j < count; j++) refresh(j);      // a failed eigen-solve above makes a node known (no split)
            lap(tm_axes);
            if (todo.empty()) continue;
            const int nr = (int)todo.size();
            size_t rpx = 0;
            for (int id : todo) rpx += hn[id].n;
            static const bool lq_trace = getenv("PAMD_LQ_TRACE") != nullptr;       // one line per split round on stderr
            if (lq_trace) fprintf(stderr, "patolette_amd: split round %zu: %d nodes, %zu pixels (%.2f of the image), %zu of %zu colours committed\n",
                                  E.stats.lq_rounds + 1, nr, rpx, (double)rpx / (double)(N ? N : 1), count, K);
            // one packet, one copy: node records, ids, children ids, then per GROUP of nodes the tile prefixes of both tilings.
            // One group = the whole round (the default).  PAMD_LQ_CHUNK_MB=m (experiment, profiles/r06_lq_chunk_major.txt): the round
            // is issued chunk-major -- groups of nodes of <= m MB, each taken through minmax -> hist -> cut -> count -> scan -> scatter
            // before the next group starts -- so that the second and third read of a node may find it in the Infinity Cache.
            static const size_t chunk_mb = getenv("PAMD_LQ_CHUNK_MB") ? (size_t)atoi(getenv("PAMD_LQ_CHUNK_MB")) : 0;
            std::vector<int> gb = {0};                              // group boundaries in todo[]
            if (chunk_mb > 0 && !sh) {
                const size_t cap_px = (chunk_mb << 20) / (planes * sizeof(double));
                size_t acc = 0;
                for (int r = 0; r < nr; r++) {
                    const size_t n = hn[todo[r]].n;
                    if (acc > 0 && acc + n > cap_px) { gb.push_back(r); acc = 0; }
                    acc += n;
                }
            }
            gb.push_back(nr);
            const int ng = (int)gb.size() - 1;
            std::vector<int> cids;
            for (int id : todo) { cids.push_back(hn[id].left); cids.push_back(hn[id].right); }
            std::vector<int> tAg, tPg;                               // the groups' prefixes one after the other (nrg + 1 entries each)
            std::vector<size_t> g_off(ng), g_px(ng);
            int ntA_max = 0, ntP_max = 0, nrg_max = 0;
            for (int g = 0; g < ng; g++) {
                g_off[g] = tAg.size();
                int a = 0, b = 0; size_t px = 0;
                tAg.push_back(0); tPg.push_back(0);
                for (int r = gb[g]; r < gb[g + 1]; r++) {
                    const unsigned long long n = hn[todo[r]].n;
                    recs[r].slot = r - gb[g];
                    a += (int)((n + kTileA - 1) / kTileA); b += (int)((n + kTileP - 1) / kTileP); px += n;
                    tAg.push_back(a); tPg.push_back(b);
                }
                g_px[g] = px;
                ntA_max = std::max(ntA_max, a); ntP_max = std::max(ntP_max, b); nrg_max = std::max(nrg_max, gb[g + 1] - gb[g]);
            }
            const size_t o_recs = 0, o_ids = o_recs + (size_t)nr * sizeof(NodeIn), o_cids = o_ids + (size_t)nr * sizeof(int),
                         o_tA = o_cids + cids.size() * sizeof(int), o_tP = o_tA + tAg.size() * sizeof(int),
                         pk_bytes = o_tP + tPg.size() * sizeof(int);
            E.h_packet.reserve(pk_bytes); E.packet.reserve(pk_bytes);
            std::memcpy(E.h_packet.p + o_recs, recs.data(), (size_t)nr * sizeof(NodeIn));
            std::memcpy(E.h_packet.p + o_ids, ids.data(), (size_t)nr * sizeof(int));
            std::memcpy(E.h_packet.p + o_cids, cids.data(), cids.size() * sizeof(int));
            std::memcpy(E.h_packet.p + o_tA, tAg.data(), tAg.size() * sizeof(int));
            std::memcpy(E.h_packet.p + o_tP, tPg.data(), tPg.size() * sizeof(int));
            lap(tm_packet);
            HIP_CHECK(hipMemcpyAsync(E.packet.p, E.h_packet.p, pk_bytes, hipMemcpyHostToDevice, s));
            const size_t lqs = (size_t)kNQ_LQ * 2 * kBuckets;
            E.hist.reserve(std::max(hs, lqs * nrg_max)); E.hsize.reserve((size_t)nrg_max * kBuckets); E.hcount.reserve((size_t)nrg_max * kBuckets);
            E.lut.reserve((size_t)nrg_max * kBuckets);
            E.tilesA.reserve(ntA_max); E.tilesP.reserve(ntP_max);
            E.tilecnt.reserve((size_t)ntP_max * kMaxChildren); E.tileoff.reserve((size_t)ntP_max * kMaxChildren);
            // the sweeps of a round alternate their direction through the pixels (the first one runs against the partition that
            // wrote them): each starts on what the previous one touched last.  PAMD_SWEEP_SNAKE=0: all forward
            const bool rev = snake && (E.stats.lq_rounds % 2 == 1);     // (the base clusters' moments were taken back to front)
            for (int g = 0; g < ng; g++) {
                const int r0 = gb[g], nrg = gb[g + 1] - r0;
                const int *d_ids = (const int *)(E.packet.p + o_ids) + r0;
                const int *d_tA0 = (const int *)(E.packet.p + o_tA) + g_off[g], *d_tP0 = (const int *)(E.packet.p + o_tP) + g_off[g];
                const int ntA = tAg[g_off[g] + nrg], ntP = tPg[g_off[g] + nrg];
                const size_t gpx = g_px[g];
                RoundSetup rs{(const NodeIn *)(E.packet.p + o_recs) + r0, d_ids, d_tA0, d_tP0, nrg, ntA, ntP,
                              E.tilesA.p, E.tilesP.p, E.hist.p, lqs * nrg, E.hsize.p, E.hcount.p, (size_t)nrg * kBuckets};
                {
                    const size_t work = std::max<size_t>(std::max<size_t>(std::max<size_t>(ntP, lqs * nrg / 4), (size_t)nrg * kNodeResetElems), 256);
                    hipLaunchKernelGGL(k_round_setup, (unsigned)std::min<size_t>((work + 255) / 256, 2048), 256, 0, s, E.nodes.p, rs);
                    HIP_CHECK(hipGetLastError());
                }
                launch_minmax(qlq, E.tilesA.p, ntA, gpx, E.nodes.p, s, rev);
                if (sh) shard_exchange_keys(E, d_ids, nrg);
                launch_hist(qlq, false, E.tilesA.p, ntA, gpx, E.nodes.p, E.hist.p, E.hsize.p, E.hcount.p, s, snake && !rev);
                if (sh) {
                    comm_sum_dev(E, E.hist.p, lqs * nrg, 0);
                    if (weighted) comm_sum_dev(E, E.hsize.p, (size_t)nrg * kBuckets, 1);
                    comm_sum_dev(E, E.hcount.p, (size_t)nrg * kBuckets, 2);
                }
                launch_cut(weighted, E.nodes.p, d_ids, nrg, E.hist.p, E.hsize.p, E.hcount.p, E.lut.p, s);
                launch_partition(qlq, E.tilesP.p, ntP, gpx, d_ids, d_tP0, nrg, E.nodes.p, E.lut.p, E.tilecnt.p, E.tileoff.p, true, s, inv_sums, rev);
                if (sh) {
                    hipLaunchKernelGGL(k_shard_children_local, (nrg + 63) / 64, 64, 0, s, E.nodes.p, d_ids, nrg);
                    HIP_CHECK(hipGetLastError());
                    shard_exchange_acc(E, (const int *)(E.packet.p + o_cids) + 2 * r0, 2 * nrg);
                }
            }
            lap(tm_enqueue);
            get_nodes_dev(E, (const int *)(E.packet.p + o_cids), (int)cids.size(), got);
            lap(tm_wait);
            for (size_t i = 0; i < cids.size(); i++) {
                HNode &c = hn[cids[i]];
                const NodeOut &d = got[i];
                c.begin = d.begin; c.n = d.n; c.gn = d.gn; c.buf = d.buf; c.sw = d.sw; c.psplit = d.psplit;
                for (int j = 0; j < 3; j++) c.mean[j] = d.mean[j];
                absorb_moments(c, d);
                leaf_bound(c);
            }
            for (int id : todo) { hn[id].split_done = true; E.stats.split_evals++; E.stats.split_px += hn[id].n; }
            for (int id : cids) leaves.push_back(id);
            for (size_t j = 0; j < count; j++) refresh(j);      // the round's nodes are known now
            E.stats.lq_rounds++;
            lap(tm_children);
        }
        if (lq_times) fprintf(stderr, "patolette_amd: split loop host ms: greedy %.3f select %.3f axes %.3f packet %.3f enqueue %.3f wait(gpu) %.3f children %.3f\n",
                              tm_greedy, tm_select, tm_axes, tm_packet, tm_enqueue, tm_wait, tm_children);
        result.resize(count);
    }
    len = count;
    centers.assign(3 * len, 0.0);
    for (size_t i = 0; i < len; i++) for (int j = 0; j < 3; j++) centers[(size_t)j * len + i] = hn[result[i]].mean[j];   // create.c:11-33
    if (max_members) { *max_members = 0; for (size_t i = 0; i < len; i++) *max_members = std::max(*max_members, hn[result[i]].gn); }
    E.stats.n_clusters = len;
    E.trace_hdr.n_clusters = (int32_t)len; E.trace_hdr.n_records = (int32_t)E.trace.size();
    E.cluster_centers = centers;
    E.stats.ms_lq = now_ms() - t0;
    return 0;
}

static int quantize_clusters(Engine &E, size_t N, size_t K, bool weighted, const Bounds &bnd,
                             std::vector<double> &centers, size_t &len, bool verbose = false, unsigned long long *max_members = nullptr) {
    int rc = quantize_clusters_run(E, N, K, weighted, bnd, centers, len, verbose, max_members, true);
    if (rc == -2) {                                             // the device-driven loop ran out of node records: once more, host-driven
        E.prep_N = 0;
        rc = quantize_clusters_run(E, N, K, weighted, bnd, centers, len, verbose, max_members, false);
    }
    return rc;
}

// --------------------------------------------------------------------------------------------
// KMeans refinement (refine.c:165-221); centres planar (k,3) f64 in/out
// --------------------------------------------------------------------------------------------
// largest_cluster: pixels of the most populous cluster the centres come from (0 = unknown): a dominant colour means one very long
// centroid chain, and the iterations then take the sorted path with the block-parallel chain from the start (kmeans_iterate)
// patolette_amd_set_kmeans_update: process-wide (the batch entry's helper threads must see the caller's choice).  -1: not set, the
// environment's default (PAMD_KMEANS_UPDATE)
std::atomic<int> g_km_update{-1};
// patolette_amd_set_subsample_cache: 1 (default) keep the subsample list on the device between calls on images of one size;
// 0 every call makes it again (on the helper thread): what a first call of a size costs, call after call
std::atomic<int> g_perm_cache{1};

// Call entry: if this image will be subsampled (assuming the quantisers deliver all K clusters; fewer only shorten the list),
// start making the list on a helper thread.  take = the longest list any cluster count <= K can ask for: min(ms, N).
static void subsample_start(Engine &E, size_t Nt, size_t K, size_t max_samples) {
    if (E.perm_job.valid()) {                                                   // a job left behind by a call that threw
        try { E.perm_job.get(); } catch (...) { E.hperm_N = 0; E.hperm_nx = 0; }   // ... and that may itself have failed: its list is void then
    }
    const size_t ms = std::max(max_samples, (size_t)(256 * 256));                // refine.c:21
    if (K == 0 || Nt < K) return;
    const size_t nxK = K * (size_t)(int)(ms / K);
    if (!(Nt > nxK)) return;                                                     // Clustering.cpp:311: no subsampling
    const size_t take = std::min(ms, Nt);
    if (!g_perm_cache.load(std::memory_order_relaxed)) { E.perm_N = 0; E.perm_nx = 0; E.hperm_N = 0; E.hperm_nx = 0; }
    if (E.perm_N == Nt && E.perm_nx >= nxK) return;                              // on the device already
    if (E.hperm_N == Nt && E.hperm_nx >= nxK) return;
    E.h_perm.reserve(take);
    E.hperm_N = 0; E.hperm_nx = 0;
    int32_t *dst = E.h_perm.p;
    hm::PermScratch *ws = &E.perm_ws;
    E.perm_job = std::async(std::launch::async, [dst, Nt, take, ws] { hm::rand_perm_prefix(Nt, take, 1234u, dst, *ws); });   // random.cpp:184-194
    E.hperm_N = Nt; E.hperm_nx = take;                                           // valid once perm_job has been waited for
}
// The KMeans stage: the first nx entries of rand_perm(Nt) on the device; no stream synchronisation (pinned staging)
static const int *subsample_list(Engine &E, size_t Nt, size_t nx, hipStream_t s) {
    if (E.perm_job.valid()) {
        try { E.perm_job.get(); } catch (...) { E.hperm_N = 0; E.hperm_nx = 0; throw; }
    }
    if (E.perm_N == Nt && E.perm_nx >= nx) return E.perm_dev.p;
    if (!(E.hperm_N == Nt && E.hperm_nx >= nx)) {                                // no helper was started (a stage-level call, fewer clusters than K)
        E.h_perm.reserve(nx);
        hm::rand_perm_prefix(Nt, nx, 1234u, E.h_perm.p, E.perm_ws);
        E.hperm_N = Nt; E.hperm_nx = nx;
    }
    E.perm_dev.reserve(E.hperm_nx);
    HIP_CHECK(hipMemcpyAsync(E.perm_dev.p, E.h_perm.p, E.hperm_nx * sizeof(int), hipMemcpyHostToDevice, s));
    E.perm_N = Nt; E.perm_nx = E.hperm_nx;
    return E.perm_dev.p;
}
static bool km_update_order_free() {
    static const bool env = getenv("PAMD_KMEANS_UPDATE") && atoi(getenv("PAMD_KMEANS_UPDATE")) != 0;
    const int v = g_km_update.load(std::memory_order_relaxed);
    return v < 0 ? env : v != 0;
}
// bound_x / bound_w: upper bounds of |colour value| and of the weights over the image (the order-free update scales by them)
static void kmeans_refine(Engine &E, size_t N, bool weighted, std::vector<double> &centers, size_t k, int niter, size_t max_samples,
                          bool nonfinite, unsigned long long largest_cluster, double bound_x, double bound_w) {
    hipStream_t s = E.stream;
    const KmSums sums_v{bound_x, bound_w};
    const KmSums *sums = km_update_order_free() ? &sums_v : nullptr;
    std::vector<float> cent(3 * k);
    for (size_t i = 0; i < k; i++) for (int j = 0; j < 3; j++) cent[3 * i + j] = (float)centers[(size_t)j * k + i];   // refine.c:102-125
    const size_t min_samples = 256 * 256;                                          // refine.c:21
    const size_t ms = std::max(max_samples, min_samples);
    const int mppc = (int)(ms / k);                                                // refine.c:87
    // Clustering.cpp:272-278 (fewer points than centroids) and :295-304 (a NaN / Inf among the f32 inputs) throw; refine.c:91
    // swallows the exception and the palette stays the initial centres
    const Shard *sh = E.shard;                                                     // sliced image: N is this GPU's part, the sample set is the whole image's
    const size_t Nt = sh ? sh->total : N;
    auto expect_longest = [&](size_t nx_) { return (size_t)((double)largest_cluster / (double)(Nt ? Nt : 1) * (double)nx_); };
    bool ok = Nt >= k && !nonfinite;
    size_t nx = Nt;
    E.stats.kmeans_samples = 0;
    if (ok) {
        const bool sub = Nt > k * (size_t)mppc;                                    // Clustering.cpp:311-319
        if (sub) nx = k * (size_t)mppc;
        E.km.reserve(nx, (int)k);
        const int *dperm = nullptr;
        if (sub) dperm = subsample_list(E, Nt, nx, s);                            // faiss draws the subsample from rand_perm(N, seed 1234)
        if (sh) {
            // every GPU contributes the samples that fall into its slice (zero bits elsewhere); the integer SUM of the bit
            // patterns hands every GPU the whole sample set, and each runs the same deterministic iterations on it
            kmeans_gather_slice(E.cvt.p, N, weighted, dperm, nx, sh->begin, E.km, s);
            comm_sum_dev(E, E.km.sx.p, nx, 2); comm_sum_dev(E, E.km.sy.p, nx, 2); comm_sum_dev(E, E.km.sz.p, nx, 2);
            if (weighted) comm_sum_dev(E, E.km.sw.p, nx, 2);
            if (nx == k) {                                                         // Clustering.cpp:331-352: centroids = first k input vectors
                std::vector<float> f(3 * k);
                HIP_CHECK(hipMemcpy(f.data(), E.km.sx.p, k * sizeof(float), hipMemcpyDeviceToHost));
                HIP_CHECK(hipMemcpy(f.data() + k, E.km.sy.p, k * sizeof(float), hipMemcpyDeviceToHost));
                HIP_CHECK(hipMemcpy(f.data() + 2 * k, E.km.sz.p, k * sizeof(float), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < k; i++) for (int j = 0; j < 3; j++) cent[3 * i + j] = f[(size_t)j * k + i];
            } else {
                HIP_CHECK(hipMemcpyAsync(E.km.cent.p, cent.data(), 3 * k * sizeof(float), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipStreamSynchronize(s));
                kmeans_iterate(E.km, nx, (int)k, weighted, niter, s, expect_longest(nx), sums);
                HIP_CHECK(hipMemcpyAsync(cent.data(), E.km.cent.p, 3 * k * sizeof(float), hipMemcpyDeviceToHost, s));
                E.sync();
                E.stats.kmeans_samples = nx;
            }
        } else if (nx == k) {                                                      // Clustering.cpp:331-352: centroids = first k input vectors
            std::vector<double> first(3 * k);
            for (int j = 0; j < 3; j++) HIP_CHECK(hipMemcpy(first.data() + (size_t)j * k, E.cvt.p + (size_t)j * N, k * sizeof(double), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < k; i++) for (int j = 0; j < 3; j++) cent[3 * i + j] = (float)first[(size_t)j * k + i];
        } else {
            kmeans_gather(E.cvt.p, N, weighted, dperm, nx, E.km, s);
            E.h_cent.reserve(3 * k);                                               // pinned: no synchronisation before the iterations
            std::memcpy(E.h_cent.p, cent.data(), 3 * k * sizeof(float));
            HIP_CHECK(hipMemcpyAsync(E.km.cent.p, E.h_cent.p, 3 * k * sizeof(float), hipMemcpyHostToDevice, s));
            kmeans_iterate(E.km, nx, (int)k, weighted, niter, s, expect_longest(nx), sums);
            HIP_CHECK(hipMemcpyAsync(E.h_cent.p, E.km.cent.p, 3 * k * sizeof(float), hipMemcpyDeviceToHost, s));
            E.sync();
            std::memcpy(cent.data(), E.h_cent.p, 3 * k * sizeof(float));
            E.stats.kmeans_samples = nx;
        }
    }
    for (size_t i = 0; i < k; i++) for (int j = 0; j < 3; j++) centers[(size_t)j * k + i] = (double)cent[3 * i + j];   // refine.c:202-212
}

static void palette_rows(std::vector<double> &pal, size_t len, void (*f)(double[3])) {
    for (size_t i = 0; i < len; i++) {
        double c[3] = {pal[i], pal[len + i], pal[2 * len + i]};
        f(c);
        pal[i] = c[0]; pal[len + i] = c[1]; pal[2 * len + i] = c[2];
    }
}

// between_copy_and_wait: work to enqueue behind the statistics' download while the host is still waiting for it
static Bounds read_bounds(Engine &E, bool weighted, const std::function<void()> &between_copy_and_wait = nullptr) {
    E.h_cstats.reserve(1);
    HIP_CHECK(hipMemcpyAsync(E.h_cstats.p, E.cstats.p, sizeof(ConvertStats), hipMemcpyDeviceToHost, E.stream));
    if (between_copy_and_wait) {                               // wait for the download alone, not for what is enqueued behind it
        if (!E.ev_stats) HIP_CHECK(hipEventCreateWithFlags(&E.ev_stats, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(E.ev_stats, E.stream));
        between_copy_and_wait();
        HIP_CHECK(hipEventSynchronize(E.ev_stats));
    } else E.sync();
    const ConvertStats &cs = *E.h_cstats.p;
    Bounds b;
    b.cmax = 0; b.range = 0;
    unsigned long long kmin[3], kmax[3], kw = 0ULL, nonfinite = cs.nonfinite_f32 != 0u ? 1ULL : 0ULL;
    double parts[3][2];
    for (int p = 0; p < 3; p++) {
        kmin[p] = ~0ULL; kmax[p] = 0ULL;
        for (int t = 0; t < kStatSlots; t++) { kmin[p] = std::min(kmin[p], cs.minkey[t][p]); kmax[p] = std::max(kmax[p], cs.maxkey[t][p]); }
        parts[p][0] = 0; parts[p][1] = 0;                      // slot partials are exact multiples of the bin grids
        for (int t = 0; t < kStatSlots; t++) { parts[p][0] += cs.sum[t][p][0]; parts[p][1] += cs.sum[t][p][1]; }
    }
    for (int t = 0; t < kStatSlots; t++) kw = std::max(kw, cs.wmaxkey[t]);
    if (E.shard) {                                             // one image over several GPUs: the bounds and sums of ALL its pixels
        const unsigned long long mine[8] = {kmin[0], kmin[1], kmin[2], kmax[0], kmax[1], kmax[2], kw, nonfinite};
        const std::vector<unsigned long long> all = comm_gather_u64(E, mine, 8);
        for (int r = 0; r < E.shard->comm.size; r++) {
            const unsigned long long *row = all.data() + (size_t)r * 8;
            for (int p = 0; p < 3; p++) { kmin[p] = std::min(kmin[p], row[p]); kmax[p] = std::max(kmax[p], row[3 + p]); }
            kw = std::max(kw, row[6]); nonfinite |= row[7];
        }
        comm_sum_host(E, &parts[0][0], 6, 0);
    }
    for (int p = 0; p < 3; p++) {
        const unsigned long long ka = kmin[p], kb = kmax[p];
        double mn = key_f64(ka), mx = key_f64(kb);
        if (!(mn <= mx)) { mn = 0; mx = 0; }
        b.lo[p] = mn; b.hi[p] = mx;
        b.cmax = std::max(b.cmax, std::max(std::fabs(mn), std::fabs(mx)));
        b.range = std::max(b.range, mx - mn);
    }
    b.wmax = weighted ? std::max(1.0, key_f64(kw)) : 1.0;
    const double lin = b.wmax * std::max(b.cmax, 1.0);
    const double quad = 3.0 * b.wmax * std::max(std::max(b.range, b.cmax), 1e-300) * std::max(std::max(b.range, b.cmax), 1e-300);
    b.e_lin = exp_bound(lin);
    b.e_quad = exp_bound(std::max(quad, 1e-300));
    for (int p = 0; p < 3; p++) { b.sum[p] = parts[p][0] + parts[p][1]; b.sum2[p][0] = parts[p][0]; b.sum2[p][1] = parts[p][1]; }
    for (int q = 0; q < 6; q++)
        for (int t = 0; t < kStatSlots; t++) { b.mom2[q][0] += cs.mom[t][q][0]; b.mom2[q][1] += cs.mom[t][q][1]; }   // exact: parts on the grids
    b.nonfinite = nonfinite != 0ULL;
    return b;
}

// --------------------------------------------------------------------------------------------
// the full path, inputs resident in HBM
// --------------------------------------------------------------------------------------------
struct Pixels {                      // device-resident input image: planar f64 sRGB, or interleaved 8-bit sRGB
    const double *f64 = nullptr;
    const unsigned char *u8 = nullptr;
    int channels = 3;
    bool rows = false;               // f64 as (N,3) row-major instead of planar
    bool converted = false;          // E.cvt already holds the converted image and E.cstats its statistics (run_host: chunk by chunk behind the upload)
};

// What stage S1 launches: the conversion and the sums that ride with it (run_device; run_host when it converts behind the upload)
struct ConvertPlan { int which; BinK sumk, momk; };
static ConvertPlan convert_plan(const Engine &E, const patolette__QuantizationOptions *opt, size_t N) {
    ConvertPlan p{PAMD_COPY, BinK{0.0, 0.0}, BinK{0.0, 0.0}};
    if (opt->color_space == patolette__CIELuv) p.which = PAMD_SRGB_TO_CIELUV;
    else if (opt->color_space == patolette__ICtCp) p.which = PAMD_SRGB_TO_ICTCP;
    // the root mean of the global quantiser (matrix2D.c:229) rides along with the conversion where the colour space bounds
    // the values a priori: |I| <= 1, |Ct|, |Cp| <= 0.5 (2^1); L <= 100, |u|, |v| < 256 (2^8).  sRGB passes user data through.
    const size_t Nt = E.shard ? E.shard->total : N;            // a slice sums on the grids of the whole image
    int rootP = 1; while ((1ULL << rootP) < (Nt > 1 ? Nt : 2)) rootP++;
    if (p.which == PAMD_SRGB_TO_ICTCP) p.sumk = make_bink(1, rootP);
    else if (p.which == PAMD_SRGB_TO_CIELUV) p.sumk = make_bink(8, rootP);
    // one GPU: the raw second moments ride along too (products < 1, resp. < 2^16), and the root's covariance needs no sweep
    static const bool mom_on = !(getenv("PAMD_ROOT_MOMENTS") && atoi(getenv("PAMD_ROOT_MOMENTS")) == 0);
    if (!E.shard && mom_on) {
        if (p.which == PAMD_SRGB_TO_ICTCP) p.momk = make_bink(1, rootP);
        else if (p.which == PAMD_SRGB_TO_CIELUV) p.momk = make_bink(16, rootP);
    }
    return p;
}

static void run_device(Engine &E, size_t width, size_t height, Pixels px, const double *d_weights, size_t K,
                       const patolette__QuantizationOptions *opt, double *palette, void *d_map, int map_elem) {
    // d_map == nullptr with palette_only unset: the palette takes the same conversions, the map kernels are skipped
    hipStream_t s = E.stream;
    const size_t N = width * height;
    const bool weighted = d_weights != nullptr;
    const double t_start = now_ms();
    E.stats = patolette_amd__Stats{};
    E.prep_N = 0;
    if (opt->kmeans_niter > 0) subsample_start(E, E.shard ? E.shard->total : N, K, opt->kmeans_max_samples);
    // S1: colour conversion into the working image (x|y|z|w planar), patolette.c:201-207
    double t0 = now_ms();
    E.cvt.reserve((weighted ? 4 : 3) * N);
    E.cstats.reserve(1);
    if (E.shard && opt->dither && !opt->palette_only)
        throw HipError("patolette_amd: dithering walks the whole image along one curve; it is not available per slice");
    const ConvertPlan cp = convert_plan(E, opt, N);
    const int which = cp.which;
    const BinK sumk = cp.sumk, momk = cp.momk;
    if (px.converted) { /* run_host has converted the image chunk by chunk behind its upload */ }
    else if (px.u8) launch_convert_u8(which, px.u8, px.channels, E.cvt.p, N, E.cstats.p, s, sumk, momk);
    else if (px.rows) launch_convert_rows(which, px.f64, E.cvt.p, N, E.cstats.p, s, sumk, momk);
    else launch_convert(which, px.f64, E.cvt.p, N, E.cstats.p, s, sumk, momk);
    if (weighted) {
        HIP_CHECK(hipMemcpyAsync(E.cvt.p + 3 * N, d_weights, N * sizeof(double), hipMemcpyDeviceToDevice, s));
        launch_weight_stats(d_weights, N, E.cstats.p, s);
    }
    // the quantiser's size-only preparations go behind the statistics' download: they run while the host derives the bounds
    Bounds bnd = read_bounds(E, weighted, [&] { gq_prepare(E, N, weighted); });
    bnd.have_sum = sumk.M0 != 0.0;
    bnd.have_mom = momk.M0 != 0.0;
    E.stats.ms_convert = now_ms() - t0;

    // S2 + S3: global + local quantiser (progress lines as patolette.c:209-229 prints them when verbose)
    if (opt->verbose) printf("patolette ======== Palette generation \n");
    std::vector<double> pal;
    size_t len = 0;
    unsigned long long largest_cluster = 0;
    if (quantize_clusters(E, N, K, weighted, bnd, pal, len, opt->verbose, &largest_cluster) != 0) throw HipError("internal quantization error");

    // S4: optional KMeans refinement
    t0 = now_ms();
    if (opt->kmeans_niter > 0) {
        if (opt->verbose) printf("patolette ======== KMeans refinement\n");                // patolette.c:249-251
        kmeans_refine(E, N, weighted, pal, len, opt->kmeans_niter, opt->kmeans_max_samples, bnd.nonfinite, largest_cluster, bnd.cmax, bnd.wmax);
    }
    E.stats.ms_kmeans = now_ms() - t0;

    // S5: palette map
    t0 = now_ms();
    if (!opt->palette_only) {
        E.dpal.reserve(3 * len);
        if (opt->verbose) printf(opt->dither ? "patolette ======== Dithering\n" : "patolette ======== NN mapping\n");
        if (opt->dither) {                                                     // patolette.c:268-299
            int pix = PAMD_SRGB_TO_REC2020;
            void (*pf)(double[3]) = hm::color::srgb_to_rec2020;
            if (opt->color_space == patolette__CIELuv) { pix = PAMD_CIELUV_TO_REC2020; pf = hm::color::cieluv_to_rec2020; }
            else if (opt->color_space == patolette__ICtCp) { pix = PAMD_ICTCP_TO_REC2020; pf = hm::color::ictcp_to_rec2020; }
            palette_rows(pal, len, pf);
            E.map_palette = pal;
            if (d_map) {
                HIP_CHECK(hipMemcpyAsync(E.dpal.p, pal.data(), 3 * len * sizeof(double), hipMemcpyHostToDevice, s));
                HIP_CHECK(hipStreamSynchronize(s));
                if (dither_lane_layout(width, height, (int)len)) {               // decided ONCE: launch_dither is told
                    // the pixels go into curve order anyway: their conversion to linear Rec2020 rides on that pass
                    launch_dither(E.cvt.p, N, pix, width, height, E.dpal.p, pal.data(), (int)len, d_map, map_elem, E.nn, s, 1);
                } else {
                    E.aux.reserve(3 * N);
                    launch_convert(pix, E.cvt.p, E.aux.p, N, nullptr, s);      // plane stride of cvt is N for x,y,z
                    launch_dither(E.aux.p, N, PAMD_COPY, width, height, E.dpal.p, pal.data(), (int)len, d_map, map_elem, E.nn, s, 0);
                }
                E.stats.dither_segments = E.nn.dither_segments; E.stats.dither_repairs = E.nn.dither_repairs; E.stats.dither_rounds = E.nn.dither_rounds; E.stats.dither_through = E.nn.dither_through;
    E.stats.dither_jumps = E.nn.dither_jumps; E.stats.dither_solo = E.nn.dither_solo;
            }
            palette_rows(pal, len, hm::color::rec2020_to_srgb);
        } else {                                                               // patolette.c:300-324
            const double *pixels = E.cvt.p;
            const double *blo = bnd.lo, *bhi = bnd.hi;                         // exact min/max of the pixels being mapped
            Bounds b2;
            if (opt->color_space == patolette__CIELuv) {
                if (d_map) {
                    E.aux.reserve(3 * N);
                    launch_convert(PAMD_CIELUV_TO_ICTCP, E.cvt.p, E.aux.p, N, E.cstats.p, s);
                    b2 = read_bounds(E, false);
                    blo = b2.lo; bhi = b2.hi;
                    pixels = E.aux.p;
                }
                palette_rows(pal, len, hm::color::cieluv_to_rec2020);
                palette_rows(pal, len, hm::color::rec2020_to_srgb);
                palette_rows(pal, len, hm::color::srgb_to_ictcp);
            }
            E.map_palette = pal;
            if (d_map) {
                // from a pinned copy: no wait for the transfer (a pageable source is staged and waited for, ~15 us with nothing queued
                // behind it); the copy is this engine's and is next written by its next call, after this one's final synchronisation
                E.h_pal.reserve(3 * len);
                std::memcpy(E.h_pal.p, pal.data(), 3 * len * sizeof(double));
                HIP_CHECK(hipMemcpyAsync(E.dpal.p, E.h_pal.p, 3 * len * sizeof(double), hipMemcpyHostToDevice, s));
                launch_nn_map(pixels, N, N, E.dpal.p, (int)len, d_map, map_elem, blo, bhi, E.nn, s);
            }
            palette_rows(pal, len, hm::color::ictcp_to_rec2020);
            palette_rows(pal, len, hm::color::rec2020_to_srgb);
        }
        E.sync();
    }
    E.stats.ms_map = now_ms() - t0;
    // S6: palette write-out, unused rows = -1 (patolette.c:327-336)
    for (size_t j = 0; j < K * 3; j++) palette[j] = -1.0;
    for (int j = 0; j < 3; j++) for (size_t i = 0; i < len; i++) palette[K * (size_t)j + i] = pal[(size_t)j * len + i];
    E.stats.ms_total = now_ms() - t_start;
}

struct CodeError : std::runtime_error {        // failure with a dedicated exit code (saliency stage)
    int code;
    CodeError(int c, const char *m) : std::runtime_error(m), code(c) {}
};

// weights the Python binding derives when tile_size > 0 (patolette.pyx:407-414), left on the device
static const double *derive_weights(Engine &E, const double *d_f64, const unsigned char *d_u8, int channels, size_t width,
                                    size_t height, double tile_size) {
    E.wsal.reserve(width * height);
    const double t0 = now_ms();
    const int rc = saliency_weights(E.sal, d_f64, d_u8, channels, width, height, tile_size, E.wsal.p, E.stream);
    if (ktimer().enabled) ktimer().collect();
    E.ms_saliency = now_ms() - t0;
    if (rc == kSalBadShape) throw CodeError(-5, "saliency weights: image shape not supported");
    if (rc == kSalSingular) throw CodeError(-6, "saliency weights: singular border covariance");
    return E.wsal.p;
}

static int validate(size_t width, size_t height, size_t K) {               // patolette.c:61-95
    const size_t px = width * height;
    if (px == 0) return -2;
    if (K < 1) return -3;
    if (px > (size_t)40000 * 40000) return -4;
    return 0;
}

static int map_elem_for(size_t K) { return K <= 256 ? 1 : 4; }

// The C ABI hands the index map back as size_t (patolette.h: `size_t *palette_map`): 8 bytes per pixel on the host for 1 (or
// 4) on the device.  The narrow map crosses PCIe in chunks through pinned staging while a few host threads widen the
// previous chunk into the caller's buffer.
static void download_map_widened(Engine &E, const void *d_map, int me, size_t N, size_t *out) {
    if (N == 0) return;
    const size_t chunk = (size_t)2 << 20;                                   // elements per chunk
    const size_t nchunks = ceil_div(N, chunk);
    E.h_mapstage.reserve(2 * chunk * (size_t)me);
    const int T = (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)std::thread::hardware_concurrency(), ceil_div(N, (size_t)1 << 18)}));
    std::atomic<size_t> ready{0};
    std::vector<std::atomic<int>> done(nchunks);
    for (auto &d : done) d.store(0);
    std::atomic<bool> failed{false};
    auto worker = [&](int tid) {
        for (size_t c = 0; c < nchunks; c++) {
            while (ready.load(std::memory_order_acquire) <= c) {
                if (failed.load(std::memory_order_relaxed)) return;
                std::this_thread::yield();
            }
            const size_t lo = c * chunk, cnt = std::min(chunk, N - lo);
            const size_t a = cnt * (size_t)tid / (size_t)T, b = cnt * (size_t)(tid + 1) / (size_t)T;
            const unsigned char *src = E.h_mapstage.p + (c & 1) * chunk * (size_t)me;
            if (me == 1) for (size_t i = a; i < b; i++) out[lo + i] = (size_t)src[i];
            else { const unsigned int *s4 = (const unsigned int *)src; for (size_t i = a; i < b; i++) out[lo + i] = (size_t)s4[i]; }
            done[c].fetch_add(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; t++) pool.emplace_back(worker, t);
    auto issue = [&](size_t c) {
        const size_t lo = c * chunk, cnt = std::min(chunk, N - lo);
        HIP_CHECK(hipMemcpyAsync(E.h_mapstage.p + (c & 1) * chunk * (size_t)me, (const unsigned char *)d_map + lo * (size_t)me, cnt * (size_t)me,
                                 hipMemcpyDeviceToHost, E.stream));
    };
    try {
        issue(0);
        for (size_t c = 0; c < nchunks; c++) {
            HIP_CHECK(hipStreamSynchronize(E.stream));
            ready.store(c + 1, std::memory_order_release);
            if (c + 1 < nchunks) {
                // chunk c+1 reuses the half chunk c-1 was staged in: every thread must be done with c-1 (the calling thread
                // is worker 0 and widens its share of chunk c only after issuing the copy)
                if (c >= 1) while (done[c - 1].load(std::memory_order_acquire) < T) std::this_thread::yield();
                issue(c + 1);
            }
            {   // worker 0's share of chunk c
                const size_t lo = c * chunk, cnt = std::min(chunk, N - lo);
                const size_t b = cnt / (size_t)T;
                const unsigned char *src = E.h_mapstage.p + (c & 1) * chunk * (size_t)me;
                if (me == 1) for (size_t i = 0; i < b; i++) out[lo + i] = (size_t)src[i];
                else { const unsigned int *s4 = (const unsigned int *)src; for (size_t i = 0; i < b; i++) out[lo + i] = (size_t)s4[i]; }
                done[c].fetch_add(1, std::memory_order_release);
            }
        }
    } catch (...) {
        failed.store(true);
        for (auto &th : pool) th.join();
        throw;
    }
    for (auto &th : pool) th.join();
}

// host-buffer entry: upload, run, download + widen
static void run_host(Engine &E, size_t width, size_t height, const double *data, const double *weights, double tile_size,
                     size_t K, const patolette__QuantizationOptions *opt, double *palette, size_t *palette_map, bool rows = false,
                     void *d_map_out = nullptr) {
    // d_map_out: leave the narrow index map (u8 for K <= 256, else u32) in HBM there instead of widening it into palette_map
    const size_t N = width * height;
    double t0 = now_ms();
    E.src.reserve(3 * N);
    // Large images go up in pieces and what a piece completes is converted (on a second stream) while the next one is on the link:
    // the conversion (0.5 ms of a 4096^2 image's 7 ms upload) disappears behind the copy.  Its statistics are exact sums and keyed
    // extrema, so the chunking changes no bit.  With derived weights too: the saliency stage reads the sRGB source, which stays where it is.
    const size_t chunk_min = getenv("PAMD_UPLOAD_CHUNK_MIN") ? (size_t)atoll(getenv("PAMD_UPLOAD_CHUNK_MIN")) : ((size_t)1 << 21);   // (read per call: tests lower it)
    const bool derive = !weights && tile_size > 0.0;
    const bool chunked = N >= chunk_min && !E.shard;
    bool converted = false;
    if (chunked) {
        const bool overlap = true;
        // planar source: the first two planes go up whole (every copy call has a fixed cost: 24 pieces made the upload 0.44 ms
        // longer than one), the third in four pieces with the conversion of the pixels it completes behind each; row-major
        // source: four pieces of whole pixels
        const size_t nch = 4, per = ((N + nch - 1) / nch + 255) & ~(size_t)255;
        ConvertPlan cp{};
        if (overlap) {
            if (!E.stream2) {
                HIP_CHECK(hipStreamCreateWithFlags(&E.stream2, hipStreamNonBlocking));
                for (hipEvent_t *e : {&E.ev_up[0], &E.ev_up[1], &E.ev_join}) HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
            }
            cp = convert_plan(E, opt, N);
            E.cvt.reserve(((weights || derive) ? 4 : 3) * N);        // (the derived weights' plane: run_device must not move what is converted here)
            E.cstats.reserve(1);
            HIP_CHECK(hipEventRecord(E.ev_join, E.stream));          // whatever the engine's stream still holds comes first
            HIP_CHECK(hipStreamWaitEvent(E.stream2, E.ev_join, 0));
        }
        if (!rows) HIP_CHECK(hipMemcpyAsync(E.src.p, data, 2 * N * sizeof(double), hipMemcpyHostToDevice, E.stream));
        size_t c = 0;
        for (size_t lo = 0; lo < N; lo += per, c++) {
            const size_t cnt = std::min(per, N - lo);
            if (rows) HIP_CHECK(hipMemcpyAsync(E.src.p + 3 * lo, data + 3 * lo, 3 * cnt * sizeof(double), hipMemcpyHostToDevice, E.stream));
            else HIP_CHECK(hipMemcpyAsync(E.src.p + 2 * N + lo, data + 2 * N + lo, cnt * sizeof(double), hipMemcpyHostToDevice, E.stream));
            if (overlap) {
                HIP_CHECK(hipEventRecord(E.ev_up[c & 1], E.stream));
                HIP_CHECK(hipStreamWaitEvent(E.stream2, E.ev_up[c & 1], 0));
                if (rows) launch_convert_rows(cp.which, E.src.p, E.cvt.p, N, E.cstats.p, E.stream2, cp.sumk, cp.momk, lo, lo + cnt, c == 0);
                else launch_convert(cp.which, E.src.p, E.cvt.p, N, E.cstats.p, E.stream2, cp.sumk, cp.momk, lo, lo + cnt, c == 0);
            }
        }
        if (overlap) {
            HIP_CHECK(hipEventRecord(E.ev_join, E.stream2));
            HIP_CHECK(hipStreamWaitEvent(E.stream, E.ev_join, 0));
            converted = true;
        }
    } else {
        HIP_CHECK(hipMemcpyAsync(E.src.p, data, 3 * N * sizeof(double), hipMemcpyHostToDevice, E.stream));
    }
    if (weights) {
        E.wsrc.reserve(N);
        HIP_CHECK(hipMemcpyAsync(E.wsrc.p, weights, N * sizeof(double), hipMemcpyHostToDevice, E.stream));
    }
    HIP_CHECK(hipStreamSynchronize(E.stream));
    const double up = now_ms() - t0;
    const double *d_w = weights ? E.wsrc.p : nullptr;
    E.ms_saliency = 0.0;
    if (derive) d_w = derive_weights(E, E.src.p, nullptr, rows ? -3 : 3, width, height, tile_size);
    const int me = map_elem_for(K);
    if (!opt->palette_only && !d_map_out) E.dmap.reserve(N * (size_t)me);
    std::vector<double> pal(3 * K);
    Pixels px{E.src.p, nullptr, 3, rows};
    px.converted = converted;
    run_device(E, width, height, px, d_w, K, opt, pal.data(), d_map_out ? d_map_out : (void *)E.dmap.p, me);
    E.stats.ms_saliency = E.ms_saliency;
    E.stats.ms_total += E.ms_saliency;
    t0 = now_ms();
    if (!opt->palette_only && !d_map_out) {
        const bool touched = !(opt->dither && std::max(width, height) <= 1);   // 1x1 dither visits nothing (riemersma.c:452-456)
        if (touched) {
            download_map_widened(E, E.dmap.p, me, N, palette_map);
        }
    }
    std::memcpy(palette, pal.data(), 3 * K * sizeof(double));
    E.stats.ms_upload = up;
    E.stats.ms_download = now_ms() - t0;
    E.stats.ms_total += up + E.stats.ms_download;
}

// palette_u8 = clip(palette * 255, 0, 255) truncated (README.md:178-181); unused rows (-1) become 0
static void palette_to_u8(const double *palette, size_t K, unsigned char *out) {
    for (size_t i = 0; i < K; i++)
        for (int c = 0; c < 3; c++) {
            double v = palette[K * (size_t)c + i] * 255.0;
            v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
            out[3 * i + c] = (unsigned char)v;
        }
}

// 8-bit adaptor around the path: interleaved u8 in; f64 palette, u8 palette, index map and
// reconstructed u8 image out.  `pixels`, `d_map_out`, `d_quant_out` are device pointers when `on_device`.
static void run_u8(Engine &E, size_t width, size_t height, const unsigned char *pixels, int channels, const double *weights,
                   double tile_size, size_t K, const patolette__QuantizationOptions *opt, double *palette, unsigned char *palette_u8,
                   void *map_out, int map_elem_out, unsigned char *quant_out, bool on_device, bool map_on_device = false) {
    const size_t N = width * height;
    hipStream_t s = E.stream;
    double t0 = now_ms();
    const unsigned char *d_px = pixels;
    const double *d_w = weights;
    bool converted = false;
    if (!on_device) {
        E.src8.reserve(N * (size_t)channels);
        d_px = E.src8.p;
        // as run_host: a large image goes up in four pieces of whole pixels, each converted (second stream) while the next is on the link
        const size_t chunk_min = getenv("PAMD_UPLOAD_CHUNK_MIN") ? (size_t)atoll(getenv("PAMD_UPLOAD_CHUNK_MIN")) : ((size_t)1 << 21);
        const bool derive = !weights && tile_size > 0.0;             // the saliency stage reads the 8-bit image itself (it stays where it is)
        if (N >= chunk_min && !E.shard) {
            if (!E.stream2) {
                HIP_CHECK(hipStreamCreateWithFlags(&E.stream2, hipStreamNonBlocking));
                for (hipEvent_t *e : {&E.ev_up[0], &E.ev_up[1], &E.ev_join}) HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
            }
            const ConvertPlan cp = convert_plan(E, opt, N);
            E.cvt.reserve(((weights || derive) ? 4 : 3) * N);        // (the derived weights' plane: run_device must not move what is converted here)
            E.cstats.reserve(1);
            HIP_CHECK(hipEventRecord(E.ev_join, E.stream));
            HIP_CHECK(hipStreamWaitEvent(E.stream2, E.ev_join, 0));
            const size_t nch = 4, per = ((N + nch - 1) / nch + 255) & ~(size_t)255;
            size_t c = 0;
            for (size_t lo = 0; lo < N; lo += per, c++) {
                const size_t cnt = std::min(per, N - lo);
                HIP_CHECK(hipMemcpyAsync(E.src8.p + lo * (size_t)channels, pixels + lo * (size_t)channels, cnt * (size_t)channels, hipMemcpyHostToDevice, s));
                HIP_CHECK(hipEventRecord(E.ev_up[c & 1], s));
                HIP_CHECK(hipStreamWaitEvent(E.stream2, E.ev_up[c & 1], 0));
                launch_convert_u8(cp.which, E.src8.p, channels, E.cvt.p, N, E.cstats.p, E.stream2, cp.sumk, cp.momk, lo, lo + cnt, c == 0);
            }
            HIP_CHECK(hipEventRecord(E.ev_join, E.stream2));
            HIP_CHECK(hipStreamWaitEvent(s, E.ev_join, 0));
            converted = true;
        } else {
            HIP_CHECK(hipMemcpyAsync(E.src8.p, pixels, N * (size_t)channels, hipMemcpyHostToDevice, s));
        }
        if (weights) {
            E.wsrc.reserve(N);
            HIP_CHECK(hipMemcpyAsync(E.wsrc.p, weights, N * sizeof(double), hipMemcpyHostToDevice, s));
            d_w = E.wsrc.p;
        }
        HIP_CHECK(hipStreamSynchronize(s));
    }
    const double up = now_ms() - t0;
    E.ms_saliency = 0.0;
    if (!weights && tile_size > 0.0) d_w = derive_weights(E, nullptr, d_px, channels, width, height, tile_size);
    const int me = map_elem_for(K);
    const bool want_map = !opt->palette_only && (map_out || quant_out);
    void *d_map = nullptr;                  // stays null when no map-derived output is wanted: the map kernels are skipped
    if (want_map) {
        if ((on_device || map_on_device) && map_out && map_elem_out == me) d_map = map_out;
        else { E.dmap.reserve(N * (size_t)me); d_map = E.dmap.p; }
    }
    std::vector<double> pal(3 * K);
    Pixels px8{nullptr, d_px, channels};
    px8.converted = converted;
    run_device(E, width, height, px8, d_w, K, opt, pal.data(), d_map, me);
    E.stats.ms_saliency = E.ms_saliency;
    E.stats.ms_total += E.ms_saliency;
    t0 = now_ms();
    std::vector<unsigned char> p8(3 * K);
    palette_to_u8(pal.data(), K, p8.data());
    if (palette) std::memcpy(palette, pal.data(), 3 * K * sizeof(double));
    if (palette_u8) std::memcpy(palette_u8, p8.data(), 3 * K);
    const bool touched = !(opt->dither && std::max(width, height) <= 1);       // 1x1 dither visits nothing
    if (want_map && touched) {
        if (quant_out) {
            E.pal8.reserve(3 * K);
            HIP_CHECK(hipMemcpyAsync(E.pal8.p, p8.data(), 3 * K, hipMemcpyHostToDevice, s));
            unsigned char *d_q = quant_out;
            if (!on_device) { E.quant8.reserve(3 * N); d_q = E.quant8.p; }
            launch_reconstruct(d_map, me, N, E.pal8.p, (int)K, d_q, s);
            if (!on_device) HIP_CHECK(hipMemcpyAsync(quant_out, d_q, 3 * N, hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
        }
        if (map_out && d_map != map_out) {
            if (on_device || map_on_device) throw HipError("patolette_amd: device map_elem_bytes must be 1 for K <= 256, else 4");
            if (map_elem_out == me) HIP_CHECK(hipMemcpy(map_out, d_map, N * (size_t)me, hipMemcpyDeviceToHost));
            else {
                std::vector<unsigned char> tmp(N * (size_t)me);
                HIP_CHECK(hipMemcpy(tmp.data(), d_map, tmp.size(), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < N; i++) {
                    const size_t v = me == 1 ? (size_t)tmp[i] : (size_t)reinterpret_cast<unsigned int *>(tmp.data())[i];
                    switch (map_elem_out) {
                        case 1: ((unsigned char *)map_out)[i] = (unsigned char)v; break;
                        case 2: ((unsigned short *)map_out)[i] = (unsigned short)v; break;
                        case 4: ((unsigned int *)map_out)[i] = (unsigned int)v; break;
                        default: ((size_t *)map_out)[i] = v; break;
                    }
                }
            }
        }
        E.sync();
    }
    E.stats.ms_upload = up;
    E.stats.ms_download = now_ms() - t0;
    E.stats.ms_total += up + E.stats.ms_download;
}

static int validate_u8(size_t K, int channels, const void *map_out, int map_elem) {
    if (channels != 3 && channels != 4) return -1;
    if (map_out) {
        if (map_elem != 1 && map_elem != 2 && map_elem != 4 && map_elem != 8) return -1;
        if (map_elem == 1 && K > 256) return -1;
        if (map_elem == 2 && K > 65536) return -1;
    }
    return 0;
}

}  // namespace pamd

// ============================================================================================
// C ABI
// ============================================================================================
using namespace pamd;

static const char *kMessages[8] = {                                        // patolette.c:32-38, then the additive codes
    "Quantization successful.", "Internal quantization error.", "Image dimensions should be greater than 0.",
    "Palette size should be greater than 0.", "Image dimensions are too big.",
    "Saliency weights: image shape not supported (needs more than 3 pixels per side, at least 100 pixels, and a "
    "border band that fits).",
    "Saliency weights: a border band has a singular colour covariance.", nullptr};

#define PAMD_GUARD_BEGIN try { Engine &E = engine(); E.init(); (void)E;
#define PAMD_GUARD_END(ret_fail)                                             \
    } catch (const std::exception &ex) {                                     \
        engine().last_error = ex.what();                                     \
        fprintf(stderr, "%s\n", ex.what());                                  \
        return ret_fail;                                                     \
    }

static int comm_ok(const patolette_amd__Comm *comm) {
    return comm && comm->allreduce_sum && comm->size >= 1 && comm->rank >= 0 && comm->rank < comm->size;
}
static int slice_args_ok(size_t total_pixels, size_t slice_begin, size_t slice_pixels, const void *slice_data,
                         const patolette__QuantizationOptions *options, const void *slice_map) {
    return slice_pixels > 0 && slice_begin <= total_pixels && slice_pixels <= total_pixels - slice_begin && slice_data &&
           (options->palette_only || slice_map);
}
// runs `body` on the calling thread's engine with the slice description attached
// args_ok: this rank's arguments passed slice_args_ok.  A rank that returned early on bad arguments would leave its peers
// waiting in the first collective for ever, so the verdicts are summed over the group first and every rank fails together.
template <typename F>
static void slice_entry(size_t total_pixels, size_t slice_begin, const patolette_amd__Comm *comm, bool args_ok, int *exit_code, F body) {
    Shard sh;
    sh.total = total_pixels; sh.begin = slice_begin; sh.comm = *comm;
    Engine *Ep = nullptr;
    try {
        Engine &E = engine();
        Ep = &E;
        E.init();
        E.shard = &sh;
        int bad = args_ok ? 0 : 1;
        comm_sum_host(E, &bad, 1, 2);
        if (bad) throw HipError(args_ok ? "patolette_amd_slice: another rank of the group passed unusable arguments"
                                        : "patolette_amd_slice: unusable arguments (empty slice, slice outside the image, missing buffer)");
        body(E);
        E.shard = nullptr;
        *exit_code = 0;
    } catch (const std::exception &ex) {
        if (Ep) Ep->shard = nullptr;
        engine().last_error = ex.what();
        fprintf(stderr, "patolette: %s\n", ex.what());
        *exit_code = -1;
    }
}

extern "C" {

void patolette(size_t width, size_t height, const double *data, const double *weights, size_t palette_size,
               const patolette__QuantizationOptions *options, double *palette, size_t *palette_map, int *exit_code) {
    *exit_code = validate(width, height, palette_size);
    if (*exit_code != 0) return;
    try {
        Engine &E = engine();
        E.init();
        run_host(E, width, height, data, weights, 0.0, palette_size, options, palette, palette_map);
        *exit_code = 0;
    } catch (const std::exception &ex) {
        engine().last_error = ex.what();
        fprintf(stderr, "patolette: %s\n", ex.what());
        *exit_code = -1;
    }
}

void patolette_amd_quantize(size_t width, size_t height, const double *data, const double *weights, double tile_size,
                            size_t palette_size, const patolette__QuantizationOptions *options, double *palette,
                            size_t *palette_map, int *exit_code) {
    *exit_code = validate(width, height, palette_size);
    if (*exit_code != 0) return;
    try {
        Engine &E = engine();
        E.init();
        run_host(E, width, height, data, weights, tile_size, palette_size, options, palette, palette_map);
        *exit_code = 0;
    } catch (const CodeError &ex) {
        engine().last_error = ex.what();
        *exit_code = ex.code;
    } catch (const std::exception &ex) {
        engine().last_error = ex.what();
        fprintf(stderr, "patolette: %s\n", ex.what());
        *exit_code = -1;
    }
}

void patolette_amd_quantize_rows(size_t width, size_t height, const double *rows, const double *weights, double tile_size,
                                 size_t palette_size, const patolette__QuantizationOptions *options, double *palette,
                                 size_t *palette_map, int *exit_code) {
    *exit_code = validate(width, height, palette_size);
    if (*exit_code != 0) return;
    try {
        Engine &E = engine();
        E.init();
        run_host(E, width, height, rows, weights, tile_size, palette_size, options, palette, palette_map, true);
        *exit_code = 0;
    } catch (const CodeError &ex) {
        engine().last_error = ex.what();
        *exit_code = ex.code;
    } catch (const std::exception &ex) {
        engine().last_error = ex.what();
        fprintf(stderr, "patolette: %s\n", ex.what());
        *exit_code = -1;
    }
}

void patolette_amd_slice(size_t total_pixels, size_t slice_begin, size_t slice_pixels, const double *slice_data,
                         const double *slice_weights, size_t palette_size, const patolette__QuantizationOptions *options,
                         const patolette_amd__Comm *comm, double *palette, size_t *slice_map, int *exit_code) {
    *exit_code = validate(total_pixels, 1, palette_size);
    if (*exit_code != 0) return;
    if (!comm_ok(comm)) { *exit_code = -1; return; }              // (validate() above is the same on every rank)
    slice_entry(total_pixels, slice_begin, comm, slice_args_ok(total_pixels, slice_begin, slice_pixels, slice_data, options, slice_map),
                exit_code, [&](Engine &E) {
        run_host(E, slice_pixels, 1, slice_data, slice_weights, 0.0, palette_size, options, palette, slice_map);
    });
}

void patolette_amd_slice_device(size_t total_pixels, size_t slice_begin, size_t slice_pixels, const double *d_slice_data,
                                const double *d_slice_weights, size_t palette_size, const patolette__QuantizationOptions *options,
                                const patolette_amd__Comm *comm, double *palette, void *d_slice_map, int map_elem_bytes,
                                int *exit_code) {
    *exit_code = validate(total_pixels, 1, palette_size);
    if (*exit_code != 0) return;
    if (!comm_ok(comm)) { *exit_code = -1; return; }
    slice_entry(total_pixels, slice_begin, comm, slice_args_ok(total_pixels, slice_begin, slice_pixels, d_slice_data, options, d_slice_map),
                exit_code, [&](Engine &E) {
        std::vector<double> pal(3 * palette_size);
        run_device(E, slice_pixels, 1, Pixels{d_slice_data, nullptr, 3}, d_slice_weights, palette_size, options, pal.data(), d_slice_map,
                   map_elem_bytes);
        std::memcpy(palette, pal.data(), 3 * palette_size * sizeof(double));
    });
}

int patolette_amd_set_invariant_sums(int on) {
    const int before = invariant_default() ? 1 : 0;
    tl_invariant = on != 0 ? 1 : 0;
    EngineHolder &h = holder();
    if (h.e) h.e->invariant = on != 0;                // the engine already serving this thread; later ones copy tl_invariant
    return before;
}

int patolette_amd_set_subsample_cache(int on) {
    dither_order_cache(on != 0);             // the other size-only table kept between calls: the dither's curve order (map.hip)
    return g_perm_cache.exchange(on != 0 ? 1 : 0, std::memory_order_relaxed);
}

int patolette_amd_set_kmeans_update(int mode) {
    const int before = km_update_order_free() ? 1 : 0;
    g_km_update.store(mode != 0 ? 1 : 0, std::memory_order_relaxed);
    return before;
}

int patolette_amd_saliency_weights(size_t width, size_t height, const double *data, double tile_size, double *weights_out) {
    const size_t N = width * height;
    if (N == 0) return kSalBadShape;
    try {
        Engine &E = engine();
        E.init();
        E.src.reserve(3 * N);
        HIP_CHECK(hipMemcpyAsync(E.src.p, data, 3 * N * sizeof(double), hipMemcpyHostToDevice, E.stream));
        const double *d_w = derive_weights(E, E.src.p, nullptr, 3, width, height, tile_size);
        HIP_CHECK(hipMemcpy(weights_out, d_w, N * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    } catch (const CodeError &ex) {
        engine().last_error = ex.what();
        return ex.code == -5 ? kSalBadShape : kSalSingular;
    } catch (const std::exception &ex) {
        engine().last_error = ex.what();
        fprintf(stderr, "patolette: %s\n", ex.what());
        return -1;
    }
}

int patolette_amd_mbd(size_t rows, size_t cols, const float *img, int iters, float *out) {
    PAMD_GUARD_BEGIN
    const int rc = mbd_device(E.sal, img, rows, cols, iters, out, E.stream);
    if (ktimer().enabled) ktimer().collect();
    return rc;
    PAMD_GUARD_END(-1)
}

const char *get_patolette_exit_code_info_message(int exit_code) {
    if (exit_code > 0 || exit_code < -6) return nullptr;
    return kMessages[-1 * exit_code];
}

patolette__QuantizationOptions *patolette_create_default_options(void) {   // patolette.c:107-119
    patolette__QuantizationOptions *o = (patolette__QuantizationOptions *)malloc(sizeof *o);
    o->dither = true; o->palette_only = false; o->color_space = patolette__ICtCp;
    o->kmeans_niter = 32; o->kmeans_max_samples = 512 * 512; o->verbose = false;
    return o;
}

int patolette_amd_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}
int patolette_amd_set_device(int ordinal) {
    if (hipSetDevice(ordinal) != hipSuccess) return -1;
    EngineHolder &h = holder();
    if (h.e && h.e->stream && h.e->device != ordinal) { pool_release(h.e); h.e = nullptr; }   // bound to another GPU: swap engines
    if (!h.e) h.e = pool_acquire(ordinal, invariant_default());
    h.e->device = ordinal;
    return 0;
}
void patolette_amd_release_workspace(void) {
    // this thread's engine and every idle pooled one: streams destroyed, device and pinned buffers freed; the next call
    // allocates afresh.  Engines in use by other threads (a batch in flight) are not touched.
    EngineHolder &h = holder();
    std::vector<Engine *> victims;
    if (h.e) { victims.push_back(h.e); h.e = nullptr; }
    { std::lock_guard<std::mutex> lk(g_pool_mu); victims.insert(victims.end(), g_pool.begin(), g_pool.end()); g_pool.clear(); }
    int cur = -1;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;      // ~Engine selects the engine's GPU: put the caller's back
    for (Engine *e : victims) delete e;
    if (have_cur) (void)hipSetDevice(cur);
}
const char *patolette_amd_last_error(void) { return engine().last_error.c_str(); }
void *patolette_amd_malloc(size_t bytes) { void *p = nullptr; if (hipMalloc(&p, bytes) != hipSuccess) return nullptr; return p; }
void patolette_amd_free(void *p) { if (p) (void)hipFree(p); }
int patolette_amd_memcpy_h2d(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1; }
int patolette_amd_memcpy_d2h(void *dst, const void *src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1; }
int patolette_amd_synchronize(void) { return hipDeviceSynchronize() == hipSuccess ? 0 : -1; }
int patolette_amd_fill_image(double *d, size_t n, uint64_t seed) {
    PAMD_GUARD_BEGIN launch_fill_image(d, n, seed, E.stream); E.sync(); return 0; PAMD_GUARD_END(-1)
}
int patolette_amd_fill_weights(double *d, size_t n, uint64_t seed) {
    PAMD_GUARD_BEGIN launch_fill_weights(d, n, seed, E.stream); E.sync(); return 0; PAMD_GUARD_END(-1)
}

void patolette_amd_device(size_t width, size_t height, const double *d_data, const double *d_weights, size_t palette_size,
                          const patolette__QuantizationOptions *options, double *palette, void *d_palette_map,
                          int map_elem_bytes, int *exit_code) {
    *exit_code = validate(width, height, palette_size);
    if (*exit_code != 0) return;
    try {
        Engine &E = engine();
        E.init();
        std::vector<double> pal(3 * palette_size);
        run_device(E, width, height, Pixels{d_data, nullptr, 3}, d_weights, palette_size, options, pal.data(), d_palette_map,
                   map_elem_bytes);
        std::memcpy(palette, pal.data(), 3 * palette_size * sizeof(double));
        *exit_code = 0;
    } catch (const std::exception &ex) {
        engine().last_error = ex.what();
        fprintf(stderr, "patolette: %s\n", ex.what());
        *exit_code = -1;
    }
}

static void u8_entry(bool on_device, size_t width, size_t height, const unsigned char *pixels, int channels,
                     const double *weights, double tile_size, size_t palette_size, const patolette__QuantizationOptions *options, double *palette,
                     unsigned char *palette_u8, void *palette_map, int map_elem_bytes, unsigned char *quantized, int *exit_code) {
    *exit_code = validate(width, height, palette_size);
    if (*exit_code != 0) return;
    if (validate_u8(palette_size, channels, palette_map, map_elem_bytes) != 0) {
        fprintf(stderr, "patolette_amd: bad channels / map_elem_bytes for the u8 entry point\n");
        *exit_code = -1;
        return;
    }
    try {
        Engine &E = engine();
        E.init();
        run_u8(E, width, height, pixels, channels, weights, tile_size, palette_size, options, palette, palette_u8, palette_map,
               map_elem_bytes, quantized, on_device);
        *exit_code = 0;
    } catch (const CodeError &ex) {
        engine().last_error = ex.what();
        *exit_code = ex.code;
    } catch (const std::exception &ex) {
        engine().last_error = ex.what();
        fprintf(stderr, "patolette: %s\n", ex.what());
        *exit_code = -1;
    }
}

void patolette_amd_u8(size_t width, size_t height, const unsigned char *pixels, int channels, const double *weights,
                      double tile_size, size_t palette_size, const patolette__QuantizationOptions *options, double *palette,
                      unsigned char *palette_u8, void *palette_map, int map_elem_bytes, unsigned char *quantized, int *exit_code) {
    u8_entry(false, width, height, pixels, channels, weights, tile_size, palette_size, options, palette, palette_u8, palette_map,
             map_elem_bytes, quantized, exit_code);
}

void patolette_amd_u8_device(size_t width, size_t height, const unsigned char *d_pixels, int channels, const double *d_weights,
                             double tile_size, size_t palette_size, const patolette__QuantizationOptions *options, double *palette,
                             unsigned char *palette_u8, void *d_palette_map, int map_elem_bytes, unsigned char *d_quantized,
                             int *exit_code) {
    u8_entry(true, width, height, d_pixels, channels, d_weights, tile_size, palette_size, options, palette, palette_u8, d_palette_map,
             map_elem_bytes, d_quantized, exit_code);
}

// Independent images: up to six are in flight at once, each on its own engine (HIP stream + workspace) driven by
// its own host thread, so the upload / host-side split-loop work of one image overlaps the kernels of another.
// Engines are pooled per device and reused across calls.

// run item(E, i) for i in [0, count) on pooled engines, `workers` of them in flight
static void batch_run(size_t count, size_t width, size_t height, const patolette__QuantizationOptions *options, int *exit_codes,
                      const std::function<void(Engine &, size_t)> &item) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) { for (size_t i = 0; i < count; i++) exit_codes[i] = -1; return; }
    if (engine().device >= 0) device = engine().device;
    // Images in flight: six keep the GPU busy through the host round trips between an image's split rounds (device-resident
    // 4096^2 images, one MI355X: 2960 Mpx/s one at a time, 3330 / 3410 / 3740 / 3990 / 3880 with 2 / 3 / 4 / 6 / 8 in flight).  The
    // Riemersma dither fills the GPU by itself since it is cut into runs (map.hip DitherSeg), so dithered batches keep the same
    // number in flight.  Bounded by the free HBM: an engine's workspace is ~170 bytes per pixel.
    const size_t base_flight = getenv("PAMD_BATCH_FLIGHT") ? std::max<size_t>(1, (size_t)atoll(getenv("PAMD_BATCH_FLIGHT"))) : 6;
    size_t workers = std::min<size_t>(count, base_flight);
    if (count > 1) {
        size_t free_b = 0, total_b = 0;
        (void)hipSetDevice(device);
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const size_t per_engine = (size_t)(170.0 * (double)width * (double)height) + ((size_t)64 << 20);
            size_t idle = 0;                                     // pooled engines whose workspace already covers this image size
            {
                std::lock_guard<std::mutex> lk(g_pool_mu);
                for (Engine *e : g_pool) if (e->device == device && e->cvt.cap >= 3 * width * height) idle++;
            }
            const size_t fit = idle + free_b / 2 / per_engine;   // leave half of what is free alone
            workers = std::min<size_t>(count, std::max<size_t>(1, std::min<size_t>(fit, base_flight)));
        }
    }
    std::atomic<size_t> next{0};
    const bool invariant = invariant_default();
    auto work = [&]() {
        Engine *E = nullptr;
        try {
            (void)hipSetDevice(device);
            E = pool_acquire(device, invariant);                 // the CALLER's setting (this may be a helper thread)
            E->init();
        } catch (const std::exception &ex) {
            fprintf(stderr, "patolette: %s\n", ex.what());
            for (size_t i; (i = next.fetch_add(1)) < count;) exit_codes[i] = -1;
            if (E) pool_release(E);
            return;
        }
        for (size_t i; (i = next.fetch_add(1)) < count;) {
            try {
                item(*E, i);
                exit_codes[i] = 0;
            } catch (const CodeError &ex) {
                exit_codes[i] = ex.code;
            } catch (const std::exception &ex) {
                fprintf(stderr, "patolette: %s\n", ex.what());
                exit_codes[i] = -1;
            }
        }
        pool_release(E);
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < workers; t++) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
}

static void batch_impl(bool rows, size_t count, size_t width, size_t height, const double *const *data, const double *const *weights,
                       double tile_size, size_t palette_size, const patolette__QuantizationOptions *options, double *const *palettes,
                       size_t *const *palette_maps, int *exit_codes) {
    const int v = validate(width, height, palette_size);
    if (v != 0) { for (size_t i = 0; i < count; i++) exit_codes[i] = v; return; }
    batch_run(count, width, height, options, exit_codes, [&](Engine &E, size_t i) {
        run_host(E, width, height, data[i], weights ? weights[i] : nullptr, tile_size, palette_size, options, palettes[i],
                 palette_maps ? palette_maps[i] : nullptr, rows);
    });
}

void patolette_amd_batch_u8(size_t count, size_t width, size_t height, const unsigned char *const *pixels, int channels,
                            const double *const *weights, double tile_size, size_t palette_size,
                            const patolette__QuantizationOptions *options, double *const *palettes, unsigned char *const *palettes_u8,
                            void *const *palette_maps, int map_elem_bytes, unsigned char *const *quantized, int *exit_codes) {
    int v = validate(width, height, palette_size);
    if (v == 0 && validate_u8(palette_size, channels, palette_maps ? (const void *)palette_maps : nullptr, map_elem_bytes) != 0) {
        fprintf(stderr, "patolette_amd: bad channels / map_elem_bytes for the u8 entry point\n");
        v = -1;
    }
    if (v != 0) { for (size_t i = 0; i < count; i++) exit_codes[i] = v; return; }
    batch_run(count, width, height, options, exit_codes, [&](Engine &E, size_t i) {
        run_u8(E, width, height, pixels[i], channels, weights ? weights[i] : nullptr, tile_size, palette_size, options, palettes[i],
               palettes_u8 ? palettes_u8[i] : nullptr, palette_maps ? palette_maps[i] : nullptr, map_elem_bytes,
               quantized ? quantized[i] : nullptr, false);
    });
}

void patolette_amd_batch_dmap(size_t count, size_t width, size_t height, const void *const *images, int pixel_format,
                              const double *const *weights, double tile_size, size_t palette_size,
                              const patolette__QuantizationOptions *options, double *const *palettes, void *d_palette_maps,
                              int map_elem_bytes, int *exit_codes) {
    int v = validate(width, height, palette_size);
    if (v == 0 && pixel_format != 0 && pixel_format != 1 && pixel_format != 3 && pixel_format != 4) v = -1;
    if (v == 0 && d_palette_maps && map_elem_bytes != map_elem_for(palette_size)) {
        fprintf(stderr, "patolette_amd: device map_elem_bytes must be 1 for K <= 256, else 4\n");
        v = -1;
    }
    if (v != 0) { for (size_t i = 0; i < count; i++) exit_codes[i] = v; return; }
    const size_t N = width * height;
    batch_run(count, width, height, options, exit_codes, [&](Engine &E, size_t i) {
        void *d_map = d_palette_maps ? (void *)((unsigned char *)d_palette_maps + i * N * (size_t)map_elem_bytes) : nullptr;
        const double *w = weights ? weights[i] : nullptr;
        if (pixel_format <= 1)
            run_host(E, width, height, (const double *)images[i], w, tile_size, palette_size, options, palettes[i], nullptr,
                     pixel_format == 1, d_map);
        else
            run_u8(E, width, height, (const unsigned char *)images[i], pixel_format, w, tile_size, palette_size, options, palettes[i],
                   nullptr, d_map, map_elem_bytes, nullptr, false, true);
    });
}

int patolette_amd_principal_axis(const double cov6[6], double axis[3]) {
    return hm::principal_axis(cov6, axis) ? 0 : -1;
}
int patolette_amd_subsample_indices(size_t n, size_t take, int32_t *out) {
    if (!out || take > n) return -1;
    hm::rand_perm_prefix(n, take, 1234u, out);                              // host code: needs no device
    return 0;
}
int patolette_amd_eigen_sym3_device(const double *a_colmajor, size_t count, double *w, double *z, int *info) {
    try {
        if (!count) return 0;
        DevBuf<double> da, dw, dz; DevBuf<int> di;
        da.reserve(9 * count); dw.reserve(3 * count); dz.reserve(9 * count); di.reserve(count);
        HIP_CHECK(hipMemcpy(da.p, a_colmajor, 9 * count * sizeof(double), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_eigen_batch, (unsigned)((count + 63) / 64), 64, 0, nullptr, da.p, count, dw.p, dz.p, di.p);
        HIP_CHECK(hipGetLastError());
        HIP_CHECK(hipMemcpy(w, dw.p, 3 * count * sizeof(double), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(z, dz.p, 9 * count * sizeof(double), hipMemcpyDeviceToHost));
        HIP_CHECK(hipMemcpy(info, di.p, count * sizeof(int), hipMemcpyDeviceToHost));
        return 0;
    } catch (const std::exception &e) { engine().last_error = e.what(); return -1; }
}
int patolette_amd_eigen_sym3(const double a_colmajor[9], double w[3], double z[9]) {
    std::memcpy(z, a_colmajor, 9 * sizeof(double));
    return hm::eigen_sym3(z, w);                                           // LAPACK info: 0 = converged
}

void patolette_amd_batch(size_t count, size_t width, size_t height, const double *const *data, const double *const *weights,
                         double tile_size, size_t palette_size, const patolette__QuantizationOptions *options, double *const *palettes,
                         size_t *const *palette_maps, int *exit_codes) {
    batch_impl(false, count, width, height, data, weights, tile_size, palette_size, options, palettes, palette_maps, exit_codes);
}
void patolette_amd_batch_rows(size_t count, size_t width, size_t height, const double *const *rows, const double *const *weights,
                              double tile_size, size_t palette_size, const patolette__QuantizationOptions *options,
                              double *const *palettes, size_t *const *palette_maps, int *exit_codes) {
    batch_impl(true, count, width, height, rows, weights, tile_size, palette_size, options, palettes, palette_maps, exit_codes);
}

int patolette_amd_pow(const double *x, double y, double *out, size_t n) {
    PAMD_GUARD_BEGIN
    if (n == 0) return 0;
    E.src.reserve(2 * n);
    HIP_CHECK(hipMemcpyAsync(E.src.p, x, n * sizeof(double), hipMemcpyHostToDevice, E.stream));
    launch_pow(E.src.p, y, E.src.p + n, n, E.stream);
    HIP_CHECK(hipMemcpyAsync(out, E.src.p + n, n * sizeof(double), hipMemcpyDeviceToHost, E.stream));
    E.sync();
    return 0;
    PAMD_GUARD_END(-1)
}

int patolette_amd_convert(int which, double *planar, size_t n) {
    PAMD_GUARD_BEGIN
    if (n == 0) return 0;
    E.src.reserve(3 * n); E.aux.reserve(3 * n);
    HIP_CHECK(hipMemcpy(E.src.p, planar, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    launch_convert(which, E.src.p, E.aux.p, n, nullptr, E.stream);
    E.sync();
    HIP_CHECK(hipMemcpy(planar, E.aux.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
    PAMD_GUARD_END(-1)
}

int patolette_amd_quantize_clusters(const double *colors, const double *weights, size_t n, size_t K, double *centers, size_t *n_clusters) {
    PAMD_GUARD_BEGIN
    if (n == 0 || K == 0) return -1;
    const bool weighted = weights != nullptr;
    E.src.reserve(3 * n);
    E.cvt.reserve((weighted ? 4 : 3) * n);
    E.cstats.reserve(1);
    HIP_CHECK(hipMemcpy(E.src.p, colors, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    launch_convert(PAMD_COPY, E.src.p, E.cvt.p, n, E.cstats.p, E.stream);
    if (weighted) {
        HIP_CHECK(hipMemcpyAsync(E.cvt.p + 3 * n, weights, n * sizeof(double), hipMemcpyHostToDevice, E.stream));
        launch_weight_stats(E.cvt.p + 3 * n, n, E.cstats.p, E.stream);
    }
    Bounds b = read_bounds(E, weighted);
    E.stats = patolette_amd__Stats{};
    std::vector<double> pal;
    size_t len = 0;
    if (quantize_clusters(E, n, K, weighted, b, pal, len) != 0) return -1;
    for (size_t j = 0; j < 3 * K; j++) centers[j] = NAN;
    for (int j = 0; j < 3; j++) for (size_t i = 0; i < len; i++) centers[(size_t)j * K + i] = pal[(size_t)j * len + i];
    *n_clusters = len;
    return 0;
    PAMD_GUARD_END(-1)
}

int patolette_amd_kmeans_refine(const double *colors, const double *weights, size_t n, double *centers_io, size_t k, int niter, size_t max_samples) {
    PAMD_GUARD_BEGIN
    const bool weighted = weights != nullptr;
    E.cvt.reserve((weighted ? 4 : 3) * n);
    HIP_CHECK(hipMemcpy(E.cvt.p, colors, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    if (weighted) HIP_CHECK(hipMemcpy(E.cvt.p + 3 * n, weights, n * sizeof(double), hipMemcpyHostToDevice));
    std::vector<double> pal(centers_io, centers_io + 3 * k);
    bool nonfinite = false;                                     // Clustering.cpp:295-304 over every colour value as f32
    for (size_t i = 0; i < 3 * n && !nonfinite; i++) nonfinite = !(std::fabs(colors[i]) < 0x1.ffffffp127);
    double bx = 0, bw = 1;
    if (!nonfinite && km_update_order_free()) {
        for (size_t i = 0; i < 3 * n; i++) bx = std::max(bx, std::fabs(colors[i]));
        if (weighted) for (size_t i = 0; i < n; i++) bw = std::max(bw, std::fabs(weights[i]));
    }
    kmeans_refine(E, n, weighted, pal, k, niter, max_samples, nonfinite, 0, bx, bw);
    std::memcpy(centers_io, pal.data(), 3 * k * sizeof(double));
    return 0;
    PAMD_GUARD_END(-1)
}

int patolette_amd_nn_map(const double *colors, size_t n, const double *palette, size_t k, size_t *map) {
    PAMD_GUARD_BEGIN
    if (n == 0) return 0;
    E.src.reserve(3 * n); E.dpal.reserve(3 * k); E.dmap.reserve(n * 4);
    HIP_CHECK(hipMemcpy(E.src.p, colors, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(E.dpal.p, palette, 3 * k * sizeof(double), hipMemcpyHostToDevice));
    launch_nn_map(E.src.p, n, n, E.dpal.p, (int)k, E.dmap.p, 4, nullptr, nullptr, E.nn, E.stream);
    E.sync();
    std::vector<unsigned int> tmp(n);
    HIP_CHECK(hipMemcpy(tmp.data(), E.dmap.p, n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) map[i] = tmp[i];
    return 0;
    PAMD_GUARD_END(-1)
}

int patolette_amd_dither(const double *colors, size_t width, size_t height, const double *palette, size_t k, size_t *map) {
    PAMD_GUARD_BEGIN
    const size_t n = width * height;
    if (n == 0) return 0;
    E.src.reserve(3 * n); E.dpal.reserve(3 * k); E.dmap.reserve(n * 4);
    HIP_CHECK(hipMemcpy(E.src.p, colors, 3 * n * sizeof(double), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(E.dpal.p, palette, 3 * k * sizeof(double), hipMemcpyHostToDevice));
    launch_dither(E.src.p, n, PAMD_COPY, width, height, E.dpal.p, palette, (int)k, E.dmap.p, 4, E.nn, E.stream);
    E.stats.dither_segments = E.nn.dither_segments; E.stats.dither_repairs = E.nn.dither_repairs; E.stats.dither_rounds = E.nn.dither_rounds; E.stats.dither_through = E.nn.dither_through;
    E.stats.dither_jumps = E.nn.dither_jumps; E.stats.dither_solo = E.nn.dither_solo;
    E.sync();
    if (std::max(width, height) > 1) {
        std::vector<unsigned int> tmp(n);
        HIP_CHECK(hipMemcpy(tmp.data(), E.dmap.p, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) map[i] = tmp[i];
    }
    return 0;
    PAMD_GUARD_END(-1)
}

int patolette_amd_debug_fault(int which) { return g_debug_fault.exchange(which); }
int patolette_amd_set_split_loop(int mode) { return g_lq_device.exchange(mode < 0 || mode > 2 ? 2 : mode); }
int patolette_amd_set_global_quantiser(int on_device) { return g_gq_device.exchange(on_device ? 1 : 0); }
void patolette_amd_dither_config(int segments, int warm) { dither_config(segments, warm); }
void patolette_amd_dither_layout(int lanes) { dither_layout(lanes); }
int patolette_amd_debug_dither_solo_cap(int cap) { return dither_solo_cap(cap); }
int patolette_amd_debug_dither_stall_passes(int n) { return dither_stall_passes(n); }
int patolette_amd_dither_layout_in_use(size_t width, size_t height, size_t k) { return dither_lane_layout(width, height, (int)k) ? 1 : 0; }
void patolette_amd_last_stats(patolette_amd__Stats *out) { *out = engine().stats; }
size_t patolette_amd_last_split_trace(patolette_amd__SplitTrace *hdr, patolette_amd__SplitRecord *recs, size_t capacity) {
    Engine &E = engine();
    try { materialise_trace(E); } catch (const std::exception &e) { E.last_error = e.what(); E.trace.clear(); }
    if (hdr) *hdr = E.trace_hdr;
    if (recs) std::memcpy(recs, E.trace.data(), sizeof *recs * std::min(capacity, E.trace.size()));
    return E.trace.size();
}
size_t patolette_amd_last_cluster_centers(double *out, size_t capacity_rows) {
    Engine &E = engine();
    const size_t len = E.cluster_centers.size() / 3;
    if (out && len <= capacity_rows)
        for (int j = 0; j < 3; j++) for (size_t i = 0; i < len; i++) out[(size_t)j * capacity_rows + i] = E.cluster_centers[(size_t)j * len + i];
    return len;
}
size_t patolette_amd_last_map_palette(double *out, size_t capacity_rows) {
    const std::vector<double> &mp = engine().map_palette;
    const size_t len = mp.size() / 3;
    if (out && capacity_rows >= len)
        for (int j = 0; j < 3; j++) for (size_t i = 0; i < len; i++) out[(size_t)j * capacity_rows + i] = mp[(size_t)j * len + i];
    return len;
}

void patolette_amd_profile_enable(int on) {
    KernelTimer &t = ktimer();
    if (engine().stream) (void)hipStreamSynchronize(engine().stream);
    t.reset();
    t.enabled = on != 0;
}
void patolette_amd_profile_only(const char *kernel_name) { ktimer().only = kernel_name ? kernel_name : ""; ktimer().only_seen = 0; }
void patolette_amd_profile_sample(int period) { ktimer().sample_period = period > 1 ? (unsigned)period : 1u; }
int patolette_amd_profile_count(void) { return (int)ktimer().names.size(); }
int patolette_amd_profile_get(int i, char *name64, double *total_ms, size_t *launches, double *total_bytes) {
    KernelTimer &t = ktimer();
    if (i < 0 || i >= (int)t.names.size()) return -1;
    snprintf(name64, 64, "%s", t.names[i].c_str());
    *total_ms = t.total_ms[i];
    *launches = t.launches[i];
    *total_bytes = t.total_bytes[i];
    return 0;
}

}  // extern "C"
