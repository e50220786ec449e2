// quant.h -- device records and launchers of the cluster-split kernels (quant.hip).
#pragma once

#include <atomic>

#include "common.h"
#include "devutil.h"

namespace pamd {

constexpr int kMaxChildren = 16;       // GQ makes <= 12 base clusters (global.c:19), LQ splits in 2
constexpr int kTileA = 8192;           // pixels per block for streaming reductions (minmax / hist / cov)
constexpr int kTileP = 2048;           // pixels per block for the stable partition (256 threads x 8 rounds)

// quantities accumulated per bucket by k_hist (two binned parts each)
//   LQ : 0..2 = sum c*w, 3 = sum w
//   GQ : 0..2 = sum c, 3 = sum |c|^2, 4..9 = sum c_r*c_s (00,01,11,02,12,22), 10..12 = sum c*w, 13 = sum w
constexpr int kNQ_LQ = 4;
constexpr int kNQ_GQ = 14;

constexpr int kSlots = 16;             // accumulator copies per node: same-address atomics serialise in L2, so tiles
                                       // of one node spread over kSlots addresses; slots are summed (exactly) on read-out

// host -> device (k_put_nodes)
struct NodeIn {
    unsigned long long begin, n, gn;   // gn: members over ALL GPUs of a within-image shard group (= n on one GPU)
    int buf, slot, child0, nchild;
    int split;                         // >= 0: the children are "bucket <= split" / "bucket > split" (two base clusters out of the global quantiser)
    double axis[3], mean[3], sw;
    BinK klin, kquad;
};
// device -> host (k_get_nodes)
struct NodeOut {
    unsigned long long begin, n, gn;
    int buf, degenerate, split, psplit;
    double sw, mean[3];
    double acc[7][2];                  // slot sums: 6 covariance sums (xx,yx,zx,yy,zy,zz) + distortion, 2 binned parts each
};

struct NodeDev {
    // ---- inputs (host or k_cut writes them)
    unsigned long long begin;          // first pixel slot of the segment
    unsigned long long n;              // pixels (of this GPU's slice of the image)
    unsigned long long gn;             // pixels over all GPUs sharing the image (= n when one GPU holds it all)
    unsigned long long gslot0;         // members held by lower-ranked GPUs: the node-wide slot of this GPU's first member
    int buf;                           // ping-pong buffer holding the segment
    int slot;                          // histogram slot in the current round (-1: not being split)
    int child0;                        // id of first child record (children are consecutive)
    int nchild;
    double axis[3];
    double mean[3];                    // weighted centre
    double sw;                         // sum of weights (n if unweighted)
    BinK klin, kquad;                  // binned-accumulation constants valid for this node and its children
    // ---- split evaluation outputs
    unsigned long long minkey[kSlots], maxkey[kSlots];   // ordered keys of the projection extrema
    int degenerate;                    // max - min < 1e-16 -> round-robin buckets (sort.c:61-79)
    int split;                         // optimal bucket index (local.c:171)
    int psplit;                        // the PARENT's cut as k_cut took it: bucket | degenerate << 16 (for the split trace; -1: a base cluster)
    int axis_state;                    // device-driven split loop: 0 moments not final yet, 1 axis solved, -1 the eigen-solver failed
    double cov6[6], dist, ub;          // ... the node's centred sums (xx,yx,zx,yy,zy,zz), distortion, upper bound of any split's benefit
    unsigned long long cbegin[kMaxChildren + 1];   // children segments in the other buffer
    // ---- moments about `mean` (k_cov / k_scatter): 6 covariance sums + distortion, 2 parts each
    double acc[kSlots][7][2];
};

__device__ __forceinline__ void node_reset_outputs(NodeDev &d) {
    for (int i = 0; i < kSlots; i++) { d.minkey[i] = ~0ULL; d.maxkey[i] = 0ULL; }
    for (int i = 0; i < kSlots; i++) for (int q = 0; q < 7; q++) { d.acc[i][q][0] = 0; d.acc[i][q][1] = 0; }
    d.degenerate = 0; d.split = -1; d.gslot0 = 0;
}
// the same, spread over threads: element e of kNodeResetElems (one store each, the scalars ride with element 0)
constexpr int kNodeResetElems = kSlots * 14 + 2 * kSlots;
__device__ __forceinline__ void node_reset_element(NodeDev &d, int e) {
    if (e < kSlots * 14) (&d.acc[0][0][0])[e] = 0.0;
    else if (e < kSlots * 15) d.minkey[e - kSlots * 14] = ~0ULL;
    else d.maxkey[e - kSlots * 15] = 0ULL;
    if (e == 0) { d.degenerate = 0; d.split = -1; d.gslot0 = 0; }
}
__device__ __forceinline__ void node_minmax(const NodeDev &d, double &mn, double &mx) {
    unsigned long long a = ~0ULL, b = 0ULL;
    for (int i = 0; i < kSlots; i++) { a = d.minkey[i] < a ? d.minkey[i] : a; b = d.maxkey[i] > b ? d.maxkey[i] : b; }
    mn = key_f64(a); mx = key_f64(b);
}

// Sizes of a split round that only the DEVICE knows (the split loop's control kernel, pipeline.hip k_lq_control, writes them):
// the sweep kernels are then launched with upper bounds and read the real extent here.  A null pointer everywhere else.
struct RoundDyn {
    int nr;                            // nodes evaluated in this round
    int ntA, ntP;                      // tiles of both tilings
    int pad;
};

// Device-driven rounds: what the round's first sweep (k_minmax) needs to set the round up itself -- the tile lists of both tilings
// and zeroed bucket tables -- instead of a launch of its own in front of it (pipeline.hip k_round_setup_dyn did this)
struct RoundLists {
    const int *round_ids, *tA0, *tP0;      // the control kernel's lists (LqCtl)
    struct Tile *tilesA, *tilesP;
    double *hist; size_t lqs;              // doubles per node in `hist`
    unsigned long long *hsize; unsigned int *hcount;
};

struct Tile {
    unsigned long long start;          // absolute pixel slot
    unsigned int count;
    unsigned int node;                 // NodeDev index
};

struct QuantBuffers {
    double *buf[2];                    // planar x|y|z(|w): plane p at buf[b] + p*N
    unsigned short *bkt;               // bucket id per pixel slot
    size_t N;
    bool weighted;
};

// all launchers enqueue on `s` and return immediately
constexpr int kSum3Slots = 32;         // d_out6 holds kSum3Slots x 6 partial sums (exact parts: the reader adds them in any order)
void launch_sum3(const double *planar, size_t N, BinK k, double *d_out6, hipStream_t s);
// global quantiser DP state on the device (global.c:189-298)
constexpr int kGqMaxK = 12;                // global.c:23 max_k
struct GqDpDev {
    unsigned long long w0[kBuckets + 1];
    double w1[3][kBuckets + 1], w2[kBuckets + 1];
    double E[2][kBuckets + 1];
    int cut[kGqMaxK + 1][kBuckets + 1];    // cut[k][n] = L[k][n] of the reference
};
void launch_gq_dp(const double *d_hist, const unsigned int *d_hcount, int kmax, GqDpDev *d_g, hipStream_t s);

// from_end (every sweep launcher): the blocks take the tiles from the end of the list.  The sweeps of a split round alternate,
// so that each starts on the lines the previous one touched last
// dyn (every launcher below): the launch covers `ntiles` / `nround` as UPPER BOUNDS and the kernels take the real sizes from *dyn;
// px_src then points at the round's pixel count (a double the control kernel wrote to pinned host memory) for the kernel timer
// lists (with dyn): the kernel builds the round's tile lists and clears its bucket tables itself (d_tiles is ignored)
void launch_minmax(const QuantBuffers &qb, const Tile *d_tiles, int ntiles, size_t px, NodeDev *d_nodes, hipStream_t s, bool from_end = false,
                   const RoundDyn *dyn = nullptr, const double *px_src = nullptr, const RoundLists *lists = nullptr);
// fixed_point (global quantiser only): block-local sums as 64-bit integers (k_hist_fix); a property of the IMAGE (its total
// pixel count), so that every GPU sharing an image takes the same path
void launch_hist(const QuantBuffers &qb, bool gq, const Tile *d_tiles, int ntiles, size_t px, NodeDev *d_nodes,
                 double *d_hist, unsigned long long *d_hsize, unsigned int *d_hcount, hipStream_t s, bool from_end = false,
                 bool fixed_point = false, const RoundDyn *dyn = nullptr, const double *px_src = nullptr);
void launch_cut(bool weighted, NodeDev *d_nodes, const int *d_round_nodes, int nround, const double *d_hist,
                const unsigned long long *d_hsize, const unsigned int *d_hcount, unsigned char *d_lut, hipStream_t s,
                const RoundDyn *dyn = nullptr, const double *nr_src = nullptr);
void launch_partition(const QuantBuffers &qb, const Tile *d_ptiles, int nptiles, size_t px, const int *d_round_nodes,
                      const int *d_node_tile0, int nround, NodeDev *d_nodes, const unsigned char *d_lut,
                      unsigned int *d_tilecnt, unsigned long long *d_tileoff, bool fuse_cov, hipStream_t s,
                      bool invariant = false, bool from_end = false, const RoundDyn *dyn = nullptr, const double *px_src = nullptr,
                      bool gated = false);
void launch_cov_children(const QuantBuffers &qb, const Tile *d_tiles, int ntiles, size_t px, NodeDev *d_nodes, hipStream_t s, bool from_end = false,
                         bool gated = false);
void launch_cov_nodes(const QuantBuffers &qb, const double *planar_override, const Tile *d_tiles, int ntiles, size_t px,
                      NodeDev *d_nodes, hipStream_t s, bool from_end = false);

extern std::atomic<int> g_debug_fault;   // patolette_amd_debug_fault (tests only): 0 none, 1 wrong cut, 2 wrong greedy step, 3 last arg-max
size_t hist_slot_doubles();            // doubles per histogram slot

}  // namespace pamd
