// kmeans.h -- KMeans refinement workspace + launchers (kmeans.hip).
#pragma once

#include "common.h"
#include "devutil.h"

namespace pamd {

constexpr int kKMeansMaxK = 4096;      // up to here the stable sort keeps per-wavefront counters in LDS (4 waves x K x 4 B); beyond: counters in memory

struct KmSamples { float *x, *y, *z, *w; };

struct KMeansWork {
    DevBuf<float> sx, sy, sz, sw;      // samples, f32 SoA
    DevBuf<int> assign;
    DevBuf<float4> sorted;             // samples grouped by centroid, sample order kept
    DevBuf<unsigned int> table, rowtot, ticket;
    DevBuf<unsigned int> flags;        // list path: [0] fixed point reached, [1 + j] centroid j dirty (kmeans.hip, k_km_prep)
    DevBuf<unsigned int> rowbase;      // k > kKMeansMaxK: exclusive prefix of rowtot
    DevBuf<float> hs;                  // k > kKMeansMaxK: split_clusters' working copy of the cluster sizes
    DevBuf<float> cent, hassign;       // interleaved xyz centroids (faiss layout)
    DevBuf<float4> c4;                 // (y0,y1,y2,|y|^2)
    DevBuf<int> perm;                  // subsample indices
    DevBuf<struct DevMT> mt;           // std::mt19937(1234) as seeded (read-only after k_km_mt_seed)
    bool mt_seeded = false;
    DevBuf<unsigned char> lut, grid, clist;   // pruned assignment: candidate records per grid cell, sample bounding box, coarse lists
    DevBuf<unsigned int> mid;                 // 32^3 table of four-candidate entries (held in LDS by k_km_assign_mid)
    DevBuf<unsigned int> bkeys;
    DevBuf<unsigned long long> fsum;          // order-free update (KmSums): per centroid 4 fixed-point sums (x, y, z, weight or count)
    void reserve(size_t nx, int k);
};

void kmeans_gather(const double *d_planar, size_t N, bool weighted, const int *d_perm, size_t nx, KMeansWork &w, hipStream_t s);
// the same for a GPU that holds pixels [begin, begin + n_local) of the image: foreign samples are written as zero bits
void kmeans_gather_slice(const double *d_planar, size_t n_local, bool weighted, const int *d_perm, size_t nx, size_t begin,
                         KMeansWork &w, hipStream_t s);
// expect_longest: samples the most populous centroid is expected to hold (0 = unknown).  The list-collecting update of the
// few-samples path replays a centroid's chain at one sample per dependent add (3.3 ns), so one long cluster paces every
// iteration (a colour covering 30 % of the image: 46 ms instead of 5); from 4096 expected samples on the iterations take the
// sorted path, whose update sums long clusters block-parallel (km_chain_coop).
// sums: null = the reference's update bit for bit (sequential f32 chains in sample order).  Otherwise the ORDER-FREE update
// (patolette_amd_set_kmeans_update(1)): a centroid's sums are taken as 64-bit fixed-point integers -- exact in any order, so still
// run-to-run and launch-geometry deterministic -- and rounded to f32 once; no sort, no chain.  The centroids then differ from the
// reference's by what ITS f32 chains round away (~1e-6 of the colour range per iteration) and from there by the samples whose
// nearest centroid that flips (|x - c| / members each).
struct KmSums { double bound_x, bound_w; };   // |coordinate| <= bound_x and 0 <= weight <= bound_w over all samples
void kmeans_iterate(KMeansWork &w, size_t nx, int k, bool weighted, int niter, hipStream_t s, size_t expect_longest = 0,
                    const KmSums *sums = nullptr);

}  // namespace pamd
