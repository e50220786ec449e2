/*
 * patolette_oracle.h -- CPU restatement of big-nacho/patolette's quantisation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under patolette_amd/ (the product) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and only as the checker / CPU baseline -- never as the thing measured or shipped.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/).  See oracle/README.md for what is pinned against a build of the
 * reference's own sources (oracle/_ref) and what is "parity unpinned".
 */
#ifndef PATOLETTE_ORACLE_H
#define PATOLETTE_ORACLE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as lib/include/patolette.h:7-20 */
typedef enum { ORC_sRGB = 0, ORC_CIELuv = 1, ORC_ICtCp = 2 } orc_ColorSpace;
typedef struct {
    bool dither;
    bool palette_only;
    orc_ColorSpace color_space;
    int kmeans_niter;
    size_t kmeans_max_samples;
    bool verbose;
} orc_Options;

/* ---- synthetic input generator: u(seed,i) = (splitmix64(seed*2^32+i)>>11)*2^-53 */
uint64_t orc_splitmix64(uint64_t x);
void orc_fill_uniform(double *out, size_t n, uint64_t seed);                 /* out[i] = u(seed,i) */
void orc_fill_image(double *planar, size_t n, uint64_t image_seed);          /* plane p -> u(1000*s+p, i) */
void orc_fill_weights(double *w, size_t n, uint64_t image_seed);             /* 1 + 3*u(1000*s+7, i) */

/* ---- colour conversions, in place on planar (n,3) column-major f64 (lib/src/color/ *.c) */
void orc_srgb_to_ictcp(double *m, size_t n);        /* ICtCp.c:120-146 */
void orc_srgb_to_cieluv(double *m, size_t n);       /* CIELuv.c:166-197 */
void orc_ictcp_to_rec2020(double *m, size_t n);     /* rec2020.c:128-148 */
void orc_cieluv_to_rec2020(double *m, size_t n);    /* rec2020.c:150-173 */
void orc_srgb_to_rec2020(double *m, size_t n);      /* rec2020.c:175-195 */
void orc_rec2020_to_srgb(double *m, size_t n);      /* sRGB.c:112-132 */

/* ---- 3x3 symmetric eigen-solve restating LAPACK dsyev('V','L') for n=3 (math/eigen.c:83-140).
 * a: column-major 3x3, lower triangle read; on exit eigenvectors as columns (ascending
 * eigenvalues in w).  Returns 0 on success. */
int orc_eigen_sym3(double a[9], double w[3]);

/* ---- weighted PCA of n rows of planar colours c (n,3) (math/pca.c:62-168). weights may be
 * NULL.  axis[3] = eigenvector of the largest eigenvalue (sign as the solver returns it). */
int orc_pca_axis(const double *c, const double *weights, size_t n, double axis[3], double vcov_out[9]);

/* ---- bucket sort along an axis (quantize/sort.c:12-91) */
void orc_axis_sort(const double *c, size_t n, const double axis[3], size_t bucket_count, size_t *bucket_map);

/* ---- GQ + LQ + PALETTE_create (quantize/global.c:388-443, local.c:318-404, palette/create.c:11-33).
 * colors: planar (n,3) in the quantisation space.  On return: *n_clusters <= K, centers
 * planar (K,3) (first *n_clusters rows valid), cluster_of[i] = palette row of pixel i
 * (may be NULL), n_base = clusters produced by GQ, split_evals / split_px = number of
 * split_cluster evaluations and the sum of their sizes (D_eff = split_px / n).
 * Returns 0, or -1 on internal error. */
int orc_quantize_clusters(const double *colors, const double *weights, size_t n, size_t K,
                          double *centers, size_t *n_clusters, uint32_t *cluster_of,
                          size_t *n_base, size_t *split_evals, size_t *split_px);

/* ---- split trace of the last orc_quantize_clusters / orc_patolette call (tests/tie_prover.py): what the global
 * quantiser decided and, per committed split of the greedy loop (local.c:347-390, in commit order), which cluster was
 * split where.  The HIP path exports the same records (include/patolette_amd.h: patolette_amd_last_split_trace). */
typedef struct {
    int32_t row;          /* index in `result` of the cluster that was split (`best`, local.c:348-352) */
    int32_t new_row;      /* index that received the LEFT child (= clusters before the commit, local.c:375) */
    int32_t split;        /* bucket index of the cut (local.c:205-208) */
    int32_t degenerate;   /* the projection took sort.c:61-79's round-robin rule */
    uint64_t n, n_left, n_right;
    double sw;            /* sum of the cluster's weights */
    double axis[3];       /* principal axis incl. sign (cluster.c:191-217) */
    double cov6[6];       /* the matrix handed to dsyev: xx, yx, zx, yy, zy, zz (pca.c:62-101) */
    double dist, dist_left, dist_right, benefit;   /* cluster.c:111-152, local.c:256-275 */
} orc_SplitRecord;
typedef struct {
    int32_t n_base, n_clusters, n_records, stopped_early;   /* stopped_early: benefit < 1e-16 (local.c:365-370) */
    double gq_axis[3];
    uint64_t gq_cuts[14];                                    /* n_base + 1 entries (global.c:290) */
    double gq_cov6[6];                                       /* xx, yx, zx, yy, zy, zz handed to dsyev (global.c:407) */
} orc_SplitTraceHeader;
size_t orc_last_split_trace(orc_SplitTraceHeader *hdr, orc_SplitRecord *recs, size_t capacity);
/* test knobs: see patolette_oracle.c */
void orc_set_sum_reversed(int on);
void orc_set_fault(int which);

/* ---- KMeans refinement restating the patched faiss 1.10 path, AVX2 flavour
 * (palette/refine.c:56-221 -> faiss/Clustering.cpp:70-120,135-263,267-554,587-603,
 *  utils/distances_fused/simdlib_based.cpp:27-277, utils/random.cpp:35-51,184-194).
 * colors planar f64 (n,3); centers_io planar f64 (k,3), initial centres in / refined out. */
void orc_kmeans_refine(const double *colors, const double *weights, size_t n,
                       double *centers_io, size_t k, int niter, size_t max_samples);
/* pieces, for stage-level parity tests (all f32, interleaved xyz like faiss) */
void orc_kmeans_subsample_indices(size_t n, size_t take, int64_t seed, int32_t *out);   /* random.cpp:184-194 prefix */
void orc_kmeans_assign(const float *x, size_t nx, const float *cent, size_t k, int64_t *assign, float *dis);
void orc_kmeans_update(const float *x, const float *w, size_t nx, const int64_t *assign,
                       float *cent, size_t k, float *hassign);
int  orc_kmeans_split_clusters(size_t k, size_t n, float *hassign, float *cent);

/* ---- exact f64 nearest-palette map (palette/nearest.c:150-209; FLANN eps=0 semantics,
 * ties -> lowest index).  colors planar (n,3), palette planar (k,3). */
void orc_nn_map(const double *colors, size_t n, const double *palette, size_t k, size_t *map);

/* ---- Riemersma dither (dither/riemersma.c:437-459).  colors/palette planar in linear
 * Rec2020.  Pixels never visited (1x1 image) leave map untouched. */
void orc_dither_riemersma(const double *colors, size_t width, size_t height,
                          const double *palette, size_t k, size_t *map);
void orc_dither_riemersma_prefix(const double *colors, size_t width, size_t height,
                                 const double *palette, size_t k, size_t *map, size_t max_visits);
/* Hilbert visiting order used by the dither: writes the in-bounds (y*width+x) sequence,
 * returns its length (== width*height unless the image is 1x1). */
size_t orc_hilbert_order(size_t width, size_t height, uint64_t *order);

/* ---- minimum-barrier-distance scans of the saliency map (src/patolette/patolette.pyx:54-201);
 * img / D: f32 row-major (rows, cols).  The numpy part of get_weights is oracle/saliency.py. */
int orc_mbd(size_t rows, size_t cols, const float *img, int iters, float *D);

/* ---- the whole thing: same signature and semantics as patolette() (lib/src/patolette.c:157-343) */
void orc_patolette(size_t width, size_t height, const double *data, const double *weights,
                   size_t palette_size, const orc_Options *options, double *palette,
                   size_t *palette_map, int *exit_code);
const char *orc_exit_message(int exit_code);
/* patolette.c:246-336 alone: KMeans, mapping / dithering, write-out behind GIVEN cluster centres (planar (len,3), quantisation space) */
void orc_patolette_from_centers(size_t width, size_t height, const double *data, const double *weights, size_t palette_size,
                                const orc_Options *options, const double *centers, size_t len, double *palette, size_t *palette_map);

/* per-stage wall seconds of the last orc_patolette call: convert, gq, lq, kmeans, map/dither, total */
void orc_last_timings(double out[6]);
/* threads for the loops the reference's dependencies run on several cores (faiss search / compute_centroids, FLANN's
 * nearest-neighbour search); results are independent of it.  Default 1. */
void orc_set_threads(int n);
int orc_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
