/*
 * patolette_amd.h -- additive C-ABI extensions of libpatolette_amd.so (not in the reference).
 *
 * The reference exports one monolithic call (include/patolette.h).  The entry points below
 * expose (1) the same call with inputs already resident in HBM, (2) the call with the Python
 * binding's `tile_size` (saliency-derived weights computed on the device) and with row-major
 * colours, (3) 8-bit adaptors around it, (4) a batch call for independent images, (5) each
 * stage of the path on its own -- every one names the reference function it replaces -- so
 * parity tests can check the HIP path stage by stage against the oracle, and (6) kernel
 * timing / run statistics for bench.py.  Plain pointers and sizes only.
 */
#ifndef PATOLETTE_AMD_H
#define PATOLETTE_AMD_H

#include <stddef.h>
#include <stdint.h>

#include "patolette.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- device management ------------------------------------------------------------------ */
int  patolette_amd_device_count(void);                 /* 0 if no usable HIP device */
int  patolette_amd_set_device(int ordinal);            /* 0 ok */
const char *patolette_amd_last_error(void);            /* message of the last failure on this thread */
void *patolette_amd_malloc(size_t bytes);              /* hipMalloc; NULL on failure */
void patolette_amd_free(void *dptr);
int  patolette_amd_memcpy_h2d(void *dst, const void *src, size_t bytes);
int  patolette_amd_memcpy_d2h(void *dst, const void *src, size_t bytes);
int  patolette_amd_synchronize(void);
/* Every calling thread works on an engine (HIP stream + workspace, ~170 bytes per pixel of the largest image it has
 * seen) taken from a per-device pool and handed back when the thread exits.  This call frees the calling thread's
 * engine and every idle pooled one (the reference frees all internals before patolette() returns,
 * lib/src/patolette.c:338-341; here the workspace is kept between calls for speed and released on request). */
void patolette_amd_release_workspace(void);
/* synthetic inputs generated directly in HBM: planar image of n pixels,
 * plane p pixel i = u(1000*seed + p, i); weights = 1 + 3*u(1000*seed + 7, i) */
int  patolette_amd_fill_image(double *d_planar, size_t n, uint64_t seed);
int  patolette_amd_fill_weights(double *d_weights, size_t n, uint64_t seed);

/* ---- the full path with HBM-resident inputs/outputs --------------------------------------
 * Same semantics as patolette() (lib/src/patolette.c:157-343) except that d_data / d_weights
 * are device pointers and the index map is left in HBM at d_palette_map with elements of
 * map_elem_bytes (1 when palette_size <= 256, else 4; 8 always allowed).  `palette` is host
 * memory, (palette_size,3) column-major. */
void patolette_amd_device(size_t width, size_t height, const double *d_data, const double *d_weights,
                          size_t palette_size, const patolette__QuantizationOptions *options,
                          double *palette, void *d_palette_map, int map_elem_bytes, int *exit_code);

/* ---- saliency-derived weights ----------------------------------------------
 * What the reference's Python binding computes before calling patolette() when tile_size > 0
 * (`get_weights`, src/patolette/patolette.pyx:203-313, called at :407-414): minimum-barrier-distance
 * saliency of the channel-mean image (3 raster scans), Mahalanobis colour contrast against the four
 * border bands in CIELAB, centre prior, sigmoid; weight = 1 + sal^2 * width*height / tile_size^2.
 *
 * patolette_amd_quantize = patolette() with the binding's tile_size semantics: `weights` non-NULL ->
 * used as given; else tile_size > 0 -> weights derived on the device (never leave HBM); else
 * unweighted.  Extra exit codes (messages via get_patolette_exit_code_info_message): -5 the image
 * shape cannot be processed (the reference raises for it too: a side <= 3 pixels, fewer than 100
 * pixels, or a border band of floor(0.1*sqrt(w*h)) pixels that does not fit), -6 a border band has
 * a singular colour covariance (numpy raises LinAlgError in the reference). */
void patolette_amd_quantize(size_t width, size_t height, const double *data, const double *weights,
                            double tile_size, size_t palette_size,
                            const patolette__QuantizationOptions *options, double *palette,
                            size_t *palette_map, int *exit_code);
/* The same with the colours as (width*height, 3) ROW-MAJOR f64 -- the layout of `img.reshape(-1, 3)` in numpy
 * (README.md:156) -- so a binding need not transpose into the planar layout patolette() takes
 * (np.asfortranarray at patolette.pyx:388-391 costs ~60 ms on a 16 MP image, six times the device time). */
void patolette_amd_quantize_rows(size_t width, size_t height, const double *rows, const double *weights,
                                 double tile_size, size_t palette_size,
                                 const patolette__QuantizationOptions *options, double *palette,
                                 size_t *palette_map, int *exit_code);
/* the stage alone, host buffers: data as for patolette(); weights_out[width*height].  Returns 0,
 * -1 (HIP error), -2 (shape), -3 (singular covariance). */
int patolette_amd_saliency_weights(size_t width, size_t height, const double *data, double tile_size,
                                   double *weights_out);

/* mbd(img, iters) alone (patolette.pyx:156-201): img / out f32 row-major (rows, cols), host buffers.
 * Returns 0, -1 (HIP error) or -2 (rows <= 3 or cols <= 3: the reference returns None). */
int patolette_amd_mbd(size_t rows, size_t cols, const float *img, int iters, float *out);

/* ---- 8-bit adaptors around the path ---------------------------------------
 * Replace what every caller of the reference does by hand around quantize():
 *   ingest          colors = img.reshape(-1,3).astype(float64) / 255           (README.md:156-158)
 *   palette_u8      clip(palette * 255, 0, 255).astype(uint8)                  (README.md:178-181)
 *   quantized       palette_u8[palette_map]                                    (README.md:186-187)
 * pixels: width*height interleaved 8-bit sRGB, `channels` (3 or 4) bytes per pixel (a 4th byte is
 * ignored).  weights / tile_size: as for patolette_amd_quantize().  Outputs, each optional (NULL):
 * palette (palette_size,3) column-major f64 exactly as patolette() returns it; palette_u8
 * palette_size x 3 interleaved (unused rows 0); palette_map with elements of map_elem_bytes
 * (1, 2, 4 or 8; must be able to hold palette_size-1); quantized = width*height x 3 interleaved.
 * The conversion v/255.0 happens in the first kernel (3 B/px cross PCIe instead of 24), results
 * are identical to the f64 entry point fed with the by-hand conversion.  The *_device flavour takes
 * device pointers for pixels / weights / palette_map / quantized (map_elem_bytes 1 when
 * palette_size <= 256, else 4); palette and palette_u8 stay host memory. */
void patolette_amd_u8(size_t width, size_t height, const unsigned char *pixels, int channels, const double *weights,
                      double tile_size, size_t palette_size, const patolette__QuantizationOptions *options, double *palette,
                      unsigned char *palette_u8, void *palette_map, int map_elem_bytes, unsigned char *quantized,
                      int *exit_code);
void patolette_amd_u8_device(size_t width, size_t height, const unsigned char *d_pixels, int channels,
                             const double *d_weights, double tile_size, size_t palette_size,
                             const patolette__QuantizationOptions *options, double *palette,
                             unsigned char *palette_u8, void *d_palette_map, int map_elem_bytes,
                             unsigned char *d_quantized, int *exit_code);

/* ---- batch of independent images (SURVEY.md 8(b) "Batch extension") -----------------------
 * count images of identical width x height; data[i] / weights[i] (weights may be NULL or hold
 * NULL entries) / tile_size / palettes[i] / palette_maps[i] / exit_codes[i] as for
 * patolette_amd_quantize(); up to six images are in flight on the current GPU (fewer if the free HBM does not hold their workspaces).  Results are
 * identical to count separate calls. */
void patolette_amd_batch(size_t count, size_t width, size_t height, const double *const *data,
                         const double *const *weights, double tile_size, size_t palette_size,
                         const patolette__QuantizationOptions *options, double *const *palettes,
                         size_t *const *palette_maps, int *exit_codes);

/* the same with every image as (width*height, 3) row-major f64 (see patolette_amd_quantize_rows) */
void patolette_amd_batch_rows(size_t count, size_t width, size_t height, const double *const *rows,
                              const double *const *weights, double tile_size, size_t palette_size,
                              const patolette__QuantizationOptions *options, double *const *palettes,
                              size_t *const *palette_maps, int *exit_codes);

/* the 8-bit entry for a batch: pixels[i] / palettes[i] / palettes_u8[i] / palette_maps[i] / quantized[i] as for
 * patolette_amd_u8() (palettes_u8, palette_maps, quantized or single entries of them may be NULL); 3 bytes per pixel cross
 * PCIe instead of 24, so a host-fed batch is no longer bound by the upload */
void patolette_amd_batch_u8(size_t count, size_t width, size_t height, const unsigned char *const *pixels, int channels,
                            const double *const *weights, double tile_size, size_t palette_size,
                            const patolette__QuantizationOptions *options, double *const *palettes,
                            unsigned char *const *palettes_u8, void *const *palette_maps, int map_elem_bytes,
                            unsigned char *const *quantized, int *exit_codes);

/* Host images in, index maps LEFT IN HBM: the entry behind patolette_amd.dist: each rank quantises its
 * shard of the batch and the maps go to rank 0 over RCCL straight from device memory -- no widening to size_t, no PCIe
 * round trip.  images[i]: pixel_format 0 = planar f64 as patolette() takes it, 1 = (N,3) row-major f64, 3 / 4 = interleaved
 * 8-bit sRGB with that many bytes per pixel.  d_palette_maps (device, may be NULL): count consecutive maps of
 * width*height elements of map_elem_bytes (1 when palette_size <= 256, else 4).  Everything else as patolette_amd_batch(). */
void patolette_amd_batch_dmap(size_t count, size_t width, size_t height, const void *const *images, int pixel_format,
                              const double *const *weights, double tile_size, size_t palette_size,
                              const patolette__QuantizationOptions *options, double *const *palettes, void *d_palette_maps,
                              int map_elem_bytes, int *exit_codes);

/* ---- ONE image over several GPUs (SURVEY.md 8(f)-4, 8(e) last row) ------------------------
 * The reference has no distributed code; this is the path's own sharding.  Every GPU of a group holds a contiguous slice
 * of the image's pixels (image order: slice r+1 follows slice r) and runs the whole path on it; the only data that crosses
 * GPUs are the per-node reductions of patolette.c's stages -- colour bounds and column sums (matrix2D.c:229), projection
 * extrema (sort.c:43-59), 512-bucket moment tables (cells.c:82-112, local.c:118-134), the children's centred moments
 * (cluster.c:111-152) -- and the KMeans subsample (refine.c:127-163, 262 144 x 12 bytes by default).  Those reductions are
 * integers, ordered-key minima / maxima, or sums of addends on fixed grids (DESIGN.md 4.1 "order-independent sums"): exact in
 * any grouping, so every GPU takes the same decisions and the result does not depend on how the pixels are dealt out --
 * the palette is identical on every GPU and bit-identical to what one GPU computes for the whole image with
 * patolette_amd_set_invariant_sums(1); each GPU's index map is its slice of that image's map.
 *
 * The library does not link a communication library: the caller lends it ONE collective, an in-place element-wise SUM over
 * the group (RCCL all-reduce through torch.distributed in patolette_amd.dist; any MPI / RCCL binding will do).  dtype: 0 =
 * f64, 1 = i64, 2 = i32.  `buffer` is device memory unless host_buffers is set (then the library stages through pinned
 * host memory -- for CPU-side transports such as gloo).  The call returns when the reduced values are in `buffer`; non-zero =
 * failure (exit code -1).  Minima, maxima and exclusive prefixes over the ranks are taken from a SUM over a table with one
 * row per rank.  Every rank must make the same call (same palette_size, options, total_pixels). */
typedef int (*patolette_amd_allreduce_sum_fn)(void *ctx, void *buffer, size_t count, int dtype);
typedef struct patolette_amd__Comm {
    int rank, size;
    patolette_amd_allreduce_sum_fn allreduce_sum;
    void *ctx;
    int host_buffers;
} patolette_amd__Comm;
/* slice_data: planar f64 (x | y | z, plane stride slice_pixels) of pixels [slice_begin, slice_begin + slice_pixels) of an
 * image of total_pixels pixels, host memory; slice_weights: NULL or slice_pixels doubles (all ranks alike); palette:
 * palette_size x 3 column-major, the same on every rank; slice_map: slice_pixels entries (NULL with palette_only).
 * Not available per slice: dithering (one curve over the whole image: exit code -1) and derived saliency weights
 * (pass explicit weights).  Exit codes as patolette(); every slice must hold at least one pixel. */
void patolette_amd_slice(size_t total_pixels, size_t slice_begin, size_t slice_pixels, const double *slice_data,
                         const double *slice_weights, size_t palette_size, const patolette__QuantizationOptions *options,
                         const patolette_amd__Comm *comm, double *palette, size_t *slice_map, int *exit_code);
/* the same with the slice (and its weights) resident in HBM and the slice's index map left there (elements of
 * map_elem_bytes: 1 when palette_size <= 256, else 4), as patolette_amd_device() */
void patolette_amd_slice_device(size_t total_pixels, size_t slice_begin, size_t slice_pixels, const double *d_slice_data,
                                const double *d_slice_weights, size_t palette_size, const patolette__QuantizationOptions *options,
                                const patolette_amd__Comm *comm, double *palette, void *d_slice_map, int map_elem_bytes,
                                int *exit_code);
/* 1: the children's moments of every split are summed product by product on the exact grids (tiling-invariant, what the
 * sliced path always does; 190 instead of 157 us per 4096^2 level in the partition kernel, +7 % on the whole step); 0 (default): per-thread partial sums first.  Applies to
 * the calling thread's later calls.  Returns the previous setting. */
int patolette_amd_set_invariant_sums(int on);
/* The centroid update of the KMeans refinement.  0 (default): the reference's, bit for bit -- every centroid is the sequential f32
 * sum of its samples in sample order (faiss Clustering.cpp:135-204), which costs a stable sort of the samples and one dependent
 * chain per centroid in every iteration.  1: order-free -- the sums are taken exactly (64-bit fixed point) and rounded to f32
 * once: deterministic, independent of how the samples spread over the centroids, 4.4x faster per KMeans stage at 67 M samples,
 * but NOT the reference's bits: per iteration the centroids differ by what its f32 chains round away (~1e-6 of the colour range),
 * and a sample on the border of two cells that therefore changes sides moves a centroid of m members by |x - c| / m (1e-4 at the
 * default 1024 members, 1e-6 at 65 536), which over the 32 iterations of the default spreads to most rows: the palette then agrees
 * with the reference's to ~1e-4 of the colour range, not to BASELINE's 1e-5, and ~0.02 % of the index map follows
 * (tests/test_gpu_kmeans_update.py prints the measured figures).  Process-wide; applies to later calls.  Returns
 * the previous setting.  Environment default: PAMD_KMEANS_UPDATE=1. */
int patolette_amd_set_kmeans_update(int mode);

/* Who drives the split loop of the local quantiser (quantize/local.c:318-404).  The device-driven loop enqueues round after round
 * without a host turn: the children's moments, bounds and eigen-solves and the next round's node list are made by two small
 * kernels, the sweep kernels read their sizes from device memory, which candidate nodes to evaluate is decided by a rule that
 * provably never skips a node the greedy loop commits, and the greedy loop of local.c:347-390 is replayed once, on the host, over
 * the evaluated tree; the host synchronises once per image instead of once per round.  Taken for palettes of up to 256 colours
 * on one GPU.  mode 2 (default): for images below 12 Mpixel, where the nine host turns are a third of the call; 1: wherever it
 * applies; 0: the host-driven loop everywhere (what sliced images, larger palettes and verbose calls always take).  Same
 * decisions and results either way (the children's moments may differ in their last bit: DESIGN.md 4.2).  Process-wide; returns
 * the previous setting.  Environment: PAMD_LQ_DEVICE=0|1|2. */
int patolette_amd_set_split_loop(int mode);
/* Where the decisions of the global quantiser are taken (lib/src/quantize/global.c:189-298: prefix sums of the 512-bucket table,
 * the bias termination test, the dynamic programme's backtrack, the bucket -> base cluster table, the base clusters' records):
 * 1 (default) on the device, in one kernel between the histogram and the partition (no host turn; palettes of more than 12
 * colours on one GPU, not verbose), 0 on the host.  Same arithmetic either way: same cuts, same clusters.  Returns the
 * previous setting. */
int patolette_amd_set_global_quantiser(int on_device);

/* The KMeans subsample list (faiss rand_perm(N, seed 1234), Clustering.cpp:311-319: a pure function of the pixel count) is made
 * on a helper thread that starts at call entry and is joined when the KMeans stage begins.  1 (default): the list stays on the
 * device between calls on images of one size; 0: every call makes it again -- the cost of a FIRST call of a size, call after
 * call (bench.py reports both).  The same switch governs the other size-only table kept between calls: the Riemersma dither's
 * curve order (rank along the Hilbert curve -> pixel number, 4 bytes per pixel, a function of width and height alone).
 * Process-wide.  Returns the previous setting. */
int patolette_amd_set_subsample_cache(int on);

/* ---- single stages, host buffers in / out (for parity tests) ------------------------------ */
/* The KMeans subsample the refinement draws (faiss Clustering.cpp:311-319: the first `take` entries of rand_perm(n, seed 1234),
 * utils/random.cpp:184-194) as the product's host code makes it (host only: needs no device).  0 = ok. */
int patolette_amd_subsample_indices(size_t n, size_t take, int32_t *out);
/* patolette__EIGEN_solve (math/eigen.c:83-140: LAPACK dsyev 'V','L', n = 3) as the split loop's host side solves it:
 * a column-major 3x3 (lower triangle read) -> w ascending, z = eigenvectors as columns; returns LAPACK's info (0 = ok).
 * patolette_amd_principal_axis: covariance as (xx,xy,xz,yy,yz,zz) -> eigenvector of the largest eigenvalue incl. its
 * sign (math/pca.c:122-149 takes column 2); 0 ok.  Pure host code (no device needed). */
int patolette_amd_eigen_sym3(const double a_colmajor[9], double w[3], double z[9]);
int patolette_amd_principal_axis(const double cov6[6], double axis[3]);
/* the same solver as the DEVICE runs it inside the split loop's control kernel (one problem per lane): `count` column-major 3x3
 * matrices in, w (3 per problem), z (9 per problem) and LAPACK's info out; host buffers.  Returns 0, or -1 on a HIP error. */
int patolette_amd_eigen_sym3_device(const double *a_colmajor, size_t count, double *w, double *z, int *info);
/* out[i] = pow(x[i], y) as the colour conversions evaluate it on the device (x >= 0; <= 0.51 ulp) */
int patolette_amd_pow(const double *x, double y, double *out, size_t n);

enum {
    PAMD_SRGB_TO_ICTCP = 0,     /* patolette__COLOR_sRGB_Matrix_to_ICtCp_Matrix            color/ICtCp.c:120-146 */
    PAMD_SRGB_TO_CIELUV = 1,    /* patolette__COLOR_sRGB_Matrix_to_CIELuv_Matrix           color/CIELuv.c:166-197 */
    PAMD_ICTCP_TO_REC2020 = 2,  /* patolette__COLOR_ICtCp_Matrix_to_Linear_Rec2020_Matrix  color/rec2020.c:128-148 */
    PAMD_CIELUV_TO_REC2020 = 3, /* patolette__COLOR_CIELuv_Matrix_to_Linear_Rec2020_Matrix color/rec2020.c:150-173 */
    PAMD_SRGB_TO_REC2020 = 4,   /* patolette__COLOR_sRGB_Matrix_to_Linear_Rec2020_Matrix   color/rec2020.c:175-195 */
    PAMD_REC2020_TO_SRGB = 5,   /* patolette__COLOR_Linear_Rec2020_Matrix_to_sRGB_Matrix   color/sRGB.c:112-132 */
    PAMD_CIELUV_TO_ICTCP = 6    /* the Luv->Rec2020->sRGB->ICtCp chain of patolette.c:305-314, fused per pixel */
};
/* in place on a planar (n,3) host matrix */
int patolette_amd_convert(int which, double *planar, size_t n);

/* patolette__GQ_quantize + patolette__LQ_quantize + patolette__PALETTE_create
 * (quantize/global.c:388-443, quantize/local.c:318-404, palette/create.c:11-33).
 * colors planar (n,3) already in the quantisation space; centers planar (K,3) out (first
 * *n_clusters rows valid, palette order).  Returns 0 or -1. */
int patolette_amd_quantize_clusters(const double *colors, const double *weights, size_t n, size_t K,
                                    double *centers, size_t *n_clusters);

/* patolette__PALETTE_get_refined_palette (palette/refine.c:165-221 -> patched faiss
 * kmeans_clustering).  centers_io planar (k,3) f64. */
int patolette_amd_kmeans_refine(const double *colors, const double *weights, size_t n,
                                double *centers_io, size_t k, int niter, size_t max_samples);

/* patolette__PALETTE_fill_palette_map_nearest (palette/nearest.c:150-209) */
int patolette_amd_nn_map(const double *colors, size_t n, const double *palette, size_t k, size_t *map);

/* patolette__DITHER_riemersma (dither/riemersma.c:437-459); colors/palette in linear Rec2020 */
int patolette_amd_dither(const double *colors, size_t width, size_t height, const double *palette,
                         size_t k, size_t *map);

/* Test / tuning knob of the segment-parallel dither (process-wide): `segments` runs (0 = chosen from the image size, 1 = the
 * single serial chain) and `warm` in-image pixels of speculative warm-up per run (< 0 = default; 0 forces every boundary to be
 * repaired).  The map is the reference's chain bit for bit for every setting. */
void patolette_amd_dither_config(int segments, int warm);
/* ... and which layout walks the runs (process-wide, 8 <= palette rows <= 256): -1 (default) = one LANE per run for images of
 * 2^23 pixels and more, one WAVEFRONT per run below; 1 = lanes from 65 536 pixels on; 0 = wavefronts everywhere.  Same map. */
void patolette_amd_dither_layout(int lanes);
/* TESTS ONLY: after `cap` passes taken by one wavefront alone the lane layout gives the image up and the wavefront layout starts
 * over (default 4096, never reached in practice; 0 forces that fall-back at the first stall); < 0 restores the default.  Returns
 * the previous value. */
int patolette_amd_debug_dither_solo_cap(int cap);
/* TESTS ONLY: verification passes without progress (fewer than an eighth of the failing boundaries fixed) after which ONE wavefront
 * walks alone from the lowest failing boundary (default 2; 0 = every repair pass is such a walk); < 0 restores the default.
 * Returns the previous value.  The map is the reference's chain for every setting. */
int patolette_amd_debug_dither_stall_passes(int n);
/* which layout a dither of this image size and palette takes under the current knobs: 1 = one lane per run, 0 = one wavefront per run */
int patolette_amd_dither_layout_in_use(size_t width, size_t height, size_t palette_rows);
/* Where the dither cuts the curve (host-side copy of the kernel's function, runs without a GPU): *d = first curve position of the
 * aligned 64-position block that holds in-image pixel number t (curve order, 0-based), *c = in-image pixels before that block. */
void patolette_amd_debug_dither_locate(size_t width, size_t height, unsigned long long t, unsigned long long *d, unsigned long long *c);

/* ---- statistics of the last full-path call on this thread -------------------------------- */
typedef struct patolette_amd__Stats {
    double ms_total, ms_upload, ms_convert, ms_gq, ms_lq, ms_kmeans, ms_map, ms_download, ms_saliency;
    size_t n_base_clusters;   /* clusters produced by the global quantiser */
    size_t n_clusters;        /* final palette rows */
    size_t split_evals;       /* split_cluster evaluations performed on the GPU */
    size_t split_px;          /* sum of their sizes: D_eff = split_px / (width*height) */
    size_t lq_rounds;         /* host<->device round trips of the split loop */
    size_t kmeans_samples;    /* samples clustered per KMeans iteration */
    size_t dither_segments;   /* runs the Hilbert curve was cut into (walked side by side, one lane or one wavefront each) */
    size_t dither_repairs;    /* runs walked again because their speculative starting state was not the chain's */
    size_t dither_rounds;     /* boundary-verification passes (the last one found nothing to repair) */
    size_t dither_through;    /* stalled verifications (a long flat stretch off the palette) resolved by walking one run through its successors */
    size_t dither_jumps;      /* lane layout: periodic jumps -- a walk over pixels of ONE colour found its period and wrote the pattern to the stretch's end */
    size_t dither_solo;       /* lane layout: passes taken by one wavefront alone after two passes without progress */
} patolette_amd__Stats;
void patolette_amd_last_stats(patolette_amd__Stats *out);
/* The palette exactly as the mapping stage of the last full-path call on this thread used it: linear Rec2020 when dithering
 * (patolette.c:268-299), ICtCp for the nearest-neighbour map (:300-324), before the conversion back to sRGB.  Written
 * column-major with `capacity_rows` rows when they suffice; returns the number of palette rows.  Lets a parity test feed
 * the oracle's dither / NN map the very inputs the device stage saw. */
size_t patolette_amd_last_map_palette(double *out, size_t capacity_rows);

/* ---- what the quantisers of the last call on this thread decided (full path or patolette_amd_quantize_clusters) -----------
 * The global quantiser's axis, covariance and cuts (quantize/global.c:388-443) and, per committed split of the greedy loop
 * (quantize/local.c:347-390) in commit order: which row of `result` was split (`best`, :348-352), where (the bucket of
 * get_optimal_bucket_index, :102-177), along which axis (cluster.c:191-217, with the very matrix the eigen-solver was handed),
 * into how many members, and the distortions behind the benefit (:256-275).  On content whose decisions are ties in exact
 * arithmetic (perfect gradients, a handful of colours) the reference's own choice hinges on the rounding of its sequential
 * sums; tests/tie_prover.py takes this trace and the pixels and checks, in exact integer arithmetic, that every decision
 * recorded here lies within that rounding envelope of the exact optimum.  The CPU oracle exports the same records. */
typedef struct patolette_amd__SplitRecord {
    int32_t row, new_row;     /* row split; row that received the LEFT child (= clusters before the commit, local.c:375-376) */
    int32_t split;            /* bucket index of the cut */
    int32_t degenerate;       /* the projection took sort.c:61-79's round-robin rule */
    uint64_t n, n_left, n_right;
    double sw;                /* sum of the cluster's weights */
    double axis[3];
    double cov6[6];           /* xx, yx, zx, yy, zy, zz as handed to the eigen-solver (pca.c:62-101) */
    double dist, dist_left, dist_right, benefit;
} patolette_amd__SplitRecord;
typedef struct patolette_amd__SplitTrace {
    int32_t n_base, n_clusters, n_records, stopped_early;   /* stopped_early: best benefit < 1e-16 (local.c:365-370) */
    double gq_axis[3];
    uint64_t gq_cuts[14];     /* n_base + 1 entries (global.c:290) */
    double gq_cov6[6];        /* the unweighted covariance of all pixels as handed to the eigen-solver (global.c:407) */
} patolette_amd__SplitTrace;
/* returns the number of records; copies min(capacity, that) of them */
size_t patolette_amd_last_split_trace(patolette_amd__SplitTrace *hdr, patolette_amd__SplitRecord *recs, size_t capacity);
/* PALETTE_create's rows of the last call (palette/create.c:11-33: the clusters' weighted centres in the quantisation space,
 * before any KMeans), column-major with `capacity_rows` rows when they suffice; returns the number of rows */
size_t patolette_amd_last_cluster_centers(double *out, size_t capacity_rows);

/* TESTS ONLY: make the quantisers take a deliberately WRONG decision, to show that tests/tie_prover.py tells a tie from a bug.
 * 0 = none (default); 1 = the cut one occupied bucket past the arg-max of local.c:171; 2 = the greedy step takes the second
 * best cluster (local.c:277-307); 3 = the cut at the LAST maximum of the objective instead of the first (a member of the tie
 * set: differs from the reference only where the objective is exactly tied).  Process-wide; returns the previous setting. */
int patolette_amd_debug_fault(int which);

/* ---- per-kernel timing with HIP events on the launch stream ------------------------------ */
void patolette_amd_profile_enable(int on);   /* also resets the accumulated numbers */
/* restrict the timing to the kernel of this name (NULL or "" = every kernel): two event records per launch
 * cost ~7 us, so a throughput measurement times only the kernel it reports the roofline of */
void patolette_amd_profile_only(const char *kernel_name);
/* with a kernel selected by patolette_amd_profile_only: put events on every `period`-th of its launches only (1 = all) */
void patolette_amd_profile_sample(int period);
/* number of distinct kernels seen; entry i: name (<= 63 chars), accumulated ms, launch count and the
 * ALGORITHMIC HBM bytes of those launches (per-unit figures in DESIGN.md) */
int  patolette_amd_profile_count(void);
int  patolette_amd_profile_get(int i, char *name64, double *total_ms, size_t *launches, double *total_bytes);

#ifdef __cplusplus
}
#endif
#endif
