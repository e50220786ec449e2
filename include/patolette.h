/*
 * patolette.h -- drop-in C ABI of the MI355X-native patolette hot path.
 *
 * These three entry points and two types are exactly what the reference exports from
 * lib/include/patolette.h:7-35 and what its only FFI consumer binds
 * (src/patolette/patolette.pyx:14-40 `cdef extern from 'patolette.h'`, call site :429-439).
 * Same names, argument meaning, buffer layouts, ownership and exit codes; the work behind
 * them runs as HIP kernels on gfx950 (libpatolette_amd.so).  There is no CPU fallback: if no
 * HIP device is usable the call fails loudly (exit code -1 and a message on stderr).
 */
#ifndef PATOLETTE_H
#define PATOLETTE_H

#include <stdbool.h>
#include <stddef.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

/* reference: lib/include/patolette.h:7-11 */
typedef enum patolette__ColorSpace {
    patolette__sRGB,
    patolette__CIELuv,
    patolette__ICtCp
} patolette__ColorSpace;

/* reference: lib/include/patolette.h:13-20 (x86-64 SysV: offsets 0,1,4,8,16,24; sizeof 32) */
typedef struct patolette__QuantizationOptions {
    bool dither;
    bool palette_only;
    patolette__ColorSpace color_space;
    int kmeans_niter;
    size_t kmeans_max_samples;
    bool verbose;
} patolette__QuantizationOptions;

/*
 * reference: lib/include/patolette.h:22-32, implemented at lib/src/patolette.c:157-343.
 *   data        (width*height, 3) f64, column-major (planar R|G|B), row-scan pixel order, sRGB[0,1]; host memory
 *   weights     width*height f64 >= 1, or NULL; host memory
 *   palette     out, (palette_size, 3) f64 column-major sRGB[0,1]; unused rows = -1
 *   palette_map out, width*height size_t; may be NULL only when options->palette_only
 *   exit_code   0 ok, -1 internal quantisation error, -2 w*h == 0, -3 palette_size < 1, -4 w*h > 40000^2
 * Inputs are never written; on error the outputs are left untouched.
 */
void patolette(
    size_t width,
    size_t height,
    const double *data,
    const double *weights,
    size_t palette_size,
    const patolette__QuantizationOptions *options,
    double *palette,
    size_t *palette_map,
    int *exit_code
);

/* reference: lib/include/patolette.h:34, lib/src/patolette.c:97-105 */
const char *get_patolette_exit_code_info_message(int exit_code);

/* reference: lib/include/patolette.h:35, lib/src/patolette.c:107-119 (malloc'd; caller frees) */
patolette__QuantizationOptions *patolette_create_default_options(void);

#ifdef __cplusplus
}
#endif
#endif
