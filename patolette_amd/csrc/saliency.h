// saliency.h -- saliency-derived pixel weights, the device-side replacement of
// `get_weights` in the reference's Python binding (src/patolette/patolette.pyx:203-313).
#pragma once

#include "color_device.h"
#include "common.h"

namespace pamd {

constexpr int kSalMax = 8;           // maxima: 0..3 border contrasts, 4 barrier distance, 5 u_final, 6 s1, 7 s2

struct SalDev {                      // device-resident scalars of one saliency evaluation
    double sum[kStatSlots][4][3][2]; // binned sums of Lab over the four border bands (slots: same-address atomics serialise)
    double cov[kStatSlots][4][6][2]; // binned centred products xx,xy,xz,yy,yz,zz
    double mean[4][3];
    double vi[4][9];                 // inverse covariance, row-major
    unsigned long long maxkey[kSalMax][kStatSlots];
    double mx[kSalMax];              // folded maxima (rounded through f32 where the reference holds a `cdef float`)
    int singular;                    // a border band has a singular covariance (numpy raises LinAlgError)
};

struct SalWork {
    DevBuf<float4> st;               // minimum-barrier state {img, D, U, L} per pixel, row-major
    DevBuf<float4> skew;             // the same, skewed by one column per row in groups of 64 rows: what the raster scans work on
    DevBuf<float> tmp;
    DevBuf<double> lab, s;           // CIELAB planes, running saliency map
    DevBuf<SalDev> dev;
    DevBuf<unsigned int> progress;   // per-strip progress flags of the raster scans (+ one stall flag)
    unsigned int *d_stall = nullptr;
    PinBuf<SalDev> host;
};

// error codes of the stage
constexpr int kSalOk = 0, kSalBadShape = -2, kSalSingular = -3;

// Returns kSalBadShape without touching the device when the reference's get_weights cannot process the
// shape.  On success d_weights (width*height f64, device) holds 1 + sal^2 * N / tile_size^2.
// d_f64 planar (channels >= 0) or (N,3) row-major (channels < 0); d_u8 interleaved with `channels` bytes per pixel
int saliency_weights(SalWork &w, const double *d_f64, const unsigned char *d_u8, int channels, size_t width, size_t height,
                     double tile_size, double *d_weights, hipStream_t s);

// mbd(img, iters) alone (patolette.pyx:156-201), host buffers, f32 row-major (rows, cols); for the parity tests
int mbd_device(SalWork &w, const float *h_img, size_t rows, size_t cols, int iters, float *h_out, hipStream_t s);

}  // namespace pamd
