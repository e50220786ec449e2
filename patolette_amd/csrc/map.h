// map.h -- launchers of the palette-mapping kernels (map.hip).
#pragma once

#include "common.h"

namespace pamd {

// colours: planar, plane p at d_colors + p*plane_stride, n pixels mapped; palette planar (k,3);
// output elements of elem_bytes (1, 4 or 8)
struct NNWork {                        // scratch of the pruned NN map
    DevBuf<unsigned char> lut;         // per grid cell: 32 candidate slots (u8 or u16), count in the last
    DevBuf<unsigned long long> keys;   // min/max keys when the caller has no bounds
    DevBuf<unsigned char> clist;       // coarse pass of the LUT build: surviving entries per 4x4x4 block of cells
    DevBuf<unsigned int> mid;          // (G/2)^3 table of up to four candidates per cell, held in LDS by k_nn_map_mid
    DevBuf<double> dtab;               // dither with more than 3200 palette entries: its two palette tables (6 k doubles)
    DevBuf<unsigned int> dside;        // segment-parallel dither: [S][16] boundary choices + the repair counter
    PinBuf<unsigned int> hrep;         // ... the counter's landing place on the host
    DevBuf<double> dsort;              // lane-per-run dither: the pixels in curve order (3 planes)
    DevBuf<unsigned int> dpos;         // ... their linear pixel numbers (a function of width and height: kept between calls)
    size_t order_w = 0, order_h = 0; int order_dev = -1;
    DevBuf<unsigned char> dsmap;       // ... the choices in curve order, and behind them the [S][16] boundary records
    DevBuf<unsigned char> dflag;       // ... [S + 1] which boundaries the last check listed
    size_t dither_segments = 0, dither_repairs = 0, dither_rounds = 0;   // of the last launch_dither
    size_t dither_through = 0;         // ... times a stalled verification was resolved by walking one run through its successors
    size_t dither_jumps = 0;           // ... periodic jumps through flat stretches (lane layout's repair walks)
    size_t dither_solo = 0;            // ... passes taken by one wavefront alone (lane layout)
    hipStream_t side_stream = nullptr, side_stream2 = nullptr;   // lane-per-run dither: the record grids are built here while the pixels are gathered
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_join2 = nullptr;
    int side_dev = -1;
    NNWork() = default;
    NNWork(const NNWork &) = delete;
    NNWork &operator=(const NNWork &) = delete;
    ~NNWork() {
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_join2) (void)hipEventDestroy(ev_join2);
        if (side_stream) (void)hipStreamDestroy(side_stream);
        if (side_stream2) (void)hipStreamDestroy(side_stream2);
    }
};
// lo/hi: exact per-plane min/max of the colours if known (else nullptr: computed with one more pass)
void launch_nn_map(const double *d_colors, size_t plane_stride, size_t n, const double *d_pal, int k,
                   void *d_out, int elem_bytes, const double *lo, const double *hi, NNWork &w, hipStream_t s);
// which: the conversion the pixels still need on their way into linear Rec2020 (PAMD_COPY: none) -- only the lane-per-run layout
// (dither_lane_layout) converts on the fly; h_pal: the same palette on the host if the caller has it (else it is read back)
bool dither_lane_layout(size_t width, size_t height, int k);
void launch_dither(const double *d_img, size_t plane_stride, int which, size_t width, size_t height, const double *d_pal, const double *h_pal, int k,
                   void *d_out, int elem_bytes, NNWork &w, hipStream_t s, int layout = -1);   // layout: 1 lanes, 0 wavefronts, -1 decide here

// test / tuning knob: runs the curve is cut into (0 = chosen from the image size) and in-image pixels of warm-up (< 0 = default)
void dither_config(int segments, int warm);
// which layout walks the runs: 1 = one lane per run where it applies (default), 0 = one wavefront per run, -1 = default
void dither_layout(int lanes);
// tests: solo passes after which the lane layout gives up and the wavefront layout takes the image (default 4096, never reached);
// returns the previous value
int dither_solo_cap(int cap);
int dither_stall_passes(int n);
// keep the curve order of an image size on the device between calls (default) or make it again in every call
void dither_order_cache(bool on);

}  // namespace pamd
