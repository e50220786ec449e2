// common.h -- shared declarations of the MI355X-native patolette pipeline (product code).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/patolette.h"
#include "../../include/patolette_amd.h"

namespace pamd {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

#define HIP_CHECK(expr)                                                                         \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            char _buf[512];                                                                     \
            snprintf(_buf, sizeof _buf, "patolette_amd: HIP error %s at %s:%d (%s)",            \
                     hipGetErrorString(_e), __FILE__, __LINE__, #expr);                         \
            throw pamd::HipError(_buf);                                                         \
        }                                                                                       \
    } while (0)

constexpr int kBuckets = 512;        // reference: quantize/global.c:22, local.c:15
constexpr double kDelta = 1e-16;     // reference: math/misc.h:5

// ---- kernel timing (HIP events on the launch stream), enabled by patolette_amd_profile_enable ----
struct KernelTimer {
    struct Rec { hipEvent_t a, b; int id; double bytes; const double *src; };   // src: bytes = bytes-per-unit x *src, read at collect()
    bool enabled = false;
    std::string only;                // when not empty: time just the kernel of this name
    unsigned sample_period = 1;      // ... and of that kernel every sample_period-th launch (two event records cost ~12 us)
    unsigned long long only_seen = 0;
    std::vector<std::string> names;
    std::vector<double> total_ms, total_bytes;
    std::vector<size_t> launches;
    std::vector<Rec> pending;
    std::vector<hipEvent_t> pool;
    int id_of(const char *name);
    hipEvent_t get_event();
    void begin(int id, hipStream_t s, double bytes, const double *src = nullptr);
    void end(hipStream_t s);
    void collect();          // call after the stream is synchronised
    void reset();
};
KernelTimer &ktimer();

struct ScopedKernel {
    bool on;
    hipStream_t s;
    ScopedKernel(const char *name, hipStream_t stream, double bytes, const double *src = nullptr)
        : on(ktimer().enabled && (ktimer().only.empty() || ktimer().only == name)), s(stream) {
        if (on && !ktimer().only.empty() && ktimer().sample_period > 1) on = (ktimer().only_seen++ % ktimer().sample_period) == 0;
        if (on) ktimer().begin(ktimer().id_of(name), s, bytes, src);
    }
    ~ScopedKernel() { if (on) ktimer().end(s); }
};
#define PAMD_CAT2(a, b) a##b
#define PAMD_CAT(a, b) PAMD_CAT2(a, b)
// bytes = ALGORITHMIC HBM bytes of the launch (DESIGN.md lists the per-unit figures)
#define KTIME(name, stream, bytes) pamd::ScopedKernel PAMD_CAT(_ktime_scope_, __LINE__)(name, stream, (double)(bytes))
// a launch whose extent only the device knows: `per_unit` bytes x the double at `src` (filled in by the time collect() runs);
// src null = the plain form with `units` known on the host.  A launch that turned out empty (*src == 0) is not counted.
#define KTIME_DYN(name, stream, per_unit, units, src) \
    pamd::ScopedKernel PAMD_CAT(_ktime_scope_, __LINE__)(name, stream, (src) ? (double)(per_unit) : (double)(per_unit) * (double)(units), src)

// ---- simple device buffer with capacity reuse ----
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        HIP_CHECK(hipMalloc((void **)&p, n * sizeof(T)));
        cap = n;
    }
    // grow keeping the first `keep` elements
    void grow(size_t n, size_t keep) {
        if (n <= cap) return;
        T *q = nullptr;
        size_t ncap = n + n / 2;
        HIP_CHECK(hipMalloc((void **)&q, ncap * sizeof(T)));
        if (p && keep) HIP_CHECK(hipMemcpy(q, p, keep * sizeof(T), hipMemcpyDeviceToDevice));
        if (p) (void)hipFree(p);
        p = q; cap = ncap;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// pinned host staging buffer
template <typename T>
struct PinBuf {
    T *p = nullptr;
    size_t cap = 0;
    void reserve(size_t n) {
        if (n <= cap) return;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        HIP_CHECK(hipHostMalloc((void **)&p, n * sizeof(T), hipHostMallocDefault));
        cap = n;
    }
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
};

inline size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute applies to the current device: a once-flag per device (one process may drive several GPUs)
struct PerDeviceOnce {
    bool done[64] = {};
    bool first() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
};

inline int num_cus() {                                      // compute units of the current device (256 on MI355X)
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
}

}  // namespace pamd
