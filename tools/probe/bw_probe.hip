// Practical HBM ceilings on this part for the access shapes the quantiser kernels use (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/bw_probe tools/probe/bw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int U, typename T>
__global__ __launch_bounds__(256) void k_read(const T *__restrict__ a, size_t n, double *out) {
    double acc = 0;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n; i += stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) { if constexpr (sizeof(T) == 8) acc += v[u]; else acc += v[u].x + v[u].y; }
    }
    if (acc == 1.2345e300) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void k_read3(const double *__restrict__ a, size_t N, double *out) {   // three planes
    double acc = 0;
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < N; i += stride) {
        double x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; u++) { x[u] = a[i + u * 256]; y[u] = a[N + i + u * 256]; z[u] = a[2 * N + i + u * 256]; }
#pragma unroll
        for (int u = 0; u < U; u++) acc += x[u] * y[u] + z[u];
    }
    if (acc == 1.2345e300) out[0] = acc;
}
template <int U, int CH>
__global__ __launch_bounds__(256) void k_read3_chunk(const double *__restrict__ a, size_t N, double *out) {   // a contiguous chunk per block
    double mn = 1e300, mx = -1e300;
    const size_t base = (size_t)blockIdx.x * CH;
    for (unsigned i = threadIdx.x; i + (U - 1) * 256 < CH; i += 256 * U) {
        double x[U], y[U], z[U];
#pragma unroll
        for (int u = 0; u < U; u++) { x[u] = a[base + i + u * 256]; y[u] = a[N + base + i + u * 256]; z[u] = a[2 * N + base + i + u * 256]; }
#pragma unroll
        for (int u = 0; u < U; u++) { double d = x[u] * 0.3 + y[u] * 0.5 + z[u] * 0.7; mn = fmin(mn, d); mx = fmax(mx, d); }
    }
    if (mn == 1.2345e300 || mx == 1.2345e300) out[0] = mn;
}
struct PTile { unsigned long long start; unsigned int count, node; };
struct PNode { double axis[3]; double pad[20]; unsigned long long minkey[16], maxkey[16]; int buf; };
__device__ __forceinline__ unsigned long long pkey(double v) { unsigned long long b = __double_as_longlong(v); return (b >> 63) ? ~b : (b | 0x8000000000000000ull); }
template <int MODE>
__global__ __launch_bounds__(256) void k_mm_real(const double *__restrict__ a, size_t N, const PTile *__restrict__ tiles, PNode *nodes) {
    const PTile t = tiles[blockIdx.x];
    PNode &nd = nodes[t.node];
    const double a0 = nd.axis[0], a1 = nd.axis[1], a2 = nd.axis[2];
    const double *px = a + (size_t)nd.buf * N, *py = px + N, *pz = py + N;
    double mn = INFINITY, mx = -INFINITY;
    unsigned i = threadIdx.x;
    for (; i + 3 * 256 < t.count; i += 4 * 256) {
        const size_t p = t.start + i;
        double x[4], y[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { x[u] = px[p + u * 256]; y[u] = py[p + u * 256]; z[u] = pz[p + u * 256]; }
#pragma unroll
        for (int u = 0; u < 4; u++) { double d = x[u] * a0 + y[u] * a1 + z[u] * a2; mn = fmin(mn, d); mx = fmax(mx, d); }
    }
    for (; i < t.count; i += 256) { size_t p = t.start + i; double d = px[p] * a0 + py[p] * a1 + pz[p] * a2; mn = fmin(mn, d); mx = fmax(mx, d); }
    if (MODE == 0) { if (mn == 1.2345e300) nd.pad[0] = mx; return; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { mn = fmin(mn, __shfl_down(mn, o, 64)); mx = fmax(mx, __shfl_down(mx, o, 64)); }
    __shared__ double smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { mn = fmin(mn, smn[w]); mx = fmax(mx, smx[w]); }
        if (mn <= mx) { atomicMin(&nd.minkey[blockIdx.x & 15], pkey(mn)); atomicMax(&nd.maxkey[blockIdx.x & 15], pkey(mx)); }
    }
}
template <int U>
__global__ __launch_bounds__(256) void k_copy(const double2 *__restrict__ a, double2 *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + (U - 1) * 256 < n; i += stride) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = a[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; u++) b[i + u * 256] = v[u];
    }
}

__global__ void k_fill_rand(double *a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        a[i] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
}
int main(int argc, char **argv) {
    const size_t N = 16777216;                 // pixels; three planes of f64 = 403 MB
    double *a, *b, *out;
    CK(hipMalloc(&a, 4 * N * 8)); CK(hipMalloc(&b, 4 * N * 8)); CK(hipMalloc(&out, 8));
    CK(hipMemset(a, 1, 4 * N * 8)); CK(hipMemset(b, 0, 4 * N * 8));
    if (argc > 1) { hipLaunchKernelGGL(k_fill_rand, 4096, 256, 0, 0, a, 4 * N); hipLaunchKernelGGL(k_fill_rand, 4096, 256, 0, 0, b, 4 * N); printf("random data\n"); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, double bytes, auto launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.1f us  %7.0f GB/s\n", name, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e9);
    };
    for (int blocks : {2048, 8192, 65536}) {
        printf("-- blocks %d\n", blocks);
        timeit("read f64 x1 (1 stream, 403MB)", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read<1, double>), blocks, 256, 0, 0, a, 3 * N, out); });
        timeit("read f64 x4", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read<4, double>), blocks, 256, 0, 0, a, 3 * N, out); });
        timeit("read f64x2 x1", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read<1, double2>), blocks, 256, 0, 0, (const double2 *)a, 3 * N / 2, out); });
        timeit("read f64x2 x4", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read<4, double2>), blocks, 256, 0, 0, (const double2 *)a, 3 * N / 2, out); });
        timeit("read 3 planes x1", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read3<1>), blocks, 256, 0, 0, a, N, out); });
        timeit("read 3 planes x4", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read3<4>), blocks, 256, 0, 0, a, N, out); });
        timeit("copy f64x2 x1 (403MB r + 403MB w)", 6.0 * N * 8, [&] { hipLaunchKernelGGL((k_copy<1>), blocks, 256, 0, 0, (const double2 *)a, (double2 *)b, 3 * N / 2); });
        timeit("copy f64x2 x4", 6.0 * N * 8, [&] { hipLaunchKernelGGL((k_copy<4>), blocks, 256, 0, 0, (const double2 *)a, (double2 *)b, 3 * N / 2); });
    }
    timeit("chunk 8192/block x1", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read3_chunk<1, 8192>), N / 8192, 256, 0, 0, a, N, out); });
    timeit("chunk 8192/block x4", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read3_chunk<4, 8192>), N / 8192, 256, 0, 0, a, N, out); });
    timeit("chunk 4096/block x4", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read3_chunk<4, 4096>), N / 4096, 256, 0, 0, a, N, out); });
    timeit("chunk 2048/block x4", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read3_chunk<4, 2048>), N / 2048, 256, 0, 0, a, N, out); });
    timeit("chunk 2048/block x8", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read3_chunk<8, 2048>), N / 2048, 256, 0, 0, a, N, out); });
    timeit("chunk 16384/block x4", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_read3_chunk<4, 16384>), N / 16384, 256, 0, 0, a, N, out); });
    {
        std::vector<PTile> ht(N / 8192);
        for (size_t i = 0; i < ht.size(); i++) ht[i] = {i * 8192ull, 8192u, 0u};
        PNode hn = {}; hn.axis[0] = 0.3; hn.axis[1] = 0.5; hn.axis[2] = 0.7;
        for (int q = 0; q < 16; q++) { hn.minkey[q] = ~0ull; hn.maxkey[q] = 0; }
        PTile *dt; PNode *dn; CK(hipMalloc(&dt, ht.size() * sizeof(PTile))); CK(hipMalloc(&dn, sizeof(PNode)));
        CK(hipMemcpy(dt, ht.data(), ht.size() * sizeof(PTile), hipMemcpyHostToDevice)); CK(hipMemcpy(dn, &hn, sizeof(PNode), hipMemcpyHostToDevice));
        timeit("minmax replica, no reduce", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_mm_real<0>), N / 8192, 256, 0, 0, a, N, dt, dn); });
        timeit("minmax replica, full", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_mm_real<1>), N / 8192, 256, 0, 0, a, N, dt, dn); });
        // same, but the planes were just written by another kernel (as in the split loop)
        timeit("copy then minmax replica (both)", 9.0 * N * 8, [&] { hipLaunchKernelGGL((k_copy<4>), 8192, 256, 0, 0, (const double2 *)b, (double2 *)a, 3 * N / 2);
                                                                      hipLaunchKernelGGL((k_mm_real<1>), N / 8192, 256, 0, 0, a, N, dt, dn); });
    }
    {
        std::vector<PTile> ht(N / 8192);
        for (size_t i = 0; i < ht.size(); i++) ht[i] = {(ht.size() - 1 - i) * 8192ull, 8192u, 0u};
        PNode hn = {}; hn.axis[0] = 0.3; hn.axis[1] = 0.5; hn.axis[2] = 0.7;
        for (int q = 0; q < 16; q++) { hn.minkey[q] = ~0ull; hn.maxkey[q] = 0; }
        PTile *dt; PNode *dn; CK(hipMalloc(&dt, ht.size() * sizeof(PTile))); CK(hipMalloc(&dn, sizeof(PNode)));
        CK(hipMemcpy(dt, ht.data(), ht.size() * sizeof(PTile), hipMemcpyHostToDevice)); CK(hipMemcpy(dn, &hn, sizeof(PNode), hipMemcpyHostToDevice));
        timeit("minmax replica reversed tiles", 3.0 * N * 8, [&] { hipLaunchKernelGGL((k_mm_real<1>), N / 8192, 256, 0, 0, a, N, dt, dn); });
        timeit("copy then REVERSED minmax (both)", 9.0 * N * 8, [&] { hipLaunchKernelGGL((k_copy<4>), 8192, 256, 0, 0, (const double2 *)b, (double2 *)a, 3 * N / 2);
                                                                      hipLaunchKernelGGL((k_mm_real<1>), N / 8192, 256, 0, 0, a, N, dt, dn); });
        timeit("copy alone", 6.0 * N * 8, [&] { hipLaunchKernelGGL((k_copy<4>), 8192, 256, 0, 0, (const double2 *)b, (double2 *)a, 3 * N / 2); });
    }
    timeit("hipMemcpyDtoD 403MB", 6.0 * N * 8, [&] { hipMemcpyAsync(b, a, 3 * N * 8, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
