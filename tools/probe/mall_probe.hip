// mall_probe.hip -- does a re-read of X MB come from the 256 MiB Infinity Cache, and how fast?  (round-6 VERDICT item 3)
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/mall_probe tools/probe/mall_probe.hip
// Prints, per working-set size: GB/s of (a) repeated reads of the same X MB, (b) a read of X MB right after a kernel wrote it,
// (c) a read of X MB after 1 GB of other traffic (cold), (d) write X MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const double2 *__restrict__ p, size_t n, double *out) {
    double a = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        const double2 v0 = p[i], v1 = p[i + stride], v2 = p[i + 2 * stride], v3 = p[i + 3 * stride];
        a += v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
    }
    for (; i < n; i += stride) { const double2 v = p[i]; a += v.x + v.y; }
    if (a == 1.2345e300) out[0] = a;
}
// contiguous chunk per block (as the sweeps' tiles are), from the end if rev
__global__ __launch_bounds__(256) void k_read_tiles(const double2 *__restrict__ p, size_t n, double *out, int rev) {
    const size_t per = 4096;                                 // double2 per tile = 64 KB
    const size_t nt = (n + per - 1) / per;
    double a = 0;
    for (size_t t = blockIdx.x; t < nt; t += gridDim.x) {
        const size_t tt = rev ? nt - 1 - t : t;
        const size_t b = tt * per, e = b + per < n ? b + per : n;
        for (size_t i = b + threadIdx.x; i < e; i += 256) { const double2 v = p[i]; a += v.x + v.y; }
    }
    if (a == 1.2345e300) out[0] = a;
}
__global__ __launch_bounds__(256) void k_write(double2 *p, size_t n, double v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = make_double2(v, v);
}
int main() {
    const size_t big = (size_t)2048 << 20;
    double2 *buf, *other; double *out;
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&other, big)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, big)); CK(hipMemset(other, 0, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = 256 * 16;
    auto timed = [&](auto f) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return (double)ms; };
    auto trash = [&]() { hipLaunchKernelGGL(k_read, grid, 256, 0, 0, other, big / 16, out); };
    printf("%8s %12s %12s %12s %12s %12s %12s\n", "MB", "reread", "rd_after_wr", "rd_cold", "write", "rd_tiles_fwd", "rd_tiles_snake");
    for (size_t mb : {16, 32, 64, 96, 128, 160, 192, 256, 384, 512, 1024}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        double best[6] = {1e9, 1e9, 1e9, 1e9, 1e9, 1e9};
        for (int rep = 0; rep < 5; rep++) {
            trash(); hipLaunchKernelGGL(k_read, grid, 256, 0, 0, buf, n, out);
            best[0] = std::min(best[0], timed([&] { hipLaunchKernelGGL(k_read, grid, 256, 0, 0, buf, n, out); }));
            trash(); hipLaunchKernelGGL(k_write, grid, 256, 0, 0, buf, n, 1.0);
            best[1] = std::min(best[1], timed([&] { hipLaunchKernelGGL(k_read, grid, 256, 0, 0, buf, n, out); }));
            trash();
            best[2] = std::min(best[2], timed([&] { hipLaunchKernelGGL(k_read, grid, 256, 0, 0, buf, n, out); }));
            trash();
            best[3] = std::min(best[3], timed([&] { hipLaunchKernelGGL(k_write, grid, 256, 0, 0, buf, n, 2.0); }));
            trash(); hipLaunchKernelGGL(k_read_tiles, grid, 256, 0, 0, buf, n, out, 0);
            best[4] = std::min(best[4], timed([&] { hipLaunchKernelGGL(k_read_tiles, grid, 256, 0, 0, buf, n, out, 0); }));
            trash(); hipLaunchKernelGGL(k_read_tiles, grid, 256, 0, 0, buf, n, out, 0);
            best[5] = std::min(best[5], timed([&] { hipLaunchKernelGGL(k_read_tiles, grid, 256, 0, 0, buf, n, out, 1); }));
        }
        printf("%8zu", mb);
        for (int q = 0; q < 6; q++) printf(" %12.0f", bytes / best[q] * 1e-6);
        printf("   GB/s\n");
    }
    return 0;
}
