// Latency of a dependent f32 add chain on one wavefront per SIMD (what bounds the bit-exact centroid update).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/chain_probe tools/probe/chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_reg_chain(float *out, int iters, float inc) {
    float acc = (float)threadIdx.x;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(inc));
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_reg_chain2(float *out, int iters, float inc) {     // two independent chains interleaved
    float a = (float)threadIdx.x, b = a + 1;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(inc)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(b) : "v"(inc)); }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a + b;
}
__global__ __launch_bounds__(64) void k_lds_chain(float *out, int iters) {                   // 16 broadcast ds_read_b128 + 64 adds per round
    __shared__ float4 st[2][16];
    if (threadIdx.x < 32) ((float4 *)st)[threadIdx.x] = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f);
    __syncthreads();
    float acc = 0.f;
    for (int i = 0; i < iters; i++) {
        const float4 *s4 = st[i & 1];
#pragma unroll
        for (int q = 0; q < 16; q++) { const float4 x = s4[q]; acc += x.x; acc += x.y; acc += x.z; acc += x.w; }
        __builtin_amdgcn_wave_barrier();
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}
int main() {
    float *out; CK(hipMalloc(&out, 1024 * 64 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    auto timeit = [&](const char *name, auto launch, int chains) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.3f ms  %6.2f ns per add step (%d chain(s))\n", name, ms, ms * 1e6 / ((double)iters * 64), chains);
        return 0;
    };
    timeit("register chain, 1 wave/CU (256 blocks)", [&] { hipLaunchKernelGGL(k_reg_chain, 256, 64, 0, 0, out, iters, 1e-3f); }, 1);
    timeit("register chain, 4 waves/CU (1024 blocks)", [&] { hipLaunchKernelGGL(k_reg_chain, 1024, 64, 0, 0, out, iters, 1e-3f); }, 1);
    timeit("two interleaved chains, 1 wave/CU", [&] { hipLaunchKernelGGL(k_reg_chain2, 256, 64, 0, 0, out, iters, 1e-3f); }, 2);
    timeit("LDS-fed chain (b128 broadcast), 1 wave/CU", [&] { hipLaunchKernelGGL(k_lds_chain, 256, 64, 0, 0, out, iters); }, 1);
    timeit("LDS-fed chain, 4 waves/CU", [&] { hipLaunchKernelGGL(k_lds_chain, 1024, 64, 0, 0, out, iters); }, 1);
    return 0;
}
