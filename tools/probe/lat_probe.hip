// Dependent-instruction latency on a LONE wavefront (one 64-thread block on the whole GPU): what bounds the serial chains of the
// raster scan (k_mbd_scan) and the dither (k_dither).  Not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/lat_probe tools/probe/lat_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define R8(S) S S S S S S S S
#define R64(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S)
constexpr int REPS = 64;                       // x 64 instructions per repetition

template <int T>
__global__ __launch_bounds__(64) void k_lat(float *out, unsigned long long *ticks) {
    float a = threadIdx.x * 0.5f + 1.0f, b = 1.0001f, c = 0.25f;
    float x0 = a, x1 = a + 1;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < REPS; r++) {
        if constexpr (T == 0) { R64(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
        else if constexpr (T == 1) { R64(asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(c));) }
        else if constexpr (T == 2) { R64(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));) }
        else if constexpr (T == 3) { R64(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a));) }
        else if constexpr (T == 4) { R64(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a));) }
        else if constexpr (T == 5) { R64(asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");) }
        else if constexpr (T == 6) { R64(asm volatile("v_cmp_le_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]" : "+v"(a) : "v"(b), "v"(c) : "s20", "s21");) }
        else if constexpr (T == 7) { R64(asm volatile("v_mov_b32 %0, %0" : "+v"(a));) }
        else if constexpr (T == 8) { R64(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(x0), "+v"(x1) : "v"(c));) }   // two independent chains
        else if constexpr (T == 9) { R64(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&x0) : "v"(*(double *)&b));) }
        else if constexpr (T == 10) { R64(asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        else if constexpr (T == 11) { R64(asm volatile("v_readlane_b32 s20, %0, 5\n v_mov_b32 %0, s20" : "+v"(a) : : "s20");) }
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) ticks[0] = t1 - t0;
    out[threadIdx.x] = a + x0 + x1;
}

template <int T>
int run(const char *name, int per_rep) {
    float *out; unsigned long long *tk;
    CK(hipMalloc(&out, 64 * 4)); CK(hipMalloc(&tk, 8));
    hipLaunchKernelGGL(k_lat<T>, 1, 64, 0, 0, out, tk);
    hipLaunchKernelGGL(k_lat<T>, 1, 64, 0, 0, out, tk);
    CK(hipDeviceSynchronize());
    unsigned long long h = 0;
    CK(hipMemcpy(&h, tk, 8, hipMemcpyDeviceToHost));
    const double ns = (double)h * 10.0;                    // wall_clock64: 100 MHz
    printf("%-52s %7.2f ns per instruction (%d in the chain)\n", name, ns / (REPS * 64.0 * per_rep), REPS * 64 * per_rep);
    return 0;
}
int main() {
    run<7>("v_mov_b32 (dependent)", 1);
    run<0>("v_add_f32 (dependent)", 1);
    run<1>("v_max_f32 (dependent)", 1);
    run<10>("v_min3_f32 (dependent)", 1);
    run<9>("v_pk_add_f32 (dependent)", 1);
    run<8>("v_add_f32 x2 independent chains, per instruction", 2);
    run<2>("s_nop 1 + v_mov_b32_dpp wave_shr:1 (dependent), per pair", 1);
    run<3>("s_nop 1 + v_mov_b32_dpp row_shr:1 (dependent), per pair", 1);
    run<4>("s_nop 1 + v_mov_b32_dpp row_bcast:15 (dependent), per pair", 1);
    run<5>("v_cmp -> vcc -> v_cndmask (dependent), per pair", 1);
    run<6>("v_cmp -> sgpr pair -> v_cndmask (dependent), per pair", 1);
    run<11>("v_readlane -> sgpr -> v_mov (dependent), per pair", 1);
    return 0;
}
