// Issue rate of the VALU / LDS instruction classes the assign-type kernels are made of, on this part (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/valu_probe tools/probe/valu_probe.hip
// Every test runs REPS x 64 copies of one instruction on 8 independent register sets, W wavefronts per SIMD on every CU;
// reported: cycles per wave-instruction per SIMD at the measured shader clock (s_memtime vs wall clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int REPS = 256;

#define R8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define R64(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S) R8(S)

template <int T>
__global__ __launch_bounds__(256) void k_probe(float *out, const unsigned *idx) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
    __syncthreads();
    float f[8]; double d[8]; unsigned u[8]; unsigned long long q[8];
    float4 f4[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { f[i] = threadIdx.x * 0.001f + i; d[i] = f[i]; u[i] = idx[(threadIdx.x + 37 * i) & 255]; q[i] = u[i] * 0x100000001ULL; f4[i] = make_float4(0, 0, 0, 0); }
    const float c0 = 1.0001f, c1 = 0.5f;
    const double e0 = 1.0001, e1 = 0.5;
    unsigned long long vcc;
    (void)vcc;
    for (int r = 0; r < REPS; r++) {
        if constexpr (T == 0) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(c0), "v"(c1));
            R64(S)
#undef S
        } else if constexpr (T == 1) {
#define S(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d[i]) : "v"(e0));
            R64(S)
#undef S
        } else if constexpr (T == 2) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(e0), "v"(e1));
            R64(S)
#undef S
        } else if constexpr (T == 3) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e0));
            R64(S)
#undef S
        } else if constexpr (T == 4) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e1));
            R64(S)
#undef S
        } else if constexpr (T == 5) {
#define S(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            R64(S)
#undef S
        } else if constexpr (T == 6) {
#define S(i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i]));
            R64(S)
#undef S
        } else if constexpr (T == 7) {
#define S(i) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(f[i]));
            R64(S)
#undef S
        } else if constexpr (T == 8) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(c0), "v"(c1) : "vcc");
            R64(S)
#undef S
        } else if constexpr (T == 9) {
#define S(i) asm volatile("v_cmp_lt_u64 vcc, %1, %2" : "=s"(vcc) : "v"(q[i]), "v"(q[(i + 1) & 7]) : "vcc");
            R64(S)
#undef S
        } else if constexpr (T == 10) {
#define S(i) asm volatile("v_cmp_lt_f64 vcc, %1, %2" : "=s"(vcc) : "v"(d[i]), "v"(d[(i + 1) & 7]) : "vcc");
            R64(S)
#undef S
        } else if constexpr (T == 11) {
#define S(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(u[i]));
            R64(S)
#undef S
        } else if constexpr (T == 12) {
#define S(i) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            R64(S)
#undef S
        } else if constexpr (T == 13) {
#define S(i) asm volatile("v_med3_i32 %0, %0, 0, 63" : "+v"(u[i]));
            R64(S)
#undef S
        } else if constexpr (T == 14) {
#define S(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c1));
            R64(S)
#undef S
        } else if constexpr (T == 15) {          // LDS b32, lane-linear addresses (conflict-free)
            unsigned a = threadIdx.x * 4;
#define S(i) asm volatile("ds_read_b32 %0, %1" : "=v"(f[i]) : "v"(a));
            R64(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (T == 16) {          // LDS b32, random addresses
#define S(i) asm volatile("ds_read_b32 %0, %1" : "=v"(f[i]) : "v"(u[i] & 0x7ffc));
            R64(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (T == 17) {          // LDS b128, random 16-byte aligned addresses
#define S(i) asm volatile("ds_read_b128 %0, %1" : "=v"(f4[i]) : "v"(u[i] & 0x7ff0));
            R64(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (T == 18) {          // LDS b64, random 8-byte aligned addresses
#define S(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(u[i] & 0x7ff8));
            R64(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (T == 19) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(c1));
            R64(S)
#undef S
        } else if constexpr (T == 20) {
#define S(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            R64(S)
#undef S
        } else if constexpr (T == 21) {
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(e0));
            R64(S)
#undef S
        } else if constexpr (T == 22) {
#define S(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(c1) : "vcc");
            R64(S)
#undef S
        } else if constexpr (T == 23) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
            R64(S)
#undef S
        } else if constexpr (T == 24) {
#define S(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            R64(S)
#undef S
        } else if constexpr (T == 25) {          // LDS atomic add, random addresses
#define S(i) asm volatile("ds_add_u32 %0, %1" : : "v"(u[i] & 0x3fc), "v"(1u));
            R64(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)");
        } else if constexpr (T == 26) {
#define S(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            R64(S)
#undef S
        }
    }
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc += f[i] + (float)d[i] + (float)u[i] + (float)q[i] + f4[i].x + f4[i].w;
    if (acc == 1.2345e30f) out[0] = acc;
}

int main() {
    float *out; unsigned *idx;
    CK(hipMalloc(&out, 64)); CK(hipMalloc(&idx, 1024));
    std::vector<unsigned> h(256);
    unsigned long long z = 88172645463325252ULL;
    for (auto &v : h) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; v = (unsigned)(z >> 20); }
    CK(hipMemcpy(idx, h.data(), 1024, hipMemcpyHostToDevice));
    int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int khz = 2400000; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("CUs %d, clock attribute %d kHz\n", cus, khz);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_cvt_f64_f32", "v_cvt_i32_f64", "v_cvt_i32_f32",
                           "v_cmp_lt_f32+v_cndmask", "v_cmp_lt_u64", "v_cmp_lt_f64", "v_bfe_u32", "v_lshl_or_b32", "v_med3_i32", "v_max_f32",
                           "ds_read_b32 linear", "ds_read_b32 random", "ds_read_b128 random", "ds_read_b64 random", "v_add_f32", "v_cvt_f32_f64",
                           "v_pk_mul_f32", "v_cmp_lt_f32", "v_cndmask_b32", "v_min_u32", "ds_add_u32 random(256)", "v_mad_u32_u24"};
    for (int wps : {1, 4}) {
        printf("-- %d wavefront(s) per SIMD\n", wps);
        const int blocks = cus * wps;                         // 256 threads = 4 waves = one per SIMD
        auto run = [&](int t, auto kern) {
            for (int i = 0; i < 2; i++) hipLaunchKernelGGL(kern, blocks, 256, 32768, 0, out, idx);
            hipEventRecord(e0);
            for (int i = 0; i < 5; i++) hipLaunchKernelGGL(kern, blocks, 256, 32768, 0, out, idx);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double ninst = (double)REPS * 64 * (t == 8 ? 2 : 1) * wps;       // wave-instructions per SIMD per launch
            const double ns = ms / 5 * 1e6;
            printf("%-26s %7.2f ns/inst/SIMD  = %5.2f cycles @2.4GHz\n", names[t], ns / ninst, ns / ninst * 2.4);
        };
#define RUN(T) run(T, k_probe<T>);
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18)
        RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26)
    }
    return 0;
}
