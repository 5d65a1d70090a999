// mfma_lab.hip -- issue rate of the two bf16 MFMA shapes on this GPU: a loop of NACC independent products per wave, W waves per SIMD,
// with constant operands (1.0 x 0.5) and with random ones (N(0,1) per lane): the second is what a GEMM feeds the pipe.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_lab.hip -o build_gpu/mfma_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// four independent products per asm statement (the accumulators stay where the register allocator put them; hipcc's own loop over the builtin rotates
// them through shifted ranges and pads with s_nop)
#define MFMA4(OP, A0, A1, A2, A3) asm volatile(OP " %0, %4, %5, %0\n\t" OP " %1, %4, %5, %1\n\t" OP " %2, %4, %5, %2\n\t" OP " %3, %4, %5, %3" \
                                              : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3) : "v"(a), "v"(b))
template <int NACC> __global__ __launch_bounds__(256) void k16(float *out, int iters, bf16x8_t a, bf16x8_t b, const bf16x8_t *rnd)
{
    if (rnd) { a = rnd[threadIdx.x]; b = rnd[256 + threadIdx.x]; }
    f32x4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; it += 8)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; i += 4) MFMA4("v_mfma_f32_16x16x32_bf16", acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> __global__ __launch_bounds__(256) void k32(float *out, int iters, bf16x8_t a, bf16x8_t b, const bf16x8_t *rnd)
{
    if (rnd) { a = rnd[threadIdx.x]; b = rnd[256 + threadIdx.x]; }
    f32x16_t acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; it += 8)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; i += 4) MFMA4("v_mfma_f32_32x32x16_bf16", acc[i], acc[i + 1], acc[i + 2], acc[i + 3]);
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class F> static void run(const char *name, F launch, double flops_per_wave_iter, int nblk, int iters)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(nblk, 100); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0)); launch(nblk, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double tf = flops_per_wave_iter * iters * nblk * 4 / (best * 1e-3) / 1e12;
    printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", name, best, tf);
}
int main(int argc, char **argv)
{
    float *out; CK(hipMalloc(&out, 4096 * 256 * 4));
    bf16x8_t a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)1.0f; b[i] = (__bf16)0.5f; }
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;                 // argv[1]: products per accumulator (long runs for clock sampling); argv[2]: 0 / 1 = only that operand kind
    const int only = argc > 2 ? atoi(argv[2]) : -1;
    bf16x8_t *rnd; CK(hipMalloc(&rnd, 512 * 16));
    {
        __bf16 h[512 * 8]; unsigned st = 12345u;
        for (int i = 0; i < 512 * 8; ++i) {
            st = st * 1664525u + 1013904223u; const float u1 = ((st >> 8) + 1) / 16777217.0f;
            st = st * 1664525u + 1013904223u; const float u2 = (st >> 8) / 16777216.0f;
            h[i] = (__bf16)(sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2));
        }
        CK(hipMemcpy(rnd, h, sizeof h, hipMemcpyHostToDevice));
    }
    for (int pass = 0; pass < 2; ++pass) {
        if (only >= 0 && pass != only) continue;
        const bf16x8_t *rp = pass ? rnd : nullptr;
        printf("--- operands: %s\n", pass ? "random N(0,1) per lane" : "constant 1.0 x 0.5");
        for (int wps = 1; wps <= 3; ++wps) {                               // waves per SIMD = blocks per CU (4 waves each)
            const int nblk = 256 * wps;
            char nm[96];
            snprintf(nm, sizeof nm, "16x16x32 bf16, 4 acc, %d wave(s) / SIMD", wps);
            run(nm, [&](int n, int it) { k16<4><<<n, 256>>>(out, it, a, b, rp); }, 4 * 2.0 * 16 * 16 * 32, nblk, iters);
            snprintf(nm, sizeof nm, "16x16x32 bf16, 16 acc, %d wave(s) / SIMD", wps);
            run(nm, [&](int n, int it) { k16<16><<<n, 256>>>(out, it, a, b, rp); }, 16 * 2.0 * 16 * 16 * 32, nblk, iters);
            snprintf(nm, sizeof nm, "32x32x16 bf16, 4 acc, %d wave(s) / SIMD", wps);
            run(nm, [&](int n, int it) { k32<4><<<n, 256>>>(out, it, a, b, rp); }, 4 * 2.0 * 32 * 32 * 16, nblk, iters);
        }
    }
    return 0;
}
