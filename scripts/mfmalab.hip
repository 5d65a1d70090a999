// mfmalab.hip -- what the bf16 matrix pipe sustains on this box: N back-to-back v_mfma_f32_16x16x32_bf16 per wave, 8 independent
// accumulator chains, no memory traffic.  build: hipcc --offload-arch=gfx950 -O3 scripts/mfmalab.hip -o build_gpu/mfmalab
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float *o, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *o; CK(hipMalloc(&o, 1024 * 1024 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wpc : {4, 8, 16}) for (int iters : {2000, 20000}) {
        const int wg = 256, th = 64 * wpc;
        k<<<wg, th>>>(o, 100); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); k<<<wg, th>>>(o, iters); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)wg * wpc * iters * 8 * (2.0 * 16 * 16 * 32);
        printf("%2d waves/CU, %6d x 8 MFMA per wave: %8.1f us  %7.1f TFLOP/s  (%.2f cycles per MFMA per SIMD at 2.4 GHz)\n", wpc, iters, ms * 1e3,
               flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wpc / 4));
    }
    return 0;
}
