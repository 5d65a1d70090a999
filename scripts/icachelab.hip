// icachelab.hip -- is straight-line code in a short kernel bound by instruction FETCH?  The same 4096 v_fma per lane either as
// 4096 straight-line instructions (32 KiB of code, every line touched once) or as a 32-iteration loop over 128 (1 KiB, fetched once).
// build: hipcc --offload-arch=gfx950 -O3 scripts/icachelab.hip -o build_gpu/icachelab
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int N> __device__ __forceinline__ void body(float &a, float &b, float &c, float &d, float k)
{
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        asm volatile("v_fma_f32 %0, %0, %4, %1\n\tv_fma_f32 %1, %1, %4, %2\n\tv_fma_f32 %2, %2, %4, %3\n\tv_fma_f32 %3, %3, %4, %0"
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));
    }
}
template <int TOTAL> __global__ void straight(float *o, float k)
{
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
    body<TOTAL>(a, b, c, d, k);
    o[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int TOTAL, int CHUNK> __global__ void looped(float *o, float k)
{
    float a = threadIdx.x, b = 1.f, c = 2.f, d = 3.f;
#pragma unroll 1
    for (int it = 0; it < TOTAL / CHUNK; ++it) body<CHUNK>(a, b, c, d, k);
    o[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ void nullk() {}

int main()
{
    float *o; CK(hipMalloc(&o, 256 * 512 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    auto timeit = [&](const char *name, auto launch) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int r = 0; r < 50; ++r) launch();
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-64s %8.3f us per launch\n", name, ms * 1e3 / 500);
        return 0;
    };
    timeit("null", [&] { nullk<<<1, 64, 0, s>>>(); });
    for (int wg : {1, 32, 256}) for (int th : {64, 512}) {
        char nm[128];
        snprintf(nm, sizeof nm, "%3d WG x %3d thr:  512 fma straight (4 KiB)", wg, th);   timeit(nm, [&] { straight<512><<<wg, th, 0, s>>>(o, 1.0001f); });
        snprintf(nm, sizeof nm, "%3d WG x %3d thr:  512 fma loop 4 x 128", wg, th);        timeit(nm, [&] { looped<512, 128><<<wg, th, 0, s>>>(o, 1.0001f); });
        snprintf(nm, sizeof nm, "%3d WG x %3d thr: 1024 fma straight (8 KiB)", wg, th);   timeit(nm, [&] { straight<1024><<<wg, th, 0, s>>>(o, 1.0001f); });
        snprintf(nm, sizeof nm, "%3d WG x %3d thr: 1024 fma loop 8 x 128", wg, th);        timeit(nm, [&] { looped<1024, 128><<<wg, th, 0, s>>>(o, 1.0001f); });
        snprintf(nm, sizeof nm, "%3d WG x %3d thr: 4096 fma straight (32 KiB)", wg, th);  timeit(nm, [&] { straight<4096><<<wg, th, 0, s>>>(o, 1.0001f); });
        snprintf(nm, sizeof nm, "%3d WG x %3d thr: 4096 fma loop 32 x 128", wg, th);       timeit(nm, [&] { looped<4096, 128><<<wg, th, 0, s>>>(o, 1.0001f); });
    }
    return 0;
}
