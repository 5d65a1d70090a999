// flaglab.hip -- what does an in-kernel producer -> consumer hand-over through a global flag cost on MI355X (8 XCDs, one L2 each)?
// A: three dependent kernels in a hipGraph (producer 24 WGs -> middle 128 WGs -> final 24 WGs), each moving a few KB.
// B: ONE kernel with the three roles by blockIdx, synchronised by release-add / acquire-spin on two counters (bounded spin).
// build: hipcc --offload-arch=gfx950 -O3 scripts/flaglab.hip -o build_gpu/flaglab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int NP = 32, NC = 128, NF = 24, T = 256, VEC = 256;       // a = NP * VEC = 8192 floats (one n = 8192 activation row)

__device__ __forceinline__ void spin_until(int *cnt, int target, int *err)
{
    if (threadIdx.x == 0) {
        int it = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++it > (1 << 22)) { *err = 1; break; }
        }
    }
    __syncthreads();
}
__device__ __forceinline__ void signal(int *cnt)
{
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ void produce(const float *in, float *a, int b, float salt)
{
    for (int i = threadIdx.x; i < VEC; i += T) a[b * VEC + i] = in[b * VEC + i] * 2.f + salt;
}
__device__ void middle(const float *a, float *y, int b)
{   // every middle WG reads ALL of a (like a GEMM WG reads all of xt)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4 *>(a)[threadIdx.x + T * u];     // 8 loads in flight
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) y[b] = red[0] + red[1] + red[2] + red[3] + b;
}
__device__ void final_(const float *y, float *out, int b)
{
    float s = 0.f;
    for (int i = threadIdx.x; i < NC; i += T) s += y[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[b] = red[0] + red[1] + red[2] + red[3];
}
__global__ void k_prod(const float *in, float *a, float salt) { produce(in, a, blockIdx.x, salt); }
__global__ void k_mid(const float *a, float *y) { middle(a, y, blockIdx.x); }
__global__ void k_fin(const float *y, float *out) { final_(y, out, blockIdx.x); }

__global__ void k_fused(const float *in, float *a, float *y, float *out, int *cnt, int *err, float salt)
{
    const int b = blockIdx.x;
    if (b < NP) {
        produce(in, a, b, salt);
        signal(cnt + 0);
    } else if (b < NP + NC) {
        spin_until(cnt + 0, NP, err);
        middle(a, y, b - NP);
        signal(cnt + 1);
    } else {
        spin_until(cnt + 1, NC, err);
        final_(y, out, b - NP - NC);
        __syncthreads();
        if (threadIdx.x == 0) {                                  // last one out resets the counters for the next launch
            if (__hip_atomic_fetch_add(cnt + 2, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == NF - 1) {
                __hip_atomic_store(cnt + 0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 2, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
__global__ void k_null() {}

int main()
{
    float *in, *a, *y, *out; int *cnt, *err;
    CK(hipMalloc(&in, NP * VEC * 4)); CK(hipMalloc(&a, NP * VEC * 4)); CK(hipMalloc(&y, NC * 4)); CK(hipMalloc(&out, NF * 4));
    CK(hipMalloc(&cnt, 64)); CK(hipMalloc(&err, 4)); CK(hipMemset(cnt, 0, 64)); CK(hipMemset(err, 0, 4));
    std::vector<float> h(NP * VEC);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 7) * 0.25f;
    CK(hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int REP = 50;
    auto timeit = [&](const char *name, auto enqueue) -> int {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int r = 0; r < REP; ++r) enqueue((float)r);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e0, s));
        for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        float o[NF]; CK(hipMemcpy(o, out, sizeof o, hipMemcpyDeviceToHost));
        int e; CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
        printf("%-44s %8.3f us per chain   out[0]=%.1f out[23]=%.1f err=%d\n", name, ms * 1e3 / (10 * REP), o[0], o[NF - 1], e);
        return 0;
    };
    timeit("null kernel", [&](float) { k_null<<<1, 64, 0, s>>>(); });
    timeit("3 kernels (32 -> 128 -> 24 workgroups)", [&](float salt) {
        k_prod<<<NP, T, 0, s>>>(in, a, salt); k_mid<<<NC, T, 0, s>>>(a, y); k_fin<<<NF, T, 0, s>>>(y, out); });
    timeit("1 kernel, two flag hand-overs", [&](float salt) { k_fused<<<NP + NC + NF, T, 0, s>>>(in, a, y, out, cnt, err, salt); });
    timeit("3 kernels again", [&](float salt) {
        k_prod<<<NP, T, 0, s>>>(in, a, salt); k_mid<<<NC, T, 0, s>>>(a, y); k_fin<<<NF, T, 0, s>>>(y, out); });
    return 0;
}
