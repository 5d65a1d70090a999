// xcdbw_lab.hip -- lab (round 6): how fast can the workgroups of ONE XCD stream from HBM?  A decode step that ran as one persistent launch on one
// XCD would trade its launch boundaries (1.6-2.6 us each, 5 / 15 per block) for in-L2 exchanges (0.55 us, scripts/xcdsync_lab.hip) -- if 32 CUs
// can pull a block's 12.6 MB of codes fast enough.  256 one-per-CU workgroups are launched; those on XCD 0 (HW_REG_XCC_ID) stream `bytes` with
// 16-byte non-temporal loads, the others leave (or, with all = 1, every workgroup streams its share: the whole-chip figure for comparison).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcdbw_lab scripts/xcdbw_lab.hip && /tmp/xcdbw_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void stream_kernel(const u32x4 *src, size_t n16, int all, int *claim, unsigned *sink, int nwg)
{
    __shared__ int slot_s;
    if (threadIdx.x == 0) {
        int slot = blockIdx.x;
        if (!all) {
            const int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;
            slot = xcc == 0 ? atomicAdd(claim, 1) : -1;
        }
        slot_s = slot;
    }
    __syncthreads();
    const int slot = slot_s;
    if (slot < 0 || slot >= nwg) return;
    unsigned acc = 0;
    const size_t per = n16 / nwg;
    const u32x4 *p = src + (size_t)slot * per;
    for (size_t i = threadIdx.x; i + 3 * 1024 < per; i += 4 * 1024) {
        const u32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + 1024);
        const u32x4 c = __builtin_nontemporal_load(p + i + 2048), d = __builtin_nontemporal_load(p + i + 3072);
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    const size_t bytes = 1024ull << 20;
    u32x4 *src;
    int *claim;
    unsigned *sink;
    hipMalloc(&src, bytes);
    hipMalloc(&claim, 4);
    hipMalloc(&sink, 4);
    hipMemset(src, 1, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int all = 0; all < 2; ++all)
        for (int threads : {256, 512, 1024})
            for (size_t mb : {16, 64, 256}) {
                const int nwg = all ? 256 : 32;
                float best = 1e30f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipMemset(claim, 0, 4);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    stream_kernel<<<256, threads>>>(src + (size_t)rep * ((192ull << 20) / 16), (mb << 20) / 16, all, claim, sink, nwg);
                    hipEventRecord(e1);
                    hipDeviceSynchronize();
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("%s  %4d threads per workgroup  %4zu MB   %8.1f us   %7.1f GB/s\n", all ? "all 256 CUs" : "XCD 0 (32) ", threads, mb, best * 1e3f,
                       (double)(mb << 20) / (best * 1e-3) / 1e9);
            }
    return 0;
}
