// xingest_lab.hip -- lab (not part of the library): how fast can EVERY workgroup of a 256-workgroup launch pull the same 128 KiB of x
// (16 rows x 4096 bf16) out of the L2s into its CU?  That phase is 2.2 of the 3.4 us dq_h_kernel spends behind the launch boundary
// (profiles/r02c_k2probe_timeline.log: "last chunk landed" at 5241 of 8232 ticks), against 128 KiB / 64 B per clock = 2048 clocks of
// pure L1 return rate.  Variants (8 waves, wave w owns the 16 x 256 slabs of chunks {w, w + 8}, like dq_h_kernel<2, ., 1, 8, 2>):
//   dma      buffer_load_dwordx4 ... lds, every workgroup in the same address order                       (what the kernel does)
//   dma_rot  the same, chunk order rotated by the workgroup index: at any moment the 32 CUs of an XCD ask different L2 channels
//   dma_rot2 ... and the 8 instructions of a slab rotated as well
//   reg      global_load_dwordx4 into registers (no LDS write), same order / reg_rot rotated
//   half     dma, 8 of the 16 rows (64 KiB): the HALF path of bs <= 8
//   own      dma, every workgroup its OWN copy of x (256 x 128 KiB = 32 MiB: no line is shared; HBM / Infinity Cache behind it)
//   xcd      dma, one copy of x per XCD (blockIdx % 8)
// Reported: per-launch period inside a hipGraph of 200 launches (null kernel of the same geometry beside it) and, from s_memtime stamps,
// the clocks from a workgroup's first wave start to its last wave's vmcnt(0) (median / p10 / p90 over workgroups).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build_gpu/xingest_lab scripts/xingest_lab.hip && build_gpu/xingest_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = 16, D = 4096, ROWB = D * 2, NW = 8, NCH = 2, KC = 256, SLAB = ROWS * KC * 2;   // slab 8 KiB

enum { V_DMA = 0, V_DMA_ROT, V_DMA_ROT2, V_REG, V_REG_ROT, V_HALF, V_OWN, V_XCD, V_NULL, NVAR };
static const char *vname[NVAR] = {"dma", "dma_rot", "dma_rot2", "reg", "reg_rot", "half", "own", "xcd", "null"};

template <int V>
__global__ __launch_bounds__(64 * NW) void ingest_kernel(const uint16_t *x, unsigned long long *stamps, uint32_t *sink)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (V == V_NULL) {
        if (stamps && lane == 0) { stamps[(blockIdx.x * NW + wave) * 2] = t0; stamps[(blockIdx.x * NW + wave) * 2 + 1] = t0; }
        return;
    }
    const uint16_t *xb = x;
    if (V == V_OWN) xb = x + (size_t)blockIdx.x * ROWS * D;
    if (V == V_XCD) xb = x + (size_t)(blockIdx.x & 7) * ROWS * D;
    const bool rot = V == V_DMA_ROT || V == V_DMA_ROT2 || V == V_REG_ROT;
    const int r0 = rot ? (int)(blockIdx.x >> 3) : 0;                  // (blockIdx & 7 is the XCD: rotate among the CUs of one XCD)
    char *myreg = smem + wave * (NCH * SLAB);
    uint32_t acc = 0;
    if (V == V_REG || V == V_REG_ROT) {
        u32x4 v[NCH][8];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int kc = (i * NW + wave + r0) & (NW * NCH - 1);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // instruction q: 128-byte column block q >> 1 of rows 8 (q & 1) .. + 7; lane = 8 * (row & 7) + 16-byte slot
                const int row = 8 * (q & 1) + (lane >> 3);
                const char *p = reinterpret_cast<const char *>(xb) + (size_t)row * ROWB + kc * (KC * 2) + (q >> 1) * 128 + (lane & 7) * 16;
                v[i][q] = *reinterpret_cast<const u32x4 *>(p);
            }
        }
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= v[i][q][0] ^ v[i][q][1] ^ v[i][q][2] ^ v[i][q][3];
    } else {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)xb, 0, ROWS * ROWB, 0x00020000);
        const uint32_t voff_lo = (lane >> 3) * ROWB + ((uint32_t)(lane & 7) << 4);
        const uint32_t voff_hi = voff_lo + 8u * ROWB;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int kc = (i * NW + wave + r0) & (NW * NCH - 1);
#pragma unroll
            for (int q0 = 0; q0 < 8; ++q0) {
                const int q = V == V_DMA_ROT2 ? (q0 + (int)(blockIdx.x >> 3)) & 7 : q0;
                if (V == V_HALF && (q & 1)) continue;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(myreg + i * SLAB + q * 1024), 16, (q & 1) ? voff_hi : voff_lo,
                                                         kc * (KC * 2) + (q >> 1) * 128, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = *reinterpret_cast<const uint32_t *>(myreg + lane * 4);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) sink[0] = acc;                                 // keep the loads alive
    if (stamps && lane == 0) { stamps[(blockIdx.x * NW + wave) * 2] = t0; stamps[(blockIdx.x * NW + wave) * 2 + 1] = t1; }
}

// second generation of variants: run-time pattern.  kc = pattern(i, wave, cu):
//   mode 0: (i NW + wave + mul cu) & 15        mode 1: (i NW + wave) ^ (mul cu & 15)        mode 2: (2 wave + i + mul cu) & 15 (a wave's two chunks adjacent)
//   mode 3: like 0, and the 8 instructions of a slab issued column block by column block for BOTH chunks interleaved
__global__ __launch_bounds__(64 * NW) void ingest2_kernel(const uint16_t *x, unsigned long long *stamps, uint32_t *sink, int mode, int mul)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_readcyclecounter();
    const int cu = (int)(blockIdx.x >> 3);
    char *myreg = smem + wave * (NCH * SLAB);
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, ROWS * ROWB, 0x00020000);
    const uint32_t voff_lo = (lane >> 3) * ROWB + ((uint32_t)(lane & 7) << 4);
    const uint32_t voff_hi = voff_lo + 8u * ROWB;
    int kcs[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        kcs[i] = mode == 1 ? ((i * NW + wave) ^ ((mul * cu) & 15)) : mode == 2 ? ((2 * wave + i + mul * cu) & 15) : ((i * NW + wave + mul * cu) & 15);
    if (mode == 4 || mode == 6) {
        // round 6: the FOOTPRINT of one DMA instruction.  Rounds 2-5: 8 rows x 128 B (eight lines 8 KiB apart).  Here: 2 rows x 512 B -- lanes
        // 0..31 walk 512 contiguous bytes of row 2 q, lanes 32..63 of row 2 q + 1 (two runs of four consecutive lines); chunks adjacent (mode 2's order)
        const uint32_t vw = (uint32_t)(lane >> 5) * ROWB + ((uint32_t)(lane & 31) << 4);
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int kc = (2 * wave + i + (mode == 6 ? mul * cu : 0)) & 15;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(myreg + i * SLAB + q * 1024), 16, vw + (uint32_t)(2 * q) * ROWB, kc * (KC * 2), 0, 0);
        }
    } else if (mode == 5) {
        // ... and 1 row x 1 KiB: the wave's two adjacent chunks of ONE row per instruction (eight consecutive lines), 16 instructions
        const uint32_t vr = (uint32_t)lane << 4;
#pragma unroll
        for (int n = 0; n < 16; ++n)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(myreg + n * 1024), 16, vr + (uint32_t)n * ROWB, ((2 * wave) & 15) * (KC * 2), 0, 0);
    } else if (mode == 3) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < NCH; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(myreg + i * SLAB + q * 1024), 16, (q & 1) ? voff_hi : voff_lo,
                                                         kcs[i] * (KC * 2) + (q >> 1) * 128, 0, 0);
    } else {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t *)(myreg + i * SLAB + q * 1024), 16, (q & 1) ? voff_hi : voff_lo,
                                                         kcs[i] * (KC * 2) + (q >> 1) * 128, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t acc = *reinterpret_cast<const uint32_t *>(myreg + lane * 4);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) sink[0] = acc;
    if (stamps && lane == 0) { stamps[(blockIdx.x * NW + wave) * 2] = t0; stamps[(blockIdx.x * NW + wave) * 2 + 1] = t1; }
}

static void run2(const char *name, int mode, int mul, const uint16_t *x, unsigned long long *stamps, uint32_t *sink, hipStream_t st)
{
    const int G = 256, steps = 200;
    const size_t lds = (size_t)NW * NCH * SLAB;
    CK(hipFuncSetAttribute((const void *)ingest2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < steps; ++i) ingest2_kernel<<<G, 64 * NW, lds, st>>>(x, nullptr, sink, mode, mul);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    std::vector<unsigned long long> h((size_t)G * NW * 2);
    for (int rep = 0; rep < 5; ++rep) { ingest2_kernel<<<G, 64 * NW, lds, st>>>(x, stamps, sink, mode, mul); CK(hipStreamSynchronize(st)); }
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> dur;
    for (int b = 0; b < G; ++b) {
        unsigned long long a = ~0ull, z = 0;
        for (int w = 0; w < NW; ++w) { a = std::min(a, h[(b * NW + w) * 2]); z = std::max(z, h[(b * NW + w) * 2 + 1]); }
        dur.push_back((double)(z - a));
    }
    std::sort(dur.begin(), dur.end());
    printf("%-22s %7.3f us/launch in a graph   workgroup ingest clocks: median %6.0f  p10 %6.0f  p90 %6.0f\n", name, best * 1e3 / steps,
           dur[dur.size() / 2], dur[dur.size() / 10], dur[dur.size() * 9 / 10]);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

template <int V> static void run(const uint16_t *x, unsigned long long *stamps, uint32_t *sink, hipStream_t st)
{
    const int G = 256, steps = 200;
    const size_t lds = (size_t)NW * NCH * SLAB;
    auto kern = ingest_kernel<V>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < steps; ++i) kern<<<G, 64 * NW, lds, st>>>(x, nullptr, sink);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    // stamps: five single launches, the last one read back
    std::vector<unsigned long long> h((size_t)G * NW * 2);
    for (int rep = 0; rep < 5; ++rep) { kern<<<G, 64 * NW, lds, st>>>(x, stamps, sink); CK(hipStreamSynchronize(st)); }
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> dur;
    for (int b = 0; b < G; ++b) {
        unsigned long long a = ~0ull, z = 0;
        for (int w = 0; w < NW; ++w) { a = std::min(a, h[(b * NW + w) * 2]); z = std::max(z, h[(b * NW + w) * 2 + 1]); }
        dur.push_back((double)(z - a));
    }
    std::sort(dur.begin(), dur.end());
    printf("%-9s %7.3f us/launch in a graph   workgroup ingest clocks: median %6.0f  p10 %6.0f  p90 %6.0f\n", vname[V], best * 1e3 / steps,
           dur[dur.size() / 2], dur[dur.size() / 10], dur[dur.size() * 9 / 10]);
    fflush(stdout);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

int main()
{
    hipStream_t st;
    CK(hipStreamCreate(&st));
    uint16_t *x; unsigned long long *stamps; uint32_t *sink;
    const size_t xbytes = (size_t)256 * ROWS * D * 2;                      // 256 copies (the `own` variant)
    CK(hipMalloc(&x, xbytes)); CK(hipMemset(x, 0x11, xbytes));
    CK(hipMalloc(&stamps, (size_t)256 * NW * 2 * 8)); CK(hipMalloc(&sink, 64));
    run<V_NULL>(x, stamps, sink, st);
    for (int rep = 0; rep < 1; ++rep) {
        run<V_DMA>(x, stamps, sink, st); run<V_REG>(x, stamps, sink, st); run<V_HALF>(x, stamps, sink, st);
    }
    printf("-- run-time patterns (mode, multiplier of the workgroup's index in its XCD)\n");
    for (int rep = 0; rep < 2; ++rep) {
        run2("add x0 (lockstep)", 0, 0, x, stamps, sink, st);
        run2("add x1", 0, 1, x, stamps, sink, st); run2("add x3", 0, 3, x, stamps, sink, st); run2("add x5", 0, 5, x, stamps, sink, st);
        run2("add x7", 0, 7, x, stamps, sink, st); run2("add x8 (two phases)", 0, 8, x, stamps, sink, st); run2("add x2", 0, 2, x, stamps, sink, st);
        run2("xor x1", 1, 1, x, stamps, sink, st); run2("xor x5", 1, 5, x, stamps, sink, st);
        run2("adjacent x1", 2, 1, x, stamps, sink, st); run2("adjacent x2", 2, 2, x, stamps, sink, st); run2("adjacent x0", 2, 0, x, stamps, sink, st);
        run2("interleaved x1", 3, 1, x, stamps, sink, st); run2("interleaved x0", 3, 0, x, stamps, sink, st);
        run2("2 rows x 512 B, adjacent", 4, 0, x, stamps, sink, st); run2("2 rows x 512 B, adj, rot", 6, 1, x, stamps, sink, st);
        run2("1 row x 1 KiB, adjacent", 5, 0, x, stamps, sink, st);
    }
    return 0;
}
