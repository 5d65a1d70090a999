// dqgemm_v2.h -- K2, second generation: three kernels that share one idea -- every byte a wave needs is REQUESTED
// up front and consumed behind COUNTED waits, so compute follows the data as it lands instead of waiting for all of it.
//
//   dq_h_kernel   small layers, bs <= 16 (the 4096 x 4096 headline): one workgroup per 16-row tile, the whole of x
//                 staged once into LDS (wave-private slabs, no barrier before compute), k-partials meet in LDS.
//   dq_s_kernel   big layers, bs <= 16: a pure weight stream.  One row tile per wave, x slabs staged ONCE per workgroup
//                 in an LDS ring (x ingest = weight bytes / (NW/8)), weight tiles prefetched P stages deep in registers.
//   dq_mb_kernel  bs > 16: (WR x WB) waves, each RTw row tiles x BTw batch tiles of accumulators; x staged 64 k at a
//                 time through a 3-deep LDS ring, every dequantised A fragment feeds BTw MFMAs, every B fragment RTw.
//
// hipcc (ROCm 7.2) waits vmcnt(0) in front of any ds_read it can see, and at the first use of any ordinary load result,
// while an LDS-DMA of the same wave is pending (cdna_hip_programming.md 5: "three .s-level traps") -- which turns a
// pipeline into "wait for everything".  dq_h_kernel (straight-line, one shot) hides its ds_reads and weight loads in
// inline asm (5.7 form (ii)); the two streaming kernels split the roles by wave instead (loader waves own the DMA queue).
#pragma once
#include "dq_common.h"
#include <type_traits>
#include <utility>

// Probe hooks: scripts/k2lab.hip defines K2_PROBE before including this file and gets s_memtime stamps / wait accounting;
// in the library they expand to nothing.
#include "probe.h"
#ifndef K2_PROBE
#define K2_STAMP(i) QA_STAMP(i)            /* nothing in the shipped library; the in-situ stamps of csrc/probe.h in the probe build */
#define K2_STAMP_FLUSH() do { } while (0)
#define K2_ACC_DECL do { } while (0)
#define K2_ACC(slot, stmt) stmt
#define K2_ACC_FLUSH() do { } while (0)
#endif

namespace {

// compile-time unrolled loop: f(std::integral_constant<int, 0>) ... f(<N-1>) -- asm immediates need constants
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}

typedef __attribute__((address_space(3))) void lds_void2_t;

__device__ __forceinline__ uint32_t lds_addr(const void *p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char *)p;
}

template <int N> __device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// the same wait, naming weight registers written by an asm load (form (ii))
template <int N> __device__ __forceinline__ void wait_vm(u32x4 &a)
{
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(N) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm(u32x4 &a, u32x4 &b)
{
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

// 16 bytes per lane, non-temporal (a weight tile is read by exactly one wave, once)
__device__ __forceinline__ void load_w_nt(u32x4 &dst, const u32x4 *p)
{
    asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
}
template <int OFF> __device__ __forceinline__ void lds_read16(u32x4 &dst, uint32_t addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds offset is 16 bit");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void wait_lgkm(u32x4 (&f)[8])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7])::"memory");
}
__device__ __forceinline__ void wait_lgkm(u32x4 (&f)[4])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
}
__device__ __forceinline__ void wait_lgkm(u32x4 (&f)[2])
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1])::"memory");
}

struct K2Args {
    const uint16_t *x;
    const u32x4 *qw;
    EpiArgs e;
    int64_t d;
};

// =====================================================================================================================
// dq_h_kernel: bs <= 16, d / KC <= NW * NCH chunks.  grid.x = m / (16 * RT).
// Wave w owns chunks w, w + NW, ... (NCH of them) of the workgroup's RT row tiles: it requests the RT weight tiles and
// DMAs the 16 x KC slab of x for each chunk into its own LDS region, all at once, then consumes chunk i behind
// vmcnt((NCH-1-i) * ops per chunk).  RT > 1 (fewer workgroups, each ingesting all of x once for RT tiles) lost at
// every shape tried (profiles/r02b_k2lab.log): the per-wave instruction stream, not the x traffic through the L2s, grows.
// x slab image (16-byte units, as in round 1): DMA instruction q = 2*cb + rh moves 128-B column block cb of rows
// 8*rh .. 8*rh+7; lane L = 8*(row&7) + slot fetches logical 16-B column slot ^ (row&7); logical (row b, column c16) sits at
// unit 64*(2*(c16>>3) + (b>>3)) + 8*(b&7) + ((c16&7) ^ (b&7)): every ds_read_b128 group hits 16 distinct slots.
// The weight tiles are asm loads (form (ii) of cdna_hip_programming.md 5.7): straight-line code, consumed right
// behind the wait that names them; tests/test_k2_isa.py audits the generated code for touches in between.
// =====================================================================================================================
template <int BITS, class ACT, int RT, int NW, int NCH, bool HALF, bool EXACT>   // HALF: bs <= 8; EXACT: d / KC == NW * NCH
__global__ __launch_bounds__(64 * NW) void dq_h_kernel(K2Args A)
{
#include "dq_h_body.inc"
}

// up to three problems of ONE shape in one launch (blockIdx.y picks; q / k / v, gate / up of a decode step with 5..16 rows: round 5).  The
// arguments of the picked problem are COPIED out of the kernarg segment first; the body is the same text as dq_h_kernel's.
struct K2GArgs {
    K2Args g[3];
};
template <int BITS, class ACT, int RT, int NW, int NCH, bool HALF, bool EXACT>
__global__ __launch_bounds__(64 * NW) void dq_hg_kernel(K2GArgs G)
{
    const K2Args A = G.g[blockIdx.y];
#include "dq_h_body.inc"
}

// =====================================================================================================================
// dq_hr_kernel (round 6): the one-pass kernel with the activations straight into REGISTERS.
// profiles/r06_k2h_stamps.txt (in-situ stamps of dq_h_kernel, 4 x 4, cold): a wave needs 3800 clocks just to ISSUE its 4 weight loads and 32
// LDS-DMA instructions (the CU's vector-memory queue fills), the last slab lands at 6150 -- 144 KiB through the LDS-DMA path at 24 B per
// clock, against 64 B per clock the L1 returns to registers.  The B fragment of MFMA step t of k-chunk c is, per lane (batch row j, k group g),
// 16 CONSECUTIVE bytes of row j of x: x[j][256 c + 32 t + 8 g .. + 8) -- row-major x as the caller hands it over needs no LDS hop and no
// swizzle at all.  So: one address register pair per wave, 32 x global_load_dwordx4 with immediate offsets (a wave's 4 chunks are adjacent:
// 2 KiB of every row), requested right behind the 4 weight tiles, each MFMA step behind `s_waitcnt vmcnt(31 - 8 i - t)` -- the asm-load
// discipline of the weight tiles (form (ii); tests/test_k2_isa.py audits this kernel too).  LDS is the meet's only.
// Exact fit only (d = 256 NW NCH), one row tile per workgroup, 2-bit; batch rows past bs re-read row bs - 1 (their columns are never stored).
// =====================================================================================================================
template <int BITS, class ACT, int NW, int NCH>
__global__ __launch_bounds__(64 * NW) void dq_hr_kernel(K2Args A)
{
    typedef DeqT<BITS, ACT> Q;
    constexpr int KC = Q::KC, NT = Q::NT;
    static_assert(BITS == 2 && NT == 8 && KC == 256, "2-bit tiles: 256 columns, 8 MFMA steps");
    static_assert(4 + NCH * NT <= 63, "vmcnt range");
    constexpr int NLD = NCH * NT;                                        // x loads of a wave, behind its NCH weight loads
    extern __shared__ __attribute__((aligned(16))) char smem[];         // [NW][1024 + 64]: the k-partials of the tile, then 16 row sums
    const EpiArgs &e = A.e;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    K2_STAMP(0);
    const int j = lane & 15, g = lane >> 4;
    const uint32_t nkc = (uint32_t)(NW * NCH);
    const uint32_t rt0 = blockIdx.x;
    char *myreg = smem + wave * (1024 + 64);

    float e_sc = 0.f, e_zr = 0.f, e_bi = 0.f;                             // reducer role q = wave: its epilogue parameters, requested now
    {
        const int64_t row = (int64_t)rt0 * 16 + (lane & 15);
        e_sc = e.qfn == QUIPAMD_QFN_B ? e.scale[0] : e.scale[row];
        if (e.qfn != QUIPAMD_QFN_B) e_zr = e.zero[row];
        if (e.bias) e_bi = e.bias[row];
    }
    // ---- request everything: the weight tiles (HBM), then the B fragments (L2) -----------------------------------------------------
    u32x4 w[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) load_w_nt(w[i], A.qw + ((uint64_t)rt0 * nkc + (uint32_t)(NCH * wave + i)) * 64 + lane);
    const int jr = j < (int)e.bs ? j : (int)e.bs - 1;
    const char *xb = reinterpret_cast<const char *>(A.x) + ((int64_t)jr * A.d + (int64_t)(NCH * wave) * KC + 8 * g) * 2;
    u32x4 xf[NCH][NT];
    static_for<NLD>([&](auto IT) {
        constexpr int i = decltype(IT)::value / NT, t = decltype(IT)::value % NT;
        u32x4 &dst = xf[i][t];                                            // (named outside the asm statement: a generic lambda captures nothing an
        const char *src = xb;                                             //  asm operand alone mentions)
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(dst) : "v"(src), "n"((i * KC + 32 * t) * 2) : "memory");
    });
    K2_STAMP(1);
    f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const u32x4 ones = {opaque(ACT::ONES), opaque(ACT::ONES), opaque(ACT::ONES), opaque(ACT::ONES)};
    static_for<NLD>([&](auto IT) {
        constexpr int i = decltype(IT)::value / NT, t = decltype(IT)::value % NT;
        // memory returns in order: behind this wait the NCH weight tiles and the fragments up to (i, t) have landed
        u32x4 &wi = w[i], &xi = xf[i][t];
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(wi), "+v"(xi) : "n"(NLD - 1 - (i * NT + t)) : "memory");
        if constexpr (i == 0 && t == 0) K2_STAMP(2);
        if constexpr (i == NCH - 1 && t == NT - 1) K2_STAMP(3);
        acc[t & 1] = ACT::mfma(Q::frag(wi, t), xi, acc[t & 1]);
        accx[t & 1] = ACT::mfma(ones, xi, accx[t & 1]);                   // row sums of x on the matrix pipe: D[.][b] = sum_k x[b,k]
    });
    K2_STAMP(4);
    // ---- meet: the NW k-partials of the tile (as dq_h_body.inc, one row tile) ----------------------------------------------------------
    {
        float *p = reinterpret_cast<float *>(myreg);
        const f32x4_t a = acc[0] + acc[1];
        p[lane] = a[0]; p[64 + lane] = a[1]; p[128 + lane] = a[2]; p[192 + lane] = a[3];
        if (lane < 16) p[256 + lane] = accx[0][0] + accx[1][0];
    }
    K2_STAMP(5);
    __syncthreads();
    K2_STAMP(6);
    asm volatile("" : "+v"(e_sc), "+v"(e_zr), "+v"(e_bi));
#pragma unroll 1
    for (int u = wave; u < 4; u += NW) {                                 // role u: batch rows 4u .. 4u + 3 x the 16 weight rows
        const int b = 4 * u + (lane >> 4), wr = lane & 15;
        const int src = (wr & 3) * 64 + b + 16 * (wr >> 2);
        float a = 0.f, xsum = 0.f;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const float *p = reinterpret_cast<const float *>(smem + v * (1024 + 64));
            a += p[src];
            xsum += p[256 + b];
        }
        const int64_t row = (int64_t)rt0 * 16 + wr;
        if (b < e.bs) {
            const float alpha = e.qfn == QUIPAMD_QFN_B ? e_sc * e.two_over_maxq : e_sc;
            const float c0 = e.qfn == QUIPAMD_QFN_B ? Q::OFF + 0.5f * (float)e.maxq : Q::OFF + e_zr;
            const float val = alpha * (a - c0 * xsum) + e_bi;
            const int64_t o = (int64_t)b * e.m + row;
            if (e.y_f32) ((float *)e.y)[o] = e.accumulate ? ((float *)e.y)[o] + val : val;
            else ((uint16_t *)e.y)[o] = e.y_f16 ? f32_to_f16_bits(val) : f32_to_bf16_bits(val);
        }
    }
    K2_STAMP(7);
}

template <int BITS, class ACT, int NW, int NCH>
int launch_hr(const K2Args &A, hipStream_t s)
{
    static_assert(NW == 4, "four reducer roles, one per wave");
    auto kern = dq_hr_kernel<BITS, ACT, NW, NCH>;
    kern<<<dim3((unsigned)(A.e.m / 16)), 64 * NW, NW * (1024 + 64), s>>>(A);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm(hr)");
    return QUIPAMD_OK;
}

template <int BITS, class ACT, int RT, int NW, int NCH, bool HALF, bool EXACT>
int launch_h2(const K2Args &A, hipStream_t s)
{
    typedef DeqT<BITS, ACT> Q;
    constexpr size_t lds = (size_t)NW * NCH * (HALF ? 8 : 16) * Q::KC * 2;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = dq_h_kernel<BITS, ACT, RT, NW, NCH, HALF, EXACT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm: cannot raise dynamic LDS to %zu", lds);
    kern<<<dim3((unsigned)(A.e.m / 16 / RT)), 64 * NW, lds, s>>>(A);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm(h)");
    return QUIPAMD_OK;
}
template <int BITS, class ACT, int RT, int NW, int NCH, bool HALF, bool EXACT>
int launch_h2g(const K2GArgs &G, int ngroups, hipStream_t s)
{
    typedef DeqT<BITS, ACT> Q;
    constexpr size_t lds = (size_t)NW * NCH * (HALF ? 8 : 16) * Q::KC * 2;
    auto kern = dq_hg_kernel<BITS, ACT, RT, NW, NCH, HALF, EXACT>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm_grouped: cannot raise dynamic LDS to %zu", lds);
    kern<<<dim3((unsigned)(G.g[0].e.m / 16 / RT), (unsigned)ngroups), 64 * NW, lds, s>>>(G);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm_grouped(h)");
    return QUIPAMD_OK;
}
template <int BITS, class ACT, int RT, int NW, int NCH>
int launch_hg(const K2GArgs &G, int ngroups, hipStream_t s)
{
    const K2Args &A = G.g[0];
    const bool exact = A.d / (512 / BITS) == NW * NCH;
    if (A.e.bs <= 8)
        return exact ? launch_h2g<BITS, ACT, RT, NW, NCH, true, true>(G, ngroups, s) : launch_h2g<BITS, ACT, RT, NW, NCH, true, false>(G, ngroups, s);
    return exact ? launch_h2g<BITS, ACT, RT, NW, NCH, false, true>(G, ngroups, s) : launch_h2g<BITS, ACT, RT, NW, NCH, false, false>(G, ngroups, s);
}

// requires (m / 16) % RT == 0, bs <= 16, d / KC <= NW * NCH
template <int BITS, class ACT, int RT, int NW, int NCH>
int launch_h(const K2Args &A, hipStream_t s)
{
    const bool exact = A.d / (512 / BITS) == NW * NCH;
    if (A.e.bs <= 8)
        return exact ? launch_h2<BITS, ACT, RT, NW, NCH, true, true>(A, s) : launch_h2<BITS, ACT, RT, NW, NCH, true, false>(A, s);
    return exact ? launch_h2<BITS, ACT, RT, NW, NCH, false, true>(A, s) : launch_h2<BITS, ACT, RT, NW, NCH, false, false>(A, s);
}

// =====================================================================================================================
// dq_s_kernel: bs <= 16, weight stream.  Workgroup = NW * KSP compute waves + TWO loader waves (weights; x).
// Compute wave (h, w) <-> row tile blockIdx.x * NW + w, the stages s = h (mod KSP) of K: one wave per row tile is a
// serial chain of ~100 instructions per KiB of weights at ~4.5 cycles each (measured, profiles/r02c_k2probe_timeline.log:
// the loaders never wait for data, the compute waves are the bottleneck), so K is split over KSP waves per tile whose
// partials meet in LDS once, at the end.  Everything a compute wave touches comes out of LDS:
// the loader waves DMA, per stage of 256 k, the NW * TPS weight tiles (1 KiB each, lane-linear = the STREAM tile as it
// is) and the 8 KiB slab of x into one slot of a D-deep ring, D-1 stages ahead, non-temporal for the weights.
// The compute waves therefore have NO vector-memory queue at all (only ds_reads, which hipcc counts exactly), and the
// loaders nothing but their DMA queue with one constant counted wait -- the engine of cdna_hip_programming.md 5.6.
// Why not weights as register loads in the compute waves: hipcc drains vmcnt at the loop header for loop-carried loads
// (observed: waits 0,3,2,1 over a 4-deep register ring -- a full HBM round trip every fourth stage), and asm loads get
// their destination registers copied before the data has landed (observed: wrong results, memory faults;
// profiles/r02a_k2lab.log).  Per-wave vmcnt + role split keeps every wait exact with no inline-asm loads.
// One barrier per stage:  loaders: wait stage s landed -> barrier(s) -> DMA stage s+D-1 into the slot of stage s-1
//                         compute: barrier(s) -> weight tile + fragments of stage s -> 8 x (dequant, MFMA).
// Past the end the loaders repeat the last stage (clamped): every counted wait stays a constant.
// x is ingested once per workgroup: 8 KiB of x per NW * TPS KiB of weights.
// =====================================================================================================================
// (Round 5, negative result: MORE weight-loader waves -- the change that took dq_mb_kernel from 998 to 1323 TF -- do nothing here:
//  28672 x 7168 bs 16 15.1-15.4 us with 2 or 4 weight loaders against 15.4 with one, profiles/r05m_k2lab_s_loaders.txt.  This kernel's ring
//  is D deep with counted waits; the compute waves are its bound, as the round-2 probe said.)
template <int BITS, class ACT, int NW, int KSP, int SPW, int D>
__global__ __launch_bounds__(64 * (NW * KSP + 2)) void dq_s_kernel(K2Args A, uint32_t ntile)
{
#include "dq_s_body.inc"
}

// up to three problems of ONE shape in one launch (blockIdx.y picks), each with its own x: the weight-stream kernel for the grouped GEMMs of
// a 5..16-row decode step (round 6).  The same body text; the loads are LDS-DMA builtins the compiler sees, so the copy of the picked
// problem's arguments out of the kernarg segment moves nothing that matters.
template <int BITS, class ACT, int NW, int KSP, int SPW, int D>
__global__ __launch_bounds__(64 * (NW * KSP + 2)) void dq_sg_kernel(K2GArgs G, uint32_t ntile)
{
    const K2Args A = G.g[blockIdx.y];
#include "dq_s_body.inc"
}

template <int BITS, class ACT, int NW, int KSP, int SPW, int D>
int launch_s(const K2Args &A, hipStream_t s)
{
    constexpr int TPS = 256 / (512 / BITS);
    constexpr size_t lds = (size_t)D * KSP * SPW * (16 * 256 * 2 + NW * TPS * 1024);
    static_assert(lds <= 160 * 1024, "LDS budget");
    QA_REQUIRE(A.e.m * A.d * BITS / 8 < ((int64_t)1 << 32), QUIPAMD_ERR_SHAPE, "dequant_gemm(s): packed weights >= 4 GiB");
    auto kern = dq_s_kernel<BITS, ACT, NW, KSP, SPW, D>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm: cannot raise dynamic LDS to %zu", lds);
    const uint32_t ntile = (uint32_t)(A.e.m / 16);
    kern<<<dim3((ntile + NW - 1) / NW), 64 * (NW * KSP + 2), lds, s>>>(A, ntile);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm(s)");
    return QUIPAMD_OK;
}

template <int BITS, class ACT, int NW, int KSP, int SPW, int D>
int launch_sg(const K2GArgs &G, int ngroups, hipStream_t s)
{
    constexpr int TPS = 256 / (512 / BITS);
    constexpr size_t lds = (size_t)D * KSP * SPW * (16 * 256 * 2 + NW * TPS * 1024);
    static_assert(lds <= 160 * 1024, "LDS budget");
    const K2Args &A = G.g[0];
    QA_REQUIRE(A.e.m * A.d * BITS / 8 < ((int64_t)1 << 32), QUIPAMD_ERR_SHAPE, "dequant_gemm_grouped(s): packed weights >= 4 GiB");
    auto kern = dq_sg_kernel<BITS, ACT, NW, KSP, SPW, D>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm_grouped: cannot raise dynamic LDS to %zu", lds);
    const uint32_t ntile = (uint32_t)(A.e.m / 16);
    kern<<<dim3((ntile + NW - 1) / NW, (unsigned)ngroups), 64 * (NW * KSP + 2), lds, s>>>(G, ntile);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm_grouped(s)");
    return QUIPAMD_OK;
}

// =====================================================================================================================
// dq_mb_kernel: bs > 16.  Workgroup = WR x WB compute waves + NL loader waves; compute wave (wr, wb) owns RTw row
// tiles x BTw batch tiles: workgroup tile = (ROWS = WR*RTw*16 rows) x (NB = WB*BTw*16 batch rows), all of K.
// A stage = 256 k: the NB x 512 B panel of x (NB/8 * 4 DMA instructions of 8 rows x 128 B, XOR-swizzled through the
// source address) plus the ROWS/16 * TPG weight tiles of the workgroup, DMA'd by the loader waves into one of TWO LDS
// buffers while the compute waves work on the other: the loaders' wait is a plain vmcnt(0), the compute waves touch
// nothing but LDS (same reasoning as dq_s_kernel), ONE barrier per 256 k.
// Per 32-k MFMA step and compute wave: BTw fragment reads, RTw dequantised A fragments (8 VALU each), RTw*BTw MFMAs:
// every A fragment feeds BTw MFMAs, every B fragment RTw.  The sums S_1, S_off of batch tile bt (dq_common.h) ride on the
// matrix pipe of the wave with wr = bt % WR (profiles/r02e: VALU and MFMA issue are equally loaded here, ~45 % each).
//     loaders: wait stage s landed -> barrier(s) -> DMA stage s+1 into the buffer of stage s-1
//     compute: barrier(s) -> 8 steps on buffer s & 1
// =====================================================================================================================
// T32 (round 5): the SAME workgroup tile, stages and loaders with the products on v_mfma_f32_32x32x16 (a bare issue loop of it reaches 2382 TFLOP/s
// against 2075 for 16x16x32: cdna_hip_programming.md).  The STREAM tile is the 16-row A fragment of 16x16x32 -- lane (row j, k-group g) holds the codes
// of k = 32 t + 8 g + 0..7 for its 8 steps t -- and the 32-row operand wants lane L = (row L % 32, k-group L / 32) with k = 16 s + 8 (L / 32) + 0..7:
// that is old lane j + 16 (2 (s & 1) + L / 32) of tile (L % 32) / 16, step t = s / 2.  The tiles pass through LDS here, so the permutation is just the
// address of two ds_read_b128 per tile PAIR (wE: even s, wO: odd s); the dequantiser, the loaders and the LDS image are unchanged.  x: lane L reads
// batch row L % 32, 16-byte chunk 2 (s % 4) + L / 32 of column block s / 4 (same swizzle).  D: register r of lane L = weight row 8 (r / 4) + 4 (L / 32) +
// r % 4, batch row L % 32 -- four consecutive weight rows per register quad, which is what epilogue_store takes.  The row sums S_1 / S_off stay on
// 16x16x32 (a 32x32 product for a 1 x 32 result would cost a third of the main loop): the owner wave of a 16-row batch tile reads that tile's
// fragments in the old layout.
// RESULT (profiles/r05x_k2lab_mb32.txt, r05y_*, r05z_*): correct at every shape tried (bf16 / fp16, ragged m and batch, 2 and 4 bit) and SLOWER --
// 1100-1130 TFLOP/s against 1330-1380 at 28672 x 7168 bs 256, 1040 / 1310 at 4096^2 x 2048, 1195 / 1490 at 8192^2 x 1024 -- with the x reads
// conflict-free (the key below: 8 -> 4 LDS cycles per ds_read_b128, no change in time) and requested two steps ahead (pinned: +1 %).  A bare issue
// loop on the same GPU (scripts/mfma_lab.hip, profiles/r05y_mfma_lab.txt) agrees: both shapes reach 2.46-2.49 PFLOP/s on constant operands, on random
// ones 16x16x32 gives 2.2 and 32x32x16 1.9 -- the 32-row shape is the slower instruction on real data.  Kept as configuration 47 (forced only).
template <int BITS, class ACT, int WR, int WB, int RTw, int BTw, int NL, bool T32 = false>
__global__ __launch_bounds__(64 * (WR * WB + NL)) void dq_mb_kernel(K2Args A, uint32_t nrb, uint32_t nby)
{
    // Lab switches of this kernel (scripts/k2lab.hip builds with -DK2_MB_...; never defined in the library; the results are WRONG by construction,
    // only the time is read -- profiles/r05y_*, r05z_*): NOSUMS drops the row-sum reads and products, NODEQ feeds the packed dwords to the MFMAs as
    // they are, NOX re-uses the x fragments of a stage's first column block (step), NODMA brings in stage 0 only, UNIFORM takes the one-offset
    // dequantiser for 2 bits as well (16 VALU per packed dword, no S_off).
#ifdef K2_MB_UNIFORM
    typedef DeqUni<BITS, ACT> Q;
#else
    typedef DeqSel<BITS, ACT> Q;                                      // 2 bits: multi-exponent dequantisation (dq_common.h)
#endif
    constexpr int KC = Q::KC, NTT = Q::NT;
    constexpr int TPG = 256 / KC;                                      // weight tiles per stage and row tile
    constexpr int NCW = WR * WB, NB = WB * BTw * 16, NRT = WR * RTw;   // compute waves, batch rows, row tiles per workgroup
    constexpr int NXP = NB / 8 * 4, NWT = NRT * TPG, NOPS = NXP + NWT; // x pieces, weight tiles, DMA instructions per stage
    constexpr int XBYTES = NB * 512, SB = XBYTES + NWT * 1024;         // stage bytes
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [2] stages {x panel [4 column blocks][NB/8 row blocks][1 KiB], weight tiles}
    const EpiArgs &e = A.e;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // XCD-aware order: the nby batch blocks of one row block run on one XCD (block id % 8), back to back
    const uint32_t id = blockIdx.x, xcd = id & 7, within = id >> 3;
    const uint32_t by = within % nby, rbk = (within / nby) * 8 + xcd;
    if (rbk >= nrb) return;                                            // whole workgroup
    const uint32_t ns = (uint32_t)(A.d / 256), nkc = ns * TPG;
    const uint32_t ntile = (uint32_t)(e.m / 16);
    const int64_t brow0 = (int64_t)by * NB;
    const uint32_t rowbytes = (uint32_t)A.d * 2u;

    if (wave >= NCW) {
        // ---- loader waves: DMA instruction o of a stage: o < NXP: x piece (column block o / (NB/8), row block o % (NB/8)); else weight tile o - NXP
        const int lw = wave - NCW;
        const int64_t rem = (e.bs - brow0) * (int64_t)rowbytes;
        __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.x + brow0 * A.d), 0, (int)rem, 0x00020000);
        __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)A.qw, 0, (int)((uint64_t)ntile * nkc * 1024u), 0x00020000);
        const uint32_t vx = (lane >> 3) * rowbytes + ((uint32_t)((lane & 7) ^ (lane >> 3)) << 4), vw = (uint32_t)lane * 16u;
        // T32: ds_read_b128 is served in four 16-lane groups ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) and a group of the 32-row operand
        // spans FOUR row blocks: with the key (row & 7) alone rows r and r + 16 of a group share a 16-byte slot (2-way conflict on every x read).  Key (row & 7) ^ ((row >> 4) & 1): the odd 16-row halves are written one chunk over.
        const uint32_t vx1 = (lane >> 3) * rowbytes + ((uint32_t)((lane & 7) ^ (lane >> 3) ^ 1) << 4);
        auto issue = [&](uint32_t st) {
            char *dst = smem + (st & 1) * SB;
#pragma unroll
            for (int o = lw; o < NOPS; o += NL) {
#ifdef K2_MB_NODMA
                if (st > 0) continue;
#endif
                if (o < NXP) {
                    const int cb = o / (NB / 8), rb = o % (NB / 8);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void2_t *)(dst + o * 1024), 16,
                                                             ((T32 && ((rb >> 1) & 1)) ? vx1 : vx) + (uint32_t)rb * 8u * rowbytes, st * 512u + cb * 128, 0, 0);
                } else {
                    const int i = o - NXP;
                    const uint32_t r = rbk * NRT + i / TPG, rc = r < ntile ? r : ntile - 1;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void2_t *)(dst + XBYTES + i * 1024), 16, vw,
                                                             (rc * nkc + st * TPG + i % TPG) * 1024u, 0, 0);
                }
            }
        };
        issue(0);
#pragma unroll 1
        for (uint32_t s = 0; s < ns; ++s) {
            wait_vm<0>();                                              // stage s has landed
            __builtin_amdgcn_s_barrier();
            if (s + 1 < ns) issue(s + 1);
        }
        __builtin_amdgcn_s_barrier();                                  // "every fragment read retired"
        __builtin_amdgcn_s_barrier();                                  // the row-sum exchange
        return;
    }

    // ---- compute waves -----------------------------------------------------------------------------------------------------
    const int wr = wave / WB, wb = wave - wr * WB;
    const int j = lane & 15, g = lane >> 4;
    const uint32_t rt0 = rbk * NRT + wr * RTw;                         // first row tile of this wave
    // fragment of batch tile bt, column block cb, step half sh: row R = (wb*BTw + bt)*16 + j -> row block R>>3, in-block row j&7
    const uint32_t rdA = (uint32_t)(wb * BTw) * 2048 + (j >> 3) * 1024 + (j & 7) * 128;
    const uint32_t rd0 = rdA + (((0 + g) ^ (j & 7)) << 4);
    const uint32_t rd1 = rdA + (((4 + g) ^ (j & 7)) << 4);
    const uint32_t wof = XBYTES + (uint32_t)(wr * RTw) * TPG * 1024 + (uint32_t)lane * 16;

    // S_1 = sum_k x and S_off = sum_k OFF_k x of batch tile bt are kept, on the matrix pipe, by the wave with wr = bt % WR
    constexpr int NSB = (BTw + WR - 1) / WR;
    if constexpr (T32) {
        static_assert(RTw % 2 == 0 && BTw % 2 == 0, "32 x 32 tiles: pairs of row tiles and of batch tiles");
        constexpr int NP = RTw / 2, NB2 = BTw / 2;
        const int l32 = lane & 31, kg = lane >> 5;
        // weights: old lane (l32 & 15) + 16 kg (even steps) / + 16 (2 + kg) (odd steps) of tile 2 p + (l32 >> 4)
        const uint32_t wofE = XBYTES + (uint32_t)(wr * RTw + (l32 >> 4)) * TPG * 1024 + (uint32_t)((l32 & 15) + 16 * kg) * 16;
        // x: batch row (wb BTw 16 + 32 b + l32) -> row block, in-block row; chunk 2 q + kg of a column block for step q of it
        const uint32_t rdB = (uint32_t)(wb * BTw) * 2048 + (l32 >> 3) * 1024 + (l32 & 7) * 128;
        const uint32_t rd0s = rdA + (((0 + g) ^ (j & 7) ^ 1) << 4), rd1s = rdA + (((4 + g) ^ (j & 7) ^ 1) << 4);   // odd 16-row tiles
        static_assert(BTw % 2 == 0, "tile parity = local tile parity");
        uint32_t xo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) xo[q] = rdB + (uint32_t)(((2 * q + kg) ^ (l32 & 7) ^ (l32 >> 4)) << 4);       // key (row & 7) ^ ((row >> 4) & 1)
        f32x16_t acc[NP][NB2];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int b = 0; b < NB2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[p][b][i] = 0.f;
        f32x4_t sum1[NSB];
#pragma unroll
        for (int q = 0; q < NSB; ++q) sum1[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const typename Q::Consts qc = Q::make_consts();
        const u32x4 ones = {opaque(ACT::ONES), opaque(ACT::ONES), opaque(ACT::ONES), opaque(ACT::ONES)};
        const u32x4 offs[2] = {Q::off_frag(0, qc), Q::off_frag(1, qc)};
        const u32x4 zero4 = {0u, 0u, 0u, 0u};                          // both row sums from one product: see the 16x16x32 path below
        const u32x4 sfr[2] = {j == 0 ? ones : (j == 1 ? offs[0] : zero4), j == 0 ? ones : (j == 1 ? offs[1] : zero4)};
        static_for<WR>([&](auto KK) {
            constexpr int kk = decltype(KK)::value;
            if (wr != kk) return;
#pragma unroll 1
            for (uint32_t s = 0; s < ns; ++s) {
                __builtin_amdgcn_s_barrier();
                const char *sl = smem + (s & 1) * SB;
                u32x4 wE[NP][TPG], wO[NP][TPG];
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int t = 0; t < TPG; ++t) {
                        wE[p][t] = *reinterpret_cast<const u32x4 *>(sl + wofE + (2 * p * TPG + t) * 1024);
                        wO[p][t] = *reinterpret_cast<const u32x4 *>(sl + wofE + (2 * p * TPG + t) * 1024 + 512);
                    }
                // The x fragments of step u + 2 are requested in front of the products of step u and pinned there (sched_barrier: VALU / SALU may
                // cross, DS reads and MFMAs may not): left to itself hipcc reloads the very registers a step has just freed and waits for them two products
                // later -- an LDS round trip per step in the open.
                u32x4 xr[3][NB2], sx[NSB][2];
                auto ldx = [&](int u, u32x4 (&dst)[NB2]) {
#pragma unroll
                    for (int b = 0; b < NB2; ++b) dst[b] = *reinterpret_cast<const u32x4 *>(sl + (u >> 2) * (NB * 128) + b * 4096 + xo[u & 3]);
                };
                ldx(0, xr[0]);
                ldx(1, xr[1]);
                static_for<16>([&](auto U) {
                    constexpr int u = decltype(U)::value, g32 = u >> 1, tile = g32 / NTT, tstep = g32 % NTT;   // 16-k step of the stage; its 32-k step
#ifdef K2_MB_NOX
                    constexpr int xi = u & 1;
#else
                    constexpr int xi = u % 3;
                    if constexpr (u + 2 < 16) ldx(u + 2, xr[(u + 2) % 3]);
#endif
#ifndef K2_MB_NOSUMS
                    if constexpr ((u & 3) == 0) {                           // the row sums of this wave's own 16-row batch tile(s): old fragment layout
#pragma unroll
                        for (int q = 0; q < NSB; ++q)
                            if (kk + q * WR < BTw) {
                                // (16-row batch tile bt = kk + q WR: its rows carry the key bit bt & 1)
                                sx[q][0] = *reinterpret_cast<const u32x4 *>(sl + (u >> 2) * (NB * 128) + (kk + q * WR) * 2048 + (((kk + q * WR) & 1) ? rd0s : rd0));
                                sx[q][1] = *reinterpret_cast<const u32x4 *>(sl + (u >> 2) * (NB * 128) + (kk + q * WR) * 2048 + (((kk + q * WR) & 1) ? rd1s : rd1));
                            }
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0x6);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
#ifdef K2_MB_NODEQ
                        const u32x4 a = (u & 1) ? wO[p][tile] : wE[p][tile];
#else
                        const u32x4 a = Q::frag((u & 1) ? wO[p][tile] : wE[p][tile], tstep, qc);
#endif
#pragma unroll
                        for (int b = 0; b < NB2; ++b) acc[p][b] = ACT::mfma32(a, xr[xi][b], acc[p][b]);
                    }
#ifndef K2_MB_NOSUMS
                    if constexpr ((u & 3) == 3) {
#pragma unroll
                        for (int q = 0; q < NSB; ++q)
                            if (kk + q * WR < BTw) {
                                sum1[q] = ACT::mfma(sfr[0], sx[q][0], sum1[q]);
                                sum1[q] = ACT::mfma(sfr[1], sx[q][1], sum1[q]);
                            }
                    }
#endif
                });
            }
        });
        __builtin_amdgcn_s_barrier();                                  // every fragment read retired: LDS is free
        float *xsh = reinterpret_cast<float *>(smem), *xso = xsh + NB;
#pragma unroll
        for (int q = 0; q < NSB; ++q) {
            const int bt = wr + q * WR;
            if (bt < BTw && lane < 16) {
                xsh[(wb * BTw + bt) * 16 + lane] = sum1[q][0];
                if constexpr (!Q::UNIFORM) xso[(wb * BTw + bt) * 16 + lane] = sum1[q][1];
            }
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB2; ++b) {
            const int brow = wb * BTw * 16 + b * 32 + l32;               // batch row inside the workgroup's NB
            const float xsum = xsh[brow], xoff = Q::UNIFORM ? 0.f : xso[brow];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const uint32_t rt = rt0 + 2 * p + (rg >> 1);
                    if (rt >= ntile) continue;
                    const int64_t r0 = (int64_t)rt * 16 + 8 * (rg & 1) + 4 * kg;
                    const EpiRow epi = load_epi(e, r0);
                    f32x4_t a = {acc[p][b][4 * rg] - xoff, acc[p][b][4 * rg + 1] - xoff, acc[p][b][4 * rg + 2] - xoff, acc[p][b][4 * rg + 3] - xoff};
                    epilogue_store(e, epi, Q::OFF, a, xsum, brow0 + brow, r0);
                }
        }
        return;
    }

    f32x4_t acc[RTw][BTw];
#pragma unroll
    for (int bt = 0; bt < BTw; ++bt)
#pragma unroll
        for (int r = 0; r < RTw; ++r) acc[r][bt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t sum1[NSB];
#pragma unroll
    for (int q = 0; q < NSB; ++q) sum1[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const typename Q::Consts qc = Q::make_consts();
    const u32x4 ones = {opaque(ACT::ONES), opaque(ACT::ONES), opaque(ACT::ONES), opaque(ACT::ONES)};
    const u32x4 offs[2] = {Q::off_frag(0, qc), Q::off_frag(1, qc)};
    // ONE product per 32-k step gives both row sums (round 5): A row 0 = ones, A row 1 = the OFF_k pattern, the other rows zero -- D[0][b] = S_1,
    // D[1][b] = S_off (registers 0 and 1 of lanes 0..15).  As two products (ones, then offs: every row of D the same sum) the sums were 16 of a
    // wave's 144 MFMAs per stage and 4.8 of 82 us (profiles/r05z_k2lab_mb16_ablations.txt).
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const u32x4 sfr[2] = {j == 0 ? ones : (j == 1 ? offs[0] : zero4), j == 0 ? ones : (j == 1 ? offs[1] : zero4)};
    // The stage loop exists WR times, once per value of wr: which batch tile's sums a wave keeps is then a compile-time
    // constant inside the loop.  Tested per MFMA step instead (`if (wr == kk)` in the body) it was four scalar test-and-branch
    // blocks per step, 116 SALU instructions and 32 taken / not-taken branches per 256-k stage.
    static_for<WR>([&](auto KK) {
        constexpr int kk = decltype(KK)::value;
        if (wr != kk) return;
#pragma unroll 1
        for (uint32_t s = 0; s < ns; ++s) {
            __builtin_amdgcn_s_barrier();
            const char *sl = smem + (s & 1) * SB;
            u32x4 wc[RTw][TPG];
#pragma unroll
            for (int r = 0; r < RTw; ++r)
#pragma unroll
                for (int t = 0; t < TPG; ++t) wc[r][t] = *reinterpret_cast<const u32x4 *>(sl + wof + (r * TPG + t) * 1024);
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                u32x4 xf[BTw][2];
#pragma unroll
                for (int bt = 0; bt < BTw; ++bt) {
#ifdef K2_MB_NOX
                    xf[bt][0] = *reinterpret_cast<const u32x4 *>(sl + bt * 2048 + rd0);
                    xf[bt][1] = *reinterpret_cast<const u32x4 *>(sl + bt * 2048 + rd1);
#else
                    xf[bt][0] = *reinterpret_cast<const u32x4 *>(sl + cb * (NB * 128) + bt * 2048 + rd0);
                    xf[bt][1] = *reinterpret_cast<const u32x4 *>(sl + cb * (NB * 128) + bt * 2048 + rd1);
#endif
                }
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    const int gstep = 2 * cb + sh, tile = gstep / NTT, tstep = gstep % NTT;   // MFMA step inside the stage
#pragma unroll
                    for (int r = 0; r < RTw; ++r) {
#ifdef K2_MB_NODEQ
                        const u32x4 a = wc[r][tile];
#else
                        const u32x4 a = Q::frag(wc[r][tile], tstep, qc);
#endif
#pragma unroll
                        for (int bt = 0; bt < BTw; ++bt) acc[r][bt] = ACT::mfma(a, xf[bt][sh], acc[r][bt]);
                    }
#ifndef K2_MB_NOSUMS
#pragma unroll
                    for (int q = 0; q < NSB; ++q)
                        if (kk + q * WR < BTw) {
                            sum1[q] = ACT::mfma(sfr[sh], xf[kk + q * WR][sh], sum1[q]);
                        }
#endif
                }
            }
        }
    });
    __builtin_amdgcn_s_barrier();                                      // every fragment read retired: LDS is free

    // row sums of x: owner waves publish [wb][bt][16], everybody reads
    float *xsh = reinterpret_cast<float *>(smem), *xso = xsh + NB;     // [NB] S_1, [NB] S_off
#pragma unroll
    for (int q = 0; q < NSB; ++q) {
        const int bt = wr + q * WR;
        if (bt < BTw && lane < 16) {
            xsh[(wb * BTw + bt) * 16 + lane] = sum1[q][0];
            if constexpr (!Q::UNIFORM) xso[(wb * BTw + bt) * 16 + lane] = sum1[q][1];
        }
    }
    __syncthreads();
    float xsum[BTw], xoff[BTw];
#pragma unroll
    for (int bt = 0; bt < BTw; ++bt) {
        xsum[bt] = xsh[(wb * BTw + bt) * 16 + j];
        xoff[bt] = Q::UNIFORM ? 0.f : xso[(wb * BTw + bt) * 16 + j];
    }
#pragma unroll
    for (int r = 0; r < RTw; ++r) {
        if (rt0 + r >= ntile) continue;
        const int64_t r0 = (int64_t)(rt0 + r) * 16 + 4 * g;
        const EpiRow epi = load_epi(e, r0);
#pragma unroll
        for (int bt = 0; bt < BTw; ++bt) {
            f32x4_t a = acc[r][bt];
            if constexpr (!Q::UNIFORM) { a[0] -= xoff[bt]; a[1] -= xoff[bt]; a[2] -= xoff[bt]; a[3] -= xoff[bt]; }
            epilogue_store(e, epi, Q::OFF, a, xsum[bt], brow0 + (wb * BTw + bt) * 16 + j, r0);
        }
    }
}

template <int BITS, class ACT, int WR, int WB, int RTw, int BTw, int NL, bool T32 = false>
int launch_mb2(const K2Args &A, hipStream_t s)
{
    constexpr int NB = WB * BTw * 16, ROWS = WR * RTw * 16, TPG = 256 / (512 / BITS);
    constexpr size_t lds = (size_t)2 * (NB * 512 + ROWS / 16 * TPG * 1024);
    static_assert(lds <= 160 * 1024 && lds >= (size_t)NB * 8, "LDS budget");
    QA_REQUIRE(A.e.m * A.d * BITS / 8 < ((int64_t)1 << 32), QUIPAMD_ERR_SHAPE, "dequant_gemm(mb): packed weights >= 4 GiB");
    auto kern = dq_mb_kernel<BITS, ACT, WR, WB, RTw, BTw, NL, T32>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm: cannot raise dynamic LDS to %zu", lds);
    const uint32_t nrb = (uint32_t)((A.e.m + ROWS - 1) / ROWS), nby = (uint32_t)((A.e.bs + NB - 1) / NB);
    const uint32_t nrb8 = (nrb + 7) / 8 * 8;                           // padded so that id -> (xcd, within) covers every row block
    const uint64_t nblk = (uint64_t)nrb8 * nby;
    QA_REQUIRE(nblk < (1ull << 31), QUIPAMD_ERR_SHAPE, "dequant_gemm: grid too large");
    kern<<<dim3((unsigned)nblk), 64 * (WR * WB + NL), lds, s>>>(A, nrb, nby);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm(mb)");
    return QUIPAMD_OK;
}

}   // namespace
