// k2lab.hip -- standalone lab for the K2 kernels (not part of the library): correctness against a host fp64 reference of
// the reference formula (quant.py:13-14, 222-233) and per-launch time inside a hipGraph, cold (weight ring > 256 MiB) and warm.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I quip_amd/csrc scripts/k2lab.hip -o build/k2lab
// run:   build/k2lab <mode> [m d bs bits act]     mode: null | h | s | mb | old
#include "../quip_amd/csrc/capi.hip"
#include "../quip_amd/csrc/dqgemm.hip"
#ifdef K2_PROBE
// probe build (-DK2_PROBE): per-wave s_memtime stamps (dq_h_kernel) and per-wave time spent in waits / barriers / issue (dq_s_kernel)
__device__ unsigned long long *g_k2_probe = nullptr;
#define K2_STAMP(i) do { if (g_k2_probe && (threadIdx.x & 63) == 0) g_k2_probe[((size_t)blockIdx.x * 32 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#define K2_STAMP_FLUSH() do { } while (0)
#define K2_ACC_DECL unsigned long long k2_acc[4] = {0, 0, 0, 0}; const unsigned long long k2_t00 = __builtin_readcyclecounter()
#define K2_ACC(slot, stmt) do { const unsigned long long t0_ = __builtin_readcyclecounter(); stmt; k2_acc[slot] += __builtin_readcyclecounter() - t0_; } while (0)
#define K2_ACC_FLUSH() do { if (g_k2_probe && (threadIdx.x & 63) == 0) { unsigned long long *q_ = g_k2_probe + ((size_t)blockIdx.x * 32 + (threadIdx.x >> 6)) * 8; \
        q_[0] = k2_acc[0]; q_[1] = k2_acc[1]; q_[2] = k2_acc[2]; q_[3] = __builtin_readcyclecounter() - k2_t00; } } while (0)
#endif
#include "../quip_amd/csrc/dqgemm_v2.h"
int k2v2_launch(const K2Call &, void *) { return K2V2_NOT_TAKEN; }     // the lab's "old" rows measure the round-1 kernels alone
int k2v2_launch_grouped(const K2Call *, int, void *) { return K2V2_NOT_TAKEN; }
namespace {
// =====================================================================================================================
// dq_hl_kernel (round 5, LAB ONLY -- a negative result, profiles/r05h_k2lab_hl.txt: 5.0 us cold / 4.07 warm against 4.68 / 3.60 for dq_h_kernel
// with the same chunking): dq_h_kernel with the roles split by wave -- NW compute waves (x slabs by LDS-DMA, dequant, MFMA, meet) and NL
// LOADER waves that bring the workgroup's weight tiles from HBM into LDS (one DMA instruction per 1 KiB tile, lane-linear = the STREAM
// tile as it is), wait for them, meet the compute waves at one barrier and leave.
// Why: vector memory returns IN ORDER per wave.  In dq_h_kernel a wave's weight loads (HBM: ~2000 clocks cold) sit in the same queue as
// its sixteen x-slab DMAs (L2 hits): with the weights first nothing of x can return before HBM answers, with the weights last they leave
// ~1500 clocks late -- either way the HBM round trip and the ingest of x (128 KiB per CU: ~3300 clocks) happen one after the other, and
// that is the whole difference between the cold launch and the warm one (4.7 vs 3.6 us, profiles/r05g_k2lab_wfirst_ab.txt).  In another
// wave's queue the weights cost the compute waves nothing: x streams in from clock 0, the tiles arrive under it.
// EXACT shapes only (d / KC == NW * NCH), one row tile per workgroup; a compute wave's chunks are adjacent (NCH w + i).
// LDS: [NW][NCH] x slabs (as dq_h_kernel), then NW * NCH weight tiles of 1 KiB.
// =====================================================================================================================
template <int BITS, class ACT, int NW, int NCH, int NL, bool HALF>
__global__ __launch_bounds__(64 * (NW + NL)) void dq_hl_kernel(K2Args A)
{
    typedef DeqT<BITS, ACT> Q;
    constexpr int KC = Q::KC, NT = Q::NT;
    constexpr int ROWB = KC * 2, NCB = ROWB / 128, NI = 2 * NCB, XB = 16 * ROWB;
    constexpr int NDMA = HALF ? NI / 2 : NI;
    constexpr int NTILE = NW * NCH, TPL = NTILE / NL;
    static_assert(NTILE % NL == 0 && (NCH - 1) * NDMA < 64 && TPL < 64, "tiles per loader wave; vmcnt range");
    static_assert(1024 + 64 <= NCH * XB, "a wave parks its partials in its own slab region");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *wreg = smem + NW * NCH * XB;
    const EpiArgs &e = A.e;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t rt0 = blockIdx.x;
    if (wave >= NW) {
        // ---- loader wave: TPL adjacent tiles of this row tile, HBM -> LDS, non-temporal -------------------------------------------------
        const int lw = wave - NW;
        __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.qw + (uint64_t)rt0 * NTILE * 64), 0, NTILE * 1024, 0x00020000);
        const uint32_t vl = (uint32_t)lane * 16u;
#pragma unroll
        for (int t = 0; t < TPL; ++t)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_void2_t *)(wreg + (lw * TPL + t) * 1024), 16, vl, (uint32_t)(lw * TPL + t) * 1024u, 0, 2 /* nt */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                 // the tiles are in LDS: hand-over to the compute waves
        return;
    }
    const int j = lane & 15, g = lane >> 4;
    const uint32_t rowbytes = (uint32_t)A.d * 2u;
    char *myreg = smem + wave * (NCH * XB);

    float e_sc = 0.f, e_zr = 0.f, e_bi = 0.f;
    if (wave < 4) {
        const int64_t row = (int64_t)rt0 * 16 + (lane & 15);
        e_sc = e.qfn == QUIPAMD_QFN_B ? e.scale[0] : e.scale[row];
        if (e.qfn != QUIPAMD_QFN_B) e_zr = e.zero[row];
        if (e.bias) e_bi = e.bias[row];
    }

    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)A.x, 0, (int)(e.bs * (int64_t)rowbytes), 0x00020000);
    const uint32_t voff_lo = (lane >> 3) * rowbytes + ((uint32_t)((lane & 7) ^ (lane >> 3)) << 4);
    const uint32_t voff_hi = voff_lo + 8u * rowbytes;
    const uint32_t rd_base = lds_addr(myreg) + (j >> 3) * 1024 + (j & 7) * 128;
    const uint32_t rd0 = rd_base + (((0 + g) ^ (j & 7)) << 4);        // even MFMA steps
    const uint32_t rd1 = rd_base + (((4 + g) ^ (j & 7)) << 4);        // odd MFMA steps
    const uint32_t rdw = lds_addr(wreg) + (uint32_t)(NCH * wave) * 1024u + (uint32_t)lane * 16u;   // this wave's first weight tile

    // ---- request all of x -------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const uint32_t kc = (uint32_t)(NCH * wave + i);
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            if ((q & 1) && HALF) continue;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void2_t *)(myreg + i * XB + q * 1024), 16, (q & 1) ? voff_hi : voff_lo,
                                                     kc * ROWB + (q >> 1) * 128, 0, 0);
        }
    }
    f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, accx[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const u32x4 ones = {opaque(ACT::ONES), opaque(ACT::ONES), opaque(ACT::ONES), opaque(ACT::ONES)};
    static_for<NCH>([&](auto I) {
        constexpr int i = decltype(I)::value;
        wait_vm<(NCH - 1 - i) * NDMA>();                              // the slabs up to mine (the epilogue parameters are older still)
        if constexpr (i == 0) __builtin_amdgcn_s_barrier();           // ... and every loader wave has seen its tiles land
        u32x4 wt;
        lds_read16<i * 1024>(wt, rdw);
        u32x4 xf[NT];
        static_for<NT>([&](auto T) {
            constexpr int t = decltype(T)::value;
            lds_read16<i * XB + (t >> 1) * 2048>(xf[t], (t & 1) ? rd1 : rd0);
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wt)::"memory");
        wait_lgkm(xf);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t & 1] = ACT::mfma(Q::frag(wt, t), xf[t], acc[t & 1]);
            accx[t & 1] = ACT::mfma(ones, xf[t], accx[t & 1]);       // row sums of x on the matrix pipe
        }
    });

    // ---- meet: the NW k-partials (as dq_h_kernel, RT = 1) -------------------------------------------------------------------------
    {
        float *p = reinterpret_cast<float *>(myreg);
        const f32x4_t a = acc[0] + acc[1];
        p[lane] = a[0]; p[64 + lane] = a[1]; p[128 + lane] = a[2]; p[192 + lane] = a[3];
        if (lane < 16) p[256 + lane] = accx[0][0] + accx[1][0];
    }
    __syncthreads();                                                  // (the loader waves have left: the compute waves only)
    asm volatile("" : "+v"(e_sc), "+v"(e_zr), "+v"(e_bi));
    if (wave < 4) {
        const int q = wave;
        const int b = 4 * q + (lane >> 4), wr = lane & 15;
        const int src = (wr & 3) * 64 + b + 16 * (wr >> 2);
        float a = 0.f, xsum = 0.f;
#pragma unroll
        for (int v = 0; v < NW; ++v) {
            const float *p = reinterpret_cast<const float *>(smem + v * (NCH * XB));
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
}

template <int BITS, class ACT, int NW, int NCH, int NL>
int launch_hl(const K2Args &A, hipStream_t s)
{
    typedef DeqT<BITS, ACT> Q;
    constexpr size_t lds = (size_t)NW * NCH * 16 * Q::KC * 2 + (size_t)NW * NCH * 1024;
    static_assert(lds <= 160 * 1024 && NW >= 4, "LDS budget; four reducer waves");
    const bool half = A.e.bs <= 8;
    auto kern = half ? dq_hl_kernel<BITS, ACT, NW, NCH, NL, true> : dq_hl_kernel<BITS, ACT, NW, NCH, NL, false>;
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm: cannot raise dynamic LDS to %zu", lds);
    kern<<<dim3((unsigned)(A.e.m / 16)), 64 * (NW + NL), lds, s>>>(A);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm(hl)");
    return QUIPAMD_OK;
}

}   // namespace
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

template <int T> __global__ __launch_bounds__(T) void null_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

// oracle pack_stream (oracle/quip_oracle.py:203-229) restated for the lab
static void pack_stream_host(const std::vector<uint8_t> &codes, int bits, int64_t m, int64_t d, std::vector<uint32_t> &out)
{
    const int KC = 512 / bits, NT = KC / 32;
    const int64_t nkc = d / KC;
    out.assign((size_t)(m / 16) * nkc * 64 * 4, 0u);
    for (int64_t rt = 0; rt < m / 16; ++rt)
        for (int64_t kc = 0; kc < nkc; ++kc)
            for (int lane = 0; lane < 64; ++lane) {
                const int j = lane & 15, g = lane >> 4;
                uint32_t *w = &out[(((size_t)rt * nkc + kc) * 64 + lane) * 4];
                for (int t = 0; t < NT; ++t)
                    for (int e = 0; e < 8; ++e) {
                        const int u = bits == 2 ? t >> 1 : t;
                        const int i = bits == 2 ? 4 * (t & 1) + e / 2 : e / 2;
                        const int sh = bits * i + ((e & 1) ? 16 : 0);
                        const uint32_t c = codes[(size_t)(rt * 16 + j) * d + kc * KC + 32 * t + 8 * g + e];
                        w[u] |= c << sh;
                    }
            }
}

struct Problem {
    int64_t m, d, bs; int bits; bool f16;
    std::vector<uint8_t> codes; std::vector<uint16_t> hx; std::vector<float> xf;
    uint8_t *w = nullptr; uint16_t *x = nullptr, *y = nullptr; float *scale = nullptr; size_t wbytes; int nring;
    float hs = 0.05f;
    std::vector<int64_t> rows; std::vector<double> ref;       // sampled rows, ref[b][sample]
};

static void make_problem(Problem &P)
{
    const int64_t m = P.m, d = P.d, bs = P.bs;
    P.codes.resize((size_t)m * d);
    uint32_t s = 12345u;
    for (auto &c : P.codes) { s = s * 1664525u + 1013904223u; c = (uint8_t)((s >> 24) & ((1u << P.bits) - 1u)); }
    std::vector<uint32_t> packed;
    pack_stream_host(P.codes, P.bits, m, d, packed);
    P.wbytes = packed.size() * 4;
    P.nring = (int)std::max<size_t>(2, std::min<size_t>(96, ((size_t)400 << 20) / P.wbytes + 1));
    CK(hipMalloc(&P.w, P.wbytes * P.nring));
    for (int r = 0; r < P.nring; ++r) CK(hipMemcpy(P.w + (size_t)r * P.wbytes, packed.data(), P.wbytes, hipMemcpyHostToDevice));
    P.hx.resize((size_t)bs * d); P.xf.resize((size_t)bs * d);
    for (size_t i = 0; i < P.hx.size(); ++i) {
        s = s * 1664525u + 1013904223u; const float u1 = ((s >> 8) + 1) / 16777217.0f;
        s = s * 1664525u + 1013904223u; const float u2 = (s >> 8) / 16777216.0f;
        const float v = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
        P.hx[i] = P.f16 ? f2h(v) : f2bf(v);
        P.xf[i] = P.f16 ? h2f(P.hx[i]) : bf2f(P.hx[i]);
    }
    CK(hipMalloc(&P.x, P.hx.size() * 2)); CK(hipMemcpy(P.x, P.hx.data(), P.hx.size() * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&P.y, (size_t)bs * m * 2));
    CK(hipMalloc(&P.scale, 4)); CK(hipMemcpy(P.scale, &P.hs, 4, hipMemcpyHostToDevice));
    // sampled rows: every row of the first, a middle and the last row tile + a stride through the rest
    for (int64_t r = 0; r < 16; ++r) { P.rows.push_back(r); P.rows.push_back(m - 16 + r); P.rows.push_back((m / 32) * 16 + r); }
    for (int64_t r = 16; r < m - 16; r += 37) P.rows.push_back(r);
    const int maxq = (1 << P.bits) - 1;
    P.ref.assign(P.rows.size() * bs, 0.0);
    for (size_t si = 0; si < P.rows.size(); ++si) {
        const uint8_t *cr = &P.codes[(size_t)P.rows[si] * d];
        for (int64_t b = 0; b < bs; ++b) {
            const float *xr = &P.xf[(size_t)b * d];
            double a = 0.0;
            for (int64_t k = 0; k < d; ++k) a += (((double)cr[k] / maxq) * 2.0 - 1.0) * (double)xr[k];
            P.ref[si * bs + b] = a * (double)P.hs;
        }
    }
}

static double check(Problem &P)
{
    std::vector<uint16_t> hy((size_t)P.bs * P.m);
    CK(hipMemcpy(hy.data(), P.y, hy.size() * 2, hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (size_t si = 0; si < P.rows.size(); ++si)
        for (int64_t b = 0; b < P.bs; ++b) {
            const double got = P.f16 ? h2f(hy[(size_t)b * P.m + P.rows[si]]) : bf2f(hy[(size_t)b * P.m + P.rows[si]]);
            const double want = P.ref[si * P.bs + b];
            num += (got - want) * (got - want); den += want * want;
        }
    return std::sqrt(num / den);
}

static hipStream_t st; static hipEvent_t e0, e1;
static const char *g_filter = nullptr;     // argv[7]: run only the variants whose name contains this
static int g_steps = 0;                    // K2LAB_STEPS: launches per graph (rocprofv3 counter passes want few)

// fn(weight copy index) enqueues one launch on st
static void bench(Problem &P, const char *name, const std::function<int(int)> &fn, int steps = 300)
{
    if (g_filter && !strstr(name, g_filter)) return;
    if (g_steps) steps = g_steps;
    CK(hipMemset(P.y, 0xff, (size_t)P.bs * P.m * 2));
    if (fn(0)) { printf("%-40s LAUNCH ERROR: %s\n", name, quipamd_last_error()); return; }
    hipError_t er = hipStreamSynchronize(st);
    if (er != hipSuccess) { printf("%-40s RUNTIME ERROR %s\n", name, hipGetErrorString(er)); exit(2); }
    const double rel = check(P);
    double us[2];
    for (int cold = 1; cold >= 0; --cold) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < steps; ++i) fn(cold ? i % P.nring : 0);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        us[cold] = best * 1e3 / steps;
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    const double bytes = (double)P.wbytes + 2.0 * P.bs * P.d + 2.0 * P.bs * P.m, flops = 2.0 * P.bs * P.m * P.d;
    printf("%-40s rel %.2e %s cold %8.3f us (%6.0f GB/s %7.1f TF)  warm %8.3f us (%7.1f TF)\n", name, rel, rel < 4e-3 ? "ok  " : "FAIL",
           us[1], bytes / us[1] / 1e3, flops / us[1] / 1e6, us[0], flops / us[0] / 1e6);
    fflush(stdout);
}

static K2Args mkargs(Problem &P, int ring)
{
    K2Args A;
    A.x = P.x; A.qw = (const u32x4 *)(P.w + (size_t)ring * P.wbytes); A.d = P.d;
    EpiArgs &e = A.e;
    e.scale = P.scale; e.zero = nullptr; e.bias = nullptr; e.y = P.y; e.qfn = QUIPAMD_QFN_B; e.maxq = (1 << P.bits) - 1;
    e.y_f32 = 0; e.y_f16 = P.f16 ? 1 : 0; e.accumulate = 0; e.two_over_maxq = 2.0f / e.maxq; e.bs = P.bs; e.m = P.m;
    return A;
}

template <int T> static void null_bench(const char *name, int grid, size_t lds)
{
    const int steps = 500;
    if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void *)null_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < steps; ++i) null_kernel<T><<<grid, T, lds, st>>>(nullptr);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    printf("null %-28s grid %5d x %4d thr, lds %6zu: %.3f us/launch\n", name, grid, T, lds, best * 1e3 / steps);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
}

template <class ACT> static void run_mode(const std::string &mode, Problem &P)
{
    auto old = [&](const char *nm, int rt, int bt, int nw, int split) {
        if (P.f16) return;
        bench(P, nm, [&, rt, bt, nw, split](int r) {
            quipamd_tune_dequant_gemm(rt, bt, nw, split);
            return quipamd_dequant_gemm(P.x, 2, (const int32_t *)(P.w + (size_t)r * P.wbytes), P.bits, 1, 1, P.scale, nullptr, nullptr, P.y, 2, 0,
                                        P.bs, P.m, P.d, st);
        });
    };
#define H_CASE(B, RT, NW, NCH) if (P.bits == B && P.d / (512 / B) <= NW * NCH && (P.m / 16) % RT == 0) \
        bench(P, "h<" #B ",rt" #RT ",nw" #NW ",nch" #NCH ">", [&](int r) { return launch_h<B, ACT, RT, NW, NCH>(mkargs(P, r), st); });
#define S_CASE(B, NW, KSP, SPW, DD) if (P.bits == B) bench(P, "s<" #B ",nw" #NW ",ksp" #KSP ",spw" #SPW ",d" #DD ">", [&](int r) { return launch_s<B, ACT, NW, KSP, SPW, DD>(mkargs(P, r), st); });
#define MB_CASE(B, WR, WB, RT, BT, NL) if (P.bits == B && P.d % 256 == 0) \
        bench(P, "mb<" #B "," #WR "x" #WB "," #RT "x" #BT ",nl" #NL ">", [&](int r) { return launch_mb2<B, ACT, WR, WB, RT, BT, NL>(mkargs(P, r), st); }, 100);
#define MB32_CASE(B, WR, WB, RT, BT, NL) if (P.bits == B && P.d % 256 == 0) \
        bench(P, "mb32<" #B "," #WR "x" #WB "," #RT "x" #BT ",nl" #NL ">", [&](int r) { return launch_mb2<B, ACT, WR, WB, RT, BT, NL, true>(mkargs(P, r), st); }, 100);
    if (mode == "old" || mode == "h" || mode == "s" || mode == "mb") old("old heuristic", 0, 0, 0, 0);
#define HL_CASE(B, NW, NCH, NL) if (P.bits == B && P.d / (512 / B) == NW * NCH) \
        bench(P, "hl<" #B ",nw" #NW ",nch" #NCH ",nl" #NL ">", [&](int r) { return launch_hl<B, ACT, NW, NCH, NL>(mkargs(P, r), st); });
    if (mode == "h") {
        HL_CASE(2, 8, 2, 4) HL_CASE(2, 8, 2, 2) HL_CASE(2, 8, 2, 8) HL_CASE(2, 8, 1, 2) HL_CASE(2, 8, 1, 4) HL_CASE(2, 4, 4, 4) HL_CASE(2, 16, 1, 4)
        HL_CASE(4, 8, 4, 4) HL_CASE(4, 8, 2, 4)
        H_CASE(2, 2, 8, 1) H_CASE(2, 2, 4, 4) H_CASE(2, 4, 4, 4) H_CASE(2, 1, 8, 2) H_CASE(2, 1, 4, 4) H_CASE(2, 1, 2, 8) H_CASE(2, 1, 16, 1) H_CASE(2, 1, 8, 1) H_CASE(2, 1, 4, 2) H_CASE(2, 1, 2, 4)
        H_CASE(4, 1, 8, 4) H_CASE(4, 1, 4, 8) H_CASE(4, 1, 8, 2) H_CASE(4, 1, 4, 4)
    }
    if (mode == "s") {
        S_CASE(2, 2, 4, 1, 3) S_CASE(2, 1, 8, 1, 2)
        S_CASE(2, 7, 2, 1, 3) S_CASE(2, 7, 2, 2, 2) S_CASE(2, 7, 1, 2, 3) S_CASE(2, 7, 1, 4, 2) S_CASE(2, 4, 3, 1, 3) S_CASE(2, 4, 3, 2, 2) S_CASE(2, 4, 2, 2, 3) S_CASE(2, 8, 1, 2, 3)
        S_CASE(4, 7, 2, 1, 3) S_CASE(4, 7, 1, 2, 2)
    }
    if (mode == "mb") {
        MB32_CASE(2, 4, 2, 4, 4, 4) MB32_CASE(2, 4, 2, 4, 4, 3) MB32_CASE(4, 4, 2, 2, 4, 4)
        MB_CASE(2, 4, 2, 4, 4, 4) MB_CASE(2, 4, 2, 4, 4, 3) MB_CASE(2, 4, 2, 2, 2, 4) MB_CASE(4, 4, 2, 2, 4, 4) MB_CASE(4, 4, 2, 2, 2, 4) MB_CASE(2, 4, 1, 4, 8, 4) MB_CASE(2, 4, 1, 4, 8, 2) MB_CASE(2, 2, 4, 4, 2, 2) MB_CASE(2, 4, 2, 2, 4, 2) MB_CASE(2, 4, 2, 4, 4, 2) MB_CASE(2, 2, 4, 4, 2, 1) MB_CASE(2, 2, 2, 4, 4, 1)
        MB_CASE(2, 2, 2, 2, 2, 1) MB_CASE(2, 2, 2, 4, 2, 1) MB_CASE(2, 4, 1, 2, 4, 1) MB_CASE(2, 2, 4, 2, 2, 2) MB_CASE(2, 4, 2, 2, 2, 2)
        MB_CASE(4, 2, 4, 4, 2, 2) MB_CASE(4, 4, 2, 2, 4, 2) MB_CASE(4, 2, 2, 2, 2, 1)
    }
}

#ifdef K2_PROBE
// one launch with the probe buffer attached; prints per-stamp statistics over all waves
static void probe_run(Problem &P, const char *name, const std::function<int(int)> &fn, int nwaves, bool stamps)
{
    const size_t nblk = 4096;
    unsigned long long *buf;
    CK(hipMalloc(&buf, nblk * 32 * 8 * 8));
    for (int warm = 0; warm < 2; ++warm) fn(1 % P.nring);
    CK(hipStreamSynchronize(st));
    CK(hipMemset(buf, 0, nblk * 32 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_k2_probe), &buf, sizeof(buf)));
    fn(2 % P.nring);
    CK(hipStreamSynchronize(st));
    unsigned long long *nul = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_k2_probe), &nul, sizeof(nul)));
    std::vector<unsigned long long> h(nblk * 32 * 8);
    CK(hipMemcpy(h.data(), buf, h.size() * 8, hipMemcpyDeviceToHost));
    CK(hipFree(buf));
    printf("probe %s\n", name);
    if (stamps) {
        // per workgroup: t0 of its earliest wave is the origin; print median / p10 / p90 over all waves of (stamp - origin)
        std::vector<double> v[8];
        std::vector<double> skew;
        unsigned long long gmin = ~0ull, gmax = 0;
        for (size_t b = 0; b < nblk; ++b) {
            unsigned long long t0 = ~0ull, t0max = 0;
            for (int w = 0; w < nwaves; ++w) { const unsigned long long t = h[(b * 32 + w) * 8]; if (t) { t0 = std::min(t0, t); t0max = std::max(t0max, t); } }
            if (t0 == ~0ull) continue;
            skew.push_back((double)(t0max - t0));
            if ((b & 7) == 0) { gmin = std::min(gmin, t0); for (int w = 0; w < nwaves; ++w) gmax = std::max(gmax, h[(b * 32 + w) * 8 + 7]); }
            for (int w = 0; w < nwaves; ++w)
                for (int i = 0; i < 8; ++i) if (h[(b * 32 + w) * 8 + i]) v[i].push_back((double)(h[(b * 32 + w) * 8 + i] - t0));
        }
        auto pct = [](std::vector<double> &a, double p) { if (a.empty()) return 0.0; std::sort(a.begin(), a.end()); return a[(size_t)(p * (a.size() - 1))]; };
        printf("  wave-start skew inside a workgroup: median %.0f p90 %.0f ticks; XCD-0 first start -> last end: %llu ticks\n", pct(skew, .5), pct(skew, .9), gmax - gmin);
        const char *nm[8] = {"start", "all requested", "chunk0 landed", "last chunk landed", "compute done", "parked", "barrier passed", "stored"};
        for (int i = 0; i < 8; ++i) printf("  t%d %-18s median %7.0f  p10 %7.0f  p90 %7.0f ticks (n=%zu)\n", i, nm[i], pct(v[i], .5), pct(v[i], .1), pct(v[i], .9), v[i].size());
    } else {
        // wave classes by index: compute waves [0, nwaves-2), weight loader nwaves-2, x loader nwaves-1
        for (int cls = 0; cls < 3; ++cls) {
            double a[4] = {0, 0, 0, 0}; size_t n = 0;
            for (size_t b = 0; b < nblk; ++b)
                for (int w = 0; w < nwaves; ++w) {
                    const int c = w < nwaves - 2 ? 0 : w == nwaves - 2 ? 1 : 2;
                    if (c != cls || !h[(b * 32 + w) * 8 + 3]) continue;
                    for (int i = 0; i < 4; ++i) a[i] += (double)h[(b * 32 + w) * 8 + i];
                    ++n;
                }
            if (n) printf("  %-14s waves %6zu: vmcnt-wait %8.0f  barrier %8.0f  issue %8.0f  total %8.0f ticks (mean per wave)\n",
                          cls == 0 ? "compute" : cls == 1 ? "weight loader" : "x loader", n, a[0] / n, a[1] / n, a[2] / n, a[3] / n);
        }
    }
    fflush(stdout);
}
#endif

int main(int argc, char **argv)
{
    const std::string mode = argc > 1 ? argv[1] : "h";
    CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if (mode == "null") {
        null_bench<64>("1 wave", 256, 0); null_bench<256>("4 waves", 256, 0); null_bench<512>("8 waves", 256, 0);
        null_bench<1024>("16 waves", 256, 0); null_bench<512>("8 waves, 128K lds", 256, 128 * 1024);
        null_bench<1024>("16 waves, 128K lds", 256, 128 * 1024); null_bench<256>("4 waves x 512 wg", 512, 0);
        null_bench<512>("8 waves x 1024 wg", 1024, 0); null_bench<64>("1 wave x 1 wg", 1, 0);
        return 0;
    }
    Problem P;
    P.m = argc > 2 ? atoll(argv[2]) : 4096; P.d = argc > 3 ? atoll(argv[3]) : 4096; P.bs = argc > 4 ? atoll(argv[4]) : 16;
    P.bits = argc > 5 ? atoi(argv[5]) : 2; P.f16 = argc > 6 && !strcmp(argv[6], "f16");
    g_filter = argc > 7 ? argv[7] : nullptr;
    if (getenv("K2LAB_STEPS")) g_steps = atoi(getenv("K2LAB_STEPS"));
    printf("== %s  m=%lld d=%lld bs=%lld bits=%d act=%s\n", mode.c_str(), (long long)P.m, (long long)P.d, (long long)P.bs, P.bits, P.f16 ? "f16" : "bf16");
    make_problem(P);
#ifdef K2_PROBE
    if (mode == "probe_h") {
        probe_run(P, "h<2,rt1,nw8,nch2>", [&](int r) { return launch_h<2, ActBF16, 1, 8, 2>(mkargs(P, r), st); }, 8, true);
        probe_run(P, "h<2,rt1,nw4,nch4>", [&](int r) { return launch_h<2, ActBF16, 1, 4, 4>(mkargs(P, r), st); }, 4, true);
        probe_run(P, "h<2,rt1,nw16,nch1>", [&](int r) { return launch_h<2, ActBF16, 1, 16, 1>(mkargs(P, r), st); }, 16, true);
        return 0;
    }
    if (mode == "probe_s") {
        probe_run(P, "s<2,nw7,ksp2,spw2,d2>", [&](int r) { return launch_s<2, ActBF16, 7, 2, 2, 2>(mkargs(P, r), st); }, 16, false);
        probe_run(P, "s<2,nw7,ksp2,spw1,d3>", [&](int r) { return launch_s<2, ActBF16, 7, 2, 1, 3>(mkargs(P, r), st); }, 16, false);
        probe_run(P, "s<2,nw7,ksp1,spw4,d2>", [&](int r) { return launch_s<2, ActBF16, 7, 1, 4, 2>(mkargs(P, r), st); }, 10, false);
        return 0;
    }
#endif
    if (P.f16) run_mode<ActF16>(mode, P); else run_mode<ActBF16>(mode, P);
    return 0;
}
