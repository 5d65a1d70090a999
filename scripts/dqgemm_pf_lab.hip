// LAB RECORD (round 4: moved out of the library).  The round-3 prefill kernel: correct, tested, and SLOWER than the mb family at every
// prefill shape (profiles/r03t_k2_prefill.jsonl: 686 vs 924 TFLOP/s at 4096^2 x 2048).  It shipped as a forced-only family (cfg family 5)
// that the heuristic never picked; it is kept here as the negative result it is, not built by __graft_entry__.build().
// To rebuild it as a lab: hipcc --offload-arch=gfx950 -O3 -I include -I quip_amd/csrc -c scripts/dqgemm_pf_lab.hip
// dqgemm_pf.hip -- K2 for prefill-sized batches (bs >= 256): y[b, r] = alpha (sum_k (OFF + q[r,k]) x[b,k] - c0 sum_k x[b,k]) + bias[r]
// (quant.py:222-233 with the qfn-b grid of quant.py:10-15; the contract of quipamd_dequant_gemm).
//
// The small-batch kernels dequantise a weight fragment in registers and use it for 1 - 4 batch tiles; at 2048 tokens that is the
// bound (0.40 of the bf16 MFMA peak, dense rocBLAS faster above ~1000 tokens).  Here every 2-bit tile is dequantised
// ONCE per workgroup into LDS as 16-bit values (OFF + code, exact) and a standard LDS-tiled GEMM mainloop runs on it:
//   * tile TM x TN x 64 per workgroup of 4 waves (2 x 2), wave tile (TM/2) x (TN/2) on v_mfma_f32_32x32x16_{bf16,f16}: with
//     TM = 256, TN = 128 a wave holds 4 x 2 accumulator tiles (128 registers) and reads 6 KiB of fragments per 8 MFMAs;
//   * the packed weights of the workgroup's 16 row tiles are read as the STREAM layout stores them (16 bytes per lane per 256-column
//     chunk, one chunk ahead), dequantised two fragments at a time (DeqT: shift + bfi per pair) and written as [row][64 k] with
//     the 16-byte slots XOR-swizzled by (row / 2) % 8 -- the activation tile goes global -> registers -> LDS the same way, one
//     64-column step ahead, so the MFMAs of step i run on LDS buffer i % 2 while step i + 1 lands in the other: ONE barrier per step;
//   * sum_k x[b,k] for the offset term falls out of the B fragments the row-half-0 waves read anyway (v_dot2 against ones);
//   * D[m][n = batch]: a lane holds 4 consecutive output features of one batch row per register quad -> 8-byte stores.
// qfn b, 2-bit, m % TM == 0, d % 256 == 0; any bs (rows past bs read as zeros and are not stored).
//
// STATUS (round 3, profiles/r03t_k2_prefill.jsonl): correct (tests/test_gpu_dqgemm_v2.py) and SLOWER than the mb kernel it was meant to
// replace -- 686 / 697 / 802 TFLOP/s at 4096^2 x 2048, 28672 x 7168 x 256, 8192^2 x 1024 against 924 / 938 / 1114 (dense bf16 rocBLAS on
// the same box: 1058 / 825 / 1252, i.e. 0.33 - 0.50 of the 2.5 PF peak itself at these shapes) -- so the heuristic never picks it
// (family 5 by cfg only).  Why: a 256 x 128 tile needs 128 + 146 registers per lane, i.e. ONE wave per SIMD (and LDS bandwidth rules
// out smaller per-wave tiles: 64 x 64 per wave would read 256 B/clk/CU of fragments); with one wave per SIMD nothing overlaps the
// MFMAs but the wave's own instruction stream, and hipcc schedules the step as  [fragment reads -> wait -> 8 MFMAs] x 4, then
// [global wait -> 12 ds_write_b128 + 64 dequant VALU], then the barrier: ~3700 cycles per 64-column step for 1024 cycles of matrix
// pipe.  The mb kernel hides the same work behind two loader waves and eight compute waves of 4 x 4 16x16x32 tiles.  What would
// be needed here is the interleave Tensile's assembly kernels have (staging instructions placed between the MFMAs of the previous
// step) -- ~4.5 fillers per MFMA at this tile, at the edge of what one wave per SIMD can hide (MI355X_MICROARCH.md: <= 5 per gap).
#include "common.h"
#include "dq_common.h"
#include "k2_dispatch.h"

namespace {

typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <class ACT> struct Mfma32;
template <> struct Mfma32<ActBF16> {
    static __device__ __forceinline__ f32x16_t run(const uint4 &a, const uint4 &b, const f32x16_t &c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mfma32<ActF16> {
    static __device__ __forceinline__ f32x16_t run(const uint4 &a, const uint4 &b, const f32x16_t &c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
};

struct PfArgs {
    const uint16_t *x;                    // [bs, d] bf16 / f16
    const uint4 *qw;                      // STREAM
    const float *scale, *bias;            // [1], [m] or null
    void *y;                              // [bs, m]
    int64_t bs, m, d;
    int y_f32, y_f16, maxq;
    float two_over_maxq;
    uint32_t ntm;                         // m / TM
};

__device__ __forceinline__ uint32_t swz(uint32_t row, uint32_t slot) { return row * 128u + ((slot ^ ((row >> 1) & 7u)) << 4); }

template <class ACT, int TM, int TN>
__global__ __launch_bounds__(256) void dq_pf_kernel(PfArgs A)
{
    typedef DeqT<2, ACT> Q;
    constexpr int WM = TM / 2, WN = TN / 2, MT = WM / 32, NT = WN / 32;
    constexpr int ABYTES = TM * 128, BBYTES = TN * 128, BUF = ABYTES + BBYTES;
    constexpr int RTW = TM / 16 / 4;                                  // 16-row weight tiles per wave
    constexpr int BLD = TN * 8 / 256;                                 // 16-byte activation loads per thread per step
    extern __shared__ __attribute__((aligned(16))) char smem[];       // [2][A | B] + xsum [TN]
    float *xsum_s = reinterpret_cast<float *>(smem + 2 * BUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const uint32_t ti = blockIdx.x % A.ntm, tj = blockIdx.x / A.ntm;
    const int64_t row0 = (int64_t)ti * TM, col0 = (int64_t)tj * TN;
    const uint32_t nch = (uint32_t)(A.d / 256);

    // ---- loaders ---------------------------------------------------------------------------------------------------------------------
    // weights: wave w owns row tiles RTW w .. RTW w + RTW - 1 of the workgroup; one uint4 per lane per tile per 256-column chunk
    const uint4 *wbase = A.qw + ((uint64_t)(row0 / 16 + RTW * wave) * nch) * 64 + lane;
    auto load_w = [&](uint4 (&w)[RTW], uint32_t c) {
#pragma unroll
        for (int i = 0; i < RTW; ++i) w[i] = wbase[((uint64_t)i * nch + c) * 64];
    };
    // activations: thread t moves 16-byte pieces t, t + 256, ...: piece p = (row p / 8, slot p % 8) of the TN x 64 tile
    auto load_x = [&](uint4 (&xr)[BLD], uint32_t k0) {
#pragma unroll
        for (int i = 0; i < BLD; ++i) {
            const uint32_t p = tid + 256 * i, r = p >> 3, sl = p & 7;
            const int64_t b = col0 + r;
            xr[i] = b < A.bs ? *reinterpret_cast<const uint4 *>(A.x + b * A.d + k0 + 8 * sl) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_x = [&](char *buf, const uint4 (&xr)[BLD]) {
#pragma unroll
        for (int i = 0; i < BLD; ++i) {
            const uint32_t p = tid + 256 * i, r = p >> 3, sl = p & 7;
            *reinterpret_cast<uint4 *>(buf + ABYTES + swz(r, sl)) = xr[i];
        }
    };
    // dequantise the 64-column window j (0..3) of the chunk held in w and write it as [row][64] (slot g <- step 2j, slot 4 + g <- 2j + 1)
    const uint32_t wr = lane & 15, wg = lane >> 4;
    auto store_w = [&](char *buf, const uint4 (&w)[RTW], int j) {
#pragma unroll
        for (int i = 0; i < RTW; ++i) {
            const uint32_t r = 16 * (RTW * wave + i) + wr;
            const u32x4 ww = {w[i].x, w[i].y, w[i].z, w[i].w};
            const u32x4 f0 = Q::frag(ww, 2 * j), f1 = Q::frag(ww, 2 * j + 1);
            *reinterpret_cast<uint4 *>(buf + swz(r, wg)) = make_uint4(f0[0], f0[1], f0[2], f0[3]);
            *reinterpret_cast<uint4 *>(buf + swz(r, 4 + wg)) = make_uint4(f1[0], f1[1], f1[2], f1[3]);
        }
    };

    f32x16_t acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float xs[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) xs[b] = 0.f;

    uint4 wcur[RTW], wnxt[RTW], xr[BLD];
    load_w(wcur, 0);
    load_x(xr, 0);
    if (nch > 1) load_w(wnxt, 1);
    store_x(smem, xr);
    store_w(smem, wcur, 0);
    __syncthreads();

    const uint32_t arow = wm * WM + (lane & 31), brow = wn * WN + (lane & 31), kh = lane >> 5;
    const uint32_t nsteps = nch * 4;
    auto compute = [&](const char *buf) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint4 af[MT], bf[NT];
#pragma unroll
            for (int a = 0; a < MT; ++a) af[a] = *reinterpret_cast<const uint4 *>(buf + swz(arow + 32 * a, 2 * s + kh));
#pragma unroll
            for (int b = 0; b < NT; ++b) bf[b] = *reinterpret_cast<const uint4 *>(buf + ABYTES + swz(brow + 32 * b, 2 * s + kh));
            if (wm == 0) {
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    xs[b] = ACT::dot2(bf[b].x, ACT::ONES, xs[b]);
                    xs[b] = ACT::dot2(bf[b].y, ACT::ONES, xs[b]);
                    xs[b] = ACT::dot2(bf[b].z, ACT::ONES, xs[b]);
                    xs[b] = ACT::dot2(bf[b].w, ACT::ONES, xs[b]);
                }
            }
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b) acc[a][b] = Mfma32<ACT>::run(af[a], bf[b], acc[a][b]);
        }
    };

    // ---- main loop: chunk c = 4 steps of 64 columns; step it computes on buffer it % 2 while step it + 1 is staged into the other -------
    for (uint32_t c = 0; c < nch; ++c) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t it = 4 * c + j;
            char *cur = smem + (it & 1) * BUF, *nxt = smem + ((it + 1) & 1) * BUF;
            const bool more = it + 1 < nsteps;
            if (more) load_x(xr, (it + 1) * 64);
            compute(cur);
            if (more) {
                store_x(nxt, xr);
                if (j < 3) store_w(nxt, wcur, j + 1);
                else store_w(nxt, wnxt, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < RTW; ++i) wcur[i] = wnxt[i];
        if (c + 2 < nch) load_w(wnxt, c + 2);
    }

    // ---- sum_k x per batch column: the two k halves of a fragment sit 32 lanes apart ----------------------------------------------------------
    if (wm == 0) {
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            xs[b] += __shfl_xor(xs[b], 32);
            if (lane < 32) xsum_s[wn * WN + 32 * b + lane] = xs[b];
        }
    }
    __syncthreads();
    const float alpha = A.scale[0] * A.two_over_maxq, c0 = Q::OFF + 0.5f * (float)A.maxq;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const uint32_t cl = wn * WN + 32 * b + (lane & 31);
        const int64_t bb = col0 + cl;
        const float off = c0 * xsum_s[cl];
#pragma unroll
        for (int a = 0; a < MT; ++a) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                              // register quad q: rows 8 q + 4 (lane / 32) + 0..3 of the 32 x 32 tile
                const int64_t r = row0 + wm * WM + 32 * a + 8 * q + 4 * (lane >> 5);
                float4 bi = make_float4(0.f, 0.f, 0.f, 0.f);
                if (A.bias) bi = *reinterpret_cast<const float4 *>(A.bias + r);
                const float v0 = alpha * (acc[a][b][4 * q + 0] - off) + bi.x, v1 = alpha * (acc[a][b][4 * q + 1] - off) + bi.y;
                const float v2 = alpha * (acc[a][b][4 * q + 2] - off) + bi.z, v3 = alpha * (acc[a][b][4 * q + 3] - off) + bi.w;
                if (bb < A.bs) {
                    if (A.y_f32) *reinterpret_cast<float4 *>((float *)A.y + bb * A.m + r) = make_float4(v0, v1, v2, v3);
                    else {
                        uint2 pk;
                        if (A.y_f16) {
                            pk.x = (uint32_t)f32_to_f16_bits(v0) | ((uint32_t)f32_to_f16_bits(v1) << 16);
                            pk.y = (uint32_t)f32_to_f16_bits(v2) | ((uint32_t)f32_to_f16_bits(v3) << 16);
                        } else {
                            pk.x = (uint32_t)f32_to_bf16_bits(v0) | ((uint32_t)f32_to_bf16_bits(v1) << 16);
                            pk.y = (uint32_t)f32_to_bf16_bits(v2) | ((uint32_t)f32_to_bf16_bits(v3) << 16);
                        }
                        *reinterpret_cast<uint2 *>((uint16_t *)A.y + bb * A.m + r) = pk;
                    }
                }
            }
        }
    }
}

template <class ACT, int TM, int TN> int launch_pf(const PfArgs &A0, hipStream_t s)
{
    PfArgs A = A0;
    A.ntm = (uint32_t)(A.m / TM);
    const size_t lds = (size_t)2 * (TM + TN) * 128 + TN * 4;
    auto kern = dq_pf_kernel<ACT, TM, TN>;
    static QaPerDevice attr;
    const int d = attr.dev();
    if (d < 0 || !attr.done[d]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm (prefill): cannot raise dynamic LDS to %zu", lds);
        if (d >= 0) attr.done[d] = true;
    }
    const unsigned grid = A.ntm * (unsigned)((A.bs + TN - 1) / TN);
    kern<<<grid, 256, lds, s>>>(A);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm (prefill)");
    return QUIPAMD_OK;
}

template <class ACT> int run_pf(const K2Call &c, const PfArgs &A, hipStream_t s)
{
    const int p1 = c.cfg[0] == K2_FAM_PF ? c.cfg[1] : 0;
    if (p1 == 22 || (p1 == 0 && c.bs > 3072 && c.m % 256 == 0 && (c.m / 256) * ((c.bs + 255) / 256) >= 200)) return launch_pf<ACT, 256, 256>(A, s);
    if (p1 == 0 || p1 == 21) return launch_pf<ACT, 256, 128>(A, s);
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: no prefill kernel %d (21: 256 x 128, 22: 256 x 256)", p1);
}

}   // namespace

bool k2pf_supported(const K2Call &c)
{
    return c.bits == 2 && c.qfn == QUIPAMD_QFN_B && c.maxq == 3 && !c.accumulate && c.m % 256 == 0 && c.d % 256 == 0 && c.bs >= 1 &&
           (c.x_dtype == QUIPAMD_BF16 || c.x_dtype == QUIPAMD_F16);
}

int k2pf_launch(const K2Call &c, void *stream)
{
    QA_REQUIRE(k2pf_supported(c), QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: the prefill kernel runs 2-bit qfn-b layers with m %% 256 == 0 and d %% 256 == 0");
    PfArgs A;
    A.x = (const uint16_t *)c.x; A.qw = (const uint4 *)c.qweight; A.scale = c.scale; A.bias = c.bias; A.y = c.y;
    A.bs = c.bs; A.m = c.m; A.d = c.d; A.y_f32 = c.y_dtype == QUIPAMD_F32; A.y_f16 = c.y_dtype == QUIPAMD_F16; A.maxq = c.maxq;
    A.two_over_maxq = 2.0f / (float)c.maxq; A.ntm = 0;
    hipStream_t s = (hipStream_t)stream;
    return c.x_dtype == QUIPAMD_F16 ? run_pf<ActF16>(c, A, s) : run_pf<ActBF16>(c, A, s);
}
