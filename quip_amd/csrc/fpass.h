// fpass.h -- the decode-step operator pass shared by csrc/decode_fused.hip (operator chain in the prologue of a dequant-GEMM) and
// csrc/decode_attn.hip (output-side operators of q / k / v in the prologue of the attention launch): a Kronecker operator p x q
// applied to one row held as fp16 images in LDS, both mix stages on v_mfma_f32_16x16x32_f16 with host-prepared B fragments
// (quipamd_fop, include/quip_amd.h).  NW = waves of the workgroup that run the pass.
#pragma once
#include "common.h"
#include "dq_common.h"

#include <type_traits>

namespace {

typedef quipamd_fop Fop;

__device__ __forceinline__ uint32_t pack_f16x2(float a, float b)
{
    return (uint32_t)f32_to_f16_bits(a) | ((uint32_t)f32_to_f16_bits(b) << 16);
}
__device__ __forceinline__ float4 f16x4_to_f32(const uint2 &r)
{
    return make_float4(f16_bits_to_f32(r.x & 0xffff), f16_bits_to_f32(r.x >> 16), f16_bits_to_f32(r.y & 0xffff), f16_bits_to_f32(r.y >> 16));
}

// wave-wide sum on the DPP network; every lane gets the total
__device__ __forceinline__ float fg_wave_sum(float v)
{
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});
    v += dpp(v, std::integral_constant<int, 0x4E>{});
    v += dpp(v, std::integral_constant<int, 0x141>{});
    v += dpp(v, std::integral_constant<int, 0x140>{});
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
// ---- the operator pass --------------------------------------------------------------------------------------------------------
// LDS images of one pass (P x Q operator, n = P Q):
//   ZT  f16 [Q][P + 8]   z^T: element at image position (a, b) sits at ZT[b][a]          input of stage 1 (A fragments: 8 consecutive a')
//   Z1  f16 [P][Q + 8]   result of stage 1 ("mix a"), row a                                 input of stage 2 (A fragments: 8 consecutive b')
//   ZF  f32 [P][Q + 4]   result of stage 2 = the operator's image, gathered by store_idx
template <int P, int Q, int NW = 16> struct PassDims {
    static constexpr int N = P * Q, PS = P + 8, QS = Q + 8, QF = Q + 4;
    static constexpr int NT = (P / 16) * (Q / 16);                 // 16 x 16 output tiles per stage
    static constexpr int TPW = (NT + NW - 1) / NW;                // tiles per wave
    static constexpr int S0 = P / 32, S1 = Q / 32;                 // k-steps of stage 1 / stage 2
    static constexpr int NV = (N / 4 + 64 * NW - 1) / (64 * NW);   // float4 slots per thread (natural order: slot v4 = tid + 64 NW u)
    static constexpr size_t ZT_B = (size_t)Q * PS * 2, Z1_B = (size_t)P * QS * 2, ZF_B = (size_t)P * QF * 4;
    static constexpr size_t BYTES = ZT_B + Z1_B + ZF_B;
    static_assert(P % 32 == 0 && Q % 32 == 0 && (Q & (Q - 1)) == 0, "operator shape");
};

// this wave's factor fragments (host layout: F0 [P/16][P/32][64 lanes] uint4, F1 [Q/16][Q/32][64] uint4).  A wave's tiles are
// wave, wave + 16, ...: because 16 is a multiple of P/16 and of Q/16 they all share the stage-1 fragment (it depends on at = tile %
// (P/16) only) and the stage-2 fragment (bt = tile % (Q/16)): ONE set per wave, loaded straight from global memory in fragment order.
template <int P, int Q> struct PassFrags {
    uint4 f0[P / 32];
    uint4 f1[Q / 32];
};

template <int P, int Q, int NW = 16> __device__ __forceinline__ void load_f0(const Fop &op, int wave, int lane, PassFrags<P, Q> &fr)
{
    typedef PassDims<P, Q, NW> D;
    static_assert(NW % (P / 16) == 0 && NW % (Q / 16) == 0, "a wave's tiles share their fragments");
    const uint4 *F0 = reinterpret_cast<const uint4 *>(op.F0);
    const int at = wave % (P / 16);
    if (wave < D::NT) {                                     // 64 x 32: eight tiles -- waves 8..15 own none and must not pull fragments
#pragma unroll                                              // through the CU's one vector-memory path (64 B per clock, the prologue's bound)
        for (int S = 0; S < D::S0; ++S) fr.f0[S] = F0[(at * D::S0 + S) * 64 + lane];
    }
}
template <int P, int Q, int NW = 16> __device__ __forceinline__ void load_f1(const Fop &op, int wave, int lane, PassFrags<P, Q> &fr)
{
    typedef PassDims<P, Q, NW> D;
    const uint4 *F1 = reinterpret_cast<const uint4 *>(op.F1);
    const int bt = wave % (Q / 16);
    if (wave < D::NT) {
#pragma unroll
        for (int S = 0; S < D::S1; ++S) fr.f1[S] = F1[(bt * D::S1 + S) * 64 + lane];
    }
}

// scatter 4 consecutive natural-order values into the stage-1 input image: value e goes to image position pos[e] = (a, b) -> ZT[b][a]
template <int P, int Q> __device__ __forceinline__ void scatter4(uint16_t *ZT, const float4 &v, const uint2 &pos)
{
    typedef PassDims<P, Q> D;
    constexpr int qsh = __builtin_ctz(Q);
    const float vv[4] = {v.x, v.y, v.z, v.w};
    const int pp[4] = {(int)(pos.x & 0xffff), (int)(pos.x >> 16), (int)(pos.y & 0xffff), (int)(pos.y >> 16)};
#pragma unroll
    for (int e = 0; e < 4; ++e) ZT[(pp[e] & (Q - 1)) * D::PS + (pp[e] >> qsh)] = f32_to_f16_bits(vv[e]);
}

// the same for 4 values that already ARE f16 bits (the previous GEMM's output)
template <int P, int Q> __device__ __forceinline__ void scatter4h(uint16_t *ZT, const uint2 &v, const uint2 &pos)
{
    typedef PassDims<P, Q> D;
    constexpr int qsh = __builtin_ctz(Q);
    const uint16_t vv[4] = {(uint16_t)(v.x & 0xffff), (uint16_t)(v.x >> 16), (uint16_t)(v.y & 0xffff), (uint16_t)(v.y >> 16)};
    const int pp[4] = {(int)(pos.x & 0xffff), (int)(pos.x >> 16), (int)(pos.y & 0xffff), (int)(pos.y >> 16)};
#pragma unroll
    for (int e = 0; e < 4; ++e) ZT[(pp[e] & (Q - 1)) * D::PS + (pp[e] >> qsh)] = vv[e];
}

// stage 1, transposed: D1[m = b][n = a] = sum_a' ZT[b][a'] M0[a][a'];  A = ZT rows (LDS), B = M0 rows (registers).  ZT -> Z1.
template <int P, int Q, int NW = 16>
__device__ __forceinline__ void mix_stage1(const uint16_t *ZT, uint16_t *Z1, const PassFrags<P, Q> &fr, int wave, int lane)
{
    typedef PassDims<P, Q, NW> D;
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < D::TPW; ++i) {
        const int tile = wave + NW * i;
        if (tile < D::NT) {
            const int at = tile % (P / 16), bt = tile / (P / 16);
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const uint16_t *arow = ZT + (16 * bt + j) * D::PS + 8 * g;
#pragma unroll
            for (int S = 0; S < D::S0; ++S) {
                const uint4 a = *reinterpret_cast<const uint4 *>(arow + 32 * S);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, fr.f0[S]), acc, 0, 0, 0);
            }
            // lane: a = 16 at + j, b = 16 bt + 4 g + {0..3}: four consecutive b of row a -> one 8-byte store
            uint2 pk;
            pk.x = pack_f16x2(acc[0], acc[1]);
            pk.y = pack_f16x2(acc[2], acc[3]);
            *reinterpret_cast<uint2 *>(Z1 + (16 * at + j) * D::QS + 16 * bt + 4 * g) = pk;
        }
    }
}
// stage 2: D2[m = a][n = b] = sum_b' Z1[a][b'] M1[b][b'];  A = Z1 rows (LDS), B = M1 rows (registers).  Z1 -> ZF.
template <int P, int Q, int NW = 16>
__device__ __forceinline__ void mix_stage2(const uint16_t *Z1, float *ZF, const PassFrags<P, Q> &fr, int wave, int lane)
{
    typedef PassDims<P, Q, NW> D;
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < D::TPW; ++i) {
        const int tile = wave + NW * i;
        if (tile < D::NT) {
            const int bt = tile % (Q / 16), at = tile / (Q / 16);
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const uint16_t *arow = Z1 + (16 * at + j) * D::QS + 8 * g;
#pragma unroll
            for (int S = 0; S < D::S1; ++S) {
                const uint4 a = *reinterpret_cast<const uint4 *>(arow + 32 * S);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, fr.f1[S]), acc, 0, 0, 0);
            }
            // lane: b = 16 bt + j, a = 16 at + 4 g + reg
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ZF[(16 * at + 4 * g + reg) * D::QF + 16 * bt + j] = acc[reg];
        }
    }
}
// stage 2 of an ACTIVATION-side pass whose consumer is the dequant-GEMM: the weights' COLUMNS are stored in image order (the output
// permutation of V is folded into the packing, free at pack time), so the image IS x~: every lane rounds its four results to f16 and
// writes them straight into the GEMM's operand row XT[a * Q + b] -- no fp32 image, no gather, no barrier in between.  Returns the
// lane's sum of the rounded values (the epilogue's sum_k x~[k]).
// The launches that read x~ dequantise their 2-bit codes with the multi-exponent scheme of dq_common.h (DeqME2: 10 instead of 16 VALU
// per packed dword): the constant a code rides on depends on where it sits in its 16-bit half, so the epilogue needs
// S_off = sum_k OFF_k x~_k beside S_1 = sum_k x~_k.  OFF_k for the element at index k of x~ (fp16 path): field i = 4 ((k >> 5) & 1) +
// ((k & 7) >> 1) of the STREAM word, OFF = 2^(10 - pos(i)) = 64, 16, 64, 16, 4, 64, 16, 4.
__device__ __forceinline__ float me2_off_f16(int k)
{
    const int i = ((k >> 3) & 4) | ((k & 7) >> 1);
    const int p = MEField<ActF16>::pos(i);
    return (float)(1 << (MEField<ActF16>::M - p));
}
struct XtSums {
    float s1, soff;
};
template <int P, int Q, int NW = 16, bool OFFS = false>      // OFFS: also sum OFF_k x~_k (the consumer dequantises with DeqME2)
__device__ __forceinline__ XtSums mix_stage2_xt(const uint16_t *Z1, uint16_t *XT, const PassFrags<P, Q> &fr, int wave, int lane)
{
    typedef PassDims<P, Q, NW> D;
    const int j = lane & 15, g = lane >> 4;
    XtSums part = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < D::TPW; ++i) {
        const int tile = wave + NW * i;
        if (tile < D::NT) {
            const int bt = tile % (Q / 16), at = tile / (Q / 16);
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            const uint16_t *arow = Z1 + (16 * at + j) * D::QS + 8 * g;
#pragma unroll
            for (int S = 0; S < D::S1; ++S) {
                const uint4 a = *reinterpret_cast<const uint4 *>(arow + 32 * S);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, fr.f1[S]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const uint16_t h = f32_to_f16_bits(acc[reg]);
                const int k = (16 * at + 4 * g + reg) * Q + 16 * bt + j;
                XT[k] = h;
                const float hv = f16_bits_to_f32(h);
                part.s1 += hv;
                if constexpr (OFFS) part.soff = fmaf(me2_off_f16(k), hv, part.soff);
            }
        }
    }
    return part;
}

// the input of an OUTPUT-side pass arrives in "ZT order": the producing GEMM's ROWS are stored so that its output vector is the
// transposed image row-major, yT[b * P + a] (the load permutation of U^T folded into the packing) -- the scatter is a straight copy
// of 16-byte chunks into the padded rows of ZT.  chunk c = (b = c / (P / 8), a8 = c % (P / 8)).
template <int P, int Q> __device__ __forceinline__ void copy_chunk_zt(uint16_t *ZT, const uint4 &v, int c)
{
    typedef PassDims<P, Q> D;
    const int b = c / (P / 8), a8 = c - b * (P / 8);
    *reinterpret_cast<uint4 *>(ZT + b * D::PS + 8 * a8) = v;
}

// the two mix stages: ZT -> Z1 -> ZF.  Caller: a barrier after the scatter; this function ends WITHOUT a barrier after writing ZF.
template <int P, int Q, int NW = 16>
__device__ __forceinline__ void mix_stages(const uint16_t *ZT, uint16_t *Z1, float *ZF, const PassFrags<P, Q> &fr, int wave, int lane)
{
    mix_stage1<P, Q, NW>(ZT, Z1, fr, wave, lane);
    __syncthreads();
    mix_stage2<P, Q, NW>(Z1, ZF, fr, wave, lane);
}

template <int P, int Q> __device__ __forceinline__ float4 gather4(const float *ZF, const uint2 &pos)
{
    typedef PassDims<P, Q> D;
    constexpr int qsh = __builtin_ctz(Q);
    const int p0 = (int)(pos.x & 0xffff), p1 = (int)(pos.x >> 16), p2 = (int)(pos.y & 0xffff), p3 = (int)(pos.y >> 16);
    return make_float4(ZF[(p0 >> qsh) * D::QF + (p0 & (Q - 1))], ZF[(p1 >> qsh) * D::QF + (p1 & (Q - 1))],
                       ZF[(p2 >> qsh) * D::QF + (p2 & (Q - 1))], ZF[(p3 >> qsh) * D::QF + (p3 & (Q - 1))]);
}

}   // namespace
