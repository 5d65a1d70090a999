// dq_common.h -- in-register dequantisation of STREAM-packed codes into bf16 MFMA A fragments, shared by dqgemm.hip (K2)
// and dqgemm_vop.hip.
#pragma once
#include "common.h"

namespace {

// (shifted & mask) | base in ONE VALU op: v_bfi_b32 D = (S0 & S1) | (~S0 & S2).  hipcc folds `base & ~mask` when
// base is a literal and then emits v_and + v_or, so the base constant is passed through opaque() (an EMPTY asm:
// it hides the value from the optimiser but contains no instruction).  The instruction itself must come from the
// compiler: a hand-written `asm("v_bfi_b32 ...")` result fed to an MFMA misses the VALU-write -> MFMA-operand wait
// states (cdna_hip_programming.md 5.7 item 2) and silently corrupts tiles -- caught by
// tests/test_gpu_dqgemm.py::test_forced_workgroup_shapes_agree.
__device__ __forceinline__ uint32_t opaque(uint32_t v)
{
    asm("" : "+v"(v));
    return v;
}

__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t shifted, uint32_t base)
{
    return (shifted & mask) | (base & ~mask);
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// ---- activation dtype of the MFMA (x and the dequantised weights share it) ---------------------------------------------
// The code lands on mantissa bits whose unit weight is exactly 1 under a fixed exponent, so value = OFF + code, exact:
//   bf16 (7 mantissa bits):  2 bit 0x4080 | c << 5 = 4 + c      4 bit 0x4180 | c << 3 = 16 + c
//   fp16 (10 mantissa bits): 2 bit 0x4400 | c << 8 = 4 + c      4 bit 0x4c00 | c << 6 = 16 + c
// (the reference operator widens x to fp32, quant.py:226-229: fp16 activations keep all their bits on the fp16 pipe,
//  bf16 is for models that already run in bf16)
struct ActBF16 {
    static constexpr int DTYPE = QUIPAMD_BF16;
    static constexpr uint32_t ONES = 0x3f803f80u;
    static __device__ __forceinline__ f32x4_t mfma(const u32x4 &a, const u32x4 &b, const f32x4_t &c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16_t mfma32(const u32x4 &a, const u32x4 &b, const f32x16_t &c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c)
    {
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
    }
};
struct ActF16 {
    static constexpr int DTYPE = QUIPAMD_F16;
    static constexpr uint32_t ONES = 0x3c003c00u;
    static __device__ __forceinline__ f32x4_t mfma(const u32x4 &a, const u32x4 &b, const f32x4_t &c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x16_t mfma32(const u32x4 &a, const u32x4 &b, const f32x16_t &c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c)
    {
        return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a), __builtin_bit_cast(f16x2_t, b), c, false);
    }
};
template <class ACT, int BITS> struct DqParams;
template <> struct DqParams<ActBF16, 2> { static constexpr uint32_t BASE = 0x40804080u, MASK = 0x00600060u; static constexpr int POS = 5; };
template <> struct DqParams<ActBF16, 4> { static constexpr uint32_t BASE = 0x41804180u, MASK = 0x00780078u; static constexpr int POS = 3; };
template <> struct DqParams<ActF16, 2> { static constexpr uint32_t BASE = 0x44004400u, MASK = 0x03000300u; static constexpr int POS = 8; };
template <> struct DqParams<ActF16, 4> { static constexpr uint32_t BASE = 0x4c004c00u, MASK = 0x03c003c0u; static constexpr int POS = 6; };

template <int BITS, class ACT> struct DeqT {
    static constexpr int KC = 512 / BITS, NT = KC / 32;
    static constexpr float OFF = BITS == 2 ? 4.0f : 16.0f;           // dequantised value = OFF + code
    typedef DqParams<ACT, BITS> P;
    // A fragment (4 dwords = 8 halves) of MFMA step t from the lane's 4 packed dwords: field `slot` of the source dword
    // (bits [BITS*slot, +BITS) of each 16-bit half) is shifted onto mantissa position POS
    static __device__ __forceinline__ u32x4 frag4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int t)
    {
        const uint32_t src = BITS == 2 ? ((t >> 1) == 0 ? w0 : (t >> 1) == 1 ? w1 : (t >> 1) == 2 ? w2 : w3)
                                       : (t == 0 ? w0 : t == 1 ? w1 : t == 2 ? w2 : w3);
        const uint32_t base = opaque(P::BASE);
        u32x4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int slot = BITS == 2 ? 4 * (t & 1) + v : v;
            const int sh = BITS * slot - P::POS;
            const uint32_t shifted = sh >= 0 ? (src >> sh) : (src << (-sh));
            o[v] = bfi(P::MASK, shifted, base);
        }
        return o;
    }
    static __device__ __forceinline__ u32x4 frag(const u32x4 &w, int t) { return frag4(w[0], w[1], w[2], w[3], t); }
    static __device__ __forceinline__ uint4 frag(const uint4 &w, int t)
    {
        const u32x4 o = frag4(w.x, w.y, w.z, w.w, t);
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};
// ---- "multi-exponent" dequantisation (2-bit codes): 10 VALU per packed dword instead of 16 --------------------------------
// The VALU port issues one instruction per 4 cycles per SIMD, and at bs <= 16 nothing amortises the dequantisation: it is
// THE bound of the streaming kernel (profiles/r02e_k2_pmc_summary.txt: 4.4 M VALU instructions per launch at 28672 x 7168 =
// 8.2 us of issue slots, as long as the HBM stream itself).  A 2-bit field can sit on ANY mantissa position p whose unit
// weight is made 1 by the exponent (value = 2^(M-p) + code, M mantissa bits), so three neighbouring fields share one shift:
//     bf16 (M = 7):  fields {0,1,2} as they are, {3,4,5} >> 6, {6,7} >> 12   -> positions 0,2,4   offsets 128, 32, 8
//     fp16 (M = 10): fields {0,1} << 4, {2,3,4} as they are, {5,6,7} >> 6     -> positions 4,6,8   offsets  64, 16, 4
// The offset now depends on k, so the epilogue needs S_off[b] = sum_k OFF_k x[b,k] beside S_1[b] = sum_k x[b,k]:
//     sum_k q x = acc - S_off,   y = alpha * (acc - S_off - z * S_1)   (z = maxq/2 for qfn b, zero[r] for qfn a)
// Both are linear in x only; the kernels compute them once per workgroup and stage on the matrix pipe with the constant
// A fragments off_frag(t) (the bf16/fp16 encodings of OFF_k -- the very BASE constants of the dequantisation) and "ones".
// Exactness: 128 + 3 needs 8 significant bits (bf16 has 8), 64 + 3 seven (fp16 has 11); products stay exact in fp32, the
// accumulation loses log2(OFF_max / 1.1) ~ 7 (bf16) / 6 (fp16) bits to cancellation: ~1e-5 relative, tested at 1e-3.
template <class ACT> struct MEField;
template <> struct MEField<ActBF16> {
    static constexpr int M = 7, BIAS = 127, P0 = 0;                    // target positions P0, P0+2, P0+4
    static constexpr int shift(int i) { return i < 3 ? 0 : i < 6 ? 6 : 12; }          // > 0: right
    static constexpr int pos(int i) { return 2 * (i < 3 ? i : i < 6 ? i - 3 : i - 6); }
};
template <> struct MEField<ActF16> {
    static constexpr int M = 10, BIAS = 15, P0 = 4;
    static constexpr int shift(int i) { return i < 2 ? -4 : i < 5 ? 0 : 6; }
    static constexpr int pos(int i) { return i < 2 ? 4 + 2 * i : i < 5 ? 2 * i : 2 * i - 6; }
};
template <class ACT> struct DeqME2 {
    static constexpr int KC = 256, NT = 8;
    static constexpr bool UNIFORM = false;
    static constexpr float OFF = 0.f;
    typedef MEField<ACT> F;
    static constexpr int P0 = F::P0;
    static constexpr uint32_t base_at(int p) { return (uint32_t)((F::BIAS + F::M - p) << F::M) * 0x10001u; }   // = encoding of OFF, both halves
    static constexpr uint32_t mask_at(int p) { return (3u << p) * 0x10001u; }
    // the three BASE constants, made opaque ONCE per kernel (hipcc would otherwise fold base & ~mask and emit and + or)
    struct Consts { uint32_t b[3]; };
    static __device__ __forceinline__ Consts make_consts()
    {
        Consts c;
#pragma unroll
        for (int k = 0; k < 3; ++k) c.b[k] = opaque(base_at(P0 + 2 * k));
        return c;
    }
    static __device__ __forceinline__ u32x4 frag(const u32x4 &w, int t, const Consts &c)
    {
        const uint32_t src = w[t >> 1];
        u32x4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int i = 4 * (t & 1) + v, sh = F::shift(i), p = F::pos(i);
            const uint32_t shifted = sh > 0 ? (src >> sh) : sh < 0 ? (src << (-sh)) : src;
            o[v] = bfi(mask_at(p), shifted, c.b[(p - P0) / 2]);
        }
        return o;
    }
    // the constant A fragment whose MFMA against x gives sum_k OFF_k x[b,k] for the k's of step t
    static __device__ __forceinline__ u32x4 off_frag(int t, const Consts &c)
    {
        u32x4 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) o[v] = c.b[(F::pos(4 * (t & 1) + v) - P0) / 2];
        return o;
    }
};
// kernel-facing selector: multi-exponent for 2 bits, the uniform form (one offset, S_off = OFF * S_1) otherwise
template <int BITS, class ACT> struct DeqSel : DeqT<BITS, ACT> {
    static constexpr bool UNIFORM = true;
    struct Consts { };
    static __device__ __forceinline__ Consts make_consts() { return Consts{}; }
    static __device__ __forceinline__ u32x4 frag(const u32x4 &w, int t, const Consts &) { return DeqT<BITS, ACT>::frag(w, t); }
    static __device__ __forceinline__ u32x4 off_frag(int, const Consts &) { return u32x4{0u, 0u, 0u, 0u}; }
};
template <class ACT> struct DeqSel<2, ACT> : DeqME2<ACT> {};
// the uniform-offset form for 2 bits as well (lab builds of dq_mb_kernel, -DK2_MB_UNIFORM: measured 5 % slower than the multi-exponent form)
template <int BITS, class ACT> struct DeqUni : DeqT<BITS, ACT> {
    static constexpr bool UNIFORM = true;
    struct Consts { };
    static __device__ __forceinline__ Consts make_consts() { return Consts{}; }
    static __device__ __forceinline__ u32x4 frag(const u32x4 &w, int t, const Consts &) { return DeqT<BITS, ACT>::frag(w, t); }
    static __device__ __forceinline__ u32x4 off_frag(int, const Consts &) { return u32x4{0u, 0u, 0u, 0u}; }
};

template <int BITS> using Deq = DeqT<BITS, ActBF16>;                  // the round-1 kernels are bf16-only

union Frag {
    uint4 u;
    bf16x8_t v;
};
union Pair {
    uint32_t u;
    bf16x2_t v;
};

__device__ __forceinline__ float dot_ones(const uint4 &x, float acc)
{
    Pair one, p;
    one.u = 0x3f803f80u;
    p.u = x.x; acc = __builtin_amdgcn_fdot2_f32_bf16(p.v, one.v, acc, false);
    p.u = x.y; acc = __builtin_amdgcn_fdot2_f32_bf16(p.v, one.v, acc, false);
    p.u = x.z; acc = __builtin_amdgcn_fdot2_f32_bf16(p.v, one.v, acc, false);
    p.u = x.w; acc = __builtin_amdgcn_fdot2_f32_bf16(p.v, one.v, acc, false);
    return acc;
}

// ---- epilogue shared by the K2 kernels ---------------------------------------------------------------------------------
struct EpiArgs {
    const float *scale;   // [1] (qfn b) or [m] (qfn a)
    const float *zero;    // [m] or null
    const float *bias;    // [m] or null
    void *y;
    int qfn, maxq, y_f32, y_f16, accumulate;
    float two_over_maxq;
    int64_t bs, m;
};

// raw epilogue parameters of 4 consecutive output rows, FETCHED at kernel start (their memory latency hides
// under the weight stream instead of extending the critical path after the reduction) and only turned into
// coefficients in epilogue_store
struct EpiRow {
    float4 sc, zr, bi;
};

__device__ __forceinline__ EpiRow load_epi(const EpiArgs &e, int64_t r0)
{
    EpiRow c;
    if (e.qfn == QUIPAMD_QFN_B) {
        const float s = e.scale[0];
        c.sc = make_float4(s, s, s, s);
        c.zr = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        c.sc = *reinterpret_cast<const float4 *>(e.scale + r0);
        c.zr = *reinterpret_cast<const float4 *>(e.zero + r0);
    }
    c.bi = e.bias ? *reinterpret_cast<const float4 *>(e.bias + r0) : make_float4(0.f, 0.f, 0.f, 0.f);
    return c;
}

__device__ __forceinline__ void epilogue_store(const EpiArgs &e, const EpiRow &c, float off, const f32x4_t &acc, float xs,
                                               int64_t b, int64_t r0)
{
    if (b >= e.bs) return;
    const float sc[4] = {c.sc.x, c.sc.y, c.sc.z, c.sc.w}, zr[4] = {c.zr.x, c.zr.y, c.zr.z, c.zr.w};
    const float bi[4] = {c.bi.x, c.bi.y, c.bi.z, c.bi.w};
    float out[4];
    if (e.qfn == QUIPAMD_QFN_B) {
        const float alpha = sc[0] * e.two_over_maxq, t = (off + 0.5f * (float)e.maxq) * xs;
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = alpha * (acc[i] - t) + bi[i];
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = sc[i] * (acc[i] - (off + zr[i]) * xs) + bi[i];
    }
    if (e.y_f32) {
        float4 *dst = reinterpret_cast<float4 *>((float *)e.y + b * e.m + r0);
        if (e.accumulate) {
            const float4 old = *dst;
            out[0] += old.x; out[1] += old.y; out[2] += old.z; out[3] += old.w;
        }
        *dst = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        uint2 pk;
        if (e.y_f16) {
            pk.x = (uint32_t)f32_to_f16_bits(out[0]) | ((uint32_t)f32_to_f16_bits(out[1]) << 16);
            pk.y = (uint32_t)f32_to_f16_bits(out[2]) | ((uint32_t)f32_to_f16_bits(out[3]) << 16);
        } else {
            pk.x = (uint32_t)f32_to_bf16_bits(out[0]) | ((uint32_t)f32_to_bf16_bits(out[1]) << 16);
            pk.y = (uint32_t)f32_to_bf16_bits(out[2]) | ((uint32_t)f32_to_bf16_bits(out[3]) << 16);
        }
        *reinterpret_cast<uint2 *>((uint16_t *)e.y + b * e.m + r0) = pk;
    }
}


}   // namespace
