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

template <int BITS> struct Deq;
template <> struct Deq<2> {
    static constexpr int KC = 256, NT = 8;
    static constexpr float OFF = 4.0f;           // dequantised value = OFF + code
    // A fragment (4 dwords = 8 bf16) of MFMA step t from the lane's 4 packed dwords
    static __device__ __forceinline__ uint4 frag(const uint4 &w, int t)
    {
        const uint32_t src = (t >> 1) == 0 ? w.x : (t >> 1) == 1 ? w.y : (t >> 1) == 2 ? w.z : w.w;
        const uint32_t base = opaque(0x40804080u);
        uint32_t o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int sh = 2 * (4 * (t & 1) + v) - 5;
            const uint32_t shifted = sh >= 0 ? (src >> sh) : (src << (-sh));
            o[v] = bfi(0x00600060u, shifted, base);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct Deq<4> {
    static constexpr int KC = 128, NT = 4;
    static constexpr float OFF = 16.0f;
    static __device__ __forceinline__ uint4 frag(const uint4 &w, int t)
    {
        const uint32_t src = t == 0 ? w.x : t == 1 ? w.y : t == 2 ? w.z : w.w;
        const uint32_t base = opaque(0x41804180u);
        uint32_t o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int sh = 4 * v - 3;
            const uint32_t shifted = sh >= 0 ? (src >> sh) : (src << (-sh));
            o[v] = bfi(0x00780078u, shifted, base);
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};

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

}   // namespace
