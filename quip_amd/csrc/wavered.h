// wavered.h -- wave-wide reductions on the DPP network: 4 DPP operands + 4 v_readlane instead of the six ds_bpermute round trips hipcc makes
// of a __shfl_xor butterfly (~100 clocks each on a lone wave).  Every lane of the wave must be active.
#pragma once
#include <hip/hip_runtime.h>

namespace {

template <int CTRL> __device__ __forceinline__ float dpp_f(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <bool MAX> __device__ __forceinline__ float wave_reduce(float v)
{
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
    v = op(v, dpp_f<0xB1>(v));       // quad_perm [1,0,3,2]
    v = op(v, dpp_f<0x4E>(v));       // quad_perm [2,3,0,1]
    v = op(v, dpp_f<0x141>(v));      // row_half_mirror
    v = op(v, dpp_f<0x140>(v));      // row_mirror: every lane of a 16-lane row holds the row's result
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return op(op(r0, r1), op(r2, r3));
}

}   // namespace
