// common.h -- shared helpers for the gfx950 kernels of libquip_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "quip_amd.h"

// error reporting (capi.hip)
int qa_fail(int code, const char *fmt, ...);

#define QA_REQUIRE(cond, code, ...)                        \
    do {                                                   \
        if (!(cond)) return qa_fail((code), __VA_ARGS__);  \
    } while (0)

#define QA_LAUNCH_CHECK(name)                                                                  \
    do {                                                                                       \
        hipError_t e__ = hipGetLastError();                                                    \
        if (e__ != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "%s: %s", (name), hipGetErrorString(e__)); \
    } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// ---- storage dtypes --------------------------------------------------------------------------
struct F32 { typedef float storage; };
struct F16 { typedef uint16_t storage; };
struct BF16 { typedef uint16_t storage; };

__device__ __forceinline__ float bf16_bits_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// round to nearest even on the gfx950 converter (v_cvt_pk_bf16_f32: ONE instruction; the integer form was seven plus a NaN branch --
// in the latency-bound decode kernels every 64 bytes of straight-line code is an instruction-cache miss of ~100 ns)
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f)
{
    return __builtin_bit_cast(uint16_t, (__bf16)f);
}

__device__ __forceinline__ float f16_bits_to_f32(uint16_t h) { return __half2float(__ushort_as_half(h)); }
// The empty asm pins the fp32 value: without it hipcc folds a preceding fp32 multiply / add into v_fma_mixlo_f16, i.e. ONE
// rounding of the exact result, where the reference (torch: fp32 op, then .half()) rounds TWICE -- on the rare exact-tie
// cases the two differ by an fp16 ulp (found by the driver-level golden test: 3 of 262144 weights of one layer).
__device__ __forceinline__ uint16_t f32_to_f16_bits(float f)
{
    asm volatile("" : "+v"(f));
    return __half_as_ushort(__float2half_rn(f));
}

template <class T> struct DT;
template <> struct DT<F32> {
    typedef float storage;
    static __device__ __forceinline__ float load(const void *p, int64_t i) { return ((const float *)p)[i]; }
    static __device__ __forceinline__ void store(void *p, int64_t i, float v) { ((float *)p)[i] = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }   // value after rounding to the dtype
};
template <> struct DT<F16> {
    typedef uint16_t storage;
    static __device__ __forceinline__ float load(const void *p, int64_t i) { return f16_bits_to_f32(((const uint16_t *)p)[i]); }
    static __device__ __forceinline__ void store(void *p, int64_t i, float v) { ((uint16_t *)p)[i] = f32_to_f16_bits(v); }
    static __device__ __forceinline__ float rnd(float v) { return f16_bits_to_f32(f32_to_f16_bits(v)); }
};
template <> struct DT<BF16> {
    typedef uint16_t storage;
    static __device__ __forceinline__ float load(const void *p, int64_t i) { return bf16_bits_to_f32(((const uint16_t *)p)[i]); }
    static __device__ __forceinline__ void store(void *p, int64_t i, float v) { ((uint16_t *)p)[i] = f32_to_bf16_bits(v); }
    static __device__ __forceinline__ float rnd(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }
};

// dispatch a runtime dtype enum to a template argument
#define QA_DISPATCH_DTYPE(dt, NAME, ...)                                         \
    switch (dt) {                                                                \
    case QUIPAMD_F32: { typedef F32 NAME; __VA_ARGS__; } break;                   \
    case QUIPAMD_F16: { typedef F16 NAME; __VA_ARGS__; } break;                   \
    case QUIPAMD_BF16: { typedef BF16 NAME; __VA_ARGS__; } break;                 \
    default: return qa_fail(QUIPAMD_ERR_ARG, "bad dtype %d", (int)(dt));         \
    }

// hipFuncSetAttribute is per DEVICE: one process driving several GPUs must raise the dynamic-LDS limit on each of them
// (a per-process `static bool` made the second device's first launch fail).  One instance per kernel instantiation.
struct QaPerDevice {
    bool done[64] = {};
    int dev() const
    {
        int d = 0;
        (void)hipGetDevice(&d);
        return (d >= 0 && d < 64) ? d : -1;
    }
};

static inline int qa_div_up(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
