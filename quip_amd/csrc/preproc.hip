// preproc.hip -- the elementwise / reduction chains of QuantMethod.preproc (method.py:134-193) as a few launches instead of ~15 torch
// ones over the d x d Hessian and the m x d weights (rescale 0.24 -> 0.15 ms, trace + ridge 0.11 -> 0.06 ms at 4096^2):
//
//   rescale (method.py:140-156):   H /= max|H|;  s = clamp(sqrt(sqrt(clamp(diag H, 1e-8) / clamp(diag(W^T W), 1e-8))), 1e-8);
//                                  W <- W s (columns), rounded to the layer's dtype;   H <- (H / s_j) / s_i
//      preproc_stats_kernel    one pass over H (max|H| through an integer atomicMax on the bit pattern: exact and order-free) and one over W
//                              (column sums of squares as per-row-chunk partials: no float atomics, the result is deterministic)
//      preproc_scale_kernel    the partials in a fixed order -> s[d]
//      preproc_apply_kernel    W and H rewritten in place, every division an IEEE division in the reference's order
//   trace scale + ridge (method.py:165):   H <- H (n / (tr H + 1e-8)) + ridge I     trace in one workgroup (fixed order), then one pass
//
// HBM-bound passes; built with -ffp-contract=off like gridmap.hip (the reference's operations one by one).
#include "common.h"

namespace {

constexpr int PP_ROWS = 64;        // rows of W per partial of the column sums

// grid.x < nbh: max|H| over a grid-stride slice;  grid.x >= nbh: workgroup (column block of 256, row chunk of PP_ROWS) of W
template <class T>
__global__ __launch_bounds__(256) void preproc_stats_kernel(const float *__restrict__ H, int64_t nh, unsigned *__restrict__ amax_bits,
                                                            const void *__restrict__ W, int64_t m, int64_t d, float *__restrict__ part, int nbh)
{
    if ((int)blockIdx.x < nbh) {
        float mx = 0.f;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += (int64_t)nbh * 256) mx = fmaxf(mx, fabsf(H[i]));
        for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off));
        if ((threadIdx.x & 63) == 0) atomicMax(amax_bits, __builtin_bit_cast(unsigned, mx));     // non-negative floats order like their bits
        return;
    }
    const int64_t b = (int64_t)blockIdx.x - nbh, ncb = (d + 255) / 256;
    const int64_t cb = b % ncb, rb = b / ncb;
    const int64_t c = cb * 256 + threadIdx.x;
    if (c >= d) return;
    const int64_t r1 = (rb + 1) * PP_ROWS < m ? (rb + 1) * PP_ROWS : m;
    float s = 0.f;
    for (int64_t r = rb * PP_ROWS; r < r1; ++r) {
        const float w = DT<T>::load(W, r * d + c);
        s += w * w;
    }
    part[rb * d + c] = s;
}

__global__ __launch_bounds__(256) void preproc_scale_kernel(const float *__restrict__ H, const unsigned *__restrict__ amax_bits,
                                                            const float *__restrict__ part, int64_t nrb, int64_t d, float *__restrict__ s_out)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= d) return;
    const float amax = __builtin_bit_cast(float, amax_bits[0]);
    float w2 = 0.f;
    for (int64_t rb = 0; rb < nrb; ++rb) w2 += part[rb * d + c];
    const float dh = fmaxf(__fdiv_rn(H[c * d + c], amax), 1e-8f);
    const float q = __fdiv_rn(dh, fmaxf(w2, 1e-8f));
    s_out[c] = fmaxf(__fsqrt_rn(__fsqrt_rn(q)), 1e-8f);
}

template <class T>
__global__ __launch_bounds__(256) void preproc_apply_kernel(float *__restrict__ H, int64_t d, const unsigned *__restrict__ amax_bits,
                                                            const float *__restrict__ s, void *__restrict__ W, int64_t m)
{
    const float amax = __builtin_bit_cast(float, amax_bits[0]);
    const int64_t nh = d * d, nw = m * d, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += stride) {
        const int64_t r = i / d, c = i - r * d;
        H[i] = __fdiv_rn(__fdiv_rn(__fdiv_rn(H[i], amax), s[c]), s[r]);      // ((H / max) / s[None, :]) / s[:, None]
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nw; i += stride) {
        const int64_t c = i % d;
        DT<T>::store(W, i, DT<T>::load(W, i) * s[c]);                          // fp32 product, one rounding to the layer's dtype
    }
}

__global__ __launch_bounds__(1024) void preproc_trace_kernel(const float *__restrict__ H, int64_t d, float *__restrict__ tr)
{
    __shared__ float part[16];
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < d; c += 1024) s += H[c * d + c];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += part[i];
        tr[0] = t;
    }
}

__global__ __launch_bounds__(256) void preproc_trace_apply_kernel(float *__restrict__ H, int64_t d, const float *__restrict__ tr, float ridge)
{
    const float f = __fdiv_rn((float)d, tr[0] + 1e-8f);
    const int64_t nh = d * d, stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += stride) {
        const int64_t r = i / d, c = i - r * d;
        const float v = H[i] * f;
        H[i] = r == c ? v + ridge : v;                                         // H * (n / (tr + 1e-8)) + ridge * eye
    }
}

}   // namespace

extern "C" int64_t quipamd_preproc_workspace_bytes(int64_t m, int64_t d)
{
    return 64 + ((m + PP_ROWS - 1) / PP_ROWS) * d * 4;
}

extern "C" int quipamd_preproc_rescale(float *H, void *W, int w_dtype, int64_t m, int64_t d, float *s_out, void *workspace, void *stream)
{
    QA_REQUIRE(m >= 1 && d >= 1, QUIPAMD_ERR_SHAPE, "preproc_rescale: bad shape");
    QA_REQUIRE(H && W && s_out && workspace, QUIPAMD_ERR_ARG, "preproc_rescale: null pointer");
    hipStream_t s = (hipStream_t)stream;
    unsigned *amax = (unsigned *)workspace;
    float *part = (float *)((char *)workspace + 64);
    if (hipMemsetAsync(amax, 0, 64, s) != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "preproc_rescale: memset failed");
    const int64_t nrb = (m + PP_ROWS - 1) / PP_ROWS, ncb = (d + 255) / 256;
    const int nbh = 1024;
    QA_DISPATCH_DTYPE(w_dtype, T, {
        preproc_stats_kernel<T><<<(unsigned)(nbh + nrb * ncb), 256, 0, s>>>(H, d * d, amax, W, m, d, part, nbh);
        preproc_scale_kernel<<<(unsigned)ncb, 256, 0, s>>>(H, amax, part, nrb, d, s_out);
        preproc_apply_kernel<T><<<2048, 256, 0, s>>>(H, d, amax, s_out, W, m);
    });
    QA_LAUNCH_CHECK("quipamd_preproc_rescale");
    return QUIPAMD_OK;
}

extern "C" int quipamd_preproc_trace_ridge(float *H, int64_t d, float ridge, void *workspace, void *stream)
{
    QA_REQUIRE(d >= 1 && H && workspace, QUIPAMD_ERR_ARG, "preproc_trace_ridge: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    preproc_trace_kernel<<<1, 1024, 0, s>>>(H, d, (float *)workspace);
    preproc_trace_apply_kernel<<<2048, 256, 0, s>>>(H, d, (const float *)workspace, ridge);
    QA_LAUNCH_CHECK("quipamd_preproc_trace_ridge");
    return QUIPAMD_OK;
}
