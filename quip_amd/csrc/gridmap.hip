// gridmap.hip -- K5: the grid functions of quant.py:6-21 and the W <-> grid-coordinate maps of
// vector_balance.py:500-532, evaluated in the SAME dtype sequence as the reference
// (on the Balance path an fp16 model runs the qfn-b RMS, divide and clamp
// in fp16; DT<T>::rnd() re-rounds after every elementwise op exactly where torch would).
//
// All kernels are elementwise / one reduction: HBM-bound.  Built with -ffp-contract=off.
#include "common.h"

namespace {

// ---- qfn b scale: 2.4*sqrt(mean(w^2)) + 1e-16 (quant.py:150, vector_balance.py:522) ---------
template <class T>
__global__ __launch_bounds__(256) void sumsq_kernel(const void *__restrict__ W, int64_t n, double *__restrict__ acc)
{
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float w = DT<T>::load(W, i);
        s += (double)DT<T>::rnd(w * w);            // x.square() is materialised in T
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

template <class T>
__global__ void qfnb_scale_finish(const double *__restrict__ acc, int64_t n, float *__restrict__ scale_out)
{
    const float mean = DT<T>::rnd((float)(acc[0] / (double)n));   // wide accumulate, one rounding to T
    const float root = DT<T>::rnd(__fsqrt_rn(mean));
    const float s = DT<T>::rnd(root * 2.4f);                       // python scalar enters in fp32
    scale_out[0] = DT<T>::rnd(s + 1e-16f);
}

// ---- grid coordinates (no rounding) -----------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void gridmap_b_kernel(const void *__restrict__ W, const float *__restrict__ scale,
                                                        float maxq, float *__restrict__ out, int64_t n)
{
    const float s = scale[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = DT<T>::rnd(__fdiv_rn(DT<T>::load(W, i), s));      // wr = w / scale
        v = DT<T>::rnd(v + 1.0f);
        v = DT<T>::rnd(v * 0.5f);                                    // /2 is exact up to the rounding
        v = DT<T>::rnd(v * maxq);
        out[i] = fminf(fmaxf(v, 0.0f), maxq);                        // torch.clamp(…, 0, maxq)
    }
}

template <class T>
__global__ __launch_bounds__(256) void gridmap_a_kernel(const void *__restrict__ W, const float *__restrict__ scale,
                                                        const float *__restrict__ zero, float maxq,
                                                        float *__restrict__ out, int64_t m, int64_t d)
{
    const int64_t n = m * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / d;
        const float v = __fdiv_rn(DT<T>::load(W, i), scale[r]) + zero[r];   // fp32: scale/zero are fp32 [m,1]
        out[i] = fminf(fmaxf(v, 0.0f), maxq);
    }
}

// ---- round-to-nearest through the grid (Quantizer.quantize, quant.py:144-157) --------------------
template <class T, int QFN>
__global__ __launch_bounds__(256) void quantize_kernel(const void *__restrict__ W, const float *__restrict__ scale,
                                                       const float *__restrict__ zero, float maxq,
                                                       uint8_t *__restrict__ codes, void *__restrict__ Wout,
                                                       int64_t m, int64_t d)
{
    const int64_t n = m * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = DT<T>::load(W, i);
        float q, deq;
        if constexpr (QFN == QUIPAMD_QFN_B) {                        // quant.py:10-15, all ops in T
            const float s = scale[0];
            float v = DT<T>::rnd(__fdiv_rn(x, s));
            v = DT<T>::rnd(v + 1.0f);
            v = DT<T>::rnd(v * 0.5f);
            v = DT<T>::rnd(v * maxq);
            q = fminf(fmaxf(rintf(v), 0.0f), maxq);                  // torch.round = half to even
            float t = DT<T>::rnd(__fdiv_rn(q, maxq));
            t = DT<T>::rnd(t * 2.0f);
            t = DT<T>::rnd(t - 1.0f);
            deq = DT<T>::rnd(t * s);
        } else {
            const int64_t r = i / d;
            const float sc = scale[r], z = zero[r];                  // fp32 [m,1]: result promotes to fp32
            if constexpr (QFN == QUIPAMD_QFN_A) q = fminf(fmaxf(rintf(__fdiv_rn(x, sc)) + z, 0.0f), maxq);   // quant.py:7
            else q = rintf(fminf(fmaxf(__fdiv_rn(x, sc) + z, 0.0f), maxq));                                    // quant.py:19-20
            deq = sc * (q - z);
        }
        if (codes) codes[i] = (uint8_t)q;
        if (Wout) DT<T>::store(Wout, i, deq);
    }
}

template <class TO, int QFN>
__global__ __launch_bounds__(256) void codes_to_weight_kernel(const uint8_t *__restrict__ codes,
                                                              const float *__restrict__ scale,
                                                              const float *__restrict__ zero, float maxq,
                                                              void *__restrict__ Wout, int64_t m, int64_t d)
{
    const int64_t n = m * d;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float q = (float)codes[i];
        float v;
        if constexpr (QFN == QUIPAMD_QFN_B) {                        // vector_balance.py:528-529 (fp32)
            v = __fdiv_rn(q, maxq) * 2.0f;
            v = v - 1.0f;
            v = v * scale[0];
        } else {                                                     // vector_balance.py:519
            const int64_t r = i / d;
            v = scale[r] * (q - zero[r]);
        }
        DT<TO>::store(Wout, i, v);                                   // .half()
    }
}

inline int ew_grid(int64_t n) { int64_t g = (n + 255) / 256; return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g)); }

}   // namespace

extern "C" int quipamd_qfnb_scale(const void *W, int dtype, int64_t numel, float *scale_out, double *workspace,
                                  void *stream)
{
    QA_REQUIRE(W && scale_out && workspace, QUIPAMD_ERR_ARG, "qfnb_scale: null pointer");
    QA_REQUIRE(numel > 0, QUIPAMD_ERR_SHAPE, "qfnb_scale: empty tensor");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, sizeof(double), s) != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "memset failed");
    QA_DISPATCH_DTYPE(dtype, T, {
        sumsq_kernel<T><<<ew_grid(numel), 256, 0, s>>>(W, numel, workspace);
        qfnb_scale_finish<T><<<1, 1, 0, s>>>(workspace, numel, scale_out);
    });
    QA_LAUNCH_CHECK("quipamd_qfnb_scale");
    return QUIPAMD_OK;
}

extern "C" int quipamd_gridmap(const void *W, int dtype, int qfn, const float *scale, const float *zero, int maxq,
                               float *Wgrid, int64_t m, int64_t d, void *stream)
{
    QA_REQUIRE(W && scale && Wgrid, QUIPAMD_ERR_ARG, "gridmap: null pointer");
    QA_REQUIRE(qfn == QUIPAMD_QFN_B || zero, QUIPAMD_ERR_ARG, "gridmap: qfn a needs zero");
    QA_REQUIRE(maxq >= 1 && maxq <= 255, QUIPAMD_ERR_ARG, "gridmap: maxq out of range");
    if (m * d == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    QA_DISPATCH_DTYPE(dtype, T, {
        if (qfn == QUIPAMD_QFN_B) gridmap_b_kernel<T><<<ew_grid(m * d), 256, 0, s>>>(W, scale, (float)maxq, Wgrid, m * d);
        else if (qfn == QUIPAMD_QFN_A) gridmap_a_kernel<T><<<ew_grid(m * d), 256, 0, s>>>(W, scale, zero, (float)maxq, Wgrid, m, d);
        else return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "gridmap: qfn %d", qfn);
    });
    QA_LAUNCH_CHECK("quipamd_gridmap");
    return QUIPAMD_OK;
}

extern "C" int quipamd_quantize(const void *W, int dtype, int qfn, const float *scale, const float *zero, int maxq,
                                uint8_t *codes_out, void *W_out, int64_t m, int64_t d, void *stream)
{
    QA_REQUIRE(W && scale, QUIPAMD_ERR_ARG, "quantize: null pointer");
    QA_REQUIRE(qfn == QUIPAMD_QFN_B || zero, QUIPAMD_ERR_ARG, "quantize: qfn a/c need zero");
    QA_REQUIRE(maxq >= 1 && maxq <= 255, QUIPAMD_ERR_ARG, "quantize: maxq out of range");
    if (m * d == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const int g = ew_grid(m * d);
    QA_DISPATCH_DTYPE(dtype, T, {
        if (qfn == QUIPAMD_QFN_A) quantize_kernel<T, QUIPAMD_QFN_A><<<g, 256, 0, s>>>(W, scale, zero, (float)maxq, codes_out, W_out, m, d);
        else if (qfn == QUIPAMD_QFN_B) quantize_kernel<T, QUIPAMD_QFN_B><<<g, 256, 0, s>>>(W, scale, zero, (float)maxq, codes_out, W_out, m, d);
        else if (qfn == QUIPAMD_QFN_C) quantize_kernel<T, QUIPAMD_QFN_C><<<g, 256, 0, s>>>(W, scale, zero, (float)maxq, codes_out, W_out, m, d);
        else return qa_fail(QUIPAMD_ERR_ARG, "quantize: qfn %d", qfn);
    });
    QA_LAUNCH_CHECK("quipamd_quantize");
    return QUIPAMD_OK;
}

extern "C" int quipamd_codes_to_weight(const uint8_t *codes, int qfn, const float *scale, const float *zero,
                                       int maxq, void *W_out, int out_dtype, int64_t m, int64_t d, void *stream)
{
    QA_REQUIRE(codes && scale && W_out, QUIPAMD_ERR_ARG, "codes_to_weight: null pointer");
    QA_REQUIRE(qfn == QUIPAMD_QFN_B || zero, QUIPAMD_ERR_ARG, "codes_to_weight: qfn a needs zero");
    if (m * d == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const int g = ew_grid(m * d);
    QA_DISPATCH_DTYPE(out_dtype, T, {
        if (qfn == QUIPAMD_QFN_B) codes_to_weight_kernel<T, QUIPAMD_QFN_B><<<g, 256, 0, s>>>(codes, scale, zero, (float)maxq, W_out, m, d);
        else codes_to_weight_kernel<T, QUIPAMD_QFN_A><<<g, 256, 0, s>>>(codes, scale, zero, (float)maxq, W_out, m, d);
    });
    QA_LAUNCH_CHECK("quipamd_codes_to_weight");
    return QUIPAMD_OK;
}
