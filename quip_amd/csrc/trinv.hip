// trinv.hip -- the feedback matrix of the OPTQ sweep straight from H, without H^-1 and without rocSOLVER
//
// GPTQ.fasterquant prepares `Hinv = cholesky(cholesky_inverse(cholesky(H)), upper=True)` (gptq.py:51-54: three
// rocSOLVER calls behind torch, 3.5 us of latency per column each) and K4's OPTQ modes (ldlq.hip MODE 1 / 3) want
//     FT[c'][i'] = -Hinv[i][c] / Hinv[i][i]     (c' = d-1-c, i' = d-1-i, i < c)          (include/quip_amd.h)
// With P the column reversal and  P H P = C C^T,  N = strictly-upper part of (C diag(C)^-1)^T  -- exactly what K8
// (quipamd_cholesky_lt) returns for the flipped Hessian -- this is
//     FT = I - (I + N)^-1
// (Hinv = U with H^-1 = U^T U;  P U^-1 P is the lower Cholesky factor of P H P by uniqueness;  rows of U divided by the
// diagonal are the inverse of the unit-triangular LDL factor).  So: flip, K8, ONE unit-upper-triangular inverse:
//   diag   each 128 x 128 diagonal block by back substitution, one workgroup per block, thread = column (columns of the
//          inverse are independent; N block and X block live in LDS, no barrier inside the sweep);
//   levels b = 128, 256, ...:  [[T11, T12], [0, T22]]^-1 = [[X11, -X11 T12 X22], [0, X22]] for all pairs of b-blocks at
//          once: Y = T12 X22, X12 = -X11 Y as two batched fp32 GEMMs on the matrix pipe (v_mfma_f32_16x16x4_f32, exact
//          fp32 FMA chains), 128 x 128 tiles, the zero halves of the triangular operands skipped tile-wise
//          (~d^3/3 MACs over all levels);
//   finish FT = strictly upper part of -X.
#include "common.h"

namespace {

constexpr int TB = 128;
constexpr int KS = 16;              // k per LDS stage of the GEMM
constexpr int LDK = KS + 4;         // padded k stride (80 B: float4-aligned rows)

__global__ __launch_bounds__(256) void flip_kernel(const float *__restrict__ H, float *__restrict__ R, int64_t d)
{
    const int64_t i = blockIdx.x;
    for (int64_t j = threadIdx.x; j < d; j += 256) R[i * d + j] = H[(d - 1 - i) * d + (d - 1 - j)];
}

// X[k0:k0+cnt, k0:k0+cnt] = (I + N[k0:k0+cnt, k0:k0+cnt])^-1, zero below the diagonal
__global__ __launch_bounds__(TB) void trinv_diag_kernel(const float *__restrict__ N, float *__restrict__ X, int64_t d)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *Ns = sm;                         // [TB][TB + 1]
    float *Xs = sm + TB * (TB + 1);         // [TB][TB]   Xs[k][j], thread j owns column j
    const int64_t k0 = (int64_t)blockIdx.x * TB;
    const int cnt = (int)((d - k0) < TB ? (d - k0) : TB);
    const int j = threadIdx.x;
    for (int i = 0; i < cnt; ++i) {
        Ns[i * (TB + 1) + j] = (j < cnt) ? N[(k0 + i) * d + k0 + j] : 0.f;
        Xs[i * TB + j] = (i == j) ? 1.f : 0.f;
    }
    __syncthreads();
    // row i of column j:  x_i = -sum_{k > i} N[i][k] x_k   (x_k = 0 for k > j, so rows i >= j keep their identity value)
    for (int i = cnt - 2; i >= 0; --i) {
        float s = 0.f;
        for (int k = i + 1; k < cnt; ++k) s = fmaf(Ns[i * (TB + 1) + k], Xs[k * TB + j], s);
        if (i < j) Xs[i * TB + j] = -s;
    }
    if (j < cnt)
        for (int i = 0; i < cnt; ++i) X[(k0 + i) * d + k0 + j] = Xs[i * TB + j];
}

// PHASE 0:  Y[r0:r0+b, c0:c0+n2] =  N[r0:r0+b, c0:c0+n2] * X[c0:c0+n2, c0:c0+n2]      (X22 upper triangular)
// PHASE 1:  X[r0:r0+b, c0:c0+n2] = -X[r0:r0+b, r0:r0+b]  * Y[r0:r0+b, c0:c0+n2]      (X11 upper triangular)
// for pair p = blockIdx.z: r0 = 2 p b, c0 = r0 + b, n2 = min(b, d - c0).  256 threads, 2 x 2 waves of 64 x 64.
template <int PHASE>
__global__ __launch_bounds__(256) void trinv_gemm_kernel(const float *__restrict__ N, float *__restrict__ X, float *__restrict__ Y,
                                                         int64_t d, int64_t b)
{
    __shared__ __attribute__((aligned(16))) float As[TB * LDK];     // [row][k]
    __shared__ __attribute__((aligned(16))) float Bs[TB * LDK];     // [col][k]
    const int64_t r0 = 2 * (int64_t)blockIdx.z * b, c0 = r0 + b;
    const int64_t n2 = (d - c0) < b ? (d - c0) : b;
    const int ct = blockIdx.x, rt = blockIdx.y;
    if (n2 <= 0 || (int64_t)ct * TB >= n2) return;
    const float *A, *B;
    float *C;
    int64_t kbeg, kend;
    if (PHASE == 0) {
        A = N + r0 * d + c0;  B = X + c0 * d + c0;  C = Y + r0 * d + c0;
        kbeg = 0;  kend = ((int64_t)(ct + 1) * TB < n2) ? (int64_t)(ct + 1) * TB : n2;     // X22[k][c] = 0 for k > c
    } else {
        A = X + r0 * d + r0;  B = Y + r0 * d + c0;  C = X + r0 * d + c0;
        kbeg = (int64_t)rt * TB;  kend = b;                                                  // X11[r][k] = 0 for k < r
    }
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, kq = lane >> 4;
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int arow = t >> 1, ak = (t & 1) * 8;                 // A tile: 128 rows x 16 k
    const int bk = t >> 4, bcol = (t & 15) * 8;                // B tile: 16 k x 128 columns
    for (int64_t k0 = kbeg; k0 < kend; k0 += KS) {
        const float *ap = A + ((int64_t)rt * TB + arow) * d + k0 + ak;
        const float4 a0 = *reinterpret_cast<const float4 *>(ap), a1 = *reinterpret_cast<const float4 *>(ap + 4);
        float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
        const int64_t col = (int64_t)ct * TB + bcol;
        const float *bp = B + (k0 + bk) * d + col;
        if (col < n2) b0 = *reinterpret_cast<const float4 *>(bp);           // n2 is a multiple of 16
        if (col + 4 < n2) b1 = *reinterpret_cast<const float4 *>(bp + 4);
        __syncthreads();                                                      // previous stage's fragments are read
        *reinterpret_cast<float4 *>(&As[arow * LDK + ak]) = a0;
        *reinterpret_cast<float4 *>(&As[arow * LDK + ak + 4]) = a1;
        Bs[(bcol + 0) * LDK + bk] = b0.x;  Bs[(bcol + 1) * LDK + bk] = b0.y;
        Bs[(bcol + 2) * LDK + bk] = b0.z;  Bs[(bcol + 3) * LDK + bk] = b0.w;
        Bs[(bcol + 4) * LDK + bk] = b1.x;  Bs[(bcol + 5) * LDK + bk] = b1.y;
        Bs[(bcol + 6) * LDK + bk] = b1.z;  Bs[(bcol + 7) * LDK + bk] = b1.w;
        __syncthreads();
        float4 fa[4], fb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i] = *reinterpret_cast<const float4 *>(&As[(wr * 64 + i * 16 + fr) * LDK + 4 * kq]);
            fb[i] = *reinterpret_cast<const float4 *>(&Bs[(wc * 64 + i * 16 + fr) * LDK + 4 * kq]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int jn = 0; jn < 4; ++jn) {
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].x, fb[jn].x, acc[i][jn], 0, 0, 0);
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].y, fb[jn].y, acc[i][jn], 0, 0, 0);
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].z, fb[jn].z, acc[i][jn], 0, 0, 0);
                acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i].w, fb[jn].w, acc[i][jn], 0, 0, 0);
            }
    }
    // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 4; ++jn) {
            const int64_t col = (int64_t)ct * TB + wc * 64 + jn * 16 + fr;
            if (col >= n2) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t row = (int64_t)rt * TB + wr * 64 + i * 16 + 4 * kq + reg;
                C[row * d + col] = PHASE == 0 ? acc[i][jn][reg] : -acc[i][jn][reg];
            }
        }
}

__global__ __launch_bounds__(256) void trinv_finish_kernel(const float *__restrict__ X, float *__restrict__ FT, int64_t d)
{
    const int64_t i = blockIdx.x;
    for (int64_t j = threadIdx.x; j < d; j += 256) FT[i * d + j] = (j > i) ? -X[i * d + j] : 0.f;
}

}   // namespace

extern "C" int quipamd_unit_upper_inverse(const float *N, float *X, float *work, int64_t d, void *stream)
{
    QA_REQUIRE(d >= 0 && d % 16 == 0, QUIPAMD_ERR_SHAPE, "unit_upper_inverse: needs d %% 16 == 0 (d=%lld)", (long long)d);
    if (d == 0) return QUIPAMD_OK;
    QA_REQUIRE(N && X && work && N != X && work != X && work != N, QUIPAMD_ERR_ARG, "unit_upper_inverse: three distinct buffers wanted");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)(TB * (TB + 1) + TB * TB) * sizeof(float);
    static bool attr_set[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !attr_set[dev]) {
        if (hipFuncSetAttribute((const void *)trinv_diag_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "unit_upper_inverse: cannot raise dynamic LDS to %zu", lds);
        attr_set[dev] = true;
    }
    trinv_diag_kernel<<<(unsigned)((d + TB - 1) / TB), TB, lds, s>>>(N, X, d);
    for (int64_t b = TB; b < d; b *= 2) {
        const unsigned pairs = (unsigned)((d + 2 * b - 1) / (2 * b));
        dim3 grid((unsigned)(b / TB), (unsigned)(b / TB), pairs);
        trinv_gemm_kernel<0><<<grid, 256, 0, s>>>(N, X, work, d, b);
        trinv_gemm_kernel<1><<<grid, 256, 0, s>>>(N, X, work, d, b);
    }
    QA_LAUNCH_CHECK("unit_upper_inverse");
    return QUIPAMD_OK;
}

extern "C" int quipamd_gptq_feedback(const float *H, float *FT, float *work, int64_t d, int *info, void *stream)
{
    QA_REQUIRE(d >= 0 && d % 16 == 0, QUIPAMD_ERR_SHAPE, "gptq_feedback: needs d %% 16 == 0 (d=%lld)", (long long)d);
    if (d == 0) return QUIPAMD_OK;
    QA_REQUIRE(H && FT && work && info, QUIPAMD_ERR_ARG, "gptq_feedback: null pointer");
    hipStream_t s = (hipStream_t)stream;
    float *w0 = work, *w1 = work + (size_t)d * d;
    flip_kernel<<<(unsigned)d, 256, 0, s>>>(H, w0, d);
    int rc = quipamd_cholesky_lt(w0, w0, d, info, stream);                 // N, in place
    if (rc != QUIPAMD_OK) return rc;
    rc = quipamd_unit_upper_inverse(w0, w1, FT, d, stream);               // X in w1; FT doubles as the GEMM workspace
    if (rc != QUIPAMD_OK) return rc;
    trinv_finish_kernel<<<(unsigned)d, 256, 0, s>>>(w1, FT, d);
    QA_LAUNCH_CHECK("gptq_feedback");
    return QUIPAMD_OK;
}
