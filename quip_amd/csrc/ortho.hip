// ortho.hip -- K3: structured random-orthogonal apply (two-factor butterfly / Kronecker, `--pre_proj`).
//
// Takes over mul_ortho_butterfly (method.py:46-67) and with it the dense U @ W @ V^T / V @ H @ V^T products
// of QuantMethod.preproc / postproc (method.py:175-176,202-203): the operator is applied in its factored
// form, never materialised (2 n c (p+q) flops instead of 2 n^2 c).
//
// For one length-n vector v (n = p*q), with z viewed as [p][q] (index i = a*q + b):
//   forward   z = v[perm_in];  z[a][b] <- sum_a' B0[b][a][a'] z[a'][b];  z[a][b] <- sum_b' B1[a][b][b'] z[a][b'];
//             out = z[perm_out]
//   transpose z[perm_out] = v;  z[a][b] <- sum_b' B1[a][b'][b] z[a][b'];  z[a][b] <- sum_a' B0[b][a'][a] z[a'][b];
//             out[perm_in] = z
// (index form verified against the reference in tests/golden/butterfly.npz through the oracle).
//
// Kernel: one workgroup owns RT whole rows in LDS (fp32), two ping-pong images per row:
//   image A: element (a,b) at a*QS + b  (QS = q|1)   -- read by "mix over a" with lanes along b (conflict free)
//   image B: element (a,b) at b*PS + a  (PS = p|1)   -- read by "mix over b" with lanes along a (conflict free)
// Each thread register-blocks 4 outputs x RT rows, so one factor value (global, L2 resident, coalesced along
// the lane axis) feeds RT FMAs and one LDS read feeds 4.  Permutations are applied as an LDS scatter on the
// coalesced global load and an LDS gather on the coalesced global store.
// Algorithmic bytes: rows*n*(in+out element size) (+ factors, n*(p+q)*4 blocked); FLOPs: 2*rows*n*(p+q).
#include "common.h"

namespace {

constexpr int OB = 4;   // outputs register-blocked per thread

struct OrthoArgs {
    const float *F0;        // blocked: [p][p][q]; kron: [p][p]
    const float *F1;        // blocked: [q][q][p]; kron: [q][q]
    const int32_t *load_idx;    // flat z index that input element k lands on
    const int32_t *store_idx;   // flat z index that output element k is taken from
    const float *colscale;      // [n] or null
    int p, q, blocked, transpose;
    int64_t ldx, ldo, rows;
};

template <int RT>
__device__ __forceinline__ void mix_a(const OrthoArgs &A, const float *__restrict__ src, float *__restrict__ dst,
                                      int rowstride, int QS, int PS, bool dst_is_B)
{
    // out[a][b] = sum_a' F(a,a',b) in[a'][b];  src is image A.  forward: F = B0[b][a][a'], transpose: B0[b][a'][a]
    const int p = A.p, q = A.q;
    const int fs = A.blocked ? q : 1, fb = A.blocked ? 1 : 0;
    const int nblk = (p + OB - 1) / OB;
    for (int task = threadIdx.x; task < q * nblk; task += blockDim.x) {
        const int b = task % q, a0 = (task / q) * OB;
        float acc[OB][RT];
#pragma unroll
        for (int k = 0; k < OB; ++k)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[k][r] = 0.f;
        for (int ap = 0; ap < p; ++ap) {
            float f[OB];
#pragma unroll
            for (int k = 0; k < OB; ++k) {
                const int a = a0 + k < p ? a0 + k : p - 1;
                const int64_t fi = A.transpose ? ((int64_t)ap * p + a) : ((int64_t)a * p + ap);
                f[k] = A.F0[fi * fs + (int64_t)b * fb];
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float z = src[r * rowstride + ap * QS + b];
#pragma unroll
                for (int k = 0; k < OB; ++k) acc[k][r] = fmaf(f[k], z, acc[k][r]);
            }
        }
#pragma unroll
        for (int k = 0; k < OB; ++k) {
            const int a = a0 + k;
            if (a < p) {
#pragma unroll
                for (int r = 0; r < RT; ++r) dst[r * rowstride + (dst_is_B ? b * PS + a : a * QS + b)] = acc[k][r];
            }
        }
    }
}

template <int RT>
__device__ __forceinline__ void mix_b(const OrthoArgs &A, const float *__restrict__ src, float *__restrict__ dst,
                                      int rowstride, int QS, int PS, bool dst_is_B)
{
    // out[a][b] = sum_b' F(b,b',a) in[a][b'];  src is image B.  forward: F = B1[a][b][b'], transpose: B1[a][b'][b]
    const int p = A.p, q = A.q;
    const int fs = A.blocked ? p : 1, fb = A.blocked ? 1 : 0;
    const int nblk = (q + OB - 1) / OB;
    for (int task = threadIdx.x; task < p * nblk; task += blockDim.x) {
        const int a = task % p, b0 = (task / p) * OB;
        float acc[OB][RT];
#pragma unroll
        for (int k = 0; k < OB; ++k)
#pragma unroll
            for (int r = 0; r < RT; ++r) acc[k][r] = 0.f;
        for (int bp = 0; bp < q; ++bp) {
            float f[OB];
#pragma unroll
            for (int k = 0; k < OB; ++k) {
                const int b = b0 + k < q ? b0 + k : q - 1;
                const int64_t fi = A.transpose ? ((int64_t)bp * q + b) : ((int64_t)b * q + bp);
                f[k] = A.F1[fi * fs + (int64_t)a * fb];
            }
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float z = src[r * rowstride + bp * PS + a];
#pragma unroll
                for (int k = 0; k < OB; ++k) acc[k][r] = fmaf(f[k], z, acc[k][r]);
            }
        }
#pragma unroll
        for (int k = 0; k < OB; ++k) {
            const int b = b0 + k;
            if (b < q) {
#pragma unroll
                for (int r = 0; r < RT; ++r) dst[r * rowstride + (dst_is_B ? b * PS + a : a * QS + b)] = acc[k][r];
            }
        }
    }
}

template <class TI, class TO, int RT>
__global__ __launch_bounds__(256) void ortho_rows_kernel(OrthoArgs A, const void *__restrict__ x, void *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int p = A.p, q = A.q, n = p * q;
    const int QS = q | 1, PS = p | 1;
    const int imgA = p * QS, imgB = q * PS;
    const int rowstride = imgA > imgB ? imgA : imgB;
    float *buf0 = smem, *buf1 = smem + RT * rowstride;
    const int64_t row0 = (int64_t)blockIdx.x * RT;

    // load: forward lands in image A (mix over a comes first), transpose in image B (mix over b first)
    for (int idx = threadIdx.x; idx < n * RT; idx += blockDim.x) {
        const int r = idx / n, k = idx - r * n;
        const int64_t row = row0 + r;
        float v = 0.f;
        if (row < A.rows) {
            v = DT<TI>::load(x, row * A.ldx + k);
            if (A.colscale) v *= A.colscale[k];
        }
        const int i = A.load_idx[k];
        const int a = i / q, b = i - a * q;
        buf0[r * rowstride + (A.transpose ? b * PS + a : a * QS + b)] = v;
    }
    __syncthreads();
    if (!A.transpose) {
        mix_a<RT>(A, buf0, buf1, rowstride, QS, PS, true);     // A -> B
        __syncthreads();
        mix_b<RT>(A, buf1, buf0, rowstride, QS, PS, false);    // B -> A
    } else {
        mix_b<RT>(A, buf0, buf1, rowstride, QS, PS, false);    // B -> A
        __syncthreads();
        mix_a<RT>(A, buf1, buf0, rowstride, QS, PS, true);     // A -> B
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < n * RT; idx += blockDim.x) {
        const int r = idx / n, k = idx - r * n;
        const int64_t row = row0 + r;
        if (row >= A.rows) continue;
        const int i = A.store_idx[k];
        const int a = i / q, b = i - a * q;
        const float v = buf0[r * rowstride + (A.transpose ? b * PS + a : a * QS + b)];
        DT<TO>::store(out, row * A.ldo + k, v);
    }
}

template <class TI, class TO>
int launch_rows(const OrthoArgs &A, const void *x, void *out, hipStream_t s)
{
    const int p = A.p, q = A.q;
    const int64_t imgA = (int64_t)p * (q | 1), imgB = (int64_t)q * (p | 1);
    const int64_t rowbytes = 2 * (imgA > imgB ? imgA : imgB) * 4;
    const int64_t budget = 160 * 1024;
    QA_REQUIRE(rowbytes <= budget, QUIPAMD_ERR_SHAPE,
               "ortho_apply_rows: n=%d needs %lld B of LDS per row (> 160 KiB)", p * q, (long long)rowbytes);
    int rt = 8;
    while (rt > 1 && rt * rowbytes > budget) rt >>= 1;                       // rows that fit in 160 KiB of LDS
    while (rt > 1 && (A.rows + rt - 1) / rt < 256) rt >>= 1;                  // but keep >= 256 workgroups if we can
    const int64_t grid = (A.rows + rt - 1) / rt;
    const size_t lds = (size_t)(rt * rowbytes);
#define QA_ORTHO_LAUNCH(RT)                                                                                   \
    do {                                                                                                      \
        auto kern = ortho_rows_kernel<TI, TO, RT>;                                                            \
        if (lds > 64 * 1024)                                                                                   \
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return qa_fail(QUIPAMD_ERR_LAUNCH, "ortho: cannot raise dynamic LDS to %zu", lds);             \
        kern<<<(unsigned)grid, 256, lds, s>>>(A, x, out);                                                      \
    } while (0)
    switch (rt) {
    case 8: QA_ORTHO_LAUNCH(8); break;
    case 4: QA_ORTHO_LAUNCH(4); break;
    case 2: QA_ORTHO_LAUNCH(2); break;
    default: QA_ORTHO_LAUNCH(1); break;
    }
#undef QA_ORTHO_LAUNCH
    QA_LAUNCH_CHECK("quipamd_ortho_apply_rows");
    return QUIPAMD_OK;
}

}   // namespace

extern "C" int quipamd_ortho_apply_rows(const float *B0t, const float *B1t, int blocked, const int32_t *load_idx,
                                        const int32_t *store_idx, int p, int q, int transpose, const float *colscale,
                                        const void *x, int x_dtype, int64_t ldx, void *out, int out_dtype, int64_t ldo,
                                        int64_t rows, void *stream)
{
    QA_REQUIRE(B0t && B1t && load_idx && store_idx && x && out, QUIPAMD_ERR_ARG, "ortho_apply_rows: null pointer");
    QA_REQUIRE(p >= 1 && q >= 1, QUIPAMD_ERR_SHAPE, "ortho_apply_rows: bad factors p=%d q=%d", p, q);
    QA_REQUIRE(ldx >= (int64_t)p * q && ldo >= (int64_t)p * q, QUIPAMD_ERR_SHAPE, "ortho_apply_rows: leading dimension < n");
    if (rows == 0) return QUIPAMD_OK;
    OrthoArgs A;
    A.F0 = B0t; A.F1 = B1t; A.load_idx = load_idx; A.store_idx = store_idx; A.colscale = colscale;
    A.p = p; A.q = q; A.blocked = blocked; A.transpose = transpose; A.ldx = ldx; A.ldo = ldo; A.rows = rows;
    hipStream_t s = (hipStream_t)stream;
    if (x_dtype == QUIPAMD_F32 && out_dtype == QUIPAMD_F32) return launch_rows<F32, F32>(A, x, out, s);
    if (x_dtype == QUIPAMD_BF16 && out_dtype == QUIPAMD_BF16) return launch_rows<BF16, BF16>(A, x, out, s);
    if (x_dtype == QUIPAMD_F16 && out_dtype == QUIPAMD_F16) return launch_rows<F16, F16>(A, x, out, s);
    if (x_dtype == QUIPAMD_F16 && out_dtype == QUIPAMD_BF16) return launch_rows<F16, BF16>(A, x, out, s);
    if (x_dtype == QUIPAMD_BF16 && out_dtype == QUIPAMD_F16) return launch_rows<BF16, F16>(A, x, out, s);
    if (x_dtype == QUIPAMD_F32 && out_dtype == QUIPAMD_BF16) return launch_rows<F32, BF16>(A, x, out, s);
    if (x_dtype == QUIPAMD_BF16 && out_dtype == QUIPAMD_F32) return launch_rows<BF16, F32>(A, x, out, s);
    if (x_dtype == QUIPAMD_F32 && out_dtype == QUIPAMD_F16) return launch_rows<F32, F16>(A, x, out, s);
    if (x_dtype == QUIPAMD_F16 && out_dtype == QUIPAMD_F32) return launch_rows<F16, F32>(A, x, out, s);
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "ortho_apply_rows: dtype pair %d -> %d", x_dtype, out_dtype);
}
