// hessian.hip -- K7: Hessian accumulation  Hacc += X^T X  in fp64, and its finish  H = fp32(sym(Hacc) / nsamples)
//
// Replaces QuantMethod.add_batch / post_batch (reference method.py:98-123): the reference casts the layer input
// [tokens, d] to fp64 and adds the full d x d product X^T X with a dense fp64 GEMM, once per calibration sample
// (opt.py:141-143) -- 2 * tokens * d^2 flops per call, 35 TFLOP of fp64 per OPT-1.3B fc2.
//
// gfx950 design
//   * X^T X is symmetric: only the block-lower triangle (tiles I >= J) is computed -- half the flops of the GEMM.
//     The accumulator Hacc holds valid data in those tiles only; quipamd_hessian_finish mirrors it while it divides
//     by nsamples and narrows to fp32 (the reference's post_batch), so the mirror costs no extra pass.
//   * the products run on the fp64 matrix pipe (v_mfma_f64_16x16x4_f64): fp16/bf16/fp32 inputs widen to fp64 exactly,
//     every product and every accumulation is an fp64 fma like the reference's dgemm -- same arithmetic, different
//     summation order only (differences ~1e-16 relative, invisible after the fp32 narrowing except on rounding ties).
//   * both MFMA operands are "row = token, 16 consecutive columns" fragments of the SAME matrix X, so the token-major
//     activations need no transpose: a workgroup stages KT=32 tokens x BN columns of the I side and of the J side in
//     LDS as fp32 (f16 / bf16 widen to it exactly), double buffered, one barrier per stage; a fragment element widens
//     to fp64 (one v_cvt_f64_f32, VALU under the 64-cycle MFMAs) when it is read.  fp32 instead of fp64 staging
//     halves the LDS writes and lets a stage hold twice the tokens in the same 72 KiB -- half the barriers: 2.41 ->
//     2.19 ms at 2048 x 8192 (the loop with staging and barriers compiled out runs 2.04 ms).
//     Row stride = BN*4 + 64 B puts the four 16-lane token groups of a ds_read_b32 on four different bank quarters.
//     The global loads of stage c+1 are issued before the MFMAs of stage c and stay packed dwords until they are
//     written to LDS after them, so their latency is hidden (unpacking at the load made hipcc wait on the spot).
//   * 4 waves = 2 x 2, each owns WT x WT accumulator tiles (WT=4: 128 x 128 workgroup tile, 128 accumulator
//     registers; WT=2 / 1: 64 / 32 columns for small d so the triangle still fills 256 CUs x 2 workgroups).
//   * the triangle rarely divides by the 512 workgroup slots (d=8192: 2080 tiles = 4 x 512 + 32, a fifth round at 6 %
//     occupancy): the remainder tiles are cut into quarter tiles scheduled last, so the tail is a quarter as long.
//   * no split over tokens, no atomics: the result is deterministic.
#include "common.h"

typedef __attribute__((ext_vector_type(4))) double f64x4_t;

namespace {

constexpr int KT = 32;    // tokens per LDS stage
constexpr int NPASS = KT / 16;   // a workgroup loads 16 token rows per pass (16 threads per row)

// element e of a run held as packed 32-bit words (the registers stay whole dwords until the LDS store, so nothing
// has to touch -- and wait for -- the loaded data before the MFMAs of the current stage); f16 / bf16 -> f32 is exact
template <class TI> __device__ __forceinline__ float widen(const uint32_t *w, int e);
template <> __device__ __forceinline__ float widen<F32>(const uint32_t *w, int e) { return __uint_as_float(w[e]); }
template <> __device__ __forceinline__ float widen<F16>(const uint32_t *w, int e)
{
    return f16_bits_to_f32((uint16_t)(w[e >> 1] >> (16 * (e & 1))));
}
template <> __device__ __forceinline__ float widen<BF16>(const uint32_t *w, int e)
{
    return __uint_as_float((e & 1) ? (w[e >> 1] & 0xffff0000u) : (w[e >> 1] << 16));
}

// EPT consecutive columns of one token row, still in the storage type: the widening to fp64 happens when the stage is
// written to LDS, AFTER the MFMAs of the current stage, so the load's latency hides under them (converting at the load
// made the compiler wait for it on the spot).  VEC (16-byte aligned rows, d % 8 == 0): one unconditional vector load
// from a clamped address, zeroed later if it was out of range -- no branch between the load and the MFMAs.
template <class TI, int EPT, bool VEC>
__device__ __forceinline__ bool load_run(const typename DT<TI>::storage *X, int64_t ldx, int64_t tok, int64_t tokens,
                                         int64_t col, int64_t d, uint32_t (&raw)[EPT * sizeof(typename DT<TI>::storage) / 4])
{
    typedef typename DT<TI>::storage S;
    constexpr int NW = EPT * (int)sizeof(S) / 4;
    const bool ok = tok < tokens && col < d;
    if constexpr (VEC) {
        const S *p = X + (ok ? tok * ldx + col : 0);
        if constexpr (NW == 1) {
            raw[0] = *reinterpret_cast<const uint32_t *>(p);
        } else if constexpr (NW == 2) {
            const uint2 v = *reinterpret_cast<const uint2 *>(p);
            raw[0] = v.x; raw[1] = v.y;
        } else {
#pragma unroll
            for (int b = 0; b < NW / 4; ++b) {
                const uint4 v = reinterpret_cast<const uint4 *>(p)[b];
                raw[4 * b] = v.x; raw[4 * b + 1] = v.y; raw[4 * b + 2] = v.z; raw[4 * b + 3] = v.w;
            }
        }
        return ok;
    } else {
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            if constexpr (sizeof(S) == 4) {
                raw[w] = (ok && col + w < d) ? reinterpret_cast<const uint32_t *>(X)[tok * ldx + col + w] : 0u;
            } else {
                const uint32_t lo = (ok && col + 2 * w < d) ? (uint32_t)X[tok * ldx + col + 2 * w] : 0u;
                const uint32_t hi = (ok && col + 2 * w + 1 < d) ? (uint32_t)X[tok * ldx + col + 2 * w + 1] : 0u;
                raw[w] = lo | (hi << 16);
            }
        }
        return true;
    }
}

// lower-triangle tile (I >= J) from a linear index: t = I (I + 1) / 2 + J
__device__ __forceinline__ void tri_tile(int t, int &I, int &J)
{
    I = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while (I * (I + 1) / 2 > t) --I;
    while ((I + 1) * (I + 2) / 2 <= t) ++I;
    J = t - I * (I + 1) / 2;
}

// one (32 WT) x (32 WT) tile of Hacc, tile coordinates (I, J) in units of 32 WT columns
template <class TI, int WT, bool VEC>
__device__ __forceinline__ void tile_body(const typename DT<TI>::storage *X, int64_t ldx, int64_t tokens, int64_t d, double *H,
                                          int I, int J, float *hs)
{
    constexpr int BN = 32 * WT, LDW = BN + 16, EPT = BN / 16, NW = EPT * (int)sizeof(typename DT<TI>::storage) / 4;
    const bool diag = I == J;
    const int64_t i0 = (int64_t)I * BN, j0 = (int64_t)J * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int stok = tid >> 4, scol = (tid & 15) * EPT;

    uint32_t ri[NPASS][NW], rj[NPASS][NW];
    bool oki[NPASS], okj[NPASS];
    auto gload = [&](int64_t t0) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            oki[ps] = load_run<TI, EPT, VEC>(X, ldx, t0 + 16 * ps + stok, tokens, i0 + scol, d, ri[ps]);
            if (!diag) okj[ps] = load_run<TI, EPT, VEC>(X, ldx, t0 + 16 * ps + stok, tokens, j0 + scol, d, rj[ps]);
        }
    };
    auto put = [&](float *dst, const uint32_t *raw, bool ok) {
        if constexpr (EPT >= 4) {
#pragma unroll
            for (int e = 0; e < EPT; e += 4)
                *reinterpret_cast<float4 *>(dst + e) = ok ? make_float4(widen<TI>(raw, e), widen<TI>(raw, e + 1), widen<TI>(raw, e + 2), widen<TI>(raw, e + 3))
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            *reinterpret_cast<float2 *>(dst) = ok ? make_float2(widen<TI>(raw, 0), widen<TI>(raw, 1)) : make_float2(0.f, 0.f);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            put(hs + ((buf * 2 + 0) * KT + 16 * ps + stok) * LDW + scol, ri[ps], oki[ps]);
            if (!diag) put(hs + ((buf * 2 + 1) * KT + 16 * ps + stok) * LDW + scol, rj[ps], okj[ps]);
        }
    };

    f64x4_t acc[WT][WT];
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y) acc[x][y] = f64x4_t{0.0, 0.0, 0.0, 0.0};

    const int64_t nchunks = (tokens + KT - 1) / KT;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int64_t c = 0; c < nchunks; ++c) {
        const int cur = (int)(c & 1);
        const bool more = c + 1 < nchunks;
        if (more) gload((c + 1) * KT);
        // A[row = column of the I side][k = token], B[k = token][col = column of the J side]: lane (l & 15, l >> 4)
        const float *As = hs + (cur * 2 + 0) * KT * LDW + wi * (WT * 16) + (lane & 15);
        const float *Bs = hs + (cur * 2 + (diag ? 0 : 1)) * KT * LDW + wj * (WT * 16) + (lane & 15);
#pragma unroll
        for (int ks = 0; ks < KT / 4; ++ks) {
            const int row = ks * 4 + (lane >> 4);
            double a[WT], b[WT];
#pragma unroll
            for (int x = 0; x < WT; ++x) a[x] = (double)As[row * LDW + x * 16];
#pragma unroll
            for (int y = 0; y < WT; ++y) b[y] = (double)Bs[row * LDW + y * 16];
#pragma unroll
            for (int x = 0; x < WT; ++x)
#pragma unroll
                for (int y = 0; y < WT; ++y)
                    acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
        }
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }

    // D layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t r = i0 + wi * (WT * 16) + x * 16 + (lane >> 4) + 4 * reg;
                const int64_t cidx = j0 + wj * (WT * 16) + y * 16 + (lane & 15);
                if (r < d && cidx < d) H[r * d + cidx] += acc[x][y][reg];
            }
}

// Workgroups [0, nbig) own whole (32 WT)^2 tiles; the tiles past nbig -- the remainder that would otherwise run as a
// nearly empty last round on the 256 CUs x 2 workgroup slots -- are cut into four quarter tiles each, scheduled last
// (longest-processing-time-first: the tail of the launch is a quarter as long).  The upper quarter of a diagonal tile is
// never read by quipamd_hessian_finish and is skipped.
template <class TI, int WT, bool VEC>
__global__ __launch_bounds__(256, 2) void hsyrk_kernel(const typename DT<TI>::storage *X, int64_t ldx, int64_t tokens,
                                                      int64_t d, double *H, int nbig)
{
    extern __shared__ __attribute__((aligned(16))) float hs[];         // [2 buffers][2 sides][KT][LDW]
    const int b = blockIdx.x;
    int I, J;
    if (b < nbig) {
        tri_tile(b, I, J);
        tile_body<TI, WT, VEC>(X, ldx, tokens, d, H, I, J, hs);
    } else if constexpr (WT > 1) {
        const int r = b - nbig;
        tri_tile(nbig + (r >> 2), I, J);
        const int si = (r >> 1) & 1, sj = r & 1;
        if (I == J && si == 0 && sj == 1) return;
        if ((int64_t)(2 * I + si) * (16 * WT) >= d || (int64_t)(2 * J + sj) * (16 * WT) >= d) return;
        tile_body<TI, WT / 2, VEC>(X, ldx, tokens, d, H, 2 * I + si, 2 * J + sj, hs);
    }
}

// H[i][j] = fp32( Hacc[max(i,j)][min(i,j)] / nsamples ): 32 x 32 tiles, upper tiles read their mirror through LDS
__global__ __launch_bounds__(256) void hfinish_kernel(const double *Hacc, double n, float *out, int64_t d)
{
    __shared__ double tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;             // column tile, row tile
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    if (by >= bx) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t r = (int64_t)by * 32 + ty + 8 * k, c = (int64_t)bx * 32 + tx;
            if (r < d && c < d) out[r * d + c] = (float)(Hacc[r * d + c] / n);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {                            // mirror tile: rows of column-tile bx, columns of row-tile by
        const int64_t r = (int64_t)bx * 32 + ty + 8 * k, c = (int64_t)by * 32 + tx;
        tile[ty + 8 * k][tx] = (r < d && c < d) ? Hacc[r * d + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t r = (int64_t)by * 32 + ty + 8 * k, c = (int64_t)bx * 32 + tx;
        if (r < d && c < d) out[r * d + c] = (float)(tile[tx][ty + 8 * k] / n);
    }
}

template <class TI, int WT, bool VEC>
int launch_syrk(const void *x, int64_t ldx, int64_t tokens, int64_t d, double *H, hipStream_t s)
{
    constexpr int BN = 32 * WT, LDW = BN + 16, SLOTS = 512;               // 256 CUs x 2 resident workgroups
    const size_t lds = (size_t)2 * 2 * KT * LDW * sizeof(float);
    const int64_t T = (d + BN - 1) / BN, N = T * (T + 1) / 2;
    int64_t nbig = N;
    if (WT > 1 && N > SLOTS && N % SLOTS) nbig = N / SLOTS * SLOTS;
    auto kern = hsyrk_kernel<TI, WT, VEC>;
    static QaPerDevice attr_done_dev;                         // per instantiation
    const int attr_done_d = attr_done_dev.dev();
    if ((attr_done_d < 0 || !attr_done_dev.done[attr_done_d])) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "hessian_accum: cannot reserve %zu B of LDS", lds);
        if (attr_done_d >= 0) attr_done_dev.done[attr_done_d] = true;
    }
    kern<<<(unsigned)(nbig + 4 * (N - nbig)), 256, lds, s>>>((const typename DT<TI>::storage *)x, ldx, tokens, d, H, (int)nbig);
    QA_LAUNCH_CHECK("hessian_accum");
    return QUIPAMD_OK;
}

// largest tile whose triangle still gives every CU work: 128 (WT 4), 64 (WT 2) or 32 (WT 1) columns
template <class TI, bool VEC>
int pick_syrk(const void *x, int64_t ldx, int64_t tokens, int64_t d, double *H, hipStream_t s)
{
    auto ntiles = [&](int64_t bn) { const int64_t T = (d + bn - 1) / bn; return T * (T + 1) / 2; };
    if (ntiles(128) >= 384) return launch_syrk<TI, 4, VEC>(x, ldx, tokens, d, H, s);
    if (ntiles(64) >= 384) return launch_syrk<TI, 2, VEC>(x, ldx, tokens, d, H, s);
    return launch_syrk<TI, 1, VEC>(x, ldx, tokens, d, H, s);
}

}   // namespace

extern "C" int quipamd_hessian_accum(const void *x, int x_dtype, int64_t ldx, int64_t tokens, int64_t d, double *Hacc,
                                     void *stream)
{
    QA_REQUIRE(tokens >= 0 && d >= 0 && ldx >= d, QUIPAMD_ERR_SHAPE, "hessian_accum: bad shape tokens=%lld d=%lld ldx=%lld",
               (long long)tokens, (long long)d, (long long)ldx);
    if (tokens == 0 || d == 0) return QUIPAMD_OK;
    QA_REQUIRE(x && Hacc, QUIPAMD_ERR_ARG, "hessian_accum: null pointer");
    QA_REQUIRE(d <= (1 << 20), QUIPAMD_ERR_SHAPE, "hessian_accum: d too large");
    const int esz = x_dtype == QUIPAMD_F32 ? 4 : 2;
    const bool vec = ((uintptr_t)x & 15) == 0 && (ldx * esz) % 16 == 0 && d % 8 == 0;
    hipStream_t s = (hipStream_t)stream;
    QA_DISPATCH_DTYPE(x_dtype, TI, return vec ? pick_syrk<TI, true>(x, ldx, tokens, d, Hacc, s)
                                              : pick_syrk<TI, false>(x, ldx, tokens, d, Hacc, s));
    return QUIPAMD_OK;
}

extern "C" int quipamd_hessian_finish(const double *Hacc, double nsamples, float *H, int64_t d, void *stream)
{
    QA_REQUIRE(d >= 0, QUIPAMD_ERR_SHAPE, "hessian_finish: bad d");
    if (d == 0) return QUIPAMD_OK;
    QA_REQUIRE(Hacc && H, QUIPAMD_ERR_ARG, "hessian_finish: null pointer");
    QA_REQUIRE((const void *)Hacc != (const void *)H, QUIPAMD_ERR_ARG, "hessian_finish: in-place not supported");
    const unsigned nt = (unsigned)((d + 31) / 32);
    hfinish_kernel<<<dim3(nt, nt), 256, 0, (hipStream_t)stream>>>(Hacc, nsamples, H, d);
    QA_LAUNCH_CHECK("hessian_finish");
    return QUIPAMD_OK;
}

// =====================================================================================================================
// Opt-in fast mode: exact products on the 16-bit matrix pipe, fp32 partial sums over FLUSH tokens, fp64 across them.
// f16 x f16 and bf16 x bf16 products are exact in fp32, so the only rounding that the fp64 path does not have is the
// fp32 accumulation inside a FLUSH-token run (|err| <= FLUSH * 2^-24 * sum|x_i x_j| worst case, ~1e-7 of the run
// typically); the runs are then summed in fp64, which averages those errors down: ~1e-9 of sqrt(H_ii H_jj) over a
// 262144-token calibration pass -- below the fp32 narrowing of post_batch (6e-8), but NOT the reference's arithmetic,
// hence opt-in (quip_amd.method.HESSIAN_FAST).  v_mfma_f32_16x16x32 wants 8 consecutive k (= tokens) per lane for a
// fixed output row (= column of X): X is transposed once per call into the caller's workspace (tokens padded to 32
// with zeros), after which a stage row is 64 contiguous bytes and fragments are single ds_read_b128.
namespace {

constexpr int FKT = 32;          // tokens per stage (one MFMA k-step)
constexpr int FRS = 40;          // LDS row stride in 16-bit elements: 64 B of tokens + 16 B pad (conflict-free b128 reads)
constexpr int FLUSH = 128;       // tokens per fp32 run

template <class TI> struct Mfma16;
template <> struct Mfma16<F16> {
    typedef _Float16 __attribute__((ext_vector_type(8))) frag;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct Mfma16<BF16> {
    typedef bf16x8_t frag;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// Xt[c][t] = X[t][c] for t < tokens, 0 for tokens <= t < tpad; 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void htranspose_kernel(const uint16_t *X, int64_t ldx, int64_t tokens, int64_t d, uint16_t *Xt,
                                                        int64_t tpad)
{
    __shared__ uint16_t tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t t0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t t = t0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (t < tokens && c < d) ? X[t * ldx + c] : (uint16_t)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t c = c0 + ty + 8 * k, t = t0 + tx;
        if (c < d && t < tpad) Xt[c * tpad + t] = tile[tx][ty + 8 * k];
    }
}

template <class TI, int WT>
__global__ __launch_bounds__(256, 2) void hsyrk_fast_kernel(const uint16_t *Xt, int64_t tpad, int64_t d, double *H)
{
    constexpr int BN = 32 * WT;
    typedef typename Mfma16<TI>::frag frag;
    extern __shared__ __attribute__((aligned(16))) uint16_t fs[];     // [2 buffers][2 sides][BN][FRS]
    int I, J;
    tri_tile(blockIdx.x, I, J);
    const bool diag = I == J;
    const int64_t i0 = (int64_t)I * BN, j0 = (int64_t)J * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;

    // staging: a side is BN rows x 4 chunks of 16 B; thread -> chunks tid, tid + 256, ... (BN * 4 / 256 = WT / 2 ... >= 1)
    constexpr int CH = BN * 4, PER = (CH + 255) / 256;
    uint4 ri[PER], rj[PER];
    auto gload = [&](int64_t t0) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + 256 * u, row = idx >> 2, ch = idx & 3;
            ri[u] = make_uint4(0, 0, 0, 0);
            rj[u] = make_uint4(0, 0, 0, 0);
            if (idx < CH) {
                if (i0 + row < d) ri[u] = *reinterpret_cast<const uint4 *>(Xt + (i0 + row) * tpad + t0 + 8 * ch);
                if (!diag && j0 + row < d) rj[u] = *reinterpret_cast<const uint4 *>(Xt + (j0 + row) * tpad + t0 + 8 * ch);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int idx = tid + 256 * u, row = idx >> 2, ch = idx & 3;
            if (idx < CH) {
                *reinterpret_cast<uint4 *>(fs + ((buf * 2 + 0) * BN + row) * FRS + 8 * ch) = ri[u];
                if (!diag) *reinterpret_cast<uint4 *>(fs + ((buf * 2 + 1) * BN + row) * FRS + 8 * ch) = rj[u];
            }
        }
    };

    f32x4_t acc[WT][WT];
    f64x4_t sum[WT][WT];
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y) {
            acc[x][y] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            sum[x][y] = f64x4_t{0.0, 0.0, 0.0, 0.0};
        }
    auto flush = [&]() {
#pragma unroll
        for (int x = 0; x < WT; ++x)
#pragma unroll
            for (int y = 0; y < WT; ++y) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[x][y][r] += (double)acc[x][y][r];
                acc[x][y] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
    };

    const int64_t nst = tpad / FKT;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int64_t c = 0; c < nst; ++c) {
        const int cur = (int)(c & 1);
        const bool more = c + 1 < nst;
        if (more) gload((c + 1) * FKT);
        const uint16_t *As = fs + ((cur * 2 + 0) * BN + wi * (WT * 16) + (lane & 15)) * FRS + 8 * (lane >> 4);
        const uint16_t *Bs = fs + ((cur * 2 + (diag ? 0 : 1)) * BN + wj * (WT * 16) + (lane & 15)) * FRS + 8 * (lane >> 4);
        frag a[WT], b[WT];
#pragma unroll
        for (int x = 0; x < WT; ++x) a[x] = *reinterpret_cast<const frag *>(As + x * 16 * FRS);
#pragma unroll
        for (int y = 0; y < WT; ++y) b[y] = *reinterpret_cast<const frag *>(Bs + y * 16 * FRS);
#pragma unroll
        for (int x = 0; x < WT; ++x)
#pragma unroll
            for (int y = 0; y < WT; ++y) acc[x][y] = Mfma16<TI>::run(a[x], b[y], acc[x][y]);
        if (((c + 1) % (FLUSH / FKT)) == 0) flush();
        if (more) sstore(cur ^ 1);
        __syncthreads();
    }
    flush();
    // D layout of the f32 MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t r = i0 + wi * (WT * 16) + x * 16 + 4 * (lane >> 4) + reg;
                const int64_t cidx = j0 + wj * (WT * 16) + y * 16 + (lane & 15);
                if (r < d && cidx < d) H[r * d + cidx] += sum[x][y][reg];
            }
}

template <class TI, int WT> int launch_fast(const uint16_t *Xt, int64_t tpad, int64_t d, double *H, hipStream_t s)
{
    constexpr int BN = 32 * WT;
    const size_t lds = (size_t)2 * 2 * BN * FRS * sizeof(uint16_t);
    const int64_t T = (d + BN - 1) / BN;
    hsyrk_fast_kernel<TI, WT><<<(unsigned)(T * (T + 1) / 2), 256, lds, s>>>(Xt, tpad, d, H);
    QA_LAUNCH_CHECK("hessian_accum_fast");
    return QUIPAMD_OK;
}

template <class TI> int pick_fast(const uint16_t *Xt, int64_t tpad, int64_t d, double *H, hipStream_t s)
{
    auto ntiles = [&](int64_t bn) { const int64_t T = (d + bn - 1) / bn; return T * (T + 1) / 2; };
    if (ntiles(128) >= 384) return launch_fast<TI, 4>(Xt, tpad, d, H, s);
    if (ntiles(64) >= 384) return launch_fast<TI, 2>(Xt, tpad, d, H, s);
    return launch_fast<TI, 1>(Xt, tpad, d, H, s);
}

}   // namespace

extern "C" int64_t quipamd_hessian_fast_workspace(int64_t tokens, int64_t d)
{
    return d * ((tokens + FKT - 1) / FKT * FKT);          // elements of x's dtype
}

extern "C" int quipamd_hessian_accum_fast(const void *x, int x_dtype, int64_t ldx, int64_t tokens, int64_t d, double *Hacc,
                                          void *workspace, void *stream)
{
    QA_REQUIRE(tokens >= 0 && d >= 0 && ldx >= d, QUIPAMD_ERR_SHAPE, "hessian_accum_fast: bad shape");
    if (tokens == 0 || d == 0) return QUIPAMD_OK;
    QA_REQUIRE(x && Hacc && workspace, QUIPAMD_ERR_ARG, "hessian_accum_fast: null pointer");
    QA_REQUIRE(x_dtype == QUIPAMD_F16 || x_dtype == QUIPAMD_BF16, QUIPAMD_ERR_UNSUPPORTED,
               "hessian_accum_fast: f16 / bf16 inputs only (f32 products are not exact in fp32)");
    QA_REQUIRE(((uintptr_t)workspace & 15) == 0, QUIPAMD_ERR_ARG, "hessian_accum_fast: workspace must be 16-byte aligned");
    QA_REQUIRE(d <= (1 << 20), QUIPAMD_ERR_SHAPE, "hessian_accum_fast: d too large");
    hipStream_t s = (hipStream_t)stream;
    const int64_t tpad = (tokens + FKT - 1) / FKT * FKT;
    uint16_t *Xt = (uint16_t *)workspace;
    htranspose_kernel<<<dim3((unsigned)(tpad / 32), (unsigned)((d + 31) / 32)), 256, 0, s>>>((const uint16_t *)x, ldx, tokens, d, Xt, tpad);
    QA_LAUNCH_CHECK("hessian_accum_fast (transpose)");
    return x_dtype == QUIPAMD_F16 ? pick_fast<F16>(Xt, tpad, d, Hacc, s) : pick_fast<BF16>(Xt, tpad, d, Hacc, s);
}
