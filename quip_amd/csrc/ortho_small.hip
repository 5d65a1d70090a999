// ortho_small.hip -- K3, activation side: the Kronecker (noblock) structured orthogonal operator applied to a FEW rows
// (the x-side V (x (/) s) and the y-side U^T y of the packed layer's forward; batch 1..64) in ONE launch.
//
// Same operator as ortho.hip (method.py:46-67 with the one-factor-per-stage generator method.py:38-39):
//     z = scatter(x * colscale);  two mix stages;  out = gather(z) (+ bias)
// but here a whole row (n = p*q floats) and both factor matrices (p*p + q*q floats) fit in one workgroup's LDS, so the
// two stages need no intermediate in memory and no second launch: one workgroup (16 waves) per row.
//   stage "a":  z1[a][b] = sum_a' M0[a][a'] z[a'][b]      D[a][b]: A = M0 tile (16 x 4, ds_read_b128), B = z (4 x 16 b's)
//   stage "b":  z2[a][b] = sum_b' M1[b][b'] z1[a][b']     D[b][a]: A = M1 tile, B = z1^T (ds_read_b128 along b')
// on v_mfma_f32_16x16x4_f32 (exact fp32).  With one factor matrix per stage the MFMA N dimension is the OTHER index
// (16 b's or 16 a's of the same row), so a single row already fills the tile.  M0 / M1 are passed already transposed
// for Q^T, and b_first selects the stage order.  Requires p, q multiples of 16 and (p*(p+4) + q*(q+4) + 2*p*(q+4) + 16)*4 B <= 160 KiB.
// A decode step is launch-latency bound: this turns 2 launches + allocations per apply into 1.
#include "common.h"

#include "small_pass.h"

namespace {

// image layout: z[a][b] at a*QS + b, QS = q + 4 (16-byte aligned rows, spreads the ds_read_b128 of stage b over banks)
template <class TI, class TO>
__global__ __launch_bounds__(1024) void ortho_small_kernel(SmallBatch Bt)
{
    const SmallArgs &A = Bt.op[blockIdx.y];
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int p = A.p, q = A.q, n = p * q, QS = q + 4;
    // factor rows are PADDED by 4 floats: with the natural strides (256 / 512 B) the 16 rows of every ds_read_b128
    // A-fragment read sat on one bank group -- a 16-way conflict that made this kernel 19-23 us at n = 8192
    const int PS0 = p + 4, PS1 = q + 4;
    float *F0 = smem;                    // [p][PS0]
    float *F1 = F0 + p * PS0;            // [q][PS1]
    float *Z0 = F1 + q * PS1;            // [p][QS]
    float *Z1 = Z0 + p * QS;             // [p][QS]
    float *red = Z1 + p * QS;            // [16] block reduction scratch
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- load: factors (coalesced float4) and the row (coalesced, scattered into the image) ------------------------
    for (int i = tid; i < p * p / 4; i += 1024) {
        const int rr = i / (p / 4), c4 = i - rr * (p / 4);
        *reinterpret_cast<float4 *>(F0 + rr * PS0 + 4 * c4) = reinterpret_cast<const float4 *>(A.M0)[i];
    }
    for (int i = tid; i < q * q / 4; i += 1024) {
        const int rr = i / (q / 4), c4 = i - rr * (q / 4);
        *reinterpret_cast<float4 *>(F1 + rr * PS1 + 4 * c4) = reinterpret_cast<const float4 *>(A.M1)[i];
    }
    // The row is handled 4 consecutive elements at a time (16-byte loads of x, column scale, LayerNorm parameters and
    // the permutation): one CU issues every memory instruction of this kernel, so instruction COUNT is the cost
    // (the element-wise form spent 8 us on ~3000 narrow loads/stores).  Element group v = tid + 1024*u covers
    // k = 4v .. 4v+3; n <= 16384 -> at most 4 groups per thread.  q is a power of two: pos -> (a, b) by shift / mask.
    constexpr int MAXV = 4;
    const int qsh = __builtin_ctz(q), qmask = q - 1, n4 = n >> 2;
    // a workgroup walks rows blockIdx.x, + gridDim.x, ... : the factors above are loaded once per workgroup, not per row
    for (int64_t row = blockIdx.x; row < Bt.rows; row += gridDim.x) {
    float4 xv[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int v4 = tid + 1024 * u;
        xv[u] = v4 < n4 ? load4<TI>(A.x, row * A.ldx + 4 * v4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (A.ln_gamma) {
        const bool rms = A.ln_beta == nullptr;                 // RMSNorm: no mean, no shift
        float mean = 0.f;
        if (!rms) {
            float s1 = 0.f;
#pragma unroll
            for (int u = 0; u < MAXV; ++u) s1 += (xv[u].x + xv[u].y) + (xv[u].z + xv[u].w);
            mean = block_sum(s1, red) / (float)n;
        }
        float s2 = 0.f;
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            if (tid + 1024 * u < n4) {
                const float d0 = xv[u].x - mean, d1 = xv[u].y - mean, d2 = xv[u].z - mean, d3 = xv[u].w - mean;
                s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        const float rstd = rsqrtf(block_sum(s2, red) / (float)n + A.ln_eps);
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            const int v4 = tid + 1024 * u;
            if (v4 < n4) {
                const float4 gm = load4_any(A.ln_gamma, A.ln_dtype, 4 * v4);
                const float4 bt = rms ? make_float4(0.f, 0.f, 0.f, 0.f) : load4_any(A.ln_beta, A.ln_dtype, 4 * v4);
                xv[u] = make_float4((xv[u].x - mean) * rstd * gm.x + bt.x, (xv[u].y - mean) * rstd * gm.y + bt.y,
                                    (xv[u].z - mean) * rstd * gm.z + bt.z, (xv[u].w - mean) * rstd * gm.w + bt.w);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int v4 = tid + 1024 * u;
        if (v4 < n4) {
            float4 v = xv[u];
            if (A.colscale) {
                const float4 c = *reinterpret_cast<const float4 *>(A.colscale + 4 * v4);
                v = make_float4(v.x * c.x, v.y * c.y, v.z * c.z, v.w * c.w);
            }
            int4 pos = make_int4(4 * v4, 4 * v4 + 1, 4 * v4 + 2, 4 * v4 + 3);
            if (A.load_idx) pos = *reinterpret_cast<const int4 *>(A.load_idx + 4 * v4);
            Z0[(pos.x >> qsh) * QS + (pos.x & qmask)] = v.x;
            Z0[(pos.y >> qsh) * QS + (pos.y & qmask)] = v.y;
            Z0[(pos.z >> qsh) * QS + (pos.z & qmask)] = v.z;
            Z0[(pos.w >> qsh) * QS + (pos.w & qmask)] = v.w;
        }
    }
    __syncthreads();

    const int j = lane & 15, g = lane >> 4;
    float *src = Z0, *dst = Z1;
    // NOTE: one row is 2 n (p + q) flops of fp32 MFMA on ONE CU (614 GFLOP/s): 5.1 us at n = 8192 -- the measured floor of
    // this kernel (ablation: skeleton 4.2 us, stages 8.6 us).  Next step: split-bf16 operands (hi + lo) on the bf16 pipe.
    for (int st = 0; st < 2; ++st) {
        const bool mix_a = (st == 0) != (A.b_first != 0);
        if (mix_a) {
            // tiles (at, bt): D[a = 16at + 4g + reg][b = 16bt + j]
            const int nat = p / 16, nbt = q / 16;
            for (int tile = wave; tile < nat * nbt; tile += 16) {
                const int at = tile / nbt, bt = tile - at * nbt;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const float *fa = F0 + (16 * at + j) * PS0 + 4 * g;              // A[i = j][k = 16S + 4g + s]
                const float *zb = src + (4 * g) * QS + 16 * bt + j;              // B[k][j] = z[a' = 16S + 4g + s][b]
                for (int S = 0; S < p / 16; ++S) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(fa + 16 * S);
                    const float *zs = zb + 16 * S * QS;
                    const float b0 = zs[0], b1 = zs[QS], b2 = zs[2 * QS], b3 = zs[3 * QS];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b0, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b1, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b2, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b3, acc, 0, 0, 0);
                }
                float *o = dst + (16 * at + 4 * g) * QS + 16 * bt + j;
                o[0] = acc[0]; o[QS] = acc[1]; o[2 * QS] = acc[2]; o[3 * QS] = acc[3];
            }
        } else {
            // tiles (bt, at): D[b = 16bt + 4g + reg][a = 16at + j]
            const int nat = p / 16, nbt = q / 16;
            for (int tile = wave; tile < nat * nbt; tile += 16) {
                const int bt = tile / nat, at = tile - bt * nat;
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const float *fa = F1 + (16 * bt + j) * PS1 + 4 * g;              // A[i = b][k = b']
                const float *zb = src + (16 * at + j) * QS + 4 * g;              // B[k = b'][j = a] = z1[a][b'], 4 consecutive b'
                for (int S = 0; S < q / 16; ++S) {
                    const float4 a4 = *reinterpret_cast<const float4 *>(fa + 16 * S);
                    const float4 b4 = *reinterpret_cast<const float4 *>(zb + 16 * S);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
                }
                // rows of D are 4 consecutive b of one a: one 16-byte store
                *reinterpret_cast<float4 *>(dst + (16 * at + j) * QS + 16 * bt + 4 * g) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
        __syncthreads();
        float *t = src; src = dst; dst = t;
    }

    // ---- store: gather from the image, bias, convert ------------------------------------------------------------------
    // epilogue, 4 outputs per step: gather from the image, bias, residual, ReLU, one 8/16-byte store
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int v4 = tid + 1024 * u;
        if (v4 < n4) {
            int4 pos = make_int4(4 * v4, 4 * v4 + 1, 4 * v4 + 2, 4 * v4 + 3);
            if (A.store_idx) pos = *reinterpret_cast<const int4 *>(A.store_idx + 4 * v4);
            float4 v = make_float4(src[(pos.x >> qsh) * QS + (pos.x & qmask)], src[(pos.y >> qsh) * QS + (pos.y & qmask)],
                                   src[(pos.z >> qsh) * QS + (pos.z & qmask)], src[(pos.w >> qsh) * QS + (pos.w & qmask)]);
            if (A.bias) {
                const float4 c = *reinterpret_cast<const float4 *>(A.bias + 4 * v4);
                v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
            }
            if (A.residual) {
                const float4 c = load4_any(A.residual, A.res_dtype, row * A.ldo + 4 * v4);
                v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
            }
            if (A.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            store4<TO>(A.out, row * A.ldo + 4 * v4, v);
        }
    }
    __syncthreads();                                             // the image is rewritten by the next row
    }
}

// NPASS = 2 ("chain"): op[0] is applied to the row first (x from memory, bias / residual / relu epilogue, result stored
// to op[0].out by the blockIdx.y == 0 workgroup when that pointer is set), then op[1 + blockIdx.y] is applied to the
// result straight from registers (LayerNorm / colscale on the way in): U^T y + residual -> LayerNorm -> V (x (/) s) of
// two consecutive packed layers in ONE launch instead of two.  The hand-over value is rounded
// to op[0].out_dtype, so the chain computes exactly what the two separate launches compute.
// CP, CQ: compile-time factor sizes for the shapes a decode step uses (0 = take them from the op).  A phase-stamp probe
// showed every phase of this kernel at 1500-3000 cycles for a few dozen MFMAs' worth of work: instruction count at 16
// waves per CU is the cost, and with p, q known the address arithmetic folds and the k loops unroll.  The op descriptor
// is copied by value once per pass (one batch of scalar loads instead of a kernarg round trip per field on first use).
template <class TI, class TO, int NPASS, int CP, int CQ>
__global__ __launch_bounds__(1024) void ortho_small_split_kernel(SmallBatch Bt)
{
    constexpr int MAXV = 4;
    float4 xv[MAXV];
    extern __shared__ __attribute__((aligned(16))) char smemc[];
    if constexpr (NPASS == 1) {
        // many rows (weight side, prefill): a workgroup walks rows blockIdx.x, + gridDim.x, ... with the factor images
        // loaded once -- per row the factors (24 - 80 KiB) were most of the traffic
        const SmallArgs A = Bt.op[blockIdx.y];
        bool first = true;
        for (int64_t row = blockIdx.x; row < Bt.rows; row += gridDim.x) {
            if (!first) __syncthreads();                             // the previous row's final image is still being read
            small_split_pass<TI, CP, CQ>(A, row, true, xv, smemc,
                                         [&](int, int v4, const float4 &v) { store4<TO>(A.out, row * A.ldo + 4 * v4, v); }, first);
            first = false;
        }
    } else {
        const int64_t row = blockIdx.x;
        for (int pass = 0; pass < NPASS; ++pass) {
            const SmallArgs A = pass == 0 ? Bt.op[0] : Bt.op[1 + blockIdx.y];
            small_split_pass<TI, CP, CQ>(A, row, pass == 0, xv, smemc, [&](int u, int v4, const float4 &v) {
                if (pass == NPASS - 1) store4<TO>(A.out, row * A.ldo + 4 * v4, v);
                else {
                    if (A.out && blockIdx.y == 0) store4_any(A.out, A.out_dtype, row * A.ldo + 4 * v4, v);
                    xv[u] = round4_any(A.out_dtype, v);
                }
            });
            if (pass < NPASS - 1) __syncthreads();                   // ZF is read above, rewritten by the next pass
        }
    }
}

size_t small_lds(int p, int q) { return ((size_t)p * (p + 4) + (size_t)q * (q + 4) + 2 * (size_t)p * (q + 4) + 16) * 4; }

template <class TI, class TO, int NPASS, int CP, int CQ>
int launch_split_pq(const SmallBatch &B, int ny, int64_t rows, hipStream_t s, const char *who)
{
    const size_t lds = small_split_lds(B.op[0].p, B.op[0].q);
    auto kern = ortho_small_split_kernel<TI, TO, NPASS, CP, CQ>;
    if (lds > 64 * 1024)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "%s: cannot raise dynamic LDS to %zu", who, lds);
    // NPASS == 1: workgroups walk the rows (factors loaded once each); a chain keeps one workgroup per row
    const unsigned gx = (unsigned)(NPASS == 1 && rows > 1024 ? 1024 : rows);
    kern<<<dim3(gx, (unsigned)ny), 1024, lds, s>>>(B);
    QA_LAUNCH_CHECK(who);
    return QUIPAMD_OK;
}

template <class A, class B> struct SameT { static constexpr bool v = false; };
template <class A> struct SameT<A, A> { static constexpr bool v = true; };

// the dtype pairs of a decode step get the shape-specialised instantiations: V side f16 -> bf16, U side f32 -> f16, chain f32 -> bf16
template <class TI, class TO, int NPASS>
int launch_split(const SmallBatch &B, int ny, int64_t rows, hipStream_t s, const char *who)
{
    constexpr bool decode_pair = NPASS == 1 ? ((SameT<TI, F16>::v && SameT<TO, BF16>::v) || (SameT<TI, F32>::v && SameT<TO, F16>::v))
                                            : (SameT<TI, F32>::v && SameT<TO, BF16>::v);
    if constexpr (decode_pair) {
        const int p = B.op[0].p, q = B.op[0].q;
        if (p == 64 && q == 32) return launch_split_pq<TI, TO, NPASS, 64, 32>(B, ny, rows, s, who);      // n = 2048
        if (p == 64 && q == 64) return launch_split_pq<TI, TO, NPASS, 64, 64>(B, ny, rows, s, who);      // n = 4096
        if (p == 128 && q == 64) return launch_split_pq<TI, TO, NPASS, 128, 64>(B, ny, rows, s, who);    // n = 8192
    }
    return launch_split_pq<TI, TO, NPASS, 0, 0>(B, ny, rows, s, who);
}

template <class TI, class TO>
int launch_small(const SmallBatch &B, int nops, int64_t rows, hipStream_t s)
{
    if (B.op[0].M0_hi) return launch_split<TI, TO, 1>(B, nops, rows, s, "quipamd_ortho_apply_small");
    const size_t lds = small_lds(B.op[0].p, B.op[0].q);
    auto kern = ortho_small_kernel<TI, TO>;
    if (lds > 64 * 1024)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "ortho_apply_small: cannot raise dynamic LDS to %zu", lds);
    kern<<<dim3((unsigned)(rows > 1024 ? 1024 : rows), (unsigned)nops), 1024, lds, s>>>(B);
    QA_LAUNCH_CHECK("quipamd_ortho_apply_small");
    return QUIPAMD_OK;
}

}   // namespace

extern "C" int quipamd_ortho_apply_small_ops(const quipamd_small_op *ops, int nops, int64_t rows, void *stream)
{
    QA_REQUIRE(ops && nops >= 1 && nops <= QUIPAMD_SMALL_MAX_OPS, QUIPAMD_ERR_ARG, "ortho_apply_small: 1..%d ops", QUIPAMD_SMALL_MAX_OPS);
    const int p = ops[0].p, q = ops[0].q, x_dtype = ops[0].x_dtype, out_dtype = ops[0].out_dtype;
    QA_REQUIRE(p >= 16 && q >= 16 && p % 16 == 0 && q % 16 == 0, QUIPAMD_ERR_SHAPE,
               "ortho_apply_small: p and q must be multiples of 16 (p=%d q=%d); use quipamd_ortho_apply_rows", p, q);
    QA_REQUIRE(small_lds(p, q) <= 160 * 1024, QUIPAMD_ERR_SHAPE, "ortho_apply_small: factors + row need %zu B of LDS (> 160 KiB)", small_lds(p, q));
    QA_REQUIRE((int64_t)p * q <= 16 * 1024, QUIPAMD_ERR_SHAPE, "ortho_apply_small: n = %d > 16384", p * q);
    QA_REQUIRE((q & (q - 1)) == 0, QUIPAMD_ERR_SHAPE, "ortho_apply_small: q = %d must be a power of two; use quipamd_ortho_apply_rows", q);
    QA_REQUIRE(rows >= 0 && rows < ((int64_t)1 << 31), QUIPAMD_ERR_SHAPE, "ortho_apply_small: bad row count");
    if (rows == 0) return QUIPAMD_OK;                           // an empty batch has null data pointers
    const bool split = ops[0].M0_hi != nullptr;
    if (split) {
        QA_REQUIRE(p % 32 == 0 && q % 32 == 0, QUIPAMD_ERR_SHAPE, "ortho_apply_small: split-bf16 factors need p, q multiples of 32");
        QA_REQUIRE(small_split_lds(p, q) <= 160 * 1024 && 2 * q >= p, QUIPAMD_ERR_SHAPE,
                   "ortho_apply_small: split-bf16 images need %zu B of LDS and q >= p/2", small_split_lds(p, q));
    }
    SmallBatch B;
    for (int i = 0; i < nops; ++i) {
        const quipamd_small_op &o = ops[i];
        QA_REQUIRE((o.M0_hi != nullptr) == split && (!split || (o.M0_lo && o.M1_hi && o.M1_lo)), QUIPAMD_ERR_ARG,
                   "ortho_apply_small: ops of one launch must all carry (or all omit) the four split-bf16 factor arrays");
        QA_REQUIRE(o.M0 && o.M1 && o.x && o.out, QUIPAMD_ERR_ARG, "ortho_apply_small: null pointer in op %d", i);
        QA_REQUIRE(o.p == p && o.q == q && o.x_dtype == x_dtype && o.out_dtype == out_dtype, QUIPAMD_ERR_ARG,
                   "ortho_apply_small: ops of one launch must share p, q and dtypes (op %d differs)", i);
        QA_REQUIRE(o.ldx >= (int64_t)p * q && o.ldo >= (int64_t)p * q && o.ldx % 4 == 0 && o.ldo % 4 == 0, QUIPAMD_ERR_SHAPE,
                   "ortho_apply_small: leading dimensions must be >= n and multiples of 4");
        B.op[i] = o;
    }
    for (int i = nops; i < QUIPAMD_SMALL_MAX_OPS; ++i) B.op[i] = ops[0];
    B.rows = rows;
    if (rows == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
#define QA_SMALL_CASE(XI, TI, XO, TO) if (x_dtype == XI && out_dtype == XO) return launch_small<TI, TO>(B, nops, rows, s)
    QA_SMALL_CASE(QUIPAMD_F32, F32, QUIPAMD_F32, F32);
    QA_SMALL_CASE(QUIPAMD_F32, F32, QUIPAMD_F16, F16);
    QA_SMALL_CASE(QUIPAMD_F32, F32, QUIPAMD_BF16, BF16);
    QA_SMALL_CASE(QUIPAMD_F16, F16, QUIPAMD_BF16, BF16);
    QA_SMALL_CASE(QUIPAMD_F16, F16, QUIPAMD_F16, F16);
    QA_SMALL_CASE(QUIPAMD_F16, F16, QUIPAMD_F32, F32);
    QA_SMALL_CASE(QUIPAMD_BF16, BF16, QUIPAMD_BF16, BF16);
    QA_SMALL_CASE(QUIPAMD_BF16, BF16, QUIPAMD_F16, F16);
    QA_SMALL_CASE(QUIPAMD_BF16, BF16, QUIPAMD_F32, F32);
#undef QA_SMALL_CASE
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "ortho_apply_small: dtype pair %d -> %d", x_dtype, out_dtype);
}

extern "C" int quipamd_ortho_apply_small_chain(const quipamd_small_op *first, const quipamd_small_op *second, int nsecond,
                                               int64_t rows, void *stream)
{
    QA_REQUIRE(first && second && nsecond >= 1 && nsecond <= QUIPAMD_SMALL_MAX_OPS - 1, QUIPAMD_ERR_ARG,
               "ortho_apply_small_chain: 1..%d second ops", QUIPAMD_SMALL_MAX_OPS - 1);
    const int p = first->p, q = first->q;
    QA_REQUIRE(p >= 32 && q >= 32 && p % 32 == 0 && q % 32 == 0 && (q & (q - 1)) == 0 && 2 * q >= p && (int64_t)p * q <= 16 * 1024 &&
                   small_split_lds(p, q) <= 160 * 1024,
               QUIPAMD_ERR_SHAPE, "ortho_apply_small_chain: p=%d q=%d not supported by the split-bf16 single-launch kernel", p, q);
    QA_REQUIRE(first->M0_hi && first->M0_lo && first->M1_hi && first->M1_lo && first->x, QUIPAMD_ERR_ARG,
               "ortho_apply_small_chain: first op needs x and the four split-bf16 factor arrays");
    QA_REQUIRE(first->ldx >= (int64_t)p * q && first->ldx % 4 == 0 && first->ldo >= (int64_t)p * q && first->ldo % 4 == 0, QUIPAMD_ERR_SHAPE,
               "ortho_apply_small_chain: leading dimensions");
    QA_REQUIRE(rows <= 65535, QUIPAMD_ERR_SHAPE, "ortho_apply_small_chain: too many rows");
    SmallBatch B;
    B.op[0] = *first;
    const int out_dtype = second[0].out_dtype;
    for (int i = 0; i < nsecond; ++i) {
        const quipamd_small_op &o = second[i];
        QA_REQUIRE(o.p == p && o.q == q && o.out_dtype == out_dtype && o.out && o.M0_hi && o.M0_lo && o.M1_hi && o.M1_lo, QUIPAMD_ERR_ARG,
                   "ortho_apply_small_chain: second op %d must share p, q, the output dtype and carry split-bf16 factors", i);
        QA_REQUIRE(o.ldo >= (int64_t)p * q && o.ldo % 4 == 0, QUIPAMD_ERR_SHAPE, "ortho_apply_small_chain: leading dimensions");
        QA_REQUIRE(!o.residual, QUIPAMD_ERR_UNSUPPORTED, "ortho_apply_small_chain: residual on a second op");
        B.op[1 + i] = o;
    }
    for (int i = 1 + nsecond; i < QUIPAMD_SMALL_MAX_OPS; ++i) B.op[i] = second[0];
    B.rows = rows;
    if (rows == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
#define QA_CHAIN_CASE(XI, TI, XO, TO) \
    if (first->x_dtype == XI && out_dtype == XO) return launch_split<TI, TO, 2>(B, nsecond, rows, s, "quipamd_ortho_apply_small_chain");
    QA_CHAIN_CASE(QUIPAMD_F32, F32, QUIPAMD_BF16, BF16)
    QA_CHAIN_CASE(QUIPAMD_F32, F32, QUIPAMD_F16, F16)
    QA_CHAIN_CASE(QUIPAMD_F32, F32, QUIPAMD_F32, F32)
#undef QA_CHAIN_CASE
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "ortho_apply_small_chain: dtype pair %d -> %d (first x must be f32)", first->x_dtype, out_dtype);
}

extern "C" int quipamd_ortho_apply_small(const float *M0, const float *M1, const int32_t *load_idx, const int32_t *store_idx,
                                         int p, int q, int b_first, const float *colscale, const float *bias,
                                         const void *x, int x_dtype, int64_t ldx, void *out, int out_dtype, int64_t ldo,
                                         int64_t rows, void *stream)
{
    quipamd_small_op o;
    o.M0 = M0; o.M1 = M1; o.load_idx = load_idx; o.store_idx = store_idx; o.p = p; o.q = q; o.b_first = b_first;
    o.colscale = colscale; o.bias = bias; o.ln_gamma = nullptr; o.ln_beta = nullptr; o.ln_eps = 0.f; o.ln_dtype = QUIPAMD_F32;
    o.residual = nullptr; o.res_dtype = QUIPAMD_F32; o.relu = 0;
    o.M0_hi = o.M0_lo = o.M1_hi = o.M1_lo = nullptr;
    o.x = x; o.x_dtype = x_dtype; o.ldx = ldx; o.out = out; o.out_dtype = out_dtype; o.ldo = ldo;
    return quipamd_ortho_apply_small_ops(&o, 1, rows, stream);
}
