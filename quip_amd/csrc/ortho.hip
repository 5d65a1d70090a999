// ortho.hip -- K3: structured random-orthogonal apply (two-factor butterfly / Kronecker, `--pre_proj`).
//
// Takes over mul_ortho_butterfly (method.py:46-67) and with it the dense U @ W @ V^T / V @ H @ V^T products
// of QuantMethod.preproc / postproc (method.py:175-176,202-203): the operator is applied in its factored
// form, never materialised (2 n c (p+q) flops instead of 2 n^2 c).
//
// For one length-n vector v (n = p*q), with z viewed as [p][q] (index pos = a*q + b):
//   forward   z = v[perm_in];  z[a][b] <- sum_a' B0[b][a][a'] z[a'][b];  z[a][b] <- sum_b' B1[a][b][b'] z[a][b'];
//             out = z[perm_out]
//   transpose z[perm_out] = v;  z[a][b] <- sum_b' B1[a][b'][b] z[a][b'];  z[a][b] <- sum_a' B0[b][a'][a] z[a'][b];
//             out[perm_in] = z
// (index form verified against the reference in tests/golden/butterfly.npz through the oracle).
//
// Design: the operator is two "mix ONE index" stages, each a batch of tiny GEMMs
//       out[r][i][c] = sum_k M_c[i][k] * in[r][k][c]          (i, k over the mixed index, c the other index)
// run as TWO launches of one stage kernel with an fp32 intermediate z[rows][n] in a caller workspace.  A workgroup
// (4 waves) owns 16 rows x QB values of c:
//   load   the [QB][16][Pm] input tile into LDS (first stage: permutation as a GATHER, optional column scale);
//   mix    on the fp32 matrix pipe, v_mfma_f32_16x16x4_f32 (exact fmaf chains, 157 TF): the 16 ROWS are the MFMA M
//          dimension (A operand = data, one ds_read_b128 per 4 MFMAs), the factor is the B operand, pre-arranged on
//          the host in B-fragment order so a lane's 16-byte load feeds 4 MFMAs (coalesced 1 KiB per wave, L2 resident);
//   store  the [QB][16][Pm] output tile (second stage: permutation as a SCATTER, conversion to the output dtype).
// Tiling over c is what lets a 16-row activation batch (packed-layer forward) use hundreds of workgroups; QB shrinks
// until the grid covers the chip.  K index order inside a group of 16 is permuted (lane group g, step s -> k = 4g + s)
// identically for both operands, which a sum does not see.
// Algorithmic bytes: rows*n*(in + out element size) (+ 8*rows*n for the intermediate); FLOPs: 2*rows*n*(p+q).
#include "common.h"

namespace {

struct StageArgs {
    const void *in;
    void *out;
    int64_t ldi, ldo, rows;
    const int32_t *gidx;      // in position pos reads in[r][gidx[pos]] (null: pos)
    const int32_t *sidx;      // out position pos is written to out[r][sidx[pos]] (null: pos)
    const float *colscale;    // multiplies in[r][k] on load, indexed by the SOURCE column k (null: none)
    const float4 *frag;       // [C][NT][KS][64] float4, C = Po (blocked) or 1 (Kronecker)
    int blocked;
    int Pm, Po, q;            // mixed / other index range, q of the [p][q] view
    int mixa;                 // 1: mix a (pos = i*q + c), 0: mix b (pos = c*q + i)
    int QB, NT, RS;           // c values per workgroup (power of 2 <= 16), ceil(Pm/16), LDS row stride (floats)
};

template <class TI, class TO>
__global__ __launch_bounds__(256) void ortho_stage_kernel(StageArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int QB = A.QB, Pm = A.Pm, RS = A.RS, NT = A.NT, q = A.q;
    float *T = smem;                        // [QB][16][RS] input tile
    float *O = smem + QB * 16 * RS;         // [QB][16][RS] output tile
    const int c0 = blockIdx.x * QB;
    const int64_t r0 = (int64_t)blockIdx.y * 16;

    // ---- load ------------------------------------------------------------------------------------------------------
    // zero the K padding [Pm, 16*NT) of every (c, row)
    const int kpad = 16 * NT - Pm;
    if (kpad > 0)
        for (int idx = threadIdx.x; idx < QB * 16 * kpad; idx += 256) {
            const int line = idx / kpad, k = Pm + idx - line * kpad;
            T[line * RS + k] = 0.f;
        }
    // Both loops are written for memory-level parallelism: UNR independent (index -> data) chains per thread are
    // issued before any is consumed (the scalar one-element-per-iteration form was latency-bound: 20x off).
    constexpr int UNR = 8;
    if (A.mixa) {
        // runs (r, i) of QB consecutive c: pos = i*q + c0 + cl
        const int rpw = 64 / QB;                               // runs per wave pass
        const int cl = lane & (QB - 1), rl = lane / QB;
        const int c = c0 + cl;
        const int nrun = 16 * Pm;
        for (int run0 = wave * rpw; run0 < nrun; run0 += 4 * rpw * UNR) {
            int src[UNR];
            float v[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int run = run0 + u * 4 * rpw + rl;
                const int r = run & 15, i = run >> 4;
                ok[u] = (run < nrun) && (r0 + r < A.rows) && (c < A.Po);
                const int pos = i * q + c;
                src[u] = ok[u] ? (A.gidx ? A.gidx[pos] : pos) : 0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int run = run0 + u * 4 * rpw + rl;
                const int r = run & 15;
                v[u] = ok[u] ? DT<TI>::load(A.in, (r0 + r) * A.ldi + src[u]) : 0.f;
                if (ok[u] && A.colscale) v[u] *= A.colscale[src[u]];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int run = run0 + u * 4 * rpw + rl;
                const int r = run & 15, i = run >> 4;
                if (run < nrun) T[(cl * 16 + r) * RS + i] = v[u];
            }
        }
    } else {
        // lines (cl, r) of Pm consecutive i: pos = (c0 + cl)*q + i; elements e = (line, ii) with i = lane + 64*ii
        const int ipl = (Pm + 63) / 64;                        // lane passes per line
        const int nel = 16 * QB * ipl;                         // (line, ii) pairs, split over the 4 waves
        for (int e0 = wave; e0 < nel; e0 += 4 * UNR) {
            int src[UNR];
            float v[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int e = e0 + 4 * u;
                const int line = e / ipl, i = lane + 64 * (e - line * ipl);
                const int r = line & 15, c = c0 + (line >> 4);
                ok[u] = (e < nel) && (i < Pm) && (r0 + r < A.rows) && (c < A.Po);
                const int pos = c * q + i;
                src[u] = ok[u] ? (A.gidx ? A.gidx[pos] : pos) : 0;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int e = e0 + 4 * u;
                const int line = e / ipl;
                const int r = line & 15;
                v[u] = ok[u] ? DT<TI>::load(A.in, (r0 + r) * A.ldi + src[u]) : 0.f;
                if (ok[u] && A.colscale) v[u] *= A.colscale[src[u]];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int e = e0 + 4 * u;
                const int line = e / ipl, i = lane + 64 * (e - line * ipl);
                if (e < nel && i < Pm) T[line * RS + i] = v[u];
            }
        }
    }
    __syncthreads();

    // ---- mix: items (cl, nt) round-robin over the 4 waves; flattened (item, S) loop with a one-step prefetch ---------
    {
        const int row = lane & 15, g = lane >> 4;
        const int nitems = QB * NT;
        const int64_t cstride = A.blocked ? (int64_t)NT * NT * 64 : 0;
        int item = wave;
        if (item < nitems) {
            int cl = item / NT, nt = item - cl * NT, S = 0;
            const float4 *fp = A.frag + (c0 + cl < A.Po ? (c0 + cl) : 0) * cstride + ((int64_t)nt * NT) * 64 + lane;
            const float *tp = T + (cl * 16 + row) * RS + 4 * g;
            float4 b_nxt = fp[0];
            float4 a_nxt = *reinterpret_cast<const float4 *>(tp);
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            while (true) {
                const float4 a = a_nxt, b = b_nxt;
                // advance the (item, S) cursor and prefetch its operands
                int cl2 = cl, nt2 = nt, S2 = S + 1, item2 = item;
                if (S2 == NT) {
                    S2 = 0;
                    item2 = item + 4;
                    if (item2 < nitems) { cl2 = item2 / NT; nt2 = item2 - cl2 * NT; }
                }
                const bool more = item2 < nitems;
                if (more) {
                    const float4 *fp2 = A.frag + (c0 + cl2 < A.Po ? (c0 + cl2) : 0) * cstride + ((int64_t)nt2 * NT + S2) * 64 + lane;
                    b_nxt = fp2[0];
                    a_nxt = *reinterpret_cast<const float4 *>(T + (cl2 * 16 + row) * RS + 16 * S2 + 4 * g);
                }
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
                if (S2 == 0) {
                    // item finished: D[row = 4g + reg][col = lane & 15] -> O[cl][4g + reg][16 nt + (lane & 15)]
                    float *op = O + (cl * 16 + 4 * g) * RS + 16 * nt + row;
                    op[0] = acc[0]; op[RS] = acc[1]; op[2 * RS] = acc[2]; op[3 * RS] = acc[3];
                    acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
                if (!more) break;
                cl = cl2; nt = nt2; S = S2; item = item2;
            }
        }
    }
    __syncthreads();

    // ---- store (same UNR-way batching: LDS reads + scatter-index loads first, then the stores) ------------------------
    if (A.mixa) {
        const int rpw = 64 / QB;
        const int cl = lane & (QB - 1), rl = lane / QB;
        const int c = c0 + cl;
        const int nrun = 16 * Pm;
        for (int run0 = wave * rpw; run0 < nrun; run0 += 4 * rpw * UNR) {
            int dst[UNR];
            float v[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int run = run0 + u * 4 * rpw + rl;
                const int r = run & 15, i = run >> 4;
                ok[u] = (run < nrun) && (r0 + r < A.rows) && (c < A.Po);
                const int pos = i * q + c;
                dst[u] = ok[u] ? (A.sidx ? A.sidx[pos] : pos) : 0;
                v[u] = ok[u] ? O[(cl * 16 + r) * RS + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int run = run0 + u * 4 * rpw + rl;
                const int r = run & 15;
                if (ok[u]) DT<TO>::store(A.out, (r0 + r) * A.ldo + dst[u], v[u]);
            }
        }
    } else {
        const int ipl = (Pm + 63) / 64;
        const int nel = 16 * QB * ipl;
        for (int e0 = wave; e0 < nel; e0 += 4 * UNR) {
            int dst[UNR];
            float v[UNR];
            bool ok[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int e = e0 + 4 * u;
                const int line = e / ipl, i = lane + 64 * (e - line * ipl);
                const int r = line & 15, c = c0 + (line >> 4);
                ok[u] = (e < nel) && (i < Pm) && (r0 + r < A.rows) && (c < A.Po);
                const int pos = c * q + i;
                dst[u] = ok[u] ? (A.sidx ? A.sidx[pos] : pos) : 0;
                v[u] = ok[u] ? O[line * RS + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int e = e0 + 4 * u;
                const int r = (e / ipl) & 15;
                if (ok[u]) DT<TO>::store(A.out, (r0 + r) * A.ldo + dst[u], v[u]);
            }
        }
    }
}

template <class TI, class TO>
int launch_stage(StageArgs A, hipStream_t s)
{
    A.NT = (A.Pm + 15) / 16;
    A.RS = 16 * A.NT + 4;                                   // 16-byte aligned rows, consecutive rows 4 slots apart
    const int64_t per_c = 2ll * 16 * A.RS * 4;              // bytes of LDS per value of c (input + output tile)
    QA_REQUIRE(per_c <= 160 * 1024, QUIPAMD_ERR_SHAPE, "ortho_apply_rows: factor size %d needs %lld B of LDS", A.Pm, (long long)per_c);
    int qb = 16;
    while (qb > 1 && qb * per_c > 72 * 1024) qb >>= 1;      // two workgroups per CU when possible
    const int64_t rgroups = (A.rows + 15) / 16;
    while (qb > 1 && rgroups * ((A.Po + qb - 1) / qb) < 512) qb >>= 1;   // cover the chip
    while (qb > 1 && qb / 2 >= A.Po) qb >>= 1;
    A.QB = qb;
    const size_t lds = (size_t)(qb * per_c);
    auto kern = ortho_stage_kernel<TI, TO>;
    if (lds > 64 * 1024)
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "ortho: cannot raise dynamic LDS to %zu", lds);
    QA_REQUIRE(rgroups <= 65535, QUIPAMD_ERR_SHAPE, "ortho_apply_rows: too many rows (%lld)", (long long)A.rows);
    kern<<<dim3((unsigned)((A.Po + qb - 1) / qb), (unsigned)rgroups), 256, lds, s>>>(A);
    QA_LAUNCH_CHECK("quipamd_ortho_apply_rows");
    return QUIPAMD_OK;
}

template <class TI>
int launch_first(const StageArgs &A, hipStream_t s) { return launch_stage<TI, F32>(A, s); }

}   // namespace

extern "C" int quipamd_ortho_apply_rows(const float *frag_first, const float *frag_second, int blocked, const int32_t *gather_idx,
                                        const int32_t *scatter_idx, int p, int q, int b_first, const float *colscale,
                                        const void *x, int x_dtype, int64_t ldx, void *out, int out_dtype, int64_t ldo,
                                        int64_t rows, float *workspace, void *stream)
{
    if (rows == 0) return QUIPAMD_OK;
    QA_REQUIRE(frag_first && frag_second && x && out && workspace, QUIPAMD_ERR_ARG, "ortho_apply_rows: null pointer");
    QA_REQUIRE(p >= 1 && q >= 1, QUIPAMD_ERR_SHAPE, "ortho_apply_rows: bad factors p=%d q=%d", p, q);
    const int64_t n = (int64_t)p * q;
    QA_REQUIRE(ldx >= n && ldo >= n, QUIPAMD_ERR_SHAPE, "ortho_apply_rows: leading dimension < n");
    QA_REQUIRE(n < ((int64_t)1 << 31), QUIPAMD_ERR_SHAPE, "ortho_apply_rows: n too large");
    if (rows == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    // stage 1: x -> workspace (fp32, ld = n), gather + colscale on load
    StageArgs A1;
    A1.in = x; A1.out = workspace; A1.ldi = ldx; A1.ldo = n; A1.rows = rows;
    A1.gidx = gather_idx; A1.sidx = nullptr; A1.colscale = colscale;
    A1.frag = (const float4 *)frag_first; A1.blocked = blocked; A1.q = q;
    A1.mixa = b_first ? 0 : 1;
    A1.Pm = b_first ? q : p; A1.Po = b_first ? p : q;
    A1.QB = A1.NT = A1.RS = 0;
    int rc;
    switch (x_dtype) {
    case QUIPAMD_F32: rc = launch_stage<F32, F32>(A1, s); break;
    case QUIPAMD_F16: rc = launch_stage<F16, F32>(A1, s); break;
    case QUIPAMD_BF16: rc = launch_stage<BF16, F32>(A1, s); break;
    default: return qa_fail(QUIPAMD_ERR_ARG, "ortho_apply_rows: bad x dtype %d", x_dtype);
    }
    if (rc) return rc;
    // stage 2: workspace -> out, scatter on store
    StageArgs A2 = A1;
    A2.in = workspace; A2.out = out; A2.ldi = n; A2.ldo = ldo;
    A2.gidx = nullptr; A2.sidx = scatter_idx; A2.colscale = nullptr;
    A2.frag = (const float4 *)frag_second;
    A2.mixa = b_first ? 1 : 0;
    A2.Pm = b_first ? p : q; A2.Po = b_first ? q : p;
    switch (out_dtype) {
    case QUIPAMD_F32: return launch_stage<F32, F32>(A2, s);
    case QUIPAMD_F16: return launch_stage<F32, F16>(A2, s);
    case QUIPAMD_BF16: return launch_stage<F32, BF16>(A2, s);
    default: return qa_fail(QUIPAMD_ERR_ARG, "ortho_apply_rows: bad out dtype %d", out_dtype);
    }
}
