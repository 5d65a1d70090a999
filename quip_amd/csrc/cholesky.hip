// cholesky.hip -- K8: the LDL factor of the Hessian for LDLQ, LT = D^-1 U strictly upper, H = U^T U
//
// Replaces  `L = torch.linalg.cholesky(H); L = L @ diag(1/diag(L))`  and the L[j][c] column walks of round_ldl
// (reference vector_balance.py:171-173, 179-180): K4 wants row c of LT = column c of the unit-lower factor, i.e.
// LT = D^-1 U with U = C^T the UPPER Cholesky factor, rows contiguous.  rocSOLVER's potrf behind torch.linalg takes
// 6.0 / 12.8 / 28.5 ms at d = 2048 / 4096 / 8192 on MI355X -- linear in d, 3.5 us per column of pure latency -- and had
// become the largest part of Balance.fasterquant once the rounding itself (K4) took 0.4 - 4 ms.
//
// Blocked right-looking factorisation in fp32, in place in the LT buffer, upper triangle, NB = 64 rows per step:
//   diag   one wavefront factors the 64 x 64 diagonal block in registers: lane = column, 64 registers = rows; the pivot
//          row reaches the other lanes with v_readlane (scalar broadcast), no LDS, no barrier: ~8 us per block instead
//          of 64 x 3.5 us.
//   panel  U_kk^T X = A_k,rest blocked 4 x 16 rows on the fp32 matrix pipe: X_i = W_i (B_i - sum_{j<i} U_ji^T X_j) with the four
//          16 x 16 inverses W_i = (U_ii^T)^-1 formed by 16-step substitutions in every workgroup (round 1: one thread per column,
//          64 substitution steps, LDS-broadcast bound at 21 us per panel on the serial chain).
//   syrk   (panels are factored in PAIRS: a 64-row strip update lets the second panel of a pair be factored, then one
//          rank-128 update covers everything behind the pair -- the rank-64 form re-read and re-wrote the trailing
//          matrix every 64 columns and was HBM-bound at d >= 4096)
//          A_ij -= sum_t X[t][i] X[t][j] for the upper tiles i <= j of the trailing matrix on the fp32 matrix pipe
//          (v_mfma_f32_16x16x4_f32 is an exact fp32 fma chain): both operands are "row = t, 16 consecutive columns"
//          fragments of the same row panel, so the structure is K7's (hessian.hip) with fp32 in place of fp64 and
//          C -= in place of H +=; one 64-row LDS stage, 128/64/32-column tiles by how many tiles there are.
//   finish LT[c][j] = U[c][j] * (1 / U[c][c]) for j > c, 0 elsewhere  (the reciprocal-multiply of vector_balance.py:172).
// A non-positive pivot sets *info = column + 1 (LAPACK convention); the host wrapper raises like torch does.
#include "common.h"

namespace {

constexpr int NB = 64;
bool g_chol_old_syrk = false;          // tests: force the guarded round-1 trailing-update kernel
bool g_chol_unblocked_diag = false;    // tests / A-B: the 64-step single-wave factorisation of rounds 1-2 for the diagonal block

__device__ __forceinline__ float lane_bcast(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// ---- diag: A[k0:k0+64, k0:k0+64] (upper part valid) -> U_kk in place ------------------------------------------------
// a[i] -= U[j][i] * a[j] for four rows i: the four v_readlane (scalar broadcast of U[j][i] out of lane i of row j) and
// their four fma as ONE asm block.  Written with the builtin, hipcc clusters all 63 v_readlane of a row ahead of the
// fmas and spills ~1900 SGPRs through v_writelane.  Wait states: gfx950 needs 2 between a VALU writing an SGPR and a
// VALU reading it (hipcc itself puts `s_nop 1` between v_cmp and v_cndmask) -- three instructions separate each pair
// here; the leading s_nop covers a[j] having been written by the instruction right before the block.
template <int J, int I> __device__ __forceinline__ void upd4(float (&a)[NB])
{
    float t0, t1, t2, t3;
    asm volatile("s_nop 1\n\t"
                 "v_readlane_b32 %4, %8, %9\n\t"
                 "v_readlane_b32 %5, %8, %10\n\t"
                 "v_readlane_b32 %6, %8, %11\n\t"
                 "v_readlane_b32 %7, %8, %12\n\t"
                 "v_fma_f32 %0, -%4, %8, %0\n\t"
                 "v_fma_f32 %1, -%5, %8, %1\n\t"
                 "v_fma_f32 %2, -%6, %8, %2\n\t"
                 "v_fma_f32 %3, -%7, %8, %3"
                 : "+v"(a[I]), "+v"(a[I + 1]), "+v"(a[I + 2]), "+v"(a[I + 3]), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3)
                 : "v"(a[J]), "n"(I), "n"(I + 1), "n"(I + 2), "n"(I + 3));
}
template <int J, int I> __device__ __forceinline__ void upd1(float (&a)[NB])
{
    float t0;
    asm volatile("s_nop 1\n\t"
                 "v_readlane_b32 %1, %2, %3\n\t"
                 "s_nop 1\n\t"
                 "v_fma_f32 %0, -%1, %2, %0"
                 : "+v"(a[I]), "=&s"(t0)
                 : "v"(a[J]), "n"(I));
}
template <int J, int I> __device__ __forceinline__ void row_update(float (&a)[NB])
{
    if constexpr (I + 4 <= NB) {
        upd4<J, I>(a);
        row_update<J, I + 4>(a);
    } else if constexpr (I < NB) {
        upd1<J, I>(a);
        row_update<J, I + 1>(a);
    }
}
template <int J> __device__ __forceinline__ void factor_rows(float (&a)[NB], int l, int64_t k0, int *info)
{
    if constexpr (J < NB) {
        const float piv = lane_bcast(a[J], J);
        if (!(piv > 0.f) && l == 0) atomicCAS(info, 0, (int)(k0 + J + 1));
        // 1/sqrt(piv): v_rsq_f32 (1 ulp) + one Newton step -- the libm sqrtf + IEEE division pair costs ~40 instructions
        // per row, a third of this single-wave kernel (17.5 -> 11 us per block); row J = a * r, u_JJ = piv * r
        float r = __builtin_amdgcn_rsqf(piv);
        r = fmaf(0.5f * r, fmaf(-piv * r, r, 1.f), r);
        a[J] = (l >= J) ? a[J] * r : 0.f;                         // row J of U (lane J holds u_JJ); lanes < J: 0
        row_update<J, J + 1>(a);                                 // only lanes >= i of row i are read later
        factor_rows<J + 1>(a, l, k0, info);
    }
}

// ---- the blocked form of the same factorisation (round 3) --------------------------------------------------------------------------------
// factor_rows is 2016 dependent (v_readlane, v_fma) pairs at ~13 cycles each: 11 us of one wave's instruction stream, 128 times per
// d = 8192 on the serial chain.  Blocked 4 x 16 rows: the 16 rows of a block are factored the same way (120 pairs), everything below
// them is updated at once on the matrix pipe -- S = U_blk^T U_blk (fp32 MFMA chains, operands from a 16 x 64 LDS image of the finished
// rows), subtracted from the register image through LDS (the D layout holds 4 rows of a column per lane group, the register image one
// column per lane).  Same arithmetic up to the order of the 16 products per element.
template <int J, int I, int END> __device__ __forceinline__ void row_update_to(float (&a)[NB])
{
    if constexpr (I + 4 <= END) {
        upd4<J, I>(a);
        row_update_to<J, I + 4, END>(a);
    } else if constexpr (I < END) {
        upd1<J, I>(a);
        row_update_to<J, I + 1, END>(a);
    }
}
template <int J, int END> __device__ __forceinline__ void factor_block(float (&a)[NB], int l, int64_t k0, int *info)
{
    if constexpr (J < END) {
        const float piv = lane_bcast(a[J], J);
        if (!(piv > 0.f) && l == 0) atomicCAS(info, 0, (int)(k0 + J + 1));
        float r = __builtin_amdgcn_rsqf(piv);
        r = fmaf(0.5f * r, fmaf(-piv * r, r, 1.f), r);
        a[J] = (l >= J) ? a[J] * r : 0.f;
        row_update_to<J, J + 1, END>(a);                          // only the rows of this 16-row block
        factor_block<J + 1, END>(a, l, k0, info);
    }
}
template <int KB> __device__ __forceinline__ void blocked_steps(float (&a)[NB], int l, int64_t k0, int *info, float (*Ub)[NB + 1], float (*Sf)[NB + 1])
{
    if constexpr (KB < 4) {
        constexpr int R0 = 16 * KB, R1 = R0 + 16;
        factor_block<R0, R1>(a, l, k0, info);
        if constexpr (KB < 3) {
            const int j = l & 15, g = l >> 4;
#pragma unroll
            for (int k = 0; k < 16; ++k) Ub[k][l] = a[R0 + k];        // finished rows (0 left of the diagonal)
            __syncthreads();
            // S[i][c] = sum_k U[R0 + k][i] U[R0 + k][c] for the tiles (ti, tj), KB < ti <= tj
#pragma unroll
            for (int ti = KB + 1; ti < 4; ++ti)
#pragma unroll
                for (int tj = ti; tj < 4; ++tj) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Ub[4 * s4 + g][16 * ti + j], Ub[4 * s4 + g][16 * tj + j], acc, 0, 0, 0);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Sf[16 * ti + 4 * g + reg][16 * tj + j] = acc[reg];     // D: row 4g + reg, col j
                }
            __syncthreads();
            const int tjl = l >> 4;                                    // this lane's column tile: rows of the tiles ti <= tjl are its upper part
#pragma unroll
            for (int i = R1; i < NB; ++i)
                if ((i >> 4) <= tjl) a[i] -= Sf[i][l];
            __syncthreads();                                           // Ub / Sf are rewritten by the next block
        }
        blocked_steps<KB + 1>(a, l, k0, info, Ub, Sf);
    }
}

// FULL: the block lies inside the matrix (every step but a ragged last one): unconditional loads, no per-row branches.
template <bool FULL, bool BLOCKED>
__global__ __launch_bounds__(64) void chol_diag_kernel(float *A, int64_t d, int64_t k0, int *info)
{
    __shared__ float Ub[BLOCKED ? 16 : 1][NB + 1], Sf[BLOCKED ? NB : 1][NB + 1];
    const int l = threadIdx.x;
    const int64_t c = k0 + l;
    float *Ac = A + k0 * d + (FULL ? c : (c < d ? c : k0));
    float a[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        if constexpr (FULL) {
            a[r] = Ac[(int64_t)r * d];
        } else {                                                    // outside the matrix: identity
            const bool in = (k0 + r < d) && (c < d);
            const float v = Ac[(k0 + r < d ? (int64_t)r : 0) * d];
            a[r] = in ? v : (r == l ? 1.f : 0.f);
        }
    }
    if constexpr (BLOCKED) blocked_steps<0>(a, l, k0, info, Ub, Sf);
    else factor_rows<0>(a, l, k0, info);
#pragma unroll
    for (int r = 0; r < NB; ++r) {
        if constexpr (FULL) {
            if (l >= r) Ac[(int64_t)r * d] = a[r];
        } else {
            if (k0 + r < d && c < d && l >= r) Ac[(int64_t)r * d] = a[r];
        }
    }
}

// ---- panel: X = U_kk^-T A[k0:k0+64, c] for the columns c >= k0+64, on the matrix pipe ---------------------------------------------
// (only launched when columns remain, so the 64 rows of the step are always inside the matrix)
// Blocked 4 x 16 rows:  X_i = W_i ( B_i - sum_{j<i} U_ji^T X_j ),  W_i = (U_ii^T)^-1,  i = 0..3, as fp32 MFMAs
// (v_mfma_f32_16x16x4_f32: exact fp32 fma chains).  The round-1 form -- one thread per column, 64 substitution steps, U_kk read as
// broadcast float4s from LDS -- was LDS-bandwidth bound at 21 us and sat, with the 17 us diagonal block, on the serial chain of
// the factorisation (2.4 of the 3.5 ms at d = 4096).  One wave = one
// 16-column tile of the panel: 64 x 16 values in the MFMA D layout (lane = column, 4 rows per register group), which IS the
// B-operand layout, so X_j feeds the next products without a shuffle; U_kk^T and the four 16 x 16 inverses are A operands read as
// float4 from LDS.  The inverses (16-step substitutions, 16 lanes each) are recomputed by every workgroup: ~1 us.
__global__ __launch_bounds__(256) void chol_panel_mfma_kernel(float *A, int64_t d, int64_t k0)
{
    __shared__ __attribute__((aligned(16))) float UT[NB][NB + 4];        // UT[c][r] = U[r][c] for r <= c, else 0
    __shared__ __attribute__((aligned(16))) float Wi[4][16][20];         // Wi[i][r][k] = (U_ii^T)^-1 [r][k]
    {
        const float *U = A + k0 * d + k0;
        constexpr int NU = NB * NB / 256;
        const int r0 = threadIdx.x >> 6, cc = threadIdx.x & 63;
        float t1[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) t1[u] = U[(int64_t)(4 * u + r0) * d + cc];     // in flight together; the select comes after
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int r = 4 * u + r0;
            UT[cc][r] = cc >= r ? t1[u] : 0.f;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 16) {
        // column t of the inverse of L = U_ii^T (lower):  L[r][k] = UT[16i + r][16i + k]
        const int i = wave, t = lane;
        float x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sacc = 0.f;
#pragma unroll
            for (int k = 0; k < r; ++k) sacc = fmaf(UT[16 * i + r][16 * i + k], x[k], sacc);      // x[k] = 0 for k < t
            const float diag = UT[16 * i + r][16 * i + r];
            x[r] = r < t ? 0.f : (r == t ? 1.f / diag : -sacc / diag);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Wi[i][r][t] = x[r];
    }
    __syncthreads();
    const int64_t c0 = k0 + NB + ((int64_t)blockIdx.x * 4 + wave) * 16;
    if (c0 >= d) return;
    const int jc = lane & 15, g = lane >> 4;
    const int64_t col = c0 + jc;
    const bool in = col < d;
    float *Ac = A + k0 * d + (in ? col : k0 + NB);
    f32x4_t X[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) X[i][reg] = in ? Ac[(int64_t)(16 * i + 4 * g + reg) * d] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4_t t = X[i];
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const float4 a4 = *reinterpret_cast<const float4 *>(&UT[16 * i + jc][16 * j + 4 * g]);   // A[r][k] = U[16j + k][16i + r]
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(-a4.x, X[j][0], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(-a4.y, X[j][1], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(-a4.z, X[j][2], t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(-a4.w, X[j][3], t, 0, 0, 0);
        }
        const float4 w4 = *reinterpret_cast<const float4 *>(&Wi[i][jc][4 * g]);
        f32x4_t xi = {0.f, 0.f, 0.f, 0.f};
        xi = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, t[0], xi, 0, 0, 0);
        xi = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, t[1], xi, 0, 0, 0);
        xi = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, t[2], xi, 0, 0, 0);
        xi = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, t[3], xi, 0, 0, 0);
        X[i] = xi;
    }
    if (in) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Ac[(int64_t)(16 * i + 4 * g + reg) * d] = X[i][reg];
    }
}

// ---- syrk: trailing update of the upper tiles ------------------------------------------------------------------------------
__device__ __forceinline__ void tri_tile(int t, int &I, int &J)     // t = J (J + 1) / 2 + I with I <= J
{
    J = (int)((sqrtf(8.f * (float)t + 1.f) - 1.f) * 0.5f);
    while (J * (J + 1) / 2 > t) --J;
    while ((J + 1) * (J + 2) / 2 <= t) ++J;
    I = t - J * (J + 1) / 2;
}

// prow0: first row of the panel(s) the update is made of; nk: 64-row panels (1 or 2) accumulated before the single
// read-modify-write of the tile (rank-128 trailing updates halve the HBM traffic the rank-64 form was bound by);
// base: first column / row of the region updated; strip: only the first 64-row tile row of it (the rows the second
// panel of a pair needs before it can be factored).
template <int WT>
__global__ __launch_bounds__(256, 2) void chol_syrk_kernel(float *A, int64_t d, int64_t prow0, int nk, int64_t base, int strip)
{
    constexpr int BN = 32 * WT, LDW = BN + 16, EPT = BN / 16;       // EPT floats per thread per token row (16 threads/row)
    extern __shared__ __attribute__((aligned(16))) float cs[];     // [2 sides][NB][LDW]
    int I, J;
    if (strip) { I = 0; J = blockIdx.x; }
    else tri_tile(blockIdx.x, I, J);
    const bool diag = I == J;
    const int64_t i0 = base + (int64_t)I * BN, j0 = base + (int64_t)J * BN;
    const float *P = A + prow0 * d;                                 // the row panel X of the current stage: P[t][c]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int stok = tid >> 4, scol = (tid & 15) * EPT;
    const bool vec = (d % 4 == 0) && ((base & 3) == 0);

    auto stage = [&](int side, int64_t c0) {
#pragma unroll
        for (int ps = 0; ps < NB / 16; ++ps) {
            const int t = 16 * ps + stok;
            float *dst = cs + (side * NB + t) * LDW + scol;
            const int64_t c = c0 + scol;
            if constexpr (EPT >= 4) {
#pragma unroll
                for (int e = 0; e < EPT; e += 4) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (vec && c + e + 4 <= d) {
                        v = *reinterpret_cast<const float4 *>(P + (int64_t)t * d + c + e);
                    } else {
                        if (c + e + 0 < d) v.x = P[(int64_t)t * d + c + e + 0];
                        if (c + e + 1 < d) v.y = P[(int64_t)t * d + c + e + 1];
                        if (c + e + 2 < d) v.z = P[(int64_t)t * d + c + e + 2];
                        if (c + e + 3 < d) v.w = P[(int64_t)t * d + c + e + 3];
                    }
                    *reinterpret_cast<float4 *>(dst + e) = v;
                }
            } else {
                float2 v = make_float2(0.f, 0.f);
                if (c + 0 < d) v.x = P[(int64_t)t * d + c + 0];
                if (c + 1 < d) v.y = P[(int64_t)t * d + c + 1];
                *reinterpret_cast<float2 *>(dst) = v;
            }
        }
    };
    f32x4_t acc[WT][WT];
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y) acc[x][y] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const float *As = cs + wi * (WT * 16) + (lane & 15);
    const float *Bs = cs + (diag ? 0 : 1) * NB * LDW + wj * (WT * 16) + (lane & 15);
    for (int sg = 0; sg < nk; ++sg) {
    if (sg) {
        __syncthreads();                                            // every read of the previous panel's stage retired
        P += (int64_t)NB * d;
    }
    stage(0, i0);
    if (!diag) stage(1, j0);
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < NB / 4; ++ks) {
        const int row = ks * 4 + (lane >> 4);
        float a[WT], b[WT];
#pragma unroll
        for (int x = 0; x < WT; ++x) a[x] = As[row * LDW + x * 16];
#pragma unroll
        for (int y = 0; y < WT; ++y) b[y] = Bs[row * LDW + y * 16];
#pragma unroll
        for (int x = 0; x < WT; ++x)
#pragma unroll
            for (int y = 0; y < WT; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[x][y], 0, 0, 0);
    }
    }
    // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int64_t r = i0 + wi * (WT * 16) + x * 16 + 4 * (lane >> 4) + reg;
                const int64_t c = j0 + wj * (WT * 16) + y * 16 + (lane & 15);
                if (r < d && c < d) A[r * d + c] -= acc[x][y][reg];
            }
}


// (the stage is written as macros: hipcc keeps the float4 arrays in registers only when the loads and the LDS writes are inlined text --
//  behind a lambda or a forceinline function taking the arrays by reference they went through scratch memory)
#define QA_SYRK_FETCH(S0, S1, PP)                                                                                  \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) _Pragma("unroll") for (int v = 0; v < NV; ++v)              \
    {                                                                                                              \
        const float *row_ = (PP) + (int64_t)(16 * ps) * d + 4 * v;                                                 \
        S0[ps * NV + v] = *reinterpret_cast<const f32x4_t *>(row_ + i0);                                            \
        S1[ps * NV + v] = *reinterpret_cast<const f32x4_t *>(row_ + j0); /* diagonal tile: the same lines, L1 hits */ \
    }
#define QA_SYRK_PUT(S0, S1)                                                                                        \
    _Pragma("unroll") for (int ps = 0; ps < NP; ++ps) _Pragma("unroll") for (int v = 0; v < NV; ++v)              \
    {                                                                                                              \
        *reinterpret_cast<f32x4_t *>(cput + (16 * ps) * LDW + 4 * v) = S0[ps * NV + v];                             \
        *reinterpret_cast<f32x4_t *>(cput + (NB + 16 * ps) * LDW + 4 * v) = S1[ps * NV + v];                        \
    }
template <int WT, int LDW> __device__ __forceinline__ void syrk_mfmas(f32x4_t (&acc)[WT][WT], const float *As, const float *Bs, int lane)
{
#pragma unroll 4
    for (int ks = 0; ks < NB / 4; ++ks) {
        const int row = ks * 4 + (lane >> 4);
        float a[WT], b[WT];
#pragma unroll
        for (int x = 0; x < WT; ++x) a[x] = As[row * LDW + x * 16];
#pragma unroll
        for (int y = 0; y < WT; ++y) b[y] = Bs[row * LDW + y * 16];
#pragma unroll
        for (int x = 0; x < WT; ++x)
#pragma unroll
            for (int y = 0; y < WT; ++y) acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[x], b[y], acc[x][y], 0, 0, 0);
    }
}

// The same update for tiles that lie wholly inside the matrix (every tile when d and `base` are multiples of the tile size: all of the
// model shapes), written for the memory system: round 1's kernel guarded every element (r < d && c < d), which hipcc turns into one
// branch + one dependent round trip per load -- 16 serial trips per stage and 64 serial read-modify-writes per wave in the epilogue,
// ~118 us for a 7 us tile's worth of MFMAs (profiles/r03y_k8_trace.txt: 4.25 of 10.5 ms at d = 8192).  Here every load of a phase is in
// flight at once (stage: 16 float4 per thread; epilogue: 64 dwords per lane), and the two workgroups of a CU cover each other's
// memory phases with their MFMAs.  (Prefetching the C tile and the second panel under the first panel's MFMAs wants 256 registers.)
template <int WT, int NK>
__global__ __launch_bounds__(256, 2) void chol_syrk_full_kernel(float *A, int64_t d, int64_t prow0, int64_t base, int strip)
{
    constexpr int BN = 32 * WT, LDW = BN + 16, EPT = BN / 16, NV = EPT / 4, NP = NB / 16;
    static_assert(EPT % 4 == 0, "float4 staging");
    extern __shared__ __attribute__((aligned(16))) float cs[];     // [2 sides][NB][LDW]
    int I, J;
    const int abl = 0;
    if (strip) { I = 0; J = blockIdx.x; }
    else tri_tile(blockIdx.x, I, J);
    const bool diag = I == J;
    const int64_t i0 = base + (int64_t)I * BN, j0 = base + (int64_t)J * BN;
    const float *P = A + prow0 * d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int stok = tid >> 4, scol = (tid & 15) * EPT;

    f32x4_t acc[WT][WT], cc[WT][WT];
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y) acc[x][y] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const float *As = cs + wi * (WT * 16) + (lane & 15);
    const float *Bs = cs + NB * LDW + wj * (WT * 16) + (lane & 15);
    float *Ct = A + (i0 + wi * (WT * 16) + 4 * (lane >> 4)) * d + j0 + wj * (WT * 16) + (lane & 15);   // D layout: col = lane & 15, row = 4 (lane >> 4) + reg

    // Software pipeline (the two workgroups of a CU start together and run the same phases: they do NOT cover each other's memory
    // phases -- ablations at d = 8192: MFMAs 1.24 ms, staging 0.43, C read-modify-write 0.60, purely additive):
    //   stage(panel 0) | [panel 1's loads in flight] MFMAs(panel 0) | stage(panel 1) | [the C tile's loads in flight] MFMAs(panel 1) | C - acc
    // acc + one of {stage registers, C tile} live at a time: 128 + addressing registers, two waves per SIMD.
    f32x4_t s0[NP * NV], s1[NP * NV];                                // (native vectors: arrays of HIP's float4 struct live across the MFMAs went to scratch)
    (void)diag;
    (void)abl;
    const float *Pp = P + (int64_t)stok * d + scol;
    float *cput = cs + stok * LDW + scol;
    QA_SYRK_FETCH(s0, s1, Pp)
    QA_SYRK_PUT(s0, s1)
    __syncthreads();
    if constexpr (NK == 2) {
        f32x4_t t0[NP * NV], t1[NP * NV];                           // (arrays of their own: one written on two paths goes to scratch)
        QA_SYRK_FETCH(t0, t1, Pp + (int64_t)NB * d)
        __builtin_amdgcn_sched_barrier(0);                          // (hipcc otherwise sinks these loads below the 256 MFMAs they are to travel under)
        syrk_mfmas<WT, LDW>(acc, As, Bs, lane);
        __syncthreads();                                            // every read of panel 0's stage retired
        QA_SYRK_PUT(t0, t1)
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) cc[x][y][reg] = Ct[(int64_t)(x * 16 + reg) * d + y * 16];
    __builtin_amdgcn_sched_barrier(0);
    syrk_mfmas<WT, LDW>(acc, As, Bs, lane);
#pragma unroll
    for (int x = 0; x < WT; ++x)
#pragma unroll
        for (int y = 0; y < WT; ++y)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Ct[(int64_t)(x * 16 + reg) * d + y * 16] = cc[x][y][reg] - acc[x][y][reg];
}

template <int WT> int launch_syrk(float *A, int64_t d, int64_t prow0, int nk, int64_t base, bool strip, hipStream_t s)
{
    constexpr int BN = 32 * WT, LDW = BN + 16;
    const size_t lds = (size_t)2 * NB * LDW * sizeof(float);
    const int64_t rem = d - base, T = (rem + BN - 1) / BN;
    auto kern = chol_syrk_kernel<WT>;
    static QaPerDevice attr_done_dev;
    const int attr_done_d = attr_done_dev.dev();
    if ((attr_done_d < 0 || !attr_done_dev.done[attr_done_d])) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: cannot reserve %zu B of LDS", lds);
        if constexpr (WT >= 2) {
            if (hipFuncSetAttribute((const void *)chol_syrk_full_kernel<WT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
                hipFuncSetAttribute((const void *)chol_syrk_full_kernel<WT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: cannot reserve %zu B of LDS", lds);
        }
        if (attr_done_d >= 0) attr_done_dev.done[attr_done_d] = true;
    }
    const unsigned grid = (unsigned)(strip ? T : T * (T + 1) / 2);
    if constexpr (WT >= 2) {
        // every tile inside the matrix, float4-aligned rows: the branch-free kernel
        if (rem % BN == 0 && d % 4 == 0 && base % 4 == 0 && !g_chol_old_syrk) {
            if (nk == 2) chol_syrk_full_kernel<WT, 2><<<grid, 256, lds, s>>>(A, d, prow0, base, strip ? 1 : 0);
            else chol_syrk_full_kernel<WT, 1><<<grid, 256, lds, s>>>(A, d, prow0, base, strip ? 1 : 0);
            return QUIPAMD_OK;
        }
    }
    kern<<<grid, 256, lds, s>>>(A, d, prow0, nk, base, strip ? 1 : 0);
    return QUIPAMD_OK;
}

// trailing update of the region starting at `base` with nk panels from row prow0: tile size by how many tiles there are
static int trailing_update(float *A, int64_t d, int64_t prow0, int nk, int64_t base, hipStream_t s)
{
    const int64_t rem = d - base;
    auto ntiles = [&](int64_t bn) { const int64_t T = (rem + bn - 1) / bn; return T * (T + 1) / 2; };
    if (ntiles(128) >= 384) return launch_syrk<4>(A, d, prow0, nk, base, false, s);
    if (ntiles(64) >= 384) return launch_syrk<2>(A, d, prow0, nk, base, false, s);
    return launch_syrk<1>(A, d, prow0, nk, base, false, s);
}

// ---- out-of-place entry: LT <- H (grid-stride, 16-byte pieces when both pointers allow) --------------------------------------------------
__global__ __launch_bounds__(256) void chol_copy_kernel(const float *__restrict__ H, float *__restrict__ LT, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    if ((((uintptr_t)H | (uintptr_t)LT) & 15) == 0) {
        const size_t n4 = n / 4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
            reinterpret_cast<float4 *>(LT)[i] = reinterpret_cast<const float4 *>(H)[i];
        for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) LT[i] = H[i];
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) LT[i] = H[i];
    }
}

// ---- finish: LT[c][j] = U[c][j] * (1 / U[c][c]) for j > c, else 0, in place -------------------------------------------------
__global__ __launch_bounds__(256) void chol_finish_kernel(float *A, int64_t d)
{
    const int64_t c = blockIdx.x;
    const float rinv = 1.f / A[c * d + c];
    __syncthreads();                                             // every thread has the pivot before it is overwritten
    for (int64_t j = threadIdx.x; j < d; j += 256) A[c * d + j] = (j > c) ? A[c * d + j] * rinv : 0.f;
}

// side stream + events of the look-ahead, one set per device, created on first use (never destroyed: process lifetime)
struct ChoLook {
    hipStream_t side = nullptr;
    hipEvent_t panels[4] = {}, trail[4] = {};
    hipEvent_t pending = nullptr;
    bool ok = false;
};
ChoLook *chol_look()
{
    static ChoLook per_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    ChoLook &L = per_dev[dev];
    if (!L.ok) {
        // LOWEST priority: the side stream's grids fill every CU (2 workgroups each: 147 of 160 KB of LDS), and the chain's kernels on
        // the caller's stream must win the slots those workgroups free
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = 0;
        if (hipStreamCreateWithPriority(&L.side, hipStreamNonBlocking, least) != hipSuccess) return nullptr;
        for (int i = 0; i < 4; ++i)
            if (hipEventCreateWithFlags(&L.panels[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&L.trail[i], hipEventDisableTiming) != hipSuccess)
                return nullptr;
        L.ok = true;
    }
    L.pending = nullptr;
    return &L;
}
bool g_chol_no_lookahead = false, g_chol_force_lookahead = false;

}   // namespace

extern "C" void quipamd_cholesky_config(int old_syrk, int lookahead)
{
    g_chol_unblocked_diag = (old_syrk & 2) != 0;
    old_syrk &= 1;
    g_chol_old_syrk = old_syrk != 0;
    g_chol_no_lookahead = lookahead == 0;
    g_chol_force_lookahead = lookahead == 1;
}

extern "C" int quipamd_cholesky_lt(const float *H, float *LT, int64_t d, int *info, void *stream)
{
    QA_REQUIRE(d >= 0, QUIPAMD_ERR_SHAPE, "cholesky_lt: bad d");
    if (d == 0) return QUIPAMD_OK;
    QA_REQUIRE(H && LT && info, QUIPAMD_ERR_ARG, "cholesky_lt: null pointer");
    QA_REQUIRE(d <= (1 << 17), QUIPAMD_ERR_SHAPE, "cholesky_lt: d too large");
    hipStream_t s = (hipStream_t)stream;
    // (round 6: the out-of-place copy is a kernel of this library, not a runtime copy: one runtime command less in front of a chain that
    //  showed ONE transient non-finite result in ~6000 runs of a stress loop -- DESIGN.md section 4 -- and 5-10 us less per call)
    if ((const void *)H != (const void *)LT)
        chol_copy_kernel<<<(unsigned)((((size_t)d * d + 3) / 4 + 255) / 256 < 65535u * 16u ? (((size_t)d * d + 3) / 4 + 255) / 256 : 65535u * 16u), 256, 0, s>>>(H, LT, (size_t)d * d);
    if (hipMemsetAsync(info, 0, sizeof(int), s) != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: memset failed");
    auto diag = [&](int64_t k0) {
        if (k0 + NB <= d) {
            if (g_chol_unblocked_diag) chol_diag_kernel<true, false><<<1, 64, 0, s>>>(LT, d, k0, info);
            else chol_diag_kernel<true, true><<<1, 64, 0, s>>>(LT, d, k0, info);
        } else {
            chol_diag_kernel<false, false><<<1, 64, 0, s>>>(LT, d, k0, info);
        }
    };
    // two 64-row panels per trailing update: diag, panel, [64-row strip update so the second panel can be factored],
    // diag, panel, then ONE rank-128 update of everything behind the pair.
    // Look-ahead (round 3): the trailing update is cut in two.  The 128 rows the NEXT pair factors are updated on the caller's stream
    // (a strip launch); everything below them goes to a side stream and runs under the next pair's diag / panel / strip / diag / panel
    // chain (~60 us of launches that occupy one to a few dozen CUs).  The three cross-stream hand-overs per pair cost ~15 us, and the
    // chain's kernels share the CUs with the update: measured on one box (profiles/r03z2_k8_ab.txt) a loss up to d = 8192 (6.2 -> 6.8
    // ms), even at 11008, a gain at 16384 (26.1 -> 24.8 ms) -- on from d = 12288 (g_chol_force_lookahead: the tests run it from 1024).
    ChoLook *look = (d >= (g_chol_force_lookahead ? 1024 : 12288) && !g_chol_no_lookahead) ? chol_look() : nullptr;
    int pair = 0;
    for (int64_t k0 = 0; k0 < d; k0 += 2 * NB, ++pair) {
        diag(k0);
        if (d - k0 - NB <= 0) break;
        chol_panel_mfma_kernel<<<(unsigned)((d - k0 - NB + 63) / 64), 256, 0, s>>>(LT, d, k0);
        int rc = launch_syrk<2>(LT, d, k0, 1, k0 + NB, true, s);                  // rows k0+64 .. k0+127, all columns behind
        if (rc != QUIPAMD_OK) return rc;
        diag(k0 + NB);
        if (d - k0 - 2 * NB <= 0) break;
        chol_panel_mfma_kernel<<<(unsigned)((d - k0 - 2 * NB + 63) / 64), 256, 0, s>>>(LT, d, k0 + NB);
        const int64_t base = k0 + 2 * NB;
        if (look && d - base > 2 * NB && (d - base) % (2 * NB) == 0) {
            hipEvent_t evp = look->panels[pair & 3], evt = look->trail[pair & 3];
            if (hipEventRecord(evp, s) != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: event record failed");
            // rows base .. base+127 wait for the previous pair's side-stream update of the same rows
            if (pair > 0 && hipStreamWaitEvent(s, look->trail[(pair - 1) & 3], 0) != hipSuccess)
                return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: stream wait failed");
            rc = launch_syrk<4>(LT, d, k0, 2, base, true, s);
            if (rc != QUIPAMD_OK) return rc;
            if (hipStreamWaitEvent(look->side, evp, 0) != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: stream wait failed");
            rc = trailing_update(LT, d, k0, 2, base + 2 * NB, look->side);
            if (rc != QUIPAMD_OK) return rc;
            if (hipEventRecord(evt, look->side) != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: event record failed");
            look->pending = evt;
        } else {
            if (look && look->pending) {                                          // the tail of the matrix: back on one stream
                if (hipStreamWaitEvent(s, look->pending, 0) != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: stream wait failed");
                look->pending = nullptr;
            }
            rc = trailing_update(LT, d, k0, 2, base, s);
            if (rc != QUIPAMD_OK) return rc;
        }
    }
    if (look && look->pending) {
        if (hipStreamWaitEvent(s, look->pending, 0) != hipSuccess) return qa_fail(QUIPAMD_ERR_LAUNCH, "cholesky_lt: stream wait failed");
        look->pending = nullptr;
    }
    chol_finish_kernel<<<(unsigned)d, 256, 0, s>>>(LT, d);
    QA_LAUNCH_CHECK("cholesky_lt");
    return QUIPAMD_OK;
}
