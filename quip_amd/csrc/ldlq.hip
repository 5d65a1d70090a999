// ldlq.hip -- K4: LDLQ adaptive rounding with lazy-batch block updates.
//
// Takes over round_ldl / round_ldl_block (vector_balance.py:155-199, 218-257; n_greedy_passes = 0):
//     for i = d-1 .. 0:  q_i = clamp(floor(w_i + sum_{j>i} (w_j - q_j) L[j,i] + eta_i), 0, 2^b - 1)
// L = unit-lower Cholesky factor of H (vector_balance.py:171-173).  Rows of W are independent, columns are a
// length-d dependent chain, so:
//   * one workgroup (8 waves: 4 'chain' + 4 'far') owns 16 rows for the WHOLE sweep -- no inter-workgroup communication,
//     one launch; the in-block chain of block k overlaps the far field of block k+1;
//   * columns are processed in 128-wide blocks from the top (the reference's `--lazy_batch` blocking,
//     vector_balance.py:243-257).  For block [i1,i2):
//       phase A  far[16 x 128] = Err[16 x (d-i2)] * L[i2:, i1:i2]  as fp32 MFMA (v_mfma_f32_16x16x4_f32, exact
//                fp32 fmaf chains): the true lazy-batch GEMM (the reference re-does it per column, :254).
//                A operand = this workgroup's error rows (global, written by itself), B operand = rows of
//                LT = L^T; both staged per 64-k step into wave-private, XOR-swizzled LDS slabs by full-line buffer
//                DMA, the next step's DMA in flight under the current step's 32 MFMAs; wave w owns column tiles
//                w and w+4.
//       phase B  in-block error feedback, right-looking: wave w owns rows 4w..4w+3, lane = column (2 per lane).
//                Step i: every lane rounds its own column (only lane i's value is final), the error of
//                column i is broadcast with v_readlane, and acc[c] = fma(err_i, L[i][c], acc[c]) runs on all
//                lanes with the L row read from an LDS image of the 128x128 diagonal block (zero above the
//                diagonal, so finished columns are untouched).
//   * summation order is fixed and documented in oracle/ldlq_oracle.c (oracle_round_ldl_kernel_order), which
//     this kernel must match BIT-EXACTLY (tests/test_gpu_ldlq.py).
//
// Work: 2*m*d^2/2 far-field MACs on the fp32 MFMA pipe (157 TF peak) + m*d*64 in-block FMAs on VALU;
// traffic: the LT panel is re-read by every workgroup from L2 (d^2/2*4 B each), W/E/codes once from HBM.
#include "common.h"

namespace {

constexpr int BS = 128;        // column block (vector_balance.py:222 blocksize)
constexpr int LDS_LD = BS + 1; // padded leading dimension of the diagonal-block image
constexpr int SLAB = 16 * 64 * 4;   // one 16-row x 64-k fp32 operand slab of the far field (4 KiB)
typedef __attribute__((address_space(3))) void lds_void_t;

struct LdlqArgs {
    const float *W;     // [m,d] grid coordinates
    const float *LT;    // [d,d] LT[c][j] = L[j][c], j > c
    const float *eta;   // [m,d] or null
    uint8_t *codes;     // [m,d]   (modes 0, 1)
    float *E;           // [m,d] workspace: w - q
    int64_t m, d;
    float maxq;
    const float *hd;    // [d]    mode 2: diagonal of the normalised H
    float *Wout;        // [m,d]  mode 2: the updated (unclamped) values; mode 3: the dequantised weights
    // mode 3 (OPTQ in weight units, quantiser of quant.py:6-21 inside the chain)
    float *gscale, *gzero;   // groupsize > 0: OUT [m, d / groupsize] (found in the kernel); groupsize <= 0: IN [m]
    int groupsize;           // 16 / 32 / 64 / 128 or <= 0 (one (scale, zero) per row)
    int sym, qfn_c;          // Quantizer.configure(sym=...), qfn 'c' (clamp before round, quant.py:17-21)
};

__device__ __forceinline__ float round_col(float w, float acc, float eta, float maxq)
{
    const float x = (w + acc) + eta;                      // vector_balance.py:180 / :253-256
    return fminf(fmaxf(floorf(x), 0.0f), maxq);
}

// 8 waves: waves 0-3 ("chain") run the in-block error feedback of block k, waves 4-7 ("far") accumulate the far field
// of block k+1 at the same time (the part that does not depend on block k: columns >= i2), one VALU-bound and one
// MFMA-bound job per SIMD.  After a barrier the far waves add block k's own 128 columns and publish Ftile for block k+1
// while the chain waves stage the next diagonal block of L.  Two barriers per block.
// MODE: which error a rounded column feeds back.  0 = LDLQ, w - q with the ORIGINAL w (vector_balance.py:179);
// 1 = OPTQ/GPTQ, (w + acc) - q with the UPDATED w (gptq.py:80-87) -- the only difference between the two recurrences
// once the feedback matrix is prepared accordingly (quipamd_gptq_round).
// MODE 2 = one greedy coordinate-descent pass of LDLQ's post-processing (vector_balance.py:186-196, 263-288):
//   Hs_i = (s H)_i - sum_{j > i} eps_j H[j][i];  new_i = round(wr_i - Hs_i / H[i][i]);  eps_i = wr_i - new_i
// -- the same right-to-left recurrence with (s H) precomputed by a GEMM (passed in the eta slot), the feedback matrix
// -H (strictly upper) and eps as the quantity fed back; values are written as floats, unclamped (the reference clamps
// after the whole pass).
// MODE 3 = OPTQ/GPTQ in WEIGHT units with the reference's quantiser inside the chain (gptq.py:60-87 with quant.py:6-21):
//   q = scale * (clamp(round(w' / scale) + zero, 0, maxq) - zero)   (qfn a; qfn c clamps first), feedback (w' - q).
// With groupsize > 0 the (scale, zero) of a group are found IN the kernel the way Quantizer.find_params_qfna does
// (quant.py:57-94, perchannel, mse off) from the columns of the group as they stand when the group's 128-column block
// starts -- far field of all finished blocks applied, nothing of the current block (gptq.py:72-75 reads the block-lazy W,
// not W1): exactly the value `w + Ftile` the chain starts from.  Groups must not straddle a block: groupsize | 128.
// RG (round 3): row groups of 16 per workgroup.  At m >= 8192 a workgroup owns 32 rows: the 4 far waves multiply every staged LT slab
// against the error rows of BOTH groups -- 16 KiB of slabs per 64 MFMAs instead of 12 KiB per 32 -- because the far field is bound by what
// a CU can pull from L2 (every workgroup streams all of L: 21 B/clk/CU at 4096^2), not by the matrix pipe; the 4 chain waves run the
// in-block chain of the two groups one after the other (it hides under the far field wherever a few thousand columns lie above the
// block; 8 chain waves would cap the kernel at 168 registers, 8 rows per wave at once spilled 33).  The summation order of every row is
// unchanged (the kernel-order oracle still matches bit for bit).
template <int MODE, int RG>
__global__ __launch_bounds__(512) void ldlq_kernel(LdlqArgs A)
{
    constexpr int NCH = 4, ROWS = 16 * RG, NSL = RG + 2;            // chain waves; rows per workgroup; slabs per far wave [A x RG | B0 | B1]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *Ldiag = smem;                       // [BS][LDS_LD]: Ldiag[i][c] = L[i1+i][i1+c] (c < i), else 0
    float *Ftile = smem + BS * LDS_LD;         // [ROWS][BS] far-field results: written in phase Y, read at the start of the next phase X
    char *slabs = reinterpret_cast<char *>(smem + BS * LDS_LD + ROWS * BS);     // 4 far waves x NSL slabs x 4 KiB

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool is_chain = wave < NCH;
    const int cw = wave & 3;                               // chain: rows 4cw .. 4cw+3 of a group; far: column-tile pair
    const int64_t d = A.d;
    const int64_t r0 = (int64_t)blockIdx.x * ROWS;
    const int fr = lane & 15, kq = lane >> 4;              // MFMA fragment coordinates

    // ---- diagonal block of L (strictly lower part) -> LDS, by the 256 threads of the chain waves ----------------------
    // 128 x 128 floats = 16 float4 per thread, all 16 loads issued before the first is consumed (the scalar
    // one-element-per-iteration form was a 64-deep chain of L2 round trips: 0.53 ms of the 3.0 ms at 4096^2).
    auto stage_diag = [&](int64_t i1, int cnt) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = u * 256 + (int)threadIdx.x;            // 0 .. 4095 = c * 32 + i4  (threadIdx.x < 256 here)
            const int c = idx >> 5, i = (idx & 31) * 4;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < cnt && i < cnt)                               // cnt is a multiple of 16: a float4 never straddles it
                v[u] = *reinterpret_cast<const float4 *>(A.LT + (i1 + c) * d + i1 + i);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = u * 256 + (int)threadIdx.x;
            const int c = idx >> 5, i = (idx & 31) * 4;
            Ldiag[(i + 0) * LDS_LD + c] = (c < i + 0) ? v[u].x : 0.f;
            Ldiag[(i + 1) * LDS_LD + c] = (c < i + 1) ? v[u].y : 0.f;
            Ldiag[(i + 2) * LDS_LD + c] = (c < i + 2) ? v[u].z : 0.f;
            Ldiag[(i + 3) * LDS_LD + c] = (c < i + 3) ? v[u].w : 0.f;
        }
    };

    // ---- far field on the fp32 matrix pipe: acc{0,1}[16 x 16] += Err[16 x (kend - kbeg)] * L[kbeg:kend, columns] --------
    // for the column block whose LT rows start at nb1 (nbcnt rows); far wave cw owns column tiles cw and cw + 4.
    // Operands go through wave-private LDS slabs filled by buffer DMA in FULL 256-byte row segments (fragment-shaped
    // global loads -- 16 rows x 64 B per instruction -- kept this phase request-rate bound; deeper register prefetch made it
    // worse).  Per 64-k step a wave stages three 16 x 64 slabs (its copy of the error rows, LT rows of tile 0, of tile 1),
    // XOR-swizzled through the DMA's source address so the ds_read_b128 fragment reads are conflict free; the fragments
    // of a step are read into registers, THEN the next step's DMA is issued, so it flies under the 32 MFMAs.
    // k order: step k0 (64 wide), sub-step ss, lane group kq, element s -> k = k0 + 16 ss + 4 kq + s (oracle: ldlq_oracle.c).
    auto far_accumulate = [&](f32x4_t (&acc)[RG][2], int64_t kbeg, int64_t kend, int64_t nb1, int nbcnt) {
        const int t0 = cw, t1 = cw + 4, nt = nbcnt / 16;
        const bool v0 = t0 < nt, v1 = t1 < nt;
        if (!v0 || kbeg >= kend) return;
        char *sl = slabs + cw * (NSL * SLAB);                                       // [A (x RG) | B0 | B1], 4 KiB each
        // ONE descriptor for the workgroup's error rows (rows past m are past its end and read zeros); one per group spilled scalars into
        // the DMA loop
        const int64_t erows = (A.m - r0) < ROWS ? (A.m - r0) : ROWS;
        __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc((void *)(A.E + r0 * d), 0, (int)(erows * d * 4), 0x00020000);
        __amdgpu_buffer_rsrc_t lrs = __builtin_amdgcn_make_buffer_rsrc((void *)(A.LT + nb1 * d), 0, (int)((int64_t)nbcnt * d * 4), 0x00020000);
        // DMA instruction q (0..3) moves rows 4q .. 4q+3: lane L -> row 4q + (L >> 4), physical 16-B column L & 15,
        // logical column (L & 15) ^ row  (k = k0 + 4 * logical column)
        const uint32_t drow = lane >> 4, dpc = lane & 15;
        uint32_t rd[4];
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) rd[ss] = fr * 256 + ((((4 * ss + kq) ^ fr) & 15) << 4);
        const uint32_t OOB = 0xfffffff0u;
        auto issue = [&](int64_t k0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t row = 4 * q + drow;
                const uint32_t kcol = (uint32_t)k0 + 4 * ((dpc ^ row) & 15);
                const bool kin = kcol < (uint32_t)kend;
#pragma unroll
                for (int gp = 0; gp < RG; ++gp) {
                    const uint32_t off = kin ? ((16 * gp + row) * (uint32_t)d + kcol) * 4u : OOB;      // out of range reads 0
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ers, (lds_void_t *)(sl + gp * SLAB + q * 1024), 16, off, 0, 0, 0);
                }
                const uint32_t off0 = kin ? ((16 * t0 + row) * (uint32_t)d + kcol) * 4u : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, (lds_void_t *)(sl + RG * SLAB + q * 1024), 16, off0, 0, 0, 0);
                if (v1) {
                    const uint32_t off1 = kin ? ((16 * t1 + row) * (uint32_t)d + kcol) * 4u : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(lrs, (lds_void_t *)(sl + (RG + 1) * SLAB + q * 1024), 16, off1, 0, 0, 0);
                }
            }
        };
        issue(kbeg);
        for (int64_t k0 = kbeg; k0 < kend; k0 += 64) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // wave-private slabs: no barrier
            float4 fa[RG][4], fb0[4], fb1[4];
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
#pragma unroll
                for (int gp = 0; gp < RG; ++gp) fa[gp][ss] = *reinterpret_cast<const float4 *>(sl + gp * SLAB + rd[ss]);
                fb0[ss] = *reinterpret_cast<const float4 *>(sl + RG * SLAB + rd[ss]);
                fb1[ss] = v1 ? *reinterpret_cast<const float4 *>(sl + (RG + 1) * SLAB + rd[ss]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // slab reads retired before it is refilled
            if (k0 + 64 < kend) issue(k0 + 64);
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                if (k0 + 16 * ss >= kend) break;                                    // wave-uniform; padded k would only add 0*0
#pragma unroll
                for (int gp = 0; gp < RG; ++gp) {
                    acc[gp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gp][ss].x, fb0[ss].x, acc[gp][0], 0, 0, 0);
                    acc[gp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gp][ss].y, fb0[ss].y, acc[gp][0], 0, 0, 0);
                    acc[gp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gp][ss].z, fb0[ss].z, acc[gp][0], 0, 0, 0);
                    acc[gp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gp][ss].w, fb0[ss].w, acc[gp][0], 0, 0, 0);
                    if (v1) {
                        acc[gp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gp][ss].x, fb1[ss].x, acc[gp][1], 0, 0, 0);
                        acc[gp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gp][ss].y, fb1[ss].y, acc[gp][1], 0, 0, 0);
                        acc[gp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gp][ss].z, fb1[ss].z, acc[gp][1], 0, 0, 0);
                        acc[gp][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[gp][ss].w, fb1[ss].w, acc[gp][1], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- in-block sequential error feedback of block [i1, i1 + cnt), chain wave cw = rows 4cw .. 4cw+3 --------------------
    auto chain = [&](int64_t i1, int cnt, const float *Ft, int gp) {
        float acc[4][2], wv[4][2], et[4][2];
        constexpr bool UPD = MODE == 1 || MODE == 3;
        float sc[4][2] = {}, zr[4][2] = {};                   // mode 3: quantiser of (row, this lane's column)
        float hdv[2] = {1.f, 1.f};                            // mode 2: H[i][i] of this lane's two columns
        if constexpr (MODE == 2) {
            hdv[0] = lane < cnt ? A.hd[i1 + lane] : 1.f;
            hdv[1] = lane + 64 < cnt ? A.hd[i1 + lane + 64] : 1.f;
        }
        // the value a column takes given its accumulated feedback
        auto code3 = [&](float x, float scale, float zero) -> float {
            const float t = __fdiv_rn(x, scale);
            return A.qfn_c ? rintf(fminf(fmaxf(t + zero, 0.f), A.maxq))              // quant.py:17-21
                           : fminf(fmaxf(rintf(t) + zero, 0.f), A.maxq);              // quant.py:6-8
        };
        auto decide = [&](float w, float a, float g, float hd) -> float {
            if constexpr (MODE == 2) return rintf(w - __fdiv_rn(g + a, hd));          // torch.round(wr - Hs / H[i,i])
            else if constexpr (MODE == 3) return __fmul_rn(g, code3(w + a, g, hd) - hd);   // g = scale, hd = zero
            else return round_col(w, a, g, A.maxq);
        };
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = r0 + 16 * gp + 4 * cw + rr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                const bool ok = (c < cnt) && (row < A.m);
                acc[rr][h] = (c < cnt) ? Ft[(16 * gp + 4 * cw + rr) * BS + c] : 0.f;
                wv[rr][h] = ok ? A.W[row * d + i1 + c] : 0.f;
                et[rr][h] = (ok && A.eta) ? A.eta[row * d + i1 + c] : (MODE == 2 ? 0.f : 0.5f);
            }
        }
        if constexpr (MODE == 3) {
            const int gs = A.groupsize;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int64_t row = r0 + 16 * gp + 4 * cw + rr;
                if (gs <= 0) {
                    const float s1 = row < A.m ? A.gscale[row] : 1.f, z1 = row < A.m ? A.gzero[row] : 0.f;
                    sc[rr][0] = sc[rr][1] = s1;
                    zr[rr][0] = zr[rr][1] = z1;
                    continue;
                }
                float lo[2], hi[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float x = (lane + 64 * h < cnt) ? wv[rr][h] + acc[rr][h] : 0.f;     // 0 is neutral: min(.., 0), max(.., 0)
                    lo[h] = fminf(x, 0.f);
                    hi[h] = fmaxf(x, 0.f);
                    for (int off = 1; off < 64 && off < gs; off <<= 1) {
                        lo[h] = fminf(lo[h], __shfl_xor(lo[h], off, 64));
                        hi[h] = fmaxf(hi[h], __shfl_xor(hi[h], off, 64));
                    }
                }
                if (gs == 128) {
                    lo[0] = lo[1] = fminf(lo[0], lo[1]);
                    hi[0] = hi[1] = fmaxf(hi[0], hi[1]);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float xmin = lo[h], xmax = hi[h];
                    if (A.sym) {                                                   // quant.py:82-86
                        xmax = fmaxf(fabsf(xmin), xmax);
                        if (xmin < 0.f) xmin = -xmax;
                    }
                    if (xmin == 0.f && xmax == 0.f) { xmin = -1.f; xmax = 1.f; }     // :87-89
                    const float scale = __fdiv_rn(xmax - xmin, A.maxq);              // :91
                    const float zero = A.sym ? (A.maxq + 1.f) * 0.5f : rintf(__fdiv_rn(-xmin, scale));   // :92-95
                    sc[rr][h] = scale;
                    zr[rr][h] = zero;
                    const int c = lane + 64 * h;
                    if (c < cnt && row < A.m && (c % gs) == gs - 1) {            // one lane per group: the group's first ORIGINAL column
                        const int64_t grp = (d - 1 - (i1 + c)) / gs;              // columns arrive reversed (quipamd_gptq_round)
                        A.gscale[row * (d / gs) + grp] = scale;
                        A.gzero[row * (d / gs) + grp] = zero;
                    }
                }
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int h = 0; h < 2; ++h) et[rr][h] = sc[rr][h];
        }
        // columns 64..cnt-1 live in register half 1
        for (int i = cnt - 1; i >= 64; --i) {
            const int ln = __builtin_amdgcn_readfirstlane(i - 64);
            const float l0 = Ldiag[i * LDS_LD + lane], l1 = Ldiag[i * LDS_LD + 64 + lane];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float q = decide(wv[rr][1], acc[rr][1], et[rr][1], MODE == 3 ? zr[rr][1] : hdv[1]);
                const float er = (UPD ? wv[rr][1] + acc[rr][1] : wv[rr][1]) - q;
                const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, er), ln));
                acc[rr][0] = fmaf(e, l0, acc[rr][0]);
                acc[rr][1] = fmaf(e, l1, acc[rr][1]);
            }
        }
        const int top0 = cnt < 64 ? cnt : 64;
        for (int i = top0 - 1; i >= 0; --i) {
            const int ln = __builtin_amdgcn_readfirstlane(i);
            const float l0 = Ldiag[i * LDS_LD + lane];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const float q = decide(wv[rr][0], acc[rr][0], et[rr][0], MODE == 3 ? zr[rr][0] : hdv[0]);
                const float er = (UPD ? wv[rr][0] + acc[rr][0] : wv[rr][0]) - q;
                const float e = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, er), ln));
                acc[rr][0] = fmaf(e, l0, acc[rr][0]);
            }
        }
        // every column is final now (later steps only added err * 0): emit codes and errors
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int64_t row = r0 + 16 * gp + 4 * cw + rr;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                if (c < cnt && row < A.m) {
                    const float q = decide(wv[rr][h], acc[rr][h], et[rr][h], MODE == 3 ? zr[rr][h] : hdv[h]);
                    if constexpr (MODE == 2) A.Wout[row * d + i1 + c] = q;
                    else if constexpr (MODE == 3) {
                        A.Wout[row * d + i1 + c] = q;
                        if (A.codes) A.codes[row * d + i1 + c] = (uint8_t)code3(wv[rr][h] + acc[rr][h], sc[rr][h], zr[rr][h]);
                    } else A.codes[row * d + i1 + c] = (uint8_t)q;
                    A.E[row * d + i1 + c] = (UPD ? wv[rr][h] + acc[rr][h] : wv[rr][h]) - q;
                }
            }
        }
    };

    // ---- prologue: first block has no far field -------------------------------------------------------------------------------
    {
        const int64_t i1 = d - BS > 0 ? d - BS : 0;
        if (is_chain) stage_diag(i1, (int)(d - i1));
        else for (int idx = (int)threadIdx.x - 256; idx < ROWS * BS; idx += 256) Ftile[idx] = 0.f;
    }
    __syncthreads();
    for (int64_t i2 = d; i2 > 0; i2 -= BS) {
        const int64_t i1 = i2 - BS > 0 ? i2 - BS : 0;
        const int cnt = (int)(i2 - i1);
        const bool has_next = i1 > 0;
        const int64_t n1 = i1 - BS > 0 ? i1 - BS : 0;              // next block [n1, i1)
        const int ncnt = (int)(i1 - n1);
        f32x4_t acc[RG][2];
#pragma unroll
        for (int gp = 0; gp < RG; ++gp) acc[gp][0] = acc[gp][1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // phase X: chain(k)  ||  far(k+1) over the columns that are already final (>= i2)
        if (is_chain) {
#pragma unroll 1
            for (int gp = 0; gp < RG; ++gp) chain(i1, cnt, Ftile, gp);             // one group of 16 rows after the other: the registers of ONE
        }
        else if (has_next) far_accumulate(acc, i2, d, n1, ncnt);
        __syncthreads();                                           // block k's errors are written; Ldiag and Ftile are free
        // phase Y: stage the next diagonal block  ||  add block k's own columns to far(k+1) and publish it
        if (has_next) {
            if (is_chain) stage_diag(n1, ncnt);
            else {
                far_accumulate(acc, i1, i2, n1, ncnt);
                const int t0 = cw, t1 = cw + 4, nt = ncnt / 16;
                // D layout: col = lane & 15, row = 4*(lane>>4) + reg
#pragma unroll
                for (int gp = 0; gp < RG; ++gp)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (t0 < nt) Ftile[(16 * gp + 4 * kq + reg) * BS + t0 * 16 + fr] = acc[gp][0][reg];
                        if (t1 < nt) Ftile[(16 * gp + 4 * kq + reg) * BS + t1 * 16 + fr] = acc[gp][1][reg];
                    }
            }
        }
        __syncthreads();
    }
}

// LT[c][j] = C[j][c] * (1 / C[c][c]) for j > c, else 0   (vector_balance.py:172-173)
__global__ __launch_bounds__(256) void unit_lower_t_kernel(const float *__restrict__ C, float *__restrict__ LT, int64_t d)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int64_t j0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    for (int jj = ty; jj < 32; jj += 8) {
        const int64_t j = j0 + jj, c = c0 + tx;
        tile[jj][tx] = (j < d && c < d) ? C[j * d + c] : 0.f;     // read rows of C, lanes along c
    }
    __syncthreads();
    for (int cc = ty; cc < 32; cc += 8) {
        const int64_t c = c0 + cc, j = j0 + tx;
        if (c < d && j < d) {
            float v = 0.f;
            if (j > c) v = __fmul_rn(tile[tx][cc], __fdiv_rn(1.0f, C[c * d + c]));
            LT[c * d + j] = v;                                    // write rows of LT, lanes along j
        }
    }
}

}   // namespace

struct QuantSpec { float *scale, *zero; int groupsize, sym, qfn_c; };
static int g_ldlq_rg = 0;        // tests / A-B runs: 1 or 2 row groups per workgroup forced, 0 = by the row count
extern "C" void quipamd_ldlq_config(int row_groups) { g_ldlq_rg = row_groups == 1 || row_groups == 2 ? row_groups : 0; }

template <int MODE>
static int launch_ldlq(const float *Wgrid, const float *LT, const float *eta, int bits, uint8_t *codes, float *err_ws, int64_t m,
                       int64_t d, void *stream, const char *who, const float *hd = nullptr, float *wout = nullptr,
                       const QuantSpec *quant = nullptr)
{
    if (m == 0 || d == 0) return QUIPAMD_OK;
    QA_REQUIRE(Wgrid && LT && (codes || wout) && err_ws, QUIPAMD_ERR_ARG, "%s: null pointer", who);
    QA_REQUIRE(bits >= 1 && bits <= 8, QUIPAMD_ERR_ARG, "%s: bits out of range", who);
    QA_REQUIRE(d % 16 == 0, QUIPAMD_ERR_SHAPE, "%s: needs d %% 16 == 0 (d=%lld)", who, (long long)d);
    LdlqArgs A;
    A.W = Wgrid; A.LT = LT; A.eta = eta; A.codes = codes; A.E = err_ws; A.m = m; A.d = d;
    A.maxq = (float)((1 << bits) - 1);
    A.hd = hd; A.Wout = wout;
    A.gscale = A.gzero = nullptr; A.groupsize = 0; A.sym = 0; A.qfn_c = 0;
    if (quant) { A.gscale = quant->scale; A.gzero = quant->zero; A.groupsize = quant->groupsize; A.sym = quant->sym; A.qfn_c = quant->qfn_c; }
    // two row groups per workgroup from 8192 rows on (>= 256 workgroups of 32 rows): the LT slabs are then shared by twice the MFMAs
    // (a 32-row workgroup takes ~1.6 x as long as a 16-row one and one workgroup fits a CU: fewer, longer rounds of 256 must pay --
    //  8192 rows: 1 round instead of 2; 28672: 4 instead of 7; 11008: 2 instead of 3 does not)
    const int64_t rounds1 = ((m + 15) / 16 + 255) / 256, rounds2 = ((m + 31) / 32 + 255) / 256;
    const bool rg2 = (g_ldlq_rg == 2) || (g_ldlq_rg == 0 && m >= 8192 && 13 * rounds2 < 8 * rounds1);
    const size_t lds1 = (size_t)(BS * LDS_LD + 16 * BS) * sizeof(float) + 4 * 3 * SLAB, lds2 = (size_t)(BS * LDS_LD + 32 * BS) * sizeof(float) + 4 * 4 * SLAB;
    static QaPerDevice attr_set_dev;                                  // per instantiation
    const int attr_set_d = attr_set_dev.dev();
    if ((attr_set_d < 0 || !attr_set_dev.done[attr_set_d])) {
        if (hipFuncSetAttribute((const void *)ldlq_kernel<MODE, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess ||
            hipFuncSetAttribute((const void *)ldlq_kernel<MODE, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "%s: cannot raise dynamic LDS to %zu", who, lds2);
        if (attr_set_d >= 0) attr_set_dev.done[attr_set_d] = true;
    }
    if (rg2) ldlq_kernel<MODE, 2><<<(unsigned)((m + 31) / 32), 512, lds2, (hipStream_t)stream>>>(A);
    else ldlq_kernel<MODE, 1><<<(unsigned)((m + 15) / 16), 512, lds1, (hipStream_t)stream>>>(A);
    QA_LAUNCH_CHECK(who);
    return QUIPAMD_OK;
}

extern "C" int quipamd_gptq_round(const float *Wgrid_rev, const float *FT, int bits, uint8_t *codes_rev, float *err_ws, int64_t m,
                                  int64_t d, void *stream)
{
    return launch_ldlq<1>(Wgrid_rev, FT, nullptr, bits, codes_rev, err_ws, m, d, stream, "gptq_round");
}

extern "C" int quipamd_gptq_round_groups(const float *W_rev, const float *FT, int bits, int groupsize, int sym, int qfn_c, float *scale,
                                         float *zero, float *Q_rev, uint8_t *codes_rev, float *err_ws, int64_t m, int64_t d, void *stream)
{
    QA_REQUIRE(m == 0 || d == 0 || (scale && zero && Q_rev), QUIPAMD_ERR_ARG, "gptq_round_groups: null pointer");
    QA_REQUIRE(groupsize <= 0 || ((groupsize == 16 || groupsize == 32 || groupsize == 64 || groupsize == 128) && d % groupsize == 0),
               QUIPAMD_ERR_SHAPE, "gptq_round_groups: groupsize must be 16, 32, 64 or 128 and divide d (groupsize=%d, d=%lld)", groupsize,
               (long long)d);
    QuantSpec q{scale, zero, groupsize, sym != 0, qfn_c != 0};
    return launch_ldlq<3>(W_rev, FT, nullptr, bits, codes_rev, err_ws, m, d, stream, "gptq_round_groups", nullptr, Q_rev, &q);
}

extern "C" int quipamd_ldlq_greedy_pass(const float *wr, const float *sH, const float *negH_upper, const float *hdiag, float *wr_out,
                                        float *eps, int64_t m, int64_t d, void *stream)
{
    QA_REQUIRE(m == 0 || d == 0 || (sH && hdiag && wr_out), QUIPAMD_ERR_ARG, "ldlq_greedy_pass: null pointer");
    return launch_ldlq<2>(wr, negH_upper, sH, 1, nullptr, eps, m, d, stream, "ldlq_greedy_pass", hdiag, wr_out);
}

extern "C" int quipamd_ldlq_round(const float *Wgrid, const float *LT, const float *eta, int bits, uint8_t *codes,
                                  float *err_ws, int64_t m, int64_t d, void *stream)
{
    return launch_ldlq<0>(Wgrid, LT, eta, bits, codes, err_ws, m, d, stream, "ldlq_round");
}

extern "C" int quipamd_unit_lower_t(const float *C, float *LT, int64_t d, void *stream)
{
    QA_REQUIRE(C && LT, QUIPAMD_ERR_ARG, "unit_lower_t: null pointer");
    if (d == 0) return QUIPAMD_OK;
    dim3 grid(qa_div_up(d, 32), qa_div_up(d, 32));
    unit_lower_t_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(C, LT, d);
    QA_LAUNCH_CHECK("quipamd_unit_lower_t");
    return QUIPAMD_OK;
}
