// gptq_qfnb.hip -- OPTQ / GPTQ with the qfn-b quantiser (`--quant gptq --incoh_processing`: gptq.py:56-93 with quant.py:148-151).
//
// Quantizer.quantize recomputes ONE scale from ALL rows of every column it is handed (scale = 2.4 sqrt(mean(w^2)) + 1e-16,
// quant.py:158-160), and GPTQ hands it the columns one at a time, each already carrying the feedback of every earlier column: the sweep
// is d grid-wide reductions in series.  K4 (ldlq.hip) gives a workgroup 16 rows for the whole sweep and cannot hold that; rounds 1-3a
// ran this configuration as the reference's column walk in torch (~10 launches per column, ~0.4 s per 4096 x 4096 Linear).  Here:
//
//   gptqb_chain_kernel   one launch per 128-column lazy block, G = ceil(m / R) co-resident workgroups of R rows.  The block of W and the
//                        128 x 128 tile of the feedback matrix live in LDS; per column: partial sum of squares of the workgroup's rows ->
//                        a data-tagged 8-byte granule per workgroup in global memory (tag = column number: no counter, no fence, no
//                        read-modify-write -- the form MI355X_MICROARCH.md measures fastest for an all-gather of small values) -> every
//                        workgroup polls the G granules, sums them in a fixed order (deterministic), forms the scale, quantises its
//                        rows, and feeds the residual to the block's remaining columns.  ~3 us per column.
//   gptqb_far_kernel     W[:, before the block] += R_block @ FT[.., block]  on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32
//                        fma chains), all CUs.
//
// Coordinates as in quipamd_gptq_round: columns REVERSED (c' = d-1-c), feedback FT[j'][c'] = -Hinv[c][j] / Hinv[c][c] strictly upper,
// raw residual r = w - q, so the sweep runs c' = d-1 .. 0 and updates j' < c'.  W, Q and the residuals are held TRANSPOSED ([d][m]): a
// column of the workgroup's rows is one contiguous run.
#include "common.h"
#include "fpass.h"
#include "wavered.h"

namespace {

constexpr int GB_NB = 128, GB_T = 256;

struct GbArgs {
    float *WT;                    // [d][m] reversed columns, updated in place
    const float *FT;              // [d][d]
    float *QT;                    // [d][m] out: dequantised weights
    float *ET;                    // [128][m] residuals of the block
    float *colscale;              // [d] out (reversed)
    unsigned long long *gran;     // [2][G] granules {tag : 32, partial : 32}
    int *abort_flag;              // device int: set when a poll ran out of patience (a workgroup of the grid never became resident); every
                                  // workgroup then leaves, later launches of the sweep return at once, the host reports QUIPAMD_ERR_LAUNCH
    long long spin_limit;         // polls of one granule before giving up
    int64_t m, d;
    int b0, nb, G;
    float maxq;
    int *claim;                   // XL: this launch's participant counter (zeroed with the granules)
    int absent;                   // XL, test hook: that many participants never show up
    int xcc, first;               // XL: the XCD the sweep runs on; first: the sweep's first launch (its first exchange is the roll call)
};

__device__ __forceinline__ float gb_quant(float w, float s, float maxq)
{
    // quantize_qfnb, quant.py:10-15, the operations in the reference's order (fp32)
    float v = __fdiv_rn(w, s);
    v = (v + 1.0f) * 0.5f;
    v = v * maxq;
    const float q = fminf(fmaxf(rintf(v), 0.0f), maxq);              // torch.round: half to even
    float t = __fdiv_rn(q, maxq);
    t = t * 2.0f - 1.0f;
    return t * s;
}

template <int R>
__global__ __launch_bounds__(GB_T) void gptqb_chain_kernel(GbArgs A)
{
    constexpr int NCG = GB_T / R;                                     // column groups: thread (r, cg) owns the columns == cg (mod NCG)
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float *W1 = gsm;                                                  // [128][R]
    float *F1 = gsm + GB_NB * R;                                      // [128][128]: F1[jl][cl] = FT[b0 + jl][b0 + cl]
    float *E = F1 + GB_NB * GB_NB;                                    // [R]
    float *red = E + R;                                               // [R] squares, then [0] = the column's sum
    float *gaveup = red + R;                                          // [1] != 0: the sweep was abandoned (bounded poll below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = tid % R, cg = tid / R;
    const int wg = blockIdx.x, G = A.G, nb = A.nb, b0 = A.b0;
    const int64_t row = (int64_t)wg * R + r;
    const bool live = row < A.m;
    if (tid == 0)                                                     // an earlier block of this sweep gave up: leave (read once per workgroup,
        gaveup[0] = __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.f : 0.f;   // so that the exit is uniform)

    const int64_t rowc = live ? row : A.m - 1;                        // (clamped address + select: a guarded load is a branch and a round trip)
    for (int cl = cg; cl < nb; cl += NCG) {
        const float v = A.WT[(int64_t)(b0 + cl) * A.m + rowc];
        W1[cl * R + r] = live ? v : 0.f;
    }
    for (int i = tid; i < nb * nb; i += GB_T) {
        const int jl = i / nb, cl = i - jl * nb;
        F1[jl * GB_NB + cl] = A.FT[(int64_t)(b0 + jl) * A.d + b0 + cl];
    }
    __syncthreads();
    if (gaveup[0] != 0.f) return;
    const float fm = (float)A.m;
    for (int cl = nb - 1; cl >= 0; --cl) {
        const int cp = b0 + cl;                                       // reversed column index
        const bool owner = cg == cl % NCG;
        float w = 0.f;
        if (owner) {
            w = W1[cl * R + r];
            red[r] = w * w;
        }
        __syncthreads();
        if (wave == 0) {
            float p = 0.f;
            for (int i = lane; i < R; i += 64) p += red[i];
            p = fg_wave_sum(p);
            const unsigned tag = (unsigned)(A.d - cp);               // 1 .. d, unique per column of the sweep
            unsigned long long *gr = A.gran + (size_t)(tag & 1) * G;
            if (lane == 0)
                __hip_atomic_store(gr + wg, ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, p), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            // gather: lane l polls granules l, l + 64, ...; the values are summed in a fixed order
            // The poll is BOUNDED: co-residency of the G workgroups is inferred from the occupancy query, not guaranteed (another stream's
            // kernel, an RCCL collective, a masked device can keep a workgroup off the chip).  A lane that has polled one granule
            // spin_limit times -- or sees the sweep's abort flag -- gives up; the wave raises the flag and the workgroup leaves.
            float s = 0.f;
            bool bad = false;
            for (int i = lane; i < G && !bad; i += 64) {
                unsigned long long v;
                long long spins = 0;
                for (;;) {
                    v = __hip_atomic_load(gr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(v >> 32) == tag) break;
                    if (++spins >= A.spin_limit ||
                        ((spins & 255) == 0 && __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        bad = true;
                        break;
                    }
                }
                s += __builtin_bit_cast(float, (unsigned)v);
            }
            const bool dead = __any(bad);
            if (dead && lane == 0) __hip_atomic_store(A.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s = fg_wave_sum(s);
            if (lane == 0) {
                red[0] = s;
                if (dead) gaveup[0] = 1.f;
            }
        }
        __syncthreads();
        if (gaveup[0] != 0.f) return;                                     // (uniform: every thread reads the same word behind the barrier)
        const float S = red[0];
        const float scale = 2.4f * sqrtf(__fdiv_rn(S, fm)) + 1e-16f;     // quant.py:159
        if (owner) {
            const float q = gb_quant(w, scale, A.maxq);
            const float res = live ? w - q : 0.f;                       // rows past m stay zero: they are part of every column's sum
            if (live) A.QT[(int64_t)cp * A.m + row] = q;
            W1[cl * R + r] = res;
            E[r] = res;
            if (wg == 0 && r == 0) A.colscale[cp] = scale;
        }
        __syncthreads();
        const float e = E[r];
        for (int jl = cg; jl < cl; jl += NCG) W1[jl * R + r] = fmaf(e, F1[jl * GB_NB + cl], W1[jl * R + r]);
        // (the next column's owner reads what this very thread wrote; E and red are rewritten behind the next two barriers)
    }
    __syncthreads();
    if (live)
        for (int cl = cg; cl < nb; cl += NCG) A.ET[(int64_t)cl * A.m + row] = W1[cl * R + r];
}

// Round 6: the PIPELINED chain.  The form above runs a column as [owner's squares -> barrier -> wave 0: reduce, publish, poll -> barrier ->
// quantise -> barrier -> every thread feeds the residual to ALL remaining columns], 3.2-4.0 us per column of which the granule hand-off is
// ~1.5-2 (scripts/xcdsync_lab.hip: 1.6-2.0 us per all-gather of 32-64 granules at agent scope): everything else sat in series with it.
// Here a CHAIN WAVE owns 64 rows for the whole block: lane = row, the current column's 64 values live in its registers, sums run on the DPP
// network, and all it does between receiving a column's sum and publishing the next column's partial is: scale, quantise, residual, ONE fma
// for the next column (its own residual times F[c - 1][c]), 64 squares, a wave sum.  Three HELPER waves per chain wave feed residual c to
// the columns < c - 1 while the chain wave's next granule travels; one barrier per column orders the two (the chain reads column c - 1
// only after the helpers have applied every residual > c to it, the helpers read residual c only after the chain has written it).  W1[c] is
// column c until the chain has quantised it and its residual afterwards -- the block of residuals the far-field kernel reads is W1 at the end.
//
// XL -- the exchange confined to ONE XCD.  A plain store is written through but STAYS in the XCD's L2, and an sc1 load bypasses the reader's
// L1 and is served by that L2 (MI355X_MICROARCH.md, "stores of each flavour"): between workgroups of one XCD an all-gather of 16-64
// granules costs 0.55 us instead of 1.4-2.4 (profiles/r06o_xcdsync_lab.txt; across XCDs the same pair of instructions reads stale lines
// forever -- the lab's control).  So up to 16384 rows the sweep runs on the 32 CUs of XCD `A.xcc`: a full grid of one-per-CU workgroups
// is launched, every workgroup reads HW_REG_XCC_ID, the ones on that XCD claim a participant slot from a counter, everybody else leaves.
// Nothing is assumed about which block lands where (HIP promises nothing; block b on XCD b % 8 is what is observed); what is needed is
// that ceil(m / rows) workgroups of that XCD become resident.  The first column's exchange is the roll call: if it times out nothing
// has been written yet, the abort word becomes 2 and the caller repeats the sweep with the cross-XCD form (ops.gptq_round_qfnb does).
// More rows per workgroup (the one-XCD form beyond 4096 rows): NBK = 64 columns per lazy block and HP = 1 helper wave per chain wave keep
// 256 / 512 rows x 64 columns + the 64 x 64 feedback tile within a workgroup's LDS and its 1024 threads.
template <int NC, bool XL, int NBK = GB_NB, int HP = 3>               // chain waves per workgroup (rows = 64 NC)
__global__ __launch_bounds__(64 * NC * (1 + HP)) void gptqb_chainp_kernel(GbArgs A)
{
    constexpr int R = 64 * NC, NTH = 64 * NC * (1 + HP);
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    float *W1 = gsm;                                                  // [128][R]
    float *F1 = gsm + NBK * R;                                        // [NBK][NBK]: F1[jl][cl] = FT[b0 + jl][b0 + cl]
    float *gaveup = F1 + NBK * NBK;                                   // [0] != 0: the sweep was abandoned (bounded poll below); [1]: the slot
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NP = A.G, nb = A.nb, b0 = A.b0;                         // participants = workgroups that hold rows
    if (tid == 0) {
        int slot = (int)blockIdx.x;
        if constexpr (XL) {
            const int xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;       // HW_REG_XCC_ID
            slot = xcc == A.xcc ? atomicAdd(A.claim, 1) : -1;
        }
        gaveup[1] = __builtin_bit_cast(float, slot);
        gaveup[0] = __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1.f : 0.f;
    }
    __syncthreads();
    const int wg = __builtin_bit_cast(int, gaveup[1]);
    if (wg < 0 || wg >= NP - (XL ? A.absent : 0) || gaveup[0] != 0.f) return;      // (uniform; `absent`: the test hook's missing workgroups)
    const int chain = wave < NC ? wave : (wave - NC) / HP;             // the 64-row slice this wave works on
    const int rl = chain * 64 + lane;
    const int64_t row = (int64_t)wg * R + rl;
    const bool live = row < A.m;
    const int64_t rowc = live ? row : A.m - 1;                        // (clamped address + select: a guarded load is a branch and a round trip)
    for (int i = tid; i < nb * R; i += NTH) {
        const int cl = i / R, r = i - cl * R;
        const int64_t rr = (int64_t)wg * R + r;
        const float v = A.WT[(int64_t)(b0 + cl) * A.m + (rr < A.m ? rr : A.m - 1)];
        W1[cl * R + r] = rr < A.m ? v : 0.f;
    }
    for (int i = tid; i < nb * nb; i += NTH) {
        const int jl = i / nb, cl = i - jl * nb;
        F1[jl * NBK + cl] = A.FT[(int64_t)(b0 + jl) * A.d + b0 + cl];
    }
    __syncthreads();
    (void)rowc;
    const float fm = (float)A.m;
    const int NG = NP * NC;                                           // granules per column: one per chain wave
    if (wave < NC) {
        float w = W1[(nb - 1) * R + rl];
        for (int cl = nb - 1; cl >= 0; --cl) {
            const int cp = b0 + cl;                                   // reversed column index
            const float p = wave_reduce<false>(w * w);
            const unsigned tag = (unsigned)(A.d - cp);               // 1 .. d, unique per column of the sweep
            unsigned long long *gr = A.gran + (size_t)(tag & 1) * NG;
            const unsigned long long mine = ((unsigned long long)tag << 32) | (unsigned long long)__builtin_bit_cast(unsigned, p);
            if (lane == 0) {
                if constexpr (XL) __hip_atomic_store(gr + wg * NC + wave, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // plain: stays in the XCD's L2
                else __hip_atomic_store(gr + wg * NC + wave, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // gather: lane l polls granules l, l + 64, ...; bounded (co-residency is inferred, not guaranteed)
            float sgr = 0.f;
            bool bad = false;
            for (int i = lane; i < NG && !bad; i += 64) {
                unsigned long long v;
                long long spins = 0;
                for (;;) {
                    v = __hip_atomic_load(gr + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // sc1: past the L1, served by the L2
                    if ((unsigned)(v >> 32) == tag) break;
                    if (++spins >= A.spin_limit ||
                        ((spins & 255) == 0 && __hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                        bad = true;
                        break;
                    }
                }
                sgr += __builtin_bit_cast(float, (unsigned)v);
            }
            const bool dead = __any(bad);
            if (dead && lane == 0) {
                // the roll call (first column of the sweep's first launch, nothing written yet) failing is code 2: repeat with the other form
                int expect = 0;
                __hip_atomic_compare_exchange_strong(A.abort_flag, &expect, (XL && A.first && cl == nb - 1) ? 2 : 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT);
                gaveup[0] = 1.f;
            }
            const float S = wave_reduce<false>(sgr);                  // (every lane active again here; fixed order: deterministic)
            const float scale = 2.4f * sqrtf(__fdiv_rn(S, fm)) + 1e-16f;     // quant.py:159
            const float q = gb_quant(w, scale, A.maxq);
            const float res = live ? w - q : 0.f;                     // rows past m stay zero: they are part of every column's sum
            if (!dead) W1[cl * R + rl] = res;
            __syncthreads();                                          // residual c is out; every residual > c has been applied to the columns < c
            if (gaveup[0] != 0.f) return;
            if (cl > 0) w = fmaf(res, F1[(cl - 1) * NBK + cl], W1[(cl - 1) * R + rl]);
            if (live) A.QT[(int64_t)cp * A.m + row] = q;             // (off the chain: behind the next column's value)
            if (wg == 0 && tid == 0) A.colscale[cp] = scale;
        }
    } else {
        const int sub = (wave - NC) % HP;
        for (int cl = nb - 1; cl >= 0; --cl) {
            __syncthreads();
            if (gaveup[0] != 0.f) return;
            const float e = W1[cl * R + rl];
            for (int jl = cl - 2 - sub; jl >= 0; jl -= HP) W1[jl * R + rl] = fmaf(e, F1[jl * NBK + cl], W1[jl * R + rl]);   // cl - 2 first: the chain needs it next
        }
    }
    __syncthreads();
    for (int i = tid; i < nb * R; i += NTH) {
        const int cl = i / R, r = i - cl * R;
        const int64_t rr = (int64_t)wg * R + r;
        if (rr < A.m) A.ET[(int64_t)cl * A.m + rr] = W1[cl * R + r];
    }
}

// WT[j'][rows] += sum_cl FT[j'][b0 + cl] ET[cl][rows] for j' < b0.  Workgroup = 64 j' x 64 rows, wave = 16 j' x 64 rows (4 tiles).
__global__ __launch_bounds__(256) void gptqb_far_kernel(GbArgs A)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int jc = lane & 15, g = lane >> 4;
    const int64_t j0 = (int64_t)blockIdx.x * 64 + 16 * wave, r0 = (int64_t)blockIdx.y * 64;
    if (j0 >= A.b0) return;
    if (__hip_atomic_load(A.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // an abandoned sweep left no residuals: W stays as it is
    const int64_t ja = j0 + jc < A.b0 ? j0 + jc : A.b0 - 1;          // A operand row (clamped: its results are not stored)
    const float *Fa = A.FT + ja * A.d + A.b0;
    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int nb = A.nb;
    for (int kk = 0; kk < nb; kk += 16) {
        float a[4], b[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = kk + 4 * g + s;                              // this lane group's k of instruction s
            const int kc = k < nb ? k : nb - 1;                        // clamped addresses, selects afterwards: every load in flight at once
            const float av = Fa[kc];
            a[s] = k < nb ? av : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int64_t rr = r0 + 16 * t + jc, rc = rr < A.m ? rr : A.m - 1;
                b[s][t] = A.ET[(int64_t)kc * A.m + rc];                // (rows past m: garbage columns of D, not stored)
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s][t], acc[t], 0, 0, 0);
    }
    // D: col = rows (jc), row = j' = 4g + reg
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t jj = j0 + 4 * g + reg, rr = r0 + 16 * t + jc;
            if (jj < A.b0 && rr < A.m) A.WT[jj * A.m + rr] += acc[t][reg];
        }
}

// test hooks (quipamd_gptq_qfnb_debug): launch that many workgroups too few -- the rest wait for granules nobody writes -- and the poll budget
int g_gb_debug_short_grid = 0;
constexpr long long GB_SPIN_LIMIT = 1ll << 22;                        // ~0.5 us per poll: a few seconds; a healthy wait is a few polls
long long g_gb_spin_limit = GB_SPIN_LIMIT;
int g_gb_force_rows = 0;                                              // lab: rows per workgroup (16 / 32 / 64 / 128) instead of the heuristic's

template <int R> int gb_chain(const GbArgs &A, hipStream_t s)
{
    const size_t lds = (size_t)(GB_NB * R + GB_NB * GB_NB + 2 * R + 4) * sizeof(float);
    auto kern = gptqb_chain_kernel<R>;
    static QaPerDevice attr;
    const int dv = attr.dev();
    if (lds > 64 * 1024 && (dv < 0 || !attr.done[dv])) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "gptq_round_qfnb: cannot raise dynamic LDS to %zu", lds);
        if (dv >= 0) attr.done[dv] = true;
    }
    kern<<<(unsigned)(A.G - g_gb_debug_short_grid > 0 ? A.G - g_gb_debug_short_grid : 1), GB_T, lds, s>>>(A);
    return QUIPAMD_OK;
}

template <int NC, bool XL, int NBK = GB_NB, int HP = 3> int gb_chainp(const GbArgs &A, int grid, hipStream_t s)
{
    constexpr size_t lds = (size_t)(NBK * 64 * NC + NBK * NBK + 4) * sizeof(float);
    auto kern = gptqb_chainp_kernel<NC, XL, NBK, HP>;
    static QaPerDevice attr;
    const int dv = attr.dev();
    if (dv < 0 || !attr.done[dv]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "gptq_round_qfnb: cannot raise dynamic LDS to %zu", lds);
        if (dv >= 0) attr.done[dv] = true;
    }
    if (!XL) grid -= g_gb_debug_short_grid;                             // (XL: GbArgs.absent -- which blocks land on the XCD is not the host's to say)
    kern<<<(unsigned)(grid > 0 ? grid : 1), 64 * NC * (1 + HP), lds, s>>>(A);
    return QUIPAMD_OK;
}
template <int NC, bool XL, int NBK = GB_NB, int HP = 3> bool gb_chainp_fits(int64_t blocks, int ncu)
{
    constexpr size_t lds = (size_t)(NBK * 64 * NC + NBK * NBK + 4) * sizeof(float);
    auto kern = gptqb_chainp_kernel<NC, XL, NBK, HP>;
    int per_cu = 0;
    return hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
           hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, 64 * NC * (1 + HP), lds) == hipSuccess && blocks <= (int64_t)per_cu * ncu;
}

}   // namespace

extern "C" void quipamd_gptq_qfnb_debug(int short_grid, int64_t spin_limit, int force_rows)
{
    g_gb_debug_short_grid = short_grid > 0 ? short_grid : 0;
    g_gb_spin_limit = spin_limit > 0 ? spin_limit : GB_SPIN_LIMIT;
    // 0: the heuristic -- the pipelined chain, confined to one XCD up to 4096 rows (gptqb_chainp_kernel); 1: the pipelined chain across the XCDs
    // (what ops.gptq_round_qfnb repeats a sweep with when the one-XCD roll call fails: abort word 2); 2: the one-XCD form or an error;
    // 16 / 32 / 64 / 128: the barrier-per-phase chain of rounds 3-5 with that many rows per workgroup (A/B runs)
    g_gb_force_rows = (force_rows == 1 || force_rows == 2 || force_rows == 16 || force_rows == 32 || force_rows == 64 || force_rows == 128) ? force_rows : 0;
}

extern "C" int64_t quipamd_gptq_qfnb_info_offset(int64_t m, int64_t d)
{
    (void)d;
    return (int64_t)GB_NB * m * 4 + 2 * ((m + 15) / 16) * 8;          // the 64 spare bytes behind the granules
}

extern "C" int64_t quipamd_gptq_qfnb_workspace_bytes(int64_t m, int64_t d)
{
    (void)d;
    // residuals of a block + two granules per possible workgroup + 64 bytes (abort word) + one participant counter per lazy block
    return (int64_t)GB_NB * m * 4 + 2 * ((m + 15) / 16) * 8 + 64 + 4 * ((d + 63) / 64 + 1);
}

extern "C" int quipamd_gptq_round_qfnb(float *WT_rev, const float *FT, int bits, float *QT_rev, float *colscale_rev, void *workspace,
                                       int64_t m, int64_t d, void *stream)
{
    QA_REQUIRE(m >= 0 && d >= 0, QUIPAMD_ERR_SHAPE, "gptq_round_qfnb: bad shape");
    if (m == 0 || d == 0) return QUIPAMD_OK;
    QA_REQUIRE(WT_rev && FT && QT_rev && colscale_rev && workspace, QUIPAMD_ERR_ARG, "gptq_round_qfnb: null pointer");
    QA_REQUIRE(bits >= 1 && bits <= 8, QUIPAMD_ERR_ARG, "gptq_round_qfnb: bits");
    // rows per workgroup: every workgroup of a launch must be RESIDENT at once (they wait for each other): the smallest R whose grid fits
    // the device as the occupancy query sees it (256 CUs x 1-2 workgroups here; a partitioned or masked device has fewer)
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
        return qa_fail(QUIPAMD_ERR_LAUNCH, "gptq_round_qfnb: cannot query the device");
    auto fits = [&](int R, const void *kern) {
        const size_t lds = (size_t)(GB_NB * R + GB_NB * GB_NB + 2 * R + 4) * sizeof(float);
        if (lds > 64 * 1024 && hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return false;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, GB_T, lds) != hipSuccess) return false;
        return (m + R - 1) / R <= (int64_t)per_cu * ncu;
    };
    int R = 0;
    // form: 0 barrier-per-phase chain (rounds 3-5), 1 pipelined across the XCDs, 2 / 3 / 4 / 5 pipelined on one XCD with 1 / 2 / 4 / 8 chain waves
    // (64 / 128 / 256 / 512 rows) per workgroup
    int form = 0;
    if (g_gb_force_rows == 0 || g_gb_force_rows == 2) {
        // one XCD = ncu / 8 compute units, one workgroup each; the full grid of ncu one-per-CU workgroups must be resident
        if (ncu % 8 == 0 && m <= (int64_t)64 * (ncu / 8) && gb_chainp_fits<1, true>(ncu, ncu)) form = 2;
        else if (ncu % 8 == 0 && m <= (int64_t)128 * (ncu / 8) && gb_chainp_fits<2, true>(ncu, ncu)) form = 3;
        else if (ncu % 8 == 0 && m <= (int64_t)256 * (ncu / 8) && gb_chainp_fits<4, true, 64, 3>(ncu, ncu)) form = 4;   // 64-column lazy blocks from here on
        else if (ncu % 8 == 0 && m <= (int64_t)512 * (ncu / 8) && gb_chainp_fits<8, true, 64, 1>(ncu, ncu)) form = 5;
        QA_REQUIRE(form != 0 || g_gb_force_rows == 0, QUIPAMD_ERR_UNSUPPORTED, "gptq_round_qfnb: %lld rows do not fit one XCD (%d CUs / 8 x 512 rows)",
                   (long long)m, ncu);
    }
    if (form == 0 && (g_gb_force_rows == 0 || g_gb_force_rows == 1) && gb_chainp_fits<1, false>((m + 63) / 64, ncu)) form = 1;
    if (form == 1 || form == 2) R = 64;
    else if (form == 3) R = 128;
    else if (form == 4) R = 256;
    else if (form == 5) R = 512;
    else if (g_gb_force_rows == 16 && fits(16, (const void *)gptqb_chain_kernel<16>)) R = 16;
    else if (g_gb_force_rows == 32 && fits(32, (const void *)gptqb_chain_kernel<32>)) R = 32;
    else if (g_gb_force_rows == 64 && fits(64, (const void *)gptqb_chain_kernel<64>)) R = 64;
    else if (g_gb_force_rows == 128 && fits(128, (const void *)gptqb_chain_kernel<128>)) R = 128;
    // (round 4 A/B, profiles/r04d_gptq_qfnb_rows.jsonl: the per-column all-gather over G = m / R granules is the cost, and 64 rows per workgroup
    //  beat 16 / 32 from 2048 rows on -- 2048^2 3.64 -> 3.08 us per column, 4096^2 4.16 -> 3.83, 8192 x 2048 4.86 -> 4.08, 2048 x 8192 4.02 -> 3.51)
    else if (m < 1024 && fits(16, (const void *)gptqb_chain_kernel<16>)) R = 16;
    else if (m < 2048 && fits(32, (const void *)gptqb_chain_kernel<32>)) R = 32;
    else if (m <= 128 * 128 && fits(64, (const void *)gptqb_chain_kernel<64>)) R = 64;
    else if (fits(128, (const void *)gptqb_chain_kernel<128>)) R = 128;
    QA_REQUIRE(R != 0, QUIPAMD_ERR_UNSUPPORTED, "gptq_round_qfnb: %lld rows do not fit this device as one grid of co-resident workgroups (%d CUs)",
               (long long)m, ncu);
    const int64_t G = (m + R - 1) / R;
    const int64_t NBmax = form >= 4 ? 64 : GB_NB;                         // columns per lazy block
    hipStream_t s = (hipStream_t)stream;
    GbArgs A;
    A.WT = WT_rev; A.FT = FT; A.QT = QT_rev; A.colscale = colscale_rev;
    A.ET = (float *)workspace;
    A.gran = (unsigned long long *)((char *)workspace + (size_t)GB_NB * m * 4);
    A.m = m; A.d = d; A.G = (int)G; A.maxq = (float)((1 << bits) - 1);
    A.abort_flag = (int *)((char *)workspace + quipamd_gptq_qfnb_info_offset(m, d));
    A.spin_limit = g_gb_spin_limit;
    A.claim = A.abort_flag + 16;                                          // (behind the 64 spare bytes)
    A.xcc = 0; A.first = 1; A.absent = g_gb_debug_short_grid;
    // granules, the abort flag behind them and the participant counters (2 * ceil(m / 16) * 8 bytes of granule space + 64 spare + counters, all zeroed)
    if (hipMemsetAsync(A.gran, 0, (size_t)(2 * ((m + 15) / 16) * 8 + 64 + 4 * ((d + 63) / 64 + 1)), s) != hipSuccess)
        return qa_fail(QUIPAMD_ERR_LAUNCH, "gptq_round_qfnb: memset failed");
    // lazy blocks from the top of the reversed order; block edges at multiples of 128, so a remainder of d is the FIRST block (where a
    // block ends only decides when its residuals reach the columns behind it, not what they are)
    int64_t b1 = d;
    while (b1 > 0) {
        const int64_t nb = (b1 % NBmax) ? (b1 % NBmax) : NBmax;
        const int64_t b0 = b1 - nb;
        A.b0 = (int)b0; A.nb = (int)nb;
        int rc = form == 5 ? gb_chainp<8, true, 64, 1>(A, ncu, s) : form == 4 ? gb_chainp<4, true, 64, 3>(A, ncu, s)
                 : form == 3 ? gb_chainp<2, true>(A, ncu, s) : form == 2 ? gb_chainp<1, true>(A, ncu, s) : form == 1 ? gb_chainp<1, false>(A, (int)G, s)
                 : R == 128 ? gb_chain<128>(A, s) : R == 64 ? gb_chain<64>(A, s) : R == 32 ? gb_chain<32>(A, s) : gb_chain<16>(A, s);
        if (rc != QUIPAMD_OK) return rc;
        A.claim += 1;
        A.first = 0;
        if (b0 > 0) gptqb_far_kernel<<<dim3((unsigned)((b0 + 63) / 64), (unsigned)((m + 63) / 64)), 256, 0, s>>>(A);
        b1 = b0;
    }
    QA_LAUNCH_CHECK("quipamd_gptq_round_qfnb");
    return QUIPAMD_OK;
}
