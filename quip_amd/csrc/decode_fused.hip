// decode_fused.hip -- one launch per packed Linear group of a batch-1 decode step (opt.py:431-482,
// llama.py:418-471 benchmark()): everything that sits BETWEEN two dequant-GEMMs of a decoder block rides in the prologue of the
// consuming GEMM, so the only launch boundaries left are the block's true all-to-all edges (the output of a GEMM is needed whole
// by every workgroup of the next one):
//
//     t    = [relu]( U_prev^T y_prev + bias_prev + residual )          output-side operator of the PREVIOUS packed layer (optional)
//     h    = LayerNorm | RMSNorm | identity (t)                        (t is also stored once: it is the new residual stream)
//     x~_i = V_i ( h (/) s_i )                                         activation-side operator of layer i (1..3 layers sharing h)
//     y_i  = What_i x~_i                                               2-bit fused dequant-GEMM, fp32 out
//
// Round 2 ran this as 2-3 launches (operator launch(es) 3.4-6 us each + GEMM 4.9 us; the operator in the GEMM prologue,
// dqgemm_vop.hip, cost 7.8 us because the split-bf16 operator pass of small_pass.h is ~4 us of serial phases).  Here the operator
// pass is rebuilt for the decode case:
//   * ONE f16 product per factor entry (v_mfma_f32_16x16x32_f16) instead of three bf16 ones: the pass's output is rounded to 16
//     bits anyway (x~ feeds the 16-bit MFMA of the GEMM, t is the fp16 residual stream), so hi + lo operands bought nothing
//     there; f16 factors carry 2^-12 relative error per entry (orthogonal factors, |entry| <= 1), ~3e-4 per stage on the result;
//   * factor matrices never touch LDS: the host stores them in MFMA B-fragment order and every wave pulls exactly the
//     fragments of its own tiles with one 16-byte load per lane per k-step, requested at kernel start;
//   * stage 1 is computed transposed (D = z^T M0^T) so that a lane ends up with 4 consecutive b of one a: one ds_write_b64 per
//     tile puts the result into exactly the layout stage 2 reads its A fragments from (no 2-byte scatter between the stages);
//   * compile-time (p, q), operands of the NEXT phase requested before the current one runs.
// The weights of the workgroup (16 KiB - 32 KiB of packed codes) are requested first of all, so their HBM round trip hides
// under the prologue; x~ is handed to the MFMAs through LDS as [batch row][k] (only lanes of real batch rows read it: bs <= 4).
// Same STREAM weight layout, same epilogue algebra as dqgemm_vop.hip / dqgemm.hip (value = OFF + code, y = alpha (acc - c0 sum x)).
#include "common.h"
#include "dq_common.h"
#include "fpass.h"
#include "probe.h"
#include "prefetch.h"

// A/B builds (python __graft_entry__.py --variant noshf -DQA_NO_SHF): round 5's per-wave fragment loads
#ifdef QA_NO_SHF
#define QA_SHF 0
#else
#define QA_SHF 1
#endif
// A/B builds (--variant unc -DQA_UNC): the prologue's per-thread operand loads unconditional -- threads past the row's last 4-element group read
// group 0 again -- so that hipcc's vmcnt waits count them (a load behind `if (v4 < N / 4)` is not counted: the wait for the ROW, the oldest
// load, becomes a wait for every operand requested behind it; csrc/ortho_blk.hip gained 2000 clocks per launch from this)
#ifdef QA_UNC
#define QA_UNCOND 1
#else
#define QA_UNCOND 0
#endif

namespace {

constexpr int FG_MAXG = 3, FG_MAXBS = 4, FG_NW = 16;

// dequantiser of a launch: multi-exponent (DeqME2) or the uniform-offset form, behind one interface
template <bool ME, int BITS = 2> struct DeqSelME : DeqT<BITS, ActF16> {        // uniform offset: 2-bit with one tile per wave, and every 4-bit launch
    struct Consts { };
    static __device__ __forceinline__ Consts make_consts() { return Consts{}; }
    static __device__ __forceinline__ u32x4 frag(const u32x4 &w, int t, const Consts &) { return DeqT<BITS, ActF16>::frag(w, t); }
};
template <> struct DeqSelME<true, 2> : DeqME2<ActF16> {};

// lab builds (scripts/fusedlab.hip, -DFG_PROBE): s_memtime stamps of wave 0 of workgroup (FG_PROBE_WG, 0) at the phase boundaries
#ifdef FG_PROBE
__device__ unsigned long long fg_probe_buf[32];
#ifndef FG_PROBE_WG
#define FG_PROBE_WG 0
#endif
#define FG_STAMP(i)                                                                                                   \
    do {                                                                                                              \
        if (blockIdx.x == FG_PROBE_WG && blockIdx.y == 0 && threadIdx.x == 0) fg_probe_buf[i] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// per-wave stamps (lane 0 of every wave of the probed workgroup): slot i of wave w
__device__ unsigned long long fg_wprobe_buf[16 * 8];
#define FG_WSTAMP(i)                                                                                                  \
    do {                                                                                                              \
        if (blockIdx.x == FG_PROBE_WG && blockIdx.y == 0 && (threadIdx.x & 63) == 0)                                  \
            fg_wprobe_buf[(threadIdx.x >> 6) * 8 + (i)] = __builtin_amdgcn_s_memtime();                               \
    } while (0)
#elif defined(QA_PROBE)
#define FG_STAMP(i) QA_STAMP(i)            // in situ (csrc/probe.h): every wave of one workgroup, the same slots
#define FG_WSTAMP(i)
#else
#define FG_STAMP(i)
#define FG_WSTAMP(i)
#endif

struct FGroup {                           // one packed layer of the launch (blockIdx.y): 72 bytes of kernarg, fetched together
    Fop V;
    const float *colscale;
    const uint4 *qw;
    const float *scale;
    void *y;                              // [bs, m] fp32, or f16 when y_f16 (the consumer's scatter rounds it to f16 anyway)
};

struct FusedArgs {
    Fop U;
    const uint16_t *u_y, *u_bias;         // [bs, n] f16; [n] f16 (zeros when the layer has no bias)
    const uint16_t *u_res;                // [bs, ld_res] f16 (HAS_RES)
    uint16_t *t_out;                      // [bs, ld_t] f16 or null
    int64_t ld_res, ld_t;
    const uint16_t *x;                    // !HAS_U: [bs, ldx] f16
    int64_t ldx;
    const uint16_t *gamma, *beta;
    float eps, floor;                     // floor: 0 (relu) or -inf
    int bs, y_f16;
    int64_t m;
    FGroup g[FG_MAXG];
    const uint4 *pair_sig, *pair_bias, *pair_cs;   // fused_pair_kernel: per-lane tables in D-fragment order (include/quip_amd.h)
    QaPfList pf;                          // operands of a later launch, touched by QA_PF_WGS extra workgroups (csrc/prefetch.h); n = 0: none
};

// keep a kernarg pointer's scalar load where it is written: hipcc fetches kernarg fields lazily, one s_load + s_waitcnt per first
// use, and the prologue of this kernel was SEVEN serial kernarg round trips (~2000 cycles) before its first global load
template <class T> __device__ __forceinline__ void touch_s(T *p) { asm volatile("" ::"s"(p)); }

// block-wide sum over the 16 waves: one LDS round, ONE barrier; each call site owns its own red[16]
__device__ __forceinline__ float fg_block_sum(float v, float *red)
{
    const float w = fg_wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < FG_NW; i += 4) {
        const float4 r = *reinterpret_cast<const float4 *>(red + i);
        t += (r.x + r.y) + (r.z + r.w);
    }
    return t;
}

// ---- the fused launch ------------------------------------------------------------------------------------------------------------
// grid = (m / (16 RT), ngroups); 1024 threads = 16 waves = (16 / RT chunk slots) x RT row tiles; d = P Q = 256 (16 / RT) CPW.
// NORM: 0 none, 1 LayerNorm, 2 RMSNorm.  Every global operand of the prologue is requested in the first instructions of the kernel
// (EARLY: all of them when a thread owns one 4-element slot, n <= 4096; at n = 8192 the V-side set follows the U-side scatter, the
// register file does not hold both): the phases between the barriers then run on registers and LDS only.
// NRT: row tiles a wave handles one after the other against the same x~ fragments (Llama's m = 4096 x 3 and 11008 x 2 would otherwise
// be 768 / 1376 workgroups on 256 CUs, each repeating the prologue): a workgroup owns 16 RT NRT rows.
// YF32: u_y is fp32 (the accumulator of quipamd_decode_bigp_v_gemm) and is rounded to fp16 on load -- what a cast launch in between would do.
// BITS: 2, or 4 (the 4-bit STREAM container: --wbits 4 and, with maxq = 7, --wbits 3): a 1 KiB tile is then 16 rows x 128 columns, a wave owns
// twice as many chunks of the same K range, the conversion is the uniform-offset one (value = 16 + code).
// OPS (round 5: more than FG_MAXBS rows per step): the prologue ALONE, one workgroup per (batch row, layer group) -- grid (bs, ngroups);
// x~ of the row goes to global memory (Gg.y = f16 [bs, N], image order) instead of the LDS operand of the MFMAs, no weights are read, and
// the dequant-GEMM is the next launch (quipamd_dequant_gemm_grouped on the same decode-order codes: the weights stream ONCE for all rows).
// A workgroup that repeats the prologue for every row (the loop below) costs ~2 us per extra row; 16 rows as 16 workgroups cost one.
template <int P, int Q, bool HAS_U, bool HAS_RES, int NORM, int RT, int CPW, int NRT, bool YF32 = false, int BITS = 2, bool OPS = false>
__global__ __launch_bounds__(1024) void fused_gemm_kernel(FusedArgs G, float two_over_maxq, float c0)
{
    typedef PassDims<P, Q> D;
    // Multi-exponent dequantisation (dq_common.h: 10 instead of 16 VALU per packed dword, paid for with sum OFF_k x~_k in the prologue and
    // the reducer) where a wave converts several row tiles against one x~ (Llama's NRT = 4 / 8: the conversion is 3.4 us of VALU issue in
    // the gate/up launch); with one tile per wave (OPT) the bookkeeping costs more than it saves (measured, profiles/r03E).
    constexpr bool ME = NRT >= 4 && BITS == 2;
    typedef DeqSelME<ME, BITS> DQ;
    constexpr int KC = 512 / BITS;                                              // columns of a 1 KiB tile
    constexpr int N = D::N, NV = D::NV, NS = FG_NW / RT, NCH = N / KC, XTS = N + 8;       // x~ row stride (halves)
    constexpr int PB = NRT == 6 ? 3 : NRT < 4 ? NRT : 4;                        // row tiles (per parallel row slot) parked per batch
    constexpr bool EARLY = NV == 1;
    static_assert(NS * CPW == NCH, "chunks = slots x chunks per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *XT = reinterpret_cast<uint16_t *>(smem);                         // [bs][N + 8] f16
    char *pass = smem + (size_t)FG_MAXBS * XTS * 2;
    uint16_t *ZT = reinterpret_cast<uint16_t *>(pass);
    uint16_t *Z1 = reinterpret_cast<uint16_t *>(pass + D::ZT_B);
    float *ZF = reinterpret_cast<float *>(pass + D::ZT_B + D::Z1_B);
    float *park = reinterpret_cast<float *>(pass);                              // [NS][RT PB][4][64]: after the last pass
    constexpr size_t PARK_B = (size_t)(FG_NW * PB * 256) * 4;
    float *red = reinterpret_cast<float *>(pass + (D::BYTES > PARK_B ? D::BYTES : PARK_B));            // [2][16] norm, [bs][16] sum x~, [bs][16] sum OFF x~
    // Round 6: the factor fragments of both operators are SHARED through LDS (p <= 64).  Rounds 3-5: every wave that owns tiles pulled its
    // own copy from global memory -- 64 x 64: 16 waves x 4 KiB per operator for 16 KiB of distinct fragments, 128 of the 288 KiB a workgroup of
    // Llama's q / k / v launch drags through its CU's one vector-memory path (in-situ stamps, profiles/r06_decode_stamps.txt: the waves finish
    // ISSUING their requests at 2100 / 4000 / 5400 / 7500 clocks, four at a time -- the launch is bound by that queue).  Now thread t loads piece
    // entry t of an operator ONCE (F0 and F1 are piece-major [tile][k-step][lane] uint4, exactly the LDS image), the fragments reach LDS with
    // the scatter in front of the pass that needs them, and a wave reads the pieces of its tiles from there (as fused_pair_kernel does at 128 x 64).
    // Measured (profiles/r06h_decode_ab_shared_fragments.jsonl, in-situ stamps r06h vs r06c): in the fused launches (<= 4 rows) the shared copy
    // is NEUTRAL to slightly negative -- the first barrier now waits for a fragment entry that is cold (OPT q / k / v launch: first barrier 3990
    // instead of 2970 clocks, the pass behind it 790 instead of 970; whole launch 11771 vs 11533; tok/s equal) -- and in the prologue-only launches
    // of a 5..16-row step it pays (Llama-2-7B, 16 sequences: 3.03 -> 2.99 ms).  So: the OPS launches only.
    constexpr bool SHF = P <= 64 && OPS && QA_SHF;
    constexpr int NF0 = (P / 16) * D::S0, NF1 = (Q / 16) * D::S1, NFR = NF0 + NF1;      // 1 KiB pieces per operator
    static_assert(!SHF || NFR * 64 <= 1024, "one fragment entry per thread");
    uint4 *FRU = reinterpret_cast<uint4 *>(red + (2 + 2 * FG_MAXBS) * FG_NW + 16);                       // [NFR][64] uint4 (SHF)
    uint4 *FRV = FRU + NFR * 64;
    const typename DQ::Consts qc = DQ::make_consts();

    if (G.pf.n > 0 && (int)blockIdx.x >= G.pf.first) {                          // a prefetch workgroup (uniform): touch the lines, leave
        if (blockIdx.y == 0) qa_pf_run(G.pf, 1024u);
        return;
    }
    const int gi = blockIdx.y;
    const FGroup &Gg = G.g[gi];
    const Fop &V = Gg.V;
    // every kernarg field the kernel will ever read, in ONE scalar round trip: hipcc fetches kernarg fields lazily -- one s_load +
    // s_waitcnt lgkmcnt(0) per first use -- and round one of this kernel opened with SEVEN serial kernarg round trips (~2000
    // cycles) and closed with three more in front of the final store.  One asm statement naming them all pins the loads here.
    asm volatile("" ::"s"(Gg.qw), "s"(V.F0), "s"(V.F1), "s"(V.load_idx), "s"(Gg.colscale), "s"(Gg.scale), "s"(Gg.y),
                 "s"(G.U.F0), "s"(G.U.F1), "s"(G.U.store_idx), "s"(G.u_y), "s"(G.u_bias), "s"(G.u_res), "s"(G.t_out),
                 "s"(G.ld_res), "s"(G.ld_t), "s"(G.x), "s"(G.ldx), "s"(G.gamma), "s"(G.beta), "s"(G.eps), "s"(G.floor), "s"(G.bs), "s"(G.m),
                 "s"(G.y_f16));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slot = wave / RT, r = wave - slot * RT;
    const int j = lane & 15, g = lane >> 4;
    const uint32_t rt0 = blockIdx.x * (RT * NRT) + r;                          // row tile of iteration k: rt0 + k RT
    const uint32_t rtmax = (uint32_t)(G.m / 16) - 1;                           // (the last workgroup of a ragged launch re-reads the last tile; its rows are not stored)
    const int bs = G.bs;
    const int b_lo = OPS ? (int)blockIdx.x : 0, b_hi = OPS ? b_lo + 1 : bs;     // the batch rows this workgroup's prologue walks
    FG_STAMP(0);
    QA_LOG(0)

    // ---- every operand of the prologue is requested NOW, in the order the phases consume them; the packed weights go LAST: vector
    // memory returns in order (s_waitcnt vmcnt counts from the oldest), so a wait for the first activations behind a cold HBM
    // weight load would be a wait for HBM (measured: +2000 cycles in front of the first scatter) ------------------------------------
    uint4 w[NRT][CPW];
    PassFrags<P, Q> frU, frV;
    float4 cs[NV];
    uint2 bi[NV], st[NV], rs[NV], gm[NV], bt_[NV], vld[NV], xr[NV];
    constexpr int NCV = (N / 8 + 1023) / 1024;                                  // 16-byte chunks of the U pass's input row per thread
    uint4 yc[NCV], yc2[NCV];
    auto load_u_row = [&](int b) {                                              // what the first scatter needs: the row itself, in ZT order
#pragma unroll
        for (int u = 0; u < NCV; ++u) {
            const int c0_ = tid + 1024 * u;
            const int c = QA_UNCOND ? (c0_ < N / 8 ? c0_ : 0) : c0_;
            if (c < N / 8) {
                if constexpr (YF32) {
                    const float *src = reinterpret_cast<const float *>(G.u_y) + (int64_t)b * N + (uint32_t)(8 * c);
                    yc[u] = *reinterpret_cast<const uint4 *>(src);
                    yc2[u] = *reinterpret_cast<const uint4 *>(src + 4);
                } else {
                    yc[u] = *reinterpret_cast<const uint4 *>((G.u_y + (int64_t)b * N) + (uint32_t)(8 * c));
                }
            }
        }
    };
    auto u_row_f16 = [&](int u) {                                               // the chunk as 8 halves
        if constexpr (YF32) {
            auto f = [](uint32_t a) { return __builtin_bit_cast(float, a); };
            return make_uint4(pack_f16x2(f(yc[u].x), f(yc[u].y)), pack_f16x2(f(yc[u].z), f(yc[u].w)),
                              pack_f16x2(f(yc2[u].x), f(yc2[u].y)), pack_f16x2(f(yc2[u].z), f(yc2[u].w)));
        } else {
            return yc[u];
        }
    };
    uint4 shU = make_uint4(0u, 0u, 0u, 0u), shV = shU;                           // SHF: this thread's entry of U's / V's fragment pieces
    auto load_shared = [&](const Fop &op) {
        const int e_ = tid < NFR * 64 ? tid : 0;                                  // (clamped address + select below: no branch around the load)
        const uint4 *src = e_ < NF0 * 64 ? reinterpret_cast<const uint4 *>(op.F0) + e_ : reinterpret_cast<const uint4 *>(op.F1) + (e_ - NF0 * 64);
        return *src;
    };
    auto frags_from_lds = [&](const uint4 *FR, PassFrags<P, Q> &fr) {
        if (wave < D::NT) {
            const int at = wave % (P / 16), bt = wave % (Q / 16);
#pragma unroll
            for (int S = 0; S < D::S0; ++S) fr.f0[S] = FR[(at * D::S0 + S) * 64 + lane];
#pragma unroll
            for (int S = 0; S < D::S1; ++S) fr.f1[S] = FR[(NF0 + bt * D::S1 + S) * 64 + lane];
        }
    };
    auto load_u_frags = [&]() {
        if constexpr (SHF) {
            shU = load_shared(G.U);
        } else {
            load_f0<P, Q>(G.U, wave, lane, frU);
            load_f1<P, Q>(G.U, wave, lane, frU);
        }
    };
    auto load_u_side = [&](int b) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v4_ = tid + 1024 * u, v4 = QA_UNCOND ? (v4_ < N / 4 ? v4_ : 0) : v4_;
            rs[u] = make_uint2(0u, 0u);
            if (v4 < N / 4) {
                st[u] = *reinterpret_cast<const uint2 *>(G.U.store_idx + 4 * v4);
                bi[u] = *reinterpret_cast<const uint2 *>(G.u_bias + 4 * v4);
                if (HAS_RES) rs[u] = *reinterpret_cast<const uint2 *>((G.u_res + (int64_t)b * G.ld_res) + (uint32_t)(4 * v4));
            }
        }
    };
    auto load_v_frags = [&]() {
        if constexpr (SHF) {
            shV = load_shared(V);
        } else {
            load_f0<P, Q>(V, wave, lane, frV);
            load_f1<P, Q>(V, wave, lane, frV);
        }
    };
    auto load_v_side = [&]() {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v4_ = tid + 1024 * u, v4 = QA_UNCOND ? (v4_ < N / 4 ? v4_ : 0) : v4_;
            gm[u] = bt_[u] = make_uint2(0u, 0u);
            if (v4 < N / 4) {
                if (NORM) gm[u] = *reinterpret_cast<const uint2 *>(G.gamma + 4 * v4);
                if (NORM == 1) bt_[u] = *reinterpret_cast<const uint2 *>(G.beta + 4 * v4);
                cs[u] = *reinterpret_cast<const float4 *>(Gg.colscale + 4 * v4);
                vld[u] = *reinterpret_cast<const uint2 *>(V.load_idx + 4 * v4);
            }
        }
    };
    auto load_x_row = [&](int b) {
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v4_ = tid + 1024 * u, v4 = QA_UNCOND ? (v4_ < N / 4 ? v4_ : 0) : v4_;
            xr[u] = make_uint2(0u, 0u);
            if (v4 < N / 4) xr[u] = *reinterpret_cast<const uint2 *>((G.x + (int64_t)b * G.ldx) + (uint32_t)(4 * v4));
        }
    };
    if (HAS_U) load_u_row(b_lo);
    else load_x_row(b_lo);
    // everything the FIRST phase needs is in the memory pipeline of every wave before anything else is requested: the CU has one
    // vector-memory path (~64 B per clock) and the prologue pulls 100 - 350 KiB through it; without this barrier (no memory wait in
    // it, the waves arrive within a few cycles) wave 15's activations queued behind the other waves' factor fragments
    __syncthreads();
    if (HAS_U && EARLY) {
        load_u_frags();
        load_u_side(b_lo);
    }
    if (EARLY || !HAS_U) {
        load_v_side();
        load_v_frags();
    }
    if constexpr (!OPS) {
#pragma unroll
    for (int k = 0; k < NRT; ++k)
#pragma unroll
        for (int i = 0; i < CPW; ++i) {                                         // HBM, streamed once: nt
            const uint32_t rtk = rt0 + k * RT < rtmax ? rt0 + k * RT : rtmax;
            const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(Gg.qw + ((uint64_t)rtk * NCH + (slot * CPW + i)) * 64 + lane));
            w[k][i] = make_uint4(t[0], t[1], t[2], t[3]);
        }
    }
    const float e_sc = OPS ? 0.f : Gg.scale[0];                                 // needed by the reducer only

    for (int b = b_lo; b < b_hi; ++b) {
        float4 tv[NV];
        if (HAS_U) {
            // ---- t = [relu](U^T y + bias + residual) --------------------------------------------------------------------------------
            if (b > b_lo) {
                load_u_row(b);
                if (EARLY && HAS_RES) {
#pragma unroll
                    for (int u = 0; u < NV; ++u)
                        if (tid + 1024 * u < N / 4) rs[u] = *reinterpret_cast<const uint2 *>((G.u_res + (int64_t)b * G.ld_res) + (uint32_t)(4 * (tid + 1024 * u)));
                }
            }
#pragma unroll
            for (int u = 0; u < NCV; ++u)
                if (tid + 1024 * u < N / 8) copy_chunk_zt<P, Q>(ZT, u_row_f16(u), tid + 1024 * u);
            if (!EARLY) {                                                       // n = 8192: fragments, the gather's operands and the V-side set follow
                load_u_frags();                                                 // the scatter (nothing row-independent stays in registers across a row)
                load_u_side(b);
                load_v_side();
            }
            if constexpr (SHF) {
                if (b == b_lo && tid < NFR * 64) FRU[tid] = shU;                 // (once per launch: the region is nobody else's)
            }
            FG_STAMP(1);                                                         // first loads landed, scatter done
            __syncthreads();
            FG_STAMP(2);
            if constexpr (SHF) frags_from_lds(FRU, frU);
            mix_stages<P, Q>(ZT, Z1, ZF, frU, wave, lane);
            if (!EARLY) load_v_frags();                                         // ... and its fragments under the gather (registers)
            FG_STAMP(3);
            __syncthreads();
            FG_STAMP(4);
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const int v4 = tid + 1024 * u;
                tv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (v4 < N / 4) {
                    float4 t = gather4<P, Q>(ZF, st[u]);
                    const float4 rr = f16x4_to_f32(rs[u]), bb4 = f16x4_to_f32(bi[u]);
                    t = make_float4(fmaxf(t.x + bb4.x + rr.x, G.floor), fmaxf(t.y + bb4.y + rr.y, G.floor),
                                    fmaxf(t.z + bb4.z + rr.z, G.floor), fmaxf(t.w + bb4.w + rr.w, G.floor));
                    uint2 pk;                                                    // the residual stream is fp16: everything downstream sees the rounded value
                    pk.x = pack_f16x2(t.x, t.y);
                    pk.y = pack_f16x2(t.z, t.w);
                    if (G.t_out && (OPS || blockIdx.x == 0) && gi == 0) *reinterpret_cast<uint2 *>((G.t_out + (int64_t)b * G.ld_t) + (uint32_t)(4 * v4)) = pk;
                    tv[u] = f16x4_to_f32(pk);
                }
            }
        } else {
            if (b > b_lo) load_x_row(b);
#pragma unroll
            for (int u = 0; u < NV; ++u) tv[u] = f16x4_to_f32(xr[u]);
        }
        FG_STAMP(5);                                                             // t in registers (gather / x load done)
        if (NORM) {
            // statistics with ONE barrier: every wave reduces its own elements to (mean_w, M2_w) on the DPP network -- two-pass inside the
            // wave, like torch's LayerNorm -- and the NWD waves that hold data are merged with Chan's formula (equal counts):
            //   mean = avg mean_w,   M2 = sum M2_w + cnt sum (mean_w - mean)^2.     RMSNorm: one wave sum of squares.
            constexpr int NWD = (N / 4 < 1024 ? N / 4 : 1024) / 64, CNT = N / NWD;      // waves with data, elements per such wave
            float mean = 0.f, rstd;
            if (NORM == 1) {
                float s1 = 0.f;
#pragma unroll
                for (int u = 0; u < NV; ++u) s1 += (tv[u].x + tv[u].y) + (tv[u].z + tv[u].w);
                const float mw = fg_wave_sum(s1) * (1.0f / (float)CNT);
                float s2 = 0.f;
#pragma unroll
                for (int u = 0; u < NV; ++u) {
                    const float d0 = tv[u].x - mw, d1 = tv[u].y - mw, d2 = tv[u].z - mw, d3 = tv[u].w - mw;
                    s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
                const float m2w = fg_wave_sum(s2);
                if (lane == 0) {
                    red[wave] = mw;
                    red[FG_NW + wave] = m2w;
                }
                __syncthreads();
                float ms = 0.f, m2 = 0.f;
#pragma unroll
                for (int i = 0; i < NWD; ++i) ms += red[i];
                mean = ms * (1.0f / (float)NWD);
#pragma unroll
                for (int i = 0; i < NWD; ++i) {
                    const float dm = red[i] - mean;
                    m2 += red[FG_NW + i] + (float)CNT * dm * dm;
                }
                rstd = rsqrtf(m2 * (1.0f / (float)N) + G.eps);
            } else {
                float s2 = 0.f;
#pragma unroll
                for (int u = 0; u < NV; ++u) s2 += (tv[u].x * tv[u].x + tv[u].y * tv[u].y) + (tv[u].z * tv[u].z + tv[u].w * tv[u].w);
                const float w2 = fg_wave_sum(s2);
                if (lane == 0) red[wave] = w2;
                __syncthreads();
                float t2 = 0.f;
#pragma unroll
                for (int i = 0; i < NWD; ++i) t2 += red[i];
                rstd = rsqrtf(t2 * (1.0f / (float)N) + G.eps);
            }
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                const float4 gmf = f16x4_to_f32(gm[u]), btf = f16x4_to_f32(bt_[u]);
                tv[u] = make_float4((tv[u].x - mean) * rstd * gmf.x + btf.x, (tv[u].y - mean) * rstd * gmf.y + btf.y,
                                    (tv[u].z - mean) * rstd * gmf.z + btf.z, (tv[u].w - mean) * rstd * gmf.w + btf.w);
            }
        }
        FG_STAMP(6);                                                             // norm done
        // ---- x~ = V (h (/) s) ------------------------------------------------------------------------------------------------------
        // (ZT is free: with a U pass its last readers finished before the barrier in front of the gather; ZF's readers -- the gather
        //  above -- finish before the barrier after this scatter, and ZF is written only after the barrier inside mix_stages)
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (tid + 1024 * u < N / 4) {
                const float4 v = make_float4(tv[u].x * cs[u].x, tv[u].y * cs[u].y, tv[u].z * cs[u].z, tv[u].w * cs[u].w);
                scatter4<P, Q>(ZT, v, vld[u]);
            }
        }
        if constexpr (SHF) {
            if (b == b_lo && tid < NFR * 64) FRV[tid] = shV;
        }
        FG_STAMP(7);
        __syncthreads();
        FG_STAMP(8);
        if constexpr (SHF) frags_from_lds(FRV, frV);
        mix_stage1<P, Q>(ZT, Z1, frV, wave, lane);
        FG_STAMP(9);
        __syncthreads();
        FG_STAMP(10);
        const XtSums xp = mix_stage2_xt<P, Q, 16, ME>(Z1, XT + (size_t)(b - b_lo) * XTS, frV, wave, lane);   // x~ in image order = the order of the weights' columns
        if constexpr (!OPS) {
            const float xs1 = fg_wave_sum(xp.s1);
            if (lane == 0) red[2 * FG_NW + b * FG_NW + wave] = xs1;            // waves without tiles publish 0
            if constexpr (ME) {
                const float xso = fg_wave_sum(xp.soff);
                if (lane == 0) red[(2 + FG_MAXBS) * FG_NW + b * FG_NW + wave] = xso;
            }
        }
        FG_STAMP(11);
        __syncthreads();                                                        // x~ row complete; ZT / Z1 free for the next row (or park)
        FG_STAMP(12);
    }

    if constexpr (OPS) {                                                        // the row's x~ leaves as it lies in XT: 16-byte chunks, coalesced
        uint16_t *dst = reinterpret_cast<uint16_t *>(Gg.y) + (int64_t)b_lo * N;
        for (int c = tid; c < N / 8; c += 1024) *reinterpret_cast<uint4 *>(dst + 8 * c) = *reinterpret_cast<const uint4 *>(XT + 8 * c);
        (void)w; (void)e_sc; (void)rt0; (void)rtmax; (void)qc; (void)park; (void)two_over_maxq; (void)c0; (void)slot; (void)r; (void)j; (void)g;
        QA_LOG(1)
        return;
    }
    // ---- dequant + MFMA: this wave's CPW chunks of 256 columns x 16 rows -------------------------------------------------------------
    // MFMA column j = batch row j.  Columns are independent (D[m][n] depends on B[:, n] only), so lanes of columns >= bs may read
    // anything: they read row j % 4 of x~ (allocated, possibly never written) and their results are not stored -- no masking.
    f32x4_t acc[NRT];
#pragma unroll
    for (int k = 0; k < NRT; ++k) acc[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint16_t *xrow = XT + (size_t)(j & (FG_MAXBS - 1)) * XTS + 8 * g;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = slot * CPW + i;
        uint4 xf[DQ::NT];
#pragma unroll
        for (int t = 0; t < DQ::NT; ++t) xf[t] = *reinterpret_cast<const uint4 *>(xrow + c * KC + 32 * t);
#pragma unroll
        for (int k = 0; k < NRT; ++k)
#pragma unroll
            for (int t = 0; t < DQ::NT; ++t) {
                const u32x4 a = DQ::frag(u32x4{w[k][i].x, w[k][i].y, w[k][i].z, w[k][i].w}, t, qc);
                acc[k] = ActF16::mfma(a, u32x4{xf[t].x, xf[t].y, xf[t].z, xf[t].w}, acc[k]);
            }
    }
    FG_STAMP(13);
    // the 16 / RT chunk slots meet in LDS, PB row tiles (per parallel row slot) at a time
#pragma unroll
    for (int h = 0; h < NRT / PB; ++h) {
        if (h > 0) __syncthreads();                                             // the previous batch's reducers are done with park
#pragma unroll
        for (int kk = 0; kk < PB; ++kk) {
            const int k = h * PB + kk;
            float *p = park + ((slot * PB + kk) * RT + r) * 256 + lane;
            p[0] = acc[k][0]; p[64] = acc[k][1]; p[128] = acc[k][2]; p[192] = acc[k][3];
        }
        __syncthreads();
        FG_STAMP(14);
        if (wave < RT * PB) {                                                   // one reducer wave per parked row tile: lane = (batch row, row in tile)
            const int pr = wave, bb = lane >> 4, wr = lane & 15;                // pr = kk RT + r
            const int src = (wr & 3) * 64 + bb + 16 * (wr >> 2);               // [acc component][mfma lane (j = bb, g = wr / 4)]
            float a = 0.f, xsum = 0.f, xoff = 0.f;
#pragma unroll
            for (int v = 0; v < NS; ++v) a += park[(v * PB * RT + pr) * 256 + src];
#pragma unroll
            for (int v = 0; v < FG_NW; ++v) {                                     // every wave published its part of sum x~ (and sum OFF x~)
                xsum += red[2 * FG_NW + bb * FG_NW + v];
                if constexpr (ME) xoff += red[(2 + FG_MAXBS) * FG_NW + bb * FG_NW + v];
            }
            const int64_t row = ((int64_t)blockIdx.x * (RT * NRT) + h * PB * RT + pr) * 16 + wr;
            const float val = e_sc * two_over_maxq * ((a - xoff) - (ME ? c0 : c0 + DeqT<BITS, ActF16>::OFF) * xsum);   // c0 = maxq / 2; ME: the offsets went with xoff
            if (bb < bs && row < G.m) {
                if (G.y_f16) reinterpret_cast<uint16_t *>(Gg.y)[(int64_t)bb * G.m + row] = f32_to_f16_bits(val);
                else reinterpret_cast<float *>(Gg.y)[(int64_t)bb * G.m + row] = val;
            }
        }
    }
    FG_STAMP(15);
    QA_LOG(1)
}

// ---- n = 8192 (OPT's fc1 -> fc2 hand-over): t = relu(U^T y + bias) feeds x~ = V (t (/) s) with no norm and no residual in between ------
// The generic kernel above spends 60 % of its 10 us at this size on (a) pulling 2 x 96 KiB of factor fragments per workgroup through
// the CU's vector-memory path (every wave its own copy) and (b) four passes of random 2- and 4-byte LDS traffic over 8192 elements
// (fp32 image, gather, scale, scatter).  Here
//   * the fragments of BOTH operators go global -> registers -> LDS once per workgroup (2 x 40 KiB, three 16-byte loads per thread each)
//     and every wave reads the ones of its tiles from LDS;
//   * the gather / scatter pair between the operators is ONE scatter: the lane that holds element (a, b) of U's image after stage 2
//     adds its bias, clamps, scales and writes the fp16 value straight to its place in V's input image.  Where that is, and the bias
//     and 1 / s that go with it, are per-lane tables the host prepares once per layer pair in the lane order of the MFMA result
//     (pair_sig: LDS offset in ZT_V, pair_bias, pair_cs: fp16; entry [(wave * 64 + lane) * 8 + 4 i + reg] for tile wave + 16 i).
// bs <= 2 (x~ rows, 2 images and 80 KiB of fragments share the LDS).
template <int P, int Q, int BITS = 2>
__global__ __launch_bounds__(1024) void fused_pair_kernel(FusedArgs G, float two_over_maxq, float c0)
{
    typedef PassDims<P, Q> D;
    typedef DeqT<BITS, ActF16> DQ;
    constexpr int KC = 512 / BITS;
    constexpr int N = D::N, NS = FG_NW, NCH = N / KC, CPW = NCH / NS, XTS = N + 8, MAXBS = 2;
    constexpr int NF0 = (P / 16) * D::S0, NF1 = (Q / 16) * D::S1, NFR = NF0 + NF1;        // 1 KiB fragment pieces per operator (32 + 8)
    constexpr int FPT = (NFR * 64 + 1023) / 1024;                                          // uint4 per thread per operator
    static_assert(D::TPW == 2 && N / 8 == 1024, "128 x 64");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *XT = reinterpret_cast<uint16_t *>(smem);                                     // [MAXBS][N + 8]
    char *pass = smem + (size_t)MAXBS * XTS * 2;
    uint16_t *ZT = reinterpret_cast<uint16_t *>(pass);
    uint16_t *Z1 = reinterpret_cast<uint16_t *>(pass + D::ZT_B);
    float *park = reinterpret_cast<float *>(pass);                                          // [NS][4][64] after the last pass (16 KiB)
    uint4 *FRU = reinterpret_cast<uint4 *>(pass + D::ZT_B + D::Z1_B);                       // [NFR][64] uint4
    uint4 *FRV = FRU + NFR * 64;
    float *red = reinterpret_cast<float *>(FRV + NFR * 64);                                 // [MAXBS][16] sum x~

    if (G.pf.n > 0 && (int)blockIdx.x >= G.pf.first) {
        qa_pf_run(G.pf, 1024u);
        return;
    }
    const FGroup &Gg = G.g[0];
    const Fop &V = Gg.V;
    asm volatile("" ::"s"(Gg.qw), "s"(V.F0), "s"(V.F1), "s"(Gg.scale), "s"(Gg.y), "s"(G.U.F0), "s"(G.U.F1), "s"(G.u_y), "s"(G.floor), "s"(G.bs),
                 "s"(G.m), "s"(G.y_f16), "s"(G.pair_sig), "s"(G.pair_bias), "s"(G.pair_cs));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int bs = G.bs;
    FG_STAMP(0);
    QA_LOG(0)
    FG_WSTAMP(0);

    // ---- requests, in the order of use: the row, U's fragments | (barrier) | the pair tables, V's fragments, the weights ---------------
    uint4 yc = *reinterpret_cast<const uint4 *>(G.u_y + 8 * (uint32_t)tid);
    // an operator's fragments, piece-major [NF0 | NF1][64 lanes]: thread t copies entries t, t + 1024 (M0's 32 pieces) and, t < 512,
    // entry t of M1's 8 pieces.  (Scalars, not an array: a conditionally written array went to scratch, with a wait behind every load.)
    static_assert(NF0 * 64 == 2048 && NF1 * 64 == 512 && FPT == 3, "128 x 64");
    const uint4 *U0 = reinterpret_cast<const uint4 *>(G.U.F0), *U1 = reinterpret_cast<const uint4 *>(G.U.F1);
    const uint4 *V0 = reinterpret_cast<const uint4 *>(V.F0), *V1 = reinterpret_cast<const uint4 *>(V.F1);
    const uint32_t t1 = tid & 511;
    const uint4 fu0 = U0[tid], fu1 = U0[tid + 1024], fu2 = U1[t1];               // (waves 8..15 load M1's entry again; they do not store it)
    FG_WSTAMP(1);
    __syncthreads();                                                            // first-phase requests of every wave are queued before the rest
    FG_WSTAMP(2);
    const uint4 sg = G.pair_sig[tid], pb = G.pair_bias[tid], pc = G.pair_cs[tid];
    const uint4 fv0 = V0[tid], fv1 = V0[tid + 1024], fv2 = V1[t1];
    uint4 w[CPW];
    const uint32_t rt = blockIdx.x;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(Gg.qw + ((uint64_t)rt * NCH + (wave * CPW + i)) * 64 + lane));
        w[i] = make_uint4(t[0], t[1], t[2], t[3]);
    }
    const float e_sc = Gg.scale[0];
    const int at1 = wave % (P / 16), bt2 = wave % (Q / 16);                     // the factor tile of this wave's stage-1 / stage-2 tiles

    for (int b = 0; b < bs; ++b) {
        if (b > 0) yc = *reinterpret_cast<const uint4 *>((G.u_y + (int64_t)b * N) + 8 * (uint32_t)tid);
        FG_WSTAMP(3);
        copy_chunk_zt<P, Q>(ZT, yc, tid);
        FG_WSTAMP(4);
        if (b == 0) {
            FRU[tid] = fu0;
            FRU[tid + 1024] = fu1;
            if (tid < 512) FRU[2048 + tid] = fu2;
        }
        FG_STAMP(1);
        FG_WSTAMP(5);
        __syncthreads();
        FG_STAMP(2);
        FG_WSTAMP(6);
        PassFrags<P, Q> fr;
#pragma unroll
        for (int S = 0; S < D::S0; ++S) fr.f0[S] = FRU[(at1 * D::S0 + S) * 64 + lane];
#pragma unroll
        for (int S = 0; S < D::S1; ++S) fr.f1[S] = FRU[(NF0 + bt2 * D::S1 + S) * 64 + lane];
        mix_stage1<P, Q>(ZT, Z1, fr, wave, lane);
        FG_STAMP(3);
        __syncthreads();                                                        // Z1 complete; ZT (U's input) dead: V's input image goes there
        FG_STAMP(4);
        // stage 2 of U^T with the hand-over in its epilogue: relu(. + bias) (/) s, fp16, into V's input image
        {
            auto half_of = [](const uint4 &v, int e) -> uint32_t {               // 16-bit entry e (0..7) of a 16-byte table row
                const uint32_t d = (e >> 1) == 0 ? v.x : (e >> 1) == 1 ? v.y : (e >> 1) == 2 ? v.z : v.w;
                return (e & 1) ? d >> 16 : d & 0xffffu;
            };
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int tile = wave + FG_NW * i;
                const int bt = tile % (Q / 16), at = tile / (Q / 16);
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                const uint16_t *arow = Z1 + (16 * at + j) * D::QS + 8 * g;
#pragma unroll
                for (int S = 0; S < D::S1; ++S) {
                    const uint4 a = *reinterpret_cast<const uint4 *>(arow + 32 * S);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, fr.f1[S]), acc, 0, 0, 0);
                }
                (void)bt;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int e = 4 * i + reg;
                    const float bia = f16_bits_to_f32((uint16_t)half_of(pb, e)), csc = f16_bits_to_f32((uint16_t)half_of(pc, e));
                    const float t = f16_bits_to_f32(f32_to_f16_bits(fmaxf(acc[reg] + bia, G.floor)));    // t exists as an fp16 value (the round-2 launches stored it)
                    ZT[half_of(sg, e)] = f32_to_f16_bits(t * csc);
                }
            }
        }
        if (b == 0) {
            FRV[tid] = fv0;
            FRV[tid + 1024] = fv1;
            if (tid < 512) FRV[2048 + tid] = fv2;
        }
        FG_STAMP(7);
        __syncthreads();
        FG_STAMP(8);
#pragma unroll
        for (int S = 0; S < D::S0; ++S) fr.f0[S] = FRV[(at1 * D::S0 + S) * 64 + lane];
#pragma unroll
        for (int S = 0; S < D::S1; ++S) fr.f1[S] = FRV[(NF0 + bt2 * D::S1 + S) * 64 + lane];
        mix_stage1<P, Q>(ZT, Z1, fr, wave, lane);
        FG_STAMP(9);
        __syncthreads();
        FG_STAMP(10);
        float xpart = mix_stage2_xt<P, Q>(Z1, XT + (size_t)b * XTS, fr, wave, lane).s1;
        xpart = fg_wave_sum(xpart);
        if (lane == 0) red[b * FG_NW + wave] = xpart;
        FG_STAMP(11);
        __syncthreads();
        FG_STAMP(12);
    }

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const uint16_t *xrow = XT + (size_t)(j & (MAXBS - 1)) * XTS + 8 * g;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int c = wave * CPW + i;
        uint4 xf[DQ::NT];
#pragma unroll
        for (int t = 0; t < DQ::NT; ++t) xf[t] = *reinterpret_cast<const uint4 *>(xrow + c * KC + 32 * t);
#pragma unroll
        for (int t = 0; t < DQ::NT; ++t) {
            const u32x4 a = DQ::frag(u32x4{w[i].x, w[i].y, w[i].z, w[i].w}, t);
            acc = ActF16::mfma(a, u32x4{xf[t].x, xf[t].y, xf[t].z, xf[t].w}, acc);
        }
    }
    FG_STAMP(13);
    {
        float *p_ = park + wave * 256 + lane;
        p_[0] = acc[0]; p_[64] = acc[1]; p_[128] = acc[2]; p_[192] = acc[3];
    }
    __syncthreads();
    FG_STAMP(14);
    if (wave == 0) {
        const int bb = lane >> 4, wr = lane & 15;
        const int src = (wr & 3) * 64 + bb + 16 * (wr >> 2);
        float a = 0.f, xsum = 0.f;
#pragma unroll
        for (int v = 0; v < NS; ++v) a += park[v * 256 + src];
#pragma unroll
        for (int v = 0; v < FG_NW; ++v) xsum += red[(bb & (MAXBS - 1)) * FG_NW + v];
        const int64_t row = (int64_t)blockIdx.x * 16 + wr;
        const float val = e_sc * two_over_maxq * (a - c0 * xsum);
        if (bb < bs) {
            if (G.y_f16) reinterpret_cast<uint16_t *>(Gg.y)[(int64_t)bb * G.m + row] = f32_to_f16_bits(val);
            else reinterpret_cast<float *>(Gg.y)[(int64_t)bb * G.m + row] = val;
        }
    }
    FG_STAMP(15);
    QA_LOG(1)
}

template <int P, int Q, int BITS> int launch_pair(const FusedArgs &A, float maxq, hipStream_t s)
{
    typedef PassDims<P, Q> D;
    constexpr int NFR = (P / 16) * D::S0 + (Q / 16) * D::S1;
    const size_t lds = (size_t)2 * (D::N + 8) * 2 + D::ZT_B + D::Z1_B + (size_t)2 * NFR * 1024 + 2 * FG_NW * 4 + 64;
    auto kern = fused_pair_kernel<P, Q, BITS>;
    static QaPerDevice attr;
    const int d = attr.dev();
    if (d < 0 || !attr.done[d]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "decode_fused_gemm (pair): cannot raise dynamic LDS to %zu", lds);
        if (d >= 0) attr.done[d] = true;
    }
    FusedArgs Ap = A;
    const unsigned gxp = (unsigned)(A.m / 16);
    Ap.pf = qa_pf_take();
    Ap.pf.first = (int)gxp;
    kern<<<dim3(gxp + (Ap.pf.n ? QA_PF_WGS : 0), 1), 1024, lds, s>>>(Ap, 2.0f / maxq, DeqT<BITS, ActF16>::OFF + 0.5f * maxq);
    QA_LAUNCH_CHECK("quipamd_decode_fused_gemm (pair)");
    return QUIPAMD_OK;
}

template <int P, int Q, int NRT, bool OPS = false> constexpr size_t fused_lds()
{
    typedef PassDims<P, Q> D;
    const size_t parkb = (size_t)(FG_NW * (NRT < 4 ? NRT : 4) * 256) * 4;
    const size_t frags = (P <= 64 && OPS && QA_SHF) ? (size_t)2 * ((P / 16) * D::S0 + (Q / 16) * D::S1) * 1024 : 0;          // shared fragment pieces of U and V (round 6)
    return (size_t)FG_MAXBS * (D::N + 8) * 2 + (D::BYTES > parkb ? D::BYTES : parkb) + (2 + 2 * FG_MAXBS) * FG_NW * 4 + 64 + frags + 64;
}

thread_local float g_fused_maxq = 3.f;      // set by the entry point right before the dispatch (same thread): 3, 7 (3-bit codes in the 4-bit container) or 15

template <int P, int Q, bool HAS_U, bool HAS_RES, int NORM, int RT, int CPW, int NRT, bool YF32 = false, int BITS = 2, bool OPS = false>
int launch_fused(const FusedArgs &A, int ngroups, hipStream_t s)
{
    const size_t lds = fused_lds<P, Q, NRT, OPS>();
    auto kern = fused_gemm_kernel<P, Q, HAS_U, HAS_RES, NORM, RT, CPW, NRT, YF32, BITS, OPS>;
    static QaPerDevice attr;
    const int d = attr.dev();
    if (d < 0 || !attr.done[d]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "decode_fused_gemm: cannot raise dynamic LDS to %zu", lds);
        if (d >= 0) attr.done[d] = true;
    }
    const float maxq = g_fused_maxq;
    const unsigned gx = OPS ? (unsigned)A.bs : (unsigned)((A.m / 16 + RT * NRT - 1) / (RT * NRT));
    FusedArgs Ap = A;
    Ap.pf = qa_pf_take();
    Ap.pf.first = (int)gx;
    kern<<<dim3(gx + (Ap.pf.n ? QA_PF_WGS : 0), (unsigned)ngroups), 1024, lds, s>>>(Ap, 2.0f / maxq, 0.5f * maxq);
    QA_LAUNCH_CHECK("quipamd_decode_fused_gemm");
    return QUIPAMD_OK;
}

// the prologue-only launches (OPS): the combinations of dispatch_fused at the base tiling of each operator shape; the code container does
// not matter (no weights are read)
template <int P, int Q, int RT, int CPW>
int dispatch_ops(const FusedArgs &A, bool u, bool res, int norm, bool yf32, int ngroups, hipStream_t s)
{
    if constexpr (P == 64 && Q == 64) {
        if (yf32) return launch_fused<P, Q, true, true, 2, RT, CPW, 1, true, 2, true>(A, ngroups, s);
    }
    if constexpr (P == 128) {
        if (u && !res && norm == 0) return launch_fused<P, Q, true, false, 0, RT, CPW, 1, false, 2, true>(A, ngroups, s);
        if (!u && norm == 0) return launch_fused<P, Q, false, false, 0, RT, CPW, 1, false, 2, true>(A, ngroups, s);
    } else {
        if (u && res) return norm == 0 ? launch_fused<P, Q, true, true, 0, RT, CPW, 1, false, 2, true>(A, ngroups, s)
                           : norm == 1 ? launch_fused<P, Q, true, true, 1, RT, CPW, 1, false, 2, true>(A, ngroups, s)
                                       : launch_fused<P, Q, true, true, 2, RT, CPW, 1, false, 2, true>(A, ngroups, s);
        if (!u) return norm == 0 ? launch_fused<P, Q, false, false, 0, RT, CPW, 1, false, 2, true>(A, ngroups, s)
                     : norm == 1 ? launch_fused<P, Q, false, false, 1, RT, CPW, 1, false, 2, true>(A, ngroups, s)
                                 : launch_fused<P, Q, false, false, 2, RT, CPW, 1, false, 2, true>(A, ngroups, s);
    }
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "decode_fused_gemm (ops only): %d x %d has no kernel for (U %d, residual %d, norm %d)", P, Q, (int)u, (int)res, norm);
}

// the combinations a decoder block needs (each is a 1300-line kernel):
//   64 x 32 (d = 2048) and 64 x 64 (d = 4096):  [U + residual | no U] x [no norm | LayerNorm | RMSNorm]
//   128 x 64 (OPT d = 8192):                    [U, no residual, no norm]  [no U, no norm]
template <int P, int Q, int RT, int CPW, int NRT, int BITS = 2>
int dispatch_fused(const FusedArgs &A, bool u, bool res, int norm, int ngroups, hipStream_t s)
{
    if (u && res) return norm == 0 ? launch_fused<P, Q, true, true, 0, RT, CPW, NRT, false, BITS>(A, ngroups, s)
                       : norm == 1 ? launch_fused<P, Q, true, true, 1, RT, CPW, NRT, false, BITS>(A, ngroups, s)
                                   : launch_fused<P, Q, true, true, 2, RT, CPW, NRT, false, BITS>(A, ngroups, s);
    if (!u) return norm == 0 ? launch_fused<P, Q, false, false, 0, RT, CPW, NRT, false, BITS>(A, ngroups, s)
                 : norm == 1 ? launch_fused<P, Q, false, false, 1, RT, CPW, NRT, false, BITS>(A, ngroups, s)
                             : launch_fused<P, Q, false, false, 2, RT, CPW, NRT, false, BITS>(A, ngroups, s);
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "decode_fused_gemm: %d x %d has no kernel for (U %d, residual %d, norm %d)", P, Q, (int)u, (int)res, norm);
}

bool fop_ok(const quipamd_fop &o, int p, int q) { return o.F0 && o.F1 && o.load_idx && o.store_idx && o.p == p && o.q == q; }

}   // namespace

extern "C" int quipamd_decode_fused_gemm(const quipamd_fused_gemm_args *a, void *stream)
{
    QA_REQUIRE(a, QUIPAMD_ERR_ARG, "decode_fused_gemm: null args");
    QA_REQUIRE(a->act_dtype == QUIPAMD_F16, QUIPAMD_ERR_UNSUPPORTED, "decode_fused_gemm: fp16 activations only");
    QA_REQUIRE(a->bits >= 2 && a->bits <= 4, QUIPAMD_ERR_UNSUPPORTED, "decode_fused_gemm: 2-, 3- or 4-bit qfn-b codes (bits = %d)", a->bits);
    const bool w4 = a->bits != 2;                                               // 3-bit codes ride in the 4-bit container (maxq = 7)
    g_fused_maxq = (float)((1 << a->bits) - 1);
    QA_REQUIRE(a->ngroups >= 1 && a->ngroups <= FG_MAXG, QUIPAMD_ERR_ARG, "decode_fused_gemm: 1..%d groups", FG_MAXG);
    const bool ops_only = a->ops_only != 0;
    QA_REQUIRE(a->bs >= 0 && a->bs <= (ops_only ? 1024 : FG_MAXBS), QUIPAMD_ERR_SHAPE, "decode_fused_gemm: bs %lld > %d", (long long)a->bs, ops_only ? 1024 : FG_MAXBS);
    if (a->bs == 0) return QUIPAMD_OK;
    const int p = a->V[0].p, q = a->V[0].q;
    const int64_t n = (int64_t)p * q;
    FusedArgs A;
    A.U = a->U;
    A.u_y = (const uint16_t *)a->u_y; A.u_bias = (const uint16_t *)a->u_bias; A.u_res = (const uint16_t *)a->u_residual; A.t_out = (uint16_t *)a->t_out;
    A.ld_res = a->ld_residual; A.ld_t = a->ld_t; A.floor = a->u_relu ? 0.f : -INFINITY;
    A.x = (const uint16_t *)a->x; A.ldx = a->ldx;
    A.gamma = (const uint16_t *)a->ln_gamma; A.beta = (const uint16_t *)a->ln_beta; A.eps = a->ln_eps;
    A.pair_sig = A.pair_bias = A.pair_cs = nullptr;
    A.pf.n = 0;
    A.bs = (int)a->bs; A.m = a->m; A.y_f16 = a->y_dtype == QUIPAMD_F16;
    QA_REQUIRE(a->y_dtype == QUIPAMD_F16 || a->y_dtype == QUIPAMD_F32, QUIPAMD_ERR_ARG, "decode_fused_gemm: y_dtype f32 or f16");
    QA_REQUIRE(a->norm >= 0 && a->norm <= 2 && (a->norm == 0 || a->ln_gamma) && (a->norm != 1 || a->ln_beta), QUIPAMD_ERR_ARG,
               "decode_fused_gemm: norm %d needs gamma (and beta for LayerNorm)", a->norm);
    if (a->has_u) {
        QA_REQUIRE(!a->t_out || a->t_out != a->u_residual, QUIPAMD_ERR_ARG, "decode_fused_gemm: t_out must not alias u_residual");
        QA_REQUIRE(fop_ok(a->U, p, q) && a->u_y && a->u_bias, QUIPAMD_ERR_ARG,
                   "decode_fused_gemm: the output-side operator must be %d x %d like the activation-side one, with u_y and u_bias "
                   "(zeros where the layer has none)", p, q);
        QA_REQUIRE((!a->u_residual || (a->ld_residual >= n && a->ld_residual % 4 == 0)) && (!a->t_out || (a->ld_t >= n && a->ld_t % 4 == 0)),
                   QUIPAMD_ERR_SHAPE, "decode_fused_gemm: residual / t_out row strides");
    } else {
        QA_REQUIRE(a->x && a->ldx >= n && a->ldx % 4 == 0, QUIPAMD_ERR_ARG, "decode_fused_gemm: x [bs, ldx] needed without an output-side operator");
    }
    for (int i = 0; i < FG_MAXG; ++i) {
        const int k = i < a->ngroups ? i : 0;
        QA_REQUIRE(fop_ok(a->V[k], p, q) && a->colscale[k] && (ops_only || (a->qweight[k] && a->scale[k])) && a->y[k], QUIPAMD_ERR_ARG,
                   "decode_fused_gemm: null pointer / operator shape in group %d", k);
        A.g[i].V = a->V[k]; A.g[i].colscale = a->colscale[k]; A.g[i].qw = (const uint4 *)a->qweight[k]; A.g[i].scale = a->scale[k]; A.g[i].y = a->y[k];
    }
    hipStream_t s = (hipStream_t)stream;
    const bool u = a->has_u != 0, res = u && a->u_residual != nullptr;
    const bool yf32 = u && a->u_y_dtype == QUIPAMD_F32;
    QA_REQUIRE(!u || yf32 || a->u_y_dtype == QUIPAMD_F16, QUIPAMD_ERR_ARG, "decode_fused_gemm: u_y_dtype f16 or f32");
    // fp32 u_y = the accumulator of quipamd_decode_bigp_v_gemm: Llama's down_proj -> next block's q / k / v
    QA_REQUIRE(!yf32 || (p == 64 && q == 64 && res && a->norm == 2 && (!w4 || ops_only)), QUIPAMD_ERR_UNSUPPORTED,
               "decode_fused_gemm: fp32 u_y has a kernel for 2-bit 64 x 64 with residual and RMSNorm only");
    if (ops_only) {                                                              // y[k] = x~ of group k, f16 [bs, n]
        QA_REQUIRE(!a->t_out || a->has_u, QUIPAMD_ERR_ARG, "decode_fused_gemm (ops only): t_out needs the output-side operator");
        if (p == 64 && q == 32) return dispatch_ops<64, 32, 2, 1>(A, u, res, a->norm, yf32, a->ngroups, s);
        if (p == 64 && q == 64) return dispatch_ops<64, 64, 1, 1>(A, u, res, a->norm, yf32, a->ngroups, s);
        if (p == 128 && q == 64) return dispatch_ops<128, 64, 1, 2>(A, u, res, a->norm, yf32, a->ngroups, s);
        return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "decode_fused_gemm (ops only): operator %d x %d (64 x 32, 64 x 64, 128 x 64)", p, q);
    }
    if (p == 64 && q == 32) {
        QA_REQUIRE(a->m > 0 && a->m % 32 == 0, QUIPAMD_ERR_SHAPE, "decode_fused_gemm: m %% 32 (m = %lld)", (long long)a->m);
        if (w4) return dispatch_fused<64, 32, 2, 2, 1, 4>(A, u, res, a->norm, a->ngroups, s);
        return dispatch_fused<64, 32, 2, 1, 1>(A, u, res, a->norm, a->ngroups, s);
    }
    if (p == 64 && q == 64) {
        QA_REQUIRE(a->m > 0 && a->m % 16 == 0, QUIPAMD_ERR_SHAPE, "decode_fused_gemm: m %% 16 (m = %lld)", (long long)a->m);
        // row tiles per wave: as few as keep the grid within one round of 256 workgroups (each repeats the prologue; the conversion of a
        // tile is ~1 us of a wave's VALU time): 11008 x 2 rows = 1376 tiles: 6 per wave = 230 workgroups (the last one of each group ragged)
        // instead of 8 = 172
        const int64_t tpg = a->m / 16;
        auto wgs = [&](int n) { return (tpg + n - 1) / n * a->ngroups; };
        const int nrt = wgs(1) <= 256 ? 1 : wgs(4) <= 256 ? 4 : wgs(6) <= 256 ? 6 : 8;
        if (w4)         // (twice the packed dwords per row tile: at most 4 tiles per wave stay in registers)
            return nrt == 1 ? dispatch_fused<64, 64, 1, 2, 1, 4>(A, u, res, a->norm, a->ngroups, s) : dispatch_fused<64, 64, 1, 2, 4, 4>(A, u, res, a->norm, a->ngroups, s);
        if (yf32) return nrt == 8 ? launch_fused<64, 64, true, true, 2, 1, 1, 8, true>(A, a->ngroups, s)
                       : nrt == 6 ? launch_fused<64, 64, true, true, 2, 1, 1, 6, true>(A, a->ngroups, s)
                       : nrt == 4 ? launch_fused<64, 64, true, true, 2, 1, 1, 4, true>(A, a->ngroups, s)
                                  : launch_fused<64, 64, true, true, 2, 1, 1, 1, true>(A, a->ngroups, s);
        return nrt == 8 ? dispatch_fused<64, 64, 1, 1, 8>(A, u, res, a->norm, a->ngroups, s)
             : nrt == 6 ? dispatch_fused<64, 64, 1, 1, 6>(A, u, res, a->norm, a->ngroups, s)
             : nrt == 4 ? dispatch_fused<64, 64, 1, 1, 4>(A, u, res, a->norm, a->ngroups, s)
                        : dispatch_fused<64, 64, 1, 1, 1>(A, u, res, a->norm, a->ngroups, s);
    }
    if (p == 128 && q == 64) {
        QA_REQUIRE(a->m > 0 && a->m % 16 == 0, QUIPAMD_ERR_SHAPE, "decode_fused_gemm: m %% 16 (m = %lld)", (long long)a->m);
        if (u && !res && a->norm == 0 && a->pair_sig && a->pair_bias && a->pair_cs && a->bs <= 2 && a->ngroups == 1 && !a->t_out) {
            A.pair_sig = (const uint4 *)a->pair_sig; A.pair_bias = (const uint4 *)a->pair_bias; A.pair_cs = (const uint4 *)a->pair_cs;
            return w4 ? launch_pair<128, 64, 4>(A, g_fused_maxq, s) : launch_pair<128, 64, 2>(A, g_fused_maxq, s);
        }
        if (w4) {
            if (u && !res && a->norm == 0) return launch_fused<128, 64, true, false, 0, 1, 4, 1, false, 4>(A, a->ngroups, s);
            if (!u && a->norm == 0) return launch_fused<128, 64, false, false, 0, 1, 4, 1, false, 4>(A, a->ngroups, s);
        }
        if (u && !res && a->norm == 0) return launch_fused<128, 64, true, false, 0, 1, 2, 1>(A, a->ngroups, s);
        if (!u && a->norm == 0) return launch_fused<128, 64, false, false, 0, 1, 2, 1>(A, a->ngroups, s);
        return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "decode_fused_gemm: 128 x 64 runs (U, no residual, no norm) and (no U, no norm) only");
    }
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "decode_fused_gemm: operator %d x %d (64 x 32, 64 x 64, 128 x 64)", p, q);
}
