// decode_attn.hip -- single-token attention for the decode loop (benchmark(), opt.py:431-482)
//
// One generated token per sequence: append this step's k, v to the static KV cache at position *pos, then
//   o[b, head] = softmax(scale * q . K[0..pos]^T) V[0..pos].
// In the reference's benchmark() this is HF OPTAttention in eager PyTorch -- nine launches per block (two index_copy,
// matmul, scale + mask, cast, softmax, cast, matmul, reshape), ~45 us of the ~130 us a packed block took once the
// Linears ran on K2 / K3, i.e. the decode loop's tok/s measured attention glue, not the low-bit path.  Here: one launch.
//   grid = bs * heads workgroups of 256 threads; `pos` is read from DEVICE memory so the launch can sit in a hipGraph
//   that is replayed for every token (the position advances on the device).
//   phase 1  scores: thread t (+256, ...) owns cache position t: 128-byte row of K (8 x 16 B), q in registers, fp32 dot;
//            scores and the running max go through LDS, exp in fp32.
//   phase 2  lane = (row slot, 8-dimension chunk): 16-byte loads, 8 rows of V per instruction and 8 instructions in
//            flight per wave; row slots fold with shuffles, the 4 waves meet in LDS; normalised, written in the
//            activation dtype.
// fp32 throughout (the eager chain rounds scores and probabilities to fp16); the result differs from it by that rounding.
#include "common.h"
#include "fpass.h"
#include "probe.h"
#include "wavered.h"
#include "prefetch.h"

// A/B builds (python __graft_entry__.py --variant nokpf -DQA_NO_KPF): round 5's request order -- no early K rows, no touched V lines
#ifdef QA_NO_KPF
#define QA_KPF 0
#else
#define QA_KPF 1
#endif

namespace {

template <class TI, int HD>
__global__ __launch_bounds__(256) void decode_attn_kernel(const typename DT<TI>::storage *q, const typename DT<TI>::storage *k,
                                                         const typename DT<TI>::storage *v, typename DT<TI>::storage *kc,
                                                         typename DT<TI>::storage *vc, const int64_t *pos_p,
                                                         typename DT<TI>::storage *out, int heads, int64_t maxlen, float scale,
                                                         int64_t ldq)
{
    typedef typename DT<TI>::storage S;
    extern __shared__ __attribute__((aligned(16))) float sm[];       // scores[maxlen] | red[8] | part[4][HD]
    float *scores = sm, *red = sm + maxlen, *part = sm + maxlen + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, head = blockIdx.x % heads;
    QA_STAMP(0);
    QA_LOG(0)
    const int64_t pos = *pos_p, T = pos + 1;
    QA_STAMP(1);                                                    // the position landed
    const S *qh = q + (int64_t)b * ldq + head * HD, *kh = k + (int64_t)b * ldq + head * HD, *vh = v + (int64_t)b * ldq + head * HD;
    S *kcb = kc + ((int64_t)b * heads + head) * maxlen * HD, *vcb = vc + ((int64_t)b * heads + head) * maxlen * HD;

    if (pos < 0 || pos >= maxlen) return;                           // uniform; a full cache is the caller's error
    // Round 6 (profiles/r06_decode_stamps.txt): the scores phase opened with a COLD round trip for the K rows of the cache, p V with one for the
    // V rows.  Order of the requests now (vector memory returns in order): this token's k, v and q first -- written by the previous launch,
    // L2 -- then the first 256 K rows of the cache into registers and a touch of the V rows' lines, which travel under the append and the
    // conversion of q.  This token's own row is not in the cache yet: its owner reads it from k.
    S knew = S(), vnew = S();
    if (tid < HD) knew = kh[tid];
    else if (tid < 2 * HD) vnew = vh[tid - HD];
    uint4 qraw[HD / 8];
#pragma unroll
    for (int e8 = 0; e8 < HD; e8 += 8) qraw[e8 / 8] = *reinterpret_cast<const uint4 *>(qh + e8);
    uint4 kpre[HD / 8];
    {
        const int64_t tpre = tid < maxlen ? tid : 0;
#pragma unroll
        for (int e8 = 0; e8 < HD; e8 += 8) kpre[e8 / 8] = *reinterpret_cast<const uint4 *>(kcb + tpre * HD + e8);
    }
    qa_sink_t sink = 0;                                             // (csrc/prefetch.h: the touches' destination, kept alive to the end)
    if constexpr (QA_KPF) qa_touch_lines<HD * 2 / 128>(vcb + (tid < T ? (int64_t)tid : T - 1) * HD, sink);
    // append this token; the barrier (workgroup-scope fence) makes it visible to the reads below
    if (tid < HD) kcb[pos * HD + tid] = knew;
    else if (tid < 2 * HD) vcb[pos * HD + tid - HD] = vnew;
    __syncthreads();
    QA_STAMP(8);                                                    // k, v of this token landed and appended + barrier

    float qr[HD];
#pragma unroll
    for (int e8 = 0; e8 < HD; e8 += 8) {
        S raw[8];
        *reinterpret_cast<uint4 *>(raw) = qraw[e8 / 8];
#pragma unroll
        for (int e = 0; e < 8; ++e) qr[e8 + e] = DT<TI>::load(raw, e) * scale;
    }
    // the first batch of V rows does not depend on the scores: request it now, one round trip earlier
    constexpr int CH = HD / 8, RS = 64 / CH;                         // chunks per row, rows per instruction (hd 64: 8, 8)
    const int ch = lane % CH, rsub = lane / CH;
    uint4 vraw0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t t = (int64_t)wave * RS + (int64_t)u * 4 * RS + rsub;
        vraw0[u] = t < T ? *reinterpret_cast<const uint4 *>(vcb + t * HD + 8 * ch) : make_uint4(0, 0, 0, 0);
    }

    // ---- phase 1: scores --------------------------------------------------------------------------------------------
    float mx = -INFINITY;
    auto dot_row = [&](auto get) {                                   // q . (one K row), the row handed over as 16-byte pieces
        float acc = 0.f;
#pragma unroll
        for (int e8 = 0; e8 < HD; e8 += 8) {
            S raw[8];
            *reinterpret_cast<uint4 *>(raw) = get(e8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qr[e8 + e], DT<TI>::load(raw, e), acc);
        }
        return acc;
    };
    if (tid < T) {                                                   // first pass: the prefetched row (this token's own row straight from k)
        const float acc = tid == pos ? dot_row([&](int e8) { return *reinterpret_cast<const uint4 *>(kh + e8); })
                          : QA_KPF ? dot_row([&](int e8) { return kpre[e8 / 8]; })
                                   : dot_row([&](int e8) { return *reinterpret_cast<const uint4 *>(kcb + (int64_t)tid * HD + e8); });
        scores[tid] = acc;
        mx = fmaxf(mx, acc);
    }
    for (int64_t t = (int64_t)tid + 256; t < T; t += 256) {
        const S *row = t == pos ? kh : kcb + t * HD;                 // (same values as the appended copy)
        const float acc = dot_row([&](int e8) { return *reinterpret_cast<const uint4 *>(row + e8); });
        scores[t] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_reduce<true>(mx);
    if (lane == 0) red[wave] = mx;
    QA_STAMP(9);                                                    // scores: q and the K rows landed
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int64_t t = tid; t < T; t += 256) {
        const float p = __expf(scores[t] - mx);
        scores[t] = p;
        sum += p;
    }
    sum = wave_reduce<false>(sum);
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    QA_STAMP(10);
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);

    // ---- phase 2: o = p V ---------------------------------------------------------------------------------------------
    // lane = (row slot, 8-dim chunk): one 16-byte load per lane covers 64 / CH rows per instruction, up to 8 such loads
    // in flight, so a ~130-position context is ONE round trip per wave (one 128-byte row per instruction was five).
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    bool first = true;
    for (int64_t tb = (int64_t)wave * RS; tb < T; tb += 4 * RS * 8) {
        uint4 raw[8];
        float pw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t t = tb + (int64_t)u * 4 * RS + rsub;
            const bool ok = t < T;
            if (first) raw[u] = vraw0[u];
            else raw[u] = ok ? *reinterpret_cast<const uint4 *>(vcb + t * HD + 8 * ch) : make_uint4(0, 0, 0, 0);
            pw[u] = ok ? scores[t] : 0.f;
        }
        first = false;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const S *rv = reinterpret_cast<const S *>(&raw[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pw[u], DT<TI>::load(rv, e), o[e]);
        }
    }
    // rows of one chunk sit CH lanes apart: fold the row slots, then the 4 waves through LDS
#pragma unroll
    for (int off = CH; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += __shfl_xor(o[e], off);
    if (rsub == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[wave * HD + 8 * ch + e] = o[e];
    }
    QA_STAMP(11);
    __syncthreads();
    if (tid < HD) {
        const float r = (part[tid] + part[HD + tid]) + (part[2 * HD + tid] + part[3 * HD + tid]);
        DT<TI>::store(out + (int64_t)b * ldq + head * HD, tid, r * inv);
    }
    qa_touch_done(sink);
    QA_STAMP(12);
    QA_LOG(1)
}

template <class TI, int HD>
int launch_attn(const void *q, const void *k, const void *v, void *kc, void *vc, const int64_t *pos, void *out, int64_t bs,
                int heads, int64_t maxlen, float scale, int64_t ldq, hipStream_t s)
{
    typedef typename DT<TI>::storage S;
    const size_t lds = (size_t)(maxlen + 8 + 4 * HD) * sizeof(float);
    auto kern = decode_attn_kernel<TI, HD>;
    static size_t reserved[64] = {};                                // per device and instantiation: the attribute is per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool known = dev >= 0 && dev < 64;
    if (lds > 48 * 1024 && (!known || lds > reserved[dev])) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "decode_attention: cannot reserve %zu B of LDS", lds);
        if (known) reserved[dev] = lds;
    }
    kern<<<(unsigned)(bs * heads), 256, lds, s>>>((const S *)q, (const S *)k, (const S *)v, (S *)kc, (S *)vc, pos, (S *)out, heads,
                                                  maxlen, scale, ldq);
    QA_LAUNCH_CHECK("decode_attention");
    return QUIPAMD_OK;
}

}   // namespace

// ---- rotary position embedding of one decode step (Llama: llama.py:418-471 benchmark() runs HF LlamaAttention, whose
// apply_rotary_pos_emb is q * cos + rotate_half(q) * sin, five eager launches each for q and k) ----------------------------------
// In place on q and k [bs, heads * hd] at position *pos (device memory, so the launch replays inside a hipGraph):
//   x[i] <- x[i] c_i - x[i + hd/2] s_i;   x[i + hd/2] <- x[i + hd/2] c_i + x[i] s_i,   c_i = cos[pos][i], s_i = sin[pos][i],  i < hd/2
// (the tables hold HF's duplicated-halves layout [maxpos, hd]; fp32 arithmetic, one rounding to the activation dtype).
template <class TI>
__global__ __launch_bounds__(256) void rope_kernel(typename DT<TI>::storage *q, typename DT<TI>::storage *k, const float *cos_t,
                                                   const float *sin_t, const int64_t *pos_p, int64_t table_rows, int64_t total, int heads,
                                                   int kv_heads, int hd, int64_t ldq, int64_t ldk)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (b, head of q then of k, i < hd/2)
    if (idx >= total) return;
    const int half = hd / 2, i = (int)(idx % half);
    const int64_t t = idx / half;
    const int hh = (int)(t % (heads + kv_heads));
    const int64_t b = t / (heads + kv_heads);
    const int64_t pos = *pos_p;
    if (pos < 0 || pos >= table_rows) return;                             // past the tables: q / k stay as they are (like decode_attention past maxlen)
    const float c = cos_t[pos * hd + i], s = sin_t[pos * hd + i];
    typename DT<TI>::storage *x = hh < heads ? q + b * ldq + (int64_t)hh * hd : k + b * ldk + (int64_t)(hh - heads) * hd;
    const float a = DT<TI>::load(x, i), bb = DT<TI>::load(x, i + half);
    DT<TI>::store(x, i, a * c - bb * s);
    DT<TI>::store(x, i + half, bb * c + a * s);
}

extern "C" int quipamd_rope_inplace(void *q, void *k, const float *cos_table, const float *sin_table, int64_t table_rows, const int64_t *pos,
                                    int dtype, int64_t bs, int heads, int kv_heads, int hd, int64_t ldq, int64_t ldk, void *stream)
{
    QA_REQUIRE(bs >= 0 && heads > 0 && kv_heads > 0 && hd > 0 && hd % 2 == 0 && table_rows > 0, QUIPAMD_ERR_SHAPE, "rope_inplace: bad shape");
    if (bs == 0) return QUIPAMD_OK;
    QA_REQUIRE(q && k && cos_table && sin_table && pos, QUIPAMD_ERR_ARG, "rope_inplace: null pointer");
    QA_REQUIRE(dtype == QUIPAMD_F16 || dtype == QUIPAMD_BF16, QUIPAMD_ERR_UNSUPPORTED, "rope_inplace: f16 / bf16 only");
    const int64_t total = bs * (heads + kv_heads) * (hd / 2);
    const unsigned grid = (unsigned)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == QUIPAMD_F16)
        rope_kernel<F16><<<grid, 256, 0, s>>>((uint16_t *)q, (uint16_t *)k, cos_table, sin_table, pos, table_rows, total, heads, kv_heads, hd, ldq, ldk);
    else
        rope_kernel<BF16><<<grid, 256, 0, s>>>((uint16_t *)q, (uint16_t *)k, cos_table, sin_table, pos, table_rows, total, heads, kv_heads, hd, ldq, ldk);
    QA_LAUNCH_CHECK("quipamd_rope_inplace");
    return QUIPAMD_OK;
}

namespace {

// ---- attention with the output-side operators of q / k / v in its prologue (decode step, fp16) -------------------------------------
// The three dequant-GEMMs of a block hand over y_q, y_k, y_v in the projected basis; q = U_q^T y_q + b_q (likewise k, v) was a
// launch of its own (4.7 us).  A head needs 64 (128) entries of each -- but behind the permutation they depend on the whole
// image, so every (batch row, head) workgroup runs the three small operator passes itself (3 x 0.4 MFLOP on the fp16 matrix pipe,
// images in LDS, fpass.h) and gathers only its head's slice; rotary (Llama) is applied to that slice in LDS.  Then the same
// phases as decode_attn_kernel, with q read from LDS and this step's k, v appended to the cache from LDS.
struct AttnUArgs {
    Fop U[3];
    const uint16_t *y[3];                 // f16 [bs, n]
    const uint16_t *bias[3];              // f16 [n]
    uint16_t *kc, *vc, *out;
    const int64_t *pos;
    const float *cos_t, *sin_t;           // rotary tables [table_rows, HD] or null
    int64_t table_rows, maxlen, ldo;
    int heads;
    float scale;
    QaPfList pf;                          // operands of a later launch, touched by QA_PF_WGS extra workgroups (csrc/prefetch.h); n = 0: none
};

// 768 threads: waves 0-3 / 4-7 / 8-11 run the operator pass of q / k / v side by side (round 3a ran the three passes one after the other on
// four waves: two passes' worth of copy, stage 1 and stage 2 on the launch's critical path); waves 4-11 leave after their pass, the first
// 256 threads go on to the gather, the rotary embedding and the attention itself.
// (NGRP = 1 is the round-3a form: the three passes one after the other on 256 threads.  With q held as fp32 head dim 128 did not fit the 168
//  registers of a 768-thread workgroup -- 43 spilled, 605 -> 594 tok/s on Llama -- so q is held as packed fp16 pairs now.)
// MH (round 6, with NGRP = 3): a workgroup serves THREE heads of its sequence -- after the operator passes (which produce the whole q, k, v rows
// anyway: every workgroup of the one-head form redoes them for its single head) the k and v wave groups do not leave: each of the three groups
// gathers ITS head's slices, appends, scores, softmax, p V in its own LDS region.  A third of the workgroups and of the redundant passes: at 16
// sequences x 32 heads 176 workgroups in one round instead of 512 in two.  (The groups stay consecutive wave quadruples: with a group = the
// waves of one residue mod 3 -- meant to put the three heads' busy first waves on different SIMDs -- the 16-sequence step took 2.03 ms
// instead of 1.35, profiles/r06K_decode_bs.jsonl.)
template <int HD, int P, int Q, int NGRP, bool MH = false>
__global__ __launch_bounds__(256 * NGRP) void decode_attn_u_kernel(AttnUArgs G)
{
    typedef uint16_t S;
    typedef F16 TI;
    constexpr int NW = 4, N = P * Q;
    typedef PassDims<P, Q, NW> D;
    extern __shared__ __attribute__((aligned(16))) char smem_u[];
    // [3 image sets][q k v slices f16 3 x HD][scores maxlen][red 8][part 4 x HD]
    char *img = smem_u;
    static_assert(!MH || NGRP == 3, "three heads per workgroup ride on the three wave groups");
    const size_t AB = (3 * HD * 2 + 32 + (size_t)(G.maxlen + 8 + 4 * HD) * sizeof(float) + 15) & ~(size_t)15;     // one group's attention buffers
    char *abuf = smem_u + 3 * D::BYTES + (MH ? (size_t)(threadIdx.x >> 8) * AB : 0);
    uint16_t *qkv = reinterpret_cast<uint16_t *>(abuf);
    float *scores = reinterpret_cast<float *>(abuf + 3 * HD * 2 + 32);
    float *red = scores + G.maxlen, *part = red + 8;
    asm volatile("" ::"s"(G.U[0].F0), "s"(G.U[0].F1), "s"(G.U[0].store_idx), "s"(G.U[1].F0), "s"(G.U[1].F1),
                 "s"(G.U[1].store_idx), "s"(G.U[2].F0), "s"(G.U[2].F1), "s"(G.U[2].store_idx),
                 "s"(G.y[0]), "s"(G.y[1]), "s"(G.y[2]), "s"(G.bias[0]), "s"(G.bias[1]), "s"(G.bias[2]), "s"(G.kc), "s"(G.vc), "s"(G.out),
                 "s"(G.pos), "s"(G.cos_t), "s"(G.sin_t), "s"(G.table_rows), "s"(G.maxlen), "s"(G.ldo), "s"(G.heads), "s"(G.scale));
    if (G.pf.n > 0 && (int)blockIdx.x >= G.pf.first) {                // a prefetch workgroup (uniform): touch the lines, leave
        qa_pf_run(G.pf, 256u * NGRP);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hpb = MH ? (G.heads + 2) / 3 : G.heads;                 // workgroups per sequence
    const int b = blockIdx.x / hpb, head_ = MH ? (int)(blockIdx.x % hpb) * 3 + (wave >> 2) : (int)(blockIdx.x % hpb);
    const bool active = head_ < G.heads;                              // (MH: the last workgroup of a sequence may hold fewer than three heads)
    const int head = active ? head_ : 0;                              // an idle group walks head 0's addresses and stores nothing
    const int ta = MH ? (tid & 255) : tid, wa = MH ? (wave & 3) : wave;   // thread / wave within the group that runs the attention
    QA_STAMP(0);
    QA_LOG(0)
    const int64_t pos = *G.pos, T = pos + 1;
    QA_STAMP(1);                                                      // the position landed (first dependent round trip)
    if (pos < 0 || pos >= G.maxlen) return;                         // uniform; a full cache is the caller's error
    S *kcb = G.kc + ((int64_t)b * G.heads + head) * G.maxlen * HD, *vcb = G.vc + ((int64_t)b * G.heads + head) * G.maxlen * HD;

    // ---- prologue: q, k, v slices of this head ---------------------------------------------------------------------------------------
    constexpr int NCV = N / 8 / 256;                                  // 16-byte chunks of one projection's output row per thread of a group
    static_assert(NCV * 256 * 8 == N, "n = 2048 or 4096");
    const int grp = wave >> 2, w4 = wave & 3, t4 = tid & 255;         // NGRP = 3: operator of this wave group (0 q, 1 k, 2 v); wave / thread within it
    constexpr int NOP = 3 / NGRP;                                     // operators a wave group runs
    uint4 yc[NOP][NCV];                                               // y arrives in ZT order (fpass.h copy_chunk_zt): no index vectors
    PassFrags<P, Q> fr[NOP];
#pragma unroll
    for (int oi = 0; oi < NOP; ++oi) {
        const int o = NGRP == 3 ? grp : oi;
        const uint16_t *ysrc = o == 0 ? G.y[0] : o == 1 ? G.y[1] : G.y[2];
#pragma unroll
        for (int u = 0; u < NCV; ++u) yc[oi][u] = *reinterpret_cast<const uint4 *>((ysrc + (int64_t)b * N) + 8 * (uint32_t)(t4 + 256 * u));
    }
#pragma unroll
    for (int oi = 0; oi < NOP; ++oi) {
        const int o = NGRP == 3 ? grp : oi;
        const Fop &Ug = o == 0 ? G.U[0] : o == 1 ? G.U[1] : G.U[2];
        load_f0<P, Q, NW>(Ug, w4, lane, fr[oi]);
        load_f1<P, Q, NW>(Ug, w4, lane, fr[oi]);
    }
    // the gather of the head's slice: element t < 3 HD = (op = t / HD, e = t % HD), owned by thread t % 256 (HD = 128: two rounds)
    constexpr int GR = (3 * HD + 255) / 256;
    uint32_t gst[GR];
    uint16_t gbi[GR];
    float rc[GR], rs_[GR];
#pragma unroll
    for (int it = 0; it < GR; ++it) {
        const int t = ta + 256 * it;
        gst[it] = 0; gbi[it] = 0; rc[it] = 1.f; rs_[it] = 0.f;
        if ((MH || wave < 4) && t < 3 * HD) {                       // (the gather is the first 256 threads' job; MH: every group's, for its head)
            const int gop = t / HD, ge = t - gop * HD, i = head * HD + ge;
            gst[it] = gop == 0 ? G.U[0].store_idx[i] : gop == 1 ? G.U[1].store_idx[i] : G.U[2].store_idx[i];
            gbi[it] = gop == 0 ? G.bias[0][i] : gop == 1 ? G.bias[1][i] : G.bias[2][i];
            if (G.cos_t && gop < 2 && pos < G.table_rows) {         // rotary on q and k: x[e] c - x[e + HD/2] s ; x[e + HD/2] c + x[e] s
                rc[it] = G.cos_t[pos * HD + (ge % (HD / 2))];
                rs_[it] = G.sin_t[pos * HD + (ge % (HD / 2))];
            }
        }
    }
    // Round 6 (profiles/r06_decode_stamps.txt): the scores phase opened with a COLD round trip -- the K rows of the cache were requested only
    // after the whole prologue, 3700 clocks on the last wave for 97 positions.  The first 256 rows (one per thread of the scoring group) do not
    // depend on anything computed here: request them NOW, behind the prologue's own operands (vector memory returns in order: they land last,
    // under the operator passes).  The row of THIS token (t == pos) is not in the cache yet: the thread that owns it takes it from the LDS
    // slice instead -- its prefetched registers hold whatever the cache held before.  Head dim 64 only: 16 more 16-byte registers per lane do
    // not fit the 768-thread form at head dim 128.
    constexpr bool KPF = HD <= 64 && QA_KPF;                          // (MH without it: no scratch, same speed -- profiles/r06F_mh_kpf.jsonl)
    uint4 kpre[KPF ? HD / 8 : 1];
    if constexpr (KPF) {
        const int64_t tpre = ta < G.maxlen ? ta : 0;
#pragma unroll
        for (int e8 = 0; e8 < HD; e8 += 8) kpre[e8 / 8] = make_uint4(0u, 0u, 0u, 0u);
        if (MH || wave < 4) {                                         // the scoring group(s) only (one head: waves 4..11 leave after their operator pass)
#pragma unroll
            for (int e8 = 0; e8 < HD; e8 += 8) kpre[e8 / 8] = *reinterpret_cast<const uint4 *>(kcb + tpre * HD + e8);
        }
    }
    // ... and the lines of the V rows (and of the K rows at head dim 128) are TOUCHED here: loads nobody waits for, behind the prologue's own
    // requests (vector memory returns in order: in front of them they would put an HBM round trip before the first activations)
    qa_sink_t sink = 0;                                               // (csrc/prefetch.h: the touches' destination, kept alive to the end)
    if constexpr (HD <= 64 && QA_KPF) {   // (no branch around the statement: threads past the end touch row T - 1 again)
        const int64_t tt = ta < T ? ta : T - 1;
        qa_touch_lines<HD * 2 / 128>(vcb + tt * HD, sink);
    }
    QA_STAMP(2);                                                      // every request of the prologue issued
#pragma unroll
    for (int oi = 0; oi < NOP; ++oi) {
        const int o = NGRP == 3 ? grp : oi;
        uint16_t *ZT = reinterpret_cast<uint16_t *>(img + o * D::BYTES);
#pragma unroll
        for (int u = 0; u < NCV; ++u) copy_chunk_zt<P, Q>(ZT, yc[oi][u], t4 + 256 * u);
    }
    QA_STAMP(3);                                                      // y landed and copied into the image
    __syncthreads();
    QA_STAMP(4);
#pragma unroll
    for (int oi = 0; oi < NOP; ++oi) {
        const int o = NGRP == 3 ? grp : oi;
        uint16_t *ZT = reinterpret_cast<uint16_t *>(img + o * D::BYTES), *Z1 = reinterpret_cast<uint16_t *>(img + o * D::BYTES + D::ZT_B);
        mix_stage1<P, Q, NW>(ZT, Z1, fr[oi], w4, lane);
    }
    __syncthreads();
    QA_STAMP(5);                                                      // stage 1 (its factor fragments landed) + barrier
#pragma unroll
    for (int oi = 0; oi < NOP; ++oi) {
        const int o = NGRP == 3 ? grp : oi;
        uint16_t *Z1 = reinterpret_cast<uint16_t *>(img + o * D::BYTES + D::ZT_B);
        float *ZF = reinterpret_cast<float *>(img + o * D::BYTES + D::ZT_B + D::Z1_B);
        mix_stage2<P, Q, NW>(Z1, ZF, fr[oi], w4, lane);
    }
    __syncthreads();
    QA_STAMP(6);                                                      // stage 2 + barrier
    if (NGRP == 3 && !MH && wave >= 4) {                              // the k and v groups are done (hardware barriers count live waves only)
        if constexpr (HD > 64 && QA_KPF) {
            // head dim 128: any early touch of the cache rows cost this instantiation a register array in scratch memory (it sits at its register
            // budget), so the waves that LEAVE here touch the K and V lines of the first 512 positions on their way out -- ~1000 clocks ahead
            // of the scores instead of ~5000, still ahead
            const int64_t tt = tid - 256 < T ? tid - 256 : T - 1;
            qa_touch_lines<HD * 2 / 128>(kcb + tt * HD, sink);
            qa_touch_lines<HD * 2 / 128>(vcb + tt * HD, sink);
        }
        qa_touch_done(sink);
        return;
    }
#pragma unroll
    for (int it = 0; it < GR; ++it) {
        const int t = ta + 256 * it;
        if (t < 3 * HD) {
            constexpr int qsh = __builtin_ctz(Q);
            const int gop = t / HD;
            const float *ZF = reinterpret_cast<const float *>(img + gop * D::BYTES + D::ZT_B + D::Z1_B);
            const float v = ZF[(gst[it] >> qsh) * D::QF + (gst[it] & (Q - 1))] + f16_bits_to_f32(gbi[it]);
            qkv[t] = f32_to_f16_bits(v);                             // q, k, v exist as fp16 values, like the separate launch wrote them
        }
    }
    __syncthreads();
    QA_STAMP(7);                                                      // gather of the head's slice (index / bias loads landed) + barrier
    if (G.cos_t) {                                                    // rotate_half form of HF's apply_rotary_pos_emb (llama.py:418-471), fp32 math
        float r[GR];
#pragma unroll
        for (int it = 0; it < GR; ++it) {
            const int t = ta + 256 * it;
            r[it] = 0.f;
            if (t < 2 * HD) {
                const int ge = t % HD;
                const float a = f16_bits_to_f32(qkv[t]);
                const float o2 = f16_bits_to_f32(qkv[ge < HD / 2 ? t + HD / 2 : t - HD / 2]);
                r[it] = ge < HD / 2 ? a * rc[it] - o2 * rs_[it] : a * rc[it] + o2 * rs_[it];
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < GR; ++it) {
            const int t = ta + 256 * it;
            if (t < 2 * HD) qkv[t] = f32_to_f16_bits(r[it]);
        }
        __syncthreads();
    }
    // append this token's k, v; the barrier makes them visible to the cache reads below
#pragma unroll
    for (int it = 0; it < GR; ++it) {
        const int t = ta + 256 * it;
        if (!active) continue;
        if (t >= HD && t < 2 * HD) kcb[pos * HD + t - HD] = qkv[t];
        else if (t >= 2 * HD && t < 3 * HD) vcb[pos * HD + t - 2 * HD] = qkv[t];
    }
    __syncthreads();
    QA_STAMP(8);                                                      // (rotary) + cache append + barrier

    // q as packed fp16 pairs (half the registers of an fp32 copy: head dim 128 then fits the 768-thread form), scores on v_dot2_f32_f16
    // (exact products, fp32 accumulation), the 1 / sqrt(hd) applied to the finished score
    uint32_t qp[HD / 2];
#pragma unroll
    for (int e8 = 0; e8 < HD; e8 += 8) {
        const uint4 r4 = *reinterpret_cast<const uint4 *>(qkv + e8);
        qp[e8 / 2 + 0] = r4.x; qp[e8 / 2 + 1] = r4.y; qp[e8 / 2 + 2] = r4.z; qp[e8 / 2 + 3] = r4.w;
    }
    constexpr int CH = HD / 8, RS = 64 / CH;
    const int ch = lane % CH, rsub = lane / CH;
    uint4 vraw0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int64_t t = (int64_t)wa * RS + (int64_t)u * 4 * RS + rsub;
        vraw0[u] = t < T ? *reinterpret_cast<const uint4 *>(vcb + t * HD + 8 * ch) : make_uint4(0, 0, 0, 0);
    }
    float mx = -INFINITY;
    for (int64_t t = ta; t < T; t += 256) {
        const S *row = kcb + t * HD;
        const bool pre = KPF && t == ta;                             // first pass: the row came in with the prologue's requests
        const bool own = t == pos;                                    // this token's row: from the LDS slice (k = qkv[HD .. 2 HD))
        float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
        for (int e8 = 0; e8 < HD; e8 += 8) {
            uint4 k4;
            if (KPF && own) k4 = *reinterpret_cast<const uint4 *>(qkv + HD + e8);
            else if (pre) k4 = kpre[KPF ? e8 / 8 : 0];
            else k4 = *reinterpret_cast<const uint4 *>(row + e8);
            acc0 = ActF16::dot2(qp[e8 / 2 + 0], k4.x, acc0);
            acc1 = ActF16::dot2(qp[e8 / 2 + 1], k4.y, acc1);
            acc0 = ActF16::dot2(qp[e8 / 2 + 2], k4.z, acc0);
            acc1 = ActF16::dot2(qp[e8 / 2 + 3], k4.w, acc1);
        }
        const float acc = (acc0 + acc1) * G.scale;
        scores[t] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_reduce<true>(mx);
    if (lane == 0) red[wa] = mx;
    QA_STAMP(9);                                                      // scores: the K rows of the cache landed
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int64_t t = ta; t < T; t += 256) {
        const float p_ = __expf(scores[t] - mx);
        scores[t] = p_;
        sum += p_;
    }
    sum = wave_reduce<false>(sum);
    if (lane == 0) red[4 + wa] = sum;
    __syncthreads();
    QA_STAMP(10);                                                     // softmax
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.f;
    bool first = true;
    for (int64_t tb = (int64_t)wa * RS; tb < T; tb += 4 * RS * 8) {
        uint4 raw[8];
        float pw[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t t = tb + (int64_t)u * 4 * RS + rsub;
            const bool ok = t < T;
            if (first) raw[u] = vraw0[u];
            else raw[u] = ok ? *reinterpret_cast<const uint4 *>(vcb + t * HD + 8 * ch) : make_uint4(0, 0, 0, 0);
            pw[u] = ok ? scores[t] : 0.f;
        }
        first = false;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const S *rv = reinterpret_cast<const S *>(&raw[u]);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaf(pw[u], DT<TI>::load(rv, e), o[e]);
        }
    }
#pragma unroll
    for (int off = CH; off < 64; off <<= 1)
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += __shfl_xor(o[e], off);
    if (rsub == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part[wa * HD + 8 * ch + e] = o[e];
    }
    QA_STAMP(11);                                                     // p V
    __syncthreads();
    if (ta < HD && active) {
        const float r = (part[ta] + part[HD + ta]) + (part[2 * HD + ta] + part[3 * HD + ta]);
        DT<TI>::store(G.out + (int64_t)b * G.ldo + head * HD, ta, r * inv);
    }
    qa_touch_done(sink);
    QA_STAMP(12);
    QA_LOG(1)
}

template <int HD, int P, int Q, int NGRP, bool MH = false> int launch_attn_u_n(const AttnUArgs &A, int64_t bs, hipStream_t s)
{
    typedef PassDims<P, Q, 4> D;
    const size_t ab = (3 * HD * 2 + 32 + (size_t)(A.maxlen + 8 + 4 * HD) * sizeof(float) + 15) & ~(size_t)15;     // one group's attention buffers
    const size_t lds = 3 * D::BYTES + (MH ? 3 : 1) * ab;
    auto kern = decode_attn_u_kernel<HD, P, Q, NGRP, MH>;
    static size_t reserved[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const bool known = dev >= 0 && dev < 64;
    if (lds > 160 * 1024) return qa_fail(QUIPAMD_ERR_SHAPE, "decode_attention_fused: %zu B of LDS (maxlen too large for this operator shape)", lds);
    if (lds > 48 * 1024 && (!known || lds > reserved[dev])) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "decode_attention_fused: cannot reserve %zu B of LDS", lds);
        if (known) reserved[dev] = lds;
    }
    const int64_t wgs = bs * (MH ? (A.heads + 2) / 3 : A.heads);
    AttnUArgs Ap = A;
    Ap.pf = qa_pf_take();
    Ap.pf.first = (int)wgs;
    kern<<<(unsigned)wgs + (Ap.pf.n ? QA_PF_WGS : 0), 256 * NGRP, lds, s>>>(Ap);
    QA_LAUNCH_CHECK("decode_attention_fused");
    return QUIPAMD_OK;
}

// Round 6 (profiles/r06C_decode_wglog_kron_bs16.txt): at 16 sequences the launch is 512 workgroups of 12 waves at 120 registers -- one per CU,
// TWO rounds (first -> last workgroup start 4.25 us, the launch 9.7 us).  Two forms were built against that:
//   * four waves (one wave group runs the three operator passes in turn; two workgroups share a CU: one round): SLOWER -- OPT-1.3B 16 sequences
//     1.41-1.42 -> 1.45 ms per step, Llama-2-7B 2.98 -> 3.07 (profiles/r06D_attn_forms.jsonl): what the second round cost, the serial passes and
//     two workgroups on one CU's memory path cost again.  Forced only (one_group_from).
//   * three heads per workgroup (MH above): a third of the workgroups and of the redundant operator passes; default from more than one
//     (sequence, head) pair per CU on.
int g_attn_u_one_group_from = 0;       // (sequence, head) pairs from which the 4-wave form is launched (0 = never)
int g_attn_u_three_heads_from = 257;   // ... from which a workgroup serves three heads (0 = never)
template <int HD, int P, int Q> int launch_attn_u(const AttnUArgs &A, int64_t bs, hipStream_t s)
{
    if (g_attn_u_one_group_from > 0 && bs * A.heads >= g_attn_u_one_group_from) return launch_attn_u_n<HD, P, Q, 1>(A, bs, s);
    if (g_attn_u_three_heads_from > 0 && bs * A.heads >= g_attn_u_three_heads_from) return launch_attn_u_n<HD, P, Q, 3, true>(A, bs, s);
    return launch_attn_u_n<HD, P, Q, 3>(A, bs, s);
}

}   // namespace

namespace {

// ---- the output side of a packed layer on its own: out = [relu](U^T y + bias + residual), y in ZT order (the end of the last block) ----
struct UOnlyArgs {
    Fop U;
    const uint16_t *y, *bias, *res;       // f16 [bs, n] (ZT order), f16 [n], f16 [bs, ld_res] or null
    uint16_t *out;                        // f16 [bs, ld_out]
    int64_t ld_res, ld_out;
    float floor;
};

template <int P, int Q> __global__ __launch_bounds__(1024) void u_only_kernel(UOnlyArgs G)
{
    typedef PassDims<P, Q> D;
    constexpr int N = D::N, NV = D::NV, NCV = (N / 8 + 1023) / 1024;
    extern __shared__ __attribute__((aligned(16))) char smem_o[];
    uint16_t *ZT = reinterpret_cast<uint16_t *>(smem_o), *Z1 = reinterpret_cast<uint16_t *>(smem_o + D::ZT_B);
    float *ZF = reinterpret_cast<float *>(smem_o + D::ZT_B + D::Z1_B);
    asm volatile("" ::"s"(G.U.F0), "s"(G.U.F1), "s"(G.U.store_idx), "s"(G.y), "s"(G.bias), "s"(G.res), "s"(G.out), "s"(G.ld_res), "s"(G.ld_out), "s"(G.floor));
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint4 yc[NCV];
#pragma unroll
    for (int u = 0; u < NCV; ++u)
        if (tid + 1024 * u < N / 8) yc[u] = *reinterpret_cast<const uint4 *>((G.y + (int64_t)b * N) + 8 * (uint32_t)(tid + 1024 * u));
    PassFrags<P, Q> fr;
    load_f0<P, Q>(G.U, wave, lane, fr);
    load_f1<P, Q>(G.U, wave, lane, fr);
    uint2 st[NV], bi[NV], rs[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int v4 = tid + 1024 * u;
        rs[u] = make_uint2(0u, 0u);
        if (v4 < N / 4) {
            st[u] = *reinterpret_cast<const uint2 *>(G.U.store_idx + 4 * v4);
            bi[u] = *reinterpret_cast<const uint2 *>(G.bias + 4 * v4);
            if (G.res) rs[u] = *reinterpret_cast<const uint2 *>((G.res + (int64_t)b * G.ld_res) + (uint32_t)(4 * v4));
        }
    }
#pragma unroll
    for (int u = 0; u < NCV; ++u)
        if (tid + 1024 * u < N / 8) copy_chunk_zt<P, Q>(ZT, yc[u], tid + 1024 * u);
    __syncthreads();
    mix_stages<P, Q>(ZT, Z1, ZF, fr, wave, lane);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int v4 = tid + 1024 * u;
        if (v4 < N / 4) {
            float4 t = gather4<P, Q>(ZF, st[u]);
            const float4 rr = f16x4_to_f32(rs[u]), bb = f16x4_to_f32(bi[u]);
            t = make_float4(fmaxf(t.x + bb.x + rr.x, G.floor), fmaxf(t.y + bb.y + rr.y, G.floor), fmaxf(t.z + bb.z + rr.z, G.floor),
                            fmaxf(t.w + bb.w + rr.w, G.floor));
            uint2 pk;
            pk.x = pack_f16x2(t.x, t.y);
            pk.y = pack_f16x2(t.z, t.w);
            *reinterpret_cast<uint2 *>((G.out + (int64_t)b * G.ld_out) + (uint32_t)(4 * v4)) = pk;
        }
    }
}

template <int P, int Q> int launch_u_only(const UOnlyArgs &A, int64_t bs, hipStream_t s)
{
    const size_t lds = PassDims<P, Q>::BYTES;
    auto kern = u_only_kernel<P, Q>;
    static QaPerDevice attr;
    const int d = attr.dev();
    if (lds > 48 * 1024 && (d < 0 || !attr.done[d])) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "decode_u_only: cannot reserve %zu B of LDS", lds);
        if (d >= 0) attr.done[d] = true;
    }
    kern<<<(unsigned)bs, 1024, lds, s>>>(A);
    QA_LAUNCH_CHECK("quipamd_decode_u_only");
    return QUIPAMD_OK;
}

// ---- greedy token of one decode step: argmax over the vocabulary, one workgroup per batch row ---------------------------------------
// (benchmark(), opt.py:463-480: `torch.argmax(out.logits[0, -1])` -- torch's generic reduction takes 18 us for 50272 logits;
// ties go to the smallest index like torch.argmax)
template <class TI> __global__ __launch_bounds__(1024) void argmax_rows_kernel(const typename DT<TI>::storage *x, int64_t n, int64_t ld, int64_t *out, int vec)
{
    __shared__ float bv[16];
    __shared__ int bi[16];
    const typename DT<TI>::storage *row = x + (int64_t)blockIdx.x * ld;
    float best = -INFINITY;
    int idx = 0x7fffffff;
    auto take = [&](float v, int i) {
        if (v > best || (v == best && i < idx)) { best = v; idx = i; }
    };
    int64_t done = 0;
    if constexpr (sizeof(typename DT<TI>::storage) == 2) {
        // 16-bit logits, rows 16-byte aligned (vec): eight values per load and EVERY load of the row in flight at once -- 50272 entries are
        // 6284 loads over 1024 threads, seven per thread (one value per load and trip: 49 dependent round trips, 19 us; eight in flight: 13)
        if (vec) {
            const int64_t n8 = n >> 3;
            const uint4 *row8 = reinterpret_cast<const uint4 *>(row);
            constexpr int VU = 8;
            for (int64_t c0 = threadIdx.x; c0 < n8; c0 += 1024 * VU) {
                uint4 v[VU];
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    const int64_t c = c0 + 1024 * u;
                    v[u] = row8[c < n8 ? c : n8 - 1];                          // clamped address: the load is unconditional
                }
#pragma unroll
                for (int u = 0; u < VU; ++u) {
                    const int64_t c = c0 + 1024 * u;
                    if (c < n8) {
                        const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const typename DT<TI>::storage h = (typename DT<TI>::storage)(w[e >> 1] >> (16 * (e & 1)));
                            take(DT<TI>::load(&h, 0), (int)(8 * c + e));
                        }
                    }
                }
            }
            done = n8 << 3;
        }
    }
    // eight independent loads in flight per thread (the tail of a vector row; rows that are not aligned; fp32 logits)
    constexpr int AU = 8;
    for (int64_t i0 = done + threadIdx.x; i0 < n; i0 += 1024 * AU) {
        float v[AU];
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int64_t i = i0 + 1024 * u;
            const float t = DT<TI>::load(row, i < n ? i : n - 1);          // clamped address + select: the load is unconditional
            v[u] = i < n ? t : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < AU; ++u) {
            const int64_t i = i0 + 1024 * u;
            if (i < n) take(v[u], (int)i);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(idx, off);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[blockIdx.x] = idx == 0x7fffffff ? 0 : idx;
    }
}

}   // namespace

extern "C" int quipamd_decode_u_only(const quipamd_fop *U, const void *y, const void *bias, const void *residual, int64_t ld_residual,
                                     int relu, void *out, int64_t ld_out, int64_t bs, void *stream)
{
    QA_REQUIRE(U && bs >= 0, QUIPAMD_ERR_ARG, "decode_u_only: bad arguments");
    if (bs == 0) return QUIPAMD_OK;
    const int64_t n = (int64_t)U->p * U->q;
    QA_REQUIRE(U->F0 && U->F1 && U->store_idx && y && bias && out && ld_out >= n && ld_out % 4 == 0 && (!residual || (ld_residual >= n && ld_residual % 4 == 0)),
               QUIPAMD_ERR_ARG, "decode_u_only: null pointer / row strides");
    QA_REQUIRE(!residual || residual != out, QUIPAMD_ERR_ARG, "decode_u_only: out must not alias the residual");
    UOnlyArgs A;
    A.U = *U; A.y = (const uint16_t *)y; A.bias = (const uint16_t *)bias; A.res = (const uint16_t *)residual; A.out = (uint16_t *)out;
    A.ld_res = ld_residual; A.ld_out = ld_out; A.floor = relu ? 0.f : -INFINITY;
    hipStream_t s = (hipStream_t)stream;
    if (U->p == 64 && U->q == 32) return launch_u_only<64, 32>(A, bs, s);
    if (U->p == 64 && U->q == 64) return launch_u_only<64, 64>(A, bs, s);
    if (U->p == 128 && U->q == 64) return launch_u_only<128, 64>(A, bs, s);
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "decode_u_only: operator %d x %d", U->p, U->q);
}

extern "C" int quipamd_argmax_rows(const void *x, int dtype, int64_t rows, int64_t n, int64_t ld, int64_t *out, void *stream)
{
    QA_REQUIRE(rows >= 0 && n > 0 && n < 0x7fffffff && ld >= n, QUIPAMD_ERR_SHAPE, "argmax_rows: bad shape");
    if (rows == 0) return QUIPAMD_OK;
    QA_REQUIRE(x && out, QUIPAMD_ERR_ARG, "argmax_rows: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int vec = ((uintptr_t)x % 16 == 0 && ld % 8 == 0 && n >= 8) ? 1 : 0;                     // every row starts on a 16-byte boundary
    QA_DISPATCH_DTYPE(dtype, TI, (argmax_rows_kernel<TI><<<(unsigned)rows, 1024, 0, s>>>((const typename DT<TI>::storage *)x, n, ld, out, vec)));
    QA_LAUNCH_CHECK("quipamd_argmax_rows");
    return QUIPAMD_OK;
}

extern "C" void quipamd_decode_attention_config(int one_group_from, int three_heads_from)
{
    g_attn_u_one_group_from = one_group_from < 0 ? 0 : one_group_from;
    g_attn_u_three_heads_from = three_heads_from < 0 ? 257 : three_heads_from;       // (negative: the default)
}

extern "C" int quipamd_decode_attention_fused(const quipamd_fop *U, const void *const *y, const void *const *bias, void *kcache, void *vcache,
                                              const int64_t *pos, void *out, const float *cos_table, const float *sin_table,
                                              int64_t table_rows, int64_t bs, int heads, int hd, int64_t maxlen, float scale, int64_t ldo,
                                              void *stream)
{
    QA_REQUIRE(bs >= 0 && heads > 0 && maxlen > 0 && maxlen <= 32768, QUIPAMD_ERR_SHAPE, "decode_attention_fused: bad shape");
    if (bs == 0) return QUIPAMD_OK;
    QA_REQUIRE(U && y && bias && kcache && vcache && pos && out, QUIPAMD_ERR_ARG, "decode_attention_fused: null pointer");
    QA_REQUIRE((cos_table == nullptr) == (sin_table == nullptr) && (!cos_table || table_rows > 0), QUIPAMD_ERR_ARG, "decode_attention_fused: rotary tables");
    const int p = U[0].p, q = U[0].q;
    QA_REQUIRE((int64_t)p * q == (int64_t)heads * hd, QUIPAMD_ERR_SHAPE, "decode_attention_fused: operator %d x %d vs heads %d x %d", p, q, heads, hd);
    QA_REQUIRE(ldo >= (int64_t)heads * hd, QUIPAMD_ERR_SHAPE, "decode_attention_fused: out row stride");
    AttnUArgs A;
    for (int o = 0; o < 3; ++o) {
        QA_REQUIRE(U[o].F0 && U[o].F1 && U[o].store_idx && U[o].p == p && U[o].q == q && y[o] && bias[o], QUIPAMD_ERR_ARG,
                   "decode_attention_fused: operator / y / bias %d", o);
        A.U[o] = U[o]; A.y[o] = (const uint16_t *)y[o]; A.bias[o] = (const uint16_t *)bias[o];
    }
    A.kc = (uint16_t *)kcache; A.vc = (uint16_t *)vcache; A.out = (uint16_t *)out; A.pos = pos; A.cos_t = cos_table; A.sin_t = sin_table;
    A.table_rows = table_rows; A.maxlen = maxlen; A.ldo = ldo; A.heads = heads; A.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    if (hd == 64 && p == 64 && q == 32) return launch_attn_u<64, 64, 32>(A, bs, s);
    if (hd == 128 && p == 64 && q == 64) return launch_attn_u<128, 64, 64>(A, bs, s);
    if (hd == 128 && p == 64 && q == 32) return launch_attn_u<128, 64, 32>(A, bs, s);
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "decode_attention_fused: head_dim %d with a %d x %d operator (64 with 64 x 32; 128 with 64 x 64 or 64 x 32)", hd, p, q);
}

extern "C" int quipamd_decode_attention(const void *q, const void *k, const void *v, void *kcache, void *vcache, const int64_t *pos,
                                        void *out, int dtype, int64_t bs, int heads, int hd, int64_t maxlen, float scale,
                                        int64_t ld, void *stream)
{
    QA_REQUIRE(bs >= 0 && heads > 0 && maxlen > 0, QUIPAMD_ERR_SHAPE, "decode_attention: bad shape");
    if (bs == 0) return QUIPAMD_OK;
    QA_REQUIRE(q && k && v && kcache && vcache && pos && out, QUIPAMD_ERR_ARG, "decode_attention: null pointer");
    QA_REQUIRE(hd == 64 || hd == 128, QUIPAMD_ERR_UNSUPPORTED, "decode_attention: head_dim %d (64 or 128)", hd);
    QA_REQUIRE(dtype == QUIPAMD_F16 || dtype == QUIPAMD_BF16, QUIPAMD_ERR_UNSUPPORTED, "decode_attention: f16 / bf16 only");
    QA_REQUIRE(ld >= (int64_t)heads * hd && ld % 8 == 0, QUIPAMD_ERR_SHAPE, "decode_attention: row stride %lld", (long long)ld);
    QA_REQUIRE(maxlen <= 32768, QUIPAMD_ERR_SHAPE, "decode_attention: maxlen %lld > 32768", (long long)maxlen);
    QA_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)kcache | (uintptr_t)vcache) & 15) == 0, QUIPAMD_ERR_ARG,
               "decode_attention: pointers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == QUIPAMD_F16)
        return hd == 64 ? launch_attn<F16, 64>(q, k, v, kcache, vcache, pos, out, bs, heads, maxlen, scale, ld, s)
                        : launch_attn<F16, 128>(q, k, v, kcache, vcache, pos, out, bs, heads, maxlen, scale, ld, s);
    return hd == 64 ? launch_attn<BF16, 64>(q, k, v, kcache, vcache, pos, out, bs, heads, maxlen, scale, ld, s)
                    : launch_attn<BF16, 128>(q, k, v, kcache, vcache, pos, out, bs, heads, maxlen, scale, ld, s);
}
