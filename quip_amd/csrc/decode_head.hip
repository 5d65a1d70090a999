// decode_head.hip -- the two ends of a decode step around the decoder blocks (benchmark(), opt.py:431-482 / llama.py:418-471: one
// forward per token, `torch.argmax(out.logits[0, -1])` picks the next one), each as ONE launch:
//
//   head_kernel   t = U_last^T y + bias + residual        output side of the last block's last packed layer (decode_fused.hip's pass)
//                 h = LayerNorm | RMSNorm (t)             the model's final norm, rounded to fp16 like the module's output
//                 logits = W h                            lm_head, W fp16 [vocab, n] row-major (OPT: the tied embedding)
//                 (val, idx)[workgroup] = max / argmax over the workgroup's rows of the fp16 logits; *pos += 1
//   embed_kernel  id = argmax over the workgroups' partials (the previous token's head launch);  x = tok[id] + pos_table[pos + offset]
//
// Round 3 ran these as ~11 launches per token (u_only, LayerNorm, rocBLAS GEMV at 5.7 TB/s, a copy, argmax, two gathers, two adds,
// pos += 1: 59 us of 866 at OPT-1.3B).  The GEMV is a stream of 206 MB (OPT) / 262 MB (Llama) against a vector that fits LDS: every
// wave owns a contiguous run of rows, one row = n / 512 coalesced 1 KiB loads, v_dot2_f32_f16 against h from LDS, DPP wave sum;
// the workgroup's prologue (the operator pass + norm, ~2 us) is repeated in each of the 256 workgroups under the first weight loads.
#include "common.h"
#include "dq_common.h"
#include "fpass.h"

namespace {

constexpr int HD_MAXBS = 4;                // batch rows

struct HeadArgs {
    Fop U;
    const void *u_y;                      // f16 / f32 [bs, n], ZT order
    const uint16_t *u_bias, *u_res;       // f16 [n]; f16 [bs, ld_res] or null
    int64_t ld_res;
    const uint16_t *x;                    // !HAS_U: f16 [bs, ldx]
    int64_t ldx;
    const uint16_t *gamma, *beta;
    float eps;
    int bs;
    const uint16_t *W;                    // f16 [vocab, n]
    int64_t vocab;
    uint16_t *logits;                     // f16 [bs, ld_logits]
    int64_t ld_logits;
    float *part_val;                      // [bs, gridDim.x] or null
    int *part_idx;
    int64_t *pos_inc;                     // null, or a device counter this launch increments
};

template <int P, int Q, bool HAS_U, int NORM, bool YF32>
__global__ __launch_bounds__(1024) void head_kernel(HeadArgs G)
{
    typedef PassDims<P, Q> D;
    constexpr int N = D::N, NV = D::NV, NCV = (N / 8 + 1023) / 1024, NI = N / 512;      // NI: 16-byte chunks of a row per lane
    constexpr int HD_RB = 4096 / N;                                             // weight rows per batch (32 registers), double buffered in the loop
    static_assert(NV == 1 && NCV == 1, "n <= 4096");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *H = reinterpret_cast<uint16_t *>(smem);                           // [4][N] f16: the normalised rows
    char *pass = smem + (size_t)HD_MAXBS * N * 2;
    uint16_t *ZT = reinterpret_cast<uint16_t *>(pass), *Z1 = reinterpret_cast<uint16_t *>(pass + D::ZT_B);
    float *ZF = reinterpret_cast<float *>(pass + D::ZT_B + D::Z1_B);
    float *red = reinterpret_cast<float *>(pass + D::BYTES);                    // [32] norm statistics; then [4][16] + [4][16] partial maxima
    asm volatile("" ::"s"(G.U.F0), "s"(G.U.F1), "s"(G.U.store_idx), "s"(G.u_y), "s"(G.u_bias), "s"(G.u_res), "s"(G.ld_res), "s"(G.x), "s"(G.ldx),
                 "s"(G.gamma), "s"(G.beta), "s"(G.eps), "s"(G.bs), "s"(G.W), "s"(G.vocab), "s"(G.logits), "s"(G.ld_logits), "s"(G.part_val),
                 "s"(G.part_idx), "s"(G.pos_inc));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bs = G.bs;
    // this wave's rows: a contiguous run
    // (runs differ by at most one row: with ceil(vocab / waves) rows each, 50272 rows left 15 of 256 workgroups without any)
    const int64_t nwaves = (int64_t)gridDim.x * 16, wid = (int64_t)blockIdx.x * 16 + wave;
    const int64_t row_lo = wid * G.vocab / nwaves, row_hi = (wid + 1) * G.vocab / nwaves;

    // ---- requests of the prologue, then the first weight rows --------------------------------------------------------------------------
    uint4 yc = make_uint4(0u, 0u, 0u, 0u), yc2 = yc;
    auto load_u_row = [&](int b) {
        if (tid < N / 8) {
            if constexpr (YF32) {
                const float *src = reinterpret_cast<const float *>(G.u_y) + (int64_t)b * N + 8 * tid;
                yc = *reinterpret_cast<const uint4 *>(src);
                yc2 = *reinterpret_cast<const uint4 *>(src + 4);
            } else {
                yc = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(G.u_y) + (int64_t)b * N + 8 * tid);
            }
        }
    };
    auto u_row_f16 = [&]() {
        if constexpr (YF32) {
            auto f = [](uint32_t a) { return __builtin_bit_cast(float, a); };
            return make_uint4(pack_f16x2(f(yc.x), f(yc.y)), pack_f16x2(f(yc.z), f(yc.w)), pack_f16x2(f(yc2.x), f(yc2.y)), pack_f16x2(f(yc2.z), f(yc2.w)));
        } else {
            return yc;
        }
    };
    PassFrags<P, Q> fr;
    uint2 st = make_uint2(0u, 0u), bi = st, rs = st, gm = st, bt = st, xr = st;
    const bool own = tid < N / 4;                                                // this thread's 4 consecutive natural-order elements
    if (HAS_U) {
        load_u_row(0);
        load_f0<P, Q>(G.U, wave, lane, fr);
        load_f1<P, Q>(G.U, wave, lane, fr);
        if (own) {
            st = *reinterpret_cast<const uint2 *>(G.U.store_idx + 4 * tid);
            bi = *reinterpret_cast<const uint2 *>(G.u_bias + 4 * tid);
            if (G.u_res) rs = *reinterpret_cast<const uint2 *>(G.u_res + 4 * tid);
        }
    } else if (own) {
        xr = *reinterpret_cast<const uint2 *>(G.x + 4 * tid);
    }
    if (own) {
        gm = *reinterpret_cast<const uint2 *>(G.gamma + 4 * tid);
        if (NORM == 1) bt = *reinterpret_cast<const uint2 *>(G.beta + 4 * tid);
    }
    uint4 w[HD_RB][NI], wn[HD_RB][NI];
    auto load_rows = [&](uint4 (&dst)[HD_RB][NI], int64_t r0) {
#pragma unroll
        for (int rr = 0; rr < HD_RB; ++rr) {
            const int64_t row = r0 + rr;
            if (row < row_hi) {                                                  // wave-uniform: rows past the run are not fetched (they are another wave's)
                const uint4 *src = reinterpret_cast<const uint4 *>(G.W + row * N);
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(src + lane + 64 * i));
                    dst[rr][i] = make_uint4(t[0], t[1], t[2], t[3]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) dst[rr][i] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    };
    load_rows(w, row_lo);

    // ---- prologue: t = U^T y + bias + residual -> norm -> H -----------------------------------------------------------------------------
    for (int b = 0; b < bs; ++b) {
        float4 tv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (HAS_U) {
            if (b > 0) {
                load_u_row(b);
                if (own && G.u_res) rs = *reinterpret_cast<const uint2 *>((G.u_res + (int64_t)b * G.ld_res) + 4 * tid);
            }
            if (tid < N / 8) copy_chunk_zt<P, Q>(ZT, u_row_f16(), tid);
            __syncthreads();
            mix_stages<P, Q>(ZT, Z1, ZF, fr, wave, lane);
            __syncthreads();
            if (own) {
                float4 t = gather4<P, Q>(ZF, st);
                const float4 rr = f16x4_to_f32(rs), bb4 = f16x4_to_f32(bi);
                uint2 pk;                                                        // the residual stream is fp16
                pk.x = pack_f16x2(t.x + bb4.x + rr.x, t.y + bb4.y + rr.y);
                pk.y = pack_f16x2(t.z + bb4.z + rr.z, t.w + bb4.w + rr.w);
                tv = f16x4_to_f32(pk);
            }
        } else {
            if (b > 0 && own) xr = *reinterpret_cast<const uint2 *>((G.x + (int64_t)b * G.ldx) + 4 * tid);
            tv = f16x4_to_f32(xr);
        }
        // statistics as in decode_fused.hip: per-wave (mean, M2) merged with Chan's formula (equal counts); RMSNorm: sum of squares
        constexpr int NWD = N / 4 / 64, CNT = N / NWD;
        float mean = 0.f, rstd;
        if (NORM == 1) {
            const float mw = fg_wave_sum((tv.x + tv.y) + (tv.z + tv.w)) * (1.0f / (float)CNT);
            const float d0 = tv.x - mw, d1 = tv.y - mw, d2 = tv.z - mw, d3 = tv.w - mw;
            const float m2w = fg_wave_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
            if (lane == 0 && wave < NWD) {
                red[wave] = mw;
                red[16 + wave] = m2w;
            }
            __syncthreads();
            float ms = 0.f, m2 = 0.f;
#pragma unroll
            for (int i = 0; i < NWD; ++i) ms += red[i];
            mean = ms * (1.0f / (float)NWD);
#pragma unroll
            for (int i = 0; i < NWD; ++i) {
                const float dm = red[i] - mean;
                m2 += red[16 + i] + (float)CNT * dm * dm;
            }
            rstd = rsqrtf(m2 * (1.0f / (float)N) + G.eps);
        } else {
            const float w2 = fg_wave_sum((tv.x * tv.x + tv.y * tv.y) + (tv.z * tv.z + tv.w * tv.w));
            if (lane == 0 && wave < NWD) red[wave] = w2;
            __syncthreads();
            float t2 = 0.f;
#pragma unroll
            for (int i = 0; i < NWD; ++i) t2 += red[i];
            rstd = rsqrtf(t2 * (1.0f / (float)N) + G.eps);
        }
        if (own) {
            const float4 gmf = f16x4_to_f32(gm), btf = f16x4_to_f32(bt);
            uint2 pk;
            pk.x = pack_f16x2((tv.x - mean) * rstd * gmf.x + btf.x, (tv.y - mean) * rstd * gmf.y + btf.y);
            pk.y = pack_f16x2((tv.z - mean) * rstd * gmf.z + btf.z, (tv.w - mean) * rstd * gmf.w + btf.w);
            *reinterpret_cast<uint2 *>(H + (size_t)b * N + 4 * tid) = pk;
        }
        __syncthreads();                                                        // H row complete; ZT / red free for the next row
    }

    // ---- the GEMV over this wave's rows ---------------------------------------------------------------------------------------------------
    float best[HD_MAXBS];
    int bidx[HD_MAXBS];
#pragma unroll
    for (int b = 0; b < HD_MAXBS; ++b) { best[b] = -INFINITY; bidx[b] = 0x7fffffff; }
    for (int64_t r0 = row_lo; r0 < row_hi; r0 += HD_RB) {
        load_rows(wn, r0 + HD_RB);                                               // the next batch travels while this one is multiplied
        float keep[HD_MAXBS] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < HD_MAXBS; ++b) {
            if (b < bs) {
                uint4 hv[NI];
#pragma unroll
                for (int i = 0; i < NI; ++i) hv[i] = *reinterpret_cast<const uint4 *>(H + (size_t)b * N + 8 * (lane + 64 * i));
#pragma unroll
                for (int rr = 0; rr < HD_RB; ++rr) {
                    float a0 = 0.f, a1 = 0.f;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        a0 = ActF16::dot2(w[rr][i].x, hv[i].x, a0);
                        a1 = ActF16::dot2(w[rr][i].y, hv[i].y, a1);
                        a0 = ActF16::dot2(w[rr][i].z, hv[i].z, a0);
                        a1 = ActF16::dot2(w[rr][i].w, hv[i].w, a1);
                    }
                    const float s = f16_bits_to_f32(f32_to_f16_bits(fg_wave_sum(a0 + a1)));    // the logit as the fp16 value torch would hold
                    keep[b] = lane == rr ? s : keep[b];
                    if (r0 + rr < row_hi && s > best[b]) { best[b] = s; bidx[b] = (int)(r0 + rr); }   // rows ascend: ties keep the smallest index
                }
            }
        }
        const int64_t mine = r0 + lane;
#pragma unroll
        for (int b = 0; b < HD_MAXBS; ++b)
            if (b < bs && lane < HD_RB && mine < row_hi) G.logits[(int64_t)b * G.ld_logits + mine] = f32_to_f16_bits(keep[b]);
#pragma unroll
        for (int rr = 0; rr < HD_RB; ++rr)
#pragma unroll
            for (int i = 0; i < NI; ++i) w[rr][i] = wn[rr][i];
    }
    if (G.part_val) {
        float *pv = red + 32;
        int *pi = reinterpret_cast<int *>(red + 32 + 64);
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < HD_MAXBS; ++b) { pv[b * 16 + wave] = best[b]; pi[b * 16 + wave] = bidx[b]; }
        }
        __syncthreads();
        if (tid < bs) {
            float bv = -INFINITY;
            int bi2 = 0x7fffffff;
            for (int wv = 0; wv < 16; ++wv) {
                const float v = pv[tid * 16 + wv];
                if (v > bv) { bv = v; bi2 = pi[tid * 16 + wv]; }
            }
            G.part_val[(int64_t)tid * gridDim.x + blockIdx.x] = bv;
            G.part_idx[(int64_t)tid * gridDim.x + blockIdx.x] = bi2;
        }
    }
    if (G.pos_inc && blockIdx.x == 0 && tid == 0) *G.pos_inc += 1;
}

struct EmbedArgs {
    const uint16_t *tok, *pos_table;      // f16 [vocab, n]; f16 [positions, n] or null
    const int64_t *pos;
    int64_t pos_offset, positions, vocab;
    int64_t *ids;                         // [bs] in / out
    const float *part_val;                // [bs, npart] or null
    const int *part_idx;
    int npart, n;
    uint16_t *out;                        // f16 [bs, ld_out]
    int64_t ld_out;
};

// one workgroup (256 threads) per batch row
__global__ __launch_bounds__(256) void embed_kernel(EmbedArgs G)
{
    __shared__ float sv[4];
    __shared__ int si[4];
    __shared__ int64_t sid;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bool have = false;
    if (G.part_val) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < G.npart; i += 256) {                               // ascending workgroups = ascending rows
            const float v = G.part_val[(int64_t)b * G.npart + i];
            const int ix = G.part_idx[(int64_t)b * G.npart + i];
            if (ix >= 0 && ix != 0x7fffffff && (v > bv || (v == bv && ix < bi))) { bv = v; bi = ix; }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int wv = 1; wv < 4; ++wv)
                if (sv[wv] > bv || (sv[wv] == bv && si[wv] < bi)) { bv = sv[wv]; bi = si[wv]; }
            have = bi != 0x7fffffff;                                             // no valid partial (first token): keep the caller's id
            sid = have ? (int64_t)bi : G.ids[b];
            if (have) G.ids[b] = (int64_t)bi;
        }
    } else if (tid == 0) {
        sid = G.ids[b];
    }
    __syncthreads();
    int64_t id = sid;
    id = id < 0 ? 0 : id >= G.vocab ? G.vocab - 1 : id;
    int64_t pr = 0;
    if (G.pos_table) {
        pr = *G.pos + G.pos_offset;
        pr = pr < 0 ? 0 : pr >= G.positions ? G.positions - 1 : pr;
    }
    for (int c = tid; c < G.n / 8; c += 256) {
        const uint4 t = *reinterpret_cast<const uint4 *>(G.tok + id * G.n + 8 * c);
        uint4 o = t;
        if (G.pos_table) {
            const uint4 q = *reinterpret_cast<const uint4 *>(G.pos_table + pr * G.n + 8 * c);
            auto add2 = [](uint32_t a, uint32_t b2) {
                return pack_f16x2(f16_bits_to_f32(a & 0xffff) + f16_bits_to_f32(b2 & 0xffff), f16_bits_to_f32(a >> 16) + f16_bits_to_f32(b2 >> 16));
            };
            o = make_uint4(add2(t.x, q.x), add2(t.y, q.y), add2(t.z, q.z), add2(t.w, q.w));
        }
        *reinterpret_cast<uint4 *>(G.out + (int64_t)b * G.ld_out + 8 * c) = o;
    }
}

template <int P, int Q, bool HAS_U, int NORM, bool YF32> int launch_head(const HeadArgs &A, int nwg, hipStream_t s)
{
    typedef PassDims<P, Q> D;
    const size_t lds = (size_t)HD_MAXBS * D::N * 2 + D::BYTES + (32 + 64 + 64) * 4;
    auto kern = head_kernel<P, Q, HAS_U, NORM, YF32>;
    static QaPerDevice attr;
    const int d = attr.dev();
    if (d < 0 || !attr.done[d]) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "decode_head: cannot raise dynamic LDS to %zu", lds);
        if (d >= 0) attr.done[d] = true;
    }
    kern<<<(unsigned)nwg, 1024, lds, s>>>(A);
    QA_LAUNCH_CHECK("quipamd_decode_head");
    return QUIPAMD_OK;
}

template <int P, int Q> int dispatch_head(const HeadArgs &A, bool u, int norm, bool yf32, int nwg, hipStream_t s)
{
    if (u) {
        if (yf32) return norm == 1 ? launch_head<P, Q, true, 1, true>(A, nwg, s) : launch_head<P, Q, true, 2, true>(A, nwg, s);
        return norm == 1 ? launch_head<P, Q, true, 1, false>(A, nwg, s) : launch_head<P, Q, true, 2, false>(A, nwg, s);
    }
    return norm == 1 ? launch_head<P, Q, false, 1, false>(A, nwg, s) : launch_head<P, Q, false, 2, false>(A, nwg, s);
}

}   // namespace

extern "C" int quipamd_decode_head(const quipamd_head_args *a, void *stream)
{
    QA_REQUIRE(a, QUIPAMD_ERR_ARG, "decode_head: null args");
    QA_REQUIRE(a->bs >= 1 && a->bs <= HD_MAXBS, QUIPAMD_ERR_SHAPE, "decode_head: 1..%d rows", HD_MAXBS);
    QA_REQUIRE(a->n == 2048 || a->n == 4096, QUIPAMD_ERR_UNSUPPORTED, "decode_head: n = %lld (2048 = 64 x 32 or 4096 = 64 x 64)", (long long)a->n);
    QA_REQUIRE((a->norm == 1 || a->norm == 2) && a->ln_gamma && (a->norm != 1 || a->ln_beta), QUIPAMD_ERR_ARG,
               "decode_head: norm 1 (LayerNorm: gamma, beta) or 2 (RMSNorm: gamma)");
    QA_REQUIRE(a->W && a->vocab > 0 && a->logits && a->ld_logits >= a->vocab, QUIPAMD_ERR_ARG, "decode_head: W [vocab, n], logits [bs, ld >= vocab]");
    QA_REQUIRE((a->part_val == nullptr) == (a->part_idx == nullptr) && (!a->part_val || a->nparts >= 1), QUIPAMD_ERR_ARG,
               "decode_head: part_val and part_idx come together, nparts = the number of workgroups");
    const int p = 64, q = a->n == 2048 ? 32 : 64;
    HeadArgs A;
    A.U = a->U;
    A.u_y = a->u_y; A.u_bias = (const uint16_t *)a->u_bias; A.u_res = (const uint16_t *)a->u_residual; A.ld_res = a->ld_residual;
    A.x = (const uint16_t *)a->x; A.ldx = a->ldx;
    A.gamma = (const uint16_t *)a->ln_gamma; A.beta = (const uint16_t *)a->ln_beta; A.eps = a->ln_eps;
    A.bs = (int)a->bs;
    A.W = (const uint16_t *)a->W; A.vocab = a->vocab; A.logits = (uint16_t *)a->logits; A.ld_logits = a->ld_logits;
    A.part_val = a->part_val; A.part_idx = a->part_idx; A.pos_inc = a->pos_inc;
    const bool u = a->has_u != 0, yf32 = u && a->u_y_dtype == QUIPAMD_F32;
    if (u) {
        QA_REQUIRE(a->U.F0 && a->U.F1 && a->U.store_idx && a->U.p == p && a->U.q == q && a->u_y && a->u_bias, QUIPAMD_ERR_ARG,
                   "decode_head: the output-side operator must be %d x %d, with u_y and u_bias (zeros where the layer has none)", p, q);
        QA_REQUIRE(yf32 || a->u_y_dtype == QUIPAMD_F16, QUIPAMD_ERR_ARG, "decode_head: u_y_dtype f16 or f32");
        QA_REQUIRE(!a->u_residual || (a->ld_residual >= a->n && a->ld_residual % 4 == 0), QUIPAMD_ERR_SHAPE, "decode_head: residual row stride");
    } else {
        QA_REQUIRE(a->x && a->ldx >= a->n && a->ldx % 4 == 0, QUIPAMD_ERR_ARG, "decode_head: x [bs, ldx] needed without an output-side operator");
    }
    int nwg = a->nparts > 0 ? a->nparts : 256;
    QA_REQUIRE(nwg >= 1 && nwg <= 4096, QUIPAMD_ERR_ARG, "decode_head: 1..4096 workgroups");
    hipStream_t s = (hipStream_t)stream;
    return q == 32 ? dispatch_head<64, 32>(A, u, a->norm, yf32, nwg, s) : dispatch_head<64, 64>(A, u, a->norm, yf32, nwg, s);
}

extern "C" int quipamd_decode_embed(const void *tok_table, int64_t vocab, const void *pos_table, int64_t positions, int64_t pos_offset,
                                    const int64_t *pos, int64_t *ids, const float *part_val, const int *part_idx, int nparts, int64_t n,
                                    void *out, int64_t ld_out, int64_t bs, void *stream)
{
    QA_REQUIRE(tok_table && ids && out && vocab > 0 && n > 0 && n % 8 == 0 && ld_out >= n && ld_out % 8 == 0, QUIPAMD_ERR_ARG,
               "decode_embed: tok_table [vocab, n], ids, out [bs, ld_out]; n %% 8 == 0");
    QA_REQUIRE(!pos_table || (pos && positions > 0), QUIPAMD_ERR_ARG, "decode_embed: a position table needs pos and its row count");
    QA_REQUIRE((part_val == nullptr) == (part_idx == nullptr) && (!part_val || nparts >= 1), QUIPAMD_ERR_ARG, "decode_embed: part_val and part_idx come together");
    QA_REQUIRE(bs >= 0 && bs <= 65535, QUIPAMD_ERR_SHAPE, "decode_embed: bad row count");
    if (bs == 0) return QUIPAMD_OK;
    EmbedArgs A{(const uint16_t *)tok_table, (const uint16_t *)pos_table, pos, pos_offset, positions, vocab, ids, part_val, part_idx, nparts, (int)n,
                (uint16_t *)out, ld_out};
    embed_kernel<<<(unsigned)bs, 256, 0, (hipStream_t)stream>>>(A);
    QA_LAUNCH_CHECK("quipamd_decode_embed");
    return QUIPAMD_OK;
}
