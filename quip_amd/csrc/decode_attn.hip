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

namespace {

// wave-wide reductions on the DPP network (4 DPP operands + 4 v_readlane instead of six ds_bpermute round trips each)
template <int CTRL> __device__ __forceinline__ float dpp_f(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, true));
}
template <bool MAX> __device__ __forceinline__ float wave_reduce(float v)
{
    auto op = [](float a, float b) { return MAX ? fmaxf(a, b) : a + b; };
    v = op(v, dpp_f<0xB1>(v));       // quad_perm [1,0,3,2]
    v = op(v, dpp_f<0x4E>(v));       // quad_perm [2,3,0,1]
    v = op(v, dpp_f<0x141>(v));      // row_half_mirror
    v = op(v, dpp_f<0x140>(v));      // row_mirror: every lane of a 16-lane row holds the row's result
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return op(op(r0, r1), op(r2, r3));
}

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
    const int64_t pos = *pos_p, T = pos + 1;
    const S *qh = q + (int64_t)b * ldq + head * HD, *kh = k + (int64_t)b * ldq + head * HD, *vh = v + (int64_t)b * ldq + head * HD;
    S *kcb = kc + ((int64_t)b * heads + head) * maxlen * HD, *vcb = vc + ((int64_t)b * heads + head) * maxlen * HD;

    if (pos < 0 || pos >= maxlen) return;                           // uniform; a full cache is the caller's error
    // append this token; the barrier (workgroup-scope fence) makes it visible to the reads below
    if (tid < HD) kcb[pos * HD + tid] = kh[tid];
    else if (tid < 2 * HD) vcb[pos * HD + tid - HD] = vh[tid - HD];
    __syncthreads();

    float qr[HD];
#pragma unroll
    for (int e8 = 0; e8 < HD; e8 += 8) {                            // 16-byte loads (HD scalar 2-byte loads per thread before)
        S raw[8];
        *reinterpret_cast<uint4 *>(raw) = *reinterpret_cast<const uint4 *>(qh + e8);
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
    for (int64_t t = tid; t < T; t += 256) {
        const S *row = kcb + t * HD;
        float acc = 0.f;
#pragma unroll
        for (int e8 = 0; e8 < HD; e8 += 8) {
            S raw[8];
            *reinterpret_cast<uint4 *>(raw) = *reinterpret_cast<const uint4 *>(row + e8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc = fmaf(qr[e8 + e], DT<TI>::load(raw, e), acc);
        }
        scores[t] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_reduce<true>(mx);
    if (lane == 0) red[wave] = mx;
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
    __syncthreads();
    if (tid < HD) {
        const float r = (part[tid] + part[HD + tid]) + (part[2 * HD + tid] + part[3 * HD + tid]);
        DT<TI>::store(out + (int64_t)b * ldq + head * HD, tid, r * inv);
    }
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
