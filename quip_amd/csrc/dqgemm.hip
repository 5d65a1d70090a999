// dqgemm.hip -- K2: fused dequant-GEMM  y[b,r] = bias[r] + sum_k What[r,k] x[b,k]
//
// Takes over quant_cuda.vecquant{3,4}matmul (quant.py:229, zeroShot/models/quant.py:207), for 2- and
// 4-bit codes, qfn a (per-row scale/zero, quant.py:8) and qfn b (scalar scale, quant.py:13-14), any bs.
//
// Design (gfx950, wave64, v_mfma_f32_16x16x32_bf16):
//   * The packed weights are the MFMA **A** operand (16 weight rows x 32 k), the activations the **B**
//     operand (32 k x 16 batch columns): D[row][batch].  The STREAM layout stores each 16-row x KC-column
//     tile as 64 lanes x 16 B in exactly the A-fragment order, so one coalesced global_load_dwordx4 per lane
//     (1 KiB per wave) feeds NT = KC/32 MFMAs with no LDS round trip (cdna_hip_programming.md: "M <= 16
//     decode weights: load straight to VGPRs").
//   * In-register dequant, 2 VALU per bf16 pair: the code is shifted onto the TOP mantissa bits of a bf16
//     with a fixed exponent:  2 bit -> 0x4080 | c<<5 = 4 + c,  4 bit -> 0x4180 | c<<3 = 16 + c   (exact),
//     pair = ((w >> s) & MASK) | BASE  (v_lshrrev/v_lshlrev + v_and_or_b32).
//   * The constant offset and the affine grid are folded into the epilogue:
//         sum_k (OFF + q) x = acc   =>   sum_k q x = acc - OFF * xsum,   xsum[b] = sum_k x[b,k]
//         qfn b:  y = (2 s / maxq) * (acc - (OFF + maxq/2) * xsum)
//         qfn a:  y = scale[r]     * (acc - (OFF + zero[r]) * xsum)
//     xsum is accumulated beside the MFMAs with v_dot2_f32_bf16 against (1,1).
//   * D layout (col = lane&15 = batch, row = 4*(lane>>4)+reg = weight row): a lane owns 4 consecutive
//     output features of one batch row -> one 8-byte (bf16) / 16-byte (fp32) store.
//
// Kernel `dqgemm_stream`: one workgroup = RT row tiles x KW k-slices (RT*KW waves); every wave streams its
// own weight tiles and reads its x fragments straight from global/L2; K-slices are reduced through LDS.
// Algorithmic bytes per call: m*d*bits/8 + 2*bs*d + (2|4)*bs*m.  FLOPs: 2*bs*m*d.
#include "common.h"

namespace {

template <int BITS> struct Deq;
template <> struct Deq<2> {
    static constexpr int KC = 256, NT = 8;
    static constexpr float OFF = 4.0f;
    // A fragment (4 dwords = 8 bf16) of MFMA step t from the lane's 4 packed dwords
    static __device__ __forceinline__ uint4 frag(const uint4 &w, int t)
    {
        const uint32_t src = (t >> 1) == 0 ? w.x : (t >> 1) == 1 ? w.y : (t >> 1) == 2 ? w.z : w.w;
        uint32_t o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int sh = 2 * (4 * (t & 1) + v) - 5;
            const uint32_t shifted = sh >= 0 ? (src >> sh) : (src << (-sh));
            o[v] = (shifted & 0x00600060u) | 0x40804080u;
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};
template <> struct Deq<4> {
    static constexpr int KC = 128, NT = 4;
    static constexpr float OFF = 16.0f;
    static __device__ __forceinline__ uint4 frag(const uint4 &w, int t)
    {
        const uint32_t src = t == 0 ? w.x : t == 1 ? w.y : t == 2 ? w.z : w.w;
        uint32_t o[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int sh = 4 * v - 3;
            const uint32_t shifted = sh >= 0 ? (src >> sh) : (src << (-sh));
            o[v] = (shifted & 0x00780078u) | 0x41804180u;
        }
        return make_uint4(o[0], o[1], o[2], o[3]);
    }
};

union Frag {
    uint4 u;
    bf16x8_t v;
};
union Pair {
    uint32_t u;
    bf16x2_t v;
};

__device__ __forceinline__ float dot_ones(const uint4 &x, float acc)
{
    Pair one, p;
    one.u = 0x3f803f80u;
    p.u = x.x; acc = __builtin_amdgcn_fdot2_f32_bf16(p.v, one.v, acc, false);
    p.u = x.y; acc = __builtin_amdgcn_fdot2_f32_bf16(p.v, one.v, acc, false);
    p.u = x.z; acc = __builtin_amdgcn_fdot2_f32_bf16(p.v, one.v, acc, false);
    p.u = x.w; acc = __builtin_amdgcn_fdot2_f32_bf16(p.v, one.v, acc, false);
    return acc;
}

struct EpiArgs {
    const float *scale;   // [1] (qfn b) or [m] (qfn a)
    const float *zero;    // [m] or null
    const float *bias;    // [m] or null
    void *y;
    int qfn, maxq, y_f32, accumulate;
    int64_t bs, m;
};

__device__ __forceinline__ void epilogue_store(const EpiArgs &e, const f32x4_t &acc, float xs, float off,
                                               int64_t b, int64_t r0)
{
    if (b >= e.bs) return;
    float out[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t r = r0 + i;
        float alpha, c0;
        if (e.qfn == QUIPAMD_QFN_B) {
            alpha = 2.0f * e.scale[0] / (float)e.maxq;
            c0 = off + 0.5f * (float)e.maxq;
        } else {
            alpha = e.scale[r];
            c0 = off + e.zero[r];
        }
        float v = alpha * (acc[i] - c0 * xs);
        if (e.bias) v += e.bias[r];
        out[i] = v;
    }
    if (e.y_f32) {
        float4 *dst = reinterpret_cast<float4 *>((float *)e.y + b * e.m + r0);
        if (e.accumulate) {
            const float4 old = *dst;
            out[0] += old.x; out[1] += old.y; out[2] += old.z; out[3] += old.w;
        }
        *dst = make_float4(out[0], out[1], out[2], out[3]);
    } else {
        uint2 pk;
        pk.x = (uint32_t)f32_to_bf16_bits(out[0]) | ((uint32_t)f32_to_bf16_bits(out[1]) << 16);
        pk.y = (uint32_t)f32_to_bf16_bits(out[2]) | ((uint32_t)f32_to_bf16_bits(out[3]) << 16);
        *reinterpret_cast<uint2 *>((uint16_t *)e.y + b * e.m + r0) = pk;
    }
}

// RT row tiles (16 rows each) x KW k-slices per workgroup; grid = (m/16/RT, ceil(bs/16)).
template <int BITS, int RT, int KW>
__global__ __launch_bounds__(64 * RT * KW) void dqgemm_stream(const uint16_t *__restrict__ x,
                                                              const uint4 *__restrict__ qw, EpiArgs e, int64_t d)
{
    typedef Deq<BITS> Q;
    constexpr int KC = Q::KC, NT = Q::NT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ks = wave % KW, rtl = wave / KW;
    const int j = lane & 15, g = lane >> 4;
    const int64_t nkc = d / KC;
    const int64_t rt = (int64_t)blockIdx.x * RT + rtl;
    const int64_t b = (int64_t)blockIdx.y * 16 + j;
    const bool bvalid = b < e.bs;
    const uint16_t *xrow = x + (bvalid ? b : 0) * d + 8 * g;
    const uint4 *wt = qw + (rt * nkc) * 64 + lane;

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float xs = 0.f;
    for (int64_t kc = ks; kc < nkc; kc += KW) {
        const uint4 w = wt[kc * 64];
        uint4 xf[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            xf[t] = *reinterpret_cast<const uint4 *>(xrow + kc * KC + 32 * t);
            if (!bvalid) xf[t] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Frag a, bb;
            a.u = Q::frag(w, t);
            bb.u = xf[t];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bb.v, acc, 0, 0, 0);
            xs = dot_ones(xf[t], xs);
        }
    }
    // the 4 lane groups hold disjoint k's of the same batch row: fold them so every lane has xsum[b=j]
    xs += __shfl_xor(xs, 16);
    xs += __shfl_xor(xs, 32);

    if constexpr (KW > 1) {
        __shared__ float red[RT][KW][5][64];
        red[rtl][ks][0][lane] = acc[0];
        red[rtl][ks][1][lane] = acc[1];
        red[rtl][ks][2][lane] = acc[2];
        red[rtl][ks][3][lane] = acc[3];
        red[rtl][ks][4][lane] = xs;
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int s = 1; s < KW; ++s) {
            acc[0] += red[rtl][s][0][lane];
            acc[1] += red[rtl][s][1][lane];
            acc[2] += red[rtl][s][2][lane];
            acc[3] += red[rtl][s][3][lane];
            xs += red[rtl][s][4][lane];
        }
    }
    epilogue_store(e, acc, xs, Q::OFF, b, rt * 16 + 4 * g);
}

template <int BITS>
int launch(const uint16_t *x, const uint4 *qw, const EpiArgs &e, int64_t d, hipStream_t s)
{
    const int64_t ntile = e.m / 16;
    const int nb = qa_div_up(e.bs, 16);
    QA_REQUIRE(nb <= 65535, QUIPAMD_ERR_SHAPE, "dequant_gemm: bs too large for this kernel (%lld)", (long long)e.bs);
    // occupancy heuristic: enough workgroups for 256 CUs first, then more k-slices per workgroup
    if (ntile * nb >= 512 && ntile % 4 == 0) {
        dqgemm_stream<BITS, 4, 1><<<dim3((unsigned)(ntile / 4), nb), 256, 0, s>>>(x, qw, e, d);
    } else {
        dqgemm_stream<BITS, 1, 4><<<dim3((unsigned)ntile, nb), 256, 0, s>>>(x, qw, e, d);
    }
    QA_LAUNCH_CHECK("quipamd_dequant_gemm");
    return QUIPAMD_OK;
}

}   // namespace

extern "C" int quipamd_dequant_gemm(const void *x, int x_dtype, const int32_t *qweight, int bits, int layout, int qfn,
                                    const float *scale, const float *zero, const float *bias, void *y, int y_dtype,
                                    int accumulate, int64_t bs, int64_t m, int64_t d, void *stream)
{
    QA_REQUIRE(x && qweight && scale && y, QUIPAMD_ERR_ARG, "dequant_gemm: null pointer");
    QA_REQUIRE(x_dtype == QUIPAMD_BF16, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: x must be bf16");
    QA_REQUIRE(y_dtype == QUIPAMD_BF16 || y_dtype == QUIPAMD_F32, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: y must be bf16 or f32");
    QA_REQUIRE(!accumulate || y_dtype == QUIPAMD_F32, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: accumulate needs f32 y");
    QA_REQUIRE(layout == QUIPAMD_LAYOUT_STREAM, QUIPAMD_ERR_UNSUPPORTED,
               "dequant_gemm: qweight must be in STREAM layout (repack with quipamd_unpack/quipamd_pack)");
    QA_REQUIRE(bits == 2 || bits == 4, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: bits must be 2 or 4");
    QA_REQUIRE(qfn == QUIPAMD_QFN_B || (qfn == QUIPAMD_QFN_A && zero), QUIPAMD_ERR_ARG, "dequant_gemm: qfn a needs zero; qfn must be a or b");
    QA_REQUIRE(m % 16 == 0 && d % (512 / bits) == 0, QUIPAMD_ERR_SHAPE,
               "dequant_gemm: needs m %% 16 == 0 and d %% %d == 0 (m=%lld d=%lld)", 512 / bits, (long long)m, (long long)d);
    if (bs == 0 || m == 0) return QUIPAMD_OK;
    EpiArgs e;
    e.scale = scale; e.zero = zero; e.bias = bias; e.y = y;
    e.qfn = qfn; e.maxq = (1 << bits) - 1; e.y_f32 = (y_dtype == QUIPAMD_F32); e.accumulate = accumulate;
    e.bs = bs; e.m = m;
    hipStream_t s = (hipStream_t)stream;
    if (bits == 2) return launch<2>((const uint16_t *)x, (const uint4 *)qweight, e, d, s);
    return launch<4>((const uint16_t *)x, (const uint4 *)qweight, e, d, s);
}
