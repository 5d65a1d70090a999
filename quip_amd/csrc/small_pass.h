// small_pass.h -- the single-workgroup Kronecker operator application shared by ortho_small.hip (K3, activation side)
// and dqgemm_vop.hip (operator fused into the dequant-GEMM prologue): helpers, LDS sizing, and the split-bf16 pass.
#pragma once
#include "common.h"
#include <type_traits>

namespace {

typedef quipamd_small_op SmallArgs;        // include/quip_amd.h

struct SmallBatch {
    SmallArgs op[QUIPAMD_SMALL_MAX_OPS];        // blockIdx.y selects the op; all ops share p, q and the dtypes
    int64_t rows;                               // rows per op (workgroups stride over them)
};

__device__ __forceinline__ float load_any(const void *p, int dt, int64_t i)
{
    return dt == QUIPAMD_F32 ? ((const float *)p)[i] : dt == QUIPAMD_F16 ? f16_bits_to_f32(((const uint16_t *)p)[i])
                                                                         : bf16_bits_to_f32(((const uint16_t *)p)[i]);
}

// 4 consecutive elements at element index i (multiple of 4) as fp32
template <class T> __device__ __forceinline__ float4 load4(const void *p, int64_t i);
template <> __device__ __forceinline__ float4 load4<F32>(const void *p, int64_t i) { return *reinterpret_cast<const float4 *>((const float *)p + i); }
template <> __device__ __forceinline__ float4 load4<F16>(const void *p, int64_t i)
{
    const uint2 u = *reinterpret_cast<const uint2 *>((const uint16_t *)p + i);
    return make_float4(f16_bits_to_f32(u.x & 0xffff), f16_bits_to_f32(u.x >> 16), f16_bits_to_f32(u.y & 0xffff), f16_bits_to_f32(u.y >> 16));
}
template <> __device__ __forceinline__ float4 load4<BF16>(const void *p, int64_t i)
{
    const uint2 u = *reinterpret_cast<const uint2 *>((const uint16_t *)p + i);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ float4 load4_any(const void *p, int dt, int64_t i)
{
    return dt == QUIPAMD_F32 ? load4<F32>(p, i) : dt == QUIPAMD_F16 ? load4<F16>(p, i) : load4<BF16>(p, i);
}
template <class T> __device__ __forceinline__ void store4(void *p, int64_t i, const float4 &v);
template <> __device__ __forceinline__ void store4<F32>(void *p, int64_t i, const float4 &v) { *reinterpret_cast<float4 *>((float *)p + i) = v; }
template <> __device__ __forceinline__ void store4<F16>(void *p, int64_t i, const float4 &v)
{
    uint2 u;
    u.x = (uint32_t)f32_to_f16_bits(v.x) | ((uint32_t)f32_to_f16_bits(v.y) << 16);
    u.y = (uint32_t)f32_to_f16_bits(v.z) | ((uint32_t)f32_to_f16_bits(v.w) << 16);
    *reinterpret_cast<uint2 *>((uint16_t *)p + i) = u;
}
template <> __device__ __forceinline__ void store4<BF16>(void *p, int64_t i, const float4 &v)
{
    uint2 u;
    u.x = (uint32_t)f32_to_bf16_bits(v.x) | ((uint32_t)f32_to_bf16_bits(v.y) << 16);
    u.y = (uint32_t)f32_to_bf16_bits(v.z) | ((uint32_t)f32_to_bf16_bits(v.w) << 16);
    *reinterpret_cast<uint2 *>((uint16_t *)p + i) = u;
}
__device__ __forceinline__ void store4_any(void *p, int dt, int64_t i, const float4 &v)
{
    if (dt == QUIPAMD_F32) store4<F32>(p, i, v);
    else if (dt == QUIPAMD_F16) store4<F16>(p, i, v);
    else store4<BF16>(p, i, v);
}
// the value a reader of the stored element would see
__device__ __forceinline__ float4 round4_any(int dt, const float4 &v)
{
    if (dt == QUIPAMD_F16) return make_float4(DT<F16>::rnd(v.x), DT<F16>::rnd(v.y), DT<F16>::rnd(v.z), DT<F16>::rnd(v.w));
    if (dt == QUIPAMD_BF16) return make_float4(DT<BF16>::rnd(v.x), DT<BF16>::rnd(v.y), DT<BF16>::rnd(v.z), DT<BF16>::rnd(v.w));
    return v;
}

// ---- loads that do not force a wait ---------------------------------------------------------------------------------------------
// load4_any converts inside the dtype branch, so hipcc has to put an s_waitcnt vmcnt(0) into every branch: N operands of
// run-time dtype became N SERIAL round trips (1.8 us of a 4.5 us operator launch, s_memtime stamps in scripts/tilelab.hip).
// raw4_load only issues the load (8 B for the 16-bit types, 16 B for fp32) and hands back the bits; raw4_cvt converts
// where the value is used, after everything else has been requested.
__device__ __forceinline__ uint4 raw4_load(const void *p, int dt, int64_t i)
{
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if (dt == QUIPAMD_F32) r = *reinterpret_cast<const uint4 *>((const float *)p + i);
    else {
        const uint2 t = *reinterpret_cast<const uint2 *>((const uint16_t *)p + i);
        r.x = t.x;
        r.y = t.y;
    }
    return r;
}
__device__ __forceinline__ float4 raw4_cvt(const uint4 &r, int dt)
{
    if (dt == QUIPAMD_F32) return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
    if (dt == QUIPAMD_F16)
        return make_float4(f16_bits_to_f32(r.x & 0xffff), f16_bits_to_f32(r.x >> 16), f16_bits_to_f32(r.y & 0xffff), f16_bits_to_f32(r.y >> 16));
    return make_float4(__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u));
}
__device__ __forceinline__ uint32_t raw1_load(const void *p, int dt, int64_t i)
{
    return dt == QUIPAMD_F32 ? ((const uint32_t *)p)[i] : (uint32_t)((const uint16_t *)p)[i];
}
__device__ __forceinline__ float raw1_cvt(uint32_t r, int dt)
{
    return dt == QUIPAMD_F32 ? __uint_as_float(r) : dt == QUIPAMD_F16 ? f16_bits_to_f32((uint16_t)r) : __uint_as_float(r << 16);
}

// wave-wide sum on the DPP network (4 v_add with a DPP operand + 4 v_readlane) instead of six ds_bpermute round trips;
// every lane gets the total
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>{});      // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4E>{});      // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x141>{});     // row_half_mirror: lanes 0-7 <-> 7-0
    v += dpp(v, std::integral_constant<int, 0x140>{});     // row_mirror: every lane of a 16-lane row holds the row sum
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
// block-wide sum over NWAVES waves: one LDS round, ONE barrier (each call site owns its red[] slots)
template <int NWAVES> __device__ __forceinline__ float block_sum_dpp(float v, float *red /* [NWAVES] */)
{
    const float w = wave_sum_dpp(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVES; i += 4) {
        const float4 r = *reinterpret_cast<const float4 *>(red + i);
        t += (r.x + r.y) + (r.z + r.w);
    }
    return t;
}

// block-wide sum over 1024 threads (16 waves): DPP wave sums + one LDS round; the trailing barrier frees red[] for the next call
__device__ __forceinline__ float block_sum(float v, float *red /* [16] */)
{
    const float t = block_sum_dpp<16>(v, red);
    __syncthreads();
    return t;
}

// ---- split-bf16 variant ------------------------------------------------------------------------------------------------
// One row is 2 n (p + q) flops on ONE CU; on the fp32 matrix pipe (614 GFLOP/s per CU) that is 5.1 us at n = 8192, the
// measured floor of the kernel above.  Here every operand is carried as bf16 hi + lo (v ~ hi + lo to 2^-17): factors are
// pre-split on the host, the row is split when it is written to LDS, and each 32-deep step is three
// v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi, fp32 accumulate) -- ~5x the fp32 MFMA rate at ~1e-5 relative error,
// two orders inside the 1e-3 contract of the projection.  Requires p, q multiples of 32.
// LDS images (bf16, rows padded by 8 elements = one 16-byte slot):
//   F0h/F0l [p][p+8], F1h/F1l [q][q+8];  ZA h/l = z^T [q][p+8] (input of "mix a": 8 consecutive a' per lane);
//   ZB h/l = z [p][q+8] (input of "mix b");  ZF fp32 [p][q+4] (final image) aliases the first stage's input.
union Frag8 {
    uint4 u;
    bf16x8_t v;
};

__device__ __forceinline__ void split_bf16(float v, uint16_t &hi, uint16_t &lo)
{
    hi = f32_to_bf16_bits(v);
    lo = f32_to_bf16_bits(v - bf16_bits_to_f32(hi));
}

inline size_t small_split_lds(int p, int q)
{
    return (size_t)2 * (2 * ((size_t)p * (p + 8) + (size_t)q * (q + 8)) + 2 * (size_t)q * (p + 8) + 2 * (size_t)p * (q + 8)) + 64;
}

// One application of a Kronecker operator to ONE row by the 1024 threads of a workgroup (the body of
// ortho_small_split_kernel, also used by the operator-fused dequant-GEMM in dqgemm_vop.hip):
//   [x from memory | xv handed over] -> [LayerNorm] -> colscale -> scatter -> mix, mix -> gather + bias + residual + relu
// and every group of 4 consecutive results is passed to epi(u, v4, value) (element index 4 * v4; u = register slot).
// smemc: small_split_lds(p, q) bytes.  All threads must call it; it ends after the epilogue WITHOUT a barrier.
// load_factors = false: the factor images of A are still in LDS from the previous call (same A, next row).
template <class TI, int CP, int CQ, class Epi>
__device__ __forceinline__ void small_split_pass(const SmallArgs &A, int64_t row, bool load_x, float4 (&xv)[4], char *smemc, Epi &&epi,
                                                 bool load_factors = true)
{
    constexpr int MAXV = 4;
    const int p = CP ? CP : A.p, q = CQ ? CQ : A.q, n = p * q;
    const int P8 = p + 8, Q8 = q + 8, QS = q + 4;
    uint16_t *F0h = reinterpret_cast<uint16_t *>(smemc), *F0l = F0h + p * P8;
    uint16_t *F1h = F0l + p * P8, *F1l = F1h + q * Q8;
    uint16_t *ZAh = F1l + q * Q8, *ZAl = ZAh + q * P8;             // [q][P8]
    uint16_t *ZBh = ZAl + q * P8, *ZBl = ZBh + p * Q8;             // [p][Q8]
    float *red = reinterpret_cast<float *>(ZBl + p * Q8);          // [16]
    const bool a_first = A.b_first == 0;
    float *ZF = reinterpret_cast<float *>(a_first ? ZAh : ZBh);    // final fp32 image over the dead first-stage input
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- factors: bf16 hi / lo, 8 elements (16 B) per thread step, into padded rows --------------------------------
    if (load_factors) {
        for (int i = tid; i < p * p / 8; i += 1024) {
            const int rr = i / (p / 8), c8 = i - rr * (p / 8);
            *reinterpret_cast<uint4 *>(F0h + rr * P8 + 8 * c8) = reinterpret_cast<const uint4 *>(A.M0_hi)[i];
            *reinterpret_cast<uint4 *>(F0l + rr * P8 + 8 * c8) = reinterpret_cast<const uint4 *>(A.M0_lo)[i];
        }
        for (int i = tid; i < q * q / 8; i += 1024) {
            const int rr = i / (q / 8), c8 = i - rr * (q / 8);
            *reinterpret_cast<uint4 *>(F1h + rr * Q8 + 8 * c8) = reinterpret_cast<const uint4 *>(A.M1_hi)[i];
            *reinterpret_cast<uint4 *>(F1l + rr * Q8 + 8 * c8) = reinterpret_cast<const uint4 *>(A.M1_lo)[i];
        }
    }
    // ---- row: 4 consecutive elements per step, optional LayerNorm, scale, split, scatter --------------------------------
    const int qsh = __builtin_ctz(q), qmask = q - 1, n4 = n >> 2;
    if (load_x) {
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            const int v4 = tid + 1024 * u;
            xv[u] = v4 < n4 ? load4<TI>(A.x, row * A.ldx + 4 * v4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    // every operand of the scatter and of the epilogue is requested NOW, before the LayerNorm reductions and the two
    // mix stages: fetched where they are used they were 2-3 exposed L2 round trips per launch (a decode step is a chain
    // of these launches; rocprof: 6-9 us each).  PF = 2 covers n <= 8192; larger rows load the rest in place.
    constexpr int PF = 2;
    // LayerNorm parameters and the residual come in the model's dtype: for f16 (the decode case) only the BITS are requested
    // here and converted where they are used -- load4_any converts inside its dtype branch, which costs a full wait per
    // operand (raw4_load above); other dtypes are loaded in place, later.
    const bool ln16 = A.ln_gamma && A.ln_dtype == QUIPAMD_F16, res16 = A.residual && A.res_dtype == QUIPAMD_F16;
    float4 pcs[PF], pbias[PF];
    uint2 rgm[PF], rbt[PF], rres[PF];
    int4 pld[PF], pst[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int v4 = tid + 1024 * u;
        pbias[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        pcs[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        pld[u] = pst[u] = make_int4(4 * v4, 4 * v4 + 1, 4 * v4 + 2, 4 * v4 + 3);
        rgm[u] = rbt[u] = rres[u] = make_uint2(0u, 0u);
        if (v4 < n4) {
            if (ln16) {
                rgm[u] = *reinterpret_cast<const uint2 *>((const uint16_t *)A.ln_gamma + 4 * v4);
                if (A.ln_beta) rbt[u] = *reinterpret_cast<const uint2 *>((const uint16_t *)A.ln_beta + 4 * v4);
            }
            if (A.colscale) pcs[u] = *reinterpret_cast<const float4 *>(A.colscale + 4 * v4);
            if (A.load_idx) pld[u] = *reinterpret_cast<const int4 *>(A.load_idx + 4 * v4);
            if (A.store_idx) pst[u] = *reinterpret_cast<const int4 *>(A.store_idx + 4 * v4);
            if (A.bias) pbias[u] = *reinterpret_cast<const float4 *>(A.bias + 4 * v4);
            if (res16) rres[u] = *reinterpret_cast<const uint2 *>((const uint16_t *)A.residual + row * A.ldo + 4 * v4);
        }
    }
    auto f16x4 = [](const uint2 &r) { return raw4_cvt(make_uint4(r.x, r.y, 0u, 0u), QUIPAMD_F16); };
    if (A.ln_gamma) {
        // ln_beta == NULL: RMSNorm (Llama) -- no mean, no shift:  x * rsqrt(mean(x^2) + eps) * gamma
        const bool rms = A.ln_beta == nullptr;
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
                const float4 gm = (u < PF && ln16) ? f16x4(rgm[u < PF ? u : 0]) : load4_any(A.ln_gamma, A.ln_dtype, 4 * v4);
                const float4 bt = rms ? make_float4(0.f, 0.f, 0.f, 0.f)
                                      : (u < PF && ln16) ? f16x4(rbt[u < PF ? u : 0]) : load4_any(A.ln_beta, A.ln_dtype, 4 * v4);
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
                const float4 c = u < PF ? pcs[u < PF ? u : 0] : *reinterpret_cast<const float4 *>(A.colscale + 4 * v4);
                v = make_float4(v.x * c.x, v.y * c.y, v.z * c.z, v.w * c.w);
            }
            int4 pos = make_int4(4 * v4, 4 * v4 + 1, 4 * v4 + 2, 4 * v4 + 3);
            if (A.load_idx) pos = u < PF ? pld[u < PF ? u : 0] : *reinterpret_cast<const int4 *>(A.load_idx + 4 * v4);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            const int pp[4] = {pos.x, pos.y, pos.z, pos.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = pp[e] >> qsh, b = pp[e] & qmask;
                uint16_t hi, lo;
                split_bf16(vv[e], hi, lo);
                const int off = a_first ? b * P8 + a : a * Q8 + b;       // z^T for "mix a" first, z for "mix b" first
                (a_first ? ZAh : ZBh)[off] = hi;
                (a_first ? ZAl : ZBl)[off] = lo;
            }
        }
    }
    __syncthreads();

    const int j = lane & 15, g = lane >> 4;
    const int nat = p / 16, nbt = q / 16;
    for (int st = 0; st < 2; ++st) {
        const bool mix_a = (st == 0) == a_first;
        const bool last = st == 1;
        if (mix_a) {
            // D[a = 16at + 4g + reg][b = 16bt + j] = sum_a' M0[a][a'] z[a'][b];  A = F0 rows, B = z^T rows (ZA)
            // n = 8192 (32 tiles on 16 waves): a wave takes two tiles that SHARE their factor rows, so the A fragments are
            // read once -- the stage is LDS-bandwidth bound (every tile re-read 16 KiB of fragments: 512 KiB = 4096 cycles)
            constexpr int TPW = (CP / 16) * (CQ / 16) == 32 ? 2 : 1;
            for (int tile = wave * TPW; tile < nat * nbt; tile += 16 * TPW) {
                const int at = tile / nbt, bt0 = tile - at * nbt;
                // three independent accumulator chains (one per product), summed small-to-large at the end
                f32x4_t acc[TPW], acc1[TPW], acc2[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t] = acc1[t] = acc2[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const int fo = (16 * at + j) * P8 + 8 * g;
#pragma unroll 4
                for (int S = 0; S < p / 32; ++S) {
                    Frag8 ah, al;
                    ah.u = *reinterpret_cast<const uint4 *>(F0h + fo + 32 * S);
                    al.u = *reinterpret_cast<const uint4 *>(F0l + fo + 32 * S);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const int zo = (16 * (bt0 + t) + j) * P8 + 8 * g;
                        Frag8 bh, bl;
                        bh.u = *reinterpret_cast<const uint4 *>(ZAh + zo + 32 * S);
                        bl.u = *reinterpret_cast<const uint4 *>(ZAl + zo + 32 * S);
                        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al.v, bh.v, acc1[t], 0, 0, 0);
                        acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bl.v, acc2[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bh.v, acc[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int bt = bt0 + t;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[t][reg] += acc1[t][reg] + acc2[t][reg];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int a = 16 * at + 4 * g + reg, b = 16 * bt + j;
                        if (last) ZF[a * QS + b] = acc[t][reg];
                        else {
                            uint16_t hi, lo;
                            split_bf16(acc[t][reg], hi, lo);
                            ZBh[a * Q8 + b] = hi;
                            ZBl[a * Q8 + b] = lo;
                        }
                    }
                }
            }
        } else {
            // D[b = 16bt + 4g + reg][a = 16at + j] = sum_b' M1[b][b'] z[a][b'];  A = F1 rows, B = z rows (ZB)
            constexpr int TPW = (CP / 16) * (CQ / 16) == 32 ? 2 : 1;
            for (int tile = wave * TPW; tile < nat * nbt; tile += 16 * TPW) {
                const int bt = tile / nat, at0 = tile - bt * nat;
                f32x4_t acc[TPW], acc1[TPW], acc2[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t] = acc1[t] = acc2[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const int fo = (16 * bt + j) * Q8 + 8 * g;
#pragma unroll 4
                for (int S = 0; S < q / 32; ++S) {
                    Frag8 ah, al;
                    ah.u = *reinterpret_cast<const uint4 *>(F1h + fo + 32 * S);
                    al.u = *reinterpret_cast<const uint4 *>(F1l + fo + 32 * S);
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const int zo = (16 * (at0 + t) + j) * Q8 + 8 * g;
                        Frag8 bh, bl;
                        bh.u = *reinterpret_cast<const uint4 *>(ZBh + zo + 32 * S);
                        bl.u = *reinterpret_cast<const uint4 *>(ZBl + zo + 32 * S);
                        acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al.v, bh.v, acc1[t], 0, 0, 0);
                        acc2[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bl.v, acc2[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah.v, bh.v, acc[t], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int at = at0 + t;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) acc[t][reg] += acc1[t][reg] + acc2[t][reg];
                    const int a = 16 * at + j, b0 = 16 * bt + 4 * g;
                    if (last) *reinterpret_cast<float4 *>(ZF + a * QS + b0) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                    else {
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            uint16_t hi, lo;
                            split_bf16(acc[t][reg], hi, lo);
                            ZAh[(b0 + reg) * P8 + a] = hi;
                            ZAl[(b0 + reg) * P8 + a] = lo;
                        }
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue, 4 outputs per step ---------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int v4 = tid + 1024 * u;
        if (v4 < n4) {
            int4 pos = make_int4(4 * v4, 4 * v4 + 1, 4 * v4 + 2, 4 * v4 + 3);
            if (A.store_idx) pos = u < PF ? pst[u < PF ? u : 0] : *reinterpret_cast<const int4 *>(A.store_idx + 4 * v4);
            float4 v = make_float4(ZF[(pos.x >> qsh) * QS + (pos.x & qmask)], ZF[(pos.y >> qsh) * QS + (pos.y & qmask)],
                                   ZF[(pos.z >> qsh) * QS + (pos.z & qmask)], ZF[(pos.w >> qsh) * QS + (pos.w & qmask)]);
            if (A.bias) {
                const float4 c = u < PF ? pbias[u < PF ? u : 0] : *reinterpret_cast<const float4 *>(A.bias + 4 * v4);
                v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
            }
            if (A.residual) {
                const float4 c = (u < PF && res16) ? f16x4(rres[u < PF ? u : 0]) : load4_any(A.residual, A.res_dtype, row * A.ldo + 4 * v4);
                v = make_float4(v.x + c.x, v.y + c.y, v.z + c.z, v.w + c.w);
            }
            if (A.relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            epi(u, v4, v);
        }
    }
}

}   // namespace
