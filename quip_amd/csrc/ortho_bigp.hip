// ortho_bigp.hip -- K3 for a decode step when the Kronecker operator is p x 16 with a LARGE p (Llama's intermediate size
// 11008 = 688 x 16, method.py:16-18 butterfly_factors): neither factor image fits the one-workgroup kernel (688 x 688 fp32 =
// 1.9 MB), and the general two-launch kernel (ortho.hip) pads one row to 16 and gives the 688-deep mix to 16 workgroups:
// ~150 us per application at batch 1, three applications per Llama block.
// Here the p index is cut into 16-row tiles, one workgroup each (43 for p = 688); q = 16 is exactly one MFMA tile, so a
// workgroup that owns rows A of the image owns whole rows and can finish both stages:
//     mix a first:   t[A, :] = M0[A, :] z            (K = p, split over the 8 waves, met in LDS)      out[A, :] = t[A, :] M1^T
//     mix b first:   z1 = z M1^T for ALL rows (p/16 tiny tiles over the 8 waves), then  out[A, :] = M0[A, :] z1
// on v_mfma_f32_16x16x4_f32 (exact fp32 chains; the factor rows come straight from L2 as A fragments, 16 x p floats per
// workgroup).  Every workgroup loads, scales and scatters the whole row itself (n floats into a 16 x (p+4) LDS image).
// Operand sets as in ortho_tile.hip: SIDE 0 = x f16 + column scale, SIDE 1 = x f32 + bias (+ f16 residual); both permutations.
// SIDE 0 with RES = the Llama MLP hand-over: the input row is silu(gate) * up, gate = x, up = `residual` (f16 [rows, n], row
// stride ldx), computed on load -- llama's  down_proj(act_fn(gate_proj(x)) * up_proj(x))  without the two elementwise launches.
#include "common.h"

#include "small_pass.h"

namespace {

constexpr int BT = 512;            // threads per workgroup
constexpr int BMAXV = 6;           // float4 groups per thread: n <= 4 * 512 * 6 = 12288

struct BigpBatch {
    SmallArgs op[QUIPAMD_SMALL_MAX_OPS];
    const int32_t *store_inv[QUIPAMD_SMALL_MAX_OPS];
};

template <int SIDE, bool RES>
__global__ __launch_bounds__(BT) void ortho_bigp_kernel(BigpBatch Bt)
{
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const SmallArgs A = Bt.op[blockIdx.y];
    const int32_t *store_inv = Bt.store_inv[blockIdx.y];
    const int64_t row = blockIdx.z;
    const int p = A.p, n = p * 16, n4 = n / 4, PS = p + 4;
    const bool a_first = A.b_first == 0;
    float *ZT = sm;                            // [16][PS]   zT[b][a']  (B operand of "mix a": 4 consecutive a' per lane)
    float *ZR = sm + 16 * PS;                  // [p][20]    z[a'][b']  (mix b first: B operand of the tiny b mixes)   | partial tiles
    const int at = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;

    // ---- request the whole row and its scatter operands ----------------------------------------------------------------------------
    const uint16_t *xrow16 = (const uint16_t *)A.x + row * A.ldx;
    const float *xrow32 = (const float *)A.x + row * A.ldx;
    const uint16_t *urow16 = (const uint16_t *)A.residual + row * A.ldx;       // SIDE 0 + RES: the `up` row
    uint4 rx[BMAXV];
    uint2 ru[BMAXV];
    float4 pcs[BMAXV];
    int4 pld[BMAXV];
#pragma unroll
    for (int u = 0; u < BMAXV; ++u) {
        const int v4 = tid + BT * u;
        if (v4 < n4) {
            if constexpr (SIDE == 0) {
                const uint2 t = *reinterpret_cast<const uint2 *>(xrow16 + 4 * v4);
                rx[u] = make_uint4(t.x, t.y, 0u, 0u);
                pcs[u] = *reinterpret_cast<const float4 *>(A.colscale + 4 * v4);
                if constexpr (RES) ru[u] = *reinterpret_cast<const uint2 *>(urow16 + 4 * v4);
            } else {
                rx[u] = *reinterpret_cast<const uint4 *>(xrow32 + 4 * v4);
            }
            pld[u] = *reinterpret_cast<const int4 *>(A.load_idx + 4 * v4);
        }
    }
    // the four outputs this lane of wave 0 will hold at the end (D: row = 4g + reg, col = j)
    int oidx[4] = {0, 0, 0, 0};
    if (wave == 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            // mix a first ends with D[b][a] (row = b, col = a); mix b first ends with D[a][b]
            const uint32_t pos = a_first ? (uint32_t)((16 * at + j) * 16 + 4 * g + reg) : (uint32_t)((16 * at + 4 * g + reg) * 16 + j);
            oidx[reg] = store_inv[pos];
        }
    }
#pragma unroll
    for (int u = 0; u < BMAXV; ++u) {
        const int v4 = tid + BT * u;
        if (v4 < n4) {
            float4 v = raw4_cvt(rx[u], SIDE == 0 ? QUIPAMD_F16 : QUIPAMD_F32);
            if constexpr (SIDE == 0 && RES) {
                // silu(g) * up, rounded to f16 like the two torch launches it replaces (F.silu(g) rounds, then the product rounds)
                const float4 up = raw4_cvt(make_uint4(ru[u].x, ru[u].y, 0u, 0u), QUIPAMD_F16);
                auto gate = [](float g, float w) {
                    const float sl = f16_bits_to_f32(f32_to_f16_bits(g / (1.f + __expf(-g))));
                    return f16_bits_to_f32(f32_to_f16_bits(sl * w));
                };
                v = make_float4(gate(v.x, up.x), gate(v.y, up.y), gate(v.z, up.z), gate(v.w, up.w));
            }
            if constexpr (SIDE == 0) v = make_float4(v.x * pcs[u].x, v.y * pcs[u].y, v.z * pcs[u].z, v.w * pcs[u].w);
            const float vv[4] = {v.x, v.y, v.z, v.w};
            const int pp[4] = {pld[u].x, pld[u].y, pld[u].z, pld[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = pp[e] >> 4, b = pp[e] & 15;
                if (a_first) ZT[b * PS + a] = vv[e];
                else ZR[a * 20 + b] = vv[e];
            }
        }
    }
    __syncthreads();
    float obias[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t rres[4] = {0u, 0u, 0u, 0u};
    if (wave == 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if constexpr (SIDE == 1) obias[reg] = A.bias[(uint32_t)oidx[reg]];
            if constexpr (SIDE == 1 && RES) rres[reg] = ((const uint16_t *)A.residual + row * A.ldo)[(uint32_t)oidx[reg]];
        }
    }

    if (!a_first) {
        // ---- mix b for ALL rows: z1T[b][a'] = sum_b' M1[b][b'] z[a'][b'];  D[b = 4g + reg][a' = 16t + j] ---------------------------------
        // K = 16: one float4 per lane covers k = 4g .. 4g+3 of row j of M1; the 4 MFMAs take .x .y .z .w (k = 4g' + s over the lane groups)
        const float4 fa = *reinterpret_cast<const float4 *>(A.M1 + j * 16 + 4 * g);
        for (int t = wave; t < p / 16; t += BT / 64) {
            const float4 zb = *reinterpret_cast<const float4 *>(ZR + (16 * t + j) * 20 + 4 * g);
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.x, zb.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.y, zb.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.z, zb.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa.w, zb.w, acc, 0, 0, 0);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) ZT[(4 * g + reg) * PS + 16 * t + j] = acc[reg];
        }
        __syncthreads();
    }

    // ---- mix a for the tile's 16 rows: D[a = 16at + 4g + reg][b = j] = sum_a' M0[a][a'] zT[b][a'],  K = p over the 8 waves ----------
    {
        const int ksteps = p / 16;                               // 16-deep steps (4 MFMAs each)
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, acc2 = acc;
        const float *m0 = A.M0 + (uint32_t)((16 * at + j) * p + 4 * g);
        const float *zt = ZT + j * PS + 4 * g;
        for (int S = wave; S < ksteps; S += BT / 64) {
            const float4 a4 = *reinterpret_cast<const float4 *>(m0 + 16 * S);
            const float4 b4 = *reinterpret_cast<const float4 *>(zt + 16 * S);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc2, 0, 0, 0);
        }
        __syncthreads();                                          // ZR (mix b first) is dead: its head holds the 8 partial tiles
        float *part = ZR;                                         // [8][16][20]:  part[w][a_local][b]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) part[(wave * 16 + 4 * g + reg) * 20 + j] = acc[reg] + acc2[reg];
        __syncthreads();
    }
    if (wave != 0) return;
    // T[a_local][b] = sum over the 8 partials;  lane (j, g) takes what its last MFMA (or its store) needs
    auto tsum = [&](int a_local, int b) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < BT / 64; ++w) t += ZR[(w * 16 + a_local) * 20 + b];
        return t;
    };
    float ov[4];
    if (a_first) {
        // out[a][b] = sum_b' M1[b][b'] T[a][b']:  D[b = 4g + reg][a = j], A = M1 rows (lane j = b, k = b'), B = T rows (lane j = a, k = b')
        const float4 a4 = *reinterpret_cast<const float4 *>(A.M1 + j * 16 + 4 * g);
        const float4 b4 = make_float4(tsum(j, 4 * g + 0), tsum(j, 4 * g + 1), tsum(j, 4 * g + 2), tsum(j, 4 * g + 3));
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, b4.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, b4.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, b4.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, b4.w, acc, 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ov[reg] = acc[reg];
    } else {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ov[reg] = tsum(4 * g + reg, j);       // D[a = 4g + reg][b = j]
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float v = (ov[reg] + obias[reg]) + ((SIDE == 1 && RES) ? f16_bits_to_f32((uint16_t)rres[reg]) : 0.f);
        ov[reg] = (SIDE == 1 && A.relu) ? fmaxf(v, 0.f) : v;
    }
    if (A.out_dtype == QUIPAMD_F32) {
        float *o = (float *)A.out + row * A.ldo;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) o[(uint32_t)oidx[reg]] = ov[reg];
    } else {
        uint16_t *o = (uint16_t *)A.out + row * A.ldo;
        const bool h = A.out_dtype == QUIPAMD_F16;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) o[(uint32_t)oidx[reg]] = h ? f32_to_f16_bits(ov[reg]) : f32_to_bf16_bits(ov[reg]);
    }
}

}   // namespace

extern "C" int quipamd_ortho_apply_bigp_supported(int p, int q)
{
    return q == 16 && p % 16 == 0 && p >= 64 && p * 16 <= 4 * BT * BMAXV;
}

extern "C" int quipamd_ortho_apply_bigp(const quipamd_small_op *ops, const int32_t *const *store_inv, int nops, int64_t rows, void *stream)
{
    QA_REQUIRE(ops && store_inv && nops >= 1 && nops <= QUIPAMD_SMALL_MAX_OPS, QUIPAMD_ERR_ARG, "ortho_apply_bigp: 1..%d ops", QUIPAMD_SMALL_MAX_OPS);
    const int p = ops[0].p, q = ops[0].q;
    QA_REQUIRE(quipamd_ortho_apply_bigp_supported(p, q), QUIPAMD_ERR_UNSUPPORTED,
               "ortho_apply_bigp: p x q = %d x %d (wants q = 16, p a multiple of 16, p * 16 <= %d)", p, q, 4 * BT * BMAXV);
    QA_REQUIRE(rows >= 0 && rows <= 65535, QUIPAMD_ERR_SHAPE, "ortho_apply_bigp: bad row count");
    if (rows == 0) return QUIPAMD_OK;
    BigpBatch B;
    int side = -1;
    bool res = false;
    for (int i = 0; i < nops; ++i) {
        const quipamd_small_op &o = ops[i];
        QA_REQUIRE(o.p == p && o.q == q && o.M0 && o.M1 && o.x && o.out && o.load_idx && o.store_idx && store_inv[i], QUIPAMD_ERR_ARG,
                   "ortho_apply_bigp: op %d: same p, q; factors, x, out and both permutations (with the inverse store map) wanted", i);
        QA_REQUIRE(o.ldx >= (int64_t)p * q && o.ldo >= (int64_t)p * q && o.ldx % 4 == 0 && !o.ln_gamma, QUIPAMD_ERR_SHAPE,
                   "ortho_apply_bigp: leading dimensions / no normalisation on this path");
        int sd = -1;
        // activation side; `residual` (f16, same row stride as x) + relu = 1 there means "x is the gate, residual is up: feed silu(x) * up"
        if (o.x_dtype == QUIPAMD_F16 && o.colscale && !o.bias && ((!o.residual && !o.relu) || (o.residual && o.relu && o.res_dtype == QUIPAMD_F16))) sd = 0;
        else if (o.x_dtype == QUIPAMD_F32 && !o.colscale && o.bias && (!o.residual || o.res_dtype == QUIPAMD_F16)) sd = 1;
        QA_REQUIRE(sd >= 0, QUIPAMD_ERR_UNSUPPORTED, "ortho_apply_bigp: op %d is neither (x f16, colscale, [silu-gate pair]) nor (x f32, bias, [f16 residual])", i);
        const bool r = o.residual != nullptr;
        QA_REQUIRE(side < 0 || (sd == side && r == res), QUIPAMD_ERR_ARG, "ortho_apply_bigp: ops of one launch must have the same operand set");
        side = sd;
        res = r;
        B.op[i] = o;
        B.store_inv[i] = store_inv[i];
    }
    for (int i = nops; i < QUIPAMD_SMALL_MAX_OPS; ++i) { B.op[i] = ops[0]; B.store_inv[i] = store_inv[0]; }
    if (rows == 0) return QUIPAMD_OK;
    const size_t lds = ((size_t)16 * (p + 4) + (size_t)(p > 128 ? p : 128) * 20) * sizeof(float);
    const dim3 grid((unsigned)(p / 16), (unsigned)nops, (unsigned)rows);
    hipStream_t s = (hipStream_t)stream;
#define QA_BIGP(SD, RS)                                                                                                              \
    do {                                                                                                                             \
        auto kern = ortho_bigp_kernel<SD, RS>;                                                                                       \
        if (lds > 64 * 1024 &&                                                                                                       \
            hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)             \
            return qa_fail(QUIPAMD_ERR_LAUNCH, "ortho_apply_bigp: cannot raise dynamic LDS to %zu", lds);                            \
        kern<<<grid, BT, lds, s>>>(B);                                                                                               \
    } while (0)
    if (side == 0 && res) QA_BIGP(0, true);
    else if (side == 0) QA_BIGP(0, false);
    else if (res) QA_BIGP(1, true);
    else QA_BIGP(1, false);
#undef QA_BIGP
    QA_LAUNCH_CHECK("quipamd_ortho_apply_bigp");
    return QUIPAMD_OK;
}
