// dqgemm_vop.hip -- K2 with the activation-side operator V (x (/) s) in its prologue: the decode step's
//     y_i = What_i ( V_i ( [LayerNorm](x) (/) s_i ) ),   i = 1..3 layers sharing x (q / k / v), d = 2048, bs <= 8
// in ONE launch instead of two.  A rocprof trace of the decode loop shows the single-workgroup
// operator launch (5.5 - 6.5 us) and the batch-1 dequant-GEMM (5.4 us, most of it waiting for cold weights) back to back;
// here every workgroup of the GEMM issues its weight loads FIRST and then applies the operator to x itself (small_pass.h,
// ~4 us of latency-bound phases that now overlap the HBM round trip), writing the bf16 result straight into the
// XOR-swizzled LDS slabs the MFMA B fragments are read from -- x~ never exists in memory.  The redundant operator work
// (one application per workgroup) is 0.4 MFLOP each; workgroups own 32 rows (RT = 2) to halve that redundancy.
// Same arithmetic in the same order as ortho_small_split_kernel followed by dqgemm_tile_kernel: bit-identical results.
#include "common.h"
#include "dq_common.h"
#include "small_pass.h"

namespace {

constexpr int VOP_MAXG = 3;

struct VopBatch {
    SmallArgs v[VOP_MAXG];            // V-side descriptors (x, LN, column scale, factors, index vectors); `out` unused
    const uint4 *qw[VOP_MAXG];
    const float *scale[VOP_MAXG];
    const float *bias[VOP_MAXG];
    float *y[VOP_MAXG];
};

template <class TI, int CP, int CQ>
__global__ __launch_bounds__(1024) void dqgemm_vop_kernel(VopBatch G, int64_t m, int bs, int maxq, float two_over_maxq)
{
    typedef Deq<2> Q;
    constexpr int KC = Q::KC, CW = 8, RT = 2, XB = 16 * KC * 2;             // 8 KiB slab per 256-column chunk
    constexpr int D = CP * CQ;
    static_assert(D == CW * KC, "one chunk per chunk slot: d = 2048");
    extern __shared__ __attribute__((aligned(16))) char smem[];            // [CW slabs][park][operator images]
    char *slabs = smem;
    float *park = reinterpret_cast<float *>(smem + CW * XB);               // [CW][RT][4][64] acc, then [CW][64] xsum
    char *opmem = smem + CW * XB + (CW * RT * 256 + CW * 64) * 4;

    const int gi = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = wave / RT, r = wave - c * RT;                            // chunk slot, row tile
    const int j = lane & 15, g = lane >> 4;
    const uint32_t rt = blockIdx.x * RT + r;

    // weights first: their HBM round trip runs under the operator pass
    const uint4 w = (G.qw[gi] + ((uint64_t)rt * CW + c) * 64)[lane];
    float e_sc = 0.f, e_bi = 0.f;
    if (wave < 4 * RT) {
        const int64_t row0 = (int64_t)(blockIdx.x * RT + (wave >> 2)) * 16 + (lane & 15);
        e_sc = G.scale[gi][0];
        if (G.bias[gi]) e_bi = G.bias[gi][row0];
    }

    // ---- operator: x~[b][k] (bf16) into slab k / 256, batch row b, 16-byte column ((k % 64) / 8) ^ (b & 7) -------------------
    const SmallArgs A = G.v[gi];
    float4 xv[4];
    for (int b = 0; b < bs; ++b) {
        small_split_pass<TI, CP, CQ>(A, (int64_t)b, true, xv, opmem, [&](int, int v4, const float4 &v) {
            const uint32_t k = 4u * (uint32_t)v4;
            const uint32_t col = k & (KC - 1), cc = col & 63;
            const uint32_t addr = (k / KC) * XB + ((col >> 6) * 2 + (b >> 3)) * 1024 + (b & 7) * 128 + (((cc >> 3) ^ (b & 7)) << 4) + (cc & 7) * 2;
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16_bits(v.x) | ((uint32_t)f32_to_bf16_bits(v.y) << 16);
            pk.y = (uint32_t)f32_to_bf16_bits(v.z) | ((uint32_t)f32_to_bf16_bits(v.w) << 16);
            *reinterpret_cast<uint2 *>(slabs + addr) = pk;
        });
        __syncthreads();                                                   // images are rewritten by the next row / slabs complete
    }

    // ---- dequant + MFMA over this wave's chunk (dqgemm_tile_kernel, DEPTH 1, one chunk group) -------------------------------------
    const uint32_t rd_base = (j >> 3) * 1024 + (j & 7) * 128;
    const uint32_t rd0 = rd_base + (((0 + g) ^ (j & 7)) << 4);
    const uint32_t rd1 = rd_base + (((4 + g) ^ (j & 7)) << 4);
    const char *slab = slabs + c * XB;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float xs = 0.f;
    {
        uint4 xf[Q::NT];
#pragma unroll
        for (int t = 0; t < Q::NT; ++t) xf[t] = *reinterpret_cast<const uint4 *>(slab + (t >> 1) * 2048 + ((t & 1) ? rd1 : rd0));
#pragma unroll
        for (int t = 0; t < Q::NT; ++t) {
            Frag a, bb;
            a.u = Q::frag(w, t);
            bb.u = xf[t];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bb.v, acc, 0, 0, 0);
            if (r == 0) xs = dot_ones(xf[t], xs);
        }
    }
    float *xpark = park + CW * RT * 256;
    {
        float *p = park + ((c * RT + r) * 4) * 64 + lane;
        p[0] = acc[0]; p[64] = acc[1]; p[128] = acc[2]; p[192] = acc[3];
        if (r == 0) {
            xs += __shfl_xor(xs, 16);
            xs += __shfl_xor(xs, 32);
            xpark[c * 64 + lane] = xs;
        }
    }
    __syncthreads();
    if (wave < 4 * RT) {                                                   // 16 waves >= 8 reducer slots: one each
        const int r2 = wave >> 2, q = wave & 3;
        const int b = 4 * q + (lane >> 4), wr = lane & 15;
        const int src = ((wr & 3) * 64) + b + 16 * (wr >> 2);              // [comp][mfma lane]
        float a = 0.f, xsum = 0.f;
#pragma unroll
        for (int v = 0; v < CW; ++v) {
            a += park[(v * RT + r2) * 256 + src];
            xsum += xpark[v * 64 + b];
        }
        const int64_t row = (int64_t)(blockIdx.x * RT + r2) * 16 + wr;
        if (b < bs) {
            const float alpha = e_sc * two_over_maxq, c0 = Q::OFF + 0.5f * (float)maxq;
            G.y[gi][(int64_t)b * m + row] = alpha * (a - c0 * xsum) + e_bi;
        }
    }
}

template <class TI>
int launch_vop(const VopBatch &B, int ngroups, int64_t m, int bs, int maxq, hipStream_t s)
{
    constexpr int CP = 64, CQ = 32, CW = 8, RT = 2, XB = 16 * 256 * 2;
    const size_t lds = (size_t)CW * XB + (size_t)(CW * RT * 256 + CW * 64) * 4 + small_split_lds(CP, CQ);
    auto kern = dqgemm_vop_kernel<TI, CP, CQ>;
    static QaPerDevice attr_set_dev;
    const int attr_set_d = attr_set_dev.dev();
    if ((attr_set_d < 0 || !attr_set_dev.done[attr_set_d])) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm_vop: cannot raise dynamic LDS to %zu", lds);
        if (attr_set_d >= 0) attr_set_dev.done[attr_set_d] = true;
    }
    kern<<<dim3((unsigned)(m / 16 / RT), (unsigned)ngroups), 1024, lds, s>>>(B, m, bs, maxq, 2.0f / (float)maxq);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm_vop");
    return QUIPAMD_OK;
}

}   // namespace

extern "C" int quipamd_dequant_gemm_vop(const quipamd_small_op *vops, const int32_t *const *qweight, const float *const *scale,
                                        const float *const *bias, float *const *y, int ngroups, int bits, int64_t bs, int64_t m,
                                        void *stream)
{
    QA_REQUIRE(vops && qweight && scale && y && ngroups >= 1 && ngroups <= VOP_MAXG, QUIPAMD_ERR_ARG, "dequant_gemm_vop: 1..%d groups", VOP_MAXG);
    QA_REQUIRE(bits == 2, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm_vop: 2-bit codes only");
    QA_REQUIRE(bs >= 0 && bs <= 8 && m > 0 && m % 32 == 0, QUIPAMD_ERR_SHAPE, "dequant_gemm_vop: needs bs <= 8 and m %% 32 == 0 (bs=%lld m=%lld)",
               (long long)bs, (long long)m);
    if (bs == 0) return QUIPAMD_OK;
    VopBatch B;
    for (int i = 0; i < ngroups; ++i) {
        const quipamd_small_op &o = vops[i];
        QA_REQUIRE(o.p == 64 && o.q == 32, QUIPAMD_ERR_SHAPE, "dequant_gemm_vop: the operator must be 64 x 32 (d = 2048); got %d x %d", o.p, o.q);
        QA_REQUIRE(o.x && o.M0_hi && o.M0_lo && o.M1_hi && o.M1_lo && qweight[i] && scale[i] && y[i], QUIPAMD_ERR_ARG,
                   "dequant_gemm_vop: null pointer in group %d", i);
        QA_REQUIRE(o.x_dtype == vops[0].x_dtype && (o.x_dtype == QUIPAMD_F16 || o.x_dtype == QUIPAMD_BF16), QUIPAMD_ERR_UNSUPPORTED,
                   "dequant_gemm_vop: x must be f16 or bf16");
        QA_REQUIRE(o.ldx >= 2048 && o.ldx % 4 == 0 && !o.residual && !o.bias && !o.relu, QUIPAMD_ERR_ARG,
                   "dequant_gemm_vop: V-side descriptor expected (no bias / residual / relu)");
        B.v[i] = o;
        B.qw[i] = (const uint4 *)qweight[i];
        B.scale[i] = scale[i];
        B.bias[i] = bias ? bias[i] : nullptr;
        B.y[i] = y[i];
    }
    for (int i = ngroups; i < VOP_MAXG; ++i) { B.v[i] = B.v[0]; B.qw[i] = B.qw[0]; B.scale[i] = B.scale[0]; B.bias[i] = B.bias[0]; B.y[i] = B.y[0]; }
    hipStream_t s = (hipStream_t)stream;
    return vops[0].x_dtype == QUIPAMD_F16 ? launch_vop<F16>(B, ngroups, m, (int)bs, 3, s) : launch_vop<BF16>(B, ngroups, m, (int)bs, 3, s);
}
