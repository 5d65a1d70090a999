// ortho_tile.hip -- K3 for a decode step: the Kronecker operator applied to ONE row by MANY workgroups
//
// ortho_small.hip gives a row to one workgroup: at batch 1 a whole operator application (n = 2048: 5.2 us, n = 8192: 10 us,
// profiles/r02j_decode_kernel_trace.txt) runs on a single CU while 255 idle, and a decode step is a chain of them.
// Here the OUTPUT image z2[a][b] (p x q) is cut into 16 x 16 tiles and each tile gets its own workgroup:
//     z2[A, B] = ( M0[A, :] z ) M1[B, :]^T          (mix a first;  mix b first: ( z M1[B, :]^T ) then M0[A, :])
// i.e. stage 1 for the 16 rows (columns) of the tile over the whole other index -- q/16 (p/16) MFMA tiles, one per wave --
// and stage 2 for the single output tile.  Every workgroup redoes the cheap part (load the row, LayerNorm statistics,
// column scale, bf16 hi/lo split, scatter into the LDS image: n elements over 512 threads) and reads only the 16 factor
// rows it needs straight from L2 as MFMA A-fragments; nothing but the row image lives in LDS.  8 (n = 2048) to 32 (n = 8192)
// workgroups per operator, times the operators of the launch (q / k / v share their input), times the rows.
// Arithmetic: split-bf16 (hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16), as small_pass.h.
// The epilogue needs the inverse of the store permutation (image position -> output index): `store_inv`.
#include "common.h"

#include "small_pass.h"

#ifndef TILE_STAMP                 // scripts/tilelab.hip defines it (s_memtime stamps per phase); nothing in the library
#define TILE_STAMP(i) do { } while (0)
#endif

namespace {

struct TileBatch {
    SmallArgs op[QUIPAMD_SMALL_MAX_OPS];
    const int32_t *store_inv[QUIPAMD_SMALL_MAX_OPS];
};

constexpr int TT = 512;            // threads per workgroup


// P, Q: factor sizes (compile time: the k loops unroll and the A-fragments of both stages are prefetched into registers).
// The operand set is compile time too -- a run-time `if (pointer)` / `switch (dtype)` around a load makes hipcc merge the arms
// through register copies, and a copy of a load result is a wait: the first version spent 1.8 us of its 4.5 us issuing its
// operands one round trip at a time (scripts/tilelab.hip stamps).  The two shapes a decode step has:
//   SIDE 0 (activation side, V):  x f16, [LayerNorm f16 gamma/beta: FLAG], column scale, both permutations, no epilogue operands
//   SIDE 1 (output side, U^T):    x f32, both permutations, bias, [residual f16: FLAG], relu at run time
// anything else is refused by quipamd_ortho_apply_tiles (the caller falls back to quipamd_ortho_apply_small_ops).
template <int P, int Q, int SIDE, bool FLAG>
__global__ __launch_bounds__(TT) void ortho_tile_kernel(TileBatch Bt)
{
    constexpr bool LN = SIDE == 0 && FLAG, RES = SIDE == 1 && FLAG;
    constexpr int N = P * Q, N4 = N / 4, MAXV = (N4 + TT - 1) / TT;
    constexpr int P8 = P + 8, Q8 = Q + 8, NAT = P / 16, NBT = Q / 16;
    constexpr int QSH = Q == 32 ? 5 : Q == 64 ? 6 : 7, QMASK = Q - 1;
    static_assert((1 << QSH) == Q && MAXV <= 4 && N4 % TT == 0, "shape");
    __shared__ __attribute__((aligned(16))) uint16_t Zh[Q * P8 > P * Q8 ? Q * P8 : P * Q8];
    __shared__ __attribute__((aligned(16))) uint16_t Zl[Q * P8 > P * Q8 ? Q * P8 : P * Q8];
    __shared__ __attribute__((aligned(16))) uint16_t Th[16 * (P8 > Q8 ? P8 : Q8)];
    __shared__ __attribute__((aligned(16))) uint16_t Tl[16 * (P8 > Q8 ? P8 : Q8)];
    __shared__ __attribute__((aligned(16))) float red[16];          // two reductions, own slots each: one barrier per reduction
    const SmallArgs A = Bt.op[blockIdx.y];
    const int32_t *store_inv = Bt.store_inv[blockIdx.y];
    const int64_t row = blockIdx.z;
    const bool a_first = A.b_first == 0;
    const int tile = blockIdx.x;
    const int at = a_first ? tile / NBT : tile % NAT, bt = a_first ? tile % NBT : tile / NAT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;

    TILE_STAMP(0);
    // ---- every global operand is requested now -----------------------------------------------------------------------------------
    const uint16_t *xrow16 = (const uint16_t *)A.x + row * A.ldx;          // wave-uniform bases + 32-bit lane offsets: SADDR loads
    const float *xrow32 = (const float *)A.x + row * A.ldx;
    uint4 rx[MAXV];
    uint2 rgm[MAXV], rbt[MAXV];
    float4 pcs[MAXV];
    int4 pld[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        const int v4 = tid + TT * u;
        if constexpr (SIDE == 0) {
            const uint2 t = *reinterpret_cast<const uint2 *>(xrow16 + 4 * v4);
            rx[u] = make_uint4(t.x, t.y, 0u, 0u);
            pcs[u] = *reinterpret_cast<const float4 *>(A.colscale + 4 * v4);
        } else {
            rx[u] = *reinterpret_cast<const uint4 *>(xrow32 + 4 * v4);
        }
        if constexpr (LN) {
            rgm[u] = *reinterpret_cast<const uint2 *>((const uint16_t *)A.ln_gamma + 4 * v4);
            rbt[u] = *reinterpret_cast<const uint2 *>((const uint16_t *)(A.ln_beta ? A.ln_beta : A.ln_gamma) + 4 * v4);   // RMSNorm: unused
        }
        pld[u] = *reinterpret_cast<const int4 *>(A.load_idx + 4 * v4);
    }
    // stage-1 factor rows of this tile (A operand: lane j = row, 8 consecutive k at 8g + 32S)
    constexpr int K1 = P > Q ? P : Q;                      // upper bound of the stage-1 depth / 1 (for array sizing)
    Frag8 f1h[K1 / 32], f1l[K1 / 32], f2h[K1 / 32], f2l[K1 / 32];
    {
        const uint16_t *m1h = (const uint16_t *)(a_first ? A.M0_hi : A.M1_hi), *m1l = (const uint16_t *)(a_first ? A.M0_lo : A.M1_lo);
        const int d1 = a_first ? P : Q, r1 = a_first ? 16 * at + j : 16 * bt + j;
        const uint32_t o1 = (uint32_t)(r1 * d1 + 8 * g);
        const bool works = wave < (a_first ? NBT : NAT);         // stage 1 has one MFMA tile per wave: the other waves need no factors
#pragma unroll
        for (int S = 0; S < K1 / 32; ++S)
            if (32 * S < d1 && works) {
                f1h[S].u = *reinterpret_cast<const uint4 *>(m1h + o1 + 32 * S);
                f1l[S].u = *reinterpret_cast<const uint4 *>(m1l + o1 + 32 * S);
            }
    }
    int oidx[4] = {0, 0, 0, 0};
    float obias[4] = {0.f, 0.f, 0.f, 0.f};
    uint32_t rres[4] = {0u, 0u, 0u, 0u};
    if (wave == 0) {
        const uint16_t *m2h = (const uint16_t *)(a_first ? A.M1_hi : A.M0_hi), *m2l = (const uint16_t *)(a_first ? A.M1_lo : A.M0_lo);
        const int d2 = a_first ? Q : P, r2 = a_first ? 16 * bt + j : 16 * at + j;
        const uint32_t o2 = (uint32_t)(r2 * d2 + 8 * g);
#pragma unroll
        for (int S = 0; S < K1 / 32; ++S)
            if (32 * S < d2) {
                f2h[S].u = *reinterpret_cast<const uint4 *>(m2h + o2 + 32 * S);
                f2l[S].u = *reinterpret_cast<const uint4 *>(m2l + o2 + 32 * S);
            }
        // the four outputs this lane will hold after stage 2 (D: row = 4g + reg, col = j)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const uint32_t pos = a_first ? (16 * at + j) * Q + 16 * bt + 4 * g + reg : (16 * at + 4 * g + reg) * Q + 16 * bt + j;
            oidx[reg] = store_inv[pos];
        }
    }

    TILE_STAMP(1);
    // ---- LayerNorm (two-pass statistics over the whole row), column scale, split, scatter ------------------------------------
    float4 xv[MAXV];
#pragma unroll
    for (int u = 0; u < MAXV; ++u) xv[u] = raw4_cvt(rx[u], SIDE == 0 ? QUIPAMD_F16 : QUIPAMD_F32);
    if constexpr (LN) {
        const bool rms = A.ln_beta == nullptr;              // RMSNorm (Llama): x * rsqrt(mean(x^2) + eps) * gamma
        float mean = 0.f;
        if (!rms) {
            float s1 = 0.f;
#pragma unroll
            for (int u = 0; u < MAXV; ++u) s1 += (xv[u].x + xv[u].y) + (xv[u].z + xv[u].w);
            mean = block_sum_dpp<TT / 64>(s1, red) / (float)N;
        }
        float s2 = 0.f;
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            const float d0 = xv[u].x - mean, d1 = xv[u].y - mean, d2 = xv[u].z - mean, d3 = xv[u].w - mean;
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        const float rstd = rsqrtf(block_sum_dpp<TT / 64>(s2, red + 8) / (float)N + A.ln_eps);
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            const float4 gm = raw4_cvt(make_uint4(rgm[u].x, rgm[u].y, 0u, 0u), QUIPAMD_F16);
            const float4 bt = rms ? make_float4(0.f, 0.f, 0.f, 0.f) : raw4_cvt(make_uint4(rbt[u].x, rbt[u].y, 0u, 0u), QUIPAMD_F16);
            xv[u] = make_float4((xv[u].x - mean) * rstd * gm.x + bt.x, (xv[u].y - mean) * rstd * gm.y + bt.y,
                                (xv[u].z - mean) * rstd * gm.z + bt.z, (xv[u].w - mean) * rstd * gm.w + bt.w);
        }
    }
    TILE_STAMP(2);
#pragma unroll
    for (int u = 0; u < MAXV; ++u) {
        {
            if constexpr (SIDE == 0) xv[u] = make_float4(xv[u].x * pcs[u].x, xv[u].y * pcs[u].y, xv[u].z * pcs[u].z, xv[u].w * pcs[u].w);
            const float vv[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
            const int pp[4] = {pld[u].x, pld[u].y, pld[u].z, pld[u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int a = pp[e] >> QSH, b = pp[e] & QMASK;
                uint16_t hi, lo;
                split_bf16(vv[e], hi, lo);
                const int off = a_first ? b * P8 + a : a * Q8 + b;          // z^T rows for "mix a", z rows for "mix b"
                Zh[off] = hi;
                Zl[off] = lo;
            }
        }
    }
    __syncthreads();
    TILE_STAMP(3);
    if (wave == 0) {                          // the output indices have landed by now: request the epilogue operands (used after stage 2)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if constexpr (SIDE == 1) obias[reg] = A.bias[(uint32_t)oidx[reg]];
            if constexpr (RES) rres[reg] = ((const uint16_t *)A.residual + row * A.ldo)[(uint32_t)oidx[reg]];
        }
    }

    // ---- stage 1: the tile's 16 rows (mix a) or 16 columns (mix b) over the whole other index, one MFMA tile per wave ---------------
    if (a_first) {
        // D[a = 16at + 4g + reg][b' = 16t + j] = sum_a' M0[a][a'] z[a'][b']
        for (int t = wave; t < NBT; t += TT / 64) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc;
            const int zo = (16 * t + j) * P8 + 8 * g;
#pragma unroll
            for (int S = 0; S < P / 32; ++S) {
                Frag8 bh, bl;
                bh.u = *reinterpret_cast<const uint4 *>(Zh + zo + 32 * S);
                bl.u = *reinterpret_cast<const uint4 *>(Zl + zo + 32 * S);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1l[S].v, bh.v, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1h[S].v, bl.v, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1h[S].v, bh.v, acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                uint16_t hi, lo;
                split_bf16(acc[reg] + (acc1[reg] + acc2[reg]), hi, lo);
                Th[(4 * g + reg) * Q8 + 16 * t + j] = hi;                  // T[a_local][b']
                Tl[(4 * g + reg) * Q8 + 16 * t + j] = lo;
            }
        }
    } else {
        // D[b = 16bt + 4g + reg][a' = 16t + j] = sum_b' M1[b][b'] z[a'][b']
        for (int t = wave; t < NAT; t += TT / 64) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc;
            const int zo = (16 * t + j) * Q8 + 8 * g;
#pragma unroll
            for (int S = 0; S < Q / 32; ++S) {
                Frag8 bh, bl;
                bh.u = *reinterpret_cast<const uint4 *>(Zh + zo + 32 * S);
                bl.u = *reinterpret_cast<const uint4 *>(Zl + zo + 32 * S);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1l[S].v, bh.v, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1h[S].v, bl.v, acc2, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f1h[S].v, bh.v, acc, 0, 0, 0);
            }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                uint16_t hi, lo;
                split_bf16(acc[reg] + (acc1[reg] + acc2[reg]), hi, lo);
                Th[(4 * g + reg) * P8 + 16 * t + j] = hi;                  // T'[b_local][a']
                Tl[(4 * g + reg) * P8 + 16 * t + j] = lo;
            }
        }
    }
    __syncthreads();
    TILE_STAMP(4);
    if (wave != 0) return;

    // ---- stage 2 (one tile) + epilogue ------------------------------------------------------------------------------------------------
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f}, acc1 = acc, acc2 = acc;
    if (a_first) {
        // D[b = 16bt + 4g + reg][a = 16at + j] = sum_b' M1[b][b'] T[a][b']
#pragma unroll
        for (int S = 0; S < Q / 32; ++S) {
            Frag8 bh, bl;
            bh.u = *reinterpret_cast<const uint4 *>(Th + j * Q8 + 8 * g + 32 * S);
            bl.u = *reinterpret_cast<const uint4 *>(Tl + j * Q8 + 8 * g + 32 * S);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2l[S].v, bh.v, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2h[S].v, bl.v, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2h[S].v, bh.v, acc, 0, 0, 0);
        }
    } else {
        // D[a = 16at + 4g + reg][b = 16bt + j] = sum_a' M0[a][a'] T'[b][a']
#pragma unroll
        for (int S = 0; S < P / 32; ++S) {
            Frag8 bh, bl;
            bh.u = *reinterpret_cast<const uint4 *>(Th + j * P8 + 8 * g + 32 * S);
            bl.u = *reinterpret_cast<const uint4 *>(Tl + j * P8 + 8 * g + 32 * S);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2l[S].v, bh.v, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2h[S].v, bl.v, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f2h[S].v, bh.v, acc, 0, 0, 0);
        }
    }
    TILE_STAMP(5);
    float ov[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        float v = acc[reg] + (acc1[reg] + acc2[reg]);
        v = (v + obias[reg]) + (RES ? f16_bits_to_f32((uint16_t)rres[reg]) : 0.f);
        ov[reg] = A.relu ? fmaxf(v, 0.f) : v;
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
    TILE_STAMP(6);
}

template <int P, int Q> int launch_tiles(const TileBatch &B, int nops, int64_t rows, int side, bool flag, hipStream_t s)
{
    const dim3 grid((P / 16) * (Q / 16), (unsigned)nops, (unsigned)rows);
    if (side == 0 && flag) ortho_tile_kernel<P, Q, 0, true><<<grid, TT, 0, s>>>(B);
    else if (side == 0) ortho_tile_kernel<P, Q, 0, false><<<grid, TT, 0, s>>>(B);
    else if (flag) ortho_tile_kernel<P, Q, 1, true><<<grid, TT, 0, s>>>(B);
    else ortho_tile_kernel<P, Q, 1, false><<<grid, TT, 0, s>>>(B);
    QA_LAUNCH_CHECK("quipamd_ortho_apply_tiles");
    return QUIPAMD_OK;
}

}   // namespace

extern "C" int quipamd_ortho_apply_tiles_supported(int p, int q)
{
    return (p == 64 && q == 32) || (p == 64 && q == 64) || (p == 128 && q == 64);
}

// the operand set of one of the compiled variants (see ortho_tile_kernel): 0 = V side, 1 = U^T side, -1 = neither
static int tile_side(const quipamd_small_op &o, const int32_t *inv, bool &flag)
{
    if (!o.load_idx || !o.store_idx || !inv) return -1;
    if (o.x_dtype == QUIPAMD_F16 && o.colscale && !o.bias && !o.residual && !o.relu && (!o.ln_gamma || o.ln_dtype == QUIPAMD_F16)) {
        flag = o.ln_gamma != nullptr;
        return 0;
    }
    if (o.x_dtype == QUIPAMD_F32 && !o.colscale && o.bias && !o.ln_gamma && (!o.residual || o.res_dtype == QUIPAMD_F16)) {
        flag = o.residual != nullptr;
        return 1;
    }
    return -1;
}

extern "C" int quipamd_ortho_apply_tiles(const quipamd_small_op *ops, const int32_t *const *store_inv, int nops, int64_t rows, void *stream)
{
    QA_REQUIRE(ops && nops >= 1 && nops <= QUIPAMD_SMALL_MAX_OPS, QUIPAMD_ERR_ARG, "ortho_apply_tiles: 1..%d ops", QUIPAMD_SMALL_MAX_OPS);
    const int p = ops[0].p, q = ops[0].q;
    QA_REQUIRE(quipamd_ortho_apply_tiles_supported(p, q), QUIPAMD_ERR_UNSUPPORTED,
               "ortho_apply_tiles: p x q = %d x %d is not one of 64x32, 64x64, 128x64; use quipamd_ortho_apply_small_ops", p, q);
    QA_REQUIRE(rows >= 0 && rows <= 65535, QUIPAMD_ERR_SHAPE, "ortho_apply_tiles: bad row count");
    if (rows == 0) return QUIPAMD_OK;                           // an empty batch has null data pointers: nothing to check, nothing to do
    TileBatch B;
    int side = -2;
    bool flag = false;
    for (int i = 0; i < nops; ++i) {
        const quipamd_small_op &o = ops[i];
        QA_REQUIRE(o.p == p && o.q == q, QUIPAMD_ERR_ARG, "ortho_apply_tiles: ops of one launch must share p and q (op %d differs)", i);
        QA_REQUIRE(o.M0_hi && o.M0_lo && o.M1_hi && o.M1_lo && o.x && o.out, QUIPAMD_ERR_ARG,
                   "ortho_apply_tiles: op %d needs x, out and the four split-bf16 factor arrays", i);
        QA_REQUIRE(o.ldx >= (int64_t)p * q && o.ldo >= (int64_t)p * q && o.ldx % 4 == 0, QUIPAMD_ERR_SHAPE, "ortho_apply_tiles: leading dimensions");
        bool f = false;
        const int sd = tile_side(o, store_inv ? store_inv[i] : nullptr, f);
        QA_REQUIRE(sd >= 0, QUIPAMD_ERR_UNSUPPORTED,
                   "ortho_apply_tiles: op %d is neither the activation-side form (x f16, colscale, permutations, [f16 LayerNorm]) nor the "
                   "output-side form (x f32, bias, permutations, [f16 residual]); use quipamd_ortho_apply_small_ops", i);
        QA_REQUIRE(side == -2 || (sd == side && f == flag), QUIPAMD_ERR_ARG, "ortho_apply_tiles: ops of one launch must have the same operand set");
        side = sd;
        flag = f;
        B.op[i] = o;
        B.store_inv[i] = store_inv[i];
    }
    for (int i = nops; i < QUIPAMD_SMALL_MAX_OPS; ++i) { B.op[i] = ops[0]; B.store_inv[i] = B.store_inv[0]; }
    if (rows == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (p == 64 && q == 32) return launch_tiles<64, 32>(B, nops, rows, side, flag, s);
    if (p == 64 && q == 64) return launch_tiles<64, 64>(B, nops, rows, side, flag, s);
    return launch_tiles<128, 64>(B, nops, rows, side, flag, s);
}
