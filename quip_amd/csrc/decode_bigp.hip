// decode_bigp.hip -- the decode step around a packed layer whose Kronecker operator is p x 16 with a LARGE p: Llama's intermediate size
// 11008 = 688 x 16 (method.py:16-18 butterfly_factors), i.e. the MLP tail of llama.py:418-471's benchmark() loop
//     g = U_gate^T y_gate,  u = U_up^T y_up,   t = silu(g) * u (/) s_down,   x~ = V_down t,   y_down = What_down x~.
// A 688 x 688 factor is 0.95 MB in fp16: not a workgroup's pass, so the two operators are cut where they are all-to-all (the mix over
// the p index) and nowhere else.  Round 2 ran this tail as four launches (ortho_bigp.hip twice at ~10 us -- fp32 factors, int32 index
// vectors and a 2-/4-byte LDS scatter of the whole row in EVERY workgroup --, the bf16 tile GEMM at 8.6 us, a cast); here it is two:
//
//   bigp_u_kernel       grid (p / 16, layers):  out[dest[pos]] = ((M0 z M1^T)[pos] + bias[pos]) * post[pos]     for pos in 16 image rows
//   bigp_v_gemm_kernel  grid (p / 16, row groups):  x~[16 image rows] = M0 silu(g) * u M1^T  ->  y += What[rows, those 256 columns] x~
//
// What makes them short:
//   * the vector a pass starts from arrives as the TRANSPOSED image, row-major (index b * p + a): the producing GEMM has its rows
//     packed in that order (include/quip_amd.h "permutations folded into the packing"), bigp_u writes its output through a uint16
//     table the host composes from U's store permutation and V_down's load permutation.  A lane's A fragment of the mix over a is then
//     ONE 16-byte global load (8 consecutive a of one b) -- no index vector, no LDS image, no scatter;
//   * the factor rows of the workgroup's 16 outputs come as fp16 MFMA B fragments (host order, zero padded to 32-deep steps):
//     22 KiB per workgroup at p = 688, one 16-byte load per lane per step; v_mfma_f32_16x16x32_f16, K = p split over the waves (two
//     steps each), the partial tiles meet in LDS as float4 per lane;
//   * q = 16 is one tile: the reduced tile IS the B operand of the second mix (four v_mfma_f32_16x16x4_f32 against M1, fp32);
//   * everything is requested in the first instructions of the kernel, weights last (in-order vmcnt, see decode_fused.hip);
//   * bigp_v_gemm: a workgroup's 16 image rows are 256 consecutive columns of the packed matrix (columns in image order) = one STREAM
//     chunk of every row tile, so the operator output goes straight into the 2-bit GEMM of that K-slice; the K-slices meet in y
//     through fp32 atomics (64 consecutive floats per instruction).  y must be ZERO on entry: bigp_u clears it (`clear`) -- it runs
//     between y's previous reader and this launch.  With `partials` set the slices meet in a FIXED order instead (each stores its partial,
//     a small second launch sums slices 0 .. p/16 - 1): bit-identical runs, and faster than the atomics from 5 rows on.
// Rows (batch): 1..4 compile-time; round 5: up to 16 -- bigp_u walks row groups of 4 over blockIdx.z, bigp_v_gemm mixes 4 rows at a
// time into a 16-row x~ image and runs ONE weight pass against all 16 MFMA columns (templates BS = 8, 16).
#include "common.h"
#include "dq_common.h"
#include "fpass.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

constexpr int BG_MAXG = 3, BG_MAXBS = 4, BG_MAXROWS = 16, BG_MAXKS = 32;

struct BUOp {
    const uint4 *F0;              // [p/16][ks][64] B fragments of M0 (fp16, zero for a >= p)
    const float *M1;              // [16][16]: z2 = M0 z M1^T
    const uint16_t *y;            // f16 [bs][n] transposed image (b p + a)
    const uint16_t *bias;         // f16 [n] image order, or null
    const float *post;            // f32 [n] image order, or null
    const uint16_t *dest;         // [n] image position -> output index
    uint16_t *out;                // f16 [bs][ldo]
    int64_t ldo;
};
struct BUArgs {
    BUOp op[BG_MAXG];
    float *clear;
    int64_t clear_n4;             // float4 count
    int p, ks;
    int rows;                     // batch rows that exist (row groups of BS over grid.z; the last group may be ragged)
};

// reduced partial tiles -> T (in the lane's D registers: b = 4g + s, a' = j) -> z2[a' = j][b' = 4g + reg] = sum_b T[a'][b] M1[b'][b]:
// D[row = b'][col = a'], A = M1 rows (lane j = b', k = b), B = T (lane j = a', k = b) -- k = 4g + s over the four instructions
__device__ __forceinline__ f32x4_t bg_mix_b(const f32x4_t &T, const float4 &m1)
{
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m1.x, T[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m1.y, T[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m1.z, T[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(m1.w, T[3], acc, 0, 0, 0);
    return acc;
}

template <int BS>
__global__ __launch_bounds__(1024) void bigp_u_kernel(BUArgs G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *part = reinterpret_cast<float4 *>(smem);                            // [BS][nw][64]
    const BUOp &O = G.op[blockIdx.y];
    asm volatile("" ::"s"(O.F0), "s"(O.M1), "s"(O.y), "s"(O.bias), "s"(O.post), "s"(O.dest), "s"(O.out), "s"(O.ldo), "s"(G.p), "s"(G.ks),
                 "s"(G.clear), "s"(G.clear_n4));
    const int row0 = (int)blockIdx.z * BS;                                      // more than 4 rows: groups of BS rows side by side (grid.z)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int at = blockIdx.x, p = G.p, ks = G.ks;
    const int nw = (ks + 1) >> 1;                                               // waves that run the mix over a (the block has max(nw, BS) waves)
    const bool mixer = wave < nw;
    const int64_t n = (int64_t)p * 16;

    // ---- requests: the row's A fragments, the factor fragments, then what the finishing waves need ------------------------------------
    const int S0 = 2 * wave, S1 = 2 * wave + 1;
    const bool two = S1 < ks;
    int ka0 = 32 * S0 + 8 * g, ka1 = 32 * S1 + 8 * g;
    ka0 = ka0 < p ? ka0 : 0;                                                    // (the factor fragment is zero there; any finite value of the row serves)
    ka1 = ka1 < p ? ka1 : 0;
    uint4 ya[BS][2], fb[2];
    fb[0] = fb[1] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int r = 0; r < BS; ++r) ya[r][0] = ya[r][1] = make_uint4(0u, 0u, 0u, 0u);
    if (mixer) {
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            const int rr = row0 + r < G.rows ? row0 + r : G.rows - 1;           // (a ragged last group re-reads the last row; not stored)
            const uint16_t *row = O.y + (int64_t)rr * n + (uint32_t)(j * p);
            ya[r][0] = *reinterpret_cast<const uint4 *>(row + ka0);
            ya[r][1] = *reinterpret_cast<const uint4 *>(row + ka1);
        }
        fb[0] = O.F0[(uint32_t)((at * ks + S0) * 64 + lane)];
        fb[1] = O.F0[(uint32_t)((at * ks + (two ? S1 : S0)) * 64 + lane)];
    }
    const uint32_t pos0 = (uint32_t)((16 * at + j) * 16 + 4 * g);              // this lane's 4 results: image positions pos0 .. pos0 + 3
    float4 m1 = make_float4(0.f, 0.f, 0.f, 0.f), po = make_float4(1.f, 1.f, 1.f, 1.f);
    uint2 bi = make_uint2(0u, 0u), de = make_uint2(0u, 0u);
    if (wave < BS) {
        m1 = *reinterpret_cast<const float4 *>(O.M1 + j * 16 + 4 * g);
        de = *reinterpret_cast<const uint2 *>(O.dest + pos0);
        if (O.bias) bi = *reinterpret_cast<const uint2 *>(O.bias + pos0);
        if (O.post) po = *reinterpret_cast<const float4 *>(O.post + pos0);
    }
    if (G.clear && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t i = tid; i < G.clear_n4; i += blockDim.x) reinterpret_cast<float4 *>(G.clear)[i] = z;
    }
    if (!two) fb[1] = make_uint4(0u, 0u, 0u, 0u);

    // ---- mix over a: D[row = b][col = a'] = sum_a zT[b][a] M0[a'][a], this wave's two 32-deep steps --------------------------------------
    if (mixer) {
#pragma unroll
        for (int r = 0; r < BS; ++r) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = ActF16::mfma(u32x4{ya[r][0].x, ya[r][0].y, ya[r][0].z, ya[r][0].w}, u32x4{fb[0].x, fb[0].y, fb[0].z, fb[0].w}, acc);
            acc = ActF16::mfma(u32x4{ya[r][1].x, ya[r][1].y, ya[r][1].z, ya[r][1].w}, u32x4{fb[1].x, fb[1].y, fb[1].z, fb[1].w}, acc);
            part[(r * nw + wave) * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
    __syncthreads();
    if (wave >= BS || row0 + wave >= G.rows) return;
    const int r = wave;
    f32x4_t T = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < nw; ++w) {
        const float4 v = part[(r * nw + w) * 64 + lane];
        T[0] += v.x; T[1] += v.y; T[2] += v.z; T[3] += v.w;
    }
    const f32x4_t z2 = bg_mix_b(T, m1);
    const float4 bb = f16x4_to_f32(bi);
    uint16_t *o = O.out + (int64_t)(row0 + r) * O.ldo;
    o[de.x & 0xffff] = f32_to_f16_bits((z2[0] + bb.x) * po.x);
    o[de.x >> 16] = f32_to_f16_bits((z2[1] + bb.y) * po.y);
    o[de.y & 0xffff] = f32_to_f16_bits((z2[2] + bb.z) * po.z);
    o[de.y >> 16] = f32_to_f16_bits((z2[3] + bb.w) * po.w);
}

struct BVArgs {
    const uint4 *F0;              // V's M0 as B fragments
    const float *M1;
    const uint16_t *gate, *up;    // f16 [bs][ldx] transposed image of V's input (GATE: t = silu(gate) * up, else t = gate)
    int64_t ldx;
    const uint4 *qw;              // 2-bit STREAM codes, columns in image order of V
    const float *scale;
    float *y;                     // fp32 [bs][m], accumulated
    int64_t m;
    int p, ks;
    int rows;                     // batch rows that exist (<= BS; rows >= it read row rows - 1 and are not stored)
    float *partials;              // fixed-order meet: [p/16 slices][rows][m] fp32, or null (atomics)
    uint16_t *xt_out;             // MIX ONLY (grid (p/16, row groups of BS)): x~ f16 [rows][16 p] in image order; no weights, no GEMM
};

__device__ __forceinline__ uint32_t bg_gate2(uint32_t g2, uint32_t u2)
{
    // silu(g) * up, rounded to f16 like the two torch launches it replaces (F.silu rounds, then the product rounds)
    auto one = [](float gv, float uv) {
        const float sl = f16_bits_to_f32(f32_to_f16_bits(gv / (1.f + __expf(-gv))));
        return sl * uv;
    };
    return pack_f16x2(one(f16_bits_to_f32(g2 & 0xffff), f16_bits_to_f32(u2 & 0xffff)), one(f16_bits_to_f32(g2 >> 16), f16_bits_to_f32(u2 >> 16)));
}

// grid = (p / 16 K-slices, m / (256 NRT) row groups); 1024 threads: wave w owns row tiles (16 blockIdx.y + w) NRT + k of the K-slice
// BITS 4: the 4-bit STREAM container (--wbits 4, and 3 with maxq = 7): the slice's 256 columns are TWO 1 KiB tiles per row tile, uniform-offset
// conversion (value = 16 + code; no per-field offsets to subtract)
template <bool ME> struct BgDq : DeqME2<ActF16> {};
template <> struct BgDq<false> : DeqT<4, ActF16> {
    struct Consts { };
    static __device__ __forceinline__ Consts make_consts() { return Consts{}; }
    static __device__ __forceinline__ u32x4 frag(const u32x4 &w, int t, const Consts &) { return DeqT<4, ActF16>::frag(w, t); }
};

template <int BS, int NRT, bool GATE, int BITS = 2>
__global__ __launch_bounds__(1024) void bigp_v_gemm_kernel(BVArgs G, float two_over_maxq, float c0)
{
    typedef BgDq<BITS == 2> DQ;                                                 // 2 bits: multi-exponent dequantisation (dq_common.h): needs sum OFF_k x~_k
    constexpr int TPS = BITS == 2 ? 1 : 2;                                       // 1 KiB tiles per (row tile, K-slice of 256 columns)
    constexpr int XTS = 256 + 8;
    // BS <= 4: as rounds 3-4.  BS = 8 / 16 (round 5): the mix over a runs RG = 4 rows at a time (the registers hold four rows' fragments),
    // NG passes fill an XR-row x~ image, then ONE pass over the weights feeds all XR MFMA columns
    constexpr int RG = BS < 4 ? BS : 4, NG = BS / RG, XR = BS <= 4 ? BG_MAXBS : BG_MAXROWS;
    static_assert(BS == RG * NG && (BS <= 4 || BS == 8 || BS == 16), "rows per launch: 1..4, 8, 16");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *XT = reinterpret_cast<uint16_t *>(smem);                          // [XR][256 + 8] f16: x~ of the slice, k = 16 a_local + b
    float *red = reinterpret_cast<float *>(smem + XR * XTS * 2);                 // [XR] sum x~, [XR] sum OFF x~
    float *park = red + 2 * XR;                                                 // [16 waves][NRT][XR][16]
    const typename DQ::Consts qc = DQ::make_consts();
    float4 *part = reinterpret_cast<float4 *>(park + 16 * NRT * XR * 16);       // [RG][nwp][64]
    asm volatile("" ::"s"(G.F0), "s"(G.M1), "s"(G.gate), "s"(G.up), "s"(G.ldx), "s"(G.qw), "s"(G.scale), "s"(G.y), "s"(G.m), "s"(G.p), "s"(G.ks), "s"(G.rows));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const int at = blockIdx.x, p = G.p, ks = G.ks, nch = p >> 4;
    const int nwp = (ks + 1) >> 1;                                              // waves that run the mix over a
    const uint32_t rt0 = (blockIdx.y * 16 + wave) * NRT;
    const int rows = G.rows;
    const bool mixonly = G.xt_out != nullptr;                                   // (uniform) blockIdx.y is then a group of BS batch rows
    const int rowbase = mixonly ? (int)blockIdx.y * BS : 0;

    const int S0 = 2 * wave, S1 = 2 * wave + 1;
    const bool mixer = wave < nwp, two = S1 < ks;
    int ka0 = 32 * S0 + 8 * g, ka1 = 32 * S1 + 8 * g;
    ka0 = ka0 < p ? ka0 : 0;
    ka1 = ka1 < p ? ka1 : 0;
    uint4 ga[RG][2], ua[RG][2], fb[2];
    fb[0] = fb[1] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int r = 0; r < RG; ++r) ga[r][0] = ga[r][1] = ua[r][0] = ua[r][1] = make_uint4(0u, 0u, 0u, 0u);
    auto load_rows = [&](int r0) {                                              // the A fragments of rows r0 .. r0 + RG - 1 (rows past the batch: the last one)
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const int rr = rowbase + r0 + r < rows ? rowbase + r0 + r : rows - 1;
            const uint16_t *grow = G.gate + (int64_t)rr * G.ldx + (uint32_t)(j * p);
            ga[r][0] = *reinterpret_cast<const uint4 *>(grow + ka0);
            ga[r][1] = *reinterpret_cast<const uint4 *>(grow + ka1);
            if (GATE) {
                const uint16_t *urow = G.up + (int64_t)rr * G.ldx + (uint32_t)(j * p);
                ua[r][0] = *reinterpret_cast<const uint4 *>(urow + ka0);
                ua[r][1] = *reinterpret_cast<const uint4 *>(urow + ka1);
            }
        }
    };
    if (mixer) {
        load_rows(0);
        fb[0] = G.F0[(uint32_t)((at * ks + S0) * 64 + lane)];
        fb[1] = G.F0[(uint32_t)((at * ks + (two ? S1 : S0)) * 64 + lane)];
    }
    float4 m1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wave < RG) m1 = *reinterpret_cast<const float4 *>(G.M1 + j * 16 + 4 * g);
    uint4 w[NRT][TPS];
    float e_sc = 0.f;
    if (!mixonly) {
#pragma unroll
        for (int k = 0; k < NRT; ++k)                                            // HBM, streamed once: nt; requested LAST (in-order vmcnt)
#pragma unroll
            for (int h = 0; h < TPS; ++h) {
                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(G.qw + ((uint64_t)(rt0 + k) * (nch * TPS) + at * TPS + h) * 64 + lane));
                w[k][h] = make_uint4(t[0], t[1], t[2], t[3]);
            }
        e_sc = G.scale[0];
    } else {
#pragma unroll
        for (int k = 0; k < NRT; ++k)
#pragma unroll
            for (int h = 0; h < TPS; ++h) w[k][h] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (!two) fb[1] = make_uint4(0u, 0u, 0u, 0u);

#pragma unroll
    for (int gq = 0; gq < NG; ++gq) {
        if (gq > 0) {
            __syncthreads();                                                    // the previous group's finishing waves are done with `part`
            if (mixer) load_rows(gq * RG);
        }
        if (mixer) {
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                u32x4 a0 = {ga[r][0].x, ga[r][0].y, ga[r][0].z, ga[r][0].w}, a1 = {ga[r][1].x, ga[r][1].y, ga[r][1].z, ga[r][1].w};
                if (GATE) {
                    a0 = u32x4{bg_gate2(ga[r][0].x, ua[r][0].x), bg_gate2(ga[r][0].y, ua[r][0].y), bg_gate2(ga[r][0].z, ua[r][0].z), bg_gate2(ga[r][0].w, ua[r][0].w)};
                    a1 = u32x4{bg_gate2(ga[r][1].x, ua[r][1].x), bg_gate2(ga[r][1].y, ua[r][1].y), bg_gate2(ga[r][1].z, ua[r][1].z), bg_gate2(ga[r][1].w, ua[r][1].w)};
                }
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
                acc = ActF16::mfma(a0, u32x4{fb[0].x, fb[0].y, fb[0].z, fb[0].w}, acc);
                acc = ActF16::mfma(a1, u32x4{fb[1].x, fb[1].y, fb[1].z, fb[1].w}, acc);
                part[(r * nwp + wave) * 64 + lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            }
        }
        __syncthreads();
        if (wave < RG) {
            const int r = wave, rx = gq * RG + r;                                // rx: the row of the x~ image
            f32x4_t T = {0.f, 0.f, 0.f, 0.f};
            for (int w2 = 0; w2 < nwp; ++w2) {
                const float4 v = part[(r * nwp + w2) * 64 + lane];
                T[0] += v.x; T[1] += v.y; T[2] += v.z; T[3] += v.w;
            }
            const f32x4_t z2 = bg_mix_b(T, m1);                                  // x~[a' = j][b' = 4g + reg]: k = 16 j + 4 g + reg of the slice
            uint2 pk;
            pk.x = pack_f16x2(z2[0], z2[1]);
            pk.y = pack_f16x2(z2[2], z2[3]);
            *reinterpret_cast<uint2 *>(XT + rx * XTS + 16 * j + 4 * g) = pk;
            if (mixonly && rowbase + rx < rows)                                 // the slice of x~ leaves for the dequant-GEMM launch behind this one
                *reinterpret_cast<uint2 *>(G.xt_out + (int64_t)(rowbase + rx) * ((int64_t)p * 16) + (uint32_t)(at * 256 + 16 * j + 4 * g)) = pk;
            const float4 rv = f16x4_to_f32(pk);                                  // the sums the epilogue subtracts are sums of what the MFMAs see
            const float s = fg_wave_sum((rv.x + rv.y) + (rv.z + rv.w));
            const int k0 = 16 * j + 4 * g;
            const float so = BITS == 2 ? fg_wave_sum(fmaf(me2_off_f16(k0), rv.x, fmaf(me2_off_f16(k0 + 1), rv.y, fmaf(me2_off_f16(k0 + 2), rv.z, me2_off_f16(k0 + 3) * rv.w))))
                                       : 0.f;                                    // 4 bits: one offset for every field, folded into c0
            if (lane == 0) {
                red[rx] = s;
                red[XR + rx] = so;
            }
        }
    }
    if (mixonly) return;
    __syncthreads();

    // ---- 2-bit GEMM of this K-slice: MFMA column j = batch row j (columns >= BS read allocated garbage and are not stored) ----------------
    f32x4_t acc[NRT];
#pragma unroll
    for (int k = 0; k < NRT; ++k) acc[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const uint16_t *xrow = XT + (j & (XR - 1)) * XTS + 8 * g;
    uint4 xf[8];                                                                // the slice's 256 k = 8 MFMA steps, whatever the container
#pragma unroll
    for (int t = 0; t < 8; ++t) xf[t] = *reinterpret_cast<const uint4 *>(xrow + 32 * t);
#pragma unroll
    for (int k = 0; k < NRT; ++k)
#pragma unroll
        for (int h = 0; h < TPS; ++h)
#pragma unroll
            for (int t = 0; t < DQ::NT; ++t) {
                const u32x4 a = DQ::frag(u32x4{w[k][h].x, w[k][h].y, w[k][h].z, w[k][h].w}, t, qc);
                const uint4 &xv = xf[h * DQ::NT + t];
                acc[k] = ActF16::mfma(a, u32x4{xv.x, xv.y, xv.z, xv.w}, acc[k]);
            }
    // D[row = 4g + reg][col = j]: lanes j < BS park their 4 rows; the wave re-reads them as 16 NRT consecutive rows per batch row
    float *mine = park + wave * (NRT * XR * 16);
    if (j < BS) {
#pragma unroll
        for (int k = 0; k < NRT; ++k) *reinterpret_cast<float4 *>(mine + (k * XR + j) * 16 + 4 * g) = make_float4(acc[k][0], acc[k][1], acc[k][2], acc[k][3]);
    }
    __syncthreads();
    const float alpha = e_sc * two_over_maxq;
    if (G.partials == nullptr) {
#pragma unroll
        for (int o = 0; o < (16 * NRT + 63) / 64; ++o) {
            const int l = lane + 64 * o;
            if (l < 16 * NRT) {
                const int k = l >> 4, wr = l & 15;
#pragma unroll
                for (int r = 0; r < BS; ++r) {
                    const float val = alpha * ((mine[(k * XR + r) * 16 + wr] - red[XR + r]) - c0 * red[r]);
                    if (BS <= 4 || r < rows) unsafeAtomicAdd(G.y + (int64_t)r * G.m + (int64_t)rt0 * 16 + l, val);
                }
            }
        }
        return;
    }
    // ---- fixed-order meet: every K-slice STORES its partial (64 consecutive floats per instruction, no atomics); bigp_reduce_kernel, the
    // next launch of the same call, sums the slices 0 .. p/16 - 1 in that order.  Bit-identical runs -- and, from 5 rows on, FASTER than the
    // atomics: 43 slices x 16 rows x 4096 outputs are 2.8 M atomic adds that resolve outside the L2s (49 us at 16 rows, 7.4 at one row;
    // profiles/r05e_bigp_tail.jsonl) against 11 MB of plain stores and one pass over them.
    float *slab = G.partials + (int64_t)at * rows * G.m;
#pragma unroll
    for (int o = 0; o < (16 * NRT + 63) / 64; ++o) {
        const int l = lane + 64 * o;
        if (l < 16 * NRT) {
            const int k = l >> 4, wr = l & 15;
#pragma unroll
            for (int r = 0; r < BS; ++r) {
                const float val = alpha * ((mine[(k * XR + r) * 16 + wr] - red[XR + r]) - c0 * red[r]);
                if (BS <= 4 || r < rows) slab[(int64_t)r * G.m + (int64_t)rt0 * 16 + l] = val;
            }
        }
    }
}

// y[r][c] = sum over slices sl = 0 .. nsl - 1, in that order, of partials[sl][r][c]      (n = rows * m floats per slice, n % 4 == 0)
__global__ __launch_bounds__(256) void bigp_reduce_kernel(const f32x4_t *__restrict__ partials, f32x4_t *__restrict__ y, int64_t n4, int nsl)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4_t *p = partials + i;
    int sl = 0;
    for (; sl + 8 <= nsl; sl += 8) {                                             // eight loads in flight, summed in slice order
        f32x4_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + (int64_t)(sl + u) * n4);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; sl < nsl; ++sl) acc += __builtin_nontemporal_load(p + (int64_t)sl * n4);
    y[i] = acc;
}

}   // namespace

extern "C" int quipamd_decode_bigp_supported(int p, int q)
{
    return q == 16 && p % 16 == 0 && p >= 64 && (p + 31) / 32 <= BG_MAXKS;
}

extern "C" int quipamd_decode_bigp_u(const quipamd_bigp_u_op *ops, int nops, int p, int64_t rows, float *clear, int64_t clear_n, void *stream)
{
    QA_REQUIRE(ops && nops >= 1 && nops <= BG_MAXG, QUIPAMD_ERR_ARG, "decode_bigp_u: 1..%d operators", BG_MAXG);
    QA_REQUIRE(quipamd_decode_bigp_supported(p, 16), QUIPAMD_ERR_UNSUPPORTED, "decode_bigp_u: p = %d (wants p %% 16 == 0, 64 <= p <= %d)", p, 32 * BG_MAXKS);
    QA_REQUIRE(rows >= 1 && rows <= BG_MAXROWS, QUIPAMD_ERR_SHAPE, "decode_bigp_u: 1..%d rows", BG_MAXROWS);
    QA_REQUIRE((!clear && clear_n == 0) || (clear && clear_n > 0 && clear_n % 4 == 0 && ((uintptr_t)clear & 15) == 0), QUIPAMD_ERR_ARG,
               "decode_bigp_u: clear wants a 16-byte aligned buffer of a multiple of 4 floats");
    BUArgs A;
    const int64_t n = (int64_t)p * 16;
    for (int i = 0; i < BG_MAXG; ++i) {
        const quipamd_bigp_u_op &o = ops[i < nops ? i : 0];
        QA_REQUIRE(o.F0 && o.M1 && o.y && o.dest && o.out && o.ld_out >= n, QUIPAMD_ERR_ARG, "decode_bigp_u: operator %d: fragments, M1, y, dest, out wanted", i);
        A.op[i] = BUOp{(const uint4 *)o.F0, o.M1, (const uint16_t *)o.y, (const uint16_t *)o.bias_img, o.post_img, o.dest, (uint16_t *)o.out, o.ld_out};
    }
    A.clear = clear;
    A.clear_n4 = clear_n / 4;
    A.p = p;
    A.ks = (p + 31) / 32;
    A.rows = (int)rows;
    const int nw = (A.ks + 1) / 2;
    const int rpg = rows <= BG_MAXBS ? (int)rows : BG_MAXBS;                       // rows per workgroup; more: row groups over grid.z
    const size_t lds = (size_t)rpg * nw * 64 * sizeof(float4);
    const dim3 grid((unsigned)(p / 16), (unsigned)nops, (unsigned)((rows + rpg - 1) / rpg));
    hipStream_t s = (hipStream_t)stream;
#define QA_BU(BS)                                                                                                                    \
    do {                                                                                                                             \
        auto kern = bigp_u_kernel<BS>;                                                                                               \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return qa_fail(QUIPAMD_ERR_LAUNCH, "decode_bigp_u: cannot raise dynamic LDS to %zu", lds);                               \
        kern<<<grid, 64 * (nw > BS ? nw : BS), lds, s>>>(A);                                                                                          \
    } while (0)
    switch (rpg) {
    case 1: QA_BU(1); break;
    case 2: QA_BU(2); break;
    case 3: QA_BU(3); break;
    default: QA_BU(4); break;
    }
#undef QA_BU
    QA_LAUNCH_CHECK("quipamd_decode_bigp_u");
    return QUIPAMD_OK;
}

extern "C" int quipamd_decode_bigp_v_gemm(const quipamd_bigp_v_gemm_args *a, void *stream)
{
    QA_REQUIRE(a, QUIPAMD_ERR_ARG, "decode_bigp_v_gemm: null args");
    const int p = a->p;
    QA_REQUIRE(quipamd_decode_bigp_supported(p, 16), QUIPAMD_ERR_UNSUPPORTED, "decode_bigp_v_gemm: p = %d (wants p %% 16 == 0, 64 <= p <= %d)", p, 32 * BG_MAXKS);
    QA_REQUIRE(a->bits >= 2 && a->bits <= 4, QUIPAMD_ERR_UNSUPPORTED, "decode_bigp_v_gemm: 2-, 3- or 4-bit qfn-b codes (bits = %d)", a->bits);
    const bool w4 = a->bits != 2;                                                // 3-bit codes ride in the 4-bit container (maxq = 7)
    const float maxq = (float)((1 << a->bits) - 1);
    QA_REQUIRE(a->rows >= 1 && a->rows <= BG_MAXROWS, QUIPAMD_ERR_SHAPE, "decode_bigp_v_gemm: 1..%d rows", BG_MAXROWS);
    QA_REQUIRE(a->F0 && a->M1 && a->gate && a->qweight && a->scale && a->y && a->ldx >= (int64_t)p * 16 && a->ldx % 8 == 0, QUIPAMD_ERR_ARG,
               "decode_bigp_v_gemm: fragments, M1, input, codes, scale, y wanted; ldx >= n, ldx %% 8 == 0");
    QA_REQUIRE(a->m > 0 && a->m % 256 == 0, QUIPAMD_ERR_SHAPE, "decode_bigp_v_gemm: m %% 256 == 0");
    int nrt = a->row_tiles_per_wave;
    if (nrt == 0) nrt = a->m % 1024 == 0 ? 4 : a->m % 512 == 0 ? 2 : 1;
    if (w4 && a->rows > 4 && nrt == 4) nrt = 2;                                  // (registers: see QA_BV)
    QA_REQUIRE((nrt == 1 || nrt == 2 || nrt == 4) && a->m % (256 * nrt) == 0, QUIPAMD_ERR_ARG, "decode_bigp_v_gemm: row_tiles_per_wave 0 / 1 / 2 / 4 with m %% (256 x it) == 0");
    QA_REQUIRE(!a->partials || (((uintptr_t)a->partials | (uintptr_t)a->y) & 15) == 0, QUIPAMD_ERR_ARG, "decode_bigp_v_gemm: partials and y 16-byte aligned");
    BVArgs A{(const uint4 *)a->F0, a->M1, (const uint16_t *)a->gate, (const uint16_t *)a->up, a->ldx, (const uint4 *)a->qweight, a->scale, a->y, a->m, p, (p + 31) / 32,
             (int)a->rows, a->partials, nullptr};
    const int nwp = (A.ks + 1) / 2;
    int bs = a->rows <= 4 ? (int)a->rows : a->rows <= 8 ? 8 : 16;                  // the kernel's row count: 1..4, 8, 16
    // c0 = maxq / 2 (2 bits: the per-field offsets are subtracted as sum OFF_k x~_k; 4-bit container: + the uniform offset 16)
    const float two_over_maxq = 2.0f / maxq, c0 = 0.5f * maxq + (w4 ? 16.0f : 0.0f);
    dim3 grid((unsigned)(p / 16), (unsigned)(a->m / (256 * nrt)));
    hipStream_t s = (hipStream_t)stream;
    // Two-launch form (xt scratch given; the Python side picks it from 5 rows on): the operator pass ALONE -- one workgroup per (16 image
    // rows, group of 4 batch rows) writes its slice of x~ -- and then the ordinary dequant-GEMM on the same decode-order codes.  In the
    // one-launch form every workgroup redoes silu(gate) * up and the mix over a for ALL of its rows (44 KB of input and 11008 gate values
    // per row), m / 1024 times per K-slice: 7.4 us at one row, 49 us at 16 (profiles/r05f_bigp_tail.jsonl) -- the atomics were not it.
    const bool two_launch = a->xt != nullptr;
    if (two_launch) {
        QA_REQUIRE(((uintptr_t)a->xt & 15) == 0, QUIPAMD_ERR_ARG, "decode_bigp_v_gemm: xt 16-byte aligned");
        A.xt_out = (uint16_t *)a->xt;
        A.partials = nullptr;
        bs = a->rows < 4 ? (int)a->rows : 4;
        nrt = 1;
        grid = dim3((unsigned)(p / 16), (unsigned)((a->rows + bs - 1) / bs));
    }
#define QA_BV(BS, NRT, GT)                                                                                                           \
    do {                                                                                                                             \
        /* (the 4-bit container with 16 rows AND 4 row tiles per wave does not fit the register file -- 20 bytes of scratch; the host  \
            picks 2 tiles per wave there, so that combination is never instantiated) */                                               \
        constexpr int B4_ = (BS > 4 && NRT == 4) ? 2 : 4;                                                                            \
        auto kern = w4 ? bigp_v_gemm_kernel<BS, NRT, GT, B4_> : bigp_v_gemm_kernel<BS, NRT, GT, 2>;                                  \
        constexpr int XR_ = BS <= 4 ? BG_MAXBS : BG_MAXROWS, RG_ = BS < 4 ? BS : 4;                                                   \
        const size_t lds = (size_t)XR_ * 264 * 2 + 2 * XR_ * 4 + (size_t)16 * NRT * XR_ * 16 * 4 + (size_t)RG_ * nwp * 64 * sizeof(float4); \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return qa_fail(QUIPAMD_ERR_LAUNCH, "decode_bigp_v_gemm: cannot raise dynamic LDS to %zu", lds);                          \
        kern<<<grid, 1024, lds, s>>>(A, two_over_maxq, c0);                                                                          \
    } while (0)
#define QA_BV_N(BS, GT)                                                                                                              \
    do {                                                                                                                             \
        if (nrt == 4) QA_BV(BS, 4, GT);                                                                                              \
        else if (nrt == 2) QA_BV(BS, 2, GT);                                                                                         \
        else QA_BV(BS, 1, GT);                                                                                                       \
    } while (0)
#define QA_BV_B(GT)                                                                                                                  \
    do {                                                                                                                             \
        switch (bs) {                                                                                                                \
        case 1: QA_BV_N(1, GT); break;                                                                                               \
        case 2: QA_BV_N(2, GT); break;                                                                                               \
        case 3: QA_BV_N(3, GT); break;                                                                                               \
        case 4: QA_BV_N(4, GT); break;                                                                                               \
        case 8: QA_BV_N(8, GT); break;                                                                                               \
        default: QA_BV_N(16, GT); break;                                                                                             \
        }                                                                                                                            \
    } while (0)
    if (a->up) QA_BV_B(true);
    else QA_BV_B(false);
#undef QA_BV_B
#undef QA_BV_N
#undef QA_BV
    if (two_launch) {
        const int rc = quipamd_dequant_gemm(a->xt, QUIPAMD_F16, (const int32_t *)a->qweight, a->bits, QUIPAMD_LAYOUT_STREAM, QUIPAMD_QFN_B, a->scale, nullptr,
                                            nullptr, a->y, QUIPAMD_F32, 0, a->rows, a->m, (int64_t)p * 16, stream);
        if (rc != QUIPAMD_OK) return rc;
    } else if (a->partials) {                                                     // the second launch of the fixed-order meet
        const int64_t n4 = a->rows * a->m / 4;
        bigp_reduce_kernel<<<(unsigned)((n4 + 255) / 256), 256, 0, s>>>((const f32x4_t *)a->partials, (f32x4_t *)a->y, n4, p / 16);
    }
    QA_LAUNCH_CHECK("quipamd_decode_bigp_v_gemm");
    return QUIPAMD_OK;
}
