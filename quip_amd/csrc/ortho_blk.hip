// ortho_blk.hip -- the BLOCKED butterfly operator (method.py:34-35 gen_rand_ortho_butterfly: B0 [q, p, p], B1 [p, q, q] -- what the
// reference's `--incoh_processing` really selects, opt.py:596 sets an unused `proj_extra`) applied to a HANDFUL of rows: the decode step
// of a model quantised by the shipped flag.
//
// A blocked operator stores n (p + q) factor values per side (n = 2048: 393 KB in fp16; n = 11008 = 688 x 16: 15.5 MB), not the
// p^2 + q^2 of the Kronecker form, so the fused decode launches (decode_fused.hip: every workgroup redoes the whole pass in its
// prologue) cannot take it, and round 3 sent such a model through the general two-stage K3 launches (ortho.hip: fp32 factors, rows padded
// to 16, a 4-byte global gather per element): 171 tok/s for OPT-1.3B, 29 tok/s for Llama-2-7B (profiles/r04c_decode_engine.jsonl).
// With one row the operator is a chain of two batched mat-vecs that is bound by the factor bytes, so it is laid out as a stream:
//
//   stage "mix a"  (q groups b, p x p each)   out[a, b] = sum_a' F[b][a][a'] in[a', b]
//   stage "mix b"  (p groups a, q x q each)   out[a, b] = sum_b' F[a][b][b'] in[a, b']
//
// one launch per stage (the stages meet all-to-all), workgroup = (group, 16 output rows), its four waves split K, factors read ONCE as
// fp16 rows (16 bytes per lane = 8 consecutive k of one output row: the v_mfma_f32_16x16x32_f16 A fragment straight from row-major
// memory), the input vector of the group staged in LDS as fp16 hi + lo (two MFMAs per step: the activations keep 22 bits, the
// factors carry the 2^-12 of fp16 -- the tolerance class of the fused decode launches, ~3e-4 per stage), fp32 accumulate.
// The first stage gathers through the operator's input permutation and applies what sits in front of the operator in a decoder block
// (silu(gate) * up, LayerNorm / RMSNorm, the 1 / scaleWH column scale); the second scatters through the output permutation and applies
// what follows it (bias, residual, ReLU) -- the operand set of the Kronecker small-batch kernels (ortho_small.hip).
// Forward = mix a (B0) then mix b (B1); transpose = mix b (B1^T) then mix a (B0^T): the host hands over the factor arrays of the
// orientation it wants (ops.OrthoOp.blk_factors).
#include "common.h"
#include "probe.h"
#include "wavered.h"

namespace {

constexpr int BK_T = 256, BK_MAXR = 16, BK_MAXROWS = 64;   // rows per workgroup (the MFMA's 16 columns); rows per call (blockIdx.z walks groups of 16)

struct BlkStage {
    const uint16_t *F;            // [G][P][P] fp16, (out index, in index)
    const uint16_t *F1;           // FUSED launches: the factors of the OTHER stage ([G1][P1][P1]), whose slice the prologue computes
    int mix_a;                    // 1: groups are b (G = q), P = p, position (i, g) = i q + g;  0: groups are a (G = p), P = q, position g q + i
    int p, q;
    const int32_t *in_idx;        // gather: element at image position pos is in[in_idx[pos]] (null: in[pos])
    const int32_t *out_idx;       // scatter: image position pos goes to out[out_idx[pos]] (null: out[pos])
    const void *in;               // [rows, ld_in] of IN
    int64_t ld_in;
    void *out;                    // [rows, ld_out] of OUT
    int64_t ld_out;
    // in front of the operator (first stage only; indexed by the NATURAL input index)
    const void *gate_up;          // IN [rows, ld_in] or null: value = silu(in) * gate_up
    int norm;                     // 0 none, 1 LayerNorm, 2 RMSNorm over the n input elements of a row
    const uint16_t *gamma, *beta; // fp16 [n]
    float eps;
    const float *colscale;        // fp32 [n] or null
    // behind it (last stage only; indexed by the NATURAL output index)
    const float *bias;            // fp32 [n] or null
    const void *residual;         // [rows, ld_res] of RES dtype or null
    int res_dtype;
    int64_t ld_res;
    int relu;
    int rows;
};

__device__ __forceinline__ float bk_wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

typedef _Float16 bk_f16x8 __attribute__((ext_vector_type(8)));

// eight consecutive elements (16-byte aligned) as floats
template <class T> __device__ __forceinline__ void bk_load8(const void *p, int64_t i, float (&v)[8]);
template <> __device__ __forceinline__ void bk_load8<F32>(const void *p, int64_t i, float (&v)[8])
{
    const float4 a = *reinterpret_cast<const float4 *>((const float *)p + i), b = *reinterpret_cast<const float4 *>((const float *)p + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <> __device__ __forceinline__ void bk_load8<F16>(const void *p, int64_t i, float (&v)[8])
{
    const uint4 a = *reinterpret_cast<const uint4 *>((const uint16_t *)p + i);
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = f16_bits_to_f32((uint16_t)(w[k >> 1] >> (16 * (k & 1))));
}
template <> __device__ __forceinline__ void bk_load8<BF16>(const void *p, int64_t i, float (&v)[8])
{
    const uint4 a = *reinterpret_cast<const uint4 *>((const uint16_t *)p + i);
    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = bf16_bits_to_f32((uint16_t)(w[k >> 1] >> (16 * (k & 1))));
}

// the same eight elements as the load delivers them: requested at the top of a launch, unpacked where they are consumed (an unpack at the
// point of the load would put the wait there)
template <class T> __device__ __forceinline__ float bk_bits_to_f32(uint16_t h);
template <> __device__ __forceinline__ float bk_bits_to_f32<F16>(uint16_t h) { return f16_bits_to_f32(h); }
template <> __device__ __forceinline__ float bk_bits_to_f32<BF16>(uint16_t h) { return bf16_bits_to_f32(h); }
template <class T> __device__ __forceinline__ float bk_cvt(typename DT<T>::storage s);
template <> __device__ __forceinline__ float bk_cvt<F32>(float s) { return s; }
template <> __device__ __forceinline__ float bk_cvt<F16>(uint16_t s) { return f16_bits_to_f32(s); }
template <> __device__ __forceinline__ float bk_cvt<BF16>(uint16_t s) { return bf16_bits_to_f32(s); }
template <class T> struct Raw8 {
    uint4 a;
    __device__ __forceinline__ void load(const void *p, int64_t i) { a = *reinterpret_cast<const uint4 *>((const uint16_t *)p + i); }
    __device__ __forceinline__ void unpack(float (&v)[8]) const
    {
        const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = bk_bits_to_f32<T>((uint16_t)(w[k >> 1] >> (16 * (k & 1))));
    }
};
template <> struct Raw8<F32> {
    float4 a, b;
    __device__ __forceinline__ void load(const void *p, int64_t i)
    {
        a = *reinterpret_cast<const float4 *>((const float *)p + i);
        b = *reinterpret_cast<const float4 *>((const float *)p + i + 4);
    }
    __device__ __forceinline__ void unpack(float (&v)[8]) const { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; }
};

// a / d for 0 <= a < 2^22 with rcp = 1.0f / d: exact (the product is off by less than a 2^-23 < 0.5 / d ... relative to the next integer, and
// a + 0.5 sits 0.5 / d away from both).  An integer division by a run-time divisor is ~50 instructions per wave; the fused prologue had
// a dozen of them between its loads.
__device__ __forceinline__ int bk_div(int a, float rcp) { return (int)(((float)a + 0.5f) * rcp); }

constexpr int BK_MAXOPS = 3;                                          // operators of one launch (q / k / v, gate / up): blockIdx.y
struct BlkStages {
    BlkStage s[BK_MAXOPS];
};

// FUSED: ONE launch per operator.  The workgroups are those of the SECOND stage; each computes, in its prologue, the slice of the first
// stage its block reads -- for every input element k of the block one dot product of length P1 between a contiguous factor row and
// entries of the (pre-processed) input row, which the workgroup holds whole in LDS as fp32:
//     second stage mixes a (block b = g):   in2[a'] = sum_j F1[a'][g][j] x[a' q + j]            (first stage = mix b, F1 [p][q][q])
//     second stage mixes b (block a = g):   in2[b'] = sum_j F1[b'][g][j] x[j q + b']            (first stage = mix a, F1 [q][p][p])
// n MACs and n factor values per workgroup and row: the first stage's factors are read (q / 16) or (p / 16) times in total instead of once,
// and the all-to-all between the stages -- a launch boundary plus a round trip through memory -- is gone.
// MAXI (FUSED): first-stage factor chunks a thread keeps in registers = ceil(n / 8 / 256): 1 up to n = 2048 (the default bound of the
// one-launch form), 8 up to 16384.  One instantiation for all n unrolled eight copies of the partial-product code (4000 lines of ISA),
// seven of which a launch at n = 2048 jumped over, one cold instruction-cache line each.
// NJ (FUSED): 16-byte chunks of the input rows a thread keeps in REGISTERS from the load to the fp32 image in LDS (R n / 8 <= 256 NJ, n a multiple
// of 512): silu(gate) * up, the norm's statistics (wave sums on the DPP network, one barrier), gains and column scale are applied there.  NJ = 0
// is the general form: the rows go to LDS as loaded and every step is a pass over LDS -- the r06i stamps put 1900 clocks on the column scale
// alone and 5300-5600 on a LayerNorm (eight serial LDS reads per statistic, six ds_bpermute steps, three barriers, the mean re-read per element).
template <class IN, class OUT, bool FUSED = false, int MAXI = 8, int NJ = 0>
__global__ __launch_bounds__(BK_T) void blk_stage_kernel(BlkStages SS)
{
    typedef typename DT<IN>::storage in_t;
    const BlkStage &S = SS.s[blockIdx.y];
    // every kernarg field the kernel reads, in ONE scalar round trip (hipcc fetches kernarg fields lazily, one s_load + s_waitcnt per first
    // use: this 5 us launch would open with a dozen serial round trips -- DESIGN.md lessons, round 3)
    asm volatile("" ::"s"(S.F), "s"(S.F1), "s"(S.mix_a), "s"(S.p), "s"(S.q), "s"(S.in_idx), "s"(S.out_idx), "s"(S.in), "s"(S.ld_in), "s"(S.out), "s"(S.ld_out),
                 "s"(S.gate_up), "s"(S.norm), "s"(S.gamma), "s"(S.beta), "s"(S.eps), "s"(S.colscale), "s"(S.bias), "s"(S.residual), "s"(S.res_dtype),
                 "s"(S.ld_res), "s"(S.relu), "s"(S.rows));
    extern __shared__ __attribute__((aligned(16))) char bk_smem[];
    const int P = S.mix_a ? S.p : S.q, q = S.q, n = S.p * S.q;
    const int PS = P + 8;                                              // LDS row stride (halves): 16-byte rows, banks staggered
    uint16_t *XH = reinterpret_cast<uint16_t *>(bk_smem);             // [BK_MAXR][PS] hi
    uint16_t *XL = XH + BK_MAXR * PS;                                 // [BK_MAXR][PS] lo
    float *part = reinterpret_cast<float *>(XL + BK_MAXR * PS);       // [4 waves][4][64]
    float *red = part + 4 * 256;                                      // [8]
    float *stat = red + 8;                                            // [BK_MAXR][2] mean, rstd
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles = P / 16;
    const int g = blockIdx.x / tiles, tile = blockIdx.x - g * tiles;
    const int gr0 = blockIdx.z * BK_MAXR;                              // this workgroup's rows: gr0 .. gr0 + R - 1
    const int R = S.rows - gr0 < BK_MAXR ? S.rows - gr0 : BK_MAXR;
    QA_STAMP(0);
    QA_LOG(0)

    const bool has_gu = S.gate_up != nullptr, has_cs = S.colscale != nullptr;
    const void *gup = has_gu ? S.gate_up : S.in;
    const uint16_t *gmp = S.norm ? S.gamma : S.F, *btp = S.norm == 1 ? S.beta : S.F;
    const float *csp = has_cs ? S.colscale : reinterpret_cast<const float *>(S.F);
    // ---- FUSED: the input rows and the permutation are what the first barrier waits for: requested FIRST (loads return in order; they used
    // to sit behind 18 cold factor / gain loads, the permutation behind a wait for all of those: two to three HBM round trips in series in
    // front of the first barrier -- "rows + permutation staged" 5400 clocks into the launch, profiles/r06_decode_stamps.txt) ------------
    const int n8 = n >> 3, n4 = n >> 2;
    const float rn8 = 1.0f / (float)n8;
    constexpr int NJR = NJ > 0 ? NJ : 1;
    Raw8<IN> rw_v[NJR], rw_u[NJR];
    in_t c0v[NJR], c0u[NJR];                                           // NJ > 0: element 0 of the chunk's row (the shift of the LayerNorm statistics)
    int jr[NJR], jc[NJR];                                              // row and column chunk of the thread's chunk j
    int4 ix_p[2];
    if constexpr (FUSED) {
#pragma unroll
        for (int j = 0; j < NJR; ++j) {
            jr[j] = jc[j] = 0;
            c0v[j] = c0u[j] = in_t();
            if (BK_T * j + 64 * wave < R * n8 || j == 0) {               // (uniform per wave: R n8 is a multiple of 64 wherever NJ > 1 is launched)
                const int e8 = tid + BK_T * j, ec = e8 < R * n8 ? e8 : 0;
                const int r = bk_div(ec, rn8), c8 = ec - r * n8;
                const int64_t at = (int64_t)(gr0 + r) * S.ld_in;
                jr[j] = r;
                jc[j] = c8;
                rw_v[j].load(S.in, at + 8 * c8);
                rw_u[j].load(gup, at + 8 * c8);                          // (no gate: the same line again)
                if constexpr (NJ > 0) {
                    c0v[j] = reinterpret_cast<const in_t *>(S.in)[at];
                    c0u[j] = reinterpret_cast<const in_t *>(gup)[at];
                }
            }
        }
        const int4 *ip = reinterpret_cast<const int4 *>(S.in_idx ? S.in_idx : reinterpret_cast<const int32_t *>(S.F));
        ix_p[0] = ip[S.in_idx && tid < n4 ? tid : 0];
        ix_p[1] = ip[S.in_idx && tid + BK_T < n4 ? tid + BK_T : 0];
    }

    // ---- the factor fragments of the stage's k-steps (the only HBM traffic of an unfused launch: requested first there; FUSED launches
    // request them LAST of their up-front loads -- the MFMAs that consume them are the launch's last phase) -----------------------------
    const int nk = (P + 31) / 32;                                     // k-steps of 32; the last one may be half (P % 32 == 16)
    // SOLO (P <= 128): wave 0 runs all of the stage's <= 4 k-steps itself and finishes the tile from its accumulator.  Split over the four
    // waves, one or two k-steps each bought nothing and cost the partials' trip through LDS, a barrier and the read back (~1000 clocks of
    // the r06i stamps' "MFMAs" + "store" phases); waves 1-3 leave at the last barrier.
    const bool solo = nk <= 4;
    const int i = lane & 15, g4 = lane >> 4;
    const uint16_t *Frow = S.F + ((int64_t)g * P + (tile * 16 + i)) * P + 8 * g4;
    constexpr int MAXS = 6;                                           // k-steps per wave held in registers: P <= 768
    uint4 af[MAXS];
    auto load_af = [&]() {                                              // (SOLO: waves 1-3 request wave 0's fragments too -- see the tail below)
#pragma unroll
        for (int s = 0; s < MAXS; ++s) {
            const int ks = solo ? s : wave + 4 * s;
            const bool ok = ks < nk && ks * 32 + 8 * g4 < P;
            // clamped address, every load unconditional and all in flight.  What a clamped load delivers is never multiplied into a result: a
            // k-step past nk is skipped, a quarter past P meets a zero B fragment (finite factor values x 0).  (A select to zero here made hipcc
            // branch around the loads, with a vmcnt(0) and a register copy inside one of the branches.)
            af[s] = *reinterpret_cast<const uint4 *>(Frow + (ok ? ks * 32 : -8 * g4));
        }
    };
    if constexpr (!FUSED) load_af();

    // ---- everything else this workgroup will read from memory is requested NOW, before the statistics and the MFMAs: the launch is a chain
    // of dependent round trips (index -> value; index -> bias / residual), and requested early they travel under each other --------------
    // (a) the group's input vector: element e = r P + k of the R real rows; the first NPF x 256 of them are prefetched (all of them up to
    //     R P = 768), with the operands of what precedes the operator (null pointers read a dummy address: a branch around a load is a
    //     round trip of its own)
    constexpr int NPF = 3;
    const int RP = R * P;
    float pv[NPF], pu[NPF], pc[NPF];
    uint16_t pg[NPF], pb[NPF];
    // every index first, then every value: written as one loop, hipcc waited for index c (vmcnt(0): and for the values of c - 1) before it
    // requested index c + 1 -- NPF index round trips in series in front of the values
    int psrc[NPF], prow[NPF];
#pragma unroll
    for (int c = 0; c < (FUSED ? 0 : NPF); ++c) {
        const int e = tid + BK_T * c, ec = e < RP ? e : 0;
        const int r = ec / P, k = ec - r * P;
        const int pos = S.mix_a ? k * q + g : g * q + k;
        prow[c] = r;
        psrc[c] = S.in_idx ? S.in_idx[pos] : pos;
    }
    // (b) [requested between the indices and the values of (a): the unfused launch is two dependent round trips -- index, then value / bias /
    //     residual -- and the destination indices used to leave only after the values, a third one]  wave 0 finishes the tile: where its four results per lane go, and the bias / residual that go with them.  EVERY wave requests them,
    // unconditionally and from the same addresses (null operands read a dummy line): vector memory returns in order and hipcc counts the
    // loads a wait may leave outstanding along the path with the FEWEST loads, so a load only wave 0 issues, or one behind a null check, turns
    // every later wait for an OLDER load into a wait for these too -- the first-stage products waited for the bias (vmcnt(0)), a cold
    // dependent round trip, in the middle of the launch.  Unsigned indices: a sign extension is an instruction on the loaded value, and
    // hipcc placed it -- and a vmcnt(0) -- right behind the loads.
    uint32_t tdst[4];
    float tbias[4];
    uint32_t tres16[4], tres32[4];
    const int trow = (lane & 15) < R ? (lane & 15) : 0;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int io = tile * 16 + 4 * g4 + reg;
        const int pos = S.mix_a ? io * q + g : g * q + io;
        const uint32_t *oi = S.out_idx ? reinterpret_cast<const uint32_t *>(S.out_idx) : reinterpret_cast<const uint32_t *>(S.F);
        tdst[reg] = oi[S.out_idx ? pos : 0];                             // (no permutation: BK_FETCH_TAIL puts the position itself)
    }
#pragma unroll
    for (int c = 0; c < (FUSED ? 0 : NPF); ++c) {
        const int src = psrc[c], r = prow[c];
        pv[c] = DT<IN>::load(S.in, (int64_t)(gr0 + r) * S.ld_in + src);
        pu[c] = DT<IN>::load(gup, (int64_t)(gr0 + r) * S.ld_in + src);
        pg[c] = gmp[S.norm ? src : 0];
        pb[c] = btp[S.norm == 1 ? src : 0];
        pc[c] = csp[has_cs ? src : 0];
    }
    // bias / residual hang on those indices: a DEPENDENT round trip.  Requested right here (rounds 4-5) it made wave 0 wait for the -- cold --
    // index vector before it could issue anything else, and in the FUSED form every wave then waited for wave 0 at the first barrier: 2000-4000
    // clocks on the launch's critical path (profiles/r06_decode_stamps.txt, "requests issued").  FUSED launches fetch them behind that barrier,
    // where the indices have long landed; the unfused ones keep the request here (all their waves chase an index of their own anyway).
    // The residual is read both as 16-bit and as 32-bit elements (the form that does not apply reads the dummy line): no branch on its dtype.
    const bool res32 = S.residual && S.res_dtype == QUIPAMD_F32, res16 = S.residual && S.res_dtype != QUIPAMD_F32;
    const float *bias_p = S.bias ? S.bias : reinterpret_cast<const float *>(S.F);
    const uint16_t *r16_p = res16 ? reinterpret_cast<const uint16_t *>(S.residual) : S.F;
    const uint32_t *r32_p = res32 ? reinterpret_cast<const uint32_t *>(S.residual) : reinterpret_cast<const uint32_t *>(S.F);
    const uint16_t *r16_row = r16_p + (res16 ? (int64_t)(gr0 + trow) * S.ld_res : 0);
    const uint32_t *r32_row = r32_p + (res32 ? (int64_t)(gr0 + trow) * S.ld_res : 0);
    const uint32_t tpos0 = (uint32_t)(S.mix_a ? (tile * 16 + 4 * g4) * q + g : g * q + tile * 16 + 4 * g4), tposs = (uint32_t)(S.mix_a ? q : 1);
    const uint32_t mb = S.bias ? ~0u : 0u, m16 = res16 ? ~0u : 0u, m32 = res32 ? ~0u : 0u, mo = S.out_idx ? ~0u : 0u;
#define BK_FETCH_TAIL()                                                                                                                       \
    _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                                                     \
        const uint32_t td_ = (tdst[reg] & mo) | ((tpos0 + reg * tposs) & ~mo);                                                                \
        tdst[reg] = td_;                                                                                                                      \
        tbias[reg] = bias_p[td_ & mb];                                                                                                        \
        tres16[reg] = r16_row[td_ & m16];                                                                                                     \
        tres32[reg] = r32_row[td_ & m32];                                                                                                     \
    }
    if constexpr (!FUSED) BK_FETCH_TAIL()

    QA_STAMP(1);                                                       // every up-front request issued (factors, index -> value, destinations)
    // ---- statistics of the rows (first stage with a norm): every workgroup reduces the whole row -- n <= 16384 values from L2 ---------
    if (!FUSED && S.norm) {
        // ONE pass over the row: sums of (x - c) and (x - c)^2 with c = the row's first element (a shift removes the cancellation of
        // E[x^2] - mean^2), both reduced behind the same pair of barriers
        for (int r = 0; r < R; ++r) {
            const float c = S.norm == 1 ? DT<IN>::load(S.in, (int64_t)(gr0 + r) * S.ld_in) : 0.f;
            float s1 = 0.f, s2 = 0.f;
            for (int e = tid; e < n; e += BK_T) {
                const float dv = DT<IN>::load(S.in, (int64_t)(gr0 + r) * S.ld_in + e) - c;
                s1 += dv;
                s2 += dv * dv;
            }
            s1 = bk_wave_sum(s1);
            s2 = bk_wave_sum(s2);
            if (lane == 0) {
                red[wave] = s1;
                red[4 + wave] = s2;
            }
            __syncthreads();
            if (tid == 0) {
                const float m1 = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n, m2 = ((red[4] + red[5]) + (red[6] + red[7])) / (float)n;
                stat[2 * r] = c + m1;
                stat[2 * r + 1] = rsqrtf((S.norm == 1 ? fmaxf(m2 - m1 * m1, 0.f) : m2) + S.eps);      // RMSNorm: mean of x^2, nothing subtracted
            }
            __syncthreads();
        }
    }

    QA_STAMP(2);                                                       // (unfused, with a norm) the row statistics
    // ---- the group's input vector into LDS as fp16 hi + lo; rows R .. 15 are whatever LDS held (MFMA columns nobody stores) --------------------------
    auto finish = [&](float v, float u, uint16_t gm, uint16_t bt, float cs, int r) {
        if (has_gu) {
            v = DT<IN>::rnd(v / (1.0f + __expf(-v))) * u;                 // silu rounded to the activation dtype like torch's op, then the product
            v = DT<IN>::rnd(v);
        }
        if (S.norm == 1) {                                               // torch LayerNorm: fp32 inside, one rounding to the model's dtype
            v = DT<IN>::rnd((v - stat[2 * r]) * stat[2 * r + 1] * f16_bits_to_f32(gm) + f16_bits_to_f32(bt));
        } else if (S.norm == 2) {                                        // HF LlamaRMSNorm: (x rsqrt(..)).to(dtype), then weight * that
            v = DT<IN>::rnd(DT<IN>::rnd(v * stat[2 * r + 1]) * f16_bits_to_f32(gm));
        }
        if (has_cs) v *= cs;
        return v;
    };
    auto put_rk = [&](int r, int k, float v) {
        const uint16_t hi = f32_to_f16_bits(v);
        XH[r * PS + k] = hi;
        XL[r * PS + k] = f32_to_f16_bits(v - f16_bits_to_f32(hi));
    };
    auto put = [&](int e, float v) {
        const int r = e / P;
        put_rk(r, e - r * P, v);
    };
    if constexpr (FUSED) {
        // Everything the prologue reads from memory is a 16-byte load at an address that depends on nothing loaded before: the factor
        // chunks, the input rows in their NATURAL order (the gather permutation is applied from LDS, where a dependent read costs 64 cycles
        // instead of a round trip through L2), the permutation itself as 16-bit entries.  (The first form of this prologue gathered from
        // memory -- index, then value, n / 256 times in sequence -- and walked its dot products 16 at a time: 14-55 us per launch.)
        float *XIN = stat + 2 * BK_MAXR;                                 // [R][n] fp32: the pre-processed input rows, natural order
        float *PART = XIN + R * n;                                       // [R][n / 8]: partial dot products, one per factor chunk
        uint16_t *IDX = reinterpret_cast<uint16_t *>(PART + R * n8);    // [n]: image position -> source column
        const int P1 = S.mix_a ? q : S.p;                               // length of the first stage's dot products
        const int C = P1 >> 3;                                           // 16-byte chunks per factor row; P C = n / 8 work items
        const float rC = 1.0f / (float)C;
        // the gains, the LayerNorm bias and the column scale of the thread's column chunks: requested with the launch's first loads (they
        // were loaded inside the "gains in place" loop once, behind three barriers: a second cold round trip in the middle of the launch)
        uint4 gm_p[NJR], bt_p[NJR];
        float4 csa_p[NJR], csb_p[NJR];
#pragma unroll
        for (int j = 0; j < NJR; ++j) {
            if (BK_T * j + 64 * wave < R * n8 || j == 0) {
                const int c8p = NJ > 0 ? jc[j] : (tid < n8 ? tid : 0);
                gm_p[j] = *reinterpret_cast<const uint4 *>(gmp + (S.norm ? 8 * c8p : 0));
                bt_p[j] = *reinterpret_cast<const uint4 *>(btp + (S.norm == 1 ? 8 * c8p : 0));
                csa_p[j] = *reinterpret_cast<const float4 *>(csp + (has_cs ? 8 * c8p : 0));
                csb_p[j] = *reinterpret_cast<const float4 *>(csp + (has_cs ? 8 * c8p + 4 : 0));
            }
        }
        uint4 fi[MAXI];                                                  // items per thread held in registers: n <= 2048 MAXI
#pragma unroll
        for (int c = 0; c < MAXI; ++c) {
            const int w = tid + BK_T * c, wc = w < n8 ? w : 0;
            const int k = bk_div(wc, rC), ch = wc - k * C;
            fi[c] = *reinterpret_cast<const uint4 *>(S.F1 + ((int64_t)k * P1 + g) * P1 + 8 * ch);
        }
        load_af();
        // (no permutation: the identity is staged -- the products below read IDX unconditionally; a null check per element was three scalar
        // branches per element in the launch's hottest code)
        auto stage_idx = [&](int e4, int4 ix) {
            if (!S.in_idx) ix = make_int4(4 * e4, 4 * e4 + 1, 4 * e4 + 2, 4 * e4 + 3);
            *reinterpret_cast<uint2 *>(IDX + 4 * e4) = make_uint2((uint32_t)ix.x | ((uint32_t)ix.y << 16), (uint32_t)ix.z | ((uint32_t)ix.w << 16));
        };
        const int4 *ipp = reinterpret_cast<const int4 *>(S.in_idx ? S.in_idx : reinterpret_cast<const int32_t *>(S.F));
        auto gate = [&](float (&v)[8], const Raw8<IN> &ru) {             // silu rounded to the activation dtype like torch's op, then the product
            float u[8];
            ru.unpack(u);
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) v[i8] = DT<IN>::rnd(DT<IN>::rnd(v[i8] / (1.0f + __expf(-v[i8]))) * u[i8]);
        };
        if constexpr (NJ > 0) {
            // ---- the rows stay in registers until they are the operator's input -------------------------------------------------------
            float *SUMS = part;                                          // [R][n8 / 64][2]: (sum, sum of squares) of a wave's 512 columns
            const int nslot = n8 >> 6;
            float v[NJ][8], c0[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (BK_T * j + 64 * wave < R * n8) {
                    rw_v[j].unpack(v[j]);
                    c0[j] = S.norm == 1 ? bk_cvt<IN>(c0v[j]) : 0.f;
                    if (has_gu) {
                        gate(v[j], rw_u[j]);
                        const float g0 = bk_cvt<IN>(c0v[j]);
                        if (S.norm == 1) c0[j] = DT<IN>::rnd(DT<IN>::rnd(g0 / (1.0f + __expf(-g0))) * bk_cvt<IN>(c0u[j]));
                    }
                    if (S.norm) {
                        float s1 = 0.f, s2 = 0.f;
#pragma unroll
                        for (int i8 = 0; i8 < 8; ++i8) {
                            const float dv = v[j][i8] - c0[j];
                            s1 += dv;
                            s2 += dv * dv;
                        }
                        s1 = wave_reduce<false>(s1);
                        s2 = wave_reduce<false>(s2);
                        if (lane == 0) *reinterpret_cast<float2 *>(SUMS + 2 * (jr[j] * nslot + (jc[j] >> 6))) = make_float2(s1, s2);
                    }
                }
            }
            {                                                            // (the permutation's LDS writes travel under the statistics' barrier)
                if (tid < n4) stage_idx(tid, ix_p[0]);
                if (tid + BK_T < n4) stage_idx(tid + BK_T, ix_p[1]);
#pragma unroll 2
                for (int e4 = tid + 2 * BK_T; e4 < n4; e4 += BK_T) stage_idx(e4, ipp[S.in_idx ? e4 : 0]);
            }
            if (S.norm) __syncthreads();
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (BK_T * j + 64 * wave < R * n8) {
                    if (S.norm) {
                        const float2 *sm = reinterpret_cast<const float2 *>(SUMS) + jr[j] * nslot;
                        float m1 = 0.f, m2 = 0.f;
                        for (int s0 = 0; s0 < nslot; s0 += 4) {              // the row's wave sums in a fixed order, four reads in flight
                            float2 t[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) t[u] = sm[s0 + u < nslot ? s0 + u : nslot - 1];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (s0 + u < nslot) {
                                    m1 += t[u].x;
                                    m2 += t[u].y;
                                }
                            }
                        }
                        m1 /= (float)n;
                        m2 /= (float)n;
                        const float mean = c0[j] + m1, rstd = rsqrtf((S.norm == 1 ? fmaxf(m2 - m1 * m1, 0.f) : m2) + S.eps);   // RMSNorm: mean of x^2
                        const uint32_t gmw[4] = {gm_p[j].x, gm_p[j].y, gm_p[j].z, gm_p[j].w}, btw[4] = {bt_p[j].x, bt_p[j].y, bt_p[j].z, bt_p[j].w};
                        if (S.norm == 1) {                               // torch LayerNorm: fp32 inside, one rounding to the model's dtype
#pragma unroll
                            for (int i8 = 0; i8 < 8; ++i8)
                                v[j][i8] = DT<IN>::rnd((v[j][i8] - mean) * rstd * f16_bits_to_f32((uint16_t)(gmw[i8 >> 1] >> (16 * (i8 & 1)))) +
                                                       f16_bits_to_f32((uint16_t)(btw[i8 >> 1] >> (16 * (i8 & 1)))));
                        } else {                                         // HF LlamaRMSNorm: (x rsqrt(..)).to(dtype), then weight * that
#pragma unroll
                            for (int i8 = 0; i8 < 8; ++i8)
                                v[j][i8] = DT<IN>::rnd(DT<IN>::rnd(v[j][i8] * rstd) * f16_bits_to_f32((uint16_t)(gmw[i8 >> 1] >> (16 * (i8 & 1)))));
                        }
                    }
                    if (has_cs) {
                        const float csv[8] = {csa_p[j].x, csa_p[j].y, csa_p[j].z, csa_p[j].w, csb_p[j].x, csb_p[j].y, csb_p[j].z, csb_p[j].w};
#pragma unroll
                        for (int i8 = 0; i8 < 8; ++i8) v[j][i8] *= csv[i8];
                    }
                    float4 *dst = reinterpret_cast<float4 *>(XIN + jr[j] * n + 8 * jc[j]);
                    dst[0] = make_float4(v[j][0], v[j][1], v[j][2], v[j][3]);
                    dst[1] = make_float4(v[j][4], v[j][5], v[j][6], v[j][7]);
                }
            }
            QA_STAMP(3);                                                 // NJ: rows landed; statistics, gains, scale applied; rows + permutation in LDS
            __syncthreads();
            QA_STAMP(4);
            BK_FETCH_TAIL()                                              // (the destination indices landed long ago: no stall)
        } else {
            // ---- the general form: the rows (silu(gate) * up applied) as fp32 and the permutation as 16-bit entries into LDS; each thread's
            // first chunk is the one it requested at the top of the launch -------------------------------------------------------------
            auto stage_row = [&](int e8, const Raw8<IN> &rv, const Raw8<IN> &ru) {
                const int r = bk_div(e8, rn8), c8 = e8 - r * n8;
                float v[8];
                rv.unpack(v);
                if (has_gu) gate(v, ru);
                float4 *dst = reinterpret_cast<float4 *>(XIN + r * n + 8 * c8);
                dst[0] = make_float4(v[0], v[1], v[2], v[3]);
                dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            };
            if (tid < R * n8) stage_row(tid, rw_v[0], rw_u[0]);
            if (tid < n4) stage_idx(tid, ix_p[0]);
            if (tid + BK_T < n4) stage_idx(tid + BK_T, ix_p[1]);
            for (int e8 = tid + BK_T; e8 < R * n8; e8 += BK_T) {         // more rows / wider operators: the rest from memory
                const int r = bk_div(e8, rn8), c8 = e8 - r * n8;
                Raw8<IN> rv, ru;
                rv.load(S.in, (int64_t)(gr0 + r) * S.ld_in + 8 * c8);
                ru.load(gup, (int64_t)(gr0 + r) * S.ld_in + 8 * c8);
                stage_row(e8, rv, ru);
            }
#pragma unroll 2
            for (int e4 = tid + 2 * BK_T; e4 < n4; e4 += BK_T) stage_idx(e4, ipp[S.in_idx ? e4 : 0]);
            QA_STAMP(3);                                                 // general form: rows + permutation landed and staged
            __syncthreads();
            QA_STAMP(4);
            BK_FETCH_TAIL()                                              // (the destination indices landed long ago: no stall)
            if (S.norm) {                                                // the statistics from LDS: one pass, shifted like the two-launch form
                for (int r = 0; r < R; ++r) {
                    const float c0 = S.norm == 1 ? XIN[r * n] : 0.f;
                    float s1 = 0.f, s2 = 0.f;
                    for (int e = tid; e < n; e += BK_T) {
                        const float dv = XIN[r * n + e] - c0;
                        s1 += dv;
                        s2 += dv * dv;
                    }
                    s1 = bk_wave_sum(s1);
                    s2 = bk_wave_sum(s2);
                    if (lane == 0) {
                        part[(r * 4 + wave) * 2] = s1;
                        part[(r * 4 + wave) * 2 + 1] = s2;
                    }
                }
                __syncthreads();
                if (tid < R) {
                    const float *pr = part + tid * 8;
                    const float c0 = S.norm == 1 ? XIN[tid * n] : 0.f;
                    const float m1 = ((pr[0] + pr[2]) + (pr[4] + pr[6])) / (float)n, m2 = ((pr[1] + pr[3]) + (pr[5] + pr[7])) / (float)n;
                    stat[2 * tid] = c0 + m1;
                    stat[2 * tid + 1] = rsqrtf((S.norm == 1 ? fmaxf(m2 - m1 * m1, 0.f) : m2) + S.eps);
                }
                __syncthreads();
            }
            if (S.norm || has_cs) {                                      // gains / column scale in place, per column chunk for all rows
                auto gains = [&](int c8, const uint4 gm4, const uint4 bt4, const float4 csa, const float4 csb) {
                    const uint32_t gmw[4] = {gm4.x, gm4.y, gm4.z, gm4.w}, btw[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
                    const float csv[8] = {csa.x, csa.y, csa.z, csa.w, csb.x, csb.y, csb.z, csb.w};
                    for (int r = 0; r < R; ++r) {
                        float *xr = XIN + r * n + 8 * c8;
                        const float mean = stat[2 * r], rstd = stat[2 * r + 1];
                        float4 xa = reinterpret_cast<const float4 *>(xr)[0], xb = reinterpret_cast<const float4 *>(xr)[1];
                        float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                        for (int i8 = 0; i8 < 8; ++i8) {
                            const uint16_t gm = (uint16_t)(gmw[i8 >> 1] >> (16 * (i8 & 1))), bt = (uint16_t)(btw[i8 >> 1] >> (16 * (i8 & 1)));
                            float v = xv[i8];
                            if (S.norm == 1) v = DT<IN>::rnd((v - mean) * rstd * f16_bits_to_f32(gm) + f16_bits_to_f32(bt));
                            else if (S.norm == 2) v = DT<IN>::rnd(DT<IN>::rnd(v * rstd) * f16_bits_to_f32(gm));
                            if (has_cs) v *= csv[i8];
                            xv[i8] = v;
                        }
                        reinterpret_cast<float4 *>(xr)[0] = make_float4(xv[0], xv[1], xv[2], xv[3]);
                        reinterpret_cast<float4 *>(xr)[1] = make_float4(xv[4], xv[5], xv[6], xv[7]);
                    }
                };
                if (tid < n8) gains(tid, gm_p[0], bt_p[0], csa_p[0], csb_p[0]);     // the chunk requested at the top of the launch
#pragma unroll 2
                for (int c8 = tid + BK_T; c8 < n8; c8 += BK_T)           // n > 2048: the rest from memory
                    gains(c8, *reinterpret_cast<const uint4 *>(gmp + (S.norm ? 8 * c8 : 0)), *reinterpret_cast<const uint4 *>(btp + (S.norm == 1 ? 8 * c8 : 0)),
                          *reinterpret_cast<const float4 *>(csp + (has_cs ? 8 * c8 : 0)), *reinterpret_cast<const float4 *>(csp + (has_cs ? 8 * c8 + 4 : 0)));
                __syncthreads();
            }
        }
        QA_STAMP(5);                                                     // FUSED: the pre-processed rows are in LDS
        // the work items: factor chunk (k, ch) against the 8 entries of the input it meets, for every row.  The C chunk partials of one
        // dot product sit in C adjacent lanes (256 is a multiple of C = 2, 4, 8, 16): summed on the DPP network in a fixed order and written
        // straight into the stage's input vector.  Any other C goes through PART and a barrier.
        const bool dppsum = C == 2 || C == 4 || C == 8 || C == 16;
        const int sk = S.mix_a ? q : 1, sj = S.mix_a ? 1 : q;
#pragma unroll
        for (int c = 0; c < MAXI; ++c) {
            const int w = tid + BK_T * c;
            if (BK_T * c < n8) {                                         // (uniform)
                const bool live = w < n8;
                const int wc = live ? w : 0;
                const int k = bk_div(wc, rC), ch = wc - k * C;
                const uint32_t fw[4] = {fi[c].x, fi[c].y, fi[c].z, fi[c].w};
                int src[8];
                const int pos0 = k * sk + 8 * ch * sj;                     // image position of (k, j) = k sk + j sj
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) src[i8] = (int)IDX[pos0 + i8 * sj];
                for (int r = 0; r < R; ++r) {
                    const float *xin = XIN + r * n;
                    float xg[8];
#pragma unroll
                    for (int i8 = 0; i8 < 8; ++i8) xg[i8] = xin[src[i8]];
                    float a1 = 0.f;
#pragma unroll
                    for (int i8 = 0; i8 < 8; ++i8) a1 = fmaf(f16_bits_to_f32((uint16_t)(fw[i8 >> 1] >> (16 * (i8 & 1)))), xg[i8], a1);
                    if (dppsum) {
                        if (!live) a1 = 0.f;
                        a1 += dpp_f<0xB1>(a1);                           // quad_perm [1,0,3,2]
                        if (C >= 4) a1 += dpp_f<0x4E>(a1);               // quad_perm [2,3,0,1]
                        if (C >= 8) a1 += dpp_f<0x141>(a1);              // row_half_mirror
                        if (C >= 16) a1 += dpp_f<0x140>(a1);             // row_mirror
                        if (live && ch == 0) put_rk(r, k, a1);
                    } else if (live) {
                        PART[r * n8 + w] = a1;
                    }
                }
            }
        }
        QA_STAMP(6);                                                     // FUSED: first-stage products (their factor chunks landed)
        if (!dppsum) {
            __syncthreads();
            const float rP = 1.0f / (float)P;
            for (int dd = tid; dd < RP; dd += BK_T) {                    // chunk partials in a fixed order: deterministic
                const int r = bk_div(dd, rP), k = dd - r * P;
                const float *pp = PART + r * n8 + k * C;
                float a1 = 0.f;
                for (int ch = 0; ch < C; ++ch) a1 += pp[ch];
                put_rk(r, k, a1);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < (FUSED ? 0 : NPF); ++c) {
        const int e = tid + BK_T * c;
        if (e < RP) put(e, finish(pv[c], pu[c], pg[c], pb[c], pc[c], prow[c]));
    }
    for (int e = tid + BK_T * NPF; e < (FUSED ? 0 : RP); e += BK_T) {   // (more than 768 real elements: 4+ rows of a wide operator)
        const int r = e / P, k = e - r * P;
        const int pos = S.mix_a ? k * q + g : g * q + k;
        const int src = S.in_idx ? S.in_idx[pos] : pos;
        put(e, finish(DT<IN>::load(S.in, (int64_t)(gr0 + r) * S.ld_in + src), DT<IN>::load(gup, (int64_t)(gr0 + r) * S.ld_in + src), gmp[S.norm ? src : 0],
                      btp[S.norm == 1 ? src : 0], csp[has_cs ? src : 0], r));
    }
    // (rows R .. 15 of XH / XL stay as they are: MFMA column j depends on B[:, j] only, and columns >= R are never stored)
    QA_STAMP(7);                                                       // the group's input vector is in LDS (unfused: index -> value landed)
    __syncthreads();
    QA_STAMP(8);
    if (solo && wave != 0) return;                                     // (no barrier behind this point in the SOLO form)

    // ---- the k-steps: D[16 out rows][16 columns = batch rows] ---------------------------------------------------------------------------
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    const int col = lane & (BK_MAXR - 1);
#pragma unroll
    for (int s = 0; s < MAXS; ++s) {
        const int ks = solo ? s : wave + 4 * s;
        if (ks < nk) {                                                   // (uniform per wave)
            const int k0 = ks * 32 + 8 * g4;
            uint4 bh = make_uint4(0u, 0u, 0u, 0u), bl = bh;
            if (k0 < P) {
                bh = *reinterpret_cast<const uint4 *>(XH + col * PS + k0);
                bl = *reinterpret_cast<const uint4 *>(XL + col * PS + k0);
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bk_f16x8, af[s]), __builtin_bit_cast(bk_f16x8, bh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(bk_f16x8, af[s]), __builtin_bit_cast(bk_f16x8, bl), acc, 0, 0, 0);
        }
    }
    float res[4] = {acc[0], acc[1], acc[2], acc[3]};
    QA_STAMP(9);                                                       // MFMAs (the stage's own factor fragments landed)
    if (!solo) {
        float *pw = part + wave * 256 + lane;
        pw[0] = acc[0]; pw[64] = acc[1]; pw[128] = acc[2]; pw[192] = acc[3];
        __syncthreads();
        QA_STAMP(10);
        if (wave != 0) return;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
            res[reg] = (part[reg * 64 + lane] + part[256 + reg * 64 + lane]) + (part[512 + reg * 64 + lane] + part[768 + reg * 64 + lane]);
    }
    {
        // D: column = lane & 15 (batch row), row = 4 (lane >> 4) + reg; destination, bias and residual were fetched at the top
        const int r = lane & 15;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            if (r < R) {
                const float tres = res32 ? __uint_as_float(tres32[reg])
                                   : !res16 ? 0.f
                                   : S.res_dtype == QUIPAMD_F16 ? f16_bits_to_f32((uint16_t)tres16[reg]) : bf16_bits_to_f32((uint16_t)tres16[reg]);
                float v = (res[reg] + (S.bias ? tbias[reg] : 0.f)) + tres;
                if (S.relu) v = fmaxf(v, 0.f);
                DT<OUT>::store(S.out, (int64_t)(gr0 + r) * S.ld_out + (int64_t)tdst[reg], v);
            }
        }
    }
    QA_STAMP(11);
    QA_LOG(1)
}

size_t blk_lds(const BlkStage &S, bool fused)
{
    const int P = S.mix_a ? S.p : S.q;
    const size_t rw = S.rows < BK_MAXR ? S.rows : BK_MAXR;           // rows a workgroup holds
    return (size_t)2 * BK_MAXR * (P + 8) * 2 + (4 * 256 + 8 + 2 * BK_MAXR) * 4 + 64 + (fused ? (size_t)S.p * S.q * 4 * rw + (size_t)(S.p * S.q / 8) * 4 * rw + (size_t)S.p * S.q * 2 + 64 : 0);
}

template <class IN, class OUT, bool FUSED = false, int MAXI = 8, int NJ = 0> int launch_stage(const BlkStages &SS, int nops, hipStream_t s)
{
    const BlkStage &S = SS.s[0];
    const int P = S.mix_a ? S.p : S.q, G = S.mix_a ? S.q : S.p;
    if constexpr (FUSED && MAXI == 8 && NJ == 0) {
        // the instantiation for the shape: one factor chunk per thread up to n = 2048; the rows in registers where they fit 1 or 4 chunks
        // per thread and every wave's chunks belong to one row
        const int n8 = S.p * S.q / 8, rw = S.rows < BK_MAXR ? S.rows : BK_MAXR;
        const bool regs = n8 % 64 == 0;
        if (n8 <= BK_T) {
            if (regs && rw * n8 <= BK_T) return launch_stage<IN, OUT, true, 1, 1>(SS, nops, s);
            if (regs && rw * n8 <= 4 * BK_T) return launch_stage<IN, OUT, true, 1, 4>(SS, nops, s);
        } else if (regs && rw * n8 <= 4 * BK_T) {
            return launch_stage<IN, OUT, true, 8, 4>(SS, nops, s);
        }
    }
    const size_t lds = blk_lds(S, FUSED);
    auto kern = blk_stage_kernel<IN, OUT, FUSED, MAXI, NJ>;
    if (lds > 64 * 1024) {
        static QaPerDevice attr;
        static size_t raised[64] = {};
        const int dv = attr.dev();
        if (dv < 0 || raised[dv] < lds) {
            if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return qa_fail(QUIPAMD_ERR_LAUNCH, "ortho_blocked_rows: cannot raise dynamic LDS to %zu", lds);
            if (dv >= 0) raised[dv] = lds;
        }
    }
    kern<<<dim3((unsigned)(G * (P / 16)), (unsigned)nops, (unsigned)((S.rows + BK_MAXR - 1) / BK_MAXR)), BK_T, lds, s>>>(SS);
    return QUIPAMD_OK;
}

int g_blk_fused_rows = 4;   // ... and up to this many rows (see the dispatch below)
int g_blk_fused_n = 2048;   // quipamd_ortho_blocked_config: one launch per operator up to this n = p q (0: always two launches).  Every workgroup of the
                           // single launch reads the whole input row, the permutation and n first-stage factors -- ~8-10 n bytes from L2, n / 16
                           // workgroups per operator.  Measured inside a decode step (profiles/r04k_decode_engine.jsonl): n = 2048 6.0 us against
                           // 2 x 5 (OPT-1.3B blocked 381 -> 427 tok/s); n = 4096 already slower than the pair (Llama-2-7B 198 -> 183 tok/s);
                           // n = 8192 / 11008 10-18 / 55-73 us

}   // namespace

extern "C" int quipamd_ortho_blocked_supported(int p, int q)
{
    return p >= 16 && q >= 16 && p % 16 == 0 && q % 16 == 0 && p <= 768 && q <= 768;
}

extern "C" int quipamd_ortho_blocked_rows_multi(const quipamd_blk_op *ops, int nops, void *workspace, void *stream)
{
    QA_REQUIRE(ops && nops >= 1 && nops <= BK_MAXOPS, QUIPAMD_ERR_ARG, "ortho_blocked_rows: 1..%d operators per launch", BK_MAXOPS);
    const quipamd_blk_op &o0 = ops[0];
    QA_REQUIRE(quipamd_ortho_blocked_supported(o0.p, o0.q), QUIPAMD_ERR_UNSUPPORTED, "ortho_blocked_rows: factors %d x %d (multiples of 16, <= 768)", o0.p, o0.q);
    QA_REQUIRE(o0.rows >= 0 && o0.rows <= BK_MAXROWS, QUIPAMD_ERR_SHAPE, "ortho_blocked_rows: %lld rows > %d", (long long)o0.rows, BK_MAXROWS);
    if (o0.rows == 0) return QUIPAMD_OK;
    QA_REQUIRE(workspace, QUIPAMD_ERR_ARG, "ortho_blocked_rows: null workspace");
    const int64_t n = (int64_t)o0.p * o0.q;
    BlkStages A, B;
    for (int k = 0; k < BK_MAXOPS; ++k) {
        const quipamd_blk_op *op = &ops[k < nops ? k : 0];
        QA_REQUIRE(op->p == o0.p && op->q == o0.q && op->rows == o0.rows && op->x_dtype == o0.x_dtype && op->out_dtype == o0.out_dtype &&
                       op->first_mixes_a == o0.first_mixes_a, QUIPAMD_ERR_ARG, "ortho_blocked_rows: the operators of one launch share shape, rows, dtypes and orientation");
        QA_REQUIRE(op->F_first && op->F_second && op->x && op->out, QUIPAMD_ERR_ARG, "ortho_blocked_rows: null pointer");
        QA_REQUIRE(op->ld_x >= n && op->ld_out >= n, QUIPAMD_ERR_SHAPE, "ortho_blocked_rows: row strides");
        QA_REQUIRE(op->norm >= 0 && op->norm <= 2 && (op->norm == 0 || op->ln_gamma) && (op->norm != 1 || op->ln_beta), QUIPAMD_ERR_ARG,
                   "ortho_blocked_rows: norm %d needs gamma (and beta for LayerNorm)", op->norm);
        QA_REQUIRE(!op->residual || (op->ld_residual >= n && (op->residual_dtype == QUIPAMD_F32 || op->residual_dtype == QUIPAMD_F16 || op->residual_dtype == QUIPAMD_BF16)),
                   QUIPAMD_ERR_ARG, "ortho_blocked_rows: residual stride / dtype");
        float *ws = (float *)workspace + (int64_t)(k < nops ? k : 0) * o0.rows * n;
        BlkStage &a = A.s[k];
        a.F = (const uint16_t *)op->F_first; a.F1 = nullptr; a.mix_a = op->first_mixes_a; a.p = op->p; a.q = op->q;
        a.in_idx = op->in_idx; a.out_idx = nullptr; a.in = op->x; a.ld_in = op->ld_x; a.out = ws; a.ld_out = n;
        a.gate_up = op->gate_up; a.norm = op->norm; a.gamma = (const uint16_t *)op->ln_gamma; a.beta = (const uint16_t *)op->ln_beta; a.eps = op->ln_eps;
        a.colscale = op->colscale; a.bias = nullptr; a.residual = nullptr; a.res_dtype = 0; a.ld_res = 0; a.relu = 0; a.rows = (int)op->rows;
        BlkStage &b = B.s[k];
        b = a;
        b.F = (const uint16_t *)op->F_second; b.mix_a = !op->first_mixes_a; b.in_idx = nullptr; b.out_idx = op->out_idx; b.in = ws; b.ld_in = n;
        b.out = op->out; b.ld_out = op->ld_out; b.gate_up = nullptr; b.norm = 0; b.colscale = nullptr;
        b.bias = op->bias; b.residual = op->residual; b.res_dtype = op->residual_dtype; b.ld_res = op->ld_residual; b.relu = op->relu;
    }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    // one launch per operator when the input rows fit a workgroup's LDS beside the stage's own buffers, for the dtype pairs a decode step uses
    const bool pair_ok = (o0.x_dtype == QUIPAMD_F32) || (o0.x_dtype == QUIPAMD_F16 && o0.out_dtype != QUIPAMD_F32) ||
                         (o0.x_dtype == QUIPAMD_BF16 && o0.out_dtype == QUIPAMD_BF16);
    bool vec_ok = n % 256 == 0 && n < 65536;                          // 16-byte loads of rows, permutation, gains; 16-bit permutation in LDS
    const int esz = o0.x_dtype == QUIPAMD_F32 ? 4 : 2;
    for (int k = 0; k < nops; ++k) {
        const quipamd_blk_op &op = ops[k];
        const uintptr_t bits = (uintptr_t)op.x | (uintptr_t)op.gate_up | (uintptr_t)op.in_idx | (uintptr_t)op.ln_gamma | (uintptr_t)op.ln_beta |
                               (uintptr_t)op.colscale | (uintptr_t)op.F_first | (uintptr_t)((int64_t)op.ld_x * esz);
        vec_ok = vec_ok && (bits & 15) == 0;
    }
    // ... and only for a few rows: the prologue's work grows with the rows (profiles/r04n: blocked OPT-1.3B at 8 sequences 4.44 ms per step in this
    // form, 4.70 ms at 16 sequences in the two-launch form; the forms tie at 4)
    // (n <= 16384: the fused prologue keeps its first-stage factor chunks in registers, 8 per thread x 256 threads x 8 values -- MAXI in the kernel)
    if (n <= g_blk_fused_n && n <= 8 * 8 * BK_T && o0.rows <= g_blk_fused_rows && vec_ok && pair_ok && blk_lds(B.s[0], true) <= 150 * 1024) {
        BlkStages Fz = B;
        for (int k = 0; k < BK_MAXOPS; ++k) {
            BlkStage &f = Fz.s[k];
            const BlkStage &a = A.s[k];
            f.F1 = a.F; f.in = a.in; f.ld_in = a.ld_in; f.in_idx = a.in_idx;
            f.gate_up = a.gate_up; f.norm = a.norm; f.gamma = a.gamma; f.beta = a.beta; f.eps = a.eps; f.colscale = a.colscale;
        }
        if (o0.x_dtype == QUIPAMD_F32)
            rc = o0.out_dtype == QUIPAMD_F16 ? launch_stage<F32, F16, true>(Fz, nops, s)
                 : o0.out_dtype == QUIPAMD_BF16 ? launch_stage<F32, BF16, true>(Fz, nops, s) : launch_stage<F32, F32, true>(Fz, nops, s);
        else if (o0.x_dtype == QUIPAMD_F16)
            rc = o0.out_dtype == QUIPAMD_F16 ? launch_stage<F16, F16, true>(Fz, nops, s) : launch_stage<F16, BF16, true>(Fz, nops, s);
        else
            rc = launch_stage<BF16, BF16, true>(Fz, nops, s);
        if (rc) return rc;
        QA_LAUNCH_CHECK("quipamd_ortho_blocked_rows (fused)");
        return QUIPAMD_OK;
    }
    switch (o0.x_dtype) {
    case QUIPAMD_F32: rc = launch_stage<F32, F32>(A, nops, s); break;
    case QUIPAMD_F16: rc = launch_stage<F16, F32>(A, nops, s); break;
    case QUIPAMD_BF16: rc = launch_stage<BF16, F32>(A, nops, s); break;
    default: return qa_fail(QUIPAMD_ERR_ARG, "ortho_blocked_rows: x dtype %d", o0.x_dtype);
    }
    if (rc) return rc;
    switch (o0.out_dtype) {
    case QUIPAMD_F32: rc = launch_stage<F32, F32>(B, nops, s); break;
    case QUIPAMD_F16: rc = launch_stage<F32, F16>(B, nops, s); break;
    case QUIPAMD_BF16: rc = launch_stage<F32, BF16>(B, nops, s); break;
    default: return qa_fail(QUIPAMD_ERR_ARG, "ortho_blocked_rows: out dtype %d", o0.out_dtype);
    }
    if (rc) return rc;
    QA_LAUNCH_CHECK("quipamd_ortho_blocked_rows");
    return QUIPAMD_OK;
}

extern "C" void quipamd_ortho_blocked_config(int max_fused_n, int max_fused_rows)
{
    g_blk_fused_n = max_fused_n < 0 ? 0 : max_fused_n;
    g_blk_fused_rows = max_fused_rows < 0 ? 0 : max_fused_rows;
}

extern "C" int quipamd_ortho_blocked_rows(const quipamd_blk_op *op, void *workspace, void *stream)
{
    QA_REQUIRE(op, QUIPAMD_ERR_ARG, "ortho_blocked_rows: null op");
    return quipamd_ortho_blocked_rows_multi(op, 1, workspace, stream);
}
