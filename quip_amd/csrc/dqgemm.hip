// dqgemm.hip -- K2: fused dequant-GEMM  y[b,r] = bias[r] + sum_k What[r,k] x[b,k]
//
// Takes over quant_cuda.vecquant{3,4}matmul (quant.py:229, zeroShot/models/quant.py:207), for 2- and
// 4-bit codes, qfn a (per-row scale/zero, quant.py:8) and qfn b (scalar scale, quant.py:13-14), any bs.
//
// Design (gfx950, wave64, v_mfma_f32_16x16x32_bf16):
//   * The packed weights are the MFMA **A** operand (16 weight rows x 32 k), the activations the **B**
//     operand (32 k x 16 batch columns): D[row][batch].  The STREAM layout stores each 16-row x KC-column
//     tile as 64 lanes x 16 B in exactly the A-fragment order, so one coalesced global_load_dwordx4 per lane
//     (1 KiB per wave) feeds NT = KC/32 MFMAs with no LDS round trip for the weight operand
//     (cdna_hip_programming.md: "M <= 16 decode weights: load straight to VGPRs").
//   * The kernel is a pure latency/ingest problem at the headline shape (4 MiB of weights = 16 KiB per CU,
//     0.56 us at 8 TB/s): every byte a CU needs must be in flight at once.  A workgroup is NW waves (16 at
//     the headline shape); wave w owns k-chunks w, w+NW, ... of the workgroup's RT row tiles, so at
//     K = 4096 every wave issues its single weight load and its x stage in the first few cycles.
//   * x goes through LDS in full lines (cdna_hip_programming.md: fragment-shaped x loads cost +18..45 %):
//     each wave DMA-stages the 16 x KC slab of x its chunk needs with global_load_lds_dwordx4 (512 B
//     contiguous per batch row) into a wave-PRIVATE LDS region -- no barrier, only the wave's own vmcnt --
//     and reads B fragments back with ds_read_b128.  The LDS image is XOR-swizzled through the DMA's
//     SOURCE address (16-B column c of batch row b lands at column c ^ b), which makes the fragment reads
//     (lanes = 16 rows x 4 k-groups at one column) bank-conflict free.
//   * In-register dequant, 2 VALU per bf16 pair: the code is shifted onto the TOP mantissa bits of a bf16
//     with a fixed exponent:  2 bit -> 0x4080 | c<<5 = 4 + c,  4 bit -> 0x4180 | c<<3 = 16 + c   (exact),
//     pair = ((w >> s) & MASK) | BASE  (v_lshrrev/v_lshlrev + v_and_or_b32).
//   * The constant offset and the affine grid are folded into the epilogue:
//         sum_k (OFF + q) x = acc   =>   sum_k q x = acc - OFF * xsum,   xsum[b] = sum_k x[b,k]
//         qfn b:  y = (2 s / maxq) * (acc - (OFF + maxq/2) * xsum)
//         qfn a:  y = scale[r]     * (acc - (OFF + zero[r]) * xsum)
//     xsum is accumulated beside the MFMAs with v_dot2_f32_bf16 against (1,1).
//   * The NW k-partials of a row tile meet in LDS (each wave parks its accumulators in its own, now dead,
//     x region), one barrier, then wave rt reduces and stores.  D layout (col = lane&15 = batch,
//     row = 4*(lane>>4)+reg = weight row): a lane owns 4 consecutive output features of one batch row
//     -> one 8-byte (bf16) / 16-byte (fp32) store.
//   * bs > 16: BT batch tiles per wave reuse every dequantised A fragment (weights are streamed once per
//     BT*16 batch rows).
//
// Algorithmic bytes per call: m*d*bits/8 + 2*bs*d + (2|4)*bs*m.  FLOPs: 2*bs*m*d.
#include "common.h"
#include "dq_common.h"
#include "k2_dispatch.h"
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

namespace {

// (the s_memtime phase probe and its ablation switches live in scripts/probe_k2.hip, not in the product kernel)
#define QA_ABL(bit) 0
#define QA_KEEP(v) do { } while (0)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// counted vmcnt wait; n is a compile-time constant after unrolling, so the switch folds to one s_waitcnt
__device__ __forceinline__ void wait_vmcnt(int n)
{
#define QA_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        QA_VMCNT_CASE(0) QA_VMCNT_CASE(2) QA_VMCNT_CASE(4) QA_VMCNT_CASE(6) QA_VMCNT_CASE(8) QA_VMCNT_CASE(10)
        QA_VMCNT_CASE(12) QA_VMCNT_CASE(14) QA_VMCNT_CASE(16) QA_VMCNT_CASE(18) QA_VMCNT_CASE(20) QA_VMCNT_CASE(22)
        QA_VMCNT_CASE(24)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef QA_VMCNT_CASE
}

// Workgroup = NW waves over RT row tiles (16 rows each) x BT batch tiles (16 batch rows each);
// grid = (m/16/RT, ceil(bs/16/BT)).  LDS: NW wave-private regions of REGION bytes.
//
// x slab (16 batch rows x KC columns of one chunk) in LDS, in 16-byte units: the slab is NCB column blocks of
// 128 B; DMA instruction i = 2*cb + rh moves column block cb of rows 8*rh .. 8*rh+7 -- one full 128-B line
// per row -- and lane L of it lands at unit 64*i + L.  Lane L = 8*(row&7) + slot fetches logical 16-B column
// w = slot ^ (row&7) of the block, so logical (row b, column c16) lives at
//     unit(b, c16) = 64*(2*(c16>>3) + (b>>3)) + 8*(b&7) + ((c16&7) ^ (b&7)),
// which spreads every ds_read_b128 lane group (8 rows x 2 k-groups) over all 16 slots of a 256-B bank row.
// MFMA step t needs column block t>>1 only, so compute starts when the first two DMAs have landed.
template <int BITS, int RT, int BT, int NW>
__global__ __launch_bounds__(64 * NW) void dqgemm_kernel(const uint16_t *__restrict__ x,
                                                         const uint4 *__restrict__ qw, EpiArgs e, int64_t d)
{
    typedef Deq<BITS> Q;
    constexpr int KC = Q::KC, NT = Q::NT;
    constexpr int ROWB = KC * 2;                 // bytes of one batch row of an x slab
    constexpr int NCB = ROWB / 128;              // 128-B column blocks per slab (4 | 2)
    constexpr int NI = 2 * NCB;                  // DMA instructions per slab (1 KiB each)
    constexpr int XB = 16 * ROWB;                // one slab
    constexpr int PART = RT * BT * 5 * 64 * 4;   // parked partials per wave
    constexpr int REGION = (BT * XB > PART) ? BT * XB : PART;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, g = lane >> 4;
    const uint32_t nkc = (uint32_t)(d / KC);
    const uint32_t rt0 = blockIdx.x * RT;
    const uint32_t bt0 = blockIdx.y * BT;
    const uint32_t rowbytes = (uint32_t)d * 2u;
    char *myreg = smem + wave * REGION;

    // epilogue coefficients for the first (row tile, batch tile) pair this wave will reduce
    EpiRow epi;
    if (wave < RT * BT) epi = load_epi(e, (int64_t)(rt0 + wave / BT) * 16 + 4 * g);

    // buffer descriptor over the rows of x from this workgroup's first batch row on: rows >= bs are out of
    // range and the DMA writes zeros for them (no clamping, no garbage columns)
    const int64_t brow0 = (int64_t)bt0 * 16;
    const int64_t rem = (e.bs - brow0) * (int64_t)rowbytes;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + brow0 * d), 0, (int)rem, 0x00020000);
    const uint32_t vrow = (lane >> 3), vslot = (lane & 7) ^ (lane >> 3);
    const uint32_t voff_lo = vrow * rowbytes + (vslot << 4);          // rows 0..7 of a slab
    const uint32_t voff_hi = voff_lo + 8u * rowbytes;                 // rows 8..15
    // fragment read addresses: lane (j, g), step t -> unit(j, 4t+g)
    const uint32_t rd_base = (j >> 3) * 1024 + (j & 7) * 128;
    const uint32_t rd0 = rd_base + (((0 + g) ^ (j & 7)) << 4);        // t even
    const uint32_t rd1 = rd_base + (((4 + g) ^ (j & 7)) << 4);        // t odd

    f32x4_t acc[RT][BT];
    float xs[BT];
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        xs[bt] = 0.f;
#pragma unroll
        for (int r = 0; r < RT; ++r) acc[r][bt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    for (uint32_t kc = wave; kc < nkc; kc += NW) {
        // ---- issue everything this chunk needs: RT weight tiles to VGPRs, BT x slabs to LDS -------------------
        uint4 w[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const uint4 *wt = qw + ((uint64_t)(rt0 + r) * nkc + kc) * 64;       // scalar base
            w[r] = wt[lane];
        }
        const uint32_t soff = kc * ROWB;
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt)
#pragma unroll
                for (int rh = 0; rh < 2; ++rh)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t *)(myreg + bt * XB + (2 * cb + rh) * 1024), 16,
                                                             (rh ? voff_hi : voff_lo) + bt * 16u * rowbytes,
                                                             soff + cb * 128, 0, 0);
        // hipcc forces vmcnt(0) before any ds_read while an LDS-DMA is pending, so per-column-block counted
        // waits would be drained anyway: one wait (wave-private data: no barrier needed)
        wait_vmcnt(0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                uint4 xf[2];
                xf[0] = *reinterpret_cast<const uint4 *>(myreg + bt * XB + cb * 2048 + rd0);
                xf[1] = *reinterpret_cast<const uint4 *>(myreg + bt * XB + cb * 2048 + rd1);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * cb + tt;
                    Frag bb;
                    bb.u = xf[tt];
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        Frag a;
                        a.u = Q::frag(w[r], t);
                        acc[r][bt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bb.v, acc[r][bt], 0, 0, 0);
                    }
                    xs[bt] = dot_ones(xf[tt], xs[bt]);
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // all slab reads retired before the next DMA lands
    }
    // the 4 lane groups hold disjoint k's of the same batch row: fold them so every lane has xsum[b=j]
#pragma unroll
    for (int bt = 0; bt < BT; ++bt) {
        xs[bt] += __shfl_xor(xs[bt], 16);
        xs[bt] += __shfl_xor(xs[bt], 32);
    }

    if constexpr (NW > 1) {
        float *park = reinterpret_cast<float *>(myreg);        // [RT][BT][5][64], the wave's own dead x region
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                float *p = park + ((r * BT + bt) * 5) * 64 + lane;
                p[0] = acc[r][bt][0]; p[64] = acc[r][bt][1]; p[128] = acc[r][bt][2]; p[192] = acc[r][bt][3];
                p[256] = xs[bt];
            }
        __syncthreads();
        // wave u reduces the (r, bt) pairs u, u + NW, ...
        for (int pr = wave; pr < RT * BT; pr += NW) {
            f32x4_t a = {0.f, 0.f, 0.f, 0.f};
            float s = 0.f;
#pragma unroll
            for (int v = 0; v < NW; ++v) {
                const float *p = reinterpret_cast<const float *>(smem + v * REGION) + (pr * 5) * 64 + lane;
                a[0] += p[0]; a[1] += p[64]; a[2] += p[128]; a[3] += p[192];
                s += p[256];
            }
            const int r = pr / BT, bt = pr - r * BT;
            const int64_t r0 = (int64_t)(rt0 + r) * 16 + 4 * g;
            if (pr != wave) epi = load_epi(e, r0);
            epilogue_store(e, epi, Q::OFF, a, s, (int64_t)(bt0 + bt) * 16 + j, r0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int bt = 0; bt < BT; ++bt) {
                const int64_t r0 = (int64_t)(rt0 + r) * 16 + 4 * g;
                if (r * BT + bt != 0) epi = load_epi(e, r0);
                epilogue_store(e, epi, Q::OFF, acc[r][bt], xs[bt], (int64_t)(bt0 + bt) * 16 + j, r0);
            }
    }
}

// ---- bs <= 16: waves specialised per (k-chunk, row tile), x slabs shared through LDS, optional split-K ---------
// Workgroup = CW x RT waves: wave (c, r) multiplies weight tile (row tile rt0 + r, chunk c of the current group)
// by x slab c, which the RT waves of chunk c stage together (NI/RT DMA instructions each).  The workgroup walks
// the chunks [z*cps, (z+1)*cps) of its k-slice z = blockIdx.z in groups of CW.  With S = gridDim.z > 1 slices the
// partial results are added into y with fp32 atomics -- only legal under the reference's in-place-accumulate
// contract (y fp32, pre-filled by the caller with the bias: quant.py:226-230), where it cuts the x bytes every
// CU has to ingest by S (at m = 4096 a full-K workgroup ingests all 128 KiB of x for 16 KiB of weights).
// up to 4 independent problems of identical shape in one launch (blockIdx.y): the q / k / v projections of a block
struct TileGroup {
    const uint16_t *x[4];
    const uint4 *qw[4];
    const float *scale[4], *zero[4], *bias[4];
    void *y[4];
};

template <int BITS, int RT, int CW, int DEPTH>
__global__ __launch_bounds__(64 * RT * CW) void dqgemm_tile_kernel(TileGroup G, EpiArgs e0, int64_t d, uint32_t cps)
{
    const int gi = blockIdx.y;
    const uint16_t *__restrict__ x = G.x[gi];
    const uint4 *__restrict__ qw = G.qw[gi];
    EpiArgs e = e0;
    e.scale = G.scale[gi]; e.zero = G.zero[gi]; e.bias = G.bias[gi]; e.y = G.y[gi];
    // NOTE on code shape (measured with scripts/probe_k2.hip): a wave issues roughly one instruction per 5 cycles, so
    // the per-wave instruction count of each phase is what a microsecond-scale launch pays for; rolling the loops to
    // shrink code did NOT help (instruction fetch is not the limit) and serialised LDS latency, so the compute is
    // unrolled, the tail is spread over reducer waves and the epilogue avoids IEEE division.
    typedef Deq<BITS> Q;
    constexpr int KC = Q::KC;
    constexpr int ROWB = KC * 2, NCB = ROWB / 128, NI = 2 * NCB, XB = 16 * ROWB;
    constexpr int DPW = NI / RT;                 // DMA instructions per wave per slab
    static_assert(NI % RT == 0, "slab DMA must split evenly over the row-tile waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [CW] slabs, then the parking area

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = wave / RT, r = wave - c * RT;   // chunk slot, row tile
    const int j = lane & 15, g = lane >> 4;
    const uint32_t nkc = (uint32_t)(d / KC);
    const uint32_t rt = blockIdx.x * RT + r;
    const uint32_t rowbytes = (uint32_t)d * 2u;
    const uint32_t k_lo = blockIdx.z * cps, k_hi = (k_lo + cps < nkc) ? k_lo + cps : nkc;
    // DEPTH slab sets: chunk slot c of set `buf` is slab (buf*CW + c)
    char *slab = smem + c * XB;
    float *park = reinterpret_cast<float *>(smem + CW * DEPTH * XB);   // [CW][RT][4][64] acc, then [CW][64] xsum

    // epilogue parameters of the one output row this wave will finish (reducer waves only), fetched NOW so their
    // latency hides under the weight stream
    float e_sc = 0.f, e_zr = 0.f, e_bi = 0.f;
    if (wave < 4 * RT) {
        const int64_t row0 = (int64_t)(blockIdx.x * RT + (wave >> 2)) * 16 + (lane & 15);
        e_sc = e.qfn == QUIPAMD_QFN_B ? e.scale[0] : e.scale[row0];
        if (e.qfn != QUIPAMD_QFN_B) e_zr = e.zero[row0];
        if (e.bias) e_bi = e.bias[row0];
    }

    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)x, 0, (int)(e.bs * (int64_t)rowbytes), 0x00020000);
    const uint32_t voff_lo = (lane >> 3) * rowbytes + ((uint32_t)((lane & 7) ^ (lane >> 3)) << 4);
    const uint32_t voff_hi = voff_lo + 8u * rowbytes;
    const uint32_t rd_base = (j >> 3) * 1024 + (j & 7) * 128;
    const uint32_t rd0 = rd_base + (((0 + g) ^ (j & 7)) << 4);
    const uint32_t rd1 = rd_base + (((4 + g) ^ (j & 7)) << 4);

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    float xs = 0.f;
    // DEPTH == 1: issue a chunk group, wait, compute, repeat (everything in flight at once when the k-slice is one group).
    // DEPTH == 2: software pipeline over two slab sets -- the NEXT group's weight load and slab DMA are issued after the
    //             current group's fragments have been read into registers and BEFORE its MFMAs, so they fly under the
    //             dequant + MFMA work (hipcc only forces vmcnt(0) in front of the next ds_read, which is where we need it).
    auto issue = [&](uint32_t kgrp, int buf, uint4 &wdst) -> bool {
        const uint32_t kc = kgrp + c;
        const bool lv = kc < k_hi;                                 // wave-uniform
        wdst = make_uint4(0, 0, 0, 0);
        if (lv) {
            if (!QA_ABL(4)) wdst = (qw + ((uint64_t)rt * nkc + kc) * 64)[lane];
            // this wave's share of the slab: DMA instructions i = r*DPW .. +DPW (i = 2*column block + row half)
            if (!QA_ABL(1))
#pragma unroll
            for (int q = 0; q < DPW; ++q) {
                const int i = r * DPW + q;
                if ((i & 1) && e.bs <= 8) continue;                 // rows 8..15 of the slab feed batch columns that are never stored
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t *)(slab + buf * CW * XB + i * 1024), 16,
                                                         (i & 1) ? voff_hi : voff_lo, kc * ROWB + (i >> 1) * 128, 0, 0);
            }
        }
        return lv;
    };
    if constexpr (DEPTH == 1) {
        // straight form (measured 6 % faster than the generic pipelined loop below when HBM latency is exposed:
        // same-process A/B against the round-1c source, 5.26 vs 5.60 us at 4096^2 cold)
#pragma unroll 1
        for (uint32_t k0 = k_lo; k0 < k_hi; k0 += CW) {
            const uint32_t kc = k0 + c;
            const bool live = kc < k_hi;                               // wave-uniform
            uint4 w = make_uint4(0, 0, 0, 0);
            if (live) {
                if (!QA_ABL(4)) w = (qw + ((uint64_t)rt * nkc + kc) * 64)[lane];
                if (!QA_ABL(1))
#pragma unroll
                for (int q = 0; q < DPW; ++q) {
                    const int i = r * DPW + q;
                    if ((i & 1) && e.bs <= 8) continue;             // rows 8..15 feed batch columns that are never stored
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t *)(slab + i * 1024), 16, (i & 1) ? voff_hi : voff_lo,
                                                             kc * ROWB + (i >> 1) * 128, 0, 0);
                }
            }
            wait_vmcnt(0);
            if constexpr (RT > 1) __syncthreads();                     // slab c complete (RT waves contributed)
            if (QA_ABL(2)) { QA_KEEP(w.x); QA_KEEP(w.y); QA_KEEP(w.z); QA_KEEP(w.w); }
            else if (live) {
                uint4 xf[Q::NT];
#pragma unroll
                for (int t = 0; t < Q::NT; ++t)
                    xf[t] = *reinterpret_cast<const uint4 *>(slab + (t >> 1) * 2048 + ((t & 1) ? rd1 : rd0));
#pragma unroll
                for (int t = 0; t < Q::NT; ++t) {
                    Frag a, bb;
                    a.u = Q::frag(w, t);
                    bb.u = xf[t];
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bb.v, acc, 0, 0, 0);
                    if (r == 0) xs = dot_ones(xf[t], xs);            // one wave per chunk keeps the row sums of x
                }
            }
            if (k0 + CW < k_hi) {                                      // every read of the slabs retired before the next DMA
                if constexpr (RT > 1) __syncthreads();
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
    } else {
    int buf = 0;
    uint4 w_nxt;
    bool live_nxt = issue(k_lo, 0, w_nxt);
#pragma unroll 1
    for (uint32_t k0 = k_lo; k0 < k_hi; k0 += CW) {
        wait_vmcnt(0);
        if constexpr (RT > 1) __syncthreads();                     // slab complete (RT waves contributed)
        const uint4 w = w_nxt;
        const bool live = live_nxt;
        const char *sl = slab + buf * CW * XB;
        uint4 xf[Q::NT];
        if (live && !QA_ABL(2)) {
#pragma unroll
            for (int t = 0; t < Q::NT; ++t)
                xf[t] = *reinterpret_cast<const uint4 *>(sl + (t >> 1) * 2048 + ((t & 1) ? rd1 : rd0));
        }
        const bool more = k0 + CW < k_hi;
        if (more) { live_nxt = issue(k0 + CW, buf ^ 1, w_nxt); buf ^= 1; }
        if (QA_ABL(2)) { QA_KEEP(w.x); QA_KEEP(w.y); QA_KEEP(w.z); QA_KEEP(w.w); }
        else if (live) {
#pragma unroll
            for (int t = 0; t < Q::NT; ++t) {
                Frag a, bb;
                a.u = Q::frag(w, t);
                bb.u = xf[t];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bb.v, acc, 0, 0, 0);
                if (r == 0) xs = dot_ones(xf[t], xs);            // one wave per chunk keeps the row sums of x
            }
        }
    }
    }

    // ---- meet in LDS: the CW chunk partials of each row tile ---------------------------------------------------------
    // park[(c*RT + r)*4 + comp][lane] = accumulator component comp; xpark[c][lane] = row sums of x over chunk slot c
    if (QA_ABL(8)) { QA_KEEP(acc[0]); QA_KEEP(acc[1]); QA_KEEP(acc[2]); QA_KEEP(acc[3]); QA_KEEP(xs); return; }
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
    // A single wave is a slow serial instruction stream, so the reduction is spread over 4*RT reducer waves.
    // Reducer u = 4*r2 + q finishes the outputs (batch rows 4q .. 4q+3) x (all 16 weight rows of row tile r2):
    // lane l -> batch row b = 4q + (l>>4), weight row l&15, i.e. 16 consecutive outputs of y per 16 lanes (64-B
    // runs for the stores / atomics).  That output is accumulator component (l&3) of MFMA lane (j = b, g = (l&15)>>2).
#pragma unroll 1
    for (int u = wave; u < 4 * RT; u += RT * CW) {
        const int r2 = u >> 2, q = u & 3;
        const int b = 4 * q + (lane >> 4), wr = lane & 15;
        const int src = ((wr & 3) * 64) + b + 16 * (wr >> 2);          // [comp][mfma lane]
        float a = 0.f, xsum = 0.f;
#pragma unroll
        for (int v = 0; v < CW; ++v) {
            a += park[(v * RT + r2) * 256 + src];
            xsum += xpark[v * 64 + b];
        }
        const int64_t row = (int64_t)(blockIdx.x * RT + r2) * 16 + wr;
        if (b < e.bs) {
            if (u != wave) {                                        // only when there are fewer waves than reducer slots
                e_sc = e.qfn == QUIPAMD_QFN_B ? e.scale[0] : e.scale[row];
                e_zr = e.qfn == QUIPAMD_QFN_B ? 0.f : e.zero[row];
                e_bi = e.bias ? e.bias[row] : 0.f;
            }
            const float alpha = e.qfn == QUIPAMD_QFN_B ? e_sc * e.two_over_maxq : e_sc;
            const float c0 = e.qfn == QUIPAMD_QFN_B ? Q::OFF + 0.5f * (float)e.maxq : Q::OFF + e_zr;
            float val = alpha * (a - c0 * xsum);
            if (blockIdx.z == 0) val += e_bi;
            const int64_t o = (int64_t)b * e.m + row;
            if (QA_ABL(16)) { QA_KEEP(val); continue; }
            if (gridDim.z > 1) unsafeAtomicAdd((float *)e.y + o, val);
            else if (e.y_f32) ((float *)e.y)[o] = e.accumulate ? ((float *)e.y)[o] + val : val;
            else ((uint16_t *)e.y)[o] = f32_to_bf16_bits(val);
        }
    }
}

template <int BITS, int RT, int CW, int DEPTH>
int launch_tile(const TileGroup &G, int ngroups, const EpiArgs &e, int64_t d, int S, hipStream_t s)
{
    typedef Deq<BITS> Q;
    constexpr int XB = 16 * Q::KC * 2;
    constexpr size_t lds = (size_t)CW * DEPTH * XB + (size_t)(CW * RT * 256 + CW * 64) * 4;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = dqgemm_tile_kernel<BITS, RT, CW, DEPTH>;
    static QaPerDevice attr_set_dev;
    const int attr_set_d = attr_set_dev.dev();
    if ((attr_set_d < 0 || !attr_set_dev.done[attr_set_d]) && lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm: cannot raise dynamic LDS to %zu", lds);
        if (attr_set_d >= 0) attr_set_dev.done[attr_set_d] = true;
    }
    const uint32_t nkc = (uint32_t)(d / Q::KC);
    const uint32_t cps = (nkc + S - 1) / S;
    const unsigned gz = (nkc + cps - 1) / cps;
    kern<<<dim3((unsigned)(e.m / 16 / RT), (unsigned)ngroups, gz), 64 * RT * CW, lds, s>>>(G, e, d, cps);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm");
    return QUIPAMD_OK;
}

template <int BITS, int RT, int BT, int NW>
int launch_cfg(const uint16_t *x, const uint4 *qw, const EpiArgs &e, int64_t d, hipStream_t s)
{
    typedef Deq<BITS> Q;
    constexpr int XB = 16 * Q::KC * 2;
    constexpr int PART = RT * BT * 5 * 64 * 4;
    constexpr int REGION = (BT * XB > PART) ? BT * XB : PART;
    constexpr size_t lds = (size_t)NW * REGION;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = dqgemm_kernel<BITS, RT, BT, NW>;
    static QaPerDevice attr_set_dev;
    const int attr_set_d = attr_set_dev.dev();
    if ((attr_set_d < 0 || !attr_set_dev.done[attr_set_d]) && lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return qa_fail(QUIPAMD_ERR_LAUNCH, "dequant_gemm: cannot raise dynamic LDS to %zu", lds);
        if (attr_set_d >= 0) attr_set_dev.done[attr_set_d] = true;
    }
    const int64_t nby = (e.bs + 16 * BT - 1) / (16 * BT);
    QA_REQUIRE(nby <= 65535, QUIPAMD_ERR_SHAPE, "dequant_gemm: bs too large for this kernel (%lld)", (long long)e.bs);
    kern<<<dim3((unsigned)(e.m / 16 / RT), (unsigned)nby), 64 * NW, lds, s>>>(x, qw, e, d);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm");
    return QUIPAMD_OK;
}

// ---- bs >= 32: batched kernel ---------------------------------------------------------------------------------------
// Workgroup = 4 waves as 2 row groups x 2 batch groups; wave (rg, bg) owns RT row tiles x 2 batch tiles of the
// (2*RT*16 rows) x (64 batch rows) workgroup tile and walks ALL k-chunks.  Per chunk the 64 x KC slab of x is staged
// once for the workgroup (each wave DMAs a quarter) into one of two LDS buffers; a wave reads its 2 x NT B fragments into
// registers, THEN issues its share of the next chunk's DMA (other buffer) and its next weight loads, THEN runs the
// RT*2*NT MFMAs -- the next chunk flies under the compute, one barrier per chunk.  Every dequantised A fragment feeds 2
// MFMAs (batch tiles), every B fragment RT MFMAs (row tiles); weights are re-streamed once per 64 batch rows (grid.y).
template <int BITS, int RT>
__global__ __launch_bounds__(256) void dqgemm_mb_kernel(const uint16_t *__restrict__ x, const uint4 *__restrict__ qw, EpiArgs e, int64_t d)
{
    typedef Deq<BITS> Q;
    constexpr int KC = Q::KC, NT = Q::NT;
    constexpr int ROWB = KC * 2, NCB = ROWB / 128, NI = 2 * NCB, XB = 16 * ROWB;   // one 16-batch-row slab
    constexpr int SLABS = 4;                                                       // 64 batch rows per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];                    // 2 buffers x 4 slabs

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rg = wave >> 1, bg = wave & 1;
    const int j = lane & 15, g = lane >> 4;
    const uint32_t nkc = (uint32_t)(d / KC);
    const uint32_t rt0 = (blockIdx.x * 2 + rg) * RT;               // first row tile of this wave
    const int64_t brow0 = (int64_t)blockIdx.y * 64;                // first batch row of the workgroup
    const uint32_t rowbytes = (uint32_t)d * 2u;

    const int64_t rem = (e.bs - brow0) * (int64_t)rowbytes;
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)(x + brow0 * d), 0, (int)rem, 0x00020000);
    const uint32_t voff_lo = (lane >> 3) * rowbytes + ((uint32_t)((lane & 7) ^ (lane >> 3)) << 4) + (uint32_t)wave * 16u * rowbytes;
    const uint32_t voff_hi = voff_lo + 8u * rowbytes;
    const uint32_t rd_base = (j >> 3) * 1024 + (j & 7) * 128;
    const uint32_t rd0 = rd_base + (((0 + g) ^ (j & 7)) << 4);
    const uint32_t rd1 = rd_base + (((4 + g) ^ (j & 7)) << 4);

    f32x4_t acc[RT][2];
    float xs[2] = {0.f, 0.f};
#pragma unroll
    for (int r = 0; r < RT; ++r) { acc[r][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc[r][1] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    // wave `wave` stages slab `wave` (batch rows 16*wave .. +15 of the workgroup) of every chunk
    auto issue_x = [&](uint32_t kc, int buf) {
#pragma unroll
        for (int i = 0; i < NI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void_t *)(smem + (buf * SLABS + wave) * XB + i * 1024), 16,
                                                     (i & 1) ? voff_hi : voff_lo, kc * ROWB + (i >> 1) * 128, 0, 0);
    };
    uint4 w_nxt[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) w_nxt[r] = (qw + ((uint64_t)(rt0 + r) * nkc) * 64)[lane];
    issue_x(0, 0);
    int buf = 0;
#pragma unroll 1
    for (uint32_t kc = 0; kc < nkc; ++kc) {
        wait_vmcnt(0);
        __syncthreads();                                           // chunk kc staged by all 4 waves; buffer buf^1 is free
        uint4 w[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r) w[r] = w_nxt[r];
        uint4 xf[2][NT];
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
            const char *sl = smem + (buf * SLABS + 2 * bg + bt) * XB;
#pragma unroll
            for (int t = 0; t < NT; ++t) xf[bt][t] = *reinterpret_cast<const uint4 *>(sl + (t >> 1) * 2048 + ((t & 1) ? rd1 : rd0));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // fragments are in registers
        if (kc + 1 < nkc) {                                        // next chunk flies under the MFMAs below
#pragma unroll
            for (int r = 0; r < RT; ++r) w_nxt[r] = (qw + ((uint64_t)(rt0 + r) * nkc + kc + 1) * 64)[lane];
            issue_x(kc + 1, buf ^ 1);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                Frag a, b0, b1;
                a.u = Q::frag(w[r], t);
                b0.u = xf[0][t];
                b1.u = xf[1][t];
                acc[r][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b0.v, acc[r][0], 0, 0, 0);
                acc[r][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b1.v, acc[r][1], 0, 0, 0);
            }
            if (rg == 0) { xs[0] = dot_ones(xf[0][t], xs[0]); xs[1] = dot_ones(xf[1][t], xs[1]); }
        }
        buf ^= 1;
    }
    // row sums of x: computed by the rg == 0 waves, handed to their rg == 1 partners through LDS
    __syncthreads();                                               // all slab reads done: LDS is free
    float *xsh = reinterpret_cast<float *>(smem);                  // [bg][bt][64]
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
        xs[bt] += __shfl_xor(xs[bt], 16);
        xs[bt] += __shfl_xor(xs[bt], 32);
        if (rg == 0) xsh[(bg * 2 + bt) * 64 + lane] = xs[bt];
    }
    __syncthreads();
#pragma unroll
    for (int bt = 0; bt < 2; ++bt) {
        const float xsum = xsh[(bg * 2 + bt) * 64 + lane];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t r0 = (int64_t)(rt0 + r) * 16 + 4 * g;
            const EpiRow epi = load_epi(e, r0);
            epilogue_store(e, epi, Q::OFF, acc[r][bt], xsum, brow0 + (2 * bg + bt) * 16 + j, r0);
        }
    }
}

template <int BITS, int RT>
int launch_mb(const uint16_t *x, const uint4 *qw, const EpiArgs &e, int64_t d, hipStream_t s)
{
    typedef Deq<BITS> Q;
    constexpr size_t lds = (size_t)2 * 4 * 16 * Q::KC * 2;                          // 64 KiB (2-bit) / 32 KiB (4-bit)
    auto kern = dqgemm_mb_kernel<BITS, RT>;
    const int64_t nby = (e.bs + 63) / 64;
    QA_REQUIRE(nby <= 65535, QUIPAMD_ERR_SHAPE, "dequant_gemm: bs too large for this kernel (%lld)", (long long)e.bs);
    kern<<<dim3((unsigned)(e.m / 16 / (2 * RT)), (unsigned)nby), 256, lds, s>>>(x, qw, e, d);
    QA_LAUNCH_CHECK("quipamd_dequant_gemm");
    return QUIPAMD_OK;
}

// Tuning override of the round-1 kernels (quipamd_tune_dequant_gemm): 0 = use the shape heuristic.  Per THREAD: the
// library's promise is thread safety per stream, and a benchmark thread forcing a shape must not change another's calls.
thread_local int g_tune_rt = 0, g_tune_bt = 0, g_tune_nw = 0, g_tune_split = 0, g_tune_depth = 0;
// per-call kernel selection (quipamd_dequant_gemm_cfg): family + parameters, all 0 = heuristic
thread_local int g_k2_cfg[4] = {0, 0, 0, 0};

#define QA_K2_CASE(RT_, BT_, NW_) \
    if (rt == RT_ && bt == BT_ && nw == NW_) return launch_cfg_each<BITS, RT_, BT_, NW_>(G, ngroups, e, d, s)

// Shape heuristic.  BT: as many batch tiles per wave as the batch has (weights streamed once per 64 batch
// rows).  RT: row tiles per workgroup -- more rows per workgroup means fewer x bytes into the chip
// (every workgroup ingests the whole x), but the grid must still cover the 256 CUs.
template <int BITS>
int launch(const TileGroup &G, int ngroups, const EpiArgs &e0, int64_t d, hipStream_t s);

// one problem of a group through the non-grouped kernels
template <int BITS, int RT, int BT, int NW>
int launch_cfg_each(const TileGroup &G, int ngroups, const EpiArgs &e0, int64_t d, hipStream_t s)
{
    for (int gi = 0; gi < ngroups; ++gi) {
        EpiArgs e = e0;
        e.scale = G.scale[gi]; e.zero = G.zero[gi]; e.bias = G.bias[gi]; e.y = G.y[gi];
        const int rc = launch_cfg<BITS, RT, BT, NW>(G.x[gi], G.qw[gi], e, d, s);
        if (rc) return rc;
    }
    return QUIPAMD_OK;
}

template <int BITS>
int launch(const TileGroup &G, int ngroups, const EpiArgs &e, int64_t d, hipStream_t s)
{
    const int64_t ntile = e.m / 16;
    const int64_t nb = (e.bs + 15) / 16;
    if (nb == 1 && g_tune_bt == 0) {
        // tile kernel.  rt: row tiles per workgroup (x slab reuse); cw: chunks in flight per workgroup;
        // S: k-slices over workgroups (atomics; accumulate contract only).
        const int64_t nkc = d / Deq<BITS>::KC;
        const bool can_split = e.accumulate && e.y_f32;
        // Choices below are the winners of the shape sweeps under profiles/r01f_* (2-bit, bs 1 and 16).  Large m / long K:
        // the per-wave kernel (every wave streams RT weight tiles per chunk: more bytes in flight per wave, private slabs);
        // small m: the tile kernel (all of a CU's bytes in flight at once, parallel reducers, optional split-K).
        const bool tuned = g_tune_rt != 0 || g_tune_nw != 0 || g_tune_split != 0 || g_tune_depth != 0;
        int rt = 1, cw = 8, S = 1;
        if (!tuned && ntile % 4 == 0 && ntile >= 1024) return launch_cfg_each<BITS, 4, 1, 8>(G, ngroups, e, d, s);
        if (!tuned && ntile % 4 == 0 && ntile >= 640 && nkc >= 16) return launch_cfg_each<BITS, 4, 1, 16>(G, ngroups, e, d, s);
        if (!tuned && ntile % 2 == 0 && ntile >= 384 && nkc >= 24) return launch_cfg_each<BITS, 2, 1, 16>(G, ngroups, e, d, s);
        if (can_split && ntile % 2 == 0 && ntile / 2 < 256) {
            // small m under the accumulate contract: 8-wave workgroups (several resident per CU), k split over
            // workgroups until there are ~512 of them (measured best at 4096x4096: rt 2, cw 4, S 4)
            rt = 2; cw = 4;
            S = (int)((512 + ntile / 2 - 1) / (ntile / 2));
            if (S > nkc / cw) S = (int)(nkc / cw);
            if (S < 1) S = 1;
        }
        if (g_tune_rt > 0 && ntile % g_tune_rt == 0) { rt = g_tune_rt; cw = 16 / rt; }
        if (g_tune_nw > 0) cw = g_tune_nw / rt > 0 ? g_tune_nw / rt : 1;
        if (g_tune_split > 0) S = can_split ? g_tune_split : 1;
        // chunks in flight per wave: enough to cover the workgroup's k-slice in one go when LDS allows (<= 128 KiB of slabs)
        const int64_t slice = (nkc + S - 1) / S;
        (void)slice;
        int depth = 1;                       // measured: 2 or 4 groups in flight buy nothing at 4096^2 (profiles/r01f)
        if (g_tune_depth > 0) depth = g_tune_depth;
#define QA_K2T_CASE(RT_, CW_) \
        if (rt == RT_ && cw == CW_) { \
            if (depth == 1) return launch_tile<BITS, RT_, CW_, 1>(G, ngroups, e, d, S, s); \
            if constexpr ((size_t)CW_ * 2 * 16 * Deq<BITS>::KC * 2 <= 128 * 1024) { if (depth == 2) return launch_tile<BITS, RT_, CW_, 2>(G, ngroups, e, d, S, s); } \
            return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: depth %d does not fit LDS for cw=%d", depth, cw); \
        }
        QA_K2T_CASE(1, 16) QA_K2T_CASE(1, 8) QA_K2T_CASE(1, 4) QA_K2T_CASE(1, 2) QA_K2T_CASE(1, 1) QA_K2T_CASE(2, 1)
        QA_K2T_CASE(2, 8)  QA_K2T_CASE(2, 4) QA_K2T_CASE(2, 2)
        QA_K2T_CASE(4, 4)  QA_K2T_CASE(4, 2) QA_K2T_CASE(4, 1)
        if constexpr (BITS == 2) { QA_K2T_CASE(8, 2) QA_K2T_CASE(8, 1) }
#undef QA_K2T_CASE
        return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: no tile kernel for rt=%d cw=%d", rt, cw);
    }
    // bs > 16: one batch tile per workgroup (grid.y walks the batch tiles, weights re-streamed from L2 / Infinity Cache)
    // with as many row tiles per wave as the grid allows beat the BT > 1 variants at every swept shape.
    const int64_t nby64 = (e.bs + 63) / 64;
    if (g_tune_bt == 0 && g_tune_rt == 0 && g_tune_nw == 0 && ngroups == 1 && !g_tune_depth && ((nby64 >= 2 && ntile >= 512) || (nb >= 3 && ntile >= 1024))) {
        // batched kernel: (2*RT*16 rows) x 64 batch rows per workgroup; RT by how many workgroups the grid then has.
        // Wins for bs > 64 on m >= 8192 (or one 64-row batch tile on m >= 16384); on shorter matrices the per-wave kernel,
        // whose 16 waves split K, is faster (profiles/r01h_k2_batched_sweep.jsonl, r01h_k2_default_config_all_shapes.jsonl).
        const int64_t nby = nby64;
        if (ntile % 8 == 0 && (ntile / 8) * nby >= 512) return launch_mb<BITS, 4>(G.x[0], G.qw[0], e, d, s);
        if (ntile % 4 == 0 && (ntile / 4) * nby >= 256) return launch_mb<BITS, 2>(G.x[0], G.qw[0], e, d, s);
        if (ntile % 2 == 0) return launch_mb<BITS, 1>(G.x[0], G.qw[0], e, d, s);
    }
    if (g_tune_depth == 9 && ngroups == 1) {                      // tuning: force the batched kernel, RT = g_tune_rt
        if (g_tune_rt == 4 && ntile % 8 == 0) return launch_mb<BITS, 4>(G.x[0], G.qw[0], e, d, s);
        if (g_tune_rt == 2 && ntile % 4 == 0) return launch_mb<BITS, 2>(G.x[0], G.qw[0], e, d, s);
        if (g_tune_rt == 1 && ntile % 2 == 0) return launch_mb<BITS, 1>(G.x[0], G.qw[0], e, d, s);
        return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: batched kernel needs rt in {1,2,4} dividing the row tiles");
    }
    int rt, bt = 1, nw = 16;
    if (ntile % 4 == 0 && (ntile / 4) * nb >= 128) rt = 4;
    else if (ntile % 2 == 0 && (ntile / 2) * nb >= 128) rt = 2;
    else rt = 1;
    if (g_tune_rt > 0 && ntile % g_tune_rt == 0) rt = g_tune_rt;
    if (g_tune_bt > 0) bt = g_tune_bt;
    if (g_tune_nw > 0) nw = g_tune_nw;
    QA_K2_CASE(1, 1, 16); QA_K2_CASE(2, 1, 16); QA_K2_CASE(4, 1, 16);
    QA_K2_CASE(1, 1, 8);  QA_K2_CASE(2, 1, 8);  QA_K2_CASE(4, 1, 8);
    QA_K2_CASE(1, 1, 4);  QA_K2_CASE(2, 1, 4);  QA_K2_CASE(4, 1, 4);
    QA_K2_CASE(1, 2, 8);  QA_K2_CASE(2, 2, 8);
    QA_K2_CASE(1, 2, 4);  QA_K2_CASE(2, 2, 4);
    QA_K2_CASE(1, 4, 4);  QA_K2_CASE(2, 4, 4);
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: no kernel for rt=%d bt=%d nw=%d", rt, bt, nw);
}
#undef QA_K2_CASE

}   // namespace

extern "C" int quipamd_tune_dequant_gemm(int rt, int bt, int nw, int split)
{
    g_tune_rt = rt; g_tune_bt = bt; g_tune_nw = nw;
    g_tune_split = split % 100;            // split + 100*depth: depth = chunk groups in flight per workgroup (0 = heuristic)
    g_tune_depth = split / 100;
    return QUIPAMD_OK;
}

static int dequant_gemm_impl(int ngroups, const void *const *x, int x_dtype, const int32_t *const *qweight, int bits, int layout, int qfn,
                             const float *const *scale, const float *const *zero, const float *const *bias, void *const *y, int y_dtype,
                             int accumulate, int64_t bs, int64_t m, int64_t d, void *stream)
{
    QA_REQUIRE(ngroups >= 1 && ngroups <= 4, QUIPAMD_ERR_ARG, "dequant_gemm: 1..4 problems per call");
    QA_REQUIRE(x_dtype == QUIPAMD_BF16 || x_dtype == QUIPAMD_F16, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: x must be bf16 or fp16");
    QA_REQUIRE(y_dtype == x_dtype || y_dtype == QUIPAMD_F32, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: y must be f32 or x's dtype");
    QA_REQUIRE(!accumulate || y_dtype == QUIPAMD_F32, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: accumulate needs f32 y");
    QA_REQUIRE(layout == QUIPAMD_LAYOUT_STREAM, QUIPAMD_ERR_UNSUPPORTED,
               "dequant_gemm: qweight must be in STREAM layout (repack with quipamd_unpack/quipamd_pack)");
    QA_REQUIRE(bits == 2 || bits == 3 || bits == 4, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: bits must be 2, 3 or 4");
    const int grid_maxq = (1 << bits) - 1;             // 3-bit codes (maxq 7) ride in the 4-bit container (pack.hip)
    if (bits == 3) bits = 4;
    QA_REQUIRE(qfn == QUIPAMD_QFN_B || qfn == QUIPAMD_QFN_A, QUIPAMD_ERR_ARG, "dequant_gemm: qfn must be a or b");
    QA_REQUIRE(bs * d * 2 < ((int64_t)1 << 31) && m * d * bits / 8 < ((int64_t)1 << 40), QUIPAMD_ERR_SHAPE,
               "dequant_gemm: x larger than 2 GiB is not supported by the 32-bit buffer offsets (bs=%lld d=%lld)", (long long)bs, (long long)d);
    QA_REQUIRE(m % 16 == 0 && d % (512 / bits) == 0, QUIPAMD_ERR_SHAPE,
               "dequant_gemm: needs m %% 16 == 0 and d %% %d == 0 (m=%lld d=%lld)", 512 / bits, (long long)m, (long long)d);
    TileGroup G;
    for (int gi = 0; gi < 4; ++gi) {
        const int k = gi < ngroups ? gi : 0;
        QA_REQUIRE(x[k] && qweight[k] && scale[k] && y[k], QUIPAMD_ERR_ARG, "dequant_gemm: null pointer (problem %d)", k);
        QA_REQUIRE(qfn == QUIPAMD_QFN_B || (zero && zero[k]), QUIPAMD_ERR_ARG, "dequant_gemm: qfn a needs zero");
        G.x[gi] = (const uint16_t *)x[k]; G.qw[gi] = (const uint4 *)qweight[k];
        G.scale[gi] = scale[k]; G.zero[gi] = zero ? zero[k] : nullptr; G.bias[gi] = bias ? bias[k] : nullptr; G.y[gi] = y[k];
    }
    if (bs == 0 || m == 0) return QUIPAMD_OK;
    const bool tuned_old = g_tune_rt || g_tune_bt || g_tune_nw || g_tune_split || g_tune_depth;
    if (ngroups == 1 && !(tuned_old && g_k2_cfg[0] == 0)) {
        // second-generation kernels first (dqgemm_v2.hip); they decline the shapes the kernels below serve better
        K2Call c;
        c.x = x[0]; c.x_dtype = x_dtype; c.qweight = qweight[0]; c.bits = bits; c.qfn = qfn; c.maxq = grid_maxq;
        c.scale = G.scale[0]; c.zero = G.zero[0]; c.bias = G.bias[0]; c.y = y[0]; c.y_dtype = y_dtype; c.accumulate = accumulate;
        c.bs = bs; c.m = m; c.d = d;
        for (int i = 0; i < 4; ++i) c.cfg[i] = g_k2_cfg[i];
        const int rc = k2v2_launch(c, stream);
        if (rc != K2V2_NOT_TAKEN) return rc;
    }
    if (ngroups > 1 && x_dtype == QUIPAMD_F16 && !(tuned_old && g_k2_cfg[0] == 0)) {
        // fp16 activations exist only in the second-generation kernels: 2..3 problems of one shape as ONE launch of the grouped h kernel
        // (dq_hg_kernel: the same body text as dq_h_kernel, the problem picked by blockIdx.y) where it holds the shape, else one launch
        // per problem.  (Routing dq_h_kernel's OWN arguments through a reference was tried first: hipcc's register allocation moved and
        //  tests/test_k2_isa.py caught a copy of an asm load's destination ahead of its wait in the ungrouped instantiation.)
        K2Call cs[4];
        for (int gi = 0; gi < ngroups; ++gi) {
            K2Call &c = cs[gi];
            c.x = x[gi]; c.x_dtype = x_dtype; c.qweight = qweight[gi]; c.bits = bits; c.qfn = qfn; c.maxq = grid_maxq;
            c.scale = G.scale[gi]; c.zero = G.zero[gi]; c.bias = G.bias[gi]; c.y = y[gi]; c.y_dtype = y_dtype; c.accumulate = accumulate;
            c.bs = bs; c.m = m; c.d = d;
            for (int i = 0; i < 4; ++i) c.cfg[i] = 0;
        }
        if (ngroups <= 3 && qfn == QUIPAMD_QFN_B) {
            const int rc = k2v2_launch_grouped(cs, ngroups, stream);
            if (rc != K2V2_NOT_TAKEN) return rc;
        }
        for (int gi = 0; gi < ngroups; ++gi) {
            K2Call c;
            c.x = x[gi]; c.x_dtype = x_dtype; c.qweight = qweight[gi]; c.bits = bits; c.qfn = qfn; c.maxq = grid_maxq;
            c.scale = G.scale[gi]; c.zero = G.zero[gi]; c.bias = G.bias[gi]; c.y = y[gi]; c.y_dtype = y_dtype; c.accumulate = accumulate;
            c.bs = bs; c.m = m; c.d = d;
            for (int i = 0; i < 4; ++i) c.cfg[i] = 0;
            const int rc = k2v2_launch(c, stream);
            QA_REQUIRE(rc != K2V2_NOT_TAKEN, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm_grouped: no fp16 kernel for this shape (m=%lld d=%lld bs=%lld)",
                       (long long)m, (long long)d, (long long)bs);
            if (rc != QUIPAMD_OK) return rc;
        }
        return QUIPAMD_OK;
    }
    QA_REQUIRE(x_dtype == QUIPAMD_BF16, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: the round-1 kernels are bf16 only");
    EpiArgs e;
    e.scale = G.scale[0]; e.zero = G.zero[0]; e.bias = G.bias[0]; e.y = G.y[0];
    e.qfn = qfn; e.maxq = grid_maxq; e.two_over_maxq = 2.0f / (float)e.maxq; e.y_f32 = (y_dtype == QUIPAMD_F32); e.y_f16 = 0; e.accumulate = accumulate;
    e.bs = bs; e.m = m;
    hipStream_t s = (hipStream_t)stream;
    if (bits == 2) return launch<2>(G, ngroups, e, d, s);
    return launch<4>(G, ngroups, e, d, s);
}

extern "C" int quipamd_dequant_gemm(const void *x, int x_dtype, const int32_t *qweight, int bits, int layout, int qfn,
                                    const float *scale, const float *zero, const float *bias, void *y, int y_dtype,
                                    int accumulate, int64_t bs, int64_t m, int64_t d, void *stream)
{
    if (bs == 0 || m == 0) return QUIPAMD_OK;                      // empty batch: nothing to do (its pointers may be null)
    QA_REQUIRE(x && qweight && scale && y, QUIPAMD_ERR_ARG, "dequant_gemm: null pointer");
    return dequant_gemm_impl(1, &x, x_dtype, &qweight, bits, layout, qfn, &scale, &zero, &bias, &y, y_dtype, accumulate, bs, m, d, stream);
}

extern "C" int quipamd_dequant_gemm_cfg(const void *x, int x_dtype, const int32_t *qweight, int bits, int layout, int qfn,
                                        const float *scale, const float *zero, const float *bias, void *y, int y_dtype,
                                        int accumulate, int64_t bs, int64_t m, int64_t d, const int32_t *cfg, void *stream)
{
    for (int i = 0; i < 4; ++i) g_k2_cfg[i] = cfg ? cfg[i] : 0;
    const int rc = quipamd_dequant_gemm(x, x_dtype, qweight, bits, layout, qfn, scale, zero, bias, y, y_dtype, accumulate, bs, m, d, stream);
    for (int i = 0; i < 4; ++i) g_k2_cfg[i] = 0;
    return rc;
}

extern "C" int quipamd_dequant_gemm_grouped(int ngroups, const void *const *x, int x_dtype, const int32_t *const *qweight, int bits,
                                            int layout, int qfn, const float *const *scale, const float *const *zero,
                                            const float *const *bias, void *const *y, int y_dtype, int accumulate, int64_t bs,
                                            int64_t m, int64_t d, void *stream)
{
    if (bs == 0 || m == 0) return QUIPAMD_OK;
    QA_REQUIRE(x && qweight && scale && y, QUIPAMD_ERR_ARG, "dequant_gemm_grouped: null pointer array");
    return dequant_gemm_impl(ngroups, x, x_dtype, qweight, bits, layout, qfn, scale, zero, bias, y, y_dtype, accumulate, bs, m, d, stream);
}
