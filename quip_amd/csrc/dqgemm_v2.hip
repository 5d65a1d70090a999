// dqgemm_v2.hip -- instantiations and shape heuristic of the second-generation K2 kernels (dqgemm_v2.h).
// k2v2_launch() is called by quipamd_dequant_gemm (dqgemm.hip) before the round-1 kernels: it returns
// K2V2_NOT_TAKEN when the shape is better served by those (bf16 only), a status otherwise.  fp16 activations
// (quant.py:226-229 widens x, it never narrows it: an fp16 model keeps all its activation bits on the fp16 MFMA pipe)
// exist only here, so every fp16 shape is taken.
#include <cstdlib>
#include "dqgemm_v2.h"
#include "k2_dispatch.h"

namespace {

template <int BITS, class ACT>
int run_family(const K2Call &c, const K2Args &A, hipStream_t s)
{
    const int64_t m = c.m, d = c.d, bs = c.bs;
    const int64_t ntile = m / 16, nkc = d / (512 / BITS);
    const bool f16 = c.x_dtype == QUIPAMD_F16;
    int fam = c.cfg[0], p1 = c.cfg[1], p2 = c.cfg[2];
    const bool h_fits = bs <= 16 && nkc <= (BITS == 2 ? 16 : 32);
    const bool mb_ok = d % 256 == 0;
    if (fam == K2_FAM_AUTO) {
        // measured on MI355X, profiles/r02*_k2lab.log (cold weights):
        //   h: every shape whose K fits one pass of LDS (d <= 4096) and whose grid is not many rounds of 1-per-CU workgroups
        //   s: tall layers (>= 1024 row tiles), where a weight stream pays; below that the round-1 kernels, which split K
        //      over 16 waves, are faster
        //   mb: bs > 16 when there are >= ~200 workgroup tiles of 256 x 128
        if (bs <= 16) {
            // round 6: 16 < chunks <= 32 (d <= 8192) at bs <= 8 -- the one-pass kernel with 8-row slabs (dq_h_body.inc HALF: 4 KiB per chunk):
            // OPT's fc2 (2048 x 8192) in a blocked-operator decode step, which the round-1 tile kernel served in 6.6 us
            if constexpr (BITS == 2) {
                if (bs <= 8 && nkc > 16 && nkc <= 32 && ntile <= 512)
                    return nkc == 32 ? launch_h2<2, ACT, 1, 8, 4, true, true>(A, s) : launch_h2<2, ACT, 1, 8, 4, true, false>(A, s);
                // 9..16 rows: the same kernel as a GROUPED launch of two problems that share the weights -- rows 0..7 and rows 8.. of x and y
                // (blockIdx.y): every weight tile is streamed twice, but 2 x ntile workgroups fit one round up to 128 row tiles, where the
                // weight-stream family (1 tile x 8 k-parts per workgroup) took 7.1 us for OPT's fc2 at 16 rows
                if (bs > 8 && nkc > 16 && nkc <= 32 && ntile <= 128 && !c.accumulate) {
                    K2GArgs G;
                    G.g[0] = A; G.g[1] = A; G.g[2] = A;
                    G.g[0].e.bs = 8;
                    G.g[1].e.bs = bs - 8;
                    G.g[1].x = A.x + 8 * d;
                    G.g[1].e.y = (char *)A.e.y + (size_t)8 * m * (A.e.y_f32 ? 4 : 2);
                    return nkc == 32 ? launch_h2g<2, ACT, 1, 8, 4, true, true>(G, 2, s) : launch_h2g<2, ACT, 1, 8, 4, true, false>(G, 2, s);
                }
            }
            if (h_fits && ntile <= 768) fam = K2_FAM_H;
            else if (ntile >= 1024 && d % 256 == 0) fam = K2_FAM_S;
            else if (f16) fam = h_fits ? K2_FAM_H : (d % 256 == 0) ? K2_FAM_S : K2_FAM_NONE;   // fp16 exists only here: any shape a kernel can hold
            else return K2V2_NOT_TAKEN;
        } else {
            const int64_t tiles = ((m + 255) / 256) * ((bs + 127) / 128);
            if (mb_ok && ((tiles >= 200 && BITS == 2) || f16)) fam = K2_FAM_MB;   // 4 bit: the big tile does not fit LDS, round 1 wins
            else if (f16) fam = K2_FAM_NONE;
            else return K2V2_NOT_TAKEN;
        }
    }
    if (fam == K2_FAM_H) {
        QA_REQUIRE(h_fits, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: kernel family h needs bs <= 16 and d <= %d", BITS == 2 ? 4096 : 4096);
        if constexpr (BITS == 2) {
            // d = 4096 (16 chunks): 4 waves x 4 adjacent chunks since round 5 -- 4.33-4.42 us cold against 4.64-4.69 for 8 x 2 with the same
            // chunk order (profiles/r05h_k2lab_hl.txt; with rounds 2-4's interleaved chunks 8 x 2 was the faster one); d <= 2048: 8 x 1
            if (p1 == 0) { p1 = nkc <= 8 ? 8 : 4; p2 = nkc <= 8 ? 1 : 4; }
            // row tiles per workgroup (cfg[3]; 0 = by the grid): more than one round of 256 one-per-CU workgroups pays the ingest of x once
            // per round -- 11008 x 4096: 10.5 us at 1 tile, 10.0 at 2, 8.2 at 4; 8192 x 2048: 4.70 / 4.39; at 256 tiles (the headline)
            // 1 tile stays the fastest: 4.37 against 5.44 / 7.63 (profiles/r05p_k2lab_rt.txt)
            int rt = c.cfg[3];
            if (rt == 0) rt = (ntile > 512 && ntile % 4 == 0 && nkc > 8) ? 4 : (ntile > 256 && ntile % 2 == 0) ? 2 : 1;
            if (rt == 2 && ntile % 2 == 0) {
                if (p1 == 8 && p2 == 1 && nkc <= 8) return launch_h<2, ACT, 2, 8, 1>(A, s);
                if (p1 == 4 && p2 == 4) return launch_h<2, ACT, 2, 4, 4>(A, s);
            }
            if (rt == 4 && ntile % 4 == 0 && p1 == 4 && p2 == 4) return launch_h<2, ACT, 4, 4, 4>(A, s);
            // round 6: x straight into registers (dq_hr_kernel), exact fit d = 4096 only.  p1 = 44 forces it, p1 = 4 keeps the LDS-DMA form
            if (p1 == 44 && p2 == 4) {
                QA_REQUIRE(nkc == 16, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: the register-x form holds d = 4096 only (d=%lld)", (long long)d);
                return launch_hr<2, ACT, 4, 4>(A, s);
            }
            if (p1 == 8 && p2 == 1 && nkc <= 8) return launch_h<2, ACT, 1, 8, 1>(A, s);
            if (p1 == 8 && p2 == 2) return launch_h<2, ACT, 1, 8, 2>(A, s);
            if (p1 == 16 && p2 == 1 && nkc <= 16) return launch_h<2, ACT, 1, 16, 1>(A, s);
            if (p1 == 4 && p2 == 4) return launch_h<2, ACT, 1, 4, 4>(A, s);
        } else {
            if (p1 == 0) { p1 = 8; p2 = nkc <= 16 ? 2 : 4; }
            if (p1 == 8 && p2 == 2 && nkc <= 16) return launch_h<4, ACT, 1, 8, 2>(A, s);
            if (p1 == 8 && p2 == 4) return launch_h<4, ACT, 1, 8, 4>(A, s);
        }
        return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: no h kernel for nw=%d nch=%d (d=%lld)", p1, p2, (long long)d);
    }
    if (fam == K2_FAM_S) {
        QA_REQUIRE(bs <= 16 && d % 256 == 0, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: kernel family s needs bs <= 16 and d %% 256 == 0");
        if (p1 == 0) {
            // one workgroup per CU and whole rounds: 7 row tiles per workgroup fit 1792 tiles (28672 rows) exactly, 8 fit 2048
            // (32768 rows: 18.0 us with 8 x 256 workgroups, 27 us with 293 workgroups of 7 -- profiles/r02l_k2lab_s.log)
            // and 4 fit 1024 (16384 rows: 13.1 us with 256 workgroups of 4, 15.0 with 147 of 7 -- profiles/r04q_k2_s_cfgs.jsonl).  Cost = rounds of
            // 256 workgroups x tiles per workgroup; ties go to the larger workgroup (32768 rows: 8 x 256 18.8 us, 4 x 512 25.5)
            const int64_t r4 = ((ntile + 3) / 4 + 255) / 256, r7 = ((ntile + 6) / 7 + 255) / 256, r8 = ((ntile + 7) / 8 + 255) / 256;
            // short and wide (reached with fp16 activations only: a 5..16-row decode step's fc2 / down_proj GEMM): fill the CUs first --
            // 2048 x 8192: 12.1 us with 4 tiles per workgroup (32 workgroups), 7.8 with 2, 7.1 with 1; 4096 x 11008: 15.5 / 9.9 / 9.95
            // (profiles/r05d_k2_s_shortwide.jsonl)
            if (ntile <= 128) { p1 = 1; p2 = 8; }
            else if (ntile <= 256) { p1 = 2; p2 = 4; }
            else if (r4 * 4 < r7 * 7 && r4 * 4 < r8 * 8) { p1 = 4; p2 = 2; }
            else if (r8 * 8 < r7 * 7) { p1 = 8; p2 = 1; }
            else { p1 = 7; p2 = 2; }
        }
        // short and wide (few row tiles, K beyond one LDS pass: OPT's fc2, 2048 x 8192, in a 5..16-row decode step): 4 tiles per workgroup
        // leave 32 workgroups on 256 CUs (12.3 us, profiles/r05c); one or two tiles per workgroup with K over 8 / 4 waves fill 128 / 64
        if (p1 == 1 && p2 == 8) return launch_s<BITS, ACT, 1, 8, 1, 2>(A, s);
        if (p1 == 2 && p2 == 4) return launch_s<BITS, ACT, 2, 4, 1, 3>(A, s);
        if (p1 == 7 && p2 == 2) return launch_s<BITS, ACT, 7, 2, 1, 3>(A, s);
        // (round 6, lab: ring depth 4 / 5 and two stages per k-half and ring step for the 7 x 2 form -- 15.2 / 16.0 / 15.6 us against 15.0 at
        //  28672 x 7168 bs 16, profiles/r06G_k2_s_cfgs.jsonl: the per-step barrier is not what the compute waves wait for; not in the library)
        if (p1 == 4 && p2 == 2) return launch_s<BITS, ACT, 4, 2, 1, 4>(A, s);
        if (p1 == 8 && p2 == 1) return launch_s<BITS, ACT, 8, 1, 2, 3>(A, s);
        return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: no s kernel for nw=%d ksp=%d", p1, p2);
    }
    if (fam == K2_FAM_MB) {
        QA_REQUIRE(mb_ok, QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: kernel family mb needs d %% 256 == 0");
        // FOUR loader waves since round 5 (45 / 23; 44 / 22 are the two-loader forms of rounds 2-4).  The loaders were the bound: a loader
        // issues its share of the stage's 80 DMA instructions (each behind an s_mov m0), waits for ALL of them, meets the barrier; with
        // one loader 593 TF, two 998, three 1272, four 1323 at 28672 x 7168 bs 256 (4096^2 x 2048: 1078 -> 1268), six 585 -- 14 waves leave 128 registers per wave, the 4 x 4 tiles need 150
        // (profiles/r05j_k2lab_mb.txt, r05k_k2lab_mb_loaders.txt).  Halving the VALU per MFMA instead (48: 4 x 8 tiles per wave) bought nothing.
        if (p1 == 0) p1 = (((m + 255) / 256) * ((bs + 127) / 128) >= 200) ? 45 : 23;
        if (p1 == 44 || p1 == 45) {                                       // 256 rows x 128 batch rows, 4 x 4 tiles per wave (4 bit: 128 rows)
            if constexpr (BITS == 2) return p1 == 45 ? launch_mb2<BITS, ACT, 4, 2, 4, 4, 4>(A, s) : launch_mb2<BITS, ACT, 4, 2, 4, 4, 2>(A, s);
            else return p1 == 45 ? launch_mb2<BITS, ACT, 4, 2, 2, 4, 4>(A, s) : launch_mb2<BITS, ACT, 4, 2, 2, 4, 2>(A, s);
        }
        if (p1 == 22) return launch_mb2<BITS, ACT, 4, 2, 2, 2, 2>(A, s);   // 128 rows x  64 batch rows
        if (p1 == 23) return launch_mb2<BITS, ACT, 4, 2, 2, 2, 4>(A, s);
        if constexpr (BITS == 2) {
            // round 5 lab configurations (forced only)
            if (p1 == 46) return launch_mb2<BITS, ACT, 4, 2, 4, 4, 3>(A, s);
            // the same tile on v_mfma_f32_32x32x16 (dq_mb_kernel<..., T32>)
            if (p1 == 47) return launch_mb2<BITS, ACT, 4, 2, 4, 4, 4, true>(A, s);
            // ONE compute wave per SIMD with 4 x 8 accumulator tiles: every dequantised A fragment feeds 8 MFMAs instead of 4 (VALU per MFMA
            // 2.35 -> ~1.2), same workgroup tile, same LDS reads per step: 992 vs 998 TF with two loaders, 1256 vs 1323 with four
            if (p1 == 48) return launch_mb2<BITS, ACT, 4, 1, 4, 8, 2>(A, s);
            if (p1 == 49) return launch_mb2<BITS, ACT, 4, 1, 4, 8, 4>(A, s);
        }
        return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: no mb kernel %d", p1);
    }
    return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: fp16 activations need d %% 256 == 0 (d=%lld)", (long long)d);
}

}   // namespace

static void k2_fill(K2Args &A, const K2Call &c)
{
    A.x = (const uint16_t *)c.x; A.qw = (const u32x4 *)c.qweight; A.d = c.d;
    EpiArgs &e = A.e;
    e.scale = c.scale; e.zero = c.zero; e.bias = c.bias; e.y = c.y; e.qfn = c.qfn; e.maxq = c.maxq;
    e.two_over_maxq = 2.0f / (float)c.maxq;
    e.y_f32 = c.y_dtype == QUIPAMD_F32; e.y_f16 = c.y_dtype == QUIPAMD_F16; e.accumulate = c.accumulate; e.bs = c.bs; e.m = c.m;
}

// A/B runs only: QUIP_HG_RT=1|2|4 forces the row tiles per workgroup of the grouped h kernel.  Read ONCE per process (a getenv on
// every grouped GEMM sat on the decode hot path, and a variable set mid-run silently changed the kernel selection).
static int g_grouped_form = 0;          // quipamd_dequant_gemm_grouped_config (tests, A/B runs); 0 = the environment's value, else the heuristic
static int hg_rt_override()
{
    static const int v = [] {
        const char *ev = getenv("QUIP_HG_RT");
        return ev ? atoi(ev) : 0;
    }();
    return g_grouped_form ? g_grouped_form : v;
}

extern "C" void quipamd_dequant_gemm_grouped_config(int form) { g_grouped_form = form; }

// which grouped weight-stream form serves the big grouped GEMMs of a 5..16-row step by default (0: none, the grouped h kernel).
// profiles/r06b_grouped_forms.jsonl (Llama-2-7B, one box, alternating, ms per step at 16 / 8 sequences): grouped h kernel with 4 row tiles
// 3.21 / 2.78; dq_sg <4,2,1,4> 3.14 / 2.78; <7,2,1,3> 3.00 / 2.63; <8,1,2,3> 3.08 / 2.70.
#ifndef K2_SG_DEFAULT
#define K2_SG_DEFAULT 72
#endif

int k2v2_launch_grouped(const K2Call *calls, int ngroups, void *stream)
{
    const K2Call &c = calls[0];
    const int64_t nkc = c.d / (512 / c.bits);
    // the shapes a decode step groups: fp16, <= 16 rows, K within one LDS pass; the h kernel's own limit of 768 row tiles per problem
    if (ngroups < 2 || ngroups > 3 || c.x_dtype != QUIPAMD_F16 || c.bs > 16 || nkc > (c.bits == 2 ? 16 : 32) || c.accumulate || c.m / 16 > 768)
        return K2V2_NOT_TAKEN;
    K2GArgs G;
    for (int i = 0; i < 3; ++i) k2_fill(G.g[i], calls[i < ngroups ? i : 0]);
    hipStream_t s = (hipStream_t)stream;
    if (c.bits == 2 && nkc > 8) {
        // Row tiles per workgroup: every workgroup pulls ALL of x~ (128 KiB at d = 4096), so a grid of many rounds of one-per-CU workgroups
        // pays that ingest once per round -- Llama's gate / up at 16 rows is 2 x 688 tiles = 5.4 rounds of 256 at RT 1.  With RT row tiles
        // per workgroup the grid is RT times smaller and each x~ slab feeds RT weight tiles.  (For ONE round -- the headline, 256 tiles -- RT > 1
        // lost at every shape in round 2; QUIP_HG_RT=1|2|4 forces it for A/B runs.)
        // (8 row tiles -- Llama's gate / up as ONE round of 172 workgroups -- measured slower than 4: Llama-2-7B at 16 sequences 3.27 ms per step against
        //  3.18, at 8 sequences 2.88 against 2.76, profiles/r05w_hg_rt8_llama.txt; the instantiation is not kept.)
        const int64_t tiles = c.m / 16;
        // Round 6: beyond two rounds of one-per-CU workgroups the grouped WEIGHT-STREAM kernel (dq_sg_kernel: x~ staged once per 4 / 7 / 8 row
        // tiles by a loader wave, weights through the LDS ring) instead of the h kernel, whose every workgroup ingests all of x~ by itself.
        // QUIP_HG_RT = 74 / 72 / 81 force its <4,2,1,4> / <7,2,1,3> / <8,1,2,3> forms, 1 / 2 / 4 the h kernel's row tiles (A/B runs).
        int sg = (tiles * ngroups > 512) ? K2_SG_DEFAULT : 0;
        if (const int f = hg_rt_override(); f == 74 || f == 72 || f == 81) sg = f;
        else if (f == 1 || f == 2 || f == 4) sg = 0;
        if (sg == 74) return launch_sg<2, ActF16, 4, 2, 1, 4>(G, ngroups, s);
        if (sg == 72) return launch_sg<2, ActF16, 7, 2, 1, 3>(G, ngroups, s);
        if (sg == 81) return launch_sg<2, ActF16, 8, 1, 2, 3>(G, ngroups, s);
        int rt = 1;
        if (tiles * ngroups > 256 && tiles % 2 == 0) rt = 2;
        if (tiles * ngroups > 512 && tiles % 4 == 0) rt = 4;
        if (const int f = hg_rt_override(); (f == 1 || f == 2 || f == 4) && tiles % f == 0) rt = f;
        if (rt == 4) return launch_hg<2, ActF16, 4, 4, 4>(G, ngroups, s);
        if (rt == 2) return launch_hg<2, ActF16, 2, 4, 4>(G, ngroups, s);
        return launch_hg<2, ActF16, 1, 4, 4>(G, ngroups, s);
    }
    if (c.bits == 2) {                                                               // d <= 2048: 8 waves x 1 chunk
        const int64_t tiles = c.m / 16;
        int rt = (tiles * ngroups > 256 && tiles % 2 == 0) ? 2 : 1;
        if (const int f = hg_rt_override(); (f == 1 || f == 2) && tiles % f == 0) {
            rt = f;
        }
        return rt == 2 ? launch_hg<2, ActF16, 2, 8, 1>(G, ngroups, s) : launch_hg<2, ActF16, 1, 8, 1>(G, ngroups, s);
    }
    return nkc <= 16 ? launch_hg<4, ActF16, 1, 8, 2>(G, ngroups, s) : launch_hg<4, ActF16, 1, 8, 4>(G, ngroups, s);
}

int k2v2_launch(const K2Call &c, void *stream)
{
    if (c.cfg[0] == K2_FAM_OLD) return K2V2_NOT_TAKEN;
    // (family 5, the round-3 prefill kernel -- every 2-bit tile dequantised once per workgroup into LDS, 32x32x16 mainloop -- measured SLOWER
    //  than the mb kernel at every prefill shape, profiles/r03t_k2_prefill.jsonl, and left the library in round 4: scripts/dqgemm_pf_lab.hip)
    if (c.cfg[0] == K2_FAM_PF) return qa_fail(QUIPAMD_ERR_UNSUPPORTED, "dequant_gemm: kernel family 5 (prefill lab) is not part of the library");
    K2Args A;
    A.x = (const uint16_t *)c.x; A.qw = (const u32x4 *)c.qweight; A.d = c.d;
    EpiArgs &e = A.e;
    e.scale = c.scale; e.zero = c.zero; e.bias = c.bias; e.y = c.y; e.qfn = c.qfn; e.maxq = c.maxq;
    e.two_over_maxq = 2.0f / (float)c.maxq;
    e.y_f32 = c.y_dtype == QUIPAMD_F32; e.y_f16 = c.y_dtype == QUIPAMD_F16; e.accumulate = c.accumulate; e.bs = c.bs; e.m = c.m;
    hipStream_t s = (hipStream_t)stream;
    if (c.x_dtype == QUIPAMD_F16) return c.bits == 2 ? run_family<2, ActF16>(c, A, s) : run_family<4, ActF16>(c, A, s);
    return c.bits == 2 ? run_family<2, ActBF16>(c, A, s) : run_family<4, ActBF16>(c, A, s);
}
