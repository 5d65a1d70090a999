/*
 * quip_amd.h -- C ABI of libquip_amd.so, the MI355X (gfx950) native replacement for the
 * low-bit linear hot path of Cornell-RelaxML/QuIP.
 *
 * The reference has exactly one native boundary: the (absent) `quant_cuda` extension,
 * called at quant.py:229 (`vecquant3matmul`) and zeroShot/models/quant.py:207
 * (`vecquant4matmul`).  Everything else on the hot path is torch tensor code in
 * quant.py / method.py / vector_balance.py; each entry point below names the reference
 * lines whose arithmetic it takes over.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch tensor data_ptr) unless it
 *     says "host"; the library never allocates, the caller owns every buffer;
 *   - matrices are row-major; W is [m, d] = [out_features, in_features] as nn.Linear;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - return value 0 = ok; non-zero = error code below, text via quipamd_last_error();
 *   - calls are asynchronous on `stream` and thread-safe per stream.
 */
#ifndef QUIP_AMD_H
#define QUIP_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QUIPAMD_VERSION 100 /* 0.1.0 */

enum quipamd_status {
    QUIPAMD_OK = 0,
    QUIPAMD_ERR_ARG = 1,         /* null pointer / bad enum */
    QUIPAMD_ERR_SHAPE = 2,       /* dimension not supported by the kernel tiling */
    QUIPAMD_ERR_LAUNCH = 3,      /* hip launch error */
    QUIPAMD_ERR_UNSUPPORTED = 4  /* combination not implemented */
};

enum quipamd_dtype { QUIPAMD_F32 = 0, QUIPAMD_F16 = 1, QUIPAMD_BF16 = 2 };

/* packed-weight layouts.
 * CANONICAL: the reference's rule, zeroShot/models/quant.py:190-199, generalised from 4 to
 *            2 bits: int32 [d/per, m], per = 32/bits, code of column i at bits
 *            [bits*(i%per), +bits) of word [i/per, row].
 * STREAM:    the permutation of it that the fused GEMM streams: 16-row x (512/bits)-column
 *            tiles of 64 lanes x 16 B in MFMA A-fragment order (oracle/quip_oracle.py
 *            pack_stream is the specification).  Requires m % 16 == 0, d % (512/bits) == 0. */
/* bits = 3 with the STREAM layout: codes 0..7 stored in the 4-bit container (K2 dequantises nibbles; the 3-bit grid,
 * maxq = 7, is applied in its epilogue).  bits = 3 with the CANONICAL layout: the reference's 32-codes-in-3-words rule. */
enum quipamd_layout { QUIPAMD_LAYOUT_CANONICAL = 0, QUIPAMD_LAYOUT_STREAM = 1 };

/* grid functions: quant.py:6-8 (a), quant.py:10-15 (b), quant.py:17-21 (c) */
enum quipamd_qfn { QUIPAMD_QFN_A = 0, QUIPAMD_QFN_B = 1, QUIPAMD_QFN_C = 2 };

int quipamd_version(void);
const char *quipamd_last_error(void); /* host string, valid until the next failing call on this thread */
/* Measurement hook (no reference counterpart): in the probe build (csrc/probe.h, -DQA_PROBE) the decode launches write s_memtime stamps
 * of their phases -- slot i of wave w of one workgroup at buf[16 w + i] -- into `buf` (device, 256 x uint64; NULL switches them off).
 * The shipped library returns QUIPAMD_ERR_UNSUPPORTED. */
int quipamd_probe_set(void *buf);
/* Operand prefetch (no reference counterpart; csrc/prefetch.h): attach up to 40 device ranges to the NEXT quipamd_decode_fused_gemm /
 * quipamd_decode_attention_fused launch of the calling thread.  That launch carries 8 extra workgroups (one per XCD) that read one dword per
 * 128-byte line of every range and leave -- the step-independent operands of a LATER launch of the decode step (factor fragments, index
 * vectors, gains, column scales, biases: 50-150 KB, cold in L2 once per token) are then L2-resident on every XCD when it starts.  Purely a
 * hint: results never depend on it.  ptrs / bytes: HOST arrays; n = 0 clears a pending list. */
int quipamd_decode_prefetch_next(const void *const *ptrs, const int64_t *bytes, int n);

/* ---- K1: integer pack / unpack (bit-exact) ------------------------------------------------
 * Replaces the Python/numpy packing loops zeroShot/models/quant.py:198-199 and
 * quant.py:199-217 ("TODO: perform packing on GPU", opt.py:302).
 * codes: uint8 [m, d], one code per byte, values < 2^bits.  bits in {2, 4}.
 * packed: int32, m*d*bits/32 words in `layout`. */
int quipamd_pack(const uint8_t *codes, int bits, int layout, int32_t *packed, int64_t m, int64_t d, void *stream);
int quipamd_unpack(const int32_t *packed, int bits, int layout, uint8_t *codes, int64_t m, int64_t d, void *stream);

/* bits = 3 with the CANONICAL layout is the reference's own 3-bit rule, Quant3Linear.pack (quant.py:192-220): 32 codes in 3
 * words, int32 [d/32*3, m] (the reference packs 1024 columns at a time; any d % 32 == 0 is accepted).
 * quipamd_repack_canonical_to_stream: a weight packed by the reference (2-bit generalisation, 3-bit or 4-bit CANONICAL)
 * becomes the STREAM layout K2 reads, on the device, no host pass (3-bit codes land in the 4-bit STREAM container).
 * stream_out: m*d*cb/32 int32 words, cb = 4 for bits 3.  Needs m % 16 == 0 and d % (512/cb) == 0. */
int quipamd_repack_canonical_to_stream(const int32_t *canonical, int bits, int32_t *stream_out, int64_t m, int64_t d,
                                       void *stream);

/* ---- K5: grid map / grid functions ---------------------------------------------------------
 * quipamd_qfnb_scale: scale = 2.4*sqrt(mean(W^2)) + 1e-16 evaluated in W's dtype
 *   (quant.py:150, vector_balance.py:522).  scale_out: device float[1].
 *   workspace: device double[1] (zeroed by the call). */
int quipamd_qfnb_scale(const void *W, int dtype, int64_t numel, float *scale_out, double *workspace, void *stream);

/* quipamd_gridmap: real-valued grid coordinates handed to LDLQ (no rounding):
 *   qfn a: clamp(w/scale[r] + zero[r], 0, maxq)            vector_balance.py:515  (fp32)
 *   qfn b: clamp(((w/s + 1)/2)*maxq, 0, maxq) in W's dtype  vector_balance.py:523-524
 * scale: device float[m] (a) or float[1] (b); zero: float[m] (a) or NULL (b). Wgrid: float [m,d]. */
int quipamd_gridmap(const void *W, int dtype, int qfn, const float *scale, const float *zero, int maxq,
                    float *Wgrid, int64_t m, int64_t d, void *stream);

/* quipamd_quantize: round-to-nearest through the grid (Quantizer.quantize, quant.py:144-157).
 *   codes_out: uint8 [m,d] or NULL; W_out: dequantised weights in `dtype` or NULL. */
int quipamd_quantize(const void *W, int dtype, int qfn, const float *scale, const float *zero, int maxq,
                     uint8_t *codes_out, void *W_out, int64_t m, int64_t d, void *stream);

/* quipamd_codes_to_weight: integer codes -> weights (vector_balance.py:519-520, 528-530):
 *   qfn a: scale[r]*(q - zero[r]);  qfn b: ((q/maxq)*2 - 1)*s;  fp32 math, stored as out_dtype. */
int quipamd_codes_to_weight(const uint8_t *codes, int qfn, const float *scale, const float *zero, int maxq,
                            void *W_out, int out_dtype, int64_t m, int64_t d, void *stream);

/* ---- K2: fused dequant-GEMM ------------------------------------------------------------------
 * Replaces quant_cuda.vecquant3matmul / vecquant4matmul (quant.py:229, zeroShot/models/quant.py:207):
 *     y[b, r] (+)= bias[r] + sum_k What[r, k] * x[b, k]
 * with What dequantised on the fly from `qweight` (STREAM layout):
 *     qfn a: What = scale[r]*(q - zero[r])      (quant.py:186-191: zeros = zero*scale)
 *     qfn b: What = ((q/maxq)*2 - 1)*scale[0]   (quant.py:13-14)
 * x: [bs, d] bf16 or fp16 row-major (the reference widens x to fp32, quant.py:226-229: an fp16 model keeps its
 *    activation bits on the fp16 MFMA pipe, bf16 is for bf16 models); y: [bs, m] in y_dtype (x's dtype or F32).
 * accumulate != 0 (F32 y only): y += result, the reference's in-place contract (quant.py:226-230);
 * bias: float[m] or NULL.  Unlike the reference (single token only, quant.py:233) any bs >= 1. */
int quipamd_dequant_gemm(const void *x, int x_dtype, const int32_t *qweight, int bits, int layout, int qfn,
                         const float *scale, const float *zero, const float *bias, void *y, int y_dtype,
                         int accumulate, int64_t bs, int64_t m, int64_t d, void *stream);

/* The reference's native entry points by name and argument meaning (quant.py:229, zeroShot/models/quant.py:207):
 *     mul[r] += sum_k (scales[r] * q[r,k] - zeros[r]) * vec[k]
 * vec: float [d] (one token); mat: the reference's CANONICAL packing ([d/32*3, m] 3-bit, [d/8, m] 4-bit); mul: float [m],
 * pre-filled by the caller (with the bias) and accumulated into; scales: float [m]; zeros: float [m] = zero * scale.
 * Adapters over quipamd_repack_canonical_to_stream + quipamd_dequant_gemm: the repacked weights, the split activations and
 * the integer zeros live in `workspace` (quipamd_vecquant_workspace_bytes).  The entry points are STATELESS: the O(m d) repack
 * runs on every call, so the same pointers with new contents give the new result -- unless the caller opts in:
 *   quipamd_vecquant_prepare(bits, mat, m, d, workspace, workspace_bytes, stream) repacks once and registers (host side)
 *   workspace -> (mat, bits, m, d, stream); calls with exactly that layer, workspace and stream then skip the repack (a decode loop
 *   pays it once, not per token).  The registration is the caller's promise that `mat` does not change: after rewriting `mat` IN
 *   PLACE, freeing it, or using the workspace for anything else, prepare again or call quipamd_vecquant_invalidate(workspace)
 *   (NULL: forget every workspace). */
int64_t quipamd_vecquant_workspace_bytes(int bits, int64_t m, int64_t d);
int quipamd_vecquant_prepare(int bits, const int32_t *mat, int64_t m, int64_t d, void *workspace, int64_t workspace_bytes, void *stream);
void quipamd_vecquant_invalidate(const void *workspace);
int quipamd_vecquant3matmul(const float *vec, const int32_t *mat, float *mul, const float *scales, const float *zeros,
                            int64_t m, int64_t d, void *workspace, int64_t workspace_bytes, void *stream);
int quipamd_vecquant4matmul(const float *vec, const int32_t *mat, float *mul, const float *scales, const float *zeros,
                            int64_t m, int64_t d, void *workspace, int64_t workspace_bytes, void *stream);

/* quipamd_dequant_gemm_cfg: the same call with the kernel chosen by the caller instead of the shape heuristic -- for
 * benchmarks and the forced-kernel parity tests; never needed for correctness.  cfg = int32[4] {family, p1, p2, p3}
 * (NULL or all 0 = heuristic): family 1 = round-1 kernels; 2 = "h" (bs <= 16, d <= 4096: p1 = waves, p2 = chunks per
 * wave, [3] = row tiles per workgroup or 0); 3 = "s" (bs <= 16 weight stream: p1 = row tiles per workgroup, p2 = k-split); 4 = "mb" (bs > 16:
 * p1 = 45 | 23 = the 256 x 128 / 128 x 64 workgroup tile with four loader waves, the defaults; 44 | 22 the two-loader forms of rounds 2-4; lab
 * forms, 2 bit only: 46 three loaders, 48 | 49 one compute wave per SIMD with 4 x 8 tiles, 47 the 256 x 128 tile on v_mfma_f32_32x32x16).  (Family 5, the round-3 prefill kernel, lost to "mb" at every shape and is no longer in the library:
 * scripts/dqgemm_pf_lab.hip.)  An unsupported combination fails with QUIPAMD_ERR_UNSUPPORTED.  Per call, thread safe. */
int quipamd_dequant_gemm_cfg(const void *x, int x_dtype, const int32_t *qweight, int bits, int layout, int qfn,
                             const float *scale, const float *zero, const float *bias, void *y, int y_dtype,
                             int accumulate, int64_t bs, int64_t m, int64_t d, const int32_t *cfg, void *stream);

/* quipamd_dequant_gemm_grouped: ngroups (1..4) independent problems of IDENTICAL shape (bs, m, d, bits, qfn, dtypes) in
 * one launch -- the q / k / v projections of a decoder block.  Every pointer argument of quipamd_dequant_gemm becomes a
 * HOST array of ngroups device pointers (zero / bias may be NULL arrays or hold NULL entries as in the single call). */
int quipamd_dequant_gemm_grouped(int ngroups, const void *const *x, int x_dtype, const int32_t *const *qweight, int bits,
                                 int layout, int qfn, const float *const *scale, const float *const *zero,
                                 const float *const *bias, void *const *y, int y_dtype, int accumulate, int64_t bs,
                                 int64_t m, int64_t d, void *stream);

/* Tuning hook for benchmarks: force the K2 workgroup shape (row tiles per workgroup, batch tiles per wave,
 * waves per workgroup, k-slices over workgroups); 0 = leave that parameter to the built-in shape heuristic.
 * bt != 0 selects the multi-batch-tile kernel also for bs <= 16; `split` carries two fields, split % 100 = k-slices
 * (> 1 only takes effect under the accumulate contract: fp32 atomics) and split / 100 = chunk groups in flight per
 * workgroup (1, 2 or 4).  Applies to the round-1 kernels and to the CALLING THREAD only; never needed for correctness.  An
 * unsupported combination makes quipamd_dequant_gemm fail with QUIPAMD_ERR_UNSUPPORTED. */
int quipamd_tune_dequant_gemm(int rt, int bt, int nw, int split);
/* tests / A-B runs: which kernel serves quipamd_dequant_gemm_grouped's fp16 problems of d = 4096 (q / k / v, gate / up of a 5..16-row decode
 * step).  0: the heuristic; 1 | 2 | 4: the grouped h kernel with that many row tiles per workgroup; 74 | 72 | 81: the grouped weight-stream
 * kernel (dq_sg_kernel) with 4 | 7 | 8 row tiles per workgroup.  Process-wide. */
void quipamd_dequant_gemm_grouped_config(int form);

/* ---- K3: structured orthogonal (two-factor butterfly / Kronecker) apply ------------------------
 * Replaces mul_ortho_butterfly (method.py:46-67) and the dense U @ W @ V^T, V @ H @ V^T products of
 * QuantMethod.preproc/postproc (method.py:175-176, 202-203) without materialising U or V.
 * Applies the n x n operator  Q = P_out * S1 * S0 * P_in  (or Q^T) to every ROW of x: out[r, :] = Q * x[r, :],
 * n = p*q, z viewed as [p][q] (pos = a*q + b); S0 mixes a per b with B0[b] (p x p), S1 mixes b per a with B1[a] (q x q);
 * blocked = 0: one B0 and one B1 for all b / a (Kronecker product, method.py:38-39).
 * The call runs two "mix one index" stages (first, second) on the fp32 matrix pipe:
 *   b_first = 0 (Q):   first = mix a with M_c = B0[c],   second = mix b with M_c = B1[c]
 *   b_first = 1 (Q^T): first = mix b with M_c = B1[c]^T, second = mix a with M_c = B0[c]^T
 *   frag_first / frag_second: the stage's matrices M_c[i][k] (out index i, in index k; P x P, P = p or q) in MFMA
 *     B-fragment order: float [C][NT][NT][64][4], C = number of matrices (q|p if blocked else 1), NT = ceil(P/16),
 *     element [c][nt][S][lane][s] = M_c[16*nt + (lane & 15)][16*S + 4*(lane >> 4) + s], zero outside P x P
 *     (quip_amd/ops.py OrthoOp builds them);
 *   gather_idx:  int32 [n] or NULL (identity): z[pos] = x[r, gather_idx[pos]]   (Q: perm_in;        Q^T: argsort(perm_out))
 *   scatter_idx: int32 [n] or NULL:            out[r, scatter_idx[pos]] = z[pos] (Q: argsort(perm_out); Q^T: perm_in)
 *     with perm_in, perm_out the torch.randperm values of method.py:35;
 *   colscale: float[n] or NULL -- x[r, k] is multiplied by colscale[k] on load (the x (/) s step of the packed
 *     layer);
 *   x: [rows, n] leading dimension ldx, out: [rows, n] with ldo; dtypes F32 / F16 / BF16 (any pair);
 *   workspace: float [16*ceil(rows/16), n], the fp32 intermediate between the two stages. */
int quipamd_ortho_apply_rows(const float *frag_first, const float *frag_second, int blocked, const int32_t *gather_idx,
                             const int32_t *scatter_idx, int p, int q, int b_first, const float *colscale,
                             const void *x, int x_dtype, int64_t ldx, void *out, int out_dtype, int64_t ldo,
                             int64_t rows, float *workspace, void *stream);

/* quipamd_ortho_apply_small: the same operator for a FEW rows (activation side of the packed layer, batch 1..64) with
 * one factor per stage (blocked = 0, the Kronecker form) in ONE launch: a workgroup keeps the row and both factor
 * matrices in LDS.  M0: float [p][p] (out a, in a'), M1: float [q][q] -- pass B0, B1 for Q and B0^T, B1^T with
 * b_first = 1 for Q^T.  load_idx: int32 [n] or NULL, input element k lands at z position load_idx[k]
 * (Q: argsort(perm_in); Q^T: perm_out); store_idx: int32 [n] or NULL, output element k is z[store_idx[k]]
 * (Q: perm_out; Q^T: argsort(perm_in)).  colscale / bias: float [n] or NULL (input scale, output offset).
 * Requires p, q multiples of 16, n <= 16384 and (p*(p+4) + q*(q+4) + 2*p*(q+4) + 16)*4 bytes <= 160 KiB; otherwise use
 * quipamd_ortho_apply_rows. */
int quipamd_ortho_apply_small(const float *M0, const float *M1, const int32_t *load_idx, const int32_t *store_idx,
                              int p, int q, int b_first, const float *colscale, const float *bias,
                              const void *x, int x_dtype, int64_t ldx, void *out, int out_dtype, int64_t ldo,
                              int64_t rows, void *stream);

/* The BLOCKED butterfly operator (method.py:34-35 gen_rand_ortho_butterfly: B0 [q, p, p], B1 [p, q, q]; what opt.py's --incoh_processing
 * really selects, opt.py:596) applied to a handful of rows -- the decode step of a model quantised by the shipped flag (csrc/ortho_blk.hip).
 *     out = [relu]( Q . ( colscale * Norm( silu(x) * gate_up | x ) ) + bias + residual ),      Q = P_out S1 S0 P_in  or its transpose
 * Two launches (the two mixing stages meet all-to-all), workgroup = (factor block, 16 output rows), factors streamed once as fp16,
 * activations as fp16 hi + lo, fp32 accumulate: ~3e-4 per stage (the tolerance class of the fused decode launches, NOT the 1e-3
 * contract of quipamd_ortho_apply_rows, which stays the path of quantisation and of every caller that does not ask for this one).
 *   F_first / F_second: fp16 [G][P][P] (out index, in index) of the stage that runs first / second; first_mixes_a != 0: the first stage
 *     mixes the index a (G = q blocks of p x p: B0 for Q), the second the index b (G = p blocks of q x q: B1); 0: the other way round
 *     (Q^T: first B1[a]^T, then B0[b]^T).  Image position of (a, b) = a q + b.
 *   in_idx: int32 [n] or NULL, image position pos reads x[in_idx[pos]] (Q: perm_in; Q^T: argsort(perm_out));
 *   out_idx: int32 [n] or NULL, image position pos is written to out[out_idx[pos]] (Q: argsort(perm_out); Q^T: perm_in).
 *   x [rows, ld_x] of x_dtype; gate_up: same dtype / stride or NULL; norm: 0 none, 1 LayerNorm, 2 RMSNorm (HF's: x rsqrt(mean x^2 + eps)
 *   rounded to x_dtype, then gamma) with fp16 ln_gamma / ln_beta [n] in natural order; colscale / bias: float [n] or NULL;
 *   residual [rows, ld_residual] of residual_dtype or NULL; out [rows, ld_out] of out_dtype; rows <= 64 (16 per workgroup, the groups side by side in the launch);
 *   workspace: float [rows * p * q].  p, q multiples of 16, <= 768 (quipamd_ortho_blocked_supported). */
typedef struct quipamd_blk_op {
    const void *F_first, *F_second;
    int first_mixes_a, p, q;
    const int32_t *in_idx, *out_idx;
    const void *x;
    int x_dtype;
    int64_t ld_x;
    void *out;
    int out_dtype;
    int64_t ld_out, rows;
    const void *gate_up;
    int norm;
    const void *ln_gamma, *ln_beta;
    float ln_eps;
    const float *colscale, *bias;
    const void *residual;
    int residual_dtype;
    int64_t ld_residual;
    int relu;
} quipamd_blk_op;
int quipamd_ortho_blocked_supported(int p, int q);
int quipamd_ortho_blocked_rows(const quipamd_blk_op *op, void *workspace, void *stream);
/* nops (1..3) operators of one shape / row count / dtypes / orientation in the SAME two launches (q / k / v, gate / up): `ops` a host array,
 * workspace float [nops * rows * p * q]. */
int quipamd_ortho_blocked_rows_multi(const quipamd_blk_op *ops, int nops, void *workspace, void *stream);
/* Up to n = p q = max_fused_n (default 2048; the rows, the permutation and the chunk partials must fit a workgroup's LDS, pointers and row
 * strides 16-byte aligned) the two stages run as ONE launch: the workgroups of the second stage compute the slice of the first stage they
 * read in their prologue (csrc/ortho_blk.hip).  Every workgroup then reads ~8-10 n bytes from L2, which from n = 4096 on costs more than
 * the launch it saves (measured, profiles/r04k_decode_engine.jsonl), and only up to max_fused_rows rows (default 4: the prologue's work
 * grows with the rows, profiles/r04n).  quipamd_ortho_blocked_config(0, 0) forces the two-launch form. */
void quipamd_ortho_blocked_config(int max_fused_n, int max_fused_rows);

/* One small-batch operator application with the elementwise work of its neighbours in the decoder block fused in
 * (a decode step is launch-latency bound):  out = [relu]( Q . ( colscale * [LayerNorm](x) ) + bias + residual ).
 * Fields as in quipamd_ortho_apply_small; ln_gamma / ln_beta (dtype ln_dtype, ln_gamma NULL = no normalisation, statistics in
 * fp32 over the row with ln_eps; ln_gamma without ln_beta = RMSNorm, x * rsqrt(mean(x^2) + eps) * gamma, the Llama block's norm); residual ([rows, n], leading dimension ldo, dtype res_dtype, NULL = none). */
typedef struct quipamd_small_op {
    const float *M0, *M1;
    const int32_t *load_idx, *store_idx;
    int p, q, b_first;
    const float *colscale, *bias;
    const void *ln_gamma, *ln_beta;
    float ln_eps;
    int ln_dtype;
    const void *residual;
    int res_dtype;
    int relu;
    const void *x;
    int x_dtype;
    int64_t ldx;
    void *out;
    int out_dtype;
    int64_t ldo;
    /* optional split-bf16 factors: M0 = M0_hi + M0_lo, M1 = M1_hi + M1_lo as bf16 [p][p] / [q][q] (all four or none).
     * When given (p, q multiples of 32, q >= p/2) the mix stages run on the bf16 matrix pipe as hi*hi + hi*lo + lo*hi
     * with fp32 accumulation: ~1e-5 relative error, ~5x the fp32-MFMA rate. */
    const void *M0_hi, *M0_lo, *M1_hi, *M1_lo;
} quipamd_small_op;
#define QUIPAMD_SMALL_MAX_OPS 4
/* nops (1..4) independent ops in ONE launch (the q / k / v projections of a block share their input): `ops` is a HOST
 * array; all ops must share p, q and the dtypes; every op is applied to `rows` rows. */
int quipamd_ortho_apply_small_ops(const quipamd_small_op *ops, int nops, int64_t rows, void *stream);
/* The decode-step form of quipamd_ortho_apply_small_ops: every 16 x 16 tile of every op's p x q output image gets its own
 * workgroup (8 to 32 per operator instead of 1), for the Kronecker shapes of a decode step: p x q = 64x32 (n = 2048),
 * 64x64 (4096), 128x64 (8192) -- quipamd_ortho_apply_tiles_supported(p, q) -- with split-bf16 factors.  Same descriptor,
 * same arithmetic (bit-identical to quipamd_ortho_apply_small_ops without LayerNorm; the LayerNorm statistics are summed in a
 * different order).  store_inv[i]: int32 [n], the inverse of ops[i].store_idx (image position -> output index); NULL entries
 * (or a NULL array) exactly where store_idx is NULL.  rows <= 65535. */
int quipamd_ortho_apply_tiles(const quipamd_small_op *ops, const int32_t *const *store_inv, int nops, int64_t rows, void *stream);
int quipamd_ortho_apply_tiles_supported(int p, int q);

/* The same for a Kronecker operator p x 16 with a large p (Llama's 11008 = 688 x 16: neither factor fits a workgroup's LDS, the
 * general quipamd_ortho_apply_rows pads one row to 16): one workgroup per 16 rows of the p x 16 image (p / 16 workgroups per
 * operator and row), fp32 MFMA, factors M0 [p, p] / M1 [16, 16] as for quipamd_ortho_apply_small.  Operand sets: (x f16, colscale)
 * or (x f32, bias, [f16 residual], relu); both permutations and their store_inv; no normalisation.  p % 16 == 0, 64 <= p <= 768.
 * Activation side with `residual` (f16, row stride ldx) and relu = 1: the input row is silu(x) * residual -- the Llama MLP's
 * act_fn(gate) * up computed on load (llama's down_proj input), rounded to f16 after the silu and after the product like torch. */
int quipamd_ortho_apply_bigp(const quipamd_small_op *ops, const int32_t *const *store_inv, int nops, int64_t rows, void *stream);
int quipamd_ortho_apply_bigp_supported(int p, int q);

/* Chain of two operator applications in ONE launch (decode: U^T y + bias + residual -> [LayerNorm] -> V (x (/) s) of two
 * consecutive packed layers):  t = epilogue_first(Q_first x_first)  is stored to first->out (when not NULL) in
 * first->out_dtype and, rounded to that dtype, is the input of every second[i] (1..3 ops that share it, e.g. the q / k / v
 * projections), whose `x` / `x_dtype` / `ldx` fields are ignored.  All ops share p, q (split-bf16 factors required); first->x
 * must be f32; the second ops share the output dtype.  Bit-identical to the two separate launches. */
int quipamd_ortho_apply_small_chain(const quipamd_small_op *first, const quipamd_small_op *second, int nsecond, int64_t rows,
                                    void *stream);

/* Decode: the dequant-GEMM with the activation-side operator in its prologue (one launch instead of
 * quipamd_ortho_apply_small_ops + quipamd_dequant_gemm_grouped):
 *   y[i][b, :] = What_i ( V_i ( [LayerNorm](x[b, :]) (/) s_i ) ) + bias_i      i < ngroups <= 3, 2-bit STREAM codes, qfn b,
 * for d = 2048 (the operator must be 64 x 32 with split-bf16 factors), bs <= 8, m % 32 == 0, fp32 y [bs, m].
 * vops[i]: the V-side descriptor (x f16 / bf16, ln_*, colscale, factors, index vectors; out / bias / residual unused).
 * Bit-identical to the two separate launches. */
int quipamd_dequant_gemm_vop(const quipamd_small_op *vops, const int32_t *const *qweight, const float *const *scale,
                             const float *const *bias, float *const *y, int ngroups, int bits, int64_t bs, int64_t m, void *stream);

/* ---- decode: one launch per packed Linear group, everything between two GEMMs in the consumer's prologue ------------------------
 * (csrc/decode_fused.hip; the decode loop of benchmark(), opt.py:431-482 / llama.py:418-471, batch <= 4)
 *     t    = [relu]( U_prev^T u_y + u_bias + u_residual )      optional (has_u): the output side of the PREVIOUS packed layer;
 *                                                              u_y fp16 [bs, d] (a fused launch writes it with y_dtype F16),
 *                                                              u_bias fp16 [d] (zeros where the layer has none), u_residual fp16
 *                                                              [bs, ld_residual] or NULL; kernels exist for the combinations a decoder
 *                                                              block needs (see dispatch_fused in csrc/decode_fused.hip);
 *                                                              t is stored to t_out (fp16, the new residual stream) when not NULL
 *     h    = norm(t)  (norm 0: none, 1: LayerNorm gamma / beta, 2: RMSNorm gamma; fp32 statistics, eps)      [t = x without has_u]
 *     x~_i = V_i ( h (/) s_i ),   y_i = What_i x~_i            i < ngroups <= 3 (q / k / v; gate / up), fp32 y [bs, m]
 * PERMUTATIONS FOLDED INTO THE PACKING (free at pack time): the decode launches take the packed codes of a layer with
 *   - its ROWS in "ZT order" of its own output-side operator U: row b * p + a of the packed matrix is the row i of the layer with
 *     load position pout_U[i] = a * q + b, so that the GEMM's output vector IS the transposed image U^T starts from (the consumer's
 *     scatter is a straight 16-byte copy; u_y and the y of quipamd_decode_attention_fused / quipamd_decode_u_only are in this order);
 *   - its COLUMNS in image order of its own activation-side operator V: column a * q + b is the column k with pout_V[k] = a * q + b,
 *     so that the image V produces IS x~ (written straight into the GEMM's operand row; no gather).
 *   quip_amd.quant.QuantLinear.decode_qweight() builds that copy of the codes once (unpack, index, pack).
 * quipamd_fop = a Kronecker operator prepared for this kernel: F0 / F1 are the two factor matrices of the wanted orientation
 * (M0 [p][p], M1 [q][q], out = (M0 (x) M1) applied to the p x q image) as FP16 in MFMA B-fragment order
 *     F0[((at * p/32 + S) * 64 + lane) * 8 + e] = M0[16 at + lane % 16][32 S + 8 (lane / 16) + e]        (F1 likewise with q),
 * load_idx / store_idx with the meaning they have in quipamd_small_op, as uint16 [n] (an output-side operator needs store_idx only,
 * an activation-side one load_idx only: the other permutation lives in the packing).  One fp16 product per factor entry: ~3e-4 relative per
 * stage, below the 16-bit rounding of the pass's output (x~ feeds the fp16 MFMA, t is the fp16 residual stream).
 * NUMERICAL CONTRACT of the decode launches (this entry point, quipamd_decode_attention_fused, quipamd_decode_u_only, quipamd_decode_bigp_*,
 * quipamd_decode_head, quipamd_ortho_blocked_rows): every y_i within 2e-3 and t within 1e-3 (relative l2) of the same chain evaluated in
 * fp64 from the layer's own tensors (tests/test_gpu_decode_fused.py; measured 4e-4 .. 1.2e-3) -- NOT the 1e-3 of quipamd_dequant_gemm and
 * quipamd_ortho_apply_*: the operator pass runs on fp16 factors and x~ is rounded to fp16 in front of the MFMA, like the fp16 model the
 * reference decodes with.  End to end: logits within 1e-2 of the Hugging Face model holding the dense equivalents, greedy tokens equal
 * (tests/test_gpu_decode_hf.py, test_gpu_decode_e2e.py: measured 2.5e-3 .. 8e-3).  A caller that needs the 1e-3 contract per layer uses
 * QuantLinear.forward on more than 8 rows or the K3 entry points directly (split-bf16 / fp32 operator arithmetic).
 * Shapes: U and V must both be p x q in {64 x 32, 64 x 64, 128 x 64}, d = p q; qfn-b STREAM codes, bits 2, or 4 / 3 (the 4-bit container;
 * fp32 u_y is 2-bit only); scale[i] float [1];
 * colscale[i] float [d] (ones when the layer has no rescale); m % 32 == 0 (64 x 32) or m % 16 == 0; all 16-bit tensors fp16;
 * t_out must not alias u_residual (other workgroups still read it).  `args` is a HOST struct. */
typedef struct quipamd_fop {
    const void *F0, *F1;
    const uint16_t *load_idx, *store_idx;      /* uint16 [n] (n <= 8192): half the registers and bytes of the int32 vectors */
    int p, q;
} quipamd_fop;
typedef struct quipamd_fused_gemm_args {
    int act_dtype, bits;
    int has_u;
    quipamd_fop U;
    const void *u_y, *u_bias;        /* fp16 [bs, d], fp16 [d] */
    const void *u_residual;
    int64_t ld_residual;
    int u_relu;
    void *t_out;
    int64_t ld_t;
    const void *x;
    int64_t ldx;
    int norm;
    const void *ln_gamma, *ln_beta;
    float ln_eps;
    int ngroups;
    quipamd_fop V[3];
    const float *colscale[3];
    const int32_t *qweight[3];
    const float *scale[3];
    void *y[3];
    int y_dtype;                     /* QUIPAMD_F32, or QUIPAMD_F16 when the consumer is another fused launch (its scatter rounds to fp16 anyway) */
    int64_t bs, m;
    /* optional, 128 x 64 with has_u, no residual, no norm, one group, bs <= 2 (OPT's fc1 -> fc2 hand-over): per-lane tables of the LAYER
     * PAIR in the lane order of the MFMA result of U's second stage -- entry [(w * 64 + lane) * 8 + 4 i + reg] belongs to image element
     * (a, b) = (16 at + 4 (lane / 16) + reg, 16 bt + lane % 16) of tile (at, bt) = ((w + 16 i) / 4, (w + 16 i) % 4), i.e. to the natural
     * element k with U.store_idx[k] = a q + b:  pair_sig uint16 = LDS position (pv % q) (p + 8) + pv / q of pv = V.load_idx[k];
     * pair_bias fp16 = u_bias[k];  pair_cs fp16 = colscale[0][k].  With them the gather / scale / scatter between the two operators is
     * one scatter in the epilogue of U's stage 2 (U.store_idx, V[0].load_idx, u_bias and colscale[0] are then not read). */
    const void *pair_sig, *pair_bias, *pair_cs;
    /* dtype of u_y: QUIPAMD_F16, or QUIPAMD_F32 -- the fp32 accumulator of quipamd_decode_bigp_v_gemm, rounded to fp16 on load (what a
     * cast launch in between would do); fp32 has a kernel for 64 x 64 with residual and RMSNorm (Llama's down_proj -> q / k / v) only. */
    int u_y_dtype;
    /* ops_only != 0 (round 5; more than 4 rows per step): the launch stops after the operator chain -- grid (bs, ngroups), one workgroup per
     * (batch row, group) -- and y[k] receives x~_k = V_k(Norm(t) (/) s_k) as fp16 [bs, n] in IMAGE order (= the order of the decode-order
     * codes' columns); qweight / scale / m / y_dtype are not read, t_out is written as usual.  The dequant-GEMM is then one
     * quipamd_dequant_gemm_grouped on those x~ and the same decode-order codes (its output rows are in ZT order like the fused
     * launch's): the weights stream once for all rows.  bs <= 1024. */
    int ops_only;
} quipamd_fused_gemm_args;
int quipamd_decode_fused_gemm(const quipamd_fused_gemm_args *args, void *stream);

/* The same attention launch with the OUTPUT-SIDE operators of the q / k / v projections in its prologue (csrc/decode_attn.hip):
 *     q = U_q^T y_q + b_q,  k = U_k^T y_k + b_k,  v = U_v^T y_v + b_v   (rounded to fp16),  [rotary on q, k],  then as above.
 * U[3]: quipamd_fop records of the TRANSPOSED operators (all p x q with p q = heads hd); y[3]: fp16 [bs, heads hd] in the
 * projected basis (a fused GEMM launch with y_dtype F16); bias[3]: fp16 [heads hd] (zeros where a layer has none);
 * cos_table / sin_table: float [table_rows, hd] (HF's duplicated-halves layout) or both NULL (OPT: learned positions);
 * out: fp16 [bs, ldo].  head_dim 64 with 64 x 32 operators (OPT-1.3B), 128 with 64 x 64 (Llama-2-7B) or 64 x 32. */
int quipamd_decode_attention_fused(const quipamd_fop *U, const void *const *y, const void *const *bias, void *kcache, void *vcache,
                                   const int64_t *pos, void *out, const float *cos_table, const float *sin_table, int64_t table_rows,
                                   int64_t bs, int heads, int hd, int64_t maxlen, float scale, int64_t ldo, void *stream);
/* decode_attention_fused launches one workgroup per (sequence, head): 12 waves, three wave groups running U_q, U_k, U_v side by side, two of
 * which leave after their pass.  From `three_heads_from` (sequence, head) pairs on (default 257: more than one per CU) a workgroup serves THREE
 * heads: the k and v groups stay and each group runs the attention of its own head -- a third of the workgroups and of the (redundant) operator
 * passes; 0 = never, negative = the default.  `one_group_from`: a 4-wave form whose one wave group runs the three passes in turn (two
 * workgroups per CU); measured slower (csrc/decode_attn.hip), default 0 = never.  All forms compute the same values in the same order. */
void quipamd_decode_attention_config(int one_group_from, int three_heads_from);

/* The output side of a packed layer on its own (the end of the last decoder block): out = [relu](U^T y + bias + residual), fp16.
 * U: the TRANSPOSED operator (quipamd_fop; 64 x 32, 64 x 64 or 128 x 64); y: fp16 [bs, n] in ZT order (see below); bias fp16 [n];
 * residual fp16 [bs, ld_residual] or NULL; out fp16 [bs, ld_out], not aliasing the residual. */
int quipamd_decode_u_only(const quipamd_fop *U, const void *y, const void *bias, const void *residual, int64_t ld_residual, int relu,
                          void *out, int64_t ld_out, int64_t bs, void *stream);

/* ---- decode around a packed layer whose operator is p x 16 with a LARGE p (csrc/decode_bigp.hip; Llama's 11008 = 688 x 16,
 * method.py:16-18 butterfly_factors): the MLP tail of llama.py:418-471's loop,  down_proj(silu(gate_proj(x)) * up_proj(x)), as two launches.
 * Both take their input vector as the TRANSPOSED image, row-major: index b * p + a holds image position (a, b) -- for an output-side
 * operator that is the "ZT order" the producing GEMM's rows are packed in (see quipamd_fused_gemm_args), for the activation-side
 * operator it is what quipamd_decode_bigp_u writes through `dest`.  Factors: F0 = M0 [p][p] as fp16 MFMA B fragments with the k index
 * zero padded to ks = ceil(p / 32) steps:  F0[((at * ks + S) * 64 + lane) * 8 + e] = M0[16 at + lane % 16][32 S + 8 (lane / 16) + e]
 * (0 where the column index >= p);  M1 = float [16][16];  result image = M0 z M1^T.  p % 16 == 0, 64 <= p <= 1024.  Rows (batch):
 * 1..16 -- quipamd_decode_bigp_u walks row groups of 4 side by side; quipamd_decode_bigp_v_gemm from 5 rows on mixes 4 rows
 * at a time, ONE pass over the weights feeds all rows).
 *
 * quipamd_decode_bigp_u: for every operator i (<= 3; gate and up in one launch)
 *       out_i[r][dest_i[pos]] = fp16( ((M0 z M1^T)[pos] + bias_img_i[pos]) * post_img_i[pos] )        pos = a * 16 + b
 *   y: fp16 [rows, n] transposed image; bias_img (fp16 [n]) / post_img (float [n]) in IMAGE order or NULL; dest: uint16 [n], any map
 *   (the host composes the operator's own store permutation with the consumer's layout).  `clear` (NULL or a 16-byte aligned float
 *   buffer of clear_n floats, clear_n % 4 == 0) is zeroed by the launch: the accumulator of the quipamd_decode_bigp_v_gemm that follows.
 * quipamd_decode_bigp_v_gemm:  t = silu(gate) * up  (up NULL: t = gate), rounded to fp16 like the two torch launches it replaces;
 *       x~ = M0 t M1^T (fp16, image order);   y[r][:] += scale * (2 / 3) * (codes - 1.5) x~        2-bit qfn-b STREAM codes [m, n]
 *   with the COLUMNS of the codes in image order of the operator (column a * 16 + b multiplies x~ at image position (a, b)).
 *   One workgroup per (16 image rows = 256 columns, 256 * row_tiles_per_wave rows); the K-slices are ADDED into y (float [rows, m])
 *   with fp32 atomics: y must be zero (or hold what is to be accumulated onto) on entry, and the summation order is not fixed
 *   (results differ in the last bits from run to run).  m % 256 == 0; row_tiles_per_wave 0 (choose) / 1 / 2 / 4. */
typedef struct quipamd_bigp_u_op {
    const void *F0;
    const float *M1;
    const void *y;
    const void *bias_img;
    const float *post_img;
    const uint16_t *dest;
    void *out;
    int64_t ld_out;
} quipamd_bigp_u_op;
typedef struct quipamd_bigp_v_gemm_args {
    const void *F0;
    const float *M1;
    const void *gate, *up;           /* fp16 [rows, ldx] */
    int64_t ldx;
    const void *qweight;
    const float *scale;              /* float [1] */
    int bits;                        /* 2 */
    float *y;
    int64_t m;
    int p;
    int64_t rows;
    int row_tiles_per_wave;
    float *partials;                 /* NULL: the K-slices meet in y through fp32 atomics (y zero on entry).  Else fp32 [p/16, rows, m]
                                        scratch, 16-byte aligned like y: every slice stores its partial and a second launch of the same call
                                        sums them in slice order into y (deterministic; y need not be cleared; faster than the atomics
                                        from 5 rows on)                                                                                */
    void *xt;                        /* NULL, or fp16 [rows, 16 p] scratch, 16-byte aligned: the TWO-LAUNCH form -- the operator pass alone writes
                                        x~ there (one workgroup per 16 image rows and 4 batch rows), then quipamd_dequant_gemm runs on it and
                                        the same codes; y is STORED (deterministic, need not be cleared).  The form for 5..16 rows: the
                                        one-launch kernel repeats the whole activation-side pass in every workgroup (49 us at 16 rows).  */
} quipamd_bigp_v_gemm_args;
int quipamd_decode_bigp_supported(int p, int q);
int quipamd_decode_bigp_u(const quipamd_bigp_u_op *ops, int nops, int p, int64_t rows, float *clear, int64_t clear_n, void *stream);
int quipamd_decode_bigp_v_gemm(const quipamd_bigp_v_gemm_args *args, void *stream);

/* ---- the two ends of a decode step around the decoder blocks (csrc/decode_head.hip; benchmark(), opt.py:431-482 / llama.py:418-471:
 * one forward per token, logits of the last position, `torch.argmax` picks the next token), each as one launch.
 * quipamd_decode_head:   t = U^T u_y + u_bias + u_residual  (has_u: the output side of the last packed layer, as in
 *   quipamd_decode_fused_gemm: U the TRANSPOSED operator 64 x 32 / 64 x 64, u_y fp16 or fp32 [bs, n] in ZT order)   or   t = x;
 *   h = LayerNorm (norm 1) / RMSNorm (norm 2) of t, rounded to fp16;   logits[r][v] = fp16( sum_k W[v][k] h[r][k] )   W fp16 [vocab, n];
 *   part_val / part_idx [bs, nparts] (or both NULL): per workgroup the maximum of its rows' fp16 logits and the smallest row index that
 *   attains it (index 0x7fffffff = the workgroup had no rows); nparts = the number of workgroups (0: 256); pos_inc: NULL, or a device
 *   counter the launch increments by one when it is done (the step's position).  n = 2048 or 4096, bs <= 4.
 * quipamd_decode_embed:  ids[r] = argmax over part_val[r][:] (ties: smallest index; entries with part_idx < 0 or 0x7fffffff are
 *   skipped; when none is valid -- the first token -- ids[r] is kept as the caller set it), then
 *   out[r][:] = tok_table[ids[r]][:] + pos_table[*pos + pos_offset][:]   (pos_table NULL: no position embedding), fp16. */
typedef struct quipamd_head_args {
    int has_u;
    quipamd_fop U;
    const void *u_y;
    int u_y_dtype;
    const void *u_bias, *u_residual;
    int64_t ld_residual;
    const void *x;
    int64_t ldx;
    int norm;
    const void *ln_gamma, *ln_beta;
    float ln_eps;
    int64_t n, bs;
    const void *W;
    int64_t vocab;
    void *logits;
    int64_t ld_logits;
    float *part_val;
    int *part_idx;
    int nparts;
    int64_t *pos_inc;
} quipamd_head_args;
int quipamd_decode_head(const quipamd_head_args *args, void *stream);
int quipamd_decode_embed(const void *tok_table, int64_t vocab, const void *pos_table, int64_t positions, int64_t pos_offset,
                         const int64_t *pos, int64_t *ids, const float *part_val, const int *part_idx, int nparts, int64_t n,
                         void *out, int64_t ld_out, int64_t bs, void *stream);

/* Greedy token of a decode step: out[r] = argmax_i x[r, i] (int64, DEVICE), the smallest index among equal maxima like torch.argmax;
 * x: [rows, n] f32 / f16 / bf16 with row stride ld.  One workgroup per row (benchmark(), opt.py:463-480 picks the next token this way). */
int quipamd_argmax_rows(const void *x, int dtype, int64_t rows, int64_t n, int64_t ld, int64_t *out, void *stream);

/* ---- K4: LDLQ rounding -------------------------------------------------------------------------
 * Replaces round_ldl / round_ldl_block (vector_balance.py:155-199, 218-257; n_greedy_passes = 0):
 *   for i = d-1 .. 0:  q_i = clamp(floor(w_i + sum_{j>i} (w_j - q_j) L[j,i] + eta_i), 0, 2^bits - 1)
 * evaluated in 128-column lazy blocks (far field as an fp32-MFMA product, in-block error feedback
 * broadcast lane to lane).  Rows are independent.
 *   Wgrid: float [m, d] grid coordinates;  LT: float [d, d], LT[c][j] = L[j][c] for j > c where L is
 *   the unit-lower Cholesky factor of H (vector_balance.py:171-173; see quipamd_unit_lower_t);
 *   eta: float [m, d] or NULL (= 0.5, vector_balance.py:174-177);
 *   codes: uint8 [m, d] out;  err_ws: float [m, d] workspace (holds w - q on return).
 * Requires d % 16 == 0. */
int quipamd_ldlq_round(const float *Wgrid, const float *LT, const float *eta, int bits, uint8_t *codes,
                       float *err_ws, int64_t m, int64_t d, void *stream);
/* tests / A-B runs: K4 with 1 or 2 groups of 16 rows per workgroup forced (0: by the row count, 2 from 8192 rows on).  Process-wide. */
void quipamd_ldlq_config(int row_groups);

/* ---- OPTQ / GPTQ rounding on the K4 machinery -----------------------------------
 * Replaces the column loop + lazy block update of GPTQ.fasterquant (gptq.py:56-93, groupsize = -1):
 *   for i = 0 .. d-1:  q_i = clamp(round(w'_i), 0, 2^bits-1);  e_i = (w'_i - q_i) / Hinv[i][i];  w'_j -= e_i Hinv[i][j]  (j > i)
 * in grid coordinates (the per-row scale cancels).  With the raw residual r_i = w'_i - q_i this is the LDLQ kernel's
 * recurrence run over the REVERSED columns with the updated-weight residual fed back through
 *   FT[c'][i'] = -Hinv[i][c] / Hinv[i][i]   (c' = d-1-c, i' = d-1-i, i < c; 0 elsewhere),
 * so the caller passes  Wgrid_rev = Wgrid[:, ::-1],  FT as above (float [d, d], strictly upper), and gets codes_rev
 * (uint8 [m, d], reversed columns) and err_ws (float [m, d], the residuals r, reversed).  Ties round half up
 * (torch.round in the reference rounds half to even).  Requires d % 16 == 0. */
int quipamd_gptq_round(const float *Wgrid_rev, const float *FT, int bits, uint8_t *codes_rev, float *err_ws, int64_t m,
                       int64_t d, void *stream);

/* The same sweep in WEIGHT units with the reference's quantiser inside the chain: what GPTQ.fasterquant does for
 * `groupsize != -1` (gptq.py:69-76) and for qfn 'c' (quant.py:17-21, 162-165).
 *   q = scale * (clamp(round(w' / scale) + zero, 0, maxq) - zero)      (qfn_c: clamp(w' / scale + zero) first, then round)
 * groupsize > 0 (16, 32, 64 or 128, dividing d): `scale`, `zero` are OUT float [m, d / groupsize]; the pair of a group is found
 *   by Quantizer.find_params_qfna (quant.py:57-94; perchannel, mse off, `sym` as configured) from the group's columns as
 *   they stand at the start of its 128-column block -- gptq.py:72-75 reads the block-lazy W, not the in-block W1.
 * groupsize <= 0: `scale`, `zero` are IN float [m] (the quantiser found beforehand on the whole row).
 *   W_rev: float [m, d] weights, columns reversed;  FT: as for quipamd_gptq_round;  Q_rev: float [m, d] out, the dequantised
 *   weights (reversed);  codes_rev: uint8 [m, d] out or NULL;  err_ws: float [m, d] workspace.  Requires d % 16 == 0. */
int quipamd_gptq_round_groups(const float *W_rev, const float *FT, int bits, int groupsize, int sym, int qfn_c, float *scale,
                              float *zero, float *Q_rev, uint8_t *codes_rev, float *err_ws, int64_t m, int64_t d, void *stream);

/* FT for the two entry points above straight from the (damped) Hessian: replaces
 *   Hinv = cholesky(cholesky_inverse(cholesky(H)), upper=True)     (gptq.py:51-54, three rocSOLVER calls behind torch)
 * by  FT = I - (I + N)^-1,  N = quipamd_cholesky_lt of the column-reversed H  (csrc/trinv.hip has the algebra).
 *   H: float [d, d] symmetric positive definite;  FT: float [d, d] out;  work: float [2 d d];  info: int [1] as for
 *   quipamd_cholesky_lt (indices refer to the REVERSED matrix).  Requires d % 16 == 0. */
int quipamd_gptq_feedback(const float *H, float *FT, float *work, int64_t d, int *info, void *stream);

/* X = (I + N)^-1 for a strictly-upper-triangular N (only the strict upper part of N is read; X comes back upper
 * triangular with a unit diagonal, its strict lower part is NOT written).  N, X, work: three distinct float [d, d]. */
int quipamd_unit_upper_inverse(const float *N, float *X, float *work, int64_t d, void *stream);

/* One greedy coordinate-descent pass of LDLQ's post-processing (round_ldl / round_ldl_block with n_greedy_passes > 0,
 * vector_balance.py:186-196, 263-288), same kernel as quipamd_ldlq_round in its third mode.  For i = d-1 .. 0:
 *   Hs_i = sH[:, i] - sum_{j > i} eps_j H[j][i];   new_i = round(wr_i - Hs_i / H[i][i]);   eps_i = wr_i - new_i
 * where sH = s @ H is the caller's GEMM of the current error s = wr - w with the normalised H (H / max diag H).
 *   wr, sH: float [m, d];  negH_upper: float [d, d], -H[i][j] for j > i, 0 elsewhere;  hdiag: float [d] = diag H;
 *   wr_out: float [m, d] the updated values, NOT clamped (the reference clamps after the pass);  eps: float [m, d] out
 *   (the caller updates s -= eps).  torch.round semantics (half to even).  Requires d % 16 == 0. */
int quipamd_ldlq_greedy_pass(const float *wr, const float *sH, const float *negH_upper, const float *hdiag, float *wr_out,
                             float *eps, int64_t m, int64_t d, void *stream);

/* quipamd_unit_lower_t: from the lower Cholesky factor C (H = C C^T, row-major [d,d]) build
 *   LT[c][j] = C[j][c] * (1 / C[c][c]) for j > c, 0 elsewhere  (vector_balance.py:172-173). */
int quipamd_unit_lower_t(const float *C, float *LT, int64_t d, void *stream);

/* ---- K7: Hessian accumulation ------------------------------------------
 * Replaces QuantMethod.add_batch's  `inp = inp.to(float64); H += inp.matmul(inp.t())`  (method.py:98-120) and
 * post_batch's  `H = (H / nsamples).to(float32)`  (method.py:122-123).
 *   quipamd_hessian_accum:  Hacc[i][j] += sum_t x[t][i] * x[t][j]  in fp64 on the fp64 matrix pipe, for the tiles of
 *     the block-lower triangle only (tiles of 64 or 128 columns with tile row >= tile column; the tiles above it are
 *     left untouched).  x: [tokens, d] f16 / bf16 / f32 with row stride ldx (elements), token-major exactly as the
 *     forward hook receives it (no transpose); Hacc: double [d, d], zero-initialised by the caller.
 *   quipamd_hessian_finish: H[i][j] = (float)(Hacc[max(i,j)][min(i,j)] / nsamples)  -- mirrors the triangle, divides
 *     in fp64 and narrows, like post_batch.  H: float [d, d], must not alias Hacc. */
int quipamd_hessian_accum(const void *x, int x_dtype, int64_t ldx, int64_t tokens, int64_t d, double *Hacc, void *stream);
int quipamd_hessian_finish(const double *Hacc, double nsamples, float *H, int64_t d, void *stream);
/* Opt-in fast variant of quipamd_hessian_accum for f16 / bf16 inputs: exact products on the 16-bit matrix pipe,
 * fp32 partial sums over runs of 128 tokens, fp64 across runs (error ~1e-9 of sqrt(H_ii H_jj) over a calibration pass
 * instead of fp64 round-off; NOT the reference's arithmetic).  Same accumulator contract (block-lower triangle).
 * workspace: quipamd_hessian_fast_workspace(tokens, d) elements of x's dtype, 16-byte aligned (holds x transposed). */
int64_t quipamd_hessian_fast_workspace(int64_t tokens, int64_t d);
int quipamd_hessian_accum_fast(const void *x, int x_dtype, int64_t ldx, int64_t tokens, int64_t d, double *Hacc,
                               void *workspace, void *stream);

/* ---- the elementwise / reduction chains of QuantMethod.preproc (method.py:134-193; csrc/preproc.hip) -------------------------------------
 * quipamd_preproc_rescale (method.py:140-156):   H /= max|H|;   s = clamp(sqrt(sqrt(clamp(diag H, 1e-8) / clamp(diag(W^T W), 1e-8))), 1e-8);
 *   W <- W s[None, :] rounded to its dtype;   H <- (H / s[None, :]) / s[:, None]       -- every operation in the reference's order, IEEE
 *   divisions; the column sums of squares are fp32 sums in a fixed order (64-row partials, then the partials).
 *   H: float [d, d] in place;  W: [m, d] of w_dtype in place;  s_out: float [d];  workspace: quipamd_preproc_workspace_bytes(m, d).
 * quipamd_preproc_trace_ridge (method.py:165):   H <- H * (d / (trace(H) + 1e-8)) + ridge * I,   in place; workspace >= 4 bytes. */
int64_t quipamd_preproc_workspace_bytes(int64_t m, int64_t d);
int quipamd_preproc_rescale(float *H, void *W, int w_dtype, int64_t m, int64_t d, float *s_out, void *workspace, void *stream);
int quipamd_preproc_trace_ridge(float *H, int64_t d, float ridge, void *workspace, void *stream);

/* OPTQ / GPTQ with the qfn-b quantiser (`--quant gptq --incoh_processing`; gptq.py:56-93 with quant.py:10-15,158-160): every column is
 * rounded on ITS OWN scale 2.4 sqrt(mean over all m rows of w'^2) + 1e-16, w' the column after the feedback of all earlier columns --
 * d grid-wide reductions in series (csrc/gptq_qfnb.hip: co-resident workgroups exchange their partial sums through data-tagged granules).
 *   q = scale * ((clamp(round((w' / scale + 1) / 2 * maxq), 0, maxq) / maxq) * 2 - 1);   r = w' - q;   w'_j += r FT[j'][c']  (j' < c')
 * in the coordinates of quipamd_gptq_round (columns reversed, FT strictly upper), with W, Q held TRANSPOSED:
 *   WT_rev: float [d, m] in / scratch (updated in place);  QT_rev: float [d, m] out;  colscale_rev: float [d] out, the scale of every
 *   column;  workspace: quipamd_gptq_qfnb_workspace_bytes(m, d) bytes.  m <= 32768 (at most 256 workgroups wait for each other: the
 *   launch assumes they are all resident, i.e. the GPU is not shared with another long-running grid).  Deterministic.
 * Co-residency is inferred from the occupancy query, not guaranteed: NOTHING ELSE may hold compute units of the device while the sweep
 * runs (another stream's grid, an RCCL collective of a sharded run, a CU-masked queue).  The wait is therefore BOUNDED: a workgroup that has
 * polled one granule ~4 M times (seconds) raises an abort word in the workspace and every workgroup of the sweep leaves; the results are
 * then undefined.  The call is asynchronous, so the caller reads that word once the stream has drained:
 *   int32 at byte quipamd_gptq_qfnb_info_offset(m, d) of `workspace`: 0 = the sweep completed, 1 = abandoned (treat as QUIPAMD_ERR_LAUNCH;
 *   quip_amd.ops.gptq_round_qfnb raises, quip_amd.gptq falls back to the column walk), 2 = NOTHING was written (WT_rev, QT_rev,
 *   colscale_rev untouched): up to 16384 rows the sweep runs on the workgroups of ONE XCD (their exchange goes through that XCD's L2: 0.55 us
 *   per column instead of 1.5-2.4), and fewer workgroups than it needs became resident there -- call quipamd_gptq_qfnb_debug(.., .., 1)
 *   and repeat the call (the exchange across the XCDs), as quip_amd.ops.gptq_round_qfnb does.
 * quipamd_gptq_qfnb_debug(short_grid, spin_limit, force_rows): test / lab hook -- `short_grid` workgroups too few take part and the rest
 * give up after `spin_limit` polls (exercises the abandon path on a healthy device); force_rows = 1: the pipelined chain across the XCDs,
 * 2: on one XCD (an error beyond 16384 rows), 16 | 32 | 64 | 128: the barrier-per-phase chain of rounds 3-5 with that many rows per
 * workgroup.  (0, 0, 0) restores the defaults. */
int64_t quipamd_gptq_qfnb_workspace_bytes(int64_t m, int64_t d);
int64_t quipamd_gptq_qfnb_info_offset(int64_t m, int64_t d);
void quipamd_gptq_qfnb_debug(int short_grid, int64_t spin_limit, int force_rows);
int quipamd_gptq_round_qfnb(float *WT_rev, const float *FT, int bits, float *QT_rev, float *colscale_rev, void *workspace, int64_t m,
                            int64_t d, void *stream);

/* ---- K8: LDL factor for LDLQ ---------------------------------------------------------------------------
 * Replaces `L = torch.linalg.cholesky(H); L = L @ diag(1/diag(L))` (vector_balance.py:171-173) plus the transpose K4
 * wants:  LT[c][j] = U[c][j] * (1 / U[c][c]) for j > c, 0 elsewhere, where H = U^T U (U upper = C^T).
 * Blocked right-looking fp32 factorisation of the upper triangle, in place in LT (H is copied first; H == LT allowed).
 *   H: float [d, d] symmetric positive definite (only the upper triangle is read);  LT: float [d, d] out;
 *   info: DEVICE int, 0 on success, else 1 + the first column whose pivot was not positive (LAPACK potrf convention) --
 *   the factor is then meaningless. */
int quipamd_cholesky_lt(const float *H, float *LT, int64_t d, int *info, void *stream);
/* tests / A-B measurements (process-wide): old_syrk bit 0 forces the guarded round-1 trailing-update kernel, bit 1 the unblocked 64-step
 * factorisation of the diagonal block; lookahead 0 = never use the
 * two-stream schedule, 1 = from d = 1024, anything else = the default (from d = 12288, where it starts to pay). */
void quipamd_cholesky_config(int old_syrk, int lookahead);

/* Rotary position embedding of one decode step, in place on q [bs, heads * hd] and k [bs, kv_heads * hd] (row strides ldq, ldk)
 * at position *pos (DEVICE memory: the launch can be replayed in a hipGraph while the position advances):
 *   x[i] <- x[i] cos[pos][i] - x[i + hd/2] sin[pos][i];  x[i + hd/2] <- x[i + hd/2] cos[pos][i] + x[i] sin[pos][i]   (i < hd/2)
 * = HF's apply_rotary_pos_emb (q * cos + rotate_half(q) * sin) behind llama.py:418-471.  cos / sin: float [table_rows, hd];
 * a position outside [0, table_rows) leaves q and k untouched (never reads past the tables). */
int quipamd_rope_inplace(void *q, void *k, const float *cos_table, const float *sin_table, int64_t table_rows, const int64_t *pos,
                         int dtype, int64_t bs, int heads, int kv_heads, int hd, int64_t ldq, int64_t ldk, void *stream);

/* ---- single-token decode attention (the decode loop of benchmark(), opt.py:431-482) -------------
 * Replaces the eager HF attention chain of one decode step (cache append, q K^T, scale + causal mask, softmax, p V --
 * nine launches per block) with one launch:
 *   kcache/vcache[b, head, *pos, :] = k/v[b, head*hd : (head+1)*hd];   out[b, head*hd + :] = softmax(scale q.K[0..*pos]^T) V[0..*pos]
 *   q, k, v, out: [bs, heads*hd] f16 / bf16 with row stride ld (elements, multiple of 8); kcache, vcache: [bs, heads, maxlen, hd]
 *   contiguous; pos: DEVICE int64 (so the launch can be replayed from a hipGraph while the position advances);
 *   hd 64 or 128; fp32 scores, softmax and accumulation.  A position outside [0, maxlen) makes the launch a no-op. */
int quipamd_decode_attention(const void *q, const void *k, const void *v, void *kcache, void *vcache, const int64_t *pos,
                             void *out, int dtype, int64_t bs, int heads, int hd, int64_t maxlen, float scale, int64_t ld,
                             void *stream);

#ifdef __cplusplus
}
#endif
#endif /* QUIP_AMD_H */
