/*
 * oracle/ldlq_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's LDLQ rounding loops and of the packed
 * single-token matmul contract, used only by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker.  Nothing under quip_amd/ may
 * link or call this file.
 *
 * Reference (Cornell-RelaxML/QuIP) lines each function follows are cited at the
 * function.  Built by oracle/build.py with:  gcc -O2 -ffp-contract=off -fopenmp
 * (-ffp-contract=off so `a*b+c` below is a separate multiply and add unless
 * fmaf() is written explicitly).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

/*
 * round_ldl -- vector_balance.py:155-199 (hot loop :179-180), n_greedy_passes=0.
 *   w    [m,d]  grid coordinates (fp32)
 *   L    [d,d]  unit-lower Cholesky factor MINUS identity (vector_balance.py:171-173), row-major
 *   eta  [m,d]  or NULL for 0.5 (vector_balance.py:174-177)
 *   what [m,d]  out: integer-valued fp32 codes
 * Dot product over j = i..d-1 ascending, multiply-then-add in fp32 (the
 * reference's torch matvec order is unspecified; see DESIGN.md "LDLQ parity").
 */
void oracle_round_ldl(const float *w, const float *L, const float *eta, float *what,
                      int64_t m, int64_t d, int nbits)
{
    const float maxq = (float)((1 << nbits) - 1);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < m; ++r) {
        const float *wr = w + r * d;
        float *hr = what + r * d;
        float *err = (float *)malloc(sizeof(float) * (size_t)d);
        memcpy(hr, wr, sizeof(float) * (size_t)d);
        for (int64_t j = 0; j < d; ++j) err[j] = 0.0f;
        for (int64_t i = d - 1; i >= 0; --i) {
            float s = 0.0f;
            for (int64_t j = i; j < d; ++j) s = s + err[j] * L[j * d + i];   /* (w - w_hat)[i:] @ L[i:, i] */
            const float e = eta ? eta[r * d + i] : 0.5f;
            const float x = (wr[i] + s) + e;
            hr[i] = clampf(floorf(x), 0.0f, maxq);
            err[i] = wr[i] - hr[i];
        }
        free(err);
    }
}

/*
 * round_ldl_gptqequiv -- vector_balance.py:381-422: forward-order variant on the
 * flipped factorisation.  Lf is the flipped/normalised factor minus identity as
 * built at :393-397 (the caller builds it; this is only the loop :403-406).
 */
void oracle_round_ldl_forward(const float *w, const float *Lf, const float *eta, float *what,
                              int64_t m, int64_t d, int nbits)
{
    const float maxq = (float)((1 << nbits) - 1);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < m; ++r) {
        const float *wr = w + r * d;
        float *hr = what + r * d;
        float *err = (float *)calloc((size_t)d, sizeof(float));
        memcpy(hr, wr, sizeof(float) * (size_t)d);
        for (int64_t i = 0; i < d; ++i) {
            float s = 0.0f;
            for (int64_t j = 0; j <= i; ++j) s = s + err[j] * Lf[j * d + i];
            const float e = eta ? eta[r * d + i] : 0.5f;
            hr[i] = clampf(floorf((wr[i] + s) + e), 0.0f, maxq);
            err[i] = wr[i] - hr[i];
        }
        free(err);
    }
}

/*
 * Kernel-order LDLQ: the SAME mathematics as round_ldl_block
 * (vector_balance.py:218-257, blocks of `bs` columns taken from the top), with
 * the floating-point evaluation order of quip_amd/csrc/ldlq.hip spelled out so
 * that the HIP kernel can be checked BIT-EXACTLY:
 *
 *   for each column block [i1,i2) descending (i1 = max(i2-bs,0)):
 *     far[c]  = fmaf-chain over the already-rounded columns j >= i2.  The chain
 *               first visits the columns BEYOND the previous block (j >= i2 + bs,
 *               which the kernel accumulates while the previous block is still
 *               being rounded), then the previous block's own columns
 *               i2 <= j < i2 + bs; both in groups of 16; inside a group the
 *               kernel issues four 16x16x4 fp32 MFMAs u=0..3, MFMA u consuming
 *               k = 4*kq+u for kq=0..3 in that order (an fmaf chain, see
 *               cdna_hip_programming.md "FP32-input MFMA ... Numerics").
 *     acc[c]  = far[c]; then for i = i2-1 .. i1 (descending) after column i is
 *               rounded: acc[c] = fmaf(err_i, L[i][c], acc[c]) for c < i.
 *     x       = (w[c] + acc[c]) + eta;  q = clamp(floor(x), 0, maxq);  err_c = w[c] - q.
 *   LT is the TRANSPOSED unit-lower factor: LT[c][j] = L[j][c] (j > c), as the kernel reads it.
 */
void oracle_round_ldl_kernel_order(const float *w, const float *LT, const float *eta,
                                   uint8_t *codes, int64_t m, int64_t d, int nbits, int bs)
{
    const float maxq = (float)((1 << nbits) - 1);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < m; ++r) {
        const float *wr = w + r * d;
        float *err = (float *)calloc((size_t)d, sizeof(float));
        float *acc = (float *)malloc(sizeof(float) * (size_t)bs);
        for (int64_t i2 = d; i2 > 0; i2 -= bs) {
            const int64_t i1 = i2 - bs > 0 ? i2 - bs : 0;
            const int64_t cnt = i2 - i1;
            for (int64_t c = 0; c < cnt; ++c) {
                float f = 0.0f;
                const float *lt = LT + (i1 + c) * d;
                const int64_t iprev = i2 + bs < d ? i2 + bs : d;    /* end of the previous block */
                for (int pass = 0; pass < 2; ++pass) {
                    const int64_t jb = pass == 0 ? iprev : i2, je = pass == 0 ? d : iprev;
                    for (int64_t j0 = jb; j0 < je; j0 += 16)
                        for (int u = 0; u < 4; ++u)
                            for (int kq = 0; kq < 4; ++kq) {
                                const int64_t j = j0 + 4 * kq + u;
                                f = fmaf(err[j], lt[j], f);
                            }
                }
                acc[c] = f;
            }
            for (int64_t i = cnt - 1; i >= 0; --i) {
                const int64_t gi = i1 + i;
                const float e = eta ? eta[r * d + gi] : 0.5f;
                const float x = (wr[gi] + acc[i]) + e;
                const float q = clampf(floorf(x), 0.0f, maxq);
                codes[r * d + gi] = (uint8_t)q;
                const float er = wr[gi] - q;
                err[gi] = er;
                for (int64_t c = 0; c < i; ++c) acc[c] = fmaf(er, LT[(i1 + c) * d + gi], acc[c]);
            }
        }
        free(acc);
        free(err);
    }
}

/*
 * Packed matmul contract of quant_cuda.vecquant{3,4}matmul as recoverable from
 * the pack code (quant.py:186-191,222-233; zeroShot/models/quant.py:187-212):
 *   y[b][r] += sum_k (scales[r]*q[r][k] - zeros[r]) * x[b][k]
 * with q unpacked from the canonical [d/per, m] int32 layout (per = 32/bits,
 * code i at bits [bits*(i%per), +bits) of word i/per), generalised to bits=2.
 * fp32 accumulation in k order.  `zeros` already holds zero*scale (quant.py:186).
 */
void oracle_packed_matmul(const float *x, const int32_t *qweight, float *y, const float *scales,
                          const float *zeros, int64_t bsz, int64_t m, int64_t d, int bits)
{
    const int per = 32 / bits;
    const uint32_t mask = (1u << bits) - 1u;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < m; ++r) {
        for (int64_t b = 0; b < bsz; ++b) {
            float s = 0.0f;
            for (int64_t k = 0; k < d; ++k) {
                const uint32_t word = (uint32_t)qweight[(k / per) * m + r];
                const float q = (float)((word >> (bits * (k % per))) & mask);
                s = s + (scales[r] * q - zeros[r]) * x[b * d + k];
            }
            y[b * m + r] += s;
        }
    }
}
