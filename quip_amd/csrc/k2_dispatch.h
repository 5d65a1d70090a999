// k2_dispatch.h -- internal interface between the C ABI (dqgemm.hip) and the second-generation K2 kernels (dqgemm_v2.hip)
#pragma once
#include <stdint.h>

enum { K2_FAM_AUTO = 0, K2_FAM_OLD = 1, K2_FAM_H = 2, K2_FAM_S = 3, K2_FAM_MB = 4, K2_FAM_PF = 5, K2_FAM_NONE = 99 };
enum { K2V2_NOT_TAKEN = -1 };

struct K2Call {
    const void *x; int x_dtype;           // [bs, d] bf16 / fp16
    const void *qweight; int bits;         // STREAM layout; container bits (2 | 4)
    int qfn, maxq;                         // grid; maxq = 2^wbits - 1 (7 for 3-bit codes in the 4-bit container)
    const float *scale, *zero, *bias;
    void *y; int y_dtype, accumulate;
    int64_t bs, m, d;
    int cfg[4];                            // quipamd_k2_config: {family, p1, p2, p3}; all 0 = heuristic
};

// returns K2V2_NOT_TAKEN (use the round-1 kernels), QUIPAMD_OK, or an error status
int k2v2_launch(const K2Call &c, void *stream);

// ngroups (2..3) problems of ONE shape as one launch of the grouped h kernel (grid.y), fp16 activations (the decode engine's 5..16-row
// steps): calls[i] differ in x, qweight, scale, y only.  Returns K2V2_NOT_TAKEN when the h kernel does not hold the shape.
int k2v2_launch_grouped(const K2Call *calls, int ngroups, void *stream);
