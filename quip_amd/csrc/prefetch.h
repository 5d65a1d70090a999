// prefetch.h -- operands of a LATER decode launch pulled into every XCD's L2 by spare workgroups of an earlier one (round 6).
//
// profiles/r06_decode_stamps.txt: in a real decode step every launch's prologue waits ~1000-2000 clocks for operands that are COLD -- factor
// fragments, index vectors, gains, column scales, biases of THIS layer were last read one token (508 MB of traffic) ago -- while the lab
// harness, whose operands stay in L2, runs the same launch 0.5-0.7 us faster (scripts/fusedlab.sh, regimes 1 vs 3).  Those operands are
// small (50-150 KB per launch), their addresses are known long before, and several launches of a block leave most of the chip idle (the
// attention launch: 32 workgroups on 256 CUs; out_proj: 64).  So a launch may carry QA_PF_WGS extra workgroups -- consecutive block indices,
// i.e. one per XCD -- that do nothing but touch one dword per 128-byte line of a list of ranges and leave: by the time the launch that
// needs them starts, every XCD's L2 holds them.
//
// Host protocol: quipamd_decode_prefetch_next(ptrs, bytes, n) attaches a list to the NEXT quipamd_decode_fused_gemm /
// quipamd_decode_attention_fused launch of the calling thread (consumed by it; inside a hipGraph capture the list is part of the captured
// kernel arguments).  No list: the launch is exactly the round-5 launch.
#pragma once
#include <stdint.h>

constexpr int QA_PF_MAX = 40;      // ranges per list
constexpr int QA_PF_WGS = 8;       // extra workgroups: consecutive linear block indices land on the 8 XCDs

struct QaPfList {
    int n;                          // 0: none
    int first;                      // blockIdx.x of the first prefetch workgroup (= the launch's own grid.x)
    uint32_t bytes[QA_PF_MAX];
    const void *ptr[QA_PF_MAX];
};

QaPfList qa_pf_take();              // capi.hip: the calling thread's pending list (n = 0 when none), cleared

#ifdef __HIPCC__
// every prefetch workgroup touches EVERY line (its XCD's L2 is private); the loads' results are dead and nobody waits for them -- the wave
// ends behind them (s_endpgm waits for outstanding memory operations by itself)
__device__ __forceinline__ void qa_pf_run(const QaPfList &L, unsigned nthreads)
{
    for (int r = 0; r < L.n; ++r) {
        const char *p = reinterpret_cast<const char *>(L.ptr[r]);
        const uint32_t nb = L.bytes[r];
        for (uint32_t off = threadIdx.x * 128u; off < nb; off += nthreads * 128u) {
            uint32_t d;
            asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p + off) : "memory");
        }
        if (threadIdx.x == 0 && nb >= 4) {                             // the last line of a range that does not start on a line boundary
            uint32_t d;
            asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p + ((nb - 4) & ~3u)) : "memory");
        }
    }
}
#endif
