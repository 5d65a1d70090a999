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
// A load nobody waits for still WRITES its destination register when it lands -- possibly thousands of clocks later.  hipcc does not see an
// asm load's pending write: with a throw-away output it hands the register to the next value and the late write-back corrupts it (round 6:
// the first form of this file did exactly that -- a prefetch workgroup's address registers were overwritten, memory access faults in every
// decode test).  So every touch of a wave writes ONE register the caller keeps alive ("+v") until qa_touch_done() at the end of its work.
typedef uint32_t qa_sink_t;

// NL consecutive 128-byte lines from one address register pair (immediate offsets): pull them towards this CU (its XCD's L2, its L1)
template <int NL> __device__ __forceinline__ void qa_touch_lines(const void *p, qa_sink_t &sink)
{
    static_assert(NL >= 1 && NL <= 4, "1..4 lines");
    asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(p));
    if constexpr (NL > 1) asm volatile("global_load_dword %0, %1, off offset:128" : "+v"(sink) : "v"(p));
    if constexpr (NL > 2) asm volatile("global_load_dword %0, %1, off offset:256" : "+v"(sink) : "v"(p));
    if constexpr (NL > 3) asm volatile("global_load_dword %0, %1, off offset:384" : "+v"(sink) : "v"(p));
}
// the register stays allocated up to here (no instruction is emitted)
__device__ __forceinline__ void qa_touch_done(qa_sink_t &sink) { asm volatile("" ::"v"(sink)); }

// every prefetch workgroup touches EVERY line (its XCD's L2 is private); nobody waits for the loads -- the wave ends behind them (s_endpgm
// waits for outstanding memory operations by itself).  The ranges are dealt to the WAVES (wave w: ranges w, w + nw, ...), a range's lines to the
// lanes: 2-4 scalar round trips for the list entries per wave, then every touch in flight at once.  (First form: every wave walked all ranges,
// one kernarg round trip each -- 33 ranges took a prefetch workgroup ~6 us, longer than the 3 us launch that carried it:
// profiles/r06c_decode_ab.jsonl, OPT-1.3B 1190 -> 1099 tok/s.)
__device__ __forceinline__ void qa_pf_run(const QaPfList &L, unsigned nthreads)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63, nw = (int)(nthreads >> 6);
    qa_sink_t sink = 0;
    for (int r = wave; r < L.n; r += nw) {
        const char *p = reinterpret_cast<const char *>(L.ptr[r]);
        const uint32_t nb = L.bytes[r];
        for (uint32_t off = (uint32_t)lane * 128u; off < nb; off += 64u * 128u) qa_touch_lines<1>(p + off, sink);
        if (lane == 0 && nb >= 4) qa_touch_lines<1>(p + ((nb - 4) & ~3u), sink);            // the last line of a range that starts inside a line
    }
    qa_touch_done(sink);
}
#endif
