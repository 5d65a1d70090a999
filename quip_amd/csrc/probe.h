// probe.h -- s_memtime phase stamps of the decode launches, IN SITU (round 6: VERDICT r5 next #1a).
// Only in builds with -DQA_PROBE (quip_amd/csrc/libquip_amd_probe.so, `python __graft_entry__.py --probe`); in the shipped library QA_STAMP
// is nothing and tests/test_k2_isa.py keeps auditing that build.  scripts/decode_stamps.py loads the probe library under the ordinary package
// (QUIP_AMD_LIB=...), hands it a device buffer (quipamd_probe_set) and runs the decode engine launch by launch: after every launch the
// buffer holds slot i (0..15) of every wave (0..15) of ONE workgroup of that launch, in shader clocks.
#pragma once
#ifdef QA_PROBE
#include <hip/hip_runtime.h>
void qa_probe_register(void (*setter)(unsigned long long *));          // capi.hip
static __constant__ unsigned long long *qa_probe_ptr = nullptr;        // one per translation unit (no relocatable device code)
static void qa_probe_set_tu(unsigned long long *p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(qa_probe_ptr), &p, sizeof(p)); }
namespace {
struct QaProbeReg {
    QaProbeReg() { qa_probe_register(qa_probe_set_tu); }
} qa_probe_reg_;
}
#ifndef QA_PROBE_WG
#define QA_PROBE_WG 5
#endif
#define QA_STAMP(i)                                                                                                                     \
    do {                                                                                                                                \
        if (blockIdx.x == (QA_PROBE_WG < gridDim.x ? QA_PROBE_WG : gridDim.x - 1) && blockIdx.y == 0 && blockIdx.z == 0 &&             \
            (threadIdx.x & 63) == 0) {                                                                                                  \
            unsigned long long *qa_p_ = qa_probe_ptr;                                                                                   \
            if (qa_p_) qa_p_[(threadIdx.x >> 6) * 16 + (i)] = __builtin_amdgcn_s_memtime();                                            \
        }                                                                                                                               \
    } while (0)
// QA_LOG(0) / QA_LOG(1) at a kernel's first / last statement, both at function scope (round 6): thread 0 of EVERY workgroup records its start and
// end in s_memrealtime ticks (100 MHz, one counter for the whole device) -- no atomics on the way (a log appended through one atomic counter
// made a 384-workgroup launch 15 us longer): buf[256] is a LAUNCH counter that workgroup 0 of an instrumented launch bumps when it ends and every
// workgroup of the next one reads when it starts (the kernel boundary orders the two), and a workgroup's record is slot
// (launch % 1024) * 1024 + workgroup of the table behind buf[258] (two words: start, end).  buf[257] != 0 switches it on.
// scripts/decode_wglog.py replays the engine's graph and splits a launch's period into dispatch skew, body, tail and the true gap.
#define QA_LOG_0                                                                                                                        \
    unsigned long long qa_lt0_ = 0, qa_lseq_ = 0;                                                                                       \
    if (threadIdx.x == 0) {                                                                                                             \
        unsigned long long *qa_p_ = qa_probe_ptr;                                                                                       \
        if (qa_p_ && qa_p_[257]) {                                                                                                      \
            qa_lt0_ = __builtin_amdgcn_s_memrealtime();                                                                                 \
            qa_lseq_ = __hip_atomic_load(qa_p_ + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                      \
        }                                                                                                                               \
    }
#define QA_LOG_1                                                                                                                        \
    if (threadIdx.x == 0 && qa_lt0_) {                                                                                                  \
        unsigned long long *qa_p_ = qa_probe_ptr;                                                                                       \
        const unsigned wg_ = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                                            \
        if (wg_ < 1024u) {                                                                                                              \
            unsigned long long *r_ = qa_p_ + 258 + ((qa_lseq_ & 1023ull) * 1024ull + wg_) * 2ull;                                       \
            r_[0] = qa_lt0_;                                                                                                            \
            r_[1] = __builtin_amdgcn_s_memrealtime();                                                                                   \
        }                                                                                                                               \
        if (wg_ == 0) atomicAdd(qa_p_ + 256, 1ull);                                                                                     \
    }
#define QA_LOG(e) QA_LOG_##e
#else
#define QA_STAMP(i)                                                                                                                     \
    do {                                                                                                                                \
    } while (0)
#define QA_LOG(e)
#endif
