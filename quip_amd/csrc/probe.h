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
#else
#define QA_STAMP(i)                                                                                                                     \
    do {                                                                                                                                \
    } while (0)
#endif
