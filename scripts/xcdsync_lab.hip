// xcdsync_lab.hip -- lab: what does one all-gather of G granules between G workgroups cost, per memory scope and per placement?
// gptq_qfnb.hip pays one such exchange per column (3.2-4.0 us per column in total).  Workgroups of one XCD share an L2; if a poll at
// WORKGROUP scope (sc0: no write-through to the fabric) is coherent between them, an exchange confined to one XCD could cost a fraction.
// The kernel checks every round's sum, so a stale read shows up as an error count (or as a bounded-poll abort), not as a hang.
//   hipcc --offload-arch=gfx950 -O3 -o build_gpu/xcdsync_lab scripts/xcdsync_lab.hip && build_gpu/xcdsync_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE, int LSCOPE = SCOPE>
__global__ __launch_bounds__(64) void gather_rounds(unsigned long long *gran, int G, int rounds, int stride, int *errors, int *xcc, long long limit)
{
    if (blockIdx.x % stride) return;
    const int wg = blockIdx.x / stride, lane = threadIdx.x;
    if (lane == 0) xcc[wg] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 15;      // HW_REG_XCC_ID
    int err = 0;
    for (int t = 1; t <= rounds; ++t) {
        unsigned long long *gr = gran + (size_t)(t & 1) * G;
        if (lane == 0) __hip_atomic_store(gr + wg, ((unsigned long long)t << 32) | (unsigned)(wg + t), __ATOMIC_RELAXED, SCOPE);
        unsigned s = 0;
        bool bad = false;
        for (int i = lane; i < G && !bad; i += 64) {
            unsigned long long v;
            long long spins = 0;
            for (;;) {
                v = __hip_atomic_load(gr + i, __ATOMIC_RELAXED, LSCOPE);
                if ((unsigned)(v >> 32) == (unsigned)t) break;
                if (++spins >= limit) { bad = true; break; }
            }
            s += (unsigned)v;
        }
        for (int off = 32; off; off >>= 1) s += __shfl_xor(s, off);
        const unsigned want = (unsigned)(G * (G - 1) / 2 + G * t);
        if (__any(bad)) { err = -1; break; }
        if (s != want) ++err;
    }
    if (lane == 0) errors[wg] = err;
}

template <int SCOPE, int LSCOPE = SCOPE> void run(const char *name, int G, int stride, int rounds)
{
    unsigned long long *gran;
    int *errors, *xcc;
    hipMalloc(&gran, sizeof(unsigned long long) * 2 * G);
    hipMalloc(&errors, sizeof(int) * G);
    hipMalloc(&xcc, sizeof(int) * G);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    std::vector<int> he(G), hx(G);
    int nerr = 0, aborted = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(gran, 0, sizeof(unsigned long long) * 2 * G);
        hipMemset(errors, 0, sizeof(int) * G);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        gather_rounds<SCOPE, LSCOPE><<<G * stride, 64>>>(gran, G, rounds, stride, errors, xcc, 1ll << 20);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
        hipMemcpy(he.data(), errors, sizeof(int) * G, hipMemcpyDeviceToHost);
        hipMemcpy(hx.data(), xcc, sizeof(int) * G, hipMemcpyDeviceToHost);
        for (int i = 0; i < G; ++i) { if (he[i] < 0) ++aborted; else nerr += he[i]; }
    }
    int xmask = 0;
    for (int i = 0; i < G; ++i) xmask |= 1 << hx[i];
    printf("%-10s G=%3d stride=%d  %8.3f us per round   wrong sums %d  aborted workgroups %d  XCC mask 0x%02x\n", name, G, stride,
           best * 1e3f / rounds, nerr, aborted, xmask);
    hipFree(gran); hipFree(errors); hipFree(xcc);
}

int main()
{
    const int rounds = 2000;
    for (int G : {16, 32, 64}) {
        run<__HIP_MEMORY_SCOPE_AGENT>("agent", G, 1, rounds);
        run<__HIP_MEMORY_SCOPE_AGENT>("agent", G, 8, rounds);
        // round 6: PLAIN (workgroup-scope) store -- written through, but the line stays in the XCD's L2 -- polled with sc1 (agent-scope) loads, which
        // bypass the reader's L1 and are served by that L2: same-XCD only
        run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_AGENT>("st wg/ld ag", G, 8, rounds);
        run<__HIP_MEMORY_SCOPE_WORKGROUP, __HIP_MEMORY_SCOPE_AGENT>("st wg/ld ag", G, 1, rounds);   // across XCDs: the control (stale or slow)
        run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup", G, 8, rounds);
        run<__HIP_MEMORY_SCOPE_WORKGROUP>("workgroup", G, 1, rounds);          // across XCDs: expected to fail (stale) or abort -- the control
    }
    return 0;
}
