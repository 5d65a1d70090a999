// pdl_lab.hip -- lab (round 6): can a decode step's dependent launches overlap their BOUNDARIES?  HIP has no programmatic dependent launch; this
// emulates one: consecutive kernels of a chain alternate between two streams of one captured graph (two branches, no graph edge between
// neighbours), and kernel i waits IN THE KERNEL for a device counter that kernel i - 1 bumps when its workgroups are done.  So kernel i is
// dispatched -- and can fetch what does not depend on its predecessor -- while kernel i - 1 still runs; the price of the dependency is one
// hand-off through memory instead of the 2.05-2.23 us gap scripts/decode_wglog.py measures between dependent launches of one stream.
// Each kernel: G workgroups, ~`work` us of timed spinning as its body, thread 0 of each workgroup bumps done[i] (agent-scope atomic) at the end;
// workgroups of kernel i poll done[i - 1] (sc1 load by one lane, s_sleep between polls) until it reaches G_{i-1} x the replay's number.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pdl_lab scripts/pdl_lab.hip && /tmp/pdl_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void link_kernel(unsigned *done, int i, unsigned prev_target, int wait, int work_ticks, long long limit, int *err)
{
    if (wait && i > 0) {
        if (threadIdx.x == 0) {
            long long spins = 0;
            while (__hip_atomic_load(done + (i - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < prev_target) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > limit) { atomicExch(err, 1); break; }
            }
        }
        __syncthreads();
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < work_ticks) { }
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(done + i, 1u);
}

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int K = 40, reps = 10;
    unsigned *done;
    int *err;
    hipMalloc(&done, sizeof(unsigned) * K);
    hipMalloc(&err, 4);
    hipStream_t sa, sb;
    hipStreamCreate(&sa);
    hipStreamCreate(&sb);
    hipEvent_t fork, join, e0, e1;
    hipEventCreate(&fork); hipEventCreate(&join); hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {32, 64, 128}) for (float work_us : {1.0f, 3.0f}) for (int mode = 0; mode < 2; ++mode) {
        // mode 0: one stream, dependent launches (today's engine); mode 1: two alternating branches + in-kernel waits
        hipGraph_t graph;
        hipGraphExec_t exec;
        hipMemset(done, 0, sizeof(unsigned) * K);
        hipMemset(err, 0, 4);
        hipDeviceSynchronize();
        // capture ONE replay; targets are per replay (counters keep growing): pass the replay number through a captured memcpy?  Simpler: the
        // counters are reset by a memset node at the head of the graph
        hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal);
        hipMemsetAsync(done, 0, sizeof(unsigned) * K, sa);
        if (mode == 1) { hipEventRecord(fork, sa); hipStreamWaitEvent(sb, fork, 0); }
        for (int i = 0; i < K; ++i) {
            hipStream_t s = (mode == 1 && (i & 1)) ? sb : sa;
            link_kernel<<<G, 256, 0, s>>>(done, i, (unsigned)G, mode, (int)(work_us * 100.f), 1ll << 12, err);
        }
        if (mode == 1) { hipEventRecord(join, sb); hipStreamWaitEvent(sa, join, 0); }
        hipStreamEndCapture(sa, &graph);
        if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
        for (int w = 0; w < 1; ++w) hipGraphLaunch(exec, sa);
        hipStreamSynchronize(sa);
        hipEventRecord(e0, sa);
        for (int r = 0; r < reps; ++r) hipGraphLaunch(exec, sa);
        hipEventRecord(e1, sa);
        hipStreamSynchronize(sa);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        int herr = 0;
        hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
        printf("G=%3d  body %.1f us  %-44s %7.2f us per kernel  (%.2f beyond the body)%s\n", G, work_us,
               mode ? "two branches + in-kernel wait on a counter" : "one stream, dependent launches", ms * 1e3f / reps / K, ms * 1e3f / reps / K - work_us,
               herr ? "  POLL TIMEOUT" : "");
        hipGraphExecDestroy(exec);
        hipGraphDestroy(graph);
    }
    return 0;
}
