// probe_k2.hip -- standalone phase-timestamp probe for K2 (not part of the library).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DQA_PROBE -I include -I quip_amd/csrc scripts/probe_k2.hip -o gpurun_out/probe_k2
#include "../quip_amd/csrc/capi.hip"
#include "../quip_amd/csrc/dqgemm.hip"
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void null_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

int main(int argc, char **argv)
{
    const int64_t m = argc > 1 ? atoll(argv[1]) : 4096, d = argc > 2 ? atoll(argv[2]) : 4096, bs = argc > 3 ? atoll(argv[3]) : 16;
    const int bits = 2;
    const size_t wbytes = m * d * bits / 8;
    const int NRING = 96;
    uint8_t *w; uint16_t *x, *y; float *scale; unsigned long long *probe;
    CK(hipMalloc(&w, wbytes * NRING)); CK(hipMalloc(&x, bs * d * 2)); CK(hipMalloc(&y, bs * m * 2)); CK(hipMalloc(&scale, 4));
    const int nwg_max = 4096 * 16;
    CK(hipMalloc(&probe, nwg_max * 8 * 8));
    std::vector<uint32_t> hw(wbytes / 4);
    for (auto &v : hw) v = rand() * 65537u + rand();
    for (int r = 0; r < NRING; ++r) CK(hipMemcpy(w + r * wbytes, hw.data(), wbytes, hipMemcpyHostToDevice));
    std::vector<uint16_t> hx(bs * d);
    for (auto &v : hx) v = 0x3c00 + (rand() & 0x3ff);
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    float hs = 0.05f; CK(hipMemcpy(scale, &hs, 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    float *yf; CK(hipMalloc(&yf, bs * m * 4)); CK(hipMemset(yf, 0, bs * m * 4));
    int ablate = 0;
    auto run = [&](const char *name, int rt, int bt, int nw, int split, bool acc, bool cold, bool null) {
        quipamd_tune_dequant_gemm(rt, bt, nw, split);
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_ablate), &ablate, sizeof(int)));
        const int steps = 500;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < steps; ++i) {
            if (null) null_kernel<<<256, 1024, 0, st>>>(nullptr);
            else if (quipamd_dequant_gemm(x, 2, (const int32_t *)(w + (cold ? (i % NRING) : 0) * wbytes), bits, 1, 1, scale, nullptr, nullptr,
                                          acc ? (void *)yf : (void *)y, acc ? 0 : 2, acc ? 1 : 0, bs, m, d, st)) { printf("err %s\n", quipamd_last_error()); hipGraph_t gg; hipStreamEndCapture(st, &gg); return; }
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        float best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        printf("abl=%2d %-10s rt=%d bt=%d nw=%2d split=%d %s %s: %.3f us/launch\n", ablate, name, rt, bt, nw, split, acc ? "f32+=" : "bf16 ", cold ? "cold" : "warm", best * 1e3 / steps);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    run("null", 0, 0, 0, 0, false, false, true);
    const bool sweep = argc > 4 && argv[4][0] == 's';
    if (argc > 4 && argv[4][0] == 'a') {
        const int abls[] = {0, 16, 8, 2, 2 | 8, 1, 1 | 2 | 8, 4, 1 | 4, 1 | 2 | 4 | 8};
        for (int cold = 0; cold < 2; ++cold)
            for (int ai = 0; ai < 10; ++ai) {
                ablate = abls[ai];
                run("tile", 1, 0, 16, 0, false, cold, false);
                run("tile", 1, 0, 8, 0, false, cold, false);
                run("tile", 2, 0, 8, 4, true, cold, false);
            }
        return 0;
    }
    for (int cold = 0; cold < 2; ++cold) {
        run("old", 1, 1, 16, 0, false, cold, false);
        // bf16 y (no split): rt x cw x depth
        const int cfg1[][3] = {{1, 16, 1}, {1, 8, 1}, {1, 8, 2}, {1, 4, 2}, {1, 4, 4}, {1, 2, 4}, {2, 8, 1}, {2, 4, 2}, {2, 4, 4}, {2, 2, 4}, {4, 4, 1}, {4, 2, 2}, {4, 1, 4}};
        for (auto &c : cfg1) run("tile", c[0], 0, c[0] * c[1], 100 * c[2], false, cold, false);
        if (!sweep) continue;
        const int rts[3] = {1, 2, 4}, cws[4] = {1, 2, 4, 8}, sps[4] = {2, 4, 8, 16}, dps[3] = {1, 2, 4};
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 4; ++b) for (int c = 0; c < 4; ++c) for (int e = 0; e < 3; ++e) {
            const int rt = rts[a], cw = cws[b], sp = sps[c], dp = dps[e];
            if (rt * cw > 16) continue;
            const long nkc = d / 256;
            if (sp * cw * dp > nkc) continue;                 // slice must need at least this depth
            if ((m / 16 / rt) * sp < 128) continue;
            run("tile", rt, 0, rt * cw, sp + 100 * dp, true, cold, false);
        }
    }
    return 0;
}
