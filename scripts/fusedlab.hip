// fusedlab.hip -- lab harness for csrc/decode_fused.hip: times quipamd_decode_fused_gemm on the decode shapes of OPT-1.3B and prints
// the s_memtime phase stamps of wave 0 of one workgroup (built with -DFG_PROBE; scripts/fusedlab.sh).  Synthetic operands (random
// permutations, random codes): the timings do not depend on the values.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#include "../quip_amd/csrc/decode_fused.hip"

int qa_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fprintf(stderr, "\n");
    return code;
}

QaPfList qa_pf_take() { return QaPfList{}; }      // the lab launches carry no operand prefetch

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class T> T *dev_alloc(size_t n, bool randomize = true)
{
    std::vector<unsigned char> h(n * sizeof(T));
    static std::mt19937 rng(1);
    for (auto &b : h) b = randomize ? (unsigned char)(rng() & 0x3f) : 0;          // small patterns: finite as f16 / f32
    T *d;
    CK(hipMalloc(&d, n * sizeof(T)));
    CK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return d;
}
uint16_t *dev_perm(int n)
{
    std::vector<uint16_t> p(n);
    std::iota(p.begin(), p.end(), 0);
    static std::mt19937 rng(7);
    std::shuffle(p.begin(), p.end(), rng);
    uint16_t *d;
    CK(hipMalloc(&d, n * 2));
    CK(hipMemcpy(d, p.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
quipamd_fop make_fop(int p, int q)
{
    quipamd_fop o;
    o.F0 = dev_alloc<uint16_t>((size_t)p * p);
    o.F1 = dev_alloc<uint16_t>((size_t)q * q);
    o.load_idx = dev_perm(p * q);
    o.store_idx = dev_perm(p * q);
    o.p = p;
    o.q = q;
    return o;
}

// regime (round 6): 0 = as rounds 3-5 (ncopies copies of the WEIGHTS only, every other operand shared: the prologue's operands are L2-warm and
// 16-48 copies of 1-6 MB sit in the 256 MiB Infinity Cache); 1 = HBM-cold: enough copies of EVERYTHING a launch reads (weights, factor
// fragments, index vectors, gains, bias, residual) to exceed 320 MiB, as a decode step finds them (508 MB stream through the caches per token);
// 2 = Infinity-Cache-warm: the same with ~48 MiB of copies (what a prefetch of layer L + 1 into the MALL would give); 3 = one copy (L2-warm)
int g_regime = 0;
void run(const char *name, int p, int q, int64_t m, int groups, bool has_u, int norm, int bs, int ncopies, bool res = true, bool pair = false)
{
    const int n = p * q;
    const size_t wbytes = (size_t)groups * m * n / 4;
    if (g_regime == 1) ncopies = (int)((320u << 20) / wbytes) + 1;
    if (g_regime == 2) ncopies = (int)((48u << 20) / wbytes) + 1;
    if (g_regime == 3) ncopies = 1;
    const bool own_ops = g_regime == 1 || g_regime == 2;
    std::vector<quipamd_fused_gemm_args> args(ncopies);
    for (int c = 0; c < ncopies; ++c) {                      // distinct weights per copy: every launch streams cold codes
        quipamd_fused_gemm_args &a = args[c];
        memset(&a, 0, sizeof(a));
        a.act_dtype = QUIPAMD_F16; a.u_y_dtype = QUIPAMD_F16; a.bits = 2; a.has_u = has_u; a.norm = norm; a.ln_eps = 1e-5f; a.ngroups = groups; a.bs = bs; a.m = m; a.y_dtype = QUIPAMD_F16;
        if (c == 0 || own_ops) {
            a.U = make_fop(p, q);
            a.u_y = dev_alloc<uint16_t>((size_t)bs * n); a.u_bias = dev_alloc<uint16_t>(n); a.u_residual = res ? dev_alloc<uint16_t>((size_t)bs * n) : nullptr;
            a.ld_residual = n; a.t_out = pair ? nullptr : dev_alloc<uint16_t>((size_t)bs * n); a.ld_t = n; a.x = dev_alloc<uint16_t>((size_t)bs * n); a.ldx = n;
            a.ln_gamma = dev_alloc<uint16_t>(n); a.ln_beta = dev_alloc<uint16_t>(n);
            if (pair) {                                      // valid LDS offsets: a permutation of the image positions
                std::vector<uint16_t> sg(n);
                std::vector<int> perm(n);
                std::iota(perm.begin(), perm.end(), 0);
                std::shuffle(perm.begin(), perm.end(), std::mt19937(3));
                for (int i = 0; i < n; ++i) sg[i] = (uint16_t)((perm[i] % q) * (p + 8) + perm[i] / q);
                uint16_t *d;
                CK(hipMalloc(&d, n * 2)); CK(hipMemcpy(d, sg.data(), n * 2, hipMemcpyHostToDevice));
                a.pair_sig = d; a.pair_bias = dev_alloc<uint16_t>(n); a.pair_cs = dev_alloc<uint16_t>(n);
            }
            for (int g = 0; g < groups; ++g) {
                a.V[g] = make_fop(p, q);
                a.colscale[g] = dev_alloc<float>(n); a.scale[g] = dev_alloc<float>(1); a.y[g] = dev_alloc<uint16_t>((size_t)bs * m, false);
            }
        } else a = args[0];
        for (int g = 0; g < groups; ++g) a.qweight[g] = dev_alloc<int32_t>((size_t)m * n / 16);
    }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 5; ++i) if (quipamd_decode_fused_gemm(&args[i % ncopies], s)) exit(2);
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int K = 400;
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < K; ++i) quipamd_decode_fused_gemm(&args[i % ncopies], s);
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(exec, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    static const char *rn[4] = {"weights x copies, operands shared (rounds 3-5)", "HBM-cold, every operand its own copy", "Infinity-Cache-warm", "L2-warm (one copy)"};
    printf("%-28s p=%d q=%d m=%lld groups=%d u=%d norm=%d bs=%d : %.3f us per launch (graph of %d, %d copies: %s)\n", name, p, q,
           (long long)m, groups, (int)has_u, norm, bs, ms * 1e3 / K, K, ncopies, rn[g_regime]);
#ifdef FG_PROBE
    CK(hipStreamSynchronize(s));
    quipamd_decode_fused_gemm(&args[1 % ncopies], s);
    CK(hipStreamSynchronize(s));
    unsigned long long st[32];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(fg_probe_buf), sizeof(st)));
    static const char *nm[16] = {"start", "U loads+scatter", "barrier", "U mix", "barrier", "gather / x load", "norm", "V scatter", "barrier",
                                 "V mix", "barrier", "x~ write", "barrier", "weights + MFMA", "park barrier", "reduce + store"};
    printf("    phase stamps of wave 0 (cycles since start; delta):\n");
    unsigned long long prev = st[0];
    for (int i = 1; i < 16; ++i) {
        if (!has_u && i >= 1 && i <= 4) continue;
        printf("      %-18s %7llu  (+%llu)\n", nm[i], st[i] - st[0], st[i] - prev);
        prev = st[i];
    }
    if (pair) {
        unsigned long long ws[16 * 8];
        CK(hipMemcpyFromSymbol(ws, HIP_SYMBOL(fg_wprobe_buf), sizeof(ws)));
        printf("    per-wave stamps (cycles since wave 0's start): start | requests issued | priority barrier passed | before copy | y landed + copied | U frags landed + stored | barrier passed\n");
        for (int w = 0; w < 16; ++w) {
            printf("      wave %2d:", w);
            for (int i = 0; i < 7; ++i) printf(" %7lld", (long long)(ws[w * 8 + i] - ws[0]));
            printf("\n");
        }
    }
#endif
}

int main(int argc, char **argv)
{
    if (argc > 1) g_regime = atoi(argv[1]);
    run("L3 out_proj (V only)", 64, 32, 2048, 1, false, 0, 1, 48);
    run("L1 qkv block 0 (LN, V)", 64, 32, 2048, 3, false, 1, 1, 16);
    run("L1 qkv (U, LN, V)", 64, 32, 2048, 3, true, 1, 1, 16);
    run("L4 fc1 (U, LN, V)", 64, 32, 8192, 1, true, 1, 1, 16);
    run("L6 fc2 (U relu, V)", 128, 64, 2048, 1, true, 0, 1, 16, false);
    run("L6 fc2 pair kernel", 128, 64, 2048, 1, true, 0, 1, 16, false, true);
    run("llama qkv (U, RMS, V)", 64, 64, 4096, 3, true, 2, 1, 8);
    return 0;
}
