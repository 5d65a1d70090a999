// tilelab.hip -- lab for csrc/ortho_tile.hip: time per launch (hipGraph, ring of distinct operators = cold factors, or one
// operator = hot) and s_memtime stamps per phase.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   -I include -I quip_amd/csrc scripts/tilelab.hip -o build_gpu/tilelab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <numeric>
__device__ unsigned long long *g_tile_probe = nullptr;
#define TILE_STAMP(i) do { if (g_tile_probe && threadIdx.x == 0) g_tile_probe[((size_t)blockIdx.x + gridDim.x * blockIdx.y) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
int qa_fail(int code, const char *fmt, ...) { printf("qa_fail %d: %s\n", code, fmt); return code; }
#include "../quip_amd/csrc/ortho_tile.hip"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class T> T *dev(const std::vector<T> &h) { T *d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }
static uint16_t bf(float f) { uint32_t u; std::memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float frand() { return (float)rand() / RAND_MAX * 2.f - 1.f; }

struct OpSet { quipamd_small_op op; const int32_t *inv; };

OpSet make_op(int p, int q, bool ln, bool perm, bool epi, const void *x, void *out, bool uside)
{
    const int n = p * q;
    std::vector<uint16_t> m0h(p * p), m0l(p * p), m1h(q * q), m1l(q * q), g(n), b(n), res(n);
    for (auto &v : m0h) v = bf(frand() * 0.2f);
    for (auto &v : m0l) v = bf(frand() * 0.001f);
    for (auto &v : m1h) v = bf(frand() * 0.2f);
    for (auto &v : m1l) v = bf(frand() * 0.001f);
    for (auto &v : g) v = 0x3c00; for (auto &v : b) v = 0; for (auto &v : res) v = 0x3800;
    std::vector<float> cs(n, 1.25f), bias(n, 0.5f);
    std::vector<int32_t> idx(n); std::iota(idx.begin(), idx.end(), 0);
    std::random_shuffle(idx.begin(), idx.end());
    std::vector<int32_t> inv(n); for (int i = 0; i < n; ++i) inv[idx[i]] = i;
    OpSet S; quipamd_small_op &o = S.op; std::memset(&o, 0, sizeof o);
    o.M0 = o.M1 = (const float *)1;   // unused by the tile kernel
    o.M0_hi = dev(m0h); o.M0_lo = dev(m0l); o.M1_hi = dev(m1h); o.M1_lo = dev(m1l);
    o.p = p; o.q = q; o.b_first = uside ? 1 : 0; o.colscale = uside ? nullptr : dev(cs);
    if (perm) { o.load_idx = dev(idx); o.store_idx = dev(idx); S.inv = dev(inv); } else S.inv = nullptr;
    if (ln) { o.ln_gamma = dev(g); o.ln_beta = dev(b); o.ln_eps = 1e-5f; o.ln_dtype = QUIPAMD_F16; }
    if (uside) o.bias = dev(bias);
    if (uside && epi) { o.residual = dev(res); o.res_dtype = QUIPAMD_F16; }
    o.x = x; o.x_dtype = uside ? QUIPAMD_F32 : QUIPAMD_F16; o.ldx = n; o.out = out; o.out_dtype = QUIPAMD_BF16; o.ldo = n;
    return S;
}

int main(int argc, char **argv)
{
    const int p = argc > 1 ? atoi(argv[1]) : 64, q = argc > 2 ? atoi(argv[2]) : 32, nops = argc > 3 ? atoi(argv[3]) : 1;
    const int n = p * q, RING = 48, REP = 48;
    std::vector<uint16_t> hx(n); for (auto &v : hx) v = 0x3c00 + (rand() & 0xff);
    void *x = dev(hx); std::vector<float> hxf(n, 0.5f); void *xf = dev(hxf); void *out; CK(hipMalloc(&out, n * 4 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int variant = 0; variant < 4; ++variant) {
        const bool uside = variant & 2, ln = !uside && (variant & 1), full = uside && (variant & 1);
        std::vector<OpSet> ring;
        for (int r = 0; r < RING * nops; ++r) ring.push_back(make_op(p, q, ln, true, full, uside ? xf : x, (char *)out + (r % nops) * n * 4, uside));
        for (int hot = 0; hot < 2; ++hot) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
            for (int r = 0; r < REP; ++r) {
                quipamd_small_op ops[4]; const int32_t *inv[4];
                for (int i = 0; i < nops; ++i) { const OpSet &S = ring[((hot ? 0 : r % RING) * nops) + i]; ops[i] = S.op; inv[i] = S.inv; }
                if (quipamd_ortho_apply_tiles(ops, inv, nops, 1, s)) return 1;
            }
            CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e0, s));
            for (int w = 0; w < 10; ++w) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%dx%d nops=%d %s ln=%d residual=%d %s: %7.3f us per launch\n", p, q, nops, uside ? "U-side" : "V-side", ln, full, hot ? "hot " : "cold", ms * 1e3 / (10 * REP));
        }
        // stamps of one cold launch
        unsigned long long *buf; const int nwg = (p / 16) * (q / 16) * nops;
        CK(hipMalloc(&buf, nwg * 64)); CK(hipMemset(buf, 0, nwg * 64));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_probe), &buf, sizeof buf));
        quipamd_small_op ops[4]; const int32_t *inv[4];
        for (int i = 0; i < nops; ++i) { ops[i] = ring[7 * nops + i].op; inv[i] = ring[7 * nops + i].inv; }
        quipamd_ortho_apply_tiles(ops, inv, nops, 1, s); CK(hipStreamSynchronize(s));
        unsigned long long *nul = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_probe), &nul, sizeof nul));
        std::vector<unsigned long long> h(nwg * 8); CK(hipMemcpy(h.data(), buf, nwg * 64, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull; for (int w = 0; w < nwg; ++w) t0 = std::min(t0, h[w * 8]);
        (void)t0;
        printf("   wg 0 phase ticks: ");
        for (int i = 1; i < 7; ++i) printf(" [%d-%d] %llu", i - 1, i, h[i] - h[i - 1]);
        printf("  total %llu\n", h[6] - h[0]);
    }
    return 0;
}
