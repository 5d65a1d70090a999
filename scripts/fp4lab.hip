// fp4lab.hip -- sizing of VERDICT r3 next #3: "make the 2-bit conversion disappear" with gfx950's block-scaled K = 128 MFMA.
//   A = the 2-bit code c as an FP4 (E2M1) nibble 0b00cc = c / 2 exactly; B = x as two e4m3 pieces (hi + lo); fp32 accumulate.
// What it measures (register-only loops, no memory traffic, 256 workgroups):
//   1. the operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with A = fp4, B = fp8: checked against a CPU product (so that a kernel built on
//      it would be right), incl. the E8M0 scale operands;
//   2. instruction rate: fp4 x fp8, fp8 x fp8, fp4 x fp4 at K = 128 next to bf16 16x16x32 (cycles per instruction per SIMD);
//   3. the compute side of one STREAM tile (16 rows x 256 columns, 64 lanes x 16 bytes of codes) both ways:
//        bf16 path   8 x [DeqT frag (4 v_bfi) + v_mfma_f32_16x16x32_bf16]                          (what K2's kernels do today)
//        ME path     8 x [multi-exponent frag] + 8 MFMA                                           (the S kernel's form: 10 VALU per dword)
//        fp4 path    nibble spread (3 VALU per packed dword) + 2 k-steps x NP pieces of v_mfma_scale K = 128     (NP = 2, 3)
//      per wave with 1, 2, 4 waves per SIMD -- the VALU / matrix-pipe balance at batch <= 16.
// build: hipcc --offload-arch=gfx950 -O3 -I include -I quip_amd/csrc scripts/fp4lab.hip -o build_gpu/fp4lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// formats of the f8f6f4 instruction (cbsz = A, blgp = B): 0 fp8 e4m3, 1 bf8 e5m2, 2 fp6 e2m3, 3 bf6 e3m2, 4 fp4 e2m1
#define FMT_FP8 0
#define FMT_FP4 4

__device__ __forceinline__ uint32_t opaque(uint32_t v) { asm("" : "+v"(v)); return v; }

// ---- 1. layout check -------------------------------------------------------------------------------------------------------------
// layout (measured slot by slot with scripts/fp4layout.hip, profiles/r04e_fp4layout.txt -- the two formats differ):
//   fp4 operand: lane l holds row (l & 15), nibble j of its 128 bits (low nibble first) = k 32 (l >> 4) + j          -- 32 consecutive k
//   fp8 operand: lane l holds column (l & 15), byte j of its 256 bits = k 16 (l >> 4) + j for j < 16, 64 + 16 (l >> 4) + (j - 16) above
//                (two K = 64 halves of 16 bytes each, like two v_mfma 16x16x64 operands back to back)
// scale operand: byte `opsel` of the VGPR, E8M0 (127 = 1.0).
__global__ void layout_kernel(const uint32_t *A4, const uint32_t *B8, float *D, int sa, int sb)
{
    const int l = threadIdx.x;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b;
    for (int i = 0; i < 4; ++i) a[i] = (int)A4[l * 4 + i];
    for (int i = 0; i < 8; ++i) b[i] = (int)B8[l * 8 + i];
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FMT_FP4, FMT_FP8, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];       // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
}

// ---- 2. rates --------------------------------------------------------------------------------------------------------------------
template <int FA, int FB> __global__ void rate_kernel(float *o, int iters)
{
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x22222222 + threadIdx.x + i; b[i] = 0x38383838 + 3 * threadIdx.x + i; }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], FA, FB, 0, 127, 0, 127);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void rate_bf16_kernel(float *o, int iters)
{
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x + 2 * i)); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    o[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- 3. one STREAM tile's compute, three ways ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t shifted, uint32_t base) { return (shifted & mask) | (base & ~mask); }

// MODE 0: uniform-offset bf16 (DeqT<2, ActBF16>), 1: multi-exponent bf16 (DeqME2), 2: fp4 x 2 fp8 pieces, 3: fp4 x 3 fp8 pieces
template <int MODE> __global__ void tile_kernel(float *o, const uint32_t *seed, int tiles)
{
    // "weights": a register quadruple that changes every tile (xorshift on the lane's own words -- 4 VALU per tile in every mode, so
    // that the compiler cannot hoist the conversion out of the loop); "x": fixed fragments, as from LDS
    u32x4 w = {seed[threadIdx.x], seed[threadIdx.x + 64], seed[threadIdx.x + 128], seed[threadIdx.x + 192]};
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const uint32_t base = opaque(0x40804080u), b0 = opaque(0x43004300u), b1 = opaque(0x42004200u), b2 = opaque(0x41004100u);
    bf16x8 xb[8];
    i32x8 x8[2][3];
    for (int t = 0; t < 8; ++t)
        for (int i = 0; i < 8; ++i) xb[t][i] = (__bf16)(0.01f * ((threadIdx.x * 7 + t * 3 + i) % 13));
    for (int s = 0; s < 2; ++s)
        for (int p = 0; p < 3; ++p)
            for (int i = 0; i < 8; ++i) x8[s][p][i] = 0x38303438 + threadIdx.x * (s + 1) + 17 * p + i;
    for (int it = 0; it < tiles; ++it) {
        w[0] ^= w[0] << 13; w[1] ^= w[1] >> 7; w[2] ^= w[2] << 5; w[3] ^= w[3] >> 11;
        if constexpr (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t src = w[t >> 1];
                u32x4 a;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int sh = 2 * (4 * (t & 1) + v) - 5;
                    a[v] = bfi(0x00600060u, sh >= 0 ? src >> sh : src << -sh, base);
                }
                if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), xb[t], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), xb[t], acc0, 0, 0, 0);
            }
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t src = w[t >> 1];
                u32x4 a;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int i = 4 * (t & 1) + v, sh = i < 3 ? 0 : i < 6 ? 6 : 12, p = 2 * (i < 3 ? i : i < 6 ? i - 3 : i - 6);
                    a[v] = bfi((3u << p) * 0x10001u, sh ? src >> sh : src, p == 0 ? b0 : p == 2 ? b1 : b2);
                }
                if (t & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), xb[t], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), xb[t], acc0, 0, 0, 0);
            }
        } else {
            constexpr int NP = MODE == 2 ? 2 : 3;
            // 2-bit -> nibble: with the tile's codes laid out so that dword j holds, at bits 4 n + 2 h .. + 1, the code of nibble n of the
            // A operand's dword (2 j + h):   lo = w & 0x33333333,  hi = (w >> 2) & 0x33333333       (3 VALU per packed dword)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0};
                a[0] = (int)(w[2 * s] & 0x33333333u);
                a[1] = (int)((w[2 * s] >> 2) & 0x33333333u);
                a[2] = (int)(w[2 * s + 1] & 0x33333333u);
                a[3] = (int)((w[2 * s + 1] >> 2) & 0x33333333u);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    if ((s * NP + p) & 1) acc1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, x8[s][p], acc1, FMT_FP4, FMT_FP8, 0, 127, 0, 127 - 4 * p);
                    else acc0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, x8[s][p], acc0, FMT_FP4, FMT_FP8, 0, 127, 0, 127 - 4 * p);
                }
            }
        }
    }
    o[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
}

static float e2m1(int n) { static const float v[8] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f}; return (n & 8) ? -v[n & 7] : v[n & 7]; }
static float e4m3(int b)
{
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? std::ldexp((float)m, -9) : std::ldexp(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}

int main()
{
    float *o; CK(hipMalloc(&o, 4 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // 1. layout
    {
        std::vector<uint32_t> A(64 * 4), B(64 * 8);
        std::vector<float> Af(16 * 128), Bf(128 * 16);
        uint32_t rng = 12345;
        auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
        for (int l = 0; l < 64; ++l) {
            for (int j = 0; j < 32; ++j) {
                const int code = next() & 3;                                   // nibble 0b00cc
                A[l * 4 + j / 8] |= (uint32_t)code << (4 * (j % 8));
                Af[(l & 15) * 128 + 32 * (l >> 4) + j] = e2m1(code);
                int byte = next() & 0xff;
                if ((byte & 0x7f) == 0x7f) byte &= 0xf7;                       // no NaN
                B[l * 8 + j / 4] |= (uint32_t)byte << (8 * (j % 4));
                const int kb = j < 16 ? 16 * (l >> 4) + j : 64 + 16 * (l >> 4) + (j - 16);   // fp8: two K = 64 halves of 16 bytes per lane group (fp4layout.hip)
                Bf[kb * 16 + (l & 15)] = e4m3(byte);
            }
        }
        uint32_t *dA, *dB; float *dD;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 256 * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
        for (int sb : {127, 123}) {
            layout_kernel<<<1, 64>>>(dA, dB, dD, 127, sb); CK(hipDeviceSynchronize());
            std::vector<float> D(256);
            CK(hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost));
            double worst = 0, norm = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double r = 0;
                    for (int k = 0; k < 128; ++k) r += (double)Af[i * 128 + k] * Bf[k * 16 + j];
                    r = std::ldexp(r, sb - 127);
                    worst = std::fmax(worst, std::fabs(r - D[i * 16 + j]));
                    norm = std::fmax(norm, std::fabs(r));
                }
            printf("layout check (A fp4: k = 32 (lane >> 4) + nibble; B fp8: k = 16 (lane >> 4) + byte | 64 + ..; scale_b E8M0 %d): max |err| %.3g of max |ref| %.3g -> %s\n", sb, worst, norm,
                   worst <= 1e-3 * norm ? "MATCH" : "MISMATCH");
        }
    }
    // 2. rates
    auto time_it = [&](auto launch) -> float { launch(100); hipDeviceSynchronize(); hipEventRecord(e0); launch(4000); hipEventRecord(e1); hipDeviceSynchronize(); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    for (int wpc : {4, 8}) {
        const int th = 64 * wpc;
        const double n = 4000.0 * 8 * wpc / 4;                                  // instructions per SIMD
        float ms = time_it([&](int it) { rate_bf16_kernel<<<256, th>>>(o, it); });
        printf("%d waves/CU  bf16 16x16x32      : %6.2f cycles per MFMA per SIMD  (%7.1f TFLOP/s)\n", wpc, ms * 1e-3 * 2.4e9 / n, 256.0 * wpc * 4000 * 8 * 2 * 16 * 16 * 32 / ms / 1e9);
        ms = time_it([&](int it) { rate_kernel<FMT_FP4, FMT_FP8><<<256, th>>>(o, it); });
        printf("%d waves/CU  fp4 x fp8 16x16x128: %6.2f cycles per MFMA per SIMD  (%7.1f TFLOP/s)\n", wpc, ms * 1e-3 * 2.4e9 / n, 256.0 * wpc * 4000 * 8 * 2 * 16 * 16 * 128 / ms / 1e9);
        ms = time_it([&](int it) { rate_kernel<FMT_FP8, FMT_FP8><<<256, th>>>(o, it); });
        printf("%d waves/CU  fp8 x fp8 16x16x128: %6.2f cycles per MFMA per SIMD  (%7.1f TFLOP/s)\n", wpc, ms * 1e-3 * 2.4e9 / n, 256.0 * wpc * 4000 * 8 * 2 * 16 * 16 * 128 / ms / 1e9);
        ms = time_it([&](int it) { rate_kernel<FMT_FP4, FMT_FP4><<<256, th>>>(o, it); });
        printf("%d waves/CU  fp4 x fp4 16x16x128: %6.2f cycles per MFMA per SIMD  (%7.1f TFLOP/s)\n", wpc, ms * 1e-3 * 2.4e9 / n, 256.0 * wpc * 4000 * 8 * 2 * 16 * 16 * 128 / ms / 1e9);
    }
    // 3. one STREAM tile
    uint32_t *seed; CK(hipMalloc(&seed, 256 * 4));
    { std::vector<uint32_t> s(256); for (int i = 0; i < 256; ++i) s[i] = 0x9e3779b9u * (i + 1); CK(hipMemcpy(seed, s.data(), 1024, hipMemcpyHostToDevice)); }
    const char *names[4] = {"bf16 uniform offset (16 VALU / dword)", "bf16 multi-exponent (10 VALU / dword)", "fp4 x 2 fp8 pieces (3 VALU / dword)  ", "fp4 x 3 fp8 pieces (3 VALU / dword)  "};
    for (int wps : {1, 2, 4}) {
        const int th = 256 * wps;
        float ms[4];
        ms[0] = time_it([&](int it) { tile_kernel<0><<<256, th>>>(o, seed, it); });
        ms[1] = time_it([&](int it) { tile_kernel<1><<<256, th>>>(o, seed, it); });
        ms[2] = time_it([&](int it) { tile_kernel<2><<<256, th>>>(o, seed, it); });
        ms[3] = time_it([&](int it) { tile_kernel<3><<<256, th>>>(o, seed, it); });
        for (int m = 0; m < 4; ++m)
            printf("%d wave(s) per SIMD  %s: %7.1f cycles per 1 KiB tile per wave, %6.1f per tile per SIMD -> %5.2f TB/s of codes on 1024 SIMDs\n", wps, names[m],
                   ms[m] * 1e-3 * 2.4e9 / 4000, ms[m] * 1e-3 * 2.4e9 / 4000 / wps, 1024.0 * 1024 * 2.4e9 / (ms[m] * 1e-3 * 2.4e9 / 4000 / wps) / 1e12);
    }
    return 0;
}
