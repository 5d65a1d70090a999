// fp4layout.hip -- which B slot does each A slot of v_mfma_scale_f32_16x16x128_f8f6f4 (A = fp4, B = fp8) multiply with?
// (scripts/fp4lab.hip's first layout guess -- lane group g holds k = 32 g .. 32 g + 31 on both sides -- did not reproduce a CPU product.)
// A slot = (lane group ga = lane >> 4, nibble ja of the lane's 128 bits); B slot = (gb, byte jb of the lane's 256 bits).
// Full pairing matrix: A one-hot (row 0, value 1.0) against B one-hot (column 0, value 1.0): D[0][0] = 1 iff the two slots hold the same k.
// build: hipcc --offload-arch=gfx950 -O2 scripts/fp4layout.hip -o build_gpu/fp4layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(int sa, float *out)                               // out[128]: D[0][0] for every B slot
{
    const int l = threadIdx.x;
    const int ga = sa >> 5, ja = sa & 31;
    uint32_t aw[4] = {0, 0, 0, 0};
    if (l == 16 * ga) aw[ja >> 3] = 0x2u << (4 * (ja & 7));              // row 0 of lane group ga, nibble ja = 1.0 (E2M1 0b0010)
    i32x8 a = {(int)aw[0], (int)aw[1], (int)aw[2], (int)aw[3], 0, 0, 0, 0};
    for (int sb = 0; sb < 128; ++sb) {
        const int gb = sb >> 5, jb = sb & 31;
        uint32_t bw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (l == 16 * gb) bw[jb >> 2] = 0x38u << (8 * (jb & 3));        // column 0 of lane group gb, byte jb = 1.0 (E4M3 0x38)
        i32x8 b = {(int)bw[0], (int)bw[1], (int)bw[2], (int)bw[3], (int)bw[4], (int)bw[5], (int)bw[6], (int)bw[7]};
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 0, 0, 127, 0, 127);
        for (int r = 0; r < 4; ++r) out[sb * 256 + l * 4 + r] = c[r];      // every D element: where the product lands says which row / column the lanes hold
    }
}

int main()
{
    float *d; hipMalloc(&d, 128 * 256 * 4);
    std::vector<float> row(128 * 256);
    int pair[128];
    bool ident = true;
    for (int sa = 0; sa < 128; ++sa) {
        probe<<<1, 64>>>(sa, d);
        hipMemcpy(row.data(), d, 128 * 256 * 4, hipMemcpyDeviceToHost);
        pair[sa] = -1;
        int hits = 0;
        for (int sb = 0; sb < 128; ++sb)
            for (int e = 0; e < 256; ++e)
                if (row[sb * 256 + e] != 0.f) {
                    pair[sa] = sb; ++hits;
                    const int lane = e >> 2, reg = e & 3, drow = 4 * (lane >> 4) + reg, dcol = lane & 15;
                    if (row[sb * 256 + e] != 1.f || drow != 0 || dcol != 0)
                        printf("A slot (%d,%2d) x B slot (%d,%2d) = %g at D[%d][%d]\n", sa >> 5, sa & 31, sb >> 5, sb & 31, row[sb * 256 + e], drow, dcol);
                }
        if (hits != 1) printf("A slot (%d,%2d): %d matching B slots\n", sa >> 5, sa & 31, hits);
        ident = ident && pair[sa] == sa;
    }
    printf("A slot (lane group, nibble) -> B slot (lane group, byte):\n");
    for (int s = 0; s < 128; ++s) printf("(%d,%2d)->(%d,%2d)%s", s >> 5, s & 31, pair[s] >> 5, pair[s] & 31, s % 8 == 7 ? "\n" : "  ");
    printf("identity pairing: %s\n", ident ? "yes" : "NO");
    return 0;
}
