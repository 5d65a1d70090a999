// fp4layout.hip -- which B slot does each A slot of v_mfma_scale_f32_16x16x128_f8f6f4 (A = fp4, B = fp8) multiply with?
// (scripts/fp4lab.hip's first layout guess -- lane group g holds k = 32 g .. 32 g + 31 on both sides -- did not reproduce a CPU product.)
// A slot = (lane group ga = lane >> 4, nibble ja of the lane's 128 bits); B slot = (gb, byte jb of the lane's 256 bits).  One-hot A
// (row 0, value 1.0) against B whose bytes encode their own slot number in three base-8 digits (values 1 .. 8, exact in e4m3):
// D[0][0] of pass t is digit t of the B slot that slot (ga, ja) meets.
// build: hipcc --offload-arch=gfx950 -O2 scripts/fp4layout.hip -o build_gpu/fp4layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__constant__ unsigned char E4M3_1_TO_8[8] = {0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};

__global__ void probe(int ga, int ja, int pass, float *out)
{
    const int l = threadIdx.x, g = l >> 4;
    i32x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (l == 16 * ga) a[ja / 8] = 0x2 << (4 * (ja % 8));                 // row 0 of lane group ga, nibble ja = 1.0
    for (int jb = 0; jb < 32; ++jb) {
        const int slot = 32 * g + jb, digit = (slot >> (3 * pass)) & 7;
        b[jb / 4] |= (int)E4M3_1_TO_8[digit] << (8 * (jb % 4));
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 0, 0, 127, 0, 127);
    if (l == 0) out[0] = c[0];                                           // D[row 0][col 0]
}

int main()
{
    float *d; hipMalloc(&d, 4);
    int map[128];
    for (int ga = 0; ga < 4; ++ga)
        for (int ja = 0; ja < 32; ++ja) {
            int slot = 0;
            for (int pass = 0; pass < 3; ++pass) {
                probe<<<1, 64>>>(ga, ja, pass, d);
                float v; hipMemcpy(&v, d, 4, hipMemcpyDeviceToHost);
                slot |= ((int)(v + 0.5f) - 1) << (3 * pass);
            }
            map[32 * ga + ja] = slot;
        }
    printf("A slot (lane group, nibble) -> B slot (lane group, byte):\n");
    bool ident = true;
    for (int s = 0; s < 128; ++s) {
        printf("(%d,%2d)->(%d,%2d)%s", s / 32, s % 32, map[s] / 32, map[s] % 32, s % 8 == 7 ? "\n" : "  ");
        ident = ident && map[s] == s;
    }
    printf("identity pairing: %s\n", ident ? "yes" : "NO");
    return 0;
}
