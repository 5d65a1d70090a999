// pack.hip -- K1: int2 / int4 pack and unpack (bit-exact integer work, HBM-bound).
//
// CANONICAL layout = the reference rule of zeroShot/models/quant.py:190-199 generalised to 2 bits:
//   qweight[i / per][r] |= code[r][i] << (bits * (i % per)),  per = 32 / bits,  int32 [d/per, m].
// STREAM layout = the MFMA A-fragment order streamed by dqgemm.hip (oracle pack_stream is the spec):
//   [row_tile = r/16][k_chunk = k/KC][lane = 16*g + (r%16)][dword u], KC = 512/bits.
//
// Roofline: bytes = m*d (uint8 codes) + m*d*bits/8 (words); no arithmetic to speak of.
#include "common.h"

namespace {

constexpr int TILE = 32;   // canonical kernels: 32 rows x 32 words per block, transposed through LDS

template <int BITS>
__global__ __launch_bounds__(256) void pack_canonical_kernel(const uint8_t *__restrict__ codes,
                                                             uint32_t *__restrict__ out, int64_t m, int64_t d)
{
    constexpr int PER = 32 / BITS;
    __shared__ uint32_t tile[TILE][TILE + 1];
    const int64_t nwords = d / PER;
    const int64_t r0 = (int64_t)blockIdx.y * TILE, w0 = (int64_t)blockIdx.x * TILE;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    // phase 1: lanes walk consecutive words of one row -> contiguous PER-byte reads
    for (int rr = ty; rr < TILE; rr += 8) {
        const int64_t r = r0 + rr, wi = w0 + tx;
        uint32_t word = 0;
        if (r < m && wi < nwords) {
            const uint8_t *src = codes + r * d + wi * PER;
            if constexpr (PER == 16) {
                const uint4 v = *reinterpret_cast<const uint4 *>(src);
                const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 16; ++e) word |= ((q[e >> 2] >> (8 * (e & 3))) & 3u) << (2 * e);
            } else {
                const uint2 v = *reinterpret_cast<const uint2 *>(src);
                const uint32_t q[2] = {v.x, v.y};
#pragma unroll
                for (int e = 0; e < 8; ++e) word |= ((q[e >> 2] >> (8 * (e & 3))) & 15u) << (4 * e);
            }
        }
        tile[rr][tx] = word;
    }
    __syncthreads();
    // phase 2: lanes walk consecutive rows of one word index -> contiguous int32 writes
    for (int ww = ty; ww < TILE; ww += 8) {
        const int64_t r = r0 + tx, wi = w0 + ww;
        if (r < m && wi < nwords) out[wi * m + r] = tile[tx][ww];
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void unpack_canonical_kernel(const uint32_t *__restrict__ packed,
                                                               uint8_t *__restrict__ codes, int64_t m, int64_t d)
{
    constexpr int PER = 32 / BITS;
    __shared__ uint32_t tile[TILE][TILE + 1];
    const int64_t nwords = d / PER;
    const int64_t r0 = (int64_t)blockIdx.y * TILE, w0 = (int64_t)blockIdx.x * TILE;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int ww = ty; ww < TILE; ww += 8) {
        const int64_t r = r0 + tx, wi = w0 + ww;
        tile[tx][ww] = (r < m && wi < nwords) ? packed[wi * m + r] : 0u;
    }
    __syncthreads();
    for (int rr = ty; rr < TILE; rr += 8) {
        const int64_t r = r0 + rr, wi = w0 + tx;
        if (r < m && wi < nwords) {
            const uint32_t word = tile[rr][tx];   // mask AFTER the shift: bit 31 may be set (negative int32)
            uint8_t *dst = codes + r * d + wi * PER;
            if constexpr (PER == 16) {
                uint32_t q[4] = {0, 0, 0, 0};
#pragma unroll
                for (int e = 0; e < 16; ++e) q[e >> 2] |= ((word >> (2 * e)) & 3u) << (8 * (e & 3));
                *reinterpret_cast<uint4 *>(dst) = make_uint4(q[0], q[1], q[2], q[3]);
            } else {
                uint32_t q[2] = {0, 0};
#pragma unroll
                for (int e = 0; e < 8; ++e) q[e >> 2] |= ((word >> (4 * e)) & 15u) << (8 * (e & 3));
                *reinterpret_cast<uint2 *>(dst) = make_uint2(q[0], q[1]);
            }
        }
    }
}

// field position of element e (0..7) of MFMA step t inside the lane's 4 dwords
template <int BITS> __device__ __forceinline__ void stream_pos(int t, int e, int &u, int &sh)
{
    if constexpr (BITS == 2) {
        u = t >> 1;
        sh = 2 * (4 * (t & 1) + (e >> 1)) + ((e & 1) ? 16 : 0);
    } else {
        u = t;
        sh = 4 * (e >> 1) + ((e & 1) ? 16 : 0);
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void pack_stream_kernel(const uint8_t *__restrict__ codes,
                                                          uint4 *__restrict__ out, int64_t m, int64_t d)
{
    constexpr int KC = 512 / BITS, NT = KC / 32;
    const int64_t nkc = d / KC;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // = (rt*nkc + kc)*64 + lane
    if (gid >= (m / 16) * nkc * 64) return;
    const int lane = (int)(gid & 63);
    const int64_t tile = gid >> 6, kc = tile % nkc, rt = tile / nkc;
    const int j = lane & 15, g = lane >> 4;
    const uint8_t *row = codes + (rt * 16 + j) * d + kc * KC + 8 * g;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint2 v = *reinterpret_cast<const uint2 *>(row + 32 * t);
        const uint32_t q[2] = {v.x, v.y};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int u, sh;
            stream_pos<BITS>(t, e, u, sh);
            w[u] |= ((q[e >> 2] >> (8 * (e & 3))) & ((1u << BITS) - 1u)) << sh;
        }
    }
    out[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

template <int BITS>
__global__ __launch_bounds__(256) void unpack_stream_kernel(const uint4 *__restrict__ packed,
                                                            uint8_t *__restrict__ codes, int64_t m, int64_t d)
{
    constexpr int KC = 512 / BITS, NT = KC / 32;
    const int64_t nkc = d / KC;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (m / 16) * nkc * 64) return;
    const int lane = (int)(gid & 63);
    const int64_t tile = gid >> 6, kc = tile % nkc, rt = tile / nkc;
    const int j = lane & 15, g = lane >> 4;
    uint8_t *row = codes + (rt * 16 + j) * d + kc * KC + 8 * g;
    const uint4 pv = packed[gid];
    const uint32_t w[4] = {pv.x, pv.y, pv.z, pv.w};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        uint32_t q[2] = {0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int u, sh;
            stream_pos<BITS>(t, e, u, sh);
            q[e >> 2] |= ((w[u] >> sh) & ((1u << BITS) - 1u)) << (8 * (e & 3));
        }
        *reinterpret_cast<uint2 *>(row + 32 * t) = make_uint2(q[0], q[1]);
    }
}

// ---- the reference's 3-bit rule (quant.py:192-220, Quant3Linear.pack): within each run of 32 input columns, code j sits at
// bits [3j, 3j+3) of a 96-bit little-endian group stored as 3 consecutive int32 rows of qweight [d/32*3, m] ------------------
__global__ __launch_bounds__(256) void pack3_canonical_kernel(const uint8_t *__restrict__ codes, uint32_t *__restrict__ out,
                                                              int64_t m, int64_t d)
{
    // thread = (group, row); lanes walk rows: the three words of a group are written row-contiguous
    const int64_t ngroups = d / 32;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= ngroups * m) return;
    const int64_t r = gid % m, grp = gid / m;
    const uint8_t *src = codes + r * d + grp * 32;
    uint64_t lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const uint64_t c = src[j] & 7u;
        const int pos = 3 * j;
        if (pos + 3 <= 64) lo |= c << pos;
        else if (pos >= 64) hi |= (uint32_t)c << (pos - 64);
        else { lo |= c << pos; hi |= (uint32_t)(c >> (64 - pos)); }          // j = 21 straddles bit 64
    }
    out[(grp * 3 + 0) * m + r] = (uint32_t)lo;
    out[(grp * 3 + 1) * m + r] = (uint32_t)(lo >> 32);
    out[(grp * 3 + 2) * m + r] = hi;
}

__device__ __forceinline__ uint32_t bits96(uint32_t w0, uint32_t w1, uint32_t w2, int pos, int n)
{   // n <= 24 bits starting at bit pos of the 96-bit little-endian number {w2, w1, w0}
    const uint64_t lo = ((uint64_t)w1 << 32) | w0, hi = ((uint64_t)w2 << 32) | w1;
    const uint64_t v = pos < 32 ? (lo >> pos) : (hi >> (pos - 32));
    return (uint32_t)(v & ((1ull << n) - 1ull));
}

__global__ __launch_bounds__(256) void unpack3_canonical_kernel(const uint32_t *__restrict__ packed, uint8_t *__restrict__ codes,
                                                                int64_t m, int64_t d)
{
    const int64_t ngroups = d / 32;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= ngroups * m) return;
    const int64_t r = gid % m, grp = gid / m;
    const uint32_t w0 = packed[(grp * 3 + 0) * m + r], w1 = packed[(grp * 3 + 1) * m + r], w2 = packed[(grp * 3 + 2) * m + r];
    uint8_t *dst = codes + r * d + grp * 32;
#pragma unroll
    for (int j = 0; j < 32; ++j) dst[j] = (uint8_t)bits96(w0, w1, w2, 3 * j, 3);
}

// ---- CANONICAL (the reference's checkpoint format) -> STREAM on the device, no host pass -----------------------------------
// One thread per STREAM lane (row tile, chunk, lane = 16 g + j): its 8 consecutive columns of MFMA step t are
//   4 bit: exactly canonical word [(k0 + 32 t + 8 g) / 8][r]                 (zeroShot/models/quant.py:190-199)
//   2 bit: half of canonical word [(k0 + 32 t + 8 g) / 16][r]
//   3 bit: bits [24 g, 24 g + 24) of the 96-bit group (k0 + 32 t) / 32        (quant.py:192-220) -> 4-bit STREAM container
// Lanes j are consecutive rows r: the canonical reads are row-contiguous.
template <int BITS>   // BITS of the canonical source; the container is 4 for BITS = 3
__global__ __launch_bounds__(256) void repack_stream_kernel(const uint32_t *__restrict__ canon, uint4 *__restrict__ out,
                                                            int64_t m, int64_t d)
{
    constexpr int CB = BITS == 3 ? 4 : BITS, KC = 512 / CB, NT = KC / 32;
    const int64_t nkc = d / KC;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (m / 16) * nkc * 64) return;
    const int lane = (int)(gid & 63);
    const int64_t tile = gid >> 6, kc = tile % nkc, rt = tile / nkc;
    const int j = lane & 15, g = lane >> 4;
    const int64_t r = rt * 16 + j, k0 = kc * KC;
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int64_t k = k0 + 32 * t + 8 * g;
        uint32_t c8;                                                   // 8 codes, code e at bits [BITS*e, +BITS)
        if constexpr (BITS == 4) c8 = canon[(k / 8) * m + r];
        else if constexpr (BITS == 2) c8 = (canon[(k / 16) * m + r] >> (16 * ((k / 8) & 1))) & 0xffffu;
        else {
            const int64_t grp = k / 32;
            c8 = bits96(canon[(grp * 3) * m + r], canon[(grp * 3 + 1) * m + r], canon[(grp * 3 + 2) * m + r], 24 * g, 24);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int u, sh;
            stream_pos<CB>(t, e, u, sh);
            w[u] |= ((c8 >> (BITS * e)) & ((1u << BITS) - 1u)) << sh;
        }
    }
    out[gid] = make_uint4(w[0], w[1], w[2], w[3]);
}

int check_pack_args(const void *a, const void *b, int bits, int layout, int64_t m, int64_t d)
{
    QA_REQUIRE((a && b) || m == 0 || d == 0, QUIPAMD_ERR_ARG, "pack/unpack: null pointer");
    QA_REQUIRE(bits == 2 || bits == 4 || (bits == 3 && layout == QUIPAMD_LAYOUT_CANONICAL), QUIPAMD_ERR_UNSUPPORTED,
               "pack/unpack: container bits must be 2 or 4, or 3 in the canonical layout (got %d)", bits);
    QA_REQUIRE(m >= 0 && d >= 0, QUIPAMD_ERR_SHAPE, "pack/unpack: negative shape");
    if (layout == QUIPAMD_LAYOUT_CANONICAL && bits == 3) {
        QA_REQUIRE(d % 32 == 0, QUIPAMD_ERR_SHAPE, "canonical 3-bit layout needs d %% 32 == 0 (d=%lld)", (long long)d);
    } else if (layout == QUIPAMD_LAYOUT_CANONICAL) {
        QA_REQUIRE(d % (32 / bits) == 0 && d % 16 == 0, QUIPAMD_ERR_SHAPE,
                   "canonical layout needs d %% %d == 0 (d=%lld)", 32 / bits > 16 ? 32 / bits : 16, (long long)d);
    } else if (layout == QUIPAMD_LAYOUT_STREAM) {
        QA_REQUIRE(m % 16 == 0 && d % (512 / bits) == 0, QUIPAMD_ERR_SHAPE,
                   "stream layout needs m %% 16 == 0 and d %% %d == 0 (m=%lld d=%lld)", 512 / bits, (long long)m,
                   (long long)d);
    } else {
        return qa_fail(QUIPAMD_ERR_ARG, "unknown layout %d", layout);
    }
    return QUIPAMD_OK;
}

}   // namespace

// 3-bit codes (--wbits 3) ride in the 4-bit STREAM container: K2 dequantises nibbles, the grid (maxq = 7) lives in its
// epilogue.  The CANONICAL 3-bit layout is the reference's own 32-codes-in-3-words rule (quant.py:185-220).
static inline int container_bits(int bits, int layout) { return (bits == 3 && layout == QUIPAMD_LAYOUT_STREAM) ? 4 : bits; }

extern "C" int quipamd_pack(const uint8_t *codes, int bits, int layout, int32_t *packed, int64_t m, int64_t d,
                            void *stream)
{
    bits = container_bits(bits, layout);
    int rc = check_pack_args(codes, packed, bits, layout, m, d);
    if (rc) return rc;
    if (m == 0 || d == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (layout == QUIPAMD_LAYOUT_CANONICAL && bits == 3) {
        pack3_canonical_kernel<<<qa_div_up((d / 32) * m, 256), 256, 0, s>>>(codes, (uint32_t *)packed, m, d);
    } else if (layout == QUIPAMD_LAYOUT_CANONICAL) {
        const int64_t nwords = d / (32 / bits);
        dim3 grid(qa_div_up(nwords, TILE), qa_div_up(m, TILE));
        if (bits == 2) pack_canonical_kernel<2><<<grid, 256, 0, s>>>(codes, (uint32_t *)packed, m, d);
        else pack_canonical_kernel<4><<<grid, 256, 0, s>>>(codes, (uint32_t *)packed, m, d);
    } else {
        const int64_t n = (m / 16) * (d / (512 / bits)) * 64;
        if (bits == 2) pack_stream_kernel<2><<<qa_div_up(n, 256), 256, 0, s>>>(codes, (uint4 *)packed, m, d);
        else pack_stream_kernel<4><<<qa_div_up(n, 256), 256, 0, s>>>(codes, (uint4 *)packed, m, d);
    }
    QA_LAUNCH_CHECK("quipamd_pack");
    return QUIPAMD_OK;
}

extern "C" int quipamd_unpack(const int32_t *packed, int bits, int layout, uint8_t *codes, int64_t m, int64_t d,
                              void *stream)
{
    bits = container_bits(bits, layout);
    int rc = check_pack_args(packed, codes, bits, layout, m, d);
    if (rc) return rc;
    if (m == 0 || d == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (layout == QUIPAMD_LAYOUT_CANONICAL && bits == 3) {
        unpack3_canonical_kernel<<<qa_div_up((d / 32) * m, 256), 256, 0, s>>>((const uint32_t *)packed, codes, m, d);
    } else if (layout == QUIPAMD_LAYOUT_CANONICAL) {
        const int64_t nwords = d / (32 / bits);
        dim3 grid(qa_div_up(nwords, TILE), qa_div_up(m, TILE));
        if (bits == 2) unpack_canonical_kernel<2><<<grid, 256, 0, s>>>((const uint32_t *)packed, codes, m, d);
        else unpack_canonical_kernel<4><<<grid, 256, 0, s>>>((const uint32_t *)packed, codes, m, d);
    } else {
        const int64_t n = (m / 16) * (d / (512 / bits)) * 64;
        if (bits == 2) unpack_stream_kernel<2><<<qa_div_up(n, 256), 256, 0, s>>>((const uint4 *)packed, codes, m, d);
        else unpack_stream_kernel<4><<<qa_div_up(n, 256), 256, 0, s>>>((const uint4 *)packed, codes, m, d);
    }
    QA_LAUNCH_CHECK("quipamd_unpack");
    return QUIPAMD_OK;
}

// CANONICAL -> STREAM on the device: a checkpoint written by the reference's packers (opt.py:303-315 opt_pack3 ->
// Quant3Linear.pack; zeroShot/models/quant.py:190-199 Quant4Linear) becomes what K2 streams without a host pass.
extern "C" int quipamd_repack_canonical_to_stream(const int32_t *canonical, int bits, int32_t *stream_out, int64_t m, int64_t d,
                                                  void *stream)
{
    QA_REQUIRE((canonical && stream_out) || m == 0 || d == 0, QUIPAMD_ERR_ARG, "repack: null pointer");
    QA_REQUIRE(bits == 2 || bits == 3 || bits == 4, QUIPAMD_ERR_UNSUPPORTED, "repack: bits must be 2, 3 or 4");
    const int cb = bits == 3 ? 4 : bits;
    QA_REQUIRE(m >= 0 && d >= 0 && m % 16 == 0 && d % (512 / cb) == 0, QUIPAMD_ERR_SHAPE,
               "repack: stream layout needs m %% 16 == 0 and d %% %d == 0 (m=%lld d=%lld)", 512 / cb, (long long)m, (long long)d);
    if (m == 0 || d == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n = (m / 16) * (d / (512 / cb)) * 64;
    if (bits == 2) repack_stream_kernel<2><<<qa_div_up(n, 256), 256, 0, s>>>((const uint32_t *)canonical, (uint4 *)stream_out, m, d);
    else if (bits == 3) repack_stream_kernel<3><<<qa_div_up(n, 256), 256, 0, s>>>((const uint32_t *)canonical, (uint4 *)stream_out, m, d);
    else repack_stream_kernel<4><<<qa_div_up(n, 256), 256, 0, s>>>((const uint32_t *)canonical, (uint4 *)stream_out, m, d);
    QA_LAUNCH_CHECK("quipamd_repack_canonical_to_stream");
    return QUIPAMD_OK;
}
