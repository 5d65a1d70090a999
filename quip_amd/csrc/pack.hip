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

int check_pack_args(const void *a, const void *b, int bits, int layout, int64_t m, int64_t d)
{
    QA_REQUIRE((a && b) || m == 0 || d == 0, QUIPAMD_ERR_ARG, "pack/unpack: null pointer");
    QA_REQUIRE(bits == 2 || bits == 4, QUIPAMD_ERR_UNSUPPORTED, "pack/unpack: container bits must be 2 or 4 (got %d)", bits);
    QA_REQUIRE(m >= 0 && d >= 0, QUIPAMD_ERR_SHAPE, "pack/unpack: negative shape");
    if (layout == QUIPAMD_LAYOUT_CANONICAL) {
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
// epilogue.  The reference's 32-codes-in-3-words rule (quant.py:185-220) is restated in the oracle only.
static inline int container_bits(int bits, int layout) { return (bits == 3 && layout == QUIPAMD_LAYOUT_STREAM) ? 4 : bits; }

extern "C" int quipamd_pack(const uint8_t *codes, int bits, int layout, int32_t *packed, int64_t m, int64_t d,
                            void *stream)
{
    bits = container_bits(bits, layout);
    int rc = check_pack_args(codes, packed, bits, layout, m, d);
    if (rc) return rc;
    if (m == 0 || d == 0) return QUIPAMD_OK;
    hipStream_t s = (hipStream_t)stream;
    if (layout == QUIPAMD_LAYOUT_CANONICAL) {
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
    if (layout == QUIPAMD_LAYOUT_CANONICAL) {
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
