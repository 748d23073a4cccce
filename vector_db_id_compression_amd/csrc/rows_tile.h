// rows_tile.h -- graph rows (faiss::nsg::Graph<int32_t>: int32 [N, K], -1 terminated; altid_impl.cpp:20-165) move between
// HBM and the "one row per LANE" kernels in TILES of 64 consecutive rows.
//
// A wavefront owns 64 rows.  Their 64*K int32 are ONE contiguous block of the row array: the wavefront reads it with
// fully coalesced 16-byte loads (lane j takes chunk i*64 + j, 1 KiB per instruction) and transposes it through LDS
// (row t, entry e at lds[t * (KP + 1) + e]: odd leading dimension, so both the row-major fill and the per-lane
// column read are free of bank conflicts), after which lane t holds row t in registers.  Decoders go the other
// way.  Before (rounds 1-5) every lane read its own row with 16-byte loads 4*K bytes apart: 64 sectors per wave
// instruction, rows fetched twice.
//
// A workgroup is ONE wavefront (blockDim.x == 64): __syncthreads() only orders the wavefront's own LDS accesses.
#pragma once
#include "wave.h"

namespace vidc {
namespace dev {

template <int KP>
struct RowsTile {
    static constexpr uint32_t LD = KP + 1;          // leading dimension of the transposed image (dwords)
    static constexpr uint32_t DWORDS = 64u * LD;    // LDS dwords a kernel has to provide
};

// host: ceil(2^32 / d): floor(f / d) == __umulhi(f, magic) for f * d < 2^32 (f < 2^16 is all the tiles need)
inline __host__ __device__ uint32_t tile_magic(uint32_t d) { return d > 1u ? (uint32_t)((0x100000000ull + d - 1u) / d) : 0u; }
__device__ __forceinline__ uint32_t tile_div(uint32_t f, uint32_t d, uint32_t magic) { return d > 1u ? __umulhi(f, magic) : f; }

// rows [row0, row0 + nrows) of the int32 [N, K] array -> r[0..KP) of lane t = row row0 + t (entries >= K and rows >= nrows: -1).
// vec: the row array starts on a 16-byte boundary (then every tile does).  Ends with the LDS free for reuse.
template <int KP>
__device__ __forceinline__ void tile_load_rows(const int32_t *__restrict__ rows, uint64_t row0, uint32_t nrows, uint32_t K,
                                               uint32_t kmagic, bool vec, uint32_t *lds, uint32_t (&r)[KP]) {
    constexpr uint32_t LD = RowsTile<KP>::LD;
    const uint32_t lane = lane_id();
    const int32_t *src = rows + row0 * K;
    if (K == (uint32_t)KP && vec) {
        constexpr int NC = KP / 4;  // 16-byte chunks per row == chunk loads per lane
        const uint32_t nchunk = nrows * (uint32_t)NC;
        int4 v[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const uint32_t c = (uint32_t)i * 64u + lane;
            v[i] = make_int4(-1, -1, -1, -1);
            if (c < nchunk) v[i] = ((const int4 *)src)[c];
        }
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const uint32_t c = (uint32_t)i * 64u + lane;
            uint32_t *d = lds + (c / (uint32_t)NC) * LD + (c % (uint32_t)NC) * 4u;
            d[0] = (uint32_t)v[i].x; d[1] = (uint32_t)v[i].y; d[2] = (uint32_t)v[i].z; d[3] = (uint32_t)v[i].w;
        }
    } else {
        const uint32_t total = nrows * K;
#pragma unroll 8
        for (int i = 0; i < KP; i++) {
            const uint32_t f = (uint32_t)i * 64u + lane;
            if (f < total) {
                const uint32_t row = tile_div(f, K, kmagic);
                lds[row * LD + (f - row * K)] = (uint32_t)src[f];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < KP; e++) r[e] = ((uint32_t)e < K && lane < nrows) ? lds[lane * LD + (uint32_t)e] : 0xffffffffu;
    __syncthreads();
}

// the transposed image (row t, entry e at lds[t * LD + e], every entry e < K of every row t < nrows defined) -> rows
// [row0, row0 + nrows) of the int32 [m, K] output, coalesced
template <int KP>
__device__ __forceinline__ void tile_store_rows(int32_t *__restrict__ out, uint64_t row0, uint32_t nrows, uint32_t K,
                                                uint32_t kmagic, bool vec, const uint32_t *lds) {
    constexpr uint32_t LD = RowsTile<KP>::LD;
    const uint32_t lane = lane_id();
    int32_t *dst = out + row0 * K;
    if (K == (uint32_t)KP && vec) {
        constexpr int NC = KP / 4;
        const uint32_t nchunk = nrows * (uint32_t)NC;
#pragma unroll
        for (int i = 0; i < NC; i++) {
            const uint32_t c = (uint32_t)i * 64u + lane;
            const uint32_t *s = lds + (c / (uint32_t)NC) * LD + (c % (uint32_t)NC) * 4u;
            const int4 v = make_int4((int)s[0], (int)s[1], (int)s[2], (int)s[3]);
            if (c < nchunk) ((int4 *)dst)[c] = v;
        }
    } else {
        const uint32_t total = nrows * K;
#pragma unroll 8
        for (int i = 0; i < KP; i++) {
            const uint32_t f = (uint32_t)i * 64u + lane;
            if (f < total) {
                const uint32_t row = tile_div(f, K, kmagic);
                dst[f] = (int32_t)lds[row * LD + (f - row * K)];
            }
        }
    }
}

// edges of a row = entries before the first -1 (altid_impl.cpp:61-68,110-117); entries from there on become 0xffffffff.
// bad: a negative id other than the terminator among them.  mx: the largest id (0 for an empty row).
template <int KP>
__device__ __forceinline__ uint32_t tile_row_edges(uint32_t (&r)[KP], uint32_t K, bool &bad, uint32_t &mx) {
    uint32_t n = K;
#pragma unroll
    for (int e = KP - 1; e >= 0; e--) n = (r[e] == 0xffffffffu && (uint32_t)e < n) ? (uint32_t)e : n;
    int32_t m = 0;
    bad = false;
#pragma unroll
    for (int e = 0; e < KP; e++) {
        r[e] = (uint32_t)e < n ? r[e] : 0xffffffffu;
        bad |= (uint32_t)e < n && (int32_t)r[e] < 0;
        m = (int32_t)r[e] > m ? (int32_t)r[e] : m;  // (signed: padding is -1)
    }
    mx = (uint32_t)m;
    return n;
}

}  // namespace dev
}  // namespace vidc
