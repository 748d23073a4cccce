// rows_tile.h -- graph rows (faiss::nsg::Graph<int32_t>: int32 [N, K], -1 terminated; altid_impl.cpp:20-165) move between
// HBM and the "one row per LANE" kernels in TILES of 64 consecutive rows.
//
// A wavefront owns 64 rows.  Their 64*K int32 are ONE contiguous block of the row array: the wavefront reads it with
// fully coalesced 16-byte loads (lane j takes chunk i*64 + j, 1 KiB per instruction) and transposes it through LDS,
// 32 columns at a time (row t, column e at lds[t * 33 + (e & 31)]: odd leading dimension, so both the row-major
// fill and the per-lane column read are free of bank conflicts), after which lane t holds row t in registers.
// Decoders go the other way.  Before (rounds 1-5) every lane read its own row with 16-byte loads 4*K bytes apart:
// 64 sectors per wave instruction, rows fetched twice.
//
// A workgroup is ONE wavefront (blockDim.x == 64): wave_lds_sync() (wave.h) orders the wavefront's own LDS accesses
// without waiting for its outstanding global loads and stores, which __syncthreads() would.
#pragma once
#include "wave.h"

namespace vidc {
namespace dev {

template <int KP>
struct RowsTile {
    static constexpr uint32_t LD = KP + 1;          // leading dimension of the transposed image (dwords)
    static constexpr uint32_t DWORDS = 64u * LD;    // LDS dwords of the image of a whole tile (rows read 4 bytes per lane)
    static constexpr uint32_t DWORDS_PF = 64u * 33u;  // ... of a block requested with tile_issue_rows: 32 columns at a time
};

// host: wavefronts (= workgroups) of a persistent tile kernel: `per_cu` resident per CU, never more than there are tiles
inline uint32_t tile_grid(int num_cu, uint64_t ntiles, uint32_t per_cu) {
    const uint64_t cap = (uint64_t)(num_cu > 0 ? num_cu : 256) * (per_cu ? per_cu : 1u);
    return (uint32_t)(ntiles < cap ? (ntiles ? ntiles : 1u) : cap);
}

// host: ceil(2^32 / d): floor(f / d) == __umulhi(f, magic) for f * d < 2^32 (f < 2^16 is all the tiles need)
inline __host__ __device__ uint32_t tile_magic(uint32_t d) { return d > 1u ? (uint32_t)((0x100000000ull + d - 1u) / d) : 0u; }
__device__ __forceinline__ uint32_t tile_div(uint32_t f, uint32_t d, uint32_t magic) { return d > 1u ? __umulhi(f, magic) : f; }

// The DECODERS are persistent: a wavefront walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and requests the records of its
// NEXT tile (tile_fetch16: 16-byte loads into registers, nothing waits) before it works on the current one, so the HBM round trip
// of a tile hides behind the arithmetic of the previous tile.  The ENCODERS take one tile per wavefront: holding a second tile's
// rows in registers across the encode costs them 254 VGPRs and a store-acknowledge wait at every commit (measured, round 6).
// tile_issue_rows requests a tile's block of rows, every load ahead of the first use.
template <int KP>
struct TileRegs {
    int4 v[KP / 4];
};
// K == KP and a 16-byte aligned row array only (the caller's `pf`): lane j requests pieces i * 64 + j of the tile's block
template <int KP>
__device__ __forceinline__ void tile_issue_rows(const int32_t *__restrict__ rows, uint64_t row0, uint32_t nrows, TileRegs<KP> &t) {
    constexpr int NC = KP / 4;
    const uint32_t lane = lane_id();
    const int4 *src = (const int4 *)(rows + row0 * (uint64_t)KP);
    const uint32_t nchunk = nrows * (uint32_t)NC;
#pragma unroll
    for (int i = 0; i < NC; i++) {
        const uint32_t c = (uint32_t)i * 64u + lane;
        t.v[i] = src[c < nchunk ? c : 0u];  // (a short last tile re-reads its first piece: never out of bounds)
    }
}
// rows [row0, row0 + nrows) -> r[0..KP) of lane t = row row0 + t (entries >= K: -1).  FULL: K == KP.  pf: the block was requested
// with tile_issue_rows (FULL kernels, aligned row array); otherwise it is read here, 4 bytes per lane and load.
// Ends with the LDS free for reuse.  Lanes >= nrows (last tile only) hold whatever the LDS held: the caller discards what it
// derives from them.
template <int KP, bool FULL>
__device__ __forceinline__ void tile_commit_rows(const int32_t *__restrict__ rows, uint64_t row0, uint32_t nrows, uint32_t K,
                                                 uint32_t kmagic, bool pf, const TileRegs<KP> &t, uint32_t *lds, uint32_t (&r)[KP]) {
    constexpr uint32_t LD = RowsTile<KP>::LD;
    const uint32_t lane = lane_id();
    if (FULL && pf) {
        // 32 columns at a time (LD = 33): the image of a K = 64 tile takes 8.25 KiB instead of 16.25, which is what bounds the
        // wavefronts per CU of the encoders
        constexpr int NC = KP / 4, HALVES = KP / 32, LDH = 33;
#pragma unroll
        for (int h = 0; h < HALVES; h++) {
            if (h) wave_lds_sync();
#pragma unroll
            for (int i = 0; i < NC; i++) {
                const uint32_t c = (uint32_t)i * 64u + lane;  // piece c: row c / NC, columns 4 (c % NC) ..
                const uint32_t col = (c % (uint32_t)NC) * 4u;
                if (col / 32u == (uint32_t)h) {
                    uint32_t *d = lds + (c / (uint32_t)NC) * LDH + (col & 31u);
                    d[0] = (uint32_t)t.v[i].x; d[1] = (uint32_t)t.v[i].y; d[2] = (uint32_t)t.v[i].z; d[3] = (uint32_t)t.v[i].w;
                }
            }
            wave_lds_sync();
#pragma unroll
            for (int e = 0; e < 32; e++) r[h * 32 + e] = lds[lane * LDH + (uint32_t)e];
        }
        wave_lds_sync();
        return;
    } else {
        const int32_t *src = rows + row0 * K;
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
    wave_lds_sync();
#pragma unroll
    for (int e = 0; e < KP; e++) r[e] = lds[lane * LD + (uint32_t)e];
    if (!FULL) {
#pragma unroll
        for (int e = 0; e < KP; e++) r[e] = (uint32_t)e < K ? r[e] : 0xffffffffu;
    }
    wave_lds_sync();
}

// edges of a row = entries before the first -1 (altid_impl.cpp:61-68,110-117); entries from there on become 0xffffffff.
// bad: a negative id other than the terminator among them.  mx: the largest id (0 for an empty row).
// (`alive` is a lane mask in a scalar register pair: its update is one v_cmp and one s_and per entry.)
template <int KP>
__device__ __forceinline__ uint32_t tile_row_edges(uint32_t (&r)[KP], bool &bad, uint32_t &mx) {
    bool alive = true;
    uint32_t n = 0;
#pragma unroll
    for (int e = 0; e < KP; e++) {  // (entries e >= K are 0xffffffff)
        alive = alive && r[e] != 0xffffffffu;
        n += alive ? 1u : 0u;
        r[e] = alive ? r[e] : 0xffffffffu;
    }
    // as signed values the padding is -1, any other negative id is below it and the ids are >= 0
    int32_t hi = 0, lo = 0;
#pragma unroll
    for (int e = 0; e < KP; e += 2) {
        const int32_t a = (int32_t)r[e], b = e + 1 < KP ? (int32_t)r[e + 1] : -1;
        hi = max(hi, max(a, b));
        lo = min(lo, min(a, b));
    }
    bad = lo < -1;
    mx = (uint32_t)hi;
    return n;
}

// (record, unit) of the flat unit index i * STEP + start over records of D units, advanced without a division
struct TileCursor {
    uint32_t row, w;
};
__device__ __forceinline__ TileCursor tile_cursor(uint32_t D, uint32_t dmagic, uint32_t start) {
    TileCursor c;
    c.row = tile_div(start, D, dmagic);
    c.w = start - c.row * D;
    return c;
}
__device__ __forceinline__ TileCursor tile_cursor(uint32_t D, uint32_t dmagic) { return tile_cursor(D, dmagic, lane_id()); }
__device__ __forceinline__ void tile_advance(TileCursor &c, uint32_t D, uint32_t step_rows, uint32_t step_w) {  // += step_rows * D + step_w
    c.w += step_w;
    c.row += step_rows;
    const bool wrap = c.w >= D;
    c.w -= wrap ? D : 0u;
    c.row += wrap ? 1u : 0u;
}

// A contiguous block of `total` 16-byte pieces (total <= 64 * NMAX) into registers, every load issued before anything waits:
// piece i * 64 + lane -> pv[i] (pieces beyond the block: zero).  A loop that loads, waits and stores one piece per trip keeps ONE
// load in flight per wavefront: ten HBM round trips in a row for a tile of compact-bit records.
template <int NMAX>
__device__ __forceinline__ void tile_fetch16(const uint4 *__restrict__ src, uint32_t total, uint4 (&pv)[NMAX]) {
    const uint32_t lane = lane_id();
#pragma unroll
    for (int i = 0; i < NMAX; i++) {
        pv[i] = make_uint4(0u, 0u, 0u, 0u);
        if ((uint32_t)i * 64u < total) {  // (uniform)
            const uint32_t f = (uint32_t)i * 64u + lane;
            pv[i] = src[f < total ? f : total - 1u];
        }
    }
}

}  // namespace dev
}  // namespace vidc
