// packed.hip -- fixed-width packed IDs (CompressedIDInvertedListsPackedBits,
// custom_invlists_impl.cpp:64-118; CompactBitNSGGraph, altid_impl.cpp:20-51).
//
// Layout per list: ceil(ls*bits/8) bytes, value i at bit i*bits, LSB-first inside a byte and
// little-endian across bytes (what BitstringReader_get_bits decodes, custom_invlists_impl.cpp:35-58).
// On the device each list starts on an 8-byte boundary so that every 64-bit word has exactly one
// owner thread (no atomics); the reported size follows the reference formula, not the padding.
// HBM-bound: 8 B read + bits/8 B written per id (encode), the mirror for decode.
#include <algorithm>
#include <memory>

#include "bits.h"
#include "chunks.h"
#include "common.h"
#include "rows_tile.h"
#include "scan.h"

using namespace vidc;
using namespace vidc::dev;

struct vidc_packed {
    int device = 0;
    uint64_t nlist = 0, ntotal = 0;
    int bits = 0;
    uint64_t compressed_bytes = 0, total_words = 0;
    std::vector<uint64_t> offsets;
    DevBuf<uint64_t> d_offsets, d_word_off, d_words;
    DevBuf<Chunk> d_chunks;
    uint64_t nchunks = 0;
    uint64_t max_list = 0;  // ids of the longest list: objects without a list of more than 256 ids take the four-register kernels
    ~vidc_packed() { vidc::vec_pool<uint64_t>().give(std::move(offsets)); }  // (back to the process-wide vector cache, common.h)
};

namespace {

// one wavefront per chunk of CHUNK_IDS ids.  The 8 ids of a lane are loaded first (coalesced, independent: 8 loads
// in flight per lane), OR-ed into an LDS image of the chunk's words (a chunk starts on a word boundary of its list:
// CHUNK_IDS * bits is a multiple of 64) and the image is written out with coalesced stores.  (A word-centric
// gather -- one owner lane per output word looping over the ids that touch it -- measured 2.3x slower: its loads
// sit in a data-dependent loop, one memory round trip per iteration.)
// R = id registers per lane: 8 = a whole chunk; 4 for objects without a list of more than 256 ids (a chunk pays for the
// unrolled code of every register, used or not; these kernels are not bound by their instructions, though: 16 M ids in
// lists of 256 encode 55 -> 53 us, decode 36 -> 35 us, against 83 -> 70 us for the Elias-Fano encoder's same change)
template <int R>
__global__ void __launch_bounds__(64) k_packed_encode(const uint64_t *ids, const uint64_t *offsets,
                                                      const uint64_t *word_off, const Chunk *chunks, uint64_t nchunks,
                                                      uint32_t bits, uint64_t id_limit, uint64_t *words, uint32_t *err) {
    __shared__ unsigned long long img[64 * R + 8];  // <= 64 R * 64 / 64 words
    const uint32_t lane = threadIdx.x;
    const uint64_t keep = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const Chunk ch = chunks[c];
        const uint64_t off = offsets[ch.list];
        const uint64_t n = offsets[ch.list + 1] - off;
        const uint32_t nc = (uint32_t)(n - ch.start < CHUNK_IDS ? n - ch.start : CHUNK_IDS);
        const uint64_t w0 = ((uint64_t)ch.start * bits) >> 6;  // exact: start * bits % 64 == 0
        const uint32_t nw = (uint32_t)(((uint64_t)nc * bits + 63) >> 6);
        const uint64_t *src = ids + off + ch.start;
        uint64_t v[R];
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
            const uint32_t i = lane + 64 * r;
            v[r] = i < nc ? src[i] : 0ull;
        }
        for (uint32_t w = lane; w < nw + 1u; w += 64) img[w] = 0;
        wave_lds_sync();
        bool bad = false;
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
            const uint32_t i = lane + 64 * r;
            if (i < nc) {
                bad |= v[r] >= id_limit || (bits < 64 && (v[r] >> bits));
                const uint64_t x = v[r] & keep;
                const uint32_t pos = i * bits, sh = pos & 63u;
                atomicOr(&img[pos >> 6], x << sh);
                if (sh + bits > 64u) atomicOr(&img[(pos >> 6) + 1u], x >> (64u - sh));
            }
        }
        if (__ballot(bad) && lane == 0) *(volatile uint32_t *)err = 1u;  // (every writer stores the same value: device or pinned host memory)
        wave_lds_sync();
        uint64_t *dst = words + word_off[ch.list] + w0;
        for (uint32_t w = lane; w < nw; w += 64) dst[w] = img[w];
        wave_lds_sync();
    }
}

// The geometry of an object in ONE launch: chunk table, word offsets and the zeroed padding word of every list.  (Chunk counts -> scan
// -> fill, a host-built word-offset array and the padding kernel were six dependent launches and a second 8-byte-per-list upload:
// 25 us of a 60 us encode of 16 M ids.)  A tile of 4096 lists counts its chunks and words, takes its place in a chained scan -- tile
// numbers handed out by a counter in the order the workgroups start, so a tile only waits for tiles that are already running (as
// k_roc_tail) -- and writes its part.  state: [gridDim.x chunk sums | gridDim.x word sums | tile counter], zero at the launch.
// inclusive prefix sums over the 64 lanes in six DPP adds (row_shr 1 2 4 8, row_bcast 15 31); 64-bit values below 2^45 as two limbs
__device__ __forceinline__ uint32_t pk_scan32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ uint64_t pk_scan64(uint64_t v) {
    const uint32_t lo = pk_scan32((uint32_t)v & 0xfffffu), hi = pk_scan32((uint32_t)(v >> 20));
    return ((uint64_t)hi << 20) + lo;
}
// Round 5: an object of ONE tile (up to 4095 lists: what a search-sized call holds) spreads its lists over all 256 threads -- `per`
// lists per thread instead of 16 for the first nlist / 16 threads -- and the prefix sums over the threads are two DPP wave scans + one
// exchange of four wave totals instead of sixteen Hillis-Steele rounds of two barriers each (1 M ids in 1024 lists: 17 -> ~8 us of a 27 us
// encode + decode).
__global__ void __launch_bounds__(256) k_packed_table(const uint64_t *__restrict__ offsets, uint32_t nlist, uint32_t bits,
                                                      unsigned long long *state, Chunk *__restrict__ chunks,
                                                      uint64_t *__restrict__ word_off, uint64_t *__restrict__ words, uint32_t per) {
    __shared__ uint64_t sh[2][256];
    __shared__ uint64_t wsum[2][4];
    __shared__ uint64_t tile_off_s[2];
    __shared__ uint32_t tile_s, qn;
    __shared__ struct { uint64_t a0; uint32_t l, cnt; } q[256];  // lists of many chunks: their items are written by the whole workgroup
    const uint32_t t = threadIdx.x, nt = gridDim.x;
    if (t == 0) qn = 0u;
    if (nt > 1u) {  // (an object of one tile needs neither the counter nor a cleared state)
        if (t == 0) tile_s = (uint32_t)atomicAdd(&state[2 * nt], 1ull);
        __syncthreads();
    }
    const uint32_t tile = nt > 1u ? tile_s : 0u;
    // `per` lists (and the index nlist, which receives the total) per thread, 256 * per in a tile (packed_table_per)
    const uint64_t base = (uint64_t)tile * (256u * per) + t * per;
    uint32_t cnt[16];
    uint64_t wc[16];
    uint64_t s0 = 0, s1 = 0;
    uint64_t prev = offsets[base < nlist ? base : nlist];
#pragma unroll
    for (int j = 0; j < 16; j++) {
        cnt[j] = 0u;
        wc[j] = 0ull;
        if ((uint32_t)j < per) {
            const uint64_t l = base + j;
            const uint64_t next = offsets[l + 1 < nlist ? l + 1 : nlist];
            const uint64_t n = next - prev;
            cnt[j] = l < nlist ? (uint32_t)((n + CHUNK_IDS - 1u) / CHUNK_IDS) : 0u;
            wc[j] = l < nlist ? (n * bits + 63) / 64 + 1 : 0ull;  // +1: read_bits may touch the next word
            prev = next;
            s0 += cnt[j];
            s1 += wc[j];
        }
    }
    // inclusive prefixes over the workgroup's threads (chunk / word counts of a tile stay far below 2^45)
    uint64_t incl0 = pk_scan64(s0), incl1 = pk_scan64(s1);
    if ((t & 63u) == 63u) { wsum[0][t >> 6] = incl0; wsum[1][t >> 6] = incl1; }
    __syncthreads();
    uint64_t sum0 = 0, sum1 = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4u; w++) {
        const uint64_t x0 = wsum[0][w], x1 = wsum[1][w];
        if (w < (t >> 6)) { incl0 += x0; incl1 += x1; }
        sum0 += x0;
        sum1 += x1;
    }
    if (t == 0 && nt > 1u) {
        // (no fence in front: the published values are the exchanges' own operands; a device-scope release writes back the XCD's whole L2)
        atomicExch(&state[tile], (1ull << 63) | sum0);
        atomicExch(&state[nt + tile], (1ull << 63) | sum1);
    }
    uint64_t t_off0 = 0, t_off1 = 0;
    if (nt > 1u) {  // (uniform) sums of the tiles before this one
        uint64_t b0 = 0, b1 = 0;
        for (uint32_t k = t; k < tile; k += 256u) {
            unsigned long long x;
            do { x = atomicAdd(&state[k], 0ull); } while (!(x >> 63));
            b0 += x & ~(1ull << 63);
            do { x = atomicAdd(&state[nt + k], 0ull); } while (!(x >> 63));
            b1 += x & ~(1ull << 63);
        }
        __syncthreads();
        sh[0][t] = b0;
        sh[1][t] = b1;
        __syncthreads();
        for (uint32_t o = 128; o > 0; o >>= 1) {
            if (t < o) { sh[0][t] += sh[0][t + o]; sh[1][t] += sh[1][t + o]; }
            __syncthreads();
        }
        if (t == 0) { tile_off_s[0] = sh[0][0]; tile_off_s[1] = sh[1][0]; }
        __syncthreads();
        t_off0 = tile_off_s[0];
        t_off1 = tile_off_s[1];
    }
    uint64_t a0 = t_off0 + incl0 - s0, a1 = t_off1 + incl1 - s1;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if ((uint32_t)j >= per) continue;
        const uint64_t l = base + j;
        if (l <= nlist) word_off[l] = a1;  // (index nlist receives the total)
        // (an index stored longest list first puts 16 lists of 128 chunks each into one thread: 10 M ids in 65 536 Zipf lists spent 40 us here)
        uint32_t slot = 256u;
        if (cnt[j] > 8u) slot = atomicAdd(&qn, 1u);
        if (slot < 256u) { q[slot].a0 = a0; q[slot].l = (uint32_t)l; q[slot].cnt = cnt[j]; }
        else for (uint32_t c = 0; c < cnt[j]; c++) chunks[a0 + c] = Chunk{(uint32_t)l, c * CHUNK_IDS};
        a0 += cnt[j];
        a1 += wc[j];
        if (l < nlist) words[a1 - 1] = 0ull;  // the padding word behind the list's stream
    }
    __syncthreads();
    const uint32_t nq = qn < 256u ? qn : 256u;
    for (uint32_t k = 0; k < nq; k++) {
        const uint64_t b = q[k].a0;
        const uint32_t l = q[k].l, n = q[k].cnt;
        for (uint32_t c = t; c < n; c += 256u) chunks[b + c] = Chunk{l, c * CHUNK_IDS};
    }
}

// one wavefront per chunk: lane t decodes ids t, t+64, ... (coalesced 8-byte stores).  Both words an id can touch
// are loaded unconditionally for all 8 ids of the lane before anything is consumed (16 independent loads in flight
// per lane; a padding word follows every list): a conditional second load costs a second memory round trip per id.
template <int R>
__global__ void __launch_bounds__(64) k_packed_decode(const uint64_t *words, const uint64_t *offsets,
                                                      const uint64_t *word_off, const Chunk *chunks, uint64_t nchunks,
                                                      uint32_t bits, uint64_t *out) {
    const uint32_t lane = threadIdx.x;
    const uint64_t keep = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const Chunk ch = chunks[c];
        const uint64_t off = offsets[ch.list];
        const uint64_t n = offsets[ch.list + 1] - off;
        const uint32_t nc = (uint32_t)(n - ch.start < CHUNK_IDS ? n - ch.start : CHUNK_IDS);
        const uint64_t *src = words + word_off[ch.list];
        uint64_t *dst = out + off + ch.start;
        uint64_t a[R], b[R];
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
            const uint32_t i = lane + 64 * r;
            const uint64_t pos = (uint64_t)(ch.start + (i < nc ? i : 0u)) * bits;
            a[r] = src[pos >> 6];
            b[r] = src[(pos >> 6) + 1];
        }
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
            const uint32_t i = lane + 64 * r;
            const uint32_t sh = (uint32_t)(((uint64_t)(ch.start + i) * bits) & 63);
            const uint64_t v = (a[r] >> sh) | (sh ? b[r] << (64 - sh) : 0ull);
            if (i < nc) dst[i] = v & keep;
        }
    }
}

// the same for a request of lists (get_ids of the lists a search touched, custom_invlists_impl.cpp:96-105 per list): one
// wavefront per (request item, chunk); item i's ids go to out + out_off[i]
struct PackedItem {
    uint32_t item;
    uint32_t start;
};
__global__ void __launch_bounds__(64) k_packed_decode_lists(const uint64_t *words, const uint64_t *offsets,
                                                            const uint64_t *word_off, const uint64_t *list_nos,
                                                            const uint64_t *out_off, const PackedItem *items,
                                                            uint64_t nitems, uint32_t bits, uint64_t *out) {
    const uint32_t lane = threadIdx.x;
    const uint64_t keep = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
    for (uint64_t c = blockIdx.x; c < nitems; c += gridDim.x) {
        const PackedItem it = items[c];
        const uint64_t l = list_nos[it.item];
        const uint64_t n = offsets[l + 1] - offsets[l];
        const uint32_t nc = (uint32_t)(n - it.start < CHUNK_IDS ? n - it.start : CHUNK_IDS);
        const uint64_t *src = words + word_off[l];
        uint64_t *dst = out + out_off[it.item] + it.start;
        uint64_t a[CHUNK_IDS / 64], b[CHUNK_IDS / 64];
#pragma unroll
        for (uint32_t r = 0; r < CHUNK_IDS / 64; r++) {
            const uint32_t i = lane + 64 * r;
            const uint64_t pos = (uint64_t)(it.start + (i < nc ? i : 0u)) * bits;
            a[r] = src[pos >> 6];
            b[r] = src[(pos >> 6) + 1];
        }
#pragma unroll
        for (uint32_t r = 0; r < CHUNK_IDS / 64; r++) {
            const uint32_t i = lane + 64 * r;
            const uint32_t sh = (uint32_t)(((uint64_t)(it.start + i) * bits) & 63);
            const uint64_t v = (a[r] >> sh) | (sh ? b[r] << (64 - sh) : 0ull);
            if (i < nc) dst[i] = v & keep;
        }
    }
}

__global__ void k_packed_get(const uint64_t *words, const uint64_t *word_off, uint32_t bits, uint64_t m,
                             const uint64_t *list_nos, const uint64_t *offs, int64_t *out) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < m) out[q] = (int64_t)read_bits(words + word_off[list_nos[q]], offs[q] * bits, bits);
}

// ---- CompactBitNSGGraph (altid_impl.cpp:20-51): row i occupies bytes [i*stride, (i+1)*stride); neighbours are
// written until the first -1, which is replaced by the sentinel N and ends the row (:28-37).
// One wavefront per row (K <= 64): lane j owns value j and ORs it into an LDS image of the row (values
// straddle bytes), which is then written out with coalesced byte stores.
__global__ void __launch_bounds__(64) k_compact_rows_encode(const int32_t *rows, uint64_t N, uint32_t K, uint32_t bits,
                                                            uint32_t stride, uint8_t *out, uint32_t *err) {
    __shared__ uint32_t img[72];  // stride <= ceil(64 * 32 / 8) = 256 bytes (+ spill word)
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t row = blockIdx.x; row < N; row += gridDim.x) {
        img[lane] = 0;
        if (lane < 8) img[64 + lane] = 0;
        __syncthreads();
        const int32_t e = lane < K ? rows[row * K + lane] : -1;
        const uint64_t endm = __ballot(e == -1);
        const uint32_t n = endm ? (uint32_t)__builtin_ctzll(endm) : 64u;  // edges before the first -1
        const bool bad = lane < n && (e < 0 || (uint64_t)e >= N);
        if (__ballot(bad) && lane == 0) *(volatile uint32_t *)err = 1u;  // (every writer stores the same value: device or pinned host memory)
        // values 0..n-1 are neighbours; value n (if n < K) is the sentinel N; nothing after it (:31-36)
        if (lane <= n && lane < K) {
            const uint64_t v = lane == n ? N : (uint64_t)(uint32_t)e;
            const uint32_t pos = lane * bits;
            const uint64_t sh = v << (pos & 31);
            atomicOr(&img[pos >> 5], (uint32_t)sh);
            if ((pos & 31) + bits > 32) atomicOr(&img[(pos >> 5) + 1], (uint32_t)(sh >> 32));
        }
        __syncthreads();
        const uint8_t *b = (const uint8_t *)img;
        for (uint32_t t = lane; t < stride; t += 64) out[row * stride + t] = b[t];
        __syncthreads();
    }
}

// one wavefront per requested row (K <= 64): lane j reads value j; the row ends at the sentinel N (:43-49).
// Four rows per iteration: a field (<= 32 bits) lies inside the two aligned dwords around its
// first byte, so every lane issues 2 dword loads per row (not 6 byte loads), all 8 loads of the four rows before
// anything is consumed.  (The buffer ends with 8 padding bytes.)
__global__ void __launch_bounds__(64) k_compact_rows_decode(const uint8_t *data, uint64_t N, uint32_t K, uint32_t bits,
                                                            uint32_t stride, uint64_t m, const uint64_t *nodes,
                                                            int32_t *out, uint32_t *counts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t mask = (1ull << bits) - 1ull;
    const uint32_t pos = lane * bits;
    for (uint64_t q0 = (uint64_t)blockIdx.x * 4u; q0 < m; q0 += (uint64_t)gridDim.x * 4u) {
        uint32_t w0[4], w1[4], sh[4];
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint64_t q = q0 + j < m ? q0 + j : q0;
            const uint8_t *p = data + (nodes ? nodes[q] : q) * stride + (pos >> 3);
            const uintptr_t a = (uintptr_t)p;
            const uint32_t *w = (const uint32_t *)(a & ~(uintptr_t)3);
            sh[j] = (uint32_t)(a & 3u) * 8u + (pos & 7u);
            w0[j] = lane < K ? w[0] : 0u;
            w1[j] = lane < K ? w[1] : 0u;
        }
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) {
            const uint64_t q = q0 + j;
            if (q >= m) break;
            const uint64_t v = lane < K ? ((((uint64_t)w1[j] << 32) | w0[j]) >> sh[j]) & mask : ~0ull;
            const uint64_t endm = __ballot(lane < K && v == N);
            const uint32_t n = endm ? (uint32_t)__builtin_ctzll(endm) : K;
            if (lane < K) out[q * K + lane] = lane < n ? (int32_t)v : -1;
            if (lane == 0) counts[q] = n;
        }
    }
}

// ---- the same two for K <= 64 rows whose stride is a whole number of dwords (every K = 32 / 64 graph), 64 consecutive rows per
// wavefront (rows_tile.h): the rows come in as one coalesced block and are transposed through LDS; lane t packs row t -- the bit position of field e is e * bits for EVERY lane, so shift amounts and dword
// positions are scalar, a row image is built 32 bits at a time in registers (no atomics, no barrier per row, no branch: every
// step stores the dword it is filling, the finished image of a dword overwrites its partial ones) and goes to the lane's LDS
// record; the 64 records leave as one contiguous block of 16-byte stores.  Every byte of the output is written (fields behind the
// sentinel are zero, :31-36): the buffer needs no memset.
template <int KP, bool FULL>  // FULL: K == KP
__global__ void __launch_bounds__(64) k_compact_rows_encode_tile(const int32_t *__restrict__ rows, uint64_t N, uint32_t K, uint32_t kmagic,
                                                                 uint32_t vec, uint32_t bits, uint32_t SD, uint32_t sdmagic,
                                                                 uint32_t *__restrict__ out, uint32_t *err) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // the tile image, then the 64 records: sized by the host
    const uint32_t lane = lane_id();
    const bool pf = FULL && vec != 0u;
    const uint32_t SD1 = SD | 1u;  // odd record stride: conflict-free for any (uniform) dword index
    const uint32_t n32 = (uint32_t)N;
    bool anybad = false;
    // One tile per wavefront.  (Measured, round 6: persistent wavefronts with the next tile's block in flight -- the form the decoders
    // use -- cost the 64 registers of the block plus a wait for the previous tile's stores at every commit (vmcnt counts loads and
    // stores in order): 0.101 ms at 8 wavefronts per CU against 0.094 ms for this form at 9.)
    {
        const uint64_t row0 = (uint64_t)blockIdx.x * 64u;
        const uint32_t nrows = (uint32_t)(N - row0 < 64u ? N - row0 : 64u);
        uint32_t r[KP];
        {
            TileRegs<KP> pre;
            if (pf) tile_issue_rows<KP>(rows, row0, nrows, pre);
            tile_commit_rows<KP, FULL>(rows, row0, nrows, K, kmagic, pf, pre, lds, r);
        }
        uint32_t *rec = lds + lane * SD1;
        bool alive = true, bad = false;  // (lane masks in scalar registers)
        uint32_t lo = 0;                 // the bits of the dword being filled
        uint32_t fill = 0, wofs = 0;     // wave-uniform: bits in `lo`, dword index
#pragma unroll
        for (int e = 0; e < KP; e++) {
            if (FULL || (uint32_t)e < K) {
                // values 0..n-1 are neighbours; value n (if n < K) is the sentinel N; nothing after it (:31-36)
                const uint32_t x = r[e];
                const bool was = alive;
                alive = alive && x != 0xffffffffu;
                bad = bad || (alive && x >= n32);  // (a negative id is >= 2^31 here)
                const uint32_t v = alive ? x : (was ? n32 : 0u);
                const uint64_t t = ((uint64_t)v << fill) | lo;
                rec[wofs] = (uint32_t)t;
                fill += bits;
                const bool full = fill >= 32u;
                lo = full ? (uint32_t)(t >> 32) : (uint32_t)t;
                wofs += full ? 1u : 0u;
                fill -= full ? 32u : 0u;
            }
        }
        if (fill) rec[wofs] = lo;
        anybad = anybad || (bad && lane < nrows);
        wave_lds_sync();
        uint32_t *dst = out + row0 * SD;
        if ((SD & 3u) == 0u) {  // 16-byte pieces never straddle two rows; SD <= 64: at most 16 pieces per lane
            const uint32_t CPR = SD >> 2, total = nrows * CPR;
            const uint32_t cmagic = tile_magic(CPR);
            const uint32_t step_rows = tile_div(64u, CPR, cmagic), step_w = 64u - step_rows * CPR;
            TileCursor c = tile_cursor(CPR, cmagic);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if ((uint32_t)i * 64u < total) {
                    const uint32_t f = (uint32_t)i * 64u + lane;
                    const uint32_t *s = lds + c.row * SD1 + 4u * c.w;
                    if (f < total) ((uint4 *)dst)[f] = make_uint4(s[0], s[1], s[2], s[3]);
                    tile_advance(c, CPR, step_rows, step_w);
                }
            }
        } else {
            const uint32_t total = nrows * SD;
            const uint32_t step_rows = tile_div(64u, SD, sdmagic), step_w = 64u - step_rows * SD;
            TileCursor c = tile_cursor(SD, sdmagic);
            for (uint32_t f = lane; f < total; f += 64u) {
                dst[f] = lds[c.row * SD1 + c.w];
                tile_advance(c, SD, step_rows, step_w);
            }
        }
    }
    if (ballot(anybad) && lane == 0) *(volatile uint32_t *)err = 1u;  // (every writer stores the same value)
}

// decode: persistent wavefronts, 64 records per tile: they come in as one block (nodes == NULL; the next tile's block is requested
// before this one is decoded) or record by record.  QUAD (K % 4 == 0): lane = (row of a group of four, four consecutive fields):
// each field is the two LDS dwords around its bit position (one v_alignbit), the sentinel of a row (:43-49) is found from four
// ballots (one per field slot) restricted to the row's 16 lanes, and a lane stores its four values, -1 behind the sentinel, as 16
// bytes: 1 KiB of four finished rows per instruction.  Otherwise lane = field, one row per step.
template <bool QUAD>
__global__ void __launch_bounds__(64) k_compact_rows_decode_tile(const uint32_t *__restrict__ data, uint64_t N, uint32_t K, uint32_t bits,
                                                                 uint32_t SD, uint32_t sdmagic, uint64_t m, const uint64_t *__restrict__ nodes,
                                                                 int32_t *__restrict__ out, uint32_t *__restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) uint32_t crec[];  // 64 records of SD | 1 dwords (+ 1 spill dword)
    const uint32_t lane = lane_id();
    const uint64_t ntiles = (m + 63u) / 64u;
    const uint32_t SD1 = SD | 1u;
    const bool blockwise = !nodes && (SD & 3u) == 0u;
    const uint32_t CPR = SD >> 2, cmagic = tile_magic(CPR);
    const uint32_t mask = bits >= 32u ? 0xffffffffu : (1u << bits) - 1u;
    const uint32_t n32 = (uint32_t)N;
    uint4 pv[16];
    uint64_t tile = blockIdx.x;
    if (blockwise && tile < ntiles)
        tile_fetch16<16>((const uint4 *)(data + tile * 64u * SD), (uint32_t)(m - tile * 64u < 64u ? m - tile * 64u : 64u) * CPR, pv);
    for (; tile < ntiles; tile += gridDim.x) {
        const uint64_t w0 = tile * 64u;
        const uint32_t nrows = (uint32_t)(m - w0 < 64u ? m - w0 : 64u);
        if (blockwise) {
            const uint32_t total = nrows * CPR;
            const uint32_t step_rows = tile_div(64u, CPR, cmagic), step_w = 64u - step_rows * CPR;
            TileCursor c = tile_cursor(CPR, cmagic);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if ((uint32_t)i * 64u < total) {
                    if ((uint32_t)i * 64u + lane < total) {
                        uint32_t *d = crec + c.row * SD1 + 4u * c.w;
                        d[0] = pv[i].x; d[1] = pv[i].y; d[2] = pv[i].z; d[3] = pv[i].w;
                    }
                    tile_advance(c, CPR, step_rows, step_w);
                }
            }
            const uint64_t nt = tile + gridDim.x;
            if (nt < ntiles)
                tile_fetch16<16>((const uint4 *)(data + nt * 64u * SD), (uint32_t)(m - nt * 64u < 64u ? m - nt * 64u : 64u) * CPR, pv);
        } else if (nodes) {
            const uint64_t row = lane < nrows ? nodes[w0 + lane] : 0ull;
            for (uint32_t w = 0; w < SD; w++) crec[lane * SD1 + w] = lane < nrows ? data[row * SD + w] : 0u;
        } else {
            const uint32_t *src = data + w0 * SD;
            const uint32_t total = nrows * SD;
            const uint32_t step_rows = tile_div(64u, SD, sdmagic), step_w = 64u - step_rows * SD;
            TileCursor c = tile_cursor(SD, sdmagic);
            for (uint32_t f = lane; f < total; f += 64u) {
                crec[c.row * SD1 + c.w] = src[f];
                tile_advance(c, SD, step_rows, step_w);
            }
        }
        wave_lds_sync();
        if (QUAD) {
            const uint32_t g = lane >> 4, q4 = (lane & 15u) * 4u, gsh = g * 16u;
            uint32_t dw[4], sh[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {  // (the second dword of a row's last field may be the record's padding dword: masked out)
                const uint32_t f = q4 + (uint32_t)k < K ? q4 + (uint32_t)k : 0u;
                dw[k] = (f * bits) >> 5;
                sh[k] = (f * bits) & 31u;
            }
            for (uint32_t r0 = 0; r0 < nrows; r0 += 4u) {
                const uint32_t rr = r0 + g;
                const uint32_t *rec = crec + rr * SD1;
                uint32_t v[4], p[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t *q = rec + dw[k];
                    v[k] = __builtin_amdgcn_alignbit(q[1], q[0], sh[k]) & mask;
                    const uint64_t bal = ballot(q4 + (uint32_t)k < K && v[k] == n32);
                    const uint32_t mk = (uint32_t)(bal >> gsh) & 0xffffu;       // the sentinel flags of slot k in this row's 16 lanes
                    p[k] = ((uint32_t)__ffs((int)mk) - 1u) * 4u + (uint32_t)k;  // (none: wraps to >= 2^32 - 4)
                }
                uint32_t n = min(min(p[0], p[1]), min(p[2], p[3]));
                n = n < K ? n : K;
                if (rr < nrows && q4 < K) {
                    int4 o;
                    o.x = q4 + 0u < n ? (int32_t)v[0] : -1; o.y = q4 + 1u < n ? (int32_t)v[1] : -1;
                    o.z = q4 + 2u < n ? (int32_t)v[2] : -1; o.w = q4 + 3u < n ? (int32_t)v[3] : -1;
                    *(int4 *)(out + (w0 + rr) * K + q4) = o;
                    if (counts && q4 == 0u) counts[w0 + rr] = n;
                }
            }
        } else {
            const uint32_t pos = (lane < K ? lane : 0u) * bits, dw = pos >> 5, sh = pos & 31u;
            uint32_t my_n = 0;
            for (uint32_t rr = 0; rr < nrows; rr++) {
                const uint32_t *rw = crec + rr * SD1 + dw;
                const uint32_t v = __builtin_amdgcn_alignbit(rw[1], rw[0], sh) & mask;
                const uint64_t endm = ballot(lane < K && v == n32);
                const uint32_t n = endm ? ff1(endm) : K;
                if (lane < K) out[(w0 + rr) * K + lane] = lane < n ? (int32_t)v : -1;
                my_n = lane == rr ? n : my_n;
            }
            if (counts && lane < nrows) counts[w0 + lane] = my_n;
        }
        wave_lds_sync();  // (the next tile's records go to the same LDS)
    }
}

// the same for rows of any width (K > 64: NSG128 / NSG256 graphs): the wavefront walks the row 64 values at a time; the
// row image lives in dynamic LDS (stride bytes + one spill word)
__global__ void __launch_bounds__(64) k_compact_rows_encode_wide(const int32_t *rows, uint64_t N, uint32_t K, uint32_t bits,
                                                                 uint32_t stride, uint8_t *out, uint32_t *err) {
    extern __shared__ uint32_t wimg[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nw = stride / 4u + 2u;
    for (uint64_t row = blockIdx.x; row < N; row += gridDim.x) {
        for (uint32_t t = lane; t < nw; t += 64) wimg[t] = 0;
        uint32_t n = K;  // edges before the first -1
        for (uint32_t j0 = 0; j0 < K && n == K; j0 += 64) {
            const uint32_t j = j0 + lane;
            const uint64_t endm = __ballot(j < K && rows[row * K + j] == -1);
            if (endm) n = j0 + (uint32_t)__builtin_ctzll(endm);
        }
        __syncthreads();
        bool bad = false;
        for (uint32_t j = lane; j <= n && j < K; j += 64) {
            const int32_t e = j < n ? rows[row * K + j] : 0;
            bad |= j < n && (e < 0 || (uint64_t)e >= N);
            const uint64_t v = j == n ? N : (uint64_t)(uint32_t)e;  // value n is the sentinel N (:31-36)
            const uint32_t pos = j * bits;
            const uint64_t sh = v << (pos & 31);
            atomicOr(&wimg[pos >> 5], (uint32_t)sh);
            if ((pos & 31) + bits > 32) atomicOr(&wimg[(pos >> 5) + 1], (uint32_t)(sh >> 32));
        }
        if (__ballot(bad) && lane == 0) *(volatile uint32_t *)err = 1u;  // (every writer stores the same value: device or pinned host memory)
        __syncthreads();
        const uint8_t *b = (const uint8_t *)wimg;
        for (uint32_t t = lane; t < stride; t += 64) out[row * stride + t] = b[t];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(64) k_compact_rows_decode_wide(const uint8_t *data, uint64_t N, uint32_t K, uint32_t bits,
                                                                 uint32_t stride, uint64_t m, const uint64_t *nodes,
                                                                 int32_t *out, uint32_t *counts) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t mask = (1ull << bits) - 1ull;
    for (uint64_t q = blockIdx.x; q < m; q += gridDim.x) {
        const uint8_t *base = data + (nodes ? nodes[q] : q) * stride;
        uint32_t n = K;
        for (uint32_t j0 = 0; j0 < K; j0 += 64) {
            const uint32_t j = j0 + lane;
            uint64_t v = ~0ull;
            if (j < K) {
                const uint32_t pos = j * bits;
                const uintptr_t a = (uintptr_t)(base + (pos >> 3));
                const uint32_t *w = (const uint32_t *)(a & ~(uintptr_t)3);
                const uint32_t sh = (uint32_t)(a & 3u) * 8u + (pos & 7u);
                v = ((((uint64_t)w[1] << 32) | w[0]) >> sh) & mask;
            }
            if (n == K) {  // the row ends at the sentinel N (:43-49)
                const uint64_t endm = __ballot(j < K && v == N);
                if (endm) n = j0 + (uint32_t)__builtin_ctzll(endm);
            }
            if (j < K) out[q * K + j] = j < n ? (int32_t)v : -1;
        }
        if (lane == 0) counts[q] = n;
    }
}

}  // namespace

struct vidc_compact {
    int device = 0;
    uint64_t N = 0;
    uint32_t K = 0, bits = 0, stride = 0;
    DevBuf<uint8_t> d_data;
};

extern "C" {

int vidc_compact_rows_encode(vidc_ctx *ctx, uint64_t N, uint32_t K, const int32_t *d_rows, vidc_compact **out) {
    if (!ctx || !out || (N && !d_rows)) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (K == 0 || K > 4096) { set_error("compact rows: K=%u unsupported (1..4096)", K); return VIDC_ERR_UNSUPPORTED; }
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_compact> c(new vidc_compact());
    c->device = ctx->device;
    c->N = N;
    c->K = K;
    c->bits = (uint32_t)vidc_packed_bits_for(N);         // while((1 << bits) < N + 1) bits++, altid_impl.cpp:22-23
    c->stride = (K * c->bits + 7) / 8;                   // :24
    VIDC_TRY(c->d_data.alloc(N * c->stride + 8, ctx->dpool));
    Scratch s_err;
    VIDC_TRY(s_err.get(ctx, 4));
    VIDC_HIP(hipMemsetAsync(s_err.p, 0, 4, ctx->stream));
    const bool tile = K <= 64 && (c->stride & 3u) == 0u;  // (rows_tile.h kernels: they write every byte of every row)
    if (tile) VIDC_HIP(hipMemsetAsync(c->d_data.p + N * c->stride, 0, 8, ctx->stream));
    else VIDC_HIP(hipMemsetAsync(c->d_data.p, 0, N * c->stride + 8, ctx->stream));
    if (N) {
        VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        uint32_t grid = (uint32_t)std::min<uint64_t>(N, (uint64_t)ctx->num_cu * 64);
        if (tile) {
            const uint32_t SD = c->stride / 4u, vec = ((uintptr_t)d_rows & 15u) == 0 ? 1u : 0u;
            const dim3 tgrid((uint32_t)((N + 63) / 64));
#define VIDC_COMPACT_ENC(KP, FULL)                                                                                                       \
    hipLaunchKernelGGL((k_compact_rows_encode_tile<KP, FULL>), tgrid, dim3(64),                                                              \
                       4u * std::max<size_t>((FULL && vec) ? dev::RowsTile<KP>::DWORDS_PF : dev::RowsTile<KP>::DWORDS, 64u * (SD | 1u)), ctx->stream, \
                       d_rows, N, K, dev::tile_magic(K), vec, c->bits, \
                       SD, dev::tile_magic(SD), (uint32_t *)c->d_data.p, s_err.as<uint32_t>())
            if (K == 32) VIDC_COMPACT_ENC(32, true);
            else if (K < 32) VIDC_COMPACT_ENC(32, false);
            else if (K == 64) VIDC_COMPACT_ENC(64, true);
            else VIDC_COMPACT_ENC(64, false);
#undef VIDC_COMPACT_ENC
        } else if (K <= 64)
            hipLaunchKernelGGL(k_compact_rows_encode, dim3(grid), dim3(64), 0, ctx->stream, d_rows, N, K, c->bits, c->stride,
                               c->d_data.p, s_err.as<uint32_t>());
        else
            hipLaunchKernelGGL(k_compact_rows_encode_wide, dim3(grid), dim3(64), (c->stride / 4u + 2u) * 4u, ctx->stream, d_rows,
                               N, K, c->bits, c->stride, c->d_data.p, s_err.as<uint32_t>());
        VIDC_HIP(hipGetLastError());
        VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    }
    uint32_t err = 0;
    VIDC_HIP(hipMemcpyAsync(&err, s_err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    if (N) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        ctx->last_kernel_ms = ms;
    }
    if (err) { set_error("compact rows: neighbour id outside [0, N)"); return VIDC_ERR_DOMAIN; }
    *out = c.release();
    return VIDC_OK;
}
void vidc_compact_destroy(vidc_compact *c) { delete c; }
uint32_t vidc_compact_bits(const vidc_compact *c) { return c ? c->bits : 0; }
uint32_t vidc_compact_stride(const vidc_compact *c) { return c ? c->stride : 0; }
uint64_t vidc_compact_size_in_bytes(const vidc_compact *c) { return c ? c->N * c->stride : 0; }

int vidc_compact_rows_decode(vidc_ctx *ctx, const vidc_compact *c, uint64_t m, const uint64_t *nodes, int32_t *d_out,
                             uint32_t *counts) {
    if (!ctx || !c || (m && !d_out)) return VIDC_ERR_INVALID;
    if (!m) return VIDC_OK;
    if (!nodes && m > c->N) { set_error("nodes == NULL selects nodes 0..m-1: m exceeds the node count"); return VIDC_ERR_INVALID; }
    for (uint64_t i = 0; nodes && i < m; i++)
        if (nodes[i] >= c->N) { set_error("node %llu out of range", (unsigned long long)nodes[i]); return VIDC_ERR_INVALID; }
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_n, s_c;
    Pinned h_io;
    const bool tile = c->K <= 64 && (c->stride & 3u) == 0u;
    if (counts || !tile) VIDC_TRY(s_c.get(ctx, m * 4));
    if (counts || nodes) VIDC_TRY(h_io.get(ctx, m * 8));
    const uint64_t *d_nodes = nullptr;  // nodes == NULL: rows 0..m-1, no index array
    if (nodes) {
        VIDC_TRY(s_n.get(ctx, m * 8));
        std::memcpy(h_io.p, nodes, m * 8);
        VIDC_HIP(hipMemcpyAsync(s_n.p, h_io.p, m * 8, hipMemcpyHostToDevice, ctx->stream));
        d_nodes = s_n.as<uint64_t>();
    }
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (tile) {
        const uint32_t SD = c->stride / 4u;
        const size_t dyn = ((size_t)64 * (SD | 1u) + 1u) * 4u;
        const dim3 tgrid(dev::tile_grid(ctx->num_cu, (m + 63) / 64, (uint32_t)std::min<size_t>(16, (150u << 10) / dyn)));
        uint32_t *d_cnt = counts ? s_c.as<uint32_t>() : nullptr;
        if ((c->K & 3u) == 0u && ((uintptr_t)d_out & 15u) == 0)
            hipLaunchKernelGGL(k_compact_rows_decode_tile<true>, tgrid, dim3(64), dyn, ctx->stream, (const uint32_t *)c->d_data.p, c->N, c->K,
                               c->bits, SD, dev::tile_magic(SD), m, d_nodes, d_out, d_cnt);
        else
            hipLaunchKernelGGL(k_compact_rows_decode_tile<false>, tgrid, dim3(64), dyn, ctx->stream, (const uint32_t *)c->d_data.p, c->N, c->K,
                               c->bits, SD, dev::tile_magic(SD), m, d_nodes, d_out, d_cnt);
    } else if (c->K <= 64)
        hipLaunchKernelGGL(k_compact_rows_decode, dim3((uint32_t)std::min<uint64_t>((m + 3) / 4, (uint64_t)ctx->num_cu * 256)), dim3(64), 0, ctx->stream,
                           c->d_data.p, c->N, c->K, c->bits, c->stride, m, d_nodes, d_out, s_c.as<uint32_t>());
    else
        hipLaunchKernelGGL(k_compact_rows_decode_wide, dim3((uint32_t)std::min<uint64_t>(m, (uint64_t)ctx->num_cu * 256)), dim3(64), 0,
                           ctx->stream, c->d_data.p, c->N, c->K, c->bits, c->stride, m, d_nodes, d_out, s_c.as<uint32_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    if (counts) VIDC_HIP(hipMemcpyAsync(h_io.p, s_c.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    if (counts) std::memcpy(counts, h_io.p, m * 4);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    return VIDC_OK;
}

int vidc_compact_export_row(vidc_ctx *ctx, const vidc_compact *c, uint64_t node, uint8_t *bytes, size_t cap) {
    if (!ctx || !c || node >= c->N || cap < c->stride) return VIDC_ERR_INVALID;
    return vidc_copy_d2h(ctx, bytes, c->d_data.p + node * c->stride, c->stride);
}

int vidc_packed_bits_for(uint64_t ntotal) {  // custom_invlists_impl.cpp:68-70
    int bits = 0;
    while (bits < 64 && (1ull << bits) < ntotal + 1) bits++;
    return bits;
}

// geometry, offsets / word offsets on the device, chunk table (shared by encode and import).  Nothing is waited for here: the staging
// block and the scan state live in `keep` until the caller has synchronised.
struct PackedSetupKeep {
    Pinned h_up;
    Scratch s_state;
};
static int packed_setup(vidc_ctx *ctx, vidc_packed *p, uint64_t nlist, const uint64_t *offsets, int bits, PackedSetupKeep &keep) {
    p->device = ctx->device;
    p->nlist = nlist;
    p->bits = bits;
    // the offsets go to the device first (pinned staging block, asynchronous copy): the host's own pass over them -- mirror, geometry --
    // runs while they cross PCIe (round 6: it used to run in front of the copy, 25 us of a 65 536-list call before the GPU saw anything)
    VIDC_TRY(keep.h_up.get(ctx, (nlist + 1) * 8 + 16));
    uint64_t *h64 = keep.h_up.as<uint64_t>();
    if (nlist) std::memcpy(h64, offsets, (nlist + 1) * 8);
    else h64[0] = 0;
    VIDC_TRY(p->d_offsets.alloc(nlist + 1, ctx->dpool));
    VIDC_HIP(hipMemcpyAsync(p->d_offsets.p, h64, (nlist + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    p->offsets = vec_pool<uint64_t>().take(nlist + 1);
    p->offsets.assign(h64, h64 + nlist + 1);
    p->ntotal = p->offsets[nlist];
    const LengthsPass lp = lengths_pass(p->offsets.data(), nlist, 9u, (uint32_t)bits);  // (four lists per instruction where the host can)
    static_assert(CHUNK_IDS == 512, "the pass above counts chunks of 2^9 ids");
    if (!lp.wide) { p->compressed_bytes += lp.bytes; p->total_words += lp.words; p->nchunks += lp.nchunks; p->max_list = std::max(p->max_list, lp.max_n); }
    else
    for (uint64_t l = 0; l < nlist; l++) {  // (a list of 2^32 ids or more, or offsets that decrease: list by list)
        if (p->offsets[l + 1] < p->offsets[l]) { set_error("offsets not monotone"); return VIDC_ERR_INVALID; }
        uint64_t n = p->offsets[l + 1] - p->offsets[l];
        p->compressed_bytes += (n * bits + 7) / 8;              // ids_all[list_no].resize((ls*bits+7)/8), :80
        p->total_words += (n * bits + 63) / 64 + 1;             // +1: read_bits may touch the next word
        p->nchunks += (n + CHUNK_IDS - 1) / CHUNK_IDS;
        p->max_list = std::max(p->max_list, n);
    }
    // word offsets, chunk table and padding words on the device (k_packed_table)
    VIDC_TRY(p->d_word_off.alloc(nlist + 1, ctx->dpool));
    VIDC_TRY(p->d_words.alloc(p->total_words ? p->total_words : 1, ctx->dpool));
    VIDC_TRY(p->d_chunks.alloc(p->nchunks ? p->nchunks : 1, ctx->dpool));
    // lists per thread of k_packed_table (256 threads per tile): an object of up to 4095 lists is ONE tile (no chained scan, no cleared
    // state), larger ones aim at ~1024 tiles -- 65 536 lists ran as 17 tiles of 4096 on 17 of the 256 CUs (16.3 us of a 102 us
    // encode + decode of 16 M ids), 2^20 lists as 257
    const uint32_t per = nlist + 1 <= 4096u ? (uint32_t)((nlist + 1 + 255u) / 256u)
                                            : (uint32_t)std::min<uint64_t>(16u, std::max<uint64_t>(1u, (nlist + 1 + 262143u) / 262144u));
    const uint32_t ntiles = (uint32_t)((nlist + 1 + 256u * per - 1u) / (256u * per));
    VIDC_TRY(keep.s_state.get(ctx, ((size_t)2 * ntiles + 1) * 8));
    // (the device work of the set-up is part of the encode's kernel time: the caller reads ev0 .. ev1)
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (ntiles > 1u) VIDC_HIP(hipMemsetAsync(keep.s_state.p, 0, ((size_t)2 * ntiles + 1) * 8, ctx->stream));
    hipLaunchKernelGGL(k_packed_table, dim3(ntiles), dim3(256), 0, ctx->stream, p->d_offsets.p, (uint32_t)nlist, (uint32_t)bits,
                       keep.s_state.as<unsigned long long>(), p->d_chunks.p, p->d_word_off.p, p->d_words.p, per);
    VIDC_HIP(hipGetLastError());
    return VIDC_OK;
}

int vidc_packed_encode(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids, int bits,
                       vidc_packed **out) {
    if (!ctx || !out || (nlist && !offsets) || bits < 0 || bits > 64) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (nlist >= 0xffffffffull) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_packed> p(new vidc_packed());
    PackedSetupKeep keep;
    struct SyncOnExit {  // (an early return must not release the staging block of copies still in flight)
        vidc_ctx *c;
        bool armed = true;
        ~SyncOnExit() { if (armed) (void)vidc::vidc_stream_wait(c->stream); }
    } guard{ctx};
    VIDC_TRY(packed_setup(ctx, p.get(), nlist, offsets, bits, keep));
    if (p->ntotal && !d_ids) return VIDC_ERR_INVALID;
    // the "an id does not fit" flag is stored by the kernels into pinned memory (no copy engine behind the last kernel)
    Pinned h_err;
    VIDC_TRY(h_err.get(ctx, 64));
    volatile uint32_t *err_flag = h_err.as<uint32_t>();
    *err_flag = 0u;
    if (p->total_words) {
        uint32_t grid = (uint32_t)std::min<uint64_t>(p->nchunks, (uint64_t)ctx->num_cu * 256);
        // ids must fit the field (FAISS_THROW_IF_NOT(ids_in[i] >= 0 && ids_in[i] < ntotal), :87)
        uint64_t limit = ~0ull;
        if (p->nchunks && p->max_list <= 256)
            hipLaunchKernelGGL(k_packed_encode<4>, dim3(grid), dim3(64), 0, ctx->stream, d_ids, p->d_offsets.p,
                               p->d_word_off.p, p->d_chunks.p, p->nchunks, (uint32_t)bits, limit, p->d_words.p,
                               h_err.as<uint32_t>());
        else if (p->nchunks)
            hipLaunchKernelGGL(k_packed_encode<CHUNK_IDS / 64>, dim3(grid), dim3(64), 0, ctx->stream, d_ids, p->d_offsets.p,
                               p->d_word_off.p, p->d_chunks.p, p->nchunks, (uint32_t)bits, limit, p->d_words.p,
                               h_err.as<uint32_t>());
        VIDC_HIP(hipGetLastError());
    }
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    guard.armed = false;
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        ctx->last_kernel_ms = ms;
    }
    const uint32_t err = *err_flag;
    if (err) {
        set_error("packed bits: an id does not fit %d bits (reference: FAISS_THROW_IF_NOT(ids_in[i] >= 0 && "
                  "ids_in[i] < ntotal), custom_invlists_impl.cpp:87)", bits);
        return VIDC_ERR_DOMAIN;
    }
    *out = p.release();
    return VIDC_OK;
}

// ---- flat image of the object (the reference keeps compressed lists in memory only): {offsets, bits, words}, words =
// the device layout (every list starts on a 64-bit word, one padding word follows it)
uint64_t vidc_packed_total_words(const vidc_packed *p) { return p ? p->total_words : 0; }
int vidc_packed_export_all(vidc_ctx *ctx, const vidc_packed *p, uint64_t *words, size_t cap) {
    if (!ctx || !p || (p->total_words && !words)) return VIDC_ERR_INVALID;
    if (p->total_words > cap) { set_error("export buffer too small"); return VIDC_ERR_INVALID; }
    return vidc_copy_d2h(ctx, words, p->d_words.p, p->total_words * 8);
}
int vidc_packed_import(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, int bits, const uint64_t *words,
                       uint64_t nwords, vidc_packed **out) {
    if (!ctx || !out || (nlist && !offsets) || bits < 0 || bits > 64) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (nlist >= 0xffffffffull) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_packed> p(new vidc_packed());
    PackedSetupKeep keep;
    {
        const int st = packed_setup(ctx, p.get(), nlist, offsets, bits, keep);
        VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));  // (the staging block / scan state of the set-up)
        VIDC_TRY(st);
    }
    if (nwords != p->total_words || (nwords && !words)) {
        set_error("packed import: %llu words given, the offsets need %llu", (unsigned long long)nwords,
                  (unsigned long long)p->total_words);
        return VIDC_ERR_INVALID;
    }
    if (nwords) VIDC_TRY(vidc_copy_h2d(ctx, p->d_words.p, words, nwords * 8));
    *out = p.release();
    return VIDC_OK;
}

void vidc_packed_destroy(vidc_packed *p) { delete p; }
uint64_t vidc_packed_compressed_bytes(const vidc_packed *p) { return p ? p->compressed_bytes : 0; }
int vidc_packed_bits(const vidc_packed *p) { return p ? p->bits : 0; }

int vidc_packed_decode_all(vidc_ctx *ctx, const vidc_packed *p, uint64_t *d_out) {
    if (!ctx || !p || (p->ntotal && !d_out)) return VIDC_ERR_INVALID;
    if (!p->ntotal) return VIDC_OK;
    VIDC_HIP(hipSetDevice(ctx->device));
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    uint32_t grid = (uint32_t)std::min<uint64_t>(p->nchunks, (uint64_t)ctx->num_cu * 256);
    if (p->max_list <= 256)
        hipLaunchKernelGGL(k_packed_decode<4>, dim3(grid), dim3(64), 0, ctx->stream, p->d_words.p, p->d_offsets.p,
                           p->d_word_off.p, p->d_chunks.p, p->nchunks, (uint32_t)p->bits, d_out);
    else
        hipLaunchKernelGGL(k_packed_decode<CHUNK_IDS / 64>, dim3(grid), dim3(64), 0, ctx->stream, p->d_words.p,
                           p->d_offsets.p, p->d_word_off.p, p->d_chunks.p, p->nchunks, (uint32_t)p->bits, d_out);
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    return VIDC_OK;
}

int vidc_packed_decode_lists(vidc_ctx *ctx, const vidc_packed *p, uint64_t m, const uint64_t *list_nos, uint64_t *d_out,
                             uint64_t *out_offsets) {
    if (!ctx || !p || (m && !list_nos) || !out_offsets) return VIDC_ERR_INVALID;
    out_offsets[0] = 0;
    std::vector<PackedItem> items;
    for (uint64_t i = 0; i < m; i++) {
        if (list_nos[i] >= p->nlist) { set_error("list number out of range"); return VIDC_ERR_INVALID; }
        const uint64_t n = p->offsets[list_nos[i] + 1] - p->offsets[list_nos[i]];
        out_offsets[i + 1] = out_offsets[i] + n;
        for (uint64_t s = 0; s < n; s += CHUNK_IDS) items.push_back(PackedItem{(uint32_t)i, (uint32_t)s});
    }
    ctx->last_kernel_ms = 0;
    if (items.empty()) return VIDC_OK;
    if (!d_out) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_l, s_o, s_i;
    VIDC_TRY(s_l.get(ctx, m * 8)); VIDC_TRY(s_o.get(ctx, (m + 1) * 8)); VIDC_TRY(s_i.get(ctx, items.size() * sizeof(PackedItem)));
    VIDC_HIP(hipMemcpyAsync(s_l.p, list_nos, m * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(s_o.p, out_offsets, (m + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(s_i.p, items.data(), items.size() * sizeof(PackedItem), hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    const uint32_t grid = (uint32_t)std::min<uint64_t>(items.size(), (uint64_t)ctx->num_cu * 256);
    hipLaunchKernelGGL(k_packed_decode_lists, dim3(grid), dim3(64), 0, ctx->stream, p->d_words.p, p->d_offsets.p,
                       p->d_word_off.p, s_l.as<uint64_t>(), s_o.as<uint64_t>(), s_i.as<PackedItem>(), (uint64_t)items.size(),
                       (uint32_t)p->bits, d_out);
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));  // the staging vectors above are pageable host memory
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    return VIDC_OK;
}

int vidc_packed_decode_gather(vidc_ctx *ctx, const vidc_packed *p, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                              const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out) {
    if (!ctx || !p) return VIDC_ERR_INVALID;
    return vidc_decode_gather_impl(ctx, p->nlist, m, list_nos, n_items, item_slot, item_off, ids_out,
                                   [&](uint64_t l) { return p->offsets[l + 1] - p->offsets[l]; },
                                   [&](uint64_t *d, uint64_t *lo) { return vidc_packed_decode_lists(ctx, p, m, list_nos, d, lo); });
}

int vidc_packed_get(vidc_ctx *ctx, const vidc_packed *p, uint64_t m, const uint64_t *list_nos, const uint64_t *offs,
                    int64_t *ids_out) {
    if (!ctx || !p || (m && (!list_nos || !offs || !ids_out))) return VIDC_ERR_INVALID;
    if (!m) return VIDC_OK;
    for (uint64_t i = 0; i < m; i++) {
        if (list_nos[i] >= p->nlist || offs[i] >= p->offsets[list_nos[i] + 1] - p->offsets[list_nos[i]]) {
            set_error("packed get: (list %llu, offset %llu) out of range", (unsigned long long)list_nos[i],
                      (unsigned long long)offs[i]);
            return VIDC_ERR_INVALID;
        }
    }
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_l, s_o, s_r;
    VIDC_TRY(s_l.get(ctx, m * 8)); VIDC_TRY(s_o.get(ctx, m * 8)); VIDC_TRY(s_r.get(ctx, m * 8));
    VIDC_HIP(hipMemcpyAsync(s_l.p, list_nos, m * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(s_o.p, offs, m * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_packed_get, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, ctx->stream, p->d_words.p,
                       p->d_word_off.p, (uint32_t)p->bits, m, s_l.as<uint64_t>(), s_o.as<uint64_t>(),
                       s_r.as<int64_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipMemcpyAsync(ids_out, s_r.p, m * 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    ctx->d2h_bytes += m * 8;
    return VIDC_OK;
}

int vidc_packed_export(vidc_ctx *ctx, const vidc_packed *p, uint64_t list_no, uint8_t *bytes, size_t cap) {
    if (!ctx || !p || list_no >= p->nlist) return VIDC_ERR_INVALID;
    uint64_t n = p->offsets[list_no + 1] - p->offsets[list_no];
    uint64_t nb = (n * p->bits + 7) / 8;
    if (nb > cap) { set_error("export buffer too small"); return VIDC_ERR_INVALID; }
    uint64_t w0 = 0;  // (the word offsets live on the device)
    VIDC_TRY(vidc_copy_d2h(ctx, &w0, p->d_word_off.p + list_no, 8));
    return vidc_copy_d2h(ctx, bytes, p->d_words.p + w0, nb);
}

}  // extern "C"
