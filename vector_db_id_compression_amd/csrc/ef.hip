// ef.hip -- Elias-Fano coded lists (CompressedIDInvertedListsEliasFano,
// custom_invlists_impl.cpp:229-339; EliasFanoNSGGraph, altid_impl.cpp:53-101) following the
// succinct::elias_fano layout described in elias_fano.hpp:22-57:
//   l = (m && u/m) ? msb(u/m) : 0          u = universe (= max id), m = count              (:28)
//   low  stream: m*l bits, element i's low l bits at bit i*l (LSB-first 64-bit words)        (:40-42)
//   high stream: (m+1) + (u>>l) + 1 bits, bit (x>>l)+i set for the i-th smallest element x   (:29,43)
// Bandwidth-bound kernels: low words have one owner thread each (no atomics); high bits are set
// with one atomicOr per element (ascending ids -> neighbouring lanes hit the same 64-bit word);
// bulk decode walks the high words with a wavefront prefix-scan of popcounts.
#include <algorithm>
#include <memory>
#include <mutex>
#include <numeric>

#include "bits.h"
#include "chunks.h"
#include "common.h"
#include "scan.h"
#include "rows_csr.h"
#include "rows_tile.h"
#include "wave.h"

using namespace vidc;
using namespace vidc::dev;

struct vidc_ef {
    int device = 0;
    uint64_t nlist = 0, ntotal = 0;
    uint64_t total_bits = 0;  // sum of low + high stream lengths in bits
    bool rows = false;        // built from graph rows (offsets live on the device until somebody asks)
    uint32_t K = 0;
    // graph objects of rows up to 64 edges: a fixed-stride arena (a_lw low + a_hw high words per row) and {universe, n | l << 8}
    // per row (see k_ef_rows_encode_tile).  The CSR members below stay empty until an entry point that works on the per-list
    // streams (export, save, decode_lists, get, decode_all) asks for them: ef_ensure_csr (guarded by mu).
    bool arena = false;
    mutable bool csr_ready = true;
    uint32_t a_lw = 0, a_hw = 0;
    DevBuf<uint64_t> d_arena;
    DevBuf<uint2> d_rmeta;
    // Host mirrors of the per-list geometry, filled lazily: the device arrays are authoritative (with 10^6 graph
    // nodes the PCIe crossings of these arrays used to cost 10x the kernels).  `offsets` is always valid for
    // objects built from host offsets.
    mutable std::vector<uint64_t> offsets, low_off, high_off, universe, high_nbits;
    mutable std::vector<uint32_t> lbits;
    mutable bool offsets_host = false, meta_host = false;
    mutable std::mutex mu;  // guards the lazy host mirrors
    DevBuf<uint64_t> d_offsets, d_low_off, d_high_off, d_low, d_high, d_universe;
    DevBuf<uint32_t> d_lbits, d_perm;
    DevBuf<Chunk> d_chunks;
    uint64_t nchunks = 0;
    // select directory (the role of succinct's darray1): ones before every batch of 64 high words, so that bulk
    // decode and select work per (list, batch) instead of scanning a list from its start
    DevBuf<uint64_t> d_batch_off;       // nlist + 1
    DevBuf<uint32_t> d_hrank;           // [total batches]
    DevBuf<Chunk> d_batches;            // (list, batch number) work items of decode_all
    uint64_t nbatches = 0;
    bool has_perm = false;
    // decode_all acceleration records (one per batch), built on the first bulk decode and kept (guarded by mu)
    mutable DevBuf<struct EfRec> d_recs;
    mutable bool recs_ready = false;
    mutable uint32_t recs_max_cnt = 0;  // largest element count of a batch: sizes the decode kernel's LDS table
    uint64_t max_list = 0;              // longest list (0: unknown): an upper bound of that count, known without a read-back
    uint64_t n_low = ~0ull, n_high = ~0ull;  // words of the two streams when the buffers were allocated by a bound (else: d_low.n / d_high.n)
    bool narrow = false;  // every id < 2^32 and every list < 2^30 ids (known from the encoder): 32-bit decode kernel
    ~vidc_ef() {  // the large host arrays go back to the process-wide vector cache (common.h)
        for (auto *v : {&offsets, &low_off, &high_off, &universe, &high_nbits}) vidc::vec_pool<uint64_t>().give(std::move(*v));
        vidc::vec_pool<uint32_t>().give(std::move(lbits));
    }
};

// Everything a wavefront needs to decode one batch of 64 high words, in one 48-byte record: with the CSR arrays
// alone a wavefront walks batch item -> list header (8 arrays) -> directory entry -> high word -> low words, five
// dependent memory round trips for ~10 KiB of payload, and the kernel is latency x occupancy bound.
struct EfRec {
    uint64_t out_pos;   // offsets[l] + done: where the batch's first element goes
    uint64_t low_base;  // word offset of the list's low stream
    uint64_t hw_base;   // word offset of the batch's first high word
    uint32_t done;      // elements of the list before this batch (select directory entry)
    uint32_t m;         // elements of the list (the batch holds those up to the NEXT record's `done`, or up to m in the list's last batch)
    uint32_t nw;        // high words of this batch (<= 64)
    uint32_t b;         // low bits per element
    uint32_t bt;        // batch number inside the list
    uint32_t nb;        // batches of the list
};
// (the record array ends with one unused record: a wavefront loads its record and the next one together)
#define EF_CLEAR_HIGH_BYTES (64ull << 20)  // high streams beyond this are cleared ahead of the chunk kernels (comment at the memset)
#define EF_SINGLE_MAX_CHUNKS 16384u  // chunk records the single-tile geometry kernel writes itself (S1: 2.4 k)
#define EF_ENC_RECS_MAX (1u << 18)  // objects of fewer batches: records written by the encoder, LDS table sized by the longest list

namespace {

struct PrepOut {
    uint64_t max_id;
    uint32_t unsorted;
    uint32_t pad;
};
__host__ __device__ inline int msb64(uint64_t x) { return 63 - __builtin_clzll(x); }
struct EfTotals {
    unsigned long long total_bits;
    unsigned int n_unsorted;
    unsigned int pad;
};

// one wavefront per chunk: max id and "is non-decreasing" (merged per list with atomics; outp zero-initialised)
__global__ void __launch_bounds__(64) k_ef_prep(const uint64_t *ids, const uint64_t *offsets, const Chunk *chunks,
                                                uint64_t nchunks, PrepOut *outp) {
    const uint32_t lane = lane_id();
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const Chunk ch = chunks[c];
        const uint64_t off = offsets[ch.list];
        const uint64_t n = offsets[ch.list + 1] - off;
        const uint32_t nc = (uint32_t)(n - ch.start < CHUNK_IDS ? n - ch.start : CHUNK_IDS);
        uint64_t mx = 0;
        bool uns = false;
        for (uint32_t i = lane; i < nc; i += 64) {
            const uint64_t j = ch.start + i;
            const uint64_t v = ids[off + j];
            if (j) uns |= ids[off + j - 1] > v;
            mx = v > mx ? v : mx;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)mx, o, 64);
            uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(mx >> 32), o, 64);
            uint64_t w = ((uint64_t)hi << 32) | lo;
            mx = w > mx ? w : mx;
        }
        const uint64_t any_uns = ballot(uns);  // all lanes vote (not inside the lane-0 branch)
        if (lane == 0) {
            atomicMax((unsigned long long *)&outp[ch.list].max_id, (unsigned long long)mx);
            if (any_uns) atomicOr(&outp[ch.list].unsorted, 1u);
        }
    }
}

// wave-level bitonic sort of (key, pos) pairs in global memory; np2 = power of two >= n, padded with ~0
__global__ void __launch_bounds__(64) k_ef_sort(const uint64_t *ids, const uint64_t *offsets, const uint32_t *lists,
                                                uint32_t nwork, const uint64_t *key_off, uint64_t *keys,
                                                uint32_t *kpos, uint64_t *sorted_ids, uint32_t *perm) {
    const uint32_t lane = lane_id();
    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        const uint32_t l = lists[wi];
        const uint64_t off = offsets[l];
        const uint32_t n = (uint32_t)(offsets[l + 1] - off);
        uint64_t *k = keys + key_off[wi];
        uint32_t *p = kpos + key_off[wi];
        const uint32_t np2 = (uint32_t)(key_off[wi + 1] - key_off[wi]);
        for (uint32_t j = lane; j < np2; j += 64) {
            k[j] = j < n ? ids[off + j] : ~0ull;
            p[j] = j < n ? j : 0xffffffffu;
        }
        __syncthreads();
        for (uint32_t kk = 2; kk <= np2; kk <<= 1) {
            for (uint32_t j = kk >> 1; j > 0; j >>= 1) {
                for (uint32_t t = lane; t < (np2 >> 1); t += 64) {
                    uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));
                    uint32_t q = i | j;
                    uint64_t x = k[i], y = k[q];
                    uint32_t px = p[i], py = p[q];
                    bool up = ((i & kk) == 0);
                    bool gt = x > y || (x == y && px > py);  // ties by position: std::sort of (id, code ptr) pairs, :336
                    if (gt == up) {
                        k[i] = y; k[q] = x;
                        p[i] = py; p[q] = px;
                    }
                }
                __syncthreads();
            }
        }
        for (uint32_t j = lane; j < n; j += 64) {
            sorted_ids[off + j] = k[j];
            perm[off + j] = p[j];
        }
        __syncthreads();
    }
}

__global__ void k_iota_perm(const uint64_t *offsets, uint32_t nlist, uint64_t ntotal, uint32_t *perm) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ntotal; g += stride) {
        const uint32_t l = find_list(offsets, nlist, g);
        perm[g] = (uint32_t)(g - offsets[l]);
    }
}

// low stream: one wavefront per chunk, one owner lane per 64-bit word (CHUNK_IDS * l is a multiple of 64)
__global__ void __launch_bounds__(64) k_ef_low(const uint64_t *sorted_ids, const uint64_t *offsets,
                                               const uint64_t *low_off, const uint32_t *lbits, const Chunk *chunks,
                                               uint64_t nchunks, uint64_t *low) {
    const uint32_t lane = lane_id();
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const Chunk ch = chunks[c];
        const uint32_t b = lbits[ch.list];
        if (!b) continue;
        const uint64_t off = offsets[ch.list];
        const uint64_t n = offsets[ch.list + 1] - off;
        const uint64_t nc = n - ch.start < CHUNK_IDS ? n - ch.start : CHUNK_IDS;
        const uint64_t keep = (1ull << b) - 1ull;
        const uint64_t w0 = (uint64_t)ch.start * b / 64, w1 = ((uint64_t)(ch.start + nc) * b + 63) / 64;
        uint64_t *dst = low + low_off[ch.list];
        for (uint64_t w = w0 + lane; w < w1; w += 64)
            dst[w] = gather_word<false>(sorted_ids + off, n, w, b, keep, ~0ull, nullptr);
    }
}

// high stream: one wavefront per chunk.  Bit positions (x >> l) + i increase strictly with i, so the chunk covers a
// contiguous bit range that only shares its first and last WORD with neighbouring chunks: the bits are assembled
// in an LDS window (ds_or) and written with plain stores, global atomics only for the two boundary words.
#define EF_WIN_WORDS 128u
__global__ void __launch_bounds__(64) k_ef_high(const uint64_t *sorted_ids, const uint64_t *offsets,
                                                const uint64_t *high_off, const uint32_t *lbits, const Chunk *chunks,
                                                uint64_t nchunks, uint64_t *high) {
    __shared__ unsigned long long win[EF_WIN_WORDS];
    const uint32_t lane = lane_id();
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const Chunk ch = chunks[c];
        const uint32_t b = lbits[ch.list];
        const uint64_t off = offsets[ch.list];
        const uint64_t n = offsets[ch.list + 1] - off;
        const uint32_t nc = (uint32_t)(n - ch.start < CHUNK_IDS ? n - ch.start : CHUNK_IDS);
        unsigned long long *dst = (unsigned long long *)(high + high_off[ch.list]);
        const uint64_t wf = ((sorted_ids[off + ch.start] >> b) + ch.start) >> 6;
        const uint64_t wl = ((sorted_ids[off + ch.start + nc - 1] >> b) + (ch.start + nc - 1)) >> 6;
        uint64_t pos[CHUNK_IDS / 64];
#pragma unroll
        for (uint32_t r = 0; r < CHUNK_IDS / 64; r++) {
            const uint32_t i = lane + 64 * r;
            pos[r] = i < nc ? (sorted_ids[off + ch.start + i] >> b) + (ch.start + i) : ~0ull;
        }
        for (uint64_t wbase = wf; wbase <= wl; wbase += EF_WIN_WORDS) {
            win[lane] = 0;
            win[lane + 64] = 0;
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < CHUNK_IDS / 64; r++) {
                const uint64_t w = pos[r] >> 6;
                if (pos[r] != ~0ull && w >= wbase && w - wbase < EF_WIN_WORDS)
                    atomicOr(&win[w - wbase], 1ull << (pos[r] & 63));
            }
            __syncthreads();
#pragma unroll
            for (uint32_t t = 0; t < 2; t++) {
                const uint32_t k = lane + 64 * t;
                const uint64_t w = wbase + k;
                const unsigned long long v = win[k];
                if (v && w <= wl) {
                    if (w == wf || w == wl) atomicOr(&dst[w], v);
                    else dst[w] = v;
                }
            }
            __syncthreads();
        }
    }
}

// ---- single-pass encoder for ascending lists (the normal case: Faiss lists are in add order, graph rows are sorted
// by k_rows_sorted).  The universe of an ascending list is its last element, so the ids are streamed from HBM once.
// Three launches whatever the number of lists:
//   k_ef_meta      per-list geometry from (size, last id) and the totals of every tile of lists
//   k_ef_offsets   word / batch / chunk offsets (tile prefix + scan inside the tile) and one 64-byte record per chunk
//                  (up to EF_SINGLE_LISTS lists: this kernel alone, it computes the geometry itself)
//   k_ef_lowhigh*  one wavefront per chunk record: both bit streams, the order check, the list's padding word and the
//                  select-directory entries whose batch starts inside the chunk
// Any order violation raises `unsorted` and the caller redoes the object with the three-pass path further down.
// (a tile = the lists of one workgroup: 512 threads with two lists each when that is the whole object, else 256 threads
// with one list each, or four each from 2^18 lists on -- every workgroup of k_ef_offsets sums the totals of the tiles before it)
// ids per chunk record of the single-pass encoder (VIDC_EF_CHUNK: build-time experiment hook)
#ifndef VIDC_EF_CHUNK
#define VIDC_EF_CHUNK 512
#endif
constexpr uint32_t EF_CHUNK = VIDC_EF_CHUNK;
#define EF_SINGLE_LISTS 1024u
#define EF_E1_MAX_LISTS 262144u
struct EfRaw {     // the four per-list counts (k_ef_meta) whose prefix sums are the offsets of a list
    uint64_t lw, hw, nb, cnt;  // low words (incl. the padding word), high words, batches of 64 high words, chunks
};
struct EfTile {    // totals of one tile
    uint64_t lw, hw, nb, cnt, bits, wide;
};
struct EfLimits {  // sizes the host allocated ahead of the geometry kernels (all ~0 / wide_ok: it allocates after them)
    uint64_t low_words, high_words, nbatches;
    uint32_t wide_ok, pad;
};
struct EfSummary {  // what the host needs before it can allocate the streams
    uint64_t low_words, high_words, nbatches, nchunks, total_bits;
    uint32_t wide, unsorted;
};
// Everything a wavefront needs to encode one chunk, in one 64-byte record (one scalar load): with the CSR arrays a
// wavefront walks chunk item -> six per-list arrays -> ids, three dependent round trips for 4 KiB of payload.
struct EfChunkRec {
    uint64_t src;        // index of the chunk's first id in the id array
    uint64_t low_word;   // word offset of the list's low stream
    uint64_t high_word;  // word offset of the list's high stream
    uint64_t batch0;     // index of the list's first select-directory entry
    uint64_t u;          // universe of the list (its last id)
    uint32_t start;      // first id of the chunk inside its list
    uint32_t n;          // ids in the list
    uint32_t list;
    uint32_t b;          // low bits per element
    uint32_t nb;         // directory entries (batches) of the list
    uint32_t pad;
};
// A list of more than EF_BIG_CHUNKS chunks leaves the records behind its first one to k_ef_big_recs: the tile holding the
// longest lists of a Zipf-sized index (S2: 1024 lists of 65 536 ids = 130 000 records) kept its 256 threads busy for 0.4 ms
// while the median tile wrote 1 500 records.
#define EF_BIG_CHUNKS 8u
struct EfBigList {
    uint64_t o0, u, lw, hw, nb, rec0;  // first id, universe, offsets of the low / high words and directory entries, first record
    uint32_t l, m, lb, pad;
};
struct EfListInfo {  // a list of the tile in LDS while the chunk records behind its first one are written
    uint64_t o0, u, lw, hw, nb;  // first id, universe, offsets of the list's low / high words and directory entries
    uint32_t c0, x0;             // chunks / chunks other than first ones of the tile before this list
    uint32_t m, lb;
};

// inclusive prefix sum over the 64 lanes in six DPP adds (row_shr 1 2 4 8, row_bcast 15 31): the shuffle version is
// six LDS-crossbar round trips per value, and these kernels are a few lone wavefronts whose time is their latency
__device__ __forceinline__ uint32_t wave_scan32(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
// the same for values below 2^45 (word / chunk counts, stream bits of a few lists): two limbs of 20 + 25 bits
__device__ __forceinline__ uint64_t wave_scan64(uint64_t v) {
    const uint32_t lo = wave_scan32((uint32_t)v & 0xfffffu), hi = wave_scan32((uint32_t)(v >> 20));
    return ((uint64_t)hi << 20) + lo;
}
// sums of N values over the NW wavefronts of the workgroup, returned to every thread (sh: N * NW words)
template <int N, int NW>
__device__ inline void block_sum(uint64_t (&v)[N], uint64_t *sh) {
#pragma unroll
    for (int k = 0; k < N; k++) v[k] = wave_scan64(v[k]);  // (lane 63 holds the sum)
    __syncthreads();  // (sh may still be read from an earlier call)
    if ((threadIdx.x & 63u) == 63u)
#pragma unroll
        for (int k = 0; k < N; k++) sh[NW * k + (threadIdx.x >> 6)] = v[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        v[k] = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) v[k] += sh[NW * k + w];
    }
}
// exclusive prefixes of N values over the threads of the workgroup (v -> prefix), totals to tot
template <int N, int NW>
__device__ inline void block_exscan(uint64_t (&v)[N], uint64_t (&tot)[N], uint64_t *sh) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint64_t incl[N];
#pragma unroll
    for (int k = 0; k < N; k++) incl[k] = wave_scan64(v[k]);
    __syncthreads();
    if (lane == 63u)
#pragma unroll
        for (int k = 0; k < N; k++) sh[NW * k + wave] = incl[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; k++) {
        uint64_t before = 0;
        tot[k] = 0;
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)NW; w++) {
            const uint64_t x = sh[NW * k + w];
            before += w < wave ? x : 0ull;
            tot[k] += x;
        }
        v[k] = before + incl[k] - v[k];
    }
}
// geometry of this thread's lists (elias_fano.hpp:28-29 per list) -> lbits, universe, raw and the registers o / u / lb / r;
// returns the tile totals.  All loads of a step are issued together: the kernel is two memory round trips long.
template <int E, int NT>
__device__ inline EfTile ef_meta_tile(const uint64_t *ids, const uint64_t *offsets, uint32_t nlist, uint32_t tile,
                                      uint32_t *lbits, uint64_t *universe, EfRaw *raw, uint64_t *sh,
                                      uint64_t (&o)[E + 1], uint64_t (&u)[E], uint32_t (&lb)[E], EfRaw (&r)[E]) {
    const uint32_t base = (tile * NT + threadIdx.x) * E;
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};  // lw, hw, nb, cnt, bits, wide
#pragma unroll
    for (uint32_t j = 0; j <= E; j++) o[j] = offsets[base + j <= nlist ? base + j : nlist];
#pragma unroll
    for (uint32_t j = 0; j < E; j++) u[j] = o[j + 1] > o[j] ? ids[o[j + 1] - 1] : 0ull;
#pragma unroll
    for (uint32_t j = 0; j < E; j++) {
        const uint32_t l = base + j;
        const uint64_t m = o[j + 1] - o[j];
        r[j] = EfRaw{0, 0, 0, 0};
        lb[j] = 0;
        if (m) {  // empty lists have no bitstream object (ef_bitstreams[list_no] stays null, :239-241)
            t[5] |= ((u[j] >> 32) != 0 || m >= (1ull << 30)) ? 1ull : 0ull;  // -> the 64-bit encoder kernel
            // msb(u / m) without the division: with k = msb(u) - msb(m), u / m lies in (2^(k-1), 2^(k+1)) and reaches
            // 2^k exactly when (u >> k) >= m  (u < m: the quotient is 0 and l = 0, elias_fano.hpp:28)
            if (u[j] >= m) {
                const uint32_t k = (uint32_t)(msb64(u[j]) - msb64(m));
                lb[j] = (u[j] >> k) >= m ? k : k - 1u;
            }
            const uint64_t hb = (m + 1) + (u[j] >> lb[j]) + 1;
            t[4] += m * lb[j] + hb;
            r[j].lw = (m * lb[j] + 63) / 64 + 1;  // +1 padding word for read_bits
            r[j].hw = (hb + 63) / 64;
            r[j].nb = (r[j].hw + 63) / 64;
            r[j].cnt = (m + EF_CHUNK - 1) / EF_CHUNK;
        }
        if (l < nlist) {
            lbits[l] = lb[j];
            universe[l] = u[j];
            if (raw) raw[l] = r[j];
        }
        t[0] += r[j].lw; t[1] += r[j].hw; t[2] += r[j].nb; t[3] += r[j].cnt;
    }
    block_sum<6, NT / 64>(t, sh);
    return EfTile{t[0], t[1], t[2], t[3], t[4], t[5]};
}
template <int E>
__global__ void __launch_bounds__(256) k_ef_meta(const uint64_t *ids, const uint64_t *offsets, uint32_t nlist,
                                                 uint32_t *lbits, uint64_t *universe, EfRaw *raw, EfTile *tiles) {
    __shared__ uint64_t sh[24];
    uint64_t o[E + 1], u[E];
    uint32_t lb[E];
    EfRaw r[E];
    const EfTile t = ef_meta_tile<E, 256>(ids, offsets, nlist, blockIdx.x, lbits, universe, raw, sh, o, u, lb, r);
    if (threadIdx.x == 0) tiles[blockIdx.x] = t;
}
__device__ inline EfChunkRec ef_make_rec(uint32_t l, uint64_t c, uint64_t o0, uint32_t m, uint64_t u, uint32_t lb,
                                         uint64_t lw, uint64_t hw, uint64_t nb0) {
    const uint64_t nhw = ((uint64_t)m + 1 + (u >> lb) + 1 + 63) / 64;
    EfChunkRec rc;
    rc.src = o0 + c * EF_CHUNK;
    rc.low_word = lw; rc.high_word = hw; rc.batch0 = nb0; rc.u = u;
    rc.start = (uint32_t)(c * EF_CHUNK);
    rc.n = m;
    rc.list = l;
    rc.b = lb;
    rc.nb = (uint32_t)((nhw + 63) / 64);
    rc.pad = 0;
    return rc;
}
// SINGLE: the whole object is one tile (grid of 1); the geometry is computed here and `raw` / `tiles` are not used
template <bool SINGLE, int E, int NT>
__global__ void __launch_bounds__(NT) k_ef_offsets(const uint64_t *ids, const uint64_t *offsets, uint32_t nlist,
                                                    uint32_t *lbits, uint64_t *universe, const EfRaw *raw,
                                                    const EfTile *tiles, uint32_t ntiles, uint64_t *low_off,
                                                    uint64_t *high_off, uint64_t *batch_off, EfChunkRec *recs,
                                                    EfSummary *sum, EfBigList *big, uint32_t *nbig, EfLimits lim, uint32_t *abort) {
    constexpr uint32_t TILE = NT * E;
    // lists of more than BIGC chunks leave their records to k_ef_big_recs (S2: the tile of the 1024 longest lists wrote 130 000 records
    // with 256 threads).  A single-tile object has no such launch: its <= NT * E lists are balanced by the binary search below whatever
    // their sizes, and the call saves a fill, a launch and the gap in front of it (S1-sized calls: 24.7 -> ~19 us of kernels)
    constexpr uint32_t BIGC = SINGLE ? 0xffffffffu : (uint32_t)EF_BIG_CHUNKS;
    __shared__ uint64_t sh[6 * (NT / 64)];
    __shared__ EfListInfo info[TILE];
    const uint32_t tile = blockIdx.x, t = threadIdx.x;
    const uint32_t base = tile * TILE + t * E;
    uint64_t P[6] = {0, 0, 0, 0, 0, 0};  // lw, hw, nb, cnt of the tiles before this one; bits, wide of all tiles
    uint64_t o[E + 1], u[E];
    uint32_t lb[E];
    EfRaw r[E];
    if (SINGLE) {
        const EfTile mine = ef_meta_tile<E, NT>(ids, offsets, nlist, 0u, lbits, universe, nullptr, sh, o, u, lb, r);
        P[4] = mine.bits; P[5] = mine.wide;
    } else {
#pragma unroll
        for (uint32_t j = 0; j <= E; j++) o[j] = offsets[base + j <= nlist ? base + j : nlist];
#pragma unroll
        for (uint32_t j = 0; j < E; j++) {
            const uint32_t l = base + j < nlist ? base + j : nlist - 1u;
            r[j] = raw[l]; u[j] = universe[l]; lb[j] = lbits[l];
            if (base + j >= nlist) r[j] = EfRaw{0, 0, 0, 0};
        }
        const bool last_tile = tile + 1u == ntiles;
        for (uint32_t i = t; i < (last_tile ? ntiles : tile); i += NT) {
            const EfTile ti = tiles[i];
            if (i < tile) { P[0] += ti.lw; P[1] += ti.hw; P[2] += ti.nb; P[3] += ti.cnt; }
            P[4] += ti.bits; P[5] += ti.wide;  // (only the last tile uses them)
        }
        block_sum<6, NT / 64>(P, sh);
    }
    uint64_t a[5] = {0, 0, 0, 0, 0}, tot[5];  // lw, hw, nb, cnt, chunks other than the first one of a list
#pragma unroll
    for (uint32_t j = 0; j < E; j++) {
        a[0] += r[j].lw; a[1] += r[j].hw; a[2] += r[j].nb; a[3] += r[j].cnt;
        a[4] += (r[j].cnt && r[j].cnt <= BIGC) ? r[j].cnt - 1 : 0;
    }
    block_exscan<5, NT / 64>(a, tot, sh);
    a[0] += P[0]; a[1] += P[1]; a[2] += P[2];
    if (nlist == 0 && t == 0) {
        low_off[0] = 0; high_off[0] = 0; batch_off[0] = 0;
        *sum = EfSummary{0, 0, 0, 0, 0, 0u, 0u};
    }
#pragma unroll
    for (uint32_t j = 0; j < E; j++) {
        const uint32_t l = base + j;
        const uint32_t m = (uint32_t)(o[j + 1] - o[j]);
        if (l < nlist) { low_off[l] = a[0]; high_off[l] = a[1]; batch_off[l] = a[2]; }
        if (r[j].cnt) recs[P[3] + a[3]] = ef_make_rec(l, 0, o[j], m, u[j], lb[j], a[0], a[1], a[2]);
        if (r[j].cnt > BIGC) {
            EfBigList bl;
            bl.o0 = o[j]; bl.u = u[j]; bl.lw = a[0]; bl.hw = a[1]; bl.nb = a[2]; bl.rec0 = P[3] + a[3];
            bl.l = l; bl.m = m; bl.lb = lb[j]; bl.pad = 0;
            big[atomicAdd(nbig, 1u)] = bl;
        }
        EfListInfo li;
        li.o0 = o[j]; li.u = u[j]; li.lw = a[0]; li.hw = a[1]; li.nb = a[2]; li.c0 = (uint32_t)a[3]; li.x0 = (uint32_t)a[4];
        li.m = m; li.lb = lb[j];
        info[t * E + j] = li;  // (lists past the end: no chunks, x0 = the tile's count)
        a[0] += r[j].lw; a[1] += r[j].hw; a[2] += r[j].nb; a[3] += r[j].cnt;
        a[4] += (r[j].cnt && r[j].cnt <= BIGC) ? r[j].cnt - 1 : 0;
        if (l + 1u == nlist) {  // (exactly one thread of the last tile): the entries behind the last list, the totals
            low_off[nlist] = a[0]; high_off[nlist] = a[1]; batch_off[nlist] = a[2];
            EfSummary out;
            out.low_words = a[0]; out.high_words = a[1]; out.nbatches = a[2]; out.nchunks = P[3] + a[3];
            out.total_bits = P[4]; out.wide = P[5] ? 1u : 0u; out.unsorted = 0u;
            *sum = out;
            // streams allocated before this kernel ran (ef_encode_fast): the chunk kernels behind it do nothing if they do not fit
            *abort = (a[0] > lim.low_words || a[1] > lim.high_words || a[2] > lim.nbatches || (P[5] && !lim.wide_ok)) ? 1u : 0u;
        }
    }
    if (!tot[4]) return;
    __syncthreads();
    // the records behind the first one of every list of at most EF_BIG_CHUNKS chunks, one per thread and step: record x
    // belongs to the last list with x0 <= x (a long list has no such record: its x0 equals its successor's)
    for (uint64_t x = t; x < tot[4]; x += NT) {
        uint32_t lo = 0, hi = TILE;  // first list with x0 > x
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (info[mid].x0 > x) hi = mid; else lo = mid + 1;
        }
        const EfListInfo li = info[lo - 1u];
        const uint64_t c = 1u + (x - li.x0);
        recs[P[3] + li.c0 + c] = ef_make_rec(tile * TILE + lo - 1u, c, li.o0, li.m, li.u, li.lb, li.lw, li.hw, li.nb);
    }
}

// the records behind the first one of the long lists: a wavefront per list, a record per lane and step
__global__ void __launch_bounds__(64) k_ef_big_recs(const EfBigList *big, const uint32_t *nbig, EfChunkRec *recs) {
    const uint32_t n = *nbig;
    for (uint32_t k = blockIdx.x; k < n; k += gridDim.x) {
        const EfBigList bl = big[k];
        const uint32_t cnt = (bl.m + EF_CHUNK - 1u) / EF_CHUNK;
        for (uint32_t c = 1u + threadIdx.x; c < cnt; c += 64u)
            recs[bl.rec0 + c] = ef_make_rec(bl.l, c, bl.o0, bl.m, bl.u, bl.lb, bl.lw, bl.hw, bl.nb);
    }
}

// select-directory entries and padding word of a chunk (what k_fill_items + k_ef_hrank + k_ef_zero_low_pads do for the
// three-pass encoder).  A chunk owns the batches whose first bit 4096 k lies in (position of the id before the chunk,
// position of its last id]; the last chunk of a list also owns the batches behind its last id.  hrank[k] = number of
// ids of the list with a position below 4096 k = start + (ids of this chunk below it).
// the decode record of batch k of the chunk's list (what k_ef_build_recs derives from the CSR arrays for objects of the other encoders)
__device__ inline void ef_store_rec(EfRec *dst, const EfChunkRec &rc, uint64_t k, uint32_t done) {
    const uint64_t nhw = ((uint64_t)rc.n + 1 + (rc.u >> rc.b) + 1 + 63) / 64;  // (= ef_make_rec's)
    const uint64_t out_pos = rc.src - rc.start + done, hw_base = rc.high_word + 64 * k;
    const uint32_t nw = (uint32_t)(64 * k >= nhw ? 0 : (nhw - 64 * k < 64 ? nhw - 64 * k : 64));
    uint4 *d = (uint4 *)dst;
    d[0] = make_uint4((uint32_t)out_pos, (uint32_t)(out_pos >> 32), (uint32_t)rc.low_word, (uint32_t)(rc.low_word >> 32));
    d[1] = make_uint4((uint32_t)hw_base, (uint32_t)(hw_base >> 32), done, rc.n);
    d[2] = make_uint4(nw, rc.b, (uint32_t)k, rc.nb);
}
template <typename PT, int R>
__device__ inline void ef_chunk_directory(const EfChunkRec &rc, const uint64_t *src, uint32_t nc, const PT (&pos)[R],
                                          PT pos_before, PT pos_last, uint64_t *low, uint32_t *hrank, Chunk *batches,
                                          EfRec *drecs) {
    const uint32_t lane = lane_id();
    const bool last_chunk = rc.start + nc == rc.n;
    if (last_chunk && lane == 0) low[rc.low_word + (((uint64_t)rc.n * rc.b + 63) >> 6)] = 0ull;
    const uint64_t kfirst = rc.start ? (uint64_t)(pos_before >> 12) + 1u : 0u;
    const uint64_t kend = last_chunk ? (uint64_t)rc.nb : (uint64_t)(pos_last >> 12) + 1u;  // one past the last owned
    if (kfirst >= kend) return;
    if (kend - kfirst <= 4u) {
        for (uint64_t k = kfirst; k < kend; k++) {
            const PT T = (PT)(k << 12);
            uint32_t below = 0;
#pragma unroll
            for (int r = 0; r < R; r++) below += popc64(ballot(pos[r] < T));  // (slots past the chunk hold the maximum)
            if (lane == 0) {
                hrank[rc.batch0 + k] = rc.start + below;
                batches[rc.batch0 + k] = Chunk{rc.list, (uint32_t)k};
                if (drecs) ef_store_rec(drecs + rc.batch0 + k, rc, k, rc.start + below);
            }
        }
    } else {  // a sparse chunk spanning many batches: a lane per batch, binary search over the chunk's ids
        for (uint64_t k = kfirst + lane; k < kend; k += 64) {
            const uint64_t T = k << 12;
            uint32_t lo = 0, hi = nc;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((src[mid] >> rc.b) + rc.start + mid >= T) hi = mid; else lo = mid + 1;
            }
            hrank[rc.batch0 + k] = rc.start + lo;
            batches[rc.batch0 + k] = Chunk{rc.list, (uint32_t)k};
            if (drecs) ef_store_rec(drecs + rc.batch0 + k, rc, k, rc.start + lo);
        }
    }
}

__global__ void __launch_bounds__(64) k_ef_lowhigh(const uint64_t *sorted_ids, const EfChunkRec *recs,
                                                   uint64_t nchunks, uint64_t *low, uint64_t *high, uint32_t *hrank,
                                                   Chunk *batches, uint32_t *unsorted, EfRec *drecs, const uint32_t *abort) {
    __shared__ unsigned long long win[EF_WIN_WORDS];
    __shared__ unsigned long long img[EF_CHUNK + 8];  // low words of the chunk (EF_CHUNK * l / 64 <= EF_CHUNK)
    const uint32_t lane = lane_id();
    if (*abort) return;  // (k_ef_offsets: the streams allocated ahead of it are too small)
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const EfChunkRec rc = recs[c];
        const uint32_t start = rc.start;
        const uint32_t b = rc.b;
        const uint64_t n = rc.n;
        const uint64_t u = rc.u;
        const uint32_t nc = (uint32_t)(n - start < EF_CHUNK ? n - start : EF_CHUNK);
        const uint64_t *src = sorted_ids + rc.src - start;  // the list's first id
        unsigned long long *dst = (unsigned long long *)(high + rc.high_word);
        // the chunk's ids: eight independent coalesced loads per lane
        uint64_t v[EF_CHUNK / 64];
        const uint64_t before = start ? src[start - 1] : 0ull;  // the id before this chunk (order check)
#pragma unroll
        for (uint32_t r = 0; r < EF_CHUNK / 64; r++) {
            const uint32_t i = lane + 64 * r;
            v[r] = i < nc ? src[start + i] : ~0ull;
        }
        const uint32_t nlw = (uint32_t)(((uint64_t)nc * b + 63) >> 6);
        for (uint32_t w = lane; w < nlw + 1u; w += 64) img[w] = 0;
        uint64_t pos[EF_CHUNK / 64];
        bool bad = false;
        uint64_t carry = before;
#pragma unroll
        for (uint32_t r = 0; r < EF_CHUNK / 64; r++) {
            const uint32_t i = lane + 64 * r;
            // predecessor = previous lane's id (lane 0: the last id of the previous 64-group)
            uint32_t plo = (uint32_t)__shfl_up((int)(uint32_t)v[r], 1, 64), phi = (uint32_t)__shfl_up((int)(uint32_t)(v[r] >> 32), 1, 64);
            const uint64_t prev = lane ? (((uint64_t)phi << 32) | plo) : carry;
            carry = rl64((uint32_t)v[r], (uint32_t)(v[r] >> 32), 63);
            pos[r] = ~0ull;
            if (i < nc) {
                bad |= v[r] > u || prev > v[r];
                pos[r] = v[r] > u ? ~0ull : (v[r] >> b) + (start + i);
            }
        }
        if (ballot(bad)) {
            if (lane == 0) *(volatile uint32_t *)unsorted = 1u;  // (pinned host memory: every writer stores the same value)
            __syncthreads();
            continue;  // the object is rebuilt by the general path
        }
        // high stream: positions increase strictly, the chunk covers a contiguous bit range (see k_ef_high)
        const uint32_t lr = (nc - 1u) >> 6, ll = (nc - 1u) & 63u;  // register / lane holding the last id
        uint64_t last = 0;
#pragma unroll
        for (uint32_t r = 0; r < EF_CHUNK / 64; r++)
            if (r == lr) last = rl64((uint32_t)pos[r], (uint32_t)(pos[r] >> 32), ll);
        const uint64_t wl = last >> 6;
        // Word ownership instead of global atomics on boundary words and instead of a zeroed stream: every high word of a list is
        // written by exactly one chunk.  A chunk owns the words behind the one that holds the id before it (a list's first chunk:
        // from word 0) up to the word of its own last id (the last chunk: up to the list's last word); it leaves its bits of the
        // previous chunk's last word to that chunk and adds to word wl the bits of the (at most 63) ids after the chunk that fall into it.
        const uint64_t wlo = start ? (((before >> b) + (start - 1)) >> 6) + 1u : 0ull;
        const uint64_t whi = start + nc == n ? ((n + 1 + (u >> b) + 1 + 63) >> 6) - 1u : wl;
        uint64_t tail_bits = 0;
        {
            const uint64_t j = (uint64_t)start + nc + lane;
            if (j < n) {
                const uint64_t vn = src[j];
                const uint64_t pn = (vn >> b) + j;
                if (vn <= u && (pn >> 6) == wl) tail_bits = 1ull << (pn & 63);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)tail_bits, o, 64);
                const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(tail_bits >> 32), o, 64);
                tail_bits |= ((uint64_t)hi << 32) | lo;
            }
        }
        if (b) {  // low stream: OR the l low bits of every id into the LDS image of the chunk's words
            __syncthreads();
            const uint64_t keep = (1ull << b) - 1ull;
#pragma unroll
            for (uint32_t r = 0; r < EF_CHUNK / 64; r++) {
                const uint32_t i = lane + 64 * r;
                if (i < nc) {
                    const uint64_t x = v[r] & keep;
                    const uint32_t p = i * b, sh = p & 63u;
                    atomicOr(&img[p >> 6], x << sh);
                    if (sh + b > 64u) atomicOr(&img[(p >> 6) + 1u], x >> (64u - sh));
                }
            }
            __syncthreads();
            uint64_t *ldst = low + rc.low_word + (((uint64_t)start * b) >> 6);
            for (uint32_t w = lane; w < nlw; w += 64) ldst[w] = img[w];
        }
        uint32_t *win32 = (uint32_t *)win;  // 32-bit LDS atomics (the 64-bit ones run at half rate)
        for (uint64_t wbase = wlo; wbase <= whi; wbase += EF_WIN_WORDS) {
            win[lane] = 0;
            win[lane + 64] = 0;
            __syncthreads();
#pragma unroll
            for (uint32_t r = 0; r < EF_CHUNK / 64; r++) {
                const uint64_t w = pos[r] >> 6;
                if (pos[r] != ~0ull && w >= wbase && w - wbase < EF_WIN_WORDS)
                    atomicOr(&win32[(uint32_t)(pos[r] - wbase * 64u) >> 5], 1u << (pos[r] & 31));
            }
            __syncthreads();
#pragma unroll
            for (uint32_t t = 0; t < 2; t++) {
                const uint32_t k = lane + 64 * t;
                const uint64_t w = wbase + k;
                unsigned long long hv = win[k];
                if (w == wl) hv |= tail_bits;
                if (w <= whi) dst[w] = hv;  // (empty words too: nothing zeroes the stream beforehand)
            }
            __syncthreads();
        }
        ef_chunk_directory<uint64_t>(rc, src + start, nc, pos, start ? (before >> b) + (start - 1) : 0ull, last,
                                     low, hrank, batches, drecs);
    }
}

// The same single-pass encoder for objects whose ids all fit 32 bits and whose lists are shorter than 2^30 (every
// Faiss index in the reference's benchmarks): ids are still loaded as u64, but order checks, bit positions and the
// low-bit image run on 32-bit values -- the 64-bit version spends about half of its instructions on two-register
// compares, shifts and shuffles, and the kernel is as much issue- as bandwidth-bound (no stores, no LDS: 162 of
// 227 us per 64 M ids).
// one chunk record with R id registers per lane (64 R >= the ids of the chunk)
// FULL: the chunk holds exactly 64 R ids (every chunk but the last of a list; every chunk of an object of equal lists that are a
// multiple of 64 R long): no lane predicate on the id slots -- the exec-mask arithmetic of `i < nc` for R slots in five loops was a
// quarter of the kernel's 538 scalar instructions, and the kernel sits at the scalar AND the vector issue limit.
// V2 (round 5, last session: the kernel is bound by its vector instructions -- DESIGN section 13): the bit offset of an id's low bits is
// lane * b (one 24-bit multiply per chunk) + a wave-uniform 64 r b instead of a full-rate-quarter 32-bit multiply per id slot; the
// "fits 32 bits" check is one OR per slot and one test per chunk, "not beyond the universe" is tested on the chunk's last id only
// (the chunk is ascending or it is handed back anyway).
template <int R, bool FULL = false, bool V2 = false>
__device__ __forceinline__ void ef_lowhigh32_chunk(const EfChunkRec &rc, const uint64_t *__restrict__ sorted_ids,
                                                   uint64_t *__restrict__ low, uint64_t *__restrict__ high,
                                                   uint32_t *__restrict__ hrank, Chunk *__restrict__ batches,
                                                   uint32_t *__restrict__ unsorted, EfRec *__restrict__ drecs, uint32_t *win32,
                                                   uint32_t *img32) {
    const uint32_t lane = lane_id();
    constexpr uint32_t NONE = 0xffffffffu;
    const uint32_t start = rc.start;
    const uint32_t b = rc.b;
    const uint32_t n = rc.n;
    const uint32_t u = (uint32_t)rc.u;
    const uint32_t nc = FULL ? 64u * R : (n - start < EF_CHUNK ? n - start : EF_CHUNK);
    const uint64_t *src = sorted_ids + rc.src - start;  // the list's first id
    uint64_t *dst = high + rc.high_word;
    uint64_t v[R];
    const uint64_t before64 = start ? src[start - 1] : 0ull;  // the id before this chunk (order check)
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        const uint32_t i = lane + 64 * r;
        v[r] = (FULL || i < nc) ? src[start + i] : 0ull;
    }
    const uint32_t jn = start + nc + lane;  // the (at most 64) ids after the chunk: last word's ownership
    const uint64_t vnext = jn < n ? src[jn] : ~0ull;
    const uint32_t nlw32 = (nc * b + 31u) >> 5;
    for (uint32_t w = lane; w < nlw32 + 2u; w += 64) img32[w] = 0;
    uint32_t pos[R];
    bool bad = (before64 >> 32) != 0;
    uint32_t hi_or = 0;  // (V2)
    uint32_t carry = (uint32_t)before64;
    const uint32_t nr = FULL ? (uint32_t)R : (nc + 63u) >> 6;  // registers in use (wave-uniform: short lists skip the rest)
#pragma unroll
    for (uint32_t r = 0; r < R; r++) {
        pos[r] = NONE;
        if (FULL || r < nr) {
            const uint32_t i = lane + 64 * r;
            const uint32_t x = (uint32_t)v[r];
            const uint32_t up = lane_shr1(x);
            const uint32_t prev = lane ? up : carry;
            carry = rl(x, 63);
            if (V2) hi_or |= (uint32_t)(v[r] >> 32);  // (slots past the chunk hold 0)
            if (FULL || i < nc) {
                if (V2) bad |= prev > x;
                else bad |= (uint32_t)(v[r] >> 32) != 0u || x > u || prev > x;
                pos[r] = (x >> b) + (start + i);
            }
        }
    }
    const uint32_t lr = (nc - 1u) >> 6, ll = (nc - 1u) & 63u;  // register / lane holding the last id
    uint32_t last = 0, xlast = 0;
#pragma unroll
    for (uint32_t r = 0; r < R; r++)
        if (r == lr) { last = rl(pos[r], ll); if (V2) xlast = rl((uint32_t)v[r], ll); }
    if (V2) bad |= hi_or != 0u;
    if (ballot(bad) || (V2 && xlast > u)) {
        if (lane == 0) *(volatile uint32_t *)unsorted = 1u;  // (pinned host memory: every writer stores the same value)
        wave_lds_sync();
        return;  // the object is rebuilt by the general path
    }
    const uint32_t wl = last >> 6;
    // word ownership as in k_ef_lowhigh: words [wlo, whi] are this chunk's
    const uint32_t before = (uint32_t)before64;
    const uint32_t wlo = start ? (((before >> b) + (start - 1u)) >> 6) + 1u : 0u;
    const uint32_t whi = start + nc == n ? (uint32_t)(((uint64_t)n + 1 + (u >> b) + 1 + 63) >> 6) - 1u : wl;
    // the ids behind the chunk whose bit falls into the chunk's last word: set through the last window
    uint32_t pnext = NONE;
    if (jn < n && (vnext >> 32) == 0) {
        const uint32_t xn = (uint32_t)vnext;
        const uint32_t pn = (xn >> b) + jn;
        if (xn <= u && (pn >> 6) == wl) pnext = pn;
    }
    if (b) {  // low stream: the l low bits of every id into the LDS image, 32-bit halves
        wave_lds_sync();
        const uint32_t keep = b >= 32u ? 0xffffffffu : ((1u << b) - 1u);
        const uint32_t lane_b = __umul24(lane, b);  // (V2)
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
            const uint32_t i = lane + 64 * r;
            if (FULL || (r < nr && i < nc)) {
                const uint32_t x = (uint32_t)v[r] & keep;
                const uint32_t p = V2 ? lane_b + (64u * r) * b : i * b, sh = p & 31u;
                atomicOr(&img32[p >> 5], x << sh);
                if (sh + b > 32u) atomicOr(&img32[(p >> 5) + 1u], x >> (32u - sh));
            }
        }
        wave_lds_sync();
        const uint32_t nlw = (nc * b + 63u) >> 6;
        uint64_t *ldst = low + rc.low_word + (((uint64_t)start * b) >> 6);
        const uint64_t *img = (const uint64_t *)img32;
        for (uint32_t w = lane; w < nlw; w += 64) ldst[w] = img[w];
    }
    const uint64_t *win = (const uint64_t *)win32;
    for (uint32_t wbase = wlo; wbase <= whi; wbase += EF_WIN_WORDS) {
        win32[lane] = 0; win32[lane + 64] = 0; win32[lane + 128] = 0; win32[lane + 192] = 0;
        wave_lds_sync();
#pragma unroll
        for (uint32_t r = 0; r < R; r++) {
            if (FULL || r < nr) {
                const uint32_t rel = pos[r] - wbase * 64u;  // bit inside the window (wraps for earlier windows / NONE)
                if ((FULL || pos[r] != NONE) && (pos[r] >> 6) >= wbase && rel < EF_WIN_WORDS * 64u)
                    atomicOr(&win32[rel >> 5], 1u << (rel & 31u));
            }
        }
        if (wl >= wbase && wl - wbase < EF_WIN_WORDS && pnext != NONE) {
            const uint32_t rel = pnext - wbase * 64u;
            atomicOr(&win32[rel >> 5], 1u << (rel & 31u));
        }
        wave_lds_sync();
#pragma unroll
        for (uint32_t t = 0; t < 2; t++) {
            const uint32_t k = lane + 64 * t;
            const uint32_t w = wbase + k;
            if (w <= whi) dst[w] = win[k];  // (empty words too: nothing zeroes the stream beforehand)
        }
        wave_lds_sync();
    }
    ef_chunk_directory<uint32_t>(rc, src + start, nc, pos, start ? (before >> b) + (start - 1u) : 0u, last,
                                 low, hrank, batches, drecs);
}
// RMAX = 8: any object.  RMAX = 4: no list of the object holds more than 256 ids (64 instead of 85 VGPRs, half the code).
// SMALL: chunks of up to 64 / 256 ids take the code unrolled for one / four id registers -- a chunk costs its unrolled
// instructions whether the slots are used or not, and the lists of a 65 536-list IVF index are mostly that short; the
// extra code paths cost a wavefront per SIMD (97 VGPRs), so objects of mostly full chunks take the plain kernel.
// (recs / sorted_ids are read-only for the whole launch: __restrict__ lets the record and the wave-uniform neighbour ids come
// through scalar loads instead of 64 identical vector loads + v_readfirstlane each)
// WITHFULL: full chunks take the predicate-free body (VIDC_EF_NO_FULL=1: the single-body kernels of round 3, for comparisons)
template <int RMAX, bool SMALL, bool WITHFULL = true, bool V2 = false>
__global__ void __launch_bounds__(64) k_ef_lowhigh32(const uint64_t *__restrict__ sorted_ids, const EfChunkRec *__restrict__ recs,
                                                     uint64_t nchunks, uint64_t *__restrict__ low, uint64_t *__restrict__ high,
                                                     uint32_t *__restrict__ hrank, Chunk *__restrict__ batches,
                                                     uint32_t *__restrict__ unsorted, EfRec *__restrict__ drecs,
                                                     const uint32_t *__restrict__ abort) {
    __shared__ uint32_t win32[EF_WIN_WORDS * 2];
    __shared__ uint32_t img32[(64 * RMAX + 8) * 2];  // low words of the chunk, as 32-bit halves
    if (*abort) return;  // (k_ef_offsets: the streams allocated ahead of it are too small)
    // (Round 6, measured and dropped: the record and the ids of the wavefront's NEXT chunk requested before the current one is worked
    // on -- 16 more registers, S2 encode 3.15-3.22 ms against 3.10: with 18 wavefronts per CU the loads of other wavefronts already
    // fill the time a wavefront waits, and the kernel sits at its issue limit.  HISTORY.md.)
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const EfChunkRec rc = recs[c];
        const uint32_t nc = rc.n - rc.start < EF_CHUNK ? rc.n - rc.start : EF_CHUNK;
        // (the predicate-free body pays for four id registers per lane -- 16 M ids in lists of 256: 66 -> 62 us, 10 M ids in 65 536
        // Zipf lists -3 % -- and not for eight: 64 M ids in lists of 1024 165 against 158 us, S2 3.19 against 3.23 ms, interleaved)
        if (!SMALL || (RMAX > 4 && nc > 256u))
            ef_lowhigh32_chunk<RMAX, false, V2>(rc, sorted_ids, low, high, hrank, batches, unsorted, drecs, win32, img32);
        else if (WITHFULL && nc == 256u)
            ef_lowhigh32_chunk<4, true, V2>(rc, sorted_ids, low, high, hrank, batches, unsorted, drecs, win32, img32);
        else if (nc > 64u)
            ef_lowhigh32_chunk<4, false, V2>(rc, sorted_ids, low, high, hrank, batches, unsorted, drecs, win32, img32);
        else
            ef_lowhigh32_chunk<1, false, V2>(rc, sorted_ids, low, high, hrank, batches, unsorted, drecs, win32, img32);
        wave_lds_sync();
    }
}

// the padding word behind every non-empty list's low stream: the only low words the encoder kernels do not write
__global__ void k_ef_zero_low_pads(const uint64_t *low_off, uint32_t nlist, uint64_t *low) {
    for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < nlist; l += gridDim.x * blockDim.x)
        if (low_off[l + 1] > low_off[l]) low[low_off[l + 1] - 1] = 0ull;
}

// the same directory rebuilt from the high stream alone (import of a saved object): one wavefront per list,
// running popcount over its batches of 64 words
__global__ void __launch_bounds__(64) k_ef_hrank_from_high(const uint64_t *high, const uint64_t *high_off,
                                                           const uint64_t *batch_off, uint32_t nlist, uint32_t *hrank) {
    const uint32_t lane = lane_id();
    for (uint32_t l = blockIdx.x; l < nlist; l += gridDim.x) {
        const uint64_t *hw = high + high_off[l];
        const uint64_t nhw = high_off[l + 1] - high_off[l];
        const uint64_t nb = batch_off[l + 1] - batch_off[l];
        uint32_t run = 0;
        for (uint64_t bt = 0; bt < nb; bt++) {
            if (lane == 0) hrank[batch_off[l] + bt] = run;
            const uint64_t w = bt * 64 + lane;
            uint32_t c = w < nhw ? popc64(hw[w]) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c += (uint32_t)__shfl_xor((int)c, o, 64);
            run += c;
        }
    }
}

// select directory: hrank[item] = number of elements whose high bit lies before batch `b` of list `l`
// = first j with (x_j >> l) + j >= 4096 * b  (positions increase with j: binary search, no scan)
__global__ void k_ef_hrank(const uint64_t *sorted_ids, const uint64_t *offsets, const uint32_t *lbits,
                           const Chunk *batches, uint64_t nbatches, uint32_t *hrank) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < nbatches; it += stride) {
        const Chunk bt = batches[it];
        const uint64_t off = offsets[bt.list];
        const uint64_t n = offsets[bt.list + 1] - off;
        const uint32_t b = lbits[bt.list];
        const uint64_t target = (uint64_t)bt.start * 4096ull;
        uint64_t lo = 0, hi = n;  // first j in [0, n] with pos_j >= target
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if ((sorted_ids[off + mid] >> b) + mid >= target) hi = mid; else lo = mid + 1;
        }
        hrank[it] = (uint32_t)lo;
    }
}

// bulk decode (select_enumerator, elias_fano.hpp:210-261): one wavefront per batch of 64 high words.
// items == nullptr: work item wi is list `worklist[wi]` (or wi) decoded batch by batch by one wave (used for
// short selections); otherwise items[wi] = (list, batch) and every batch is independent.
// Per batch: lane t popcounts high word t, a prefix scan gives every set bit its rank, the bit positions are
// TRANSPOSED through LDS (pos[rank]) and the values are then produced rank-major: lane r reads pos[r], the low bits
// of element r (both words unconditionally, eight elements per lane in flight) and stores out[r] -- every global
// access is coalesced and independent of the others.
#define EF_BATCH_BITS 4096u
__global__ void __launch_bounds__(64) k_ef_decode(const uint64_t *low, const uint64_t *high, const uint64_t *offsets,
                                                  const uint64_t *low_off, const uint64_t *high_off,
                                                  const uint32_t *lbits, const uint64_t *batch_off,
                                                  const uint32_t *hrank, uint32_t nwork, const Chunk *items,
                                                  const uint64_t *worklist, const uint64_t *out_off, uint64_t *out,
                                                  int32_t *out_rows, uint32_t K, uint32_t nsplit) {
    __shared__ uint16_t spos[EF_BATCH_BITS];  // bit position (inside the batch) of the element with in-batch rank r
    const uint32_t lane = lane_id();
    // nsplit > 1 (decode_lists of few long lists): `nsplit` workgroups share a list, workgroup s takes its batches s, s + nsplit, ...
    // (a search that touched 16 lists of up to 52 000 ids had 16 wavefronts walk up to 102 batches each: 0.16 ms for 97 000 ids)
    for (uint32_t vi = blockIdx.x; vi < nwork * nsplit; vi += gridDim.x) {
        const uint32_t wi = vi / nsplit, split = vi - wi * nsplit;
        const uint64_t l = items ? items[wi].list : (worklist ? worklist[wi] : wi);
        const uint64_t off = out_rows ? (uint64_t)wi * K : (out_off ? out_off[wi] : offsets[l]);
        const uint64_t m = offsets[l + 1] - offsets[l];
        if (out_rows && split == 0) {
            for (uint32_t j = lane; j < K; j += 64)
                if (j >= m) out_rows[off + j] = -1;
        }
        if (!m) continue;
        const uint32_t b = lbits[l];
        const uint64_t keep = b ? ((b >= 64 ? 0ull : (1ull << b)) - 1ull) : 0ull;
        const uint64_t *lw = low + low_off[l];
        const uint64_t *hw = high + high_off[l];
        const uint64_t nhw = high_off[l + 1] - high_off[l];
        const uint64_t nb = batch_off[l + 1] - batch_off[l];
        const uint64_t b_first = items ? items[wi].start : 0, b_last = items ? b_first + 1 : nb;
        for (uint64_t bt = b_first + split; bt < b_last; bt += nsplit) {
            const uint64_t done = hrank[batch_off[l] + bt];
            if (done >= m) break;
            const uint64_t wi64 = bt * 64 + lane;
            uint64_t word = wi64 < nhw ? hw[wi64] : 0ull;
            const uint32_t c = popc64(word);
            uint32_t incl = c;  // inclusive prefix sum over the 64 lanes
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
                if (lane >= (uint32_t)o) incl += v;
            }
            const uint32_t tot0 = rl(incl, 63);
            const uint32_t tot = (uint32_t)(done + tot0 <= m ? tot0 : m - done);  // elements of this batch
            uint32_t r = incl - c;
            while (word) {  // transpose: position of every set bit, indexed by its rank inside the batch
                const uint32_t bit = (uint32_t)__builtin_ctzll(word);
                word &= word - 1;
                spos[r++] = (uint16_t)(lane * 64u + bit);
            }
            __syncthreads();
            const uint64_t pbase = bt * EF_BATCH_BITS;
            for (uint32_t r0 = 0; r0 < tot; r0 += 512) {
                uint64_t a[8], bw[8];
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {  // low bits: both words, unconditionally (a padding word follows)
                    const uint32_t rr = r0 + lane + 64 * k;
                    const uint64_t bp = (done + (rr < tot ? rr : 0u)) * b;
                    a[k] = b ? lw[bp >> 6] : 0ull;
                    bw[k] = b ? lw[(bp >> 6) + 1] : 0ull;
                }
#pragma unroll
                for (uint32_t k = 0; k < 8; k++) {
                    const uint32_t rr = r0 + lane + 64 * k;
                    if (rr < tot) {
                        const uint64_t rank = done + rr;
                        const uint32_t sh = (uint32_t)((rank * b) & 63);
                        const uint64_t lo = ((a[k] >> sh) | (sh ? bw[k] << (64 - sh) : 0ull)) & keep;
                        const uint64_t val = ((pbase + spos[rr] - rank) << b) | lo;
                        if (out_rows) out_rows[off + rank] = (int32_t)val;
                        else out[off + rank] = val;
                    }
                }
            }
            __syncthreads();
        }
    }
}

// one thread per batch: the record of EfRec from the CSR arrays (coalesced over batches, off every decode's path).  The records
// of a workgroup leave through LDS as contiguous 16-byte pieces (a 48-byte struct store per thread is three stores of 16 bytes
// at a stride of 48: every 64-byte line of the record array was written in three partial passes).
__global__ void __launch_bounds__(256) k_ef_build_recs(const uint64_t *__restrict__ offsets, const uint64_t *__restrict__ low_off,
                                const uint64_t *__restrict__ high_off, const uint32_t *__restrict__ lbits,
                                const uint64_t *__restrict__ batch_off, const uint32_t *__restrict__ hrank,
                                const Chunk *__restrict__ items, uint64_t nbatches, EfRec *recs, unsigned int *max_cnt) {
    static_assert(sizeof(EfRec) == 48, "three 16-byte pieces per record");
    __shared__ uint4 stage[256 * 3];
    unsigned int mx = 0;
    for (uint64_t base = (uint64_t)blockIdx.x * 256u; base < nbatches; base += (uint64_t)gridDim.x * 256u) {
        const uint64_t it = base + threadIdx.x;
        if (it < nbatches) {
            const uint64_t l = items[it].list, bt = items[it].start;
            const uint64_t o0 = offsets[l], m = offsets[l + 1] - o0;
            const uint64_t h0 = high_off[l], nhw = high_off[l + 1] - h0;
            const uint64_t b0 = batch_off[l], nb = batch_off[l + 1] - b0;
            const uint64_t done = hrank[b0 + bt];
            const uint64_t next = bt + 1 < nb ? (uint64_t)hrank[b0 + bt + 1] : m;
            EfRec r;
            r.out_pos = o0 + done;
            r.low_base = low_off[l];
            r.hw_base = h0 + bt * 64;
            r.done = (uint32_t)done;
            const uint32_t cnt = done >= m ? 0u : (uint32_t)((next < m ? next : m) - done);
            r.m = (uint32_t)m;
            r.nw = (uint32_t)(bt * 64 >= nhw ? 0 : (nhw - bt * 64 < 64 ? nhw - bt * 64 : 64));
            r.b = lbits[l];
            r.bt = (uint32_t)bt;
            r.nb = (uint32_t)nb;
            uint4 *dst = stage + threadIdx.x * 3u;
            dst[0] = make_uint4((uint32_t)r.out_pos, (uint32_t)(r.out_pos >> 32), (uint32_t)r.low_base, (uint32_t)(r.low_base >> 32));
            dst[1] = make_uint4((uint32_t)r.hw_base, (uint32_t)(r.hw_base >> 32), r.done, r.m);
            dst[2] = make_uint4(r.nw, r.b, r.bt, r.nb);
            mx = cnt > mx ? cnt : mx;
        }
        __syncthreads();
        const uint64_t left = nbatches - base < 256u ? nbatches - base : 256u;  // records of this pass
        uint4 *out = (uint4 *)(recs + base);
        for (uint32_t k = threadIdx.x; k < left * 3u; k += 256u) out[k] = stage[k];
        __syncthreads();
    }
    // the largest batch: one atomic per WORKGROUP, and only while it still raises the maximum (the compiler already folds the
    // atomics of a wavefront into one; 20 000 of them on one address were 0.17 of this kernel's 0.19 ms on S2)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int other = (unsigned int)__shfl_xor((int)mx, o, 64);
        mx = other > mx ? other : mx;
    }
    __shared__ unsigned int wmx[4];
    if ((threadIdx.x & 63u) == 0u) wmx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int b = wmx[0];
        for (int k = 1; k < 4; k++) b = wmx[k] > b ? wmx[k] : b;
        if (max_cnt && b > *(volatile unsigned int *)max_cnt) atomicMax(max_cnt, b);
    }
}

// decode_all with batch records: record -> {high word, low words of the first 512 elements} -> stores.  The low
// words of rank-major element r only depend on `done` (in the record), not on the high word, so both loads leave
// together; two dependent round trips instead of five.
// LW = uint64_t: any ids.  LW = uint32_t: objects whose ids fit 32 bits -- low bits come as 32-bit word pairs and the
// value is assembled in one register (the 64-bit version reads 16 bytes of low words per element and does
// two-register shifts: 0.19 vs 0.13 ms per 64 M ids).
template <typename LW, int R, bool V2A = true>  // R: elements per lane and pass (4: no batch of the object holds more than 256)
__global__ void __launch_bounds__(64) k_ef_decode_rec(const uint64_t *low, const uint64_t *high, const EfRec *recs,
                                                      uint32_t nwork, uint64_t *out) {
    extern __shared__ uint16_t spos[];  // max elements per batch of this object (<= EF_BATCH_BITS) entries
    constexpr uint32_t WB = sizeof(LW) * 8u, WSH = sizeof(LW) == 8 ? 6u : 5u;
    const uint32_t lane = lane_id();
    for (uint32_t wi = blockIdx.x; wi < nwork; wi += gridDim.x) {
        // the record and the next one (its `done` ends this batch) as 24 dwords through one vector load, broadcast to SGPRs
        const uint32_t *rp = (const uint32_t *)(recs + wi);
        const uint32_t rv = lane < 24u ? rp[lane] : 0u;
        const uint64_t out_pos = ((uint64_t)rl(rv, 1) << 32) | rl(rv, 0);
        const uint64_t low_base = ((uint64_t)rl(rv, 3) << 32) | rl(rv, 2);
        const uint64_t hw_base = ((uint64_t)rl(rv, 5) << 32) | rl(rv, 4);
        const uint32_t done = rl(rv, 6), m = rl(rv, 7), nw = rl(rv, 8), b = rl(rv, 9), bt = rl(rv, 10), nb = rl(rv, 11);
        const uint32_t next = rl(rv, 18);
        const uint32_t tot = done >= m ? 0u : ((bt + 1u < nb && next < m) ? next : m) - done;
        if (!tot) continue;
        const LW keep = b ? (LW)(((b >= WB ? (LW)0 : ((LW)1 << b))) - (LW)1) : (LW)0;
        const LW *lw = (const LW *)(low + low_base);
        // V2 (LW = uint32_t, round 5): the bit offset (done + rr) * b of element rr's low bits was a 64-bit multiply per element and pass,
        // twice (v_mad_u64_u32 + v_mul_lo_u32: quarter rate).  done * b is wave-uniform (scalar), rr * b < 2^17 is a 24-bit multiply, the
        // word index relative to the uniform word (done * b) >> 5 fits 32 bits (uniform base + 32-bit offset loads), and the two words
        // are joined by one v_alignbit: 517 -> 374 vector instructions (R = 8), 56 -> 44 VGPRs.  The second word's index is written
        // (bo + 32) >> 5 on purpose: as (bo >> 5) + 1 the compiler merges the pair into ONE 8-byte load at a 4-byte boundary, which costs
        // 30 % of the kernel, and an empty asm on the index to keep them apart costs 25 %.  VIDC_EF_DEC_V1=1
        // (V2A = false) keeps the round-4 arithmetic.
        constexpr bool V2 = V2A && sizeof(LW) == 4;
        const uint64_t done_b = (uint64_t)done * b;
        const uint32_t *lwu = (const uint32_t *)lw + (done_b >> 5);
        const uint32_t db5 = (uint32_t)done_b & 31u;
        const uint32_t sh = nw <= 8u ? 3u : (nw <= 16u ? 2u : (nw <= 32u ? 1u : 0u));
        const uint32_t wi_ = lane >> sh, pw = 64u >> sh, sub = lane & ((1u << sh) - 1u);
        const uint64_t word = wi_ < nw ? high[hw_base + wi_] : 0ull;
        LW a[R], bw[R];
#pragma unroll
        for (uint32_t k = 0; k < R; k++) {
            const uint32_t rr = lane + 64 * k;
            if (V2) {
                const uint32_t bo = db5 + __umul24(rr < tot ? rr : 0u, b);
                uint32_t i1 = ((bo + 32u) >> 5);
                a[k] = b ? (LW)lwu[bo >> 5] : (LW)0;
                bw[k] = b ? (LW)lwu[i1] : (LW)0;
            } else {
                const uint64_t bp = ((uint64_t)done + (rr < tot ? rr : 0u)) * b;
                a[k] = b ? lw[bp >> WSH] : (LW)0;
                bw[k] = b ? lw[(bp >> WSH) + 1] : (LW)0;
            }
        }
        {
            const uint64_t piece = sh ? (word >> (pw * sub)) & ((1ull << pw) - 1ull) : word;
            const uint32_t c = popc64(piece);
            uint32_t r = wave_scan32(c) - c;
            uint32_t h0 = (uint32_t)piece, h1 = (uint32_t)(piece >> 32);
            const uint32_t p0 = wi_ * 64u + pw * sub;
            while (h0) {
                spos[r++] = (uint16_t)(p0 + (uint32_t)__builtin_ctz(h0));
                h0 &= h0 - 1u;
            }
            while (h1) {
                spos[r++] = (uint16_t)(p0 + 32u + (uint32_t)__builtin_ctz(h1));
                h1 &= h1 - 1u;
            }
        }
        wave_lds_sync();
        const LW pbase = (LW)bt * EF_BATCH_BITS;
        for (uint32_t r0 = 0; r0 < tot; r0 += 64 * R) {
            if (r0) {
#pragma unroll
                for (uint32_t k = 0; k < R; k++) {
                    const uint32_t rr = r0 + lane + 64 * k;
                    if (V2) {
                        const uint32_t bo = db5 + __umul24(rr < tot ? rr : 0u, b);
                        uint32_t i1 = ((bo + 32u) >> 5);
                        a[k] = b ? (LW)lwu[bo >> 5] : (LW)0;
                        bw[k] = b ? (LW)lwu[i1] : (LW)0;
                    } else {
                        const uint64_t bp = ((uint64_t)done + (rr < tot ? rr : 0u)) * b;
                        a[k] = b ? lw[bp >> WSH] : (LW)0;
                        bw[k] = b ? lw[(bp >> WSH) + 1] : (LW)0;
                    }
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < R; k++) {
                const uint32_t rr = r0 + lane + 64 * k;
                if (rr < tot) {
                    const uint32_t rank = done + rr;
                    if (V2) {
                        const uint32_t shb = (db5 + __umul24(rr, b)) & 31u;
                        const uint32_t lo = __builtin_amdgcn_alignbit((uint32_t)bw[k], (uint32_t)a[k], shb) & (uint32_t)keep;
                        out[out_pos + rr] = (uint64_t)((((uint32_t)pbase + spos[rr] - rank) << b) | lo);
                    } else {
                        const uint32_t sh = (uint32_t)(((uint64_t)rank * b) & (WB - 1u));
                        const LW lo = (LW)((a[k] >> sh) | (sh ? (LW)(bw[k] << (WB - sh)) : (LW)0)) & keep;
                        out[out_pos + rr] = (uint64_t)((LW)((LW)(pbase + spos[rr] - rank) << b) | lo);
                    }
                }
            }
        }
        wave_lds_sync();
    }
}

// ================================================================================================================
// graph rows (EliasFanoNSGGraph, altid_impl.cpp:53-101; rows of <= 64 edges): one row per LANE, 64 consecutive rows per
// wavefront (rows_tile.h).  The object is a fixed-stride ARENA: row i owns words [i * S, (i + 1) * S) of d_arena, LW low
// words followed by HW high words, with S = LW + HW known from (K, largest id) alone:
//     low  <= max over n <= K of n * msb(U / n) bits     (elias_fano.hpp:28: l = msb(universe / n))
//     high =  (n + 1) + (u >> l) + 1 <= 3 n + 1 bits     (elias_fano.hpp:29; u >> l < 2 n)
// so the encoder needs no geometry pass, no offset scan and no size read-back, a wavefront stores its 64 records as one
// contiguous block, and a row is found by a multiplication ("fixed-stride output arena instead of 10^6 heap
// objects").  The words of a record are bit-for-bit the words of the per-list streams (what vidc_ef_export returns); unused
// words are zero.  Per row the arena keeps {universe, n | l << 8}.
// ================================================================================================================
struct EfRowsTot {  // 64 slots, summed by the host (a single counter would serialise 15 625 atomics)
    unsigned long long bits, edges;
};
struct EfRowsFlags {
    uint32_t bad, over, maxid, pad;
};

template <int KP, bool FULL>  // FULL: K == KP
__global__ void __launch_bounds__(64) k_ef_rows_encode_tile(const int32_t *__restrict__ rows, uint64_t N, uint32_t K, uint32_t kmagic,
                                                            uint32_t vec, uint32_t LW, uint32_t HW, uint32_t smagic, uint32_t ubound,
                                                            uint64_t *__restrict__ arena, uint2 *__restrict__ meta, EfRowsTot *tot,
                                                            EfRowsFlags *flags) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];  // the tile image, then the 64 records: sized by the host
    const uint32_t lane = lane_id();
    const bool pf = FULL && vec != 0u;
    // LDS image: lane t's record at dword t * SDW, SDW = 2 S + 1 (odd: a per-lane dword index is conflict-free)
    const uint32_t S = LW + HW, SDW = 2u * S + 1u;
    const uint32_t step_rows = tile_div(64u, S, smagic), step_w = 64u - step_rows * S;
    unsigned long long bits = 0, edges = 0;
    uint32_t mx = 0;
    bool anybad = false, anyover = false;
    // One tile per wavefront (the persistent form with the next tile's block in flight, which the decoders use, measured slower here:
    // 254 registers and a wait for the previous tile's stores at every commit -- see k_compact_rows_encode_tile).
    {
        const uint64_t row0 = (uint64_t)blockIdx.x * 64u;
        const uint32_t nrows = (uint32_t)(N - row0 < 64u ? N - row0 : 64u);
        uint32_t r[KP];
        {
            TileRegs<KP> pre;
            if (pf) tile_issue_rows<KP>(rows, row0, nrows, pre);
            tile_commit_rows<KP, FULL>(rows, row0, nrows, K, kmagic, pf, pre, lds, r);
        }
        bool bad;
        uint32_t u;  // universe = the largest id of the row, taken before the sort (altid_impl.cpp:75)
        uint32_t n = tile_row_edges<KP>(r, bad, u);
        if (lane >= nrows) { n = 0; u = 0; bad = false; }  // (lanes behind the last row of the last tile)
        const bool over = u > ubound;
        lane_sort<KP>(r);  // altid_impl.cpp:76 (padding 0xffffffff sorts to the end)
        uint32_t l = 0;    // elias_fano.hpp:28: msb(u / n), 0 when u / n == 0
        if (n && u >= n) {
            l = (uint32_t)__builtin_clz(n) - (uint32_t)__builtin_clz(u);
            if ((n << l) > u) l--;
        }
        const uint32_t hb = n ? (n + 1u) + (u >> l) + 1u : 0u;  // :29; empty rows have no bitstream object (altid_impl.cpp:69-71)
        {
            uint4 *z = (uint4 *)lds;
            const uint32_t nz = (64u * SDW + 3u) / 4u;
            for (uint32_t f = lane; f < nz; f += 64u) z[f] = make_uint4(0u, 0u, 0u, 0u);
        }
        wave_lds_sync();
        {
            // Branch-free: an element that is not one (e >= n, or a row that is reported as bad) ORs zeros, which is harmless anywhere
            // inside the LDS image (bit e * l <= 63 * 30: dword 60 of the lane's record at most, and the host adds 64 dwords behind the records).
            uint32_t *rec = lds + lane * SDW;
            uint32_t *rech = rec + 2u * LW;
            const uint32_t nn = (bad || over) ? 0u : n;
            const uint32_t keep = (1u << l) - 1u;  // l <= 30
            uint32_t bp = 0;
#pragma unroll
            for (int e = 0; e < KP; e++) {
                const bool on = (uint32_t)e < nn;
                const uint32_t x = on ? r[e] : 0u;
                // low stream: l bits per element, LSB first (elias_fano.hpp:40-42)
                const uint64_t v = (uint64_t)(x & keep) << (bp & 31u);
                uint32_t *d = rec + (bp >> 5);
                atomicOr(d, (uint32_t)v);
                atomicOr(d + 1, (uint32_t)(v >> 32));
                // high stream: bit (x >> l) + e (:43)
                const uint32_t pos = (x >> l) + (uint32_t)e;
                atomicOr(&rech[pos >> 5], (on ? 1u : 0u) << (pos & 31u));
                bp += l;
            }
        }
        wave_lds_sync();
        {
            uint64_t *dst = arena + row0 * S;
            const uint32_t total = nrows * S;  // S <= 32 words per row: at most 32 per lane
            TileCursor c = tile_cursor(S, smagic);
#pragma unroll 4
            for (uint32_t f = lane; f < total; f += 64u) {
                const uint32_t *q = lds + c.row * SDW + 2u * c.w;
                dst[f] = ((uint64_t)q[1] << 32) | q[0];
                tile_advance(c, S, step_rows, step_w);
            }
        }
        if (lane < nrows) meta[row0 + lane] = make_uint2(u, n | (l << 8));
        bits += (unsigned long long)n * l + hb;
        edges += n;
        mx = mx > u ? mx : u;
        anybad = anybad || bad;
        anyover = anyover || over;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bits += __shfl_xor(bits, o, 64);
        edges += __shfl_xor(edges, o, 64);
        const uint32_t w = (uint32_t)__shfl_xor((int)mx, o, 64);
        mx = mx > w ? mx : w;
    }
    const uint64_t wbad = ballot(anybad), wover = ballot(anyover);
    if (lane == 0) {
        if (edges) {
            atomicAdd(&tot[blockIdx.x & 63u].bits, bits);
            atomicAdd(&tot[blockIdx.x & 63u].edges, edges);
        }
        if (wbad) atomicOr(&flags->bad, 1u);
        if (wover) { atomicOr(&flags->over, 1u); atomicMax(&flags->maxid, mx); }
    }
}

// decode: persistent wavefronts, 64 records per tile: they come in as one block (nodes == NULL: rows w0 .. w0+63; the next tile's
// block is requested before this one is decoded) or record by record (a node list).
// Pass 1, lane = row: the set bits of the row's high words are enumerated 32 bits at a time (elias_fano.hpp:233-249) and element
// e's high part -- x >> l < 2 n <= 128: one byte -- goes to stage[row][e].  Pass 2, lane = (row of a group of four, four
// consecutive elements): the four high parts are one LDS dword, the low bits come from the LDS record (:235), value = high << l |
// low, -1 behind the row's edges, one 16-byte store per lane: 1 KiB of four finished rows per instruction.
template <int KP, int HWC, bool QUAD>
__global__ void __launch_bounds__(64) k_ef_rows_decode_tile(const uint64_t *__restrict__ arena, const uint2 *__restrict__ meta, uint64_t m,
                                                            const uint64_t *__restrict__ nodes, uint32_t K, uint32_t LW, uint32_t smagic,
                                                            int32_t *__restrict__ out, uint32_t *__restrict__ counts) {
    constexpr uint32_t SROW = KP + 4;  // bytes per stage row: a whole, odd number of dwords
    __shared__ __attribute__((aligned(16))) uint8_t stage[64 * SROW];
    __shared__ uint32_t rowmeta[64];
    extern __shared__ __attribute__((aligned(16))) uint64_t rec64[];  // 64 records of S | 1 words (+ 1 spill word)
    const uint32_t lane = lane_id();
    const uint64_t ntiles = (m + 63u) / 64u;
    const uint32_t S = LW + (uint32_t)HWC, S1 = S | 1u;
    const uint32_t step_rows = tile_div(128u, S, smagic), step_w = 128u - step_rows * S;
    uint4 pv[16];
    uint2 mt = make_uint2(0u, 0u);
    uint64_t tile = blockIdx.x;
    if (!nodes && tile < ntiles) {
        // the block of nrows * S words as 16-byte pieces (w0 * S words is a multiple of 64 words: aligned); S <= 32: <= 16 per lane
        const uint32_t nr = (uint32_t)(m - tile * 64u < 64u ? m - tile * 64u : 64u);
        tile_fetch16<16>((const uint4 *)(arena + tile * 64u * S), (nr * S + 1u) >> 1, pv);  // (odd: the last piece's second word is unused)
        if (lane < nr) mt = meta[tile * 64u + lane];
    }
    for (; tile < ntiles; tile += gridDim.x) {
        const uint64_t w0 = tile * 64u;
        const uint32_t nrows = (uint32_t)(m - w0 < 64u ? m - w0 : 64u);
        const bool have = lane < nrows;
        if (nodes) {
            const uint64_t row = have ? nodes[w0 + lane] : 0ull;
            mt = have ? meta[row] : make_uint2(0u, 0u);
            for (uint32_t w = 0; w < S; w++) rec64[lane * S1 + w] = have ? arena[row * S + w] : 0ull;
        } else {
            const uint32_t total = nrows * S, pieces = (total + 1u) >> 1;
            TileCursor c = tile_cursor(S, smagic, 2u * lane);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if ((uint32_t)i * 64u < pieces) {
                    const uint32_t f = 2u * ((uint32_t)i * 64u + lane);
                    if (f < total) rec64[c.row * S1 + c.w] = ((uint64_t)pv[i].y << 32) | pv[i].x;
                    if (f + 1u < total) {
                        const bool wrap = c.w + 1u == S;
                        rec64[(wrap ? c.row + 1u : c.row) * S1 + (wrap ? 0u : c.w + 1u)] = ((uint64_t)pv[i].w << 32) | pv[i].z;
                    }
                    tile_advance(c, S, step_rows, step_w);
                }
            }
        }
        uint32_t n = have ? mt.y & 0xffu : 0u;
        n = n < (uint32_t)KP ? n : (uint32_t)KP;
        rowmeta[lane] = n | (mt.y & 0x1f00u);
        if (!nodes) {
            const uint64_t nt = tile + gridDim.x;
            if (nt < ntiles) {
                const uint32_t nr = (uint32_t)(m - nt * 64u < 64u ? m - nt * 64u : 64u);
                tile_fetch16<16>((const uint4 *)(arena + nt * 64u * S), (nr * S + 1u) >> 1, pv);
                mt = lane < nr ? meta[nt * 64u + lane] : make_uint2(0u, 0u);
            }
        }
        wave_lds_sync();
        {
            uint32_t hd[2 * HWC];
#pragma unroll
            for (int w = 0; w < HWC; w++) {
                const uint64_t h = n ? rec64[lane * S1 + LW + (uint32_t)w] : 0ull;
                hd[2 * w] = (uint32_t)h;
                hd[2 * w + 1] = (uint32_t)(h >> 32);
            }
            uint8_t *sp = stage + lane * SROW;  // -> stage[row][e]
            uint32_t off = 0;                   // 32 j - e
            uint32_t left = n;                  // (a record holds exactly n set bits; `left` keeps a damaged one inside the stage)
#pragma unroll
            for (int j = 0; j < 2 * HWC; j++) {
                uint32_t cur = left ? hd[j] : 0u;
                while (ballot(cur != 0u)) {
                    if (cur != 0u) {
                        const uint32_t bit = (uint32_t)__builtin_ctz(cur);
                        cur &= cur - 1u;
                        *sp++ = (uint8_t)(off + bit);
                        off--;
                        left--;
                        cur = left ? cur : 0u;
                    }
                }
                off += 32u;
            }
        }
        wave_lds_sync();
        if (QUAD) {  // K % 4 == 0, out 16-byte aligned
            const uint32_t g = lane >> 4, q4 = (lane & 15u) * 4u;
            const uint32_t qs = q4 < (uint32_t)KP ? q4 : 0u;
#pragma unroll 2
            for (uint32_t r0 = 0; r0 < nrows; r0 += 4u) {
                const uint32_t rr = r0 + g;
                const uint32_t rm = rowmeta[rr];
                const uint32_t his = *(const uint32_t *)(stage + rr * SROW + qs);
                const uint32_t n_r = rm & 0xffu, l_r = rm >> 8;
                const uint32_t *rw = (const uint32_t *)(rec64 + rr * S1);
                const uint32_t lmask = (1u << l_r) - 1u;
                int32_t v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t e = q4 + (uint32_t)k;
                    const uint32_t bp = (e < n_r ? e : 0u) * l_r;  // (slots behind the row's edges read element 0 and discard it)
                    const uint32_t *q = rw + (bp >> 5);
                    const uint32_t low = __builtin_amdgcn_alignbit(q[1], q[0], bp & 31u) & lmask;
                    v[k] = e < n_r ? (int32_t)((((his >> (8 * k)) & 0xffu) << l_r) | low) : -1;
                }
                if (rr < nrows && q4 < K) *(int4 *)(out + (w0 + rr) * K + q4) = make_int4(v[0], v[1], v[2], v[3]);
            }
        } else {
            for (uint32_t rr = 0; rr < nrows; rr++) {
                const uint32_t rm = rowmeta[rr];
                const uint32_t n_r = rm & 0xffu, l_r = rm >> 8;
                if (lane < K) {
                    int32_t v = -1;
                    if (lane < n_r) {
                        const uint32_t hi = stage[rr * SROW + lane];
                        const uint32_t bp = lane * l_r;
                        const uint32_t *q = (const uint32_t *)(rec64 + rr * S1) + (bp >> 5);
                        v = (int32_t)((hi << l_r) | (__builtin_amdgcn_alignbit(q[1], q[0], bp & 31u) & ((1u << l_r) - 1u)));
                    }
                    out[(w0 + rr) * K + lane] = v;
                }
            }
        }
        if (counts && have) counts[w0 + lane] = n;
        wave_lds_sync();  // (the next tile's records go to the same LDS)
    }
}

// arena -> the CSR streams of a list object (ef_ensure_csr: export, save, decode_lists, get on a graph object)
__global__ void k_ef_arena_geom(const uint2 *meta, uint64_t N, uint32_t *sizes, uint32_t *lbits, uint64_t *universe, uint32_t *low_words,
                                uint32_t *high_words, uint32_t *nbatch) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 mt = meta[i];
        const uint32_t n = mt.y & 0xffu, l = (mt.y >> 8) & 0x1fu;
        uint32_t lw = 0, hw = 0;
        if (n) {
            lw = (n * l + 63u) / 64u + 1u;  // +1 padding word for read_bits
            hw = ((n + 1u) + (mt.x >> l) + 1u + 63u) / 64u;
        }
        sizes[i] = n; lbits[i] = l; universe[i] = mt.x; low_words[i] = lw; high_words[i] = hw; nbatch[i] = (hw + 63u) / 64u;
    }
}
__global__ void k_ef_arena_to_csr(const uint64_t *arena, const uint2 *meta, uint64_t N, uint32_t LW, uint32_t HW, const uint64_t *low_off,
                                  const uint64_t *high_off, uint64_t *low, uint64_t *high) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t *rec = arena + i * (LW + HW);
        const uint32_t lw = (uint32_t)(low_off[i + 1] - low_off[i]), hw = (uint32_t)(high_off[i + 1] - high_off[i]);
        for (uint32_t w = 0; w < lw; w++) low[low_off[i] + w] = w + 1u < lw ? rec[w] : 0ull;
        for (uint32_t w = 0; w < hw; w++) high[high_off[i] + w] = rec[LW + w];
    }
}


// random access (ef->select(offset), elias_fano.hpp:141-145): one wavefront per query; the select directory
// gives the batch of 64 high words holding the wanted one, a prefix scan of popcounts finds the word
__global__ void __launch_bounds__(64) k_ef_get(const uint64_t *low, const uint64_t *high, const uint64_t *low_off,
                                               const uint64_t *high_off, const uint32_t *lbits,
                                               const uint64_t *batch_off, const uint32_t *hrank, uint64_t m,
                                               const uint64_t *list_nos, const uint64_t *offs, int64_t *out) {
    const uint32_t lane = lane_id();
    for (uint64_t q = blockIdx.x; q < m; q += gridDim.x) {
        const uint64_t l = list_nos[q];
        const uint64_t i = offs[q];
        const uint32_t b = lbits[l];
        const uint64_t *hw = high + high_off[l];
        const uint64_t nhw = high_off[l + 1] - high_off[l];
        const uint32_t *hr = hrank + batch_off[l];
        const uint64_t nb = batch_off[l + 1] - batch_off[l];
        uint64_t lo = 0, hi = nb;  // largest batch with hrank <= i
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (hr[mid] <= i) lo = mid; else hi = mid;
        }
        const uint64_t w0 = lo * 64;
        const uint64_t wi = w0 + lane;
        const uint64_t word = wi < nhw ? hw[wi] : 0ull;
        const uint32_t c = popc64(word);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
            if (lane >= (uint32_t)o) incl += v;
        }
        const uint64_t need = i - hr[lo];  // 0-based inside this batch
        const uint64_t hit = ballot((uint64_t)incl > need);
        int64_t res = -1;
        if (hit) {
            const uint32_t wl_ = ff1(hit);  // lane wl_ owns the word; k-th set bit inside it
            uint32_t k = (uint32_t)need - (rl(incl, wl_) - rl(c, wl_));
            uint64_t ww = rl64((uint32_t)word, (uint32_t)(word >> 32), wl_);
            for (; k; k--) ww &= ww - 1;
            const uint64_t pos = (w0 + wl_) * 64 + (uint32_t)__builtin_ctzll(ww);
            res = (int64_t)(((pos - i) << b) | read_bits(low + low_off[l], i * b, b));
        }
        if (lane == 0) out[q] = res;
    }
}


// ---- per-list geometry and work-item tables, built on the device
// elias_fano.hpp:28-29 per list; word counts of the two streams and of the select directory
__global__ void k_ef_geom(const uint64_t *offsets, const PrepOut *prep, uint32_t nlist, uint32_t *lbits,
                          uint64_t *universe, uint32_t *low_words, uint32_t *high_words, uint32_t *nbatch,
                          EfTotals *tot) {
    unsigned long long bits = 0;
    unsigned int uns = 0;
    bool wide = false;  // an id beyond 32 bits or a list of 2^30 ids: the 64-bit encoder kernel is needed
    for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < nlist; l += gridDim.x * blockDim.x) {
        const uint64_t m = offsets[l + 1] - offsets[l];
        uint32_t lb = 0, lw = 0, hw = 0;
        uint64_t u = 0;
        if (m) {  // empty lists have no bitstream object (ef_bitstreams[list_no] stays null, :239-241)
            u = prep[l].max_id;
            wide |= (u >> 32) != 0 || m >= (1ull << 30);
            lb = (u / m) ? (uint32_t)msb64(u / m) : 0u;
            const uint64_t hb = (m + 1) + (u >> lb) + 1;
            bits += m * lb + hb;
            lw = (uint32_t)((m * lb + 63) / 64 + 1);  // +1 padding word for read_bits
            hw = (uint32_t)((hb + 63) / 64);
            uns += prep[l].unsorted ? 1u : 0u;
        }
        lbits[l] = lb; universe[l] = u; low_words[l] = lw; high_words[l] = hw; nbatch[l] = (hw + 63u) / 64u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bits += __shfl_xor(bits, o, 64);
        uns += (unsigned int)__shfl_xor((int)uns, o, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (bits) atomicAdd(&tot->total_bits, bits);
        if (uns) atomicAdd(&tot->n_unsorted, uns);
    }
    if (wide) atomicOr(&tot->pad, 1u);
}

}  // namespace

namespace {

template <typename T>
int ef_mirror(std::vector<T> &dst, const T *d_src, size_t count) {
    dst.resize(count);
    if (count) VIDC_HIP(hipMemcpy(dst.data(), d_src, count * sizeof(T), hipMemcpyDeviceToHost));
    return VIDC_OK;
}
// words of the two streams (a buffer allocated by a bound is longer than its stream; an empty stream is kept as one word)
inline uint64_t ef_low_count(const vidc_ef *e) { return e->n_low != ~0ull ? std::max<uint64_t>(e->n_low, 1) : e->d_low.n; }
inline uint64_t ef_high_count(const vidc_ef *e) { return e->n_high != ~0ull ? std::max<uint64_t>(e->n_high, 1) : e->d_high.n; }

// arena objects (graph rows): every host mirror at once from the per-row {universe, n | l << 8}
int ef_arena_mirror_locked(const vidc_ef *e) {
    if (e->offsets_host && e->meta_host) return VIDC_OK;
    VIDC_HIP(hipSetDevice(e->device));
    const uint64_t N = e->nlist;
    std::vector<uint2> mt(N);
    if (N) VIDC_HIP(hipMemcpy(mt.data(), e->d_rmeta.p, N * sizeof(uint2), hipMemcpyDeviceToHost));
    e->offsets.assign(N + 1, 0); e->low_off.assign(N + 1, 0); e->high_off.assign(N + 1, 0);
    e->lbits.assign(N, 0); e->universe.assign(N, 0); e->high_nbits.assign(N, 0);
    for (uint64_t i = 0; i < N; i++) {
        const uint32_t n = mt[i].y & 0xffu, l = (mt[i].y >> 8) & 0x1fu;
        uint64_t lw = 0, hw = 0;
        if (n) {
            e->high_nbits[i] = (uint64_t)(n + 1u) + (mt[i].x >> l) + 1u;
            lw = ((uint64_t)n * l + 63) / 64 + 1;  // (the CSR form keeps one padding word per list for read_bits)
            hw = (e->high_nbits[i] + 63) / 64;
        }
        e->lbits[i] = l; e->universe[i] = mt[i].x;
        e->offsets[i + 1] = e->offsets[i] + n;
        e->low_off[i + 1] = e->low_off[i] + lw;
        e->high_off[i + 1] = e->high_off[i] + hw;
    }
    e->offsets_host = e->meta_host = true;
    return VIDC_OK;
}
int ef_ensure_offsets_locked(const vidc_ef *e) {
    if (e->arena) return ef_arena_mirror_locked(e);
    if (e->offsets_host) return VIDC_OK;
    VIDC_HIP(hipSetDevice(e->device));
    VIDC_TRY(ef_mirror(e->offsets, (const uint64_t *)e->d_offsets.p, e->nlist + 1));
    e->offsets_host = true;
    return VIDC_OK;
}
int ef_ensure_offsets(const vidc_ef *e) {
    std::lock_guard<std::mutex> g(e->mu);
    return ef_ensure_offsets_locked(e);
}
int ef_ensure_meta(const vidc_ef *e) {
    std::lock_guard<std::mutex> g(e->mu);
    VIDC_TRY(ef_ensure_offsets_locked(e));
    if (e->meta_host) return VIDC_OK;
    VIDC_HIP(hipSetDevice(e->device));
    VIDC_TRY(ef_mirror(e->low_off, (const uint64_t *)e->d_low_off.p, e->nlist + 1));
    VIDC_TRY(ef_mirror(e->high_off, (const uint64_t *)e->d_high_off.p, e->nlist + 1));
    VIDC_TRY(ef_mirror(e->universe, (const uint64_t *)e->d_universe.p, e->nlist));
    VIDC_TRY(ef_mirror(e->lbits, (const uint32_t *)e->d_lbits.p, e->nlist));
    e->high_nbits.assign(e->nlist, 0);
    for (uint64_t l = 0; l < e->nlist; l++) {
        const uint64_t m = e->offsets[l + 1] - e->offsets[l];
        if (m) e->high_nbits[l] = (m + 1) + (e->universe[l] >> e->lbits[l]) + 1;
    }
    e->meta_host = true;
    return VIDC_OK;
}

// The per-list CSR streams of an arena object, built on the device the first time an entry point needs them (export / save,
// decode_lists, get, decode_all on a graph object: none of them is on the graph search path).
int ef_ensure_csr(vidc_ctx *ctx, const vidc_ef *ce) {
    if (!ce->arena) return VIDC_OK;
    std::lock_guard<std::mutex> g(ce->mu);
    if (ce->csr_ready) return VIDC_OK;
    vidc_ef *e = const_cast<vidc_ef *>(ce);
    VIDC_HIP(hipSetDevice(ctx->device));
    const uint64_t N = e->nlist;
    const uint32_t n32 = (uint32_t)N;
    Scratch s_cnt, s_lw, s_hw, s_nb, s_t0;
    Pinned tail;
    VIDC_TRY(tail.get(ctx, 64));
    unsigned long long *t = tail.as<unsigned long long>();
    VIDC_TRY(s_cnt.get(ctx, (N + 1) * 4)); VIDC_TRY(s_lw.get(ctx, (N + 1) * 4));
    VIDC_TRY(s_hw.get(ctx, (N + 1) * 4)); VIDC_TRY(s_nb.get(ctx, (N + 1) * 4));
    VIDC_TRY(e->d_offsets.alloc(N + 1, ctx->dpool));
    VIDC_TRY(e->d_low_off.alloc(N + 1, ctx->dpool)); VIDC_TRY(e->d_high_off.alloc(N + 1, ctx->dpool));
    VIDC_TRY(e->d_batch_off.alloc(N + 1, ctx->dpool));
    VIDC_TRY(e->d_lbits.alloc(N ? N : 1, ctx->dpool)); VIDC_TRY(e->d_universe.alloc(N ? N : 1, ctx->dpool));
    const dim3 grid((uint32_t)std::min<uint64_t>((N + 255) / 256 + 1, (uint64_t)ctx->num_cu * 16));
    hipLaunchKernelGGL(k_ef_arena_geom, grid, dim3(256), 0, ctx->stream, e->d_rmeta.p, N, s_cnt.as<uint32_t>(), e->d_lbits.p,
                       e->d_universe.p, s_lw.as<uint32_t>(), s_hw.as<uint32_t>(), s_nb.as<uint32_t>());
    {
        Scan4 sc;
        sc.in[0] = s_cnt.as<uint32_t>(); sc.in[1] = s_lw.as<uint32_t>(); sc.in[2] = s_hw.as<uint32_t>(); sc.in[3] = s_nb.as<uint32_t>();
        sc.out[0] = e->d_offsets.p; sc.out[1] = e->d_low_off.p; sc.out[2] = e->d_high_off.p; sc.out[3] = e->d_batch_off.p;
        VIDC_TRY(device_exscan4(ctx, sc, 4, n32, s_t0));
    }
    VIDC_HIP(hipMemcpyAsync(t + 1, e->d_low_off.p + N, 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(t + 2, e->d_high_off.p + N, 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(t + 3, e->d_batch_off.p + N, 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    const uint64_t low_words = t[1], high_words = t[2];
    e->nbatches = t[3];
    VIDC_TRY(e->d_low.alloc(low_words ? low_words : 1, ctx->dpool));
    VIDC_TRY(e->d_high.alloc(high_words ? high_words : 1, ctx->dpool));
    VIDC_TRY(e->d_chunks.alloc(1, ctx->dpool));  // chunk table: encoder-only
    VIDC_TRY(e->d_batches.alloc(e->nbatches ? e->nbatches : 1, ctx->dpool));
    VIDC_TRY(e->d_hrank.alloc(e->nbatches ? e->nbatches : 1, ctx->dpool));
    // a row's high stream is a handful of words: one batch per row, no element before it
    VIDC_HIP(hipMemsetAsync(e->d_hrank.p, 0, (e->nbatches ? e->nbatches : 1) * 4, ctx->stream));
    if (!low_words) VIDC_HIP(hipMemsetAsync(e->d_low.p, 0, 8, ctx->stream));
    if (!high_words) VIDC_HIP(hipMemsetAsync(e->d_high.p, 0, 8, ctx->stream));
    if (N) {
        hipLaunchKernelGGL(k_ef_arena_to_csr, grid, dim3(256), 0, ctx->stream, e->d_arena.p, e->d_rmeta.p, N, e->a_lw, e->a_hw,
                           e->d_low_off.p, e->d_high_off.p, e->d_low.p, e->d_high.p);
        if (e->nbatches)
            launch_fill_items(ctx->stream, e->d_batch_off.p, n32, 1u, e->d_batches.p, e->nbatches, (uint32_t)ctx->num_cu);
    }
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    e->csr_ready = true;
    return VIDC_OK;
}

// the three-pass encoder for objects with lists that are not ascending (max + order check, sort, streams):
// e->d_offsets, nlist, ntotal are set
int ef_encode_general(vidc_ctx *ctx, vidc_ef *e, const uint64_t *d_ids, uint32_t flags) {
    const uint64_t nlist = e->nlist;
    const uint32_t nl32 = (uint32_t)nlist;
    double kernel_ms = 0;
    VidcPhaseTimer pt(ctx);  // phases are timed without a host synchronisation each
    auto timed = [&](auto &&fn) -> int {
        pt.begin();
        fn();
        VIDC_HIP(hipGetLastError());
        pt.end();
        return VIDC_OK;
    };
    const uint32_t lgrid = (uint32_t)std::min<uint64_t>((nlist + 255) / 256 + 1, 2048);
    Scratch s_cnt, s_coff, s_tmp, s_prep, s_lw, s_hw, s_nb, s_tot, s_tmp2;
    Pinned tail;
    VIDC_TRY(tail.get(ctx, 64));
    unsigned long long *t = tail.as<unsigned long long>();

    // chunk table (one wavefront of the streaming kernels per 512 ids)
    VIDC_TRY(s_cnt.get(ctx, (nlist + 1) * 4));
    VIDC_TRY(s_coff.get(ctx, (nlist + 1) * 8));
    hipLaunchKernelGGL(k_count_chunks, dim3(lgrid), dim3(256), 0, ctx->stream, e->d_offsets.p, nl32, s_cnt.as<uint32_t>());
    VIDC_TRY(device_exscan(ctx, s_cnt.as<uint32_t>(), nl32, s_coff.as<uint64_t>(), s_tmp));
    VIDC_HIP(hipMemcpyAsync(t, s_coff.as<uint64_t>() + nlist, 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    e->nchunks = t[0];
    VIDC_TRY(e->d_chunks.alloc(e->nchunks ? e->nchunks : 1, ctx->dpool));
    if (e->nchunks)
        launch_fill_items(ctx->stream, s_coff.as<uint64_t>(), nl32, CHUNK_IDS, e->d_chunks.p, e->nchunks, (uint32_t)ctx->num_cu);
    VIDC_HIP(hipGetLastError());
    const uint32_t cgrid = (uint32_t)std::min<uint64_t>(e->nchunks ? e->nchunks : 1, (uint64_t)ctx->num_cu * 256);

    // pass 1: universe (max id) and sortedness per list, then the geometry (elias_fano.hpp:28-29)
    VIDC_TRY(s_prep.get(ctx, (nlist + 1) * sizeof(PrepOut)));
    VIDC_TRY(s_lw.get(ctx, (nlist + 1) * 4)); VIDC_TRY(s_hw.get(ctx, (nlist + 1) * 4)); VIDC_TRY(s_nb.get(ctx, (nlist + 1) * 4));
    VIDC_TRY(s_tot.get(ctx, sizeof(EfTotals)));
    VIDC_TRY(e->d_low_off.alloc(nlist + 1, ctx->dpool)); VIDC_TRY(e->d_high_off.alloc(nlist + 1, ctx->dpool));
    VIDC_TRY(e->d_batch_off.alloc(nlist + 1, ctx->dpool));
    VIDC_TRY(e->d_lbits.alloc(nlist ? nlist : 1, ctx->dpool)); VIDC_TRY(e->d_universe.alloc(nlist ? nlist : 1, ctx->dpool));
    VIDC_HIP(hipMemsetAsync(s_prep.p, 0, (nlist + 1) * sizeof(PrepOut), ctx->stream));
    VIDC_HIP(hipMemsetAsync(s_tot.p, 0, sizeof(EfTotals), ctx->stream));
    if (e->nchunks)
        VIDC_TRY(timed([&] {
            hipLaunchKernelGGL(k_ef_prep, dim3(cgrid), dim3(64), 0, ctx->stream, d_ids, e->d_offsets.p, e->d_chunks.p,
                               e->nchunks, s_prep.as<PrepOut>());
        }));
    hipLaunchKernelGGL(k_ef_geom, dim3(lgrid), dim3(256), 0, ctx->stream, e->d_offsets.p, s_prep.as<PrepOut>(), nl32,
                       e->d_lbits.p, e->d_universe.p, s_lw.as<uint32_t>(), s_hw.as<uint32_t>(), s_nb.as<uint32_t>(),
                       s_tot.as<EfTotals>());
    {
        Scan4 sc;
        sc.in[0] = s_lw.as<uint32_t>(); sc.in[1] = s_hw.as<uint32_t>(); sc.in[2] = s_nb.as<uint32_t>(); sc.in[3] = nullptr;
        sc.out[0] = e->d_low_off.p; sc.out[1] = e->d_high_off.p; sc.out[2] = e->d_batch_off.p; sc.out[3] = nullptr;
        VIDC_TRY(device_exscan4(ctx, sc, 3, nl32, s_tmp2));
    }
    VIDC_HIP(hipMemcpyAsync(t + 0, e->d_low_off.p + nlist, 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(t + 1, e->d_high_off.p + nlist, 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(t + 2, e->d_batch_off.p + nlist, 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(t + 3, s_tot.p, sizeof(EfTotals), hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    const uint64_t low_words = t[0], high_words = t[1];
    e->nbatches = t[2];
    e->total_bits = t[3];
    const uint32_t n_unsorted = (uint32_t)(t[4] & 0xffffffffu);
    const bool wide_ids = (t[4] >> 32) != 0;
    e->narrow = !wide_ids;
    VIDC_TRY(e->d_low.alloc(low_words ? low_words : 1, ctx->dpool));
    VIDC_TRY(e->d_high.alloc(high_words ? high_words : 1, ctx->dpool));
    VIDC_HIP(hipMemsetAsync(e->d_high.p, 0, (high_words ? high_words : 1) * 8, ctx->stream));

    // pass 2 (only when some list is not ascending): sorted copy + permutation
    const uint64_t *d_sorted = d_ids;
    Scratch s_sorted, s_keys, s_kpos, s_key_off, s_lists;
    const bool want_perm = (flags & VIDC_EF_WANT_PERM) != 0;
    if (want_perm || n_unsorted) {
        VIDC_TRY(e->d_perm.alloc(e->ntotal ? e->ntotal : 1, ctx->dpool));
        e->has_perm = true;
        if (e->ntotal) {
            uint32_t grid = (uint32_t)std::min<uint64_t>((e->ntotal + 255) / 256, (uint64_t)ctx->num_cu * 32);
            VIDC_TRY(timed([&] {
                hipLaunchKernelGGL(k_iota_perm, dim3(grid), dim3(256), 0, ctx->stream, e->d_offsets.p, nl32, e->ntotal,
                                   e->d_perm.p);
            }));
        }
    }
    if (n_unsorted) {
        // rare (Faiss lists are in add order): the list of unsorted lists is built on the host
        VIDC_TRY(ef_ensure_offsets(e));
        std::vector<PrepOut> prep(nlist);
        VIDC_HIP(hipMemcpy(prep.data(), s_prep.p, nlist * sizeof(PrepOut), hipMemcpyDeviceToHost));
        std::vector<uint32_t> unsorted;
        for (uint64_t l = 0; l < nlist; l++)
            if (e->offsets[l + 1] > e->offsets[l] && prep[l].unsorted) unsorted.push_back((uint32_t)l);
        std::stable_sort(unsorted.begin(), unsorted.end(), [&](uint32_t a, uint32_t b) {
            return e->offsets[a + 1] - e->offsets[a] > e->offsets[b + 1] - e->offsets[b];
        });
        std::vector<uint64_t> key_off(unsorted.size() + 1, 0);
        for (size_t i = 0; i < unsorted.size(); i++) {
            uint64_t n = e->offsets[unsorted[i] + 1] - e->offsets[unsorted[i]], p2 = 1;
            while (p2 < n) p2 <<= 1;
            key_off[i + 1] = key_off[i] + p2;
        }
        VIDC_TRY(s_sorted.get(ctx, e->ntotal * 8));
        VIDC_HIP(hipMemcpyAsync(s_sorted.p, d_ids, e->ntotal * 8, hipMemcpyDeviceToDevice, ctx->stream));
        VIDC_TRY(s_keys.get(ctx, key_off.back() * 8));
        VIDC_TRY(s_kpos.get(ctx, key_off.back() * 4));
        VIDC_TRY(s_key_off.get(ctx, key_off.size() * 8));
        VIDC_TRY(s_lists.get(ctx, unsorted.size() * 4));
        VIDC_HIP(hipMemcpyAsync(s_key_off.p, key_off.data(), key_off.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        VIDC_HIP(hipMemcpyAsync(s_lists.p, unsorted.data(), unsorted.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        VIDC_TRY(timed([&] {
            hipLaunchKernelGGL(k_ef_sort, dim3((uint32_t)unsorted.size()), dim3(64), 0, ctx->stream, d_ids, e->d_offsets.p,
                               s_lists.as<uint32_t>(), (uint32_t)unsorted.size(), s_key_off.as<uint64_t>(),
                               s_keys.as<uint64_t>(), s_kpos.as<uint32_t>(), s_sorted.as<uint64_t>(), e->d_perm.p);
        }));
        d_sorted = s_sorted.as<uint64_t>();
    }
    // pass 3: the two bit streams
    if (e->ntotal) {
        // (a memset of the whole low stream -- 2 bytes per id -- cost more than the geometry kernels together)
        hipLaunchKernelGGL(k_ef_zero_low_pads, dim3((uint32_t)std::min<uint64_t>((nlist + 255) / 256, 2048)), dim3(256), 0,
                           ctx->stream, e->d_low_off.p, (uint32_t)nlist, e->d_low.p);
        VIDC_TRY(timed([&] {
            hipLaunchKernelGGL(k_ef_low, dim3(cgrid), dim3(64), 0, ctx->stream, d_sorted, e->d_offsets.p,
                               e->d_low_off.p, e->d_lbits.p, e->d_chunks.p, e->nchunks, e->d_low.p);
            hipLaunchKernelGGL(k_ef_high, dim3(cgrid), dim3(64), 0, ctx->stream, d_sorted, e->d_offsets.p,
                               e->d_high_off.p, e->d_lbits.p, e->d_chunks.p, e->nchunks, e->d_high.p);
        }));
    }
    // select directory
    VIDC_TRY(e->d_batches.alloc(e->nbatches ? e->nbatches : 1, ctx->dpool));
    VIDC_TRY(e->d_hrank.alloc(e->nbatches ? e->nbatches : 1, ctx->dpool));
    if (e->nbatches) {
        VIDC_TRY(timed([&] {
            launch_fill_items(ctx->stream, e->d_batch_off.p, nl32, 1u, e->d_batches.p, e->nbatches, (uint32_t)ctx->num_cu);
            hipLaunchKernelGGL(k_ef_hrank, dim3((uint32_t)std::min<uint64_t>((e->nbatches + 255) / 256, 4096)), dim3(256),
                               0, ctx->stream, d_sorted, e->d_offsets.p, e->d_lbits.p, e->d_batches.p, e->nbatches,
                               e->d_hrank.p);
        }));
    }
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    kernel_ms += pt.collect();
    ctx->last_kernel_ms = kernel_ms;
    return VIDC_OK;
}

// the single-pass encoder (kernels above); *retry is raised when some list turned out not to be ascending.
// nchunks: number of EF_CHUNK-sized pieces of all lists (the caller walks the host offsets anyway)
// spec: the streams are allocated BEFORE the geometry kernels run, by bounds that need only the list lengths and a guess of the
// largest universe U (ids of an index are 0 .. ntotal-1 unless the caller numbers them himself; the context remembers a larger
// one it has met):   low  <= sum n_i msb(U / n_i) <= ntotal log2(U nlist / ntotal) bits (n log(U / n) is concave) + 2 words per list,
//                    high  = (n + 1) + (u >> l) + 1 <= 3 n + 1 bits per list (u >> l < 2 n) whatever U is.
// Geometry and chunk kernels are then queued back to back and the call waits ONCE, at its end (before: a wait between them for the
// sizes, 20-25 us of an S1-sized call's 60).  k_ef_offsets checks the exact sizes against the allocation on the device; if they
// do not fit (a universe beyond the guess, ids >= 2^32) it raises *abort, the chunk kernels return at once, *respec is set and the
// caller runs the call again the old way.  Cost of the bound: the buffers are up to ~15 % larger than the streams (n_low / n_high
// hold the exact word counts).
int ef_encode_fast(vidc_ctx *ctx, vidc_ef *e, const uint64_t *d_ids, uint32_t flags, uint64_t nchunks, uint64_t max_list,
                   bool *retry, bool spec, bool *respec) {
    const uint64_t nlist = e->nlist;
    const uint32_t nl32 = (uint32_t)nlist;
    // (one tile only for objects whose chunk records one workgroup can write: 1024 lists of a million ids each are two million records,
    // and spreading those over the GPU is what k_ef_big_recs is for)
    const bool single = nl32 <= EF_SINGLE_LISTS && nchunks <= EF_SINGLE_MAX_CHUNKS;  // (one tile, the geometry kernel alone)
    const uint32_t tile_lists = single ? EF_SINGLE_LISTS : (nl32 <= EF_E1_MAX_LISTS ? 256u : 1024u);
    const uint32_t ntiles = nl32 ? (nl32 + tile_lists - 1u) / tile_lists : 1u;
    VidcPhaseTimer pt(ctx);
    HostTrace tr("ef encode_fast");
    Scratch s_raw, s_tiles, s_recs, s_big;
    Pinned tail;
    VIDC_TRY(tail.get(ctx, 2 * sizeof(EfSummary)));
    EfSummary *hs = tail.as<EfSummary>();
    VIDC_TRY(s_raw.get(ctx, (nlist + 1) * sizeof(EfRaw)));
    VIDC_TRY(s_tiles.get(ctx, (size_t)ntiles * sizeof(EfTile)));
    VIDC_TRY(s_recs.get(ctx, (nchunks ? nchunks : 1) * sizeof(EfChunkRec)));
    // long lists (more than EF_BIG_CHUNKS chunks: at most nchunks / (EF_BIG_CHUNKS + 1) of them) + their count
    const uint64_t big_cap = nchunks / (EF_BIG_CHUNKS + 1u) + 1u;
    VIDC_TRY(s_big.get(ctx, big_cap * sizeof(EfBigList) + 16));
    EfBigList *d_big = (EfBigList *)((char *)s_big.p + 16);
    uint32_t *d_nbig = s_big.as<uint32_t>();
    uint32_t *d_abort = d_nbig + 1;  // (written by k_ef_offsets whenever there are lists)
    if (!single || !nlist) VIDC_HIP(hipMemsetAsync(d_nbig, 0, 16, ctx->stream));  // (a single tile writes every record itself)
    EfLimits lim{~0ull, ~0ull, ~0ull, 1u, 0u};
    if (spec) {
        const uint64_t U = std::max<uint64_t>(e->ntotal ? e->ntotal - 1 : 0, ctx->ef_universe_hint);
        const double ratio = e->ntotal ? (double)U * (double)nlist / (double)e->ntotal : 0.0;
        const double low_bits = ratio > 1.0 ? (double)e->ntotal * std::log2(ratio) * 1.0001 : 0.0;
        lim.low_words = (uint64_t)(low_bits / 64.0) + 2 * nlist + 4;
        lim.high_words = (3 * e->ntotal + nlist) / 64 + nlist + 2;
        lim.nbatches = lim.high_words / 64 + nlist + 2;
        lim.wide_ok = 0u;
    }
    VIDC_TRY(e->d_low_off.alloc(nlist + 1, ctx->dpool)); VIDC_TRY(e->d_high_off.alloc(nlist + 1, ctx->dpool));
    VIDC_TRY(e->d_batch_off.alloc(nlist + 1, ctx->dpool));
    VIDC_TRY(e->d_lbits.alloc(nlist ? nlist : 1, ctx->dpool)); VIDC_TRY(e->d_universe.alloc(nlist ? nlist : 1, ctx->dpool));
    VIDC_TRY(e->d_chunks.alloc(1, ctx->dpool));  // (the chunk table of the three-pass encoder: not needed here)
    tr.mark("scratch + geometry arrays");
    pt.begin();
    if (single) {
        hipLaunchKernelGGL((k_ef_offsets<true, 2, 512>), dim3(1), dim3(512), 0, ctx->stream, d_ids, e->d_offsets.p, nl32,
                           e->d_lbits.p, e->d_universe.p, (const EfRaw *)nullptr, (const EfTile *)nullptr, 1u,
                           e->d_low_off.p, e->d_high_off.p, e->d_batch_off.p, s_recs.as<EfChunkRec>(),
                           hs, d_big, d_nbig, lim, d_abort);
    } else if (tile_lists == 256u) {
        hipLaunchKernelGGL(k_ef_meta<1>, dim3(ntiles), dim3(256), 0, ctx->stream, d_ids, e->d_offsets.p, nl32,
                           e->d_lbits.p, e->d_universe.p, s_raw.as<EfRaw>(), s_tiles.as<EfTile>());
        hipLaunchKernelGGL((k_ef_offsets<false, 1, 256>), dim3(ntiles), dim3(256), 0, ctx->stream, d_ids, e->d_offsets.p,
                           nl32, e->d_lbits.p, e->d_universe.p, s_raw.as<EfRaw>(), s_tiles.as<EfTile>(), ntiles,
                           e->d_low_off.p, e->d_high_off.p, e->d_batch_off.p, s_recs.as<EfChunkRec>(),
                           hs, d_big, d_nbig, lim, d_abort);
    } else {
        hipLaunchKernelGGL(k_ef_meta<4>, dim3(ntiles), dim3(256), 0, ctx->stream, d_ids, e->d_offsets.p, nl32,
                           e->d_lbits.p, e->d_universe.p, s_raw.as<EfRaw>(), s_tiles.as<EfTile>());
        hipLaunchKernelGGL((k_ef_offsets<false, 4, 256>), dim3(ntiles), dim3(256), 0, ctx->stream, d_ids, e->d_offsets.p,
                           nl32, e->d_lbits.p, e->d_universe.p, s_raw.as<EfRaw>(), s_tiles.as<EfTile>(), ntiles,
                           e->d_low_off.p, e->d_high_off.p, e->d_batch_off.p, s_recs.as<EfChunkRec>(),
                           hs, d_big, d_nbig, lim, d_abort);
    }
    if (!single && max_list > (uint64_t)EF_CHUNK * EF_BIG_CHUNKS)  // (some list is that long)
        hipLaunchKernelGGL(k_ef_big_recs, dim3((uint32_t)std::min<uint64_t>(big_cap, (uint64_t)ctx->num_cu * 32)), dim3(64), 0, ctx->stream,
                           d_big, d_nbig, s_recs.as<EfChunkRec>());
    VIDC_HIP(hipGetLastError());
    // (ONE pair of timing events around the call's kernels: every record is a packet the queue processes between two kernels)
    // (the geometry kernel stores its summary into the pinned block itself, and the chunk kernels their "not ascending" flag: a copy
    // engine between a kernel and the host's wake-up costs more than these kernels)
    tr.mark("geometry kernels queued");
    uint64_t low_words = lim.low_words, high_words = lim.high_words;  // words to allocate
    bool wide_ids = false;
    e->nchunks = nchunks;
    e->nbatches = lim.nbatches;  // (spec: an upper bound until the call's only wait)
    if (!spec) {
        VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
        if (hs->nchunks != nchunks) {
            set_error("elias-fano encoder: chunk count mismatch (%llu vs %llu)", (unsigned long long)hs->nchunks,
                      (unsigned long long)nchunks);
            return VIDC_ERR_OVERFLOW;
        }
        low_words = hs->low_words; high_words = hs->high_words;
        e->nbatches = hs->nbatches;
        e->total_bits = hs->total_bits;
        wide_ids = hs->wide != 0;
        e->narrow = !wide_ids;
    }
    VIDC_TRY(e->d_low.alloc(low_words ? low_words : 1, ctx->dpool));
    VIDC_TRY(e->d_high.alloc(high_words ? high_words : 1, ctx->dpool));
    VIDC_TRY(e->d_batches.alloc(e->nbatches ? e->nbatches : 1, ctx->dpool));
    VIDC_TRY(e->d_hrank.alloc(e->nbatches ? e->nbatches : 1, ctx->dpool));
    // The chunk kernels write every word of the high stream (k_ef_lowhigh), so nothing has to zero it.  A stream that does not
    // fit the caches is cleared all the same, inside the timed region: the kernels store a chunk's 100-200 bytes at a time, lines shared
    // between wavefronts reach HBM as partial writes, and the same kernel runs 8 % faster on lines a fill has just left in the
    // memory-side cache (S2, 250 MB of high words: 2.95 + 0.08 ms instead of 3.19; 16 M ids, 4 MB: 68.7 instead of 65.8 + 4.5 us).
    // VIDC_EF_MEMSET=1 / 0 (measurements): always / never.
    const char *ms_env = std::getenv("VIDC_EF_MEMSET");
    const bool clear_high = ms_env ? ms_env[0] == '1' : high_words * 8 > EF_CLEAR_HIGH_BYTES;
    // Objects of up to 2^18 batches get their bulk-decode records (EfRec) from the encoder: the wavefront that writes a batch's
    // directory entry holds everything the record needs, and the first decode_all of a fresh object is one launch instead of
    // k_ef_build_recs + decode (16 M ids in lists of 256: 53 -> ~47 us; S1-sized: 20 -> ~15 us).  Larger objects keep the lazy
    // build: it also finds the exact size of the largest batch, which sets the occupancy of their millisecond-long decode.
    EfRec *d_drecs = nullptr;
    const bool enc_recs = nchunks && max_list && e->nbatches && e->nbatches < EF_ENC_RECS_MAX && !std::getenv("VIDC_EF_LAZY_RECS");
    if (enc_recs) {
        std::lock_guard<std::mutex> g(e->mu);
        e->recs_ready = false;
        VIDC_TRY(e->d_recs.alloc(e->nbatches + 1, ctx->dpool));
        d_drecs = e->d_recs.p;
    }
    if ((flags & VIDC_EF_WANT_PERM) != 0) {
        VIDC_TRY(e->d_perm.alloc(e->ntotal ? e->ntotal : 1, ctx->dpool));
        e->has_perm = true;
        if (e->ntotal) {
            const uint32_t grid = (uint32_t)std::min<uint64_t>((e->ntotal + 255) / 256, (uint64_t)ctx->num_cu * 32);
            hipLaunchKernelGGL(k_iota_perm, dim3(grid), dim3(256), 0, ctx->stream, e->d_offsets.p, nl32, e->ntotal,
                               e->d_perm.p);
        }
    }
    if (nchunks) {
        const uint32_t cgrid = (uint32_t)std::min<uint64_t>(nchunks, (uint64_t)ctx->num_cu * 256);
        hs[1].unsorted = 0u;
        uint32_t *d_flag = &hs[1].unsorted;
        if (clear_high) VIDC_HIP(hipMemsetAsync(e->d_high.p, 0, high_words * 8, ctx->stream));
        if (wide_ids)
            hipLaunchKernelGGL(k_ef_lowhigh, dim3(cgrid), dim3(64), 0, ctx->stream, d_ids, s_recs.as<EfChunkRec>(), nchunks,
                               e->d_low.p, e->d_high.p, e->d_hrank.p, e->d_batches.p, d_flag, d_drecs, d_abort);
        else if (max_list <= 256)
            hipLaunchKernelGGL((k_ef_lowhigh32<4, true, true, true>), dim3(cgrid), dim3(64), 0, ctx->stream, d_ids, s_recs.as<EfChunkRec>(),
                               nchunks, e->d_low.p, e->d_high.p, e->d_hrank.p, e->d_batches.p, d_flag, d_drecs, d_abort);
        else if (e->ntotal < 256 * nchunks)  // (chunks half full on average)
            hipLaunchKernelGGL((k_ef_lowhigh32<EF_CHUNK / 64, true, true, true>), dim3(cgrid), dim3(64), 0, ctx->stream, d_ids,
                               s_recs.as<EfChunkRec>(), nchunks, e->d_low.p, e->d_high.p, e->d_hrank.p, e->d_batches.p, d_flag, d_drecs, d_abort);
        else
            hipLaunchKernelGGL((k_ef_lowhigh32<EF_CHUNK / 64, false, true, true>), dim3(cgrid), dim3(64), 0, ctx->stream, d_ids,
                               s_recs.as<EfChunkRec>(), nchunks, e->d_low.p, e->d_high.p, e->d_hrank.p, e->d_batches.p, d_flag, d_drecs, d_abort);
        VIDC_HIP(hipGetLastError());
    }
    pt.end();
    tr.mark("streams allocated, chunk kernels queued");
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    tr.mark("wait");
    ctx->last_kernel_ms = pt.collect();
    tr.mark("event times");
    if (spec) {
        if (hs->nchunks != nchunks) {
            set_error("elias-fano encoder: chunk count mismatch (%llu vs %llu)", (unsigned long long)hs->nchunks,
                      (unsigned long long)nchunks);
            return VIDC_ERR_OVERFLOW;
        }
        if (hs->low_words > lim.low_words || hs->high_words > lim.high_words || hs->nbatches > lim.nbatches || hs->wide) {
            *respec = true;  // (the chunk kernels saw the same verdict in *abort and wrote nothing)
            return VIDC_OK;
        }
        e->nbatches = hs->nbatches;
        e->total_bits = hs->total_bits;
        e->narrow = true;
        e->n_low = hs->low_words; e->n_high = hs->high_words;
    }
    if (nchunks && hs[1].unsorted) *retry = true;
    if (enc_recs && !*retry) {
        std::lock_guard<std::mutex> g(e->mu);
        e->recs_max_cnt = (uint32_t)std::min<uint64_t>(max_list, EF_BATCH_BITS);
        e->recs_ready = true;
    }
    return VIDC_OK;
}

}  // namespace

extern "C" {

int vidc_ef_encode(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids, uint32_t flags,
                   vidc_ef **out) {
    if (!ctx || !out || (nlist && !offsets)) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (nlist >= 0xffffffffull) return VIDC_ERR_INVALID;
    HostTrace tr("ef encode");
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_ef> e(new vidc_ef());
    e->device = ctx->device;
    e->nlist = nlist;
    // (the host mirror of the offsets is filled from the device array the first time an entry point needs it: copying 8 bytes per list
    // here was a sixth of the host side of a 65 536-list call)
    e->ntotal = nlist ? offsets[nlist] : 0;
    // the offsets go to the device first (pinned staging block, asynchronous copy); the host's pass over them runs while they cross PCIe
    VIDC_TRY(e->d_offsets.alloc(nlist + 1, ctx->dpool));
    Pinned h_off;
    VIDC_TRY(h_off.get(ctx, (nlist + 1) * 8));
    struct SyncOnExit {  // (an early return must not release the staging block of a copy in flight)
        vidc_ctx *c;
        bool armed = true;
        ~SyncOnExit() { if (armed) (void)vidc::vidc_stream_wait(c->stream); }
    } guard{ctx};
    if (nlist) std::memcpy(h_off.p, offsets, (nlist + 1) * 8);
    else *h_off.as<uint64_t>() = 0;
    VIDC_HIP(hipMemcpyAsync(e->d_offsets.p, h_off.p, (nlist + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    tr.mark("offsets staged");
    uint64_t nchunks = 0, max_list = 0;
    static_assert((EF_CHUNK & (EF_CHUNK - 1u)) == 0u, "the pass below counts chunks by a shift");
    const LengthsPass lp = lengths_pass(offsets, nlist, (uint32_t)__builtin_ctz(EF_CHUNK), 0u);  // (four lists per instruction where the host can)
    if (!lp.wide && lp.max_n <= 0xfffffff0ull) { nchunks = lp.nchunks; max_list = lp.max_n; }
    else
    for (uint64_t l = 0; l < nlist; l++) {  // (some length of 2^32 or more, or offsets that decrease: list by list, with the message)
        if (offsets[l + 1] < offsets[l] || offsets[l + 1] - offsets[l] > 0xfffffff0ull) {
            set_error("bad offsets at list %llu", (unsigned long long)l);
            return VIDC_ERR_INVALID;
        }
        nchunks += (offsets[l + 1] - offsets[l] + EF_CHUNK - 1) / EF_CHUNK;
        max_list = std::max(max_list, offsets[l + 1] - offsets[l]);
    }
    if (e->ntotal && !d_ids) return VIDC_ERR_INVALID;
    tr.mark("host offsets pass");
    bool retry = false;
    e->max_list = max_list;
    tr.mark("offsets staged");
    static const bool no_spec = std::getenv("VIDC_EF_NO_SPEC") != nullptr;  // (measurement switch: always size the streams after the geometry kernels)
    bool respec = false;
    const bool spec = !no_spec && e->ntotal && e->ntotal <= 0xffffffffull;
    VIDC_TRY(ef_encode_fast(ctx, e.get(), d_ids, flags, nchunks, max_list, &retry, spec, &respec));
    if (respec) {  // the guess of the universe was too small: size the streams from the ids, and remember how large they are
        e->d_recs.release();
        e->recs_ready = false;
        e->has_perm = false;
        e->n_low = e->n_high = ~0ull;
        VIDC_TRY(ef_encode_fast(ctx, e.get(), d_ids, flags, nchunks, max_list, &retry, false, &respec));
        std::vector<uint64_t> uni(nlist);
        VIDC_HIP(hipMemcpy(uni.data(), e->d_universe.p, nlist * 8, hipMemcpyDeviceToHost));
        for (uint64_t u : uni) ctx->ef_universe_hint = std::max(ctx->ef_universe_hint, u);
    }
    guard.armed = false;  // (ef_encode_fast has waited)
    tr.mark("encode_fast (launches + wait)");
    if (retry) {  // some list is not ascending: general three-pass encoder with the sort
        e->offsets = vec_pool<uint64_t>().take(nlist + 1);
        if (nlist) e->offsets.assign(offsets, offsets + nlist + 1);
        else e->offsets.assign(1, 0);
        e->offsets_host = true;
        e->n_low = e->n_high = ~0ull;
        e->total_bits = 0;
        e->has_perm = false;
        e->recs_ready = false;  // (the general encoder's objects build their records on the first bulk decode)
        e->d_recs.release();
        VIDC_TRY(ef_encode_general(ctx, e.get(), d_ids, flags));
    }
    *out = e.release();
    return VIDC_OK;
}

void vidc_ef_destroy(vidc_ef *e) { delete e; }

// compressed_ids_size_in_bytes = (sum of low.size() + high.size() in bits) / 8, custom_invlists_impl.cpp:272-282
uint64_t vidc_ef_compressed_bytes(const vidc_ef *e) { return e ? e->total_bits / 8 : 0; }

int vidc_ef_list_info(const vidc_ef *e, uint32_t *sizes, uint32_t *low_bits, uint64_t *universes) {
    if (!e) return VIDC_ERR_INVALID;
    VIDC_TRY(ef_ensure_meta(e));
    for (uint64_t l = 0; l < e->nlist; l++) {
        if (sizes) sizes[l] = (uint32_t)(e->offsets[l + 1] - e->offsets[l]);
        if (low_bits) low_bits[l] = e->lbits[l];
        if (universes) universes[l] = e->universe[l];
    }
    return VIDC_OK;
}

int vidc_ef_decode_all(vidc_ctx *ctx, const vidc_ef *e, uint64_t *d_out) {
    if (!ctx || !e || (e->ntotal && !d_out)) return VIDC_ERR_INVALID;
    if (!e->ntotal) return VIDC_OK;
    VIDC_TRY(ef_ensure_csr(ctx, e));
    VIDC_HIP(hipSetDevice(ctx->device));
    double recs_ms = 0;
    bool recs_timed = false, publish_after_sync = false;
    uint32_t pending_max_cnt = 0;
    std::unique_lock<std::mutex> g(e->mu, std::defer_lock);
    if (e->nbatches) {  // batch records: built once per object, on its first bulk decode (its time is part of that decode's)
        g.lock();
        if (!e->recs_ready) {
            VIDC_TRY(e->d_recs.alloc(e->nbatches + 1, ctx->dpool));  // (+1: see EfRec)
            // The decode kernel's LDS table is sized by the largest batch.  A small object whose longest list bounds it well
            // (no batch holds more elements than its list) takes that bound: the records are built and used back to back,
            // without the read-back of the exact maximum and its synchronisation (S1-sized calls: two launches of ~5 + ~12 us
            // instead of fill + build + copy + wait + decode).  Large objects keep the exact value (occupancy of a 2 ms kernel).
            const bool bounded = e->max_list && e->nbatches < EF_ENC_RECS_MAX;
            VIDC_HIP(hipEventRecord(ctx->ev_chain[0], ctx->stream));
            Scratch s_mx;
            Pinned h_mx;
            if (!bounded) {
                VIDC_TRY(s_mx.get(ctx, 16));
                VIDC_TRY(h_mx.get(ctx, 16));
                VIDC_HIP(hipMemsetAsync(s_mx.p, 0, 4, ctx->stream));
            }
            hipLaunchKernelGGL(k_ef_build_recs, dim3((uint32_t)std::min<uint64_t>((e->nbatches + 255) / 256, 4096)),
                               dim3(256), 0, ctx->stream, e->d_offsets.p, e->d_low_off.p, e->d_high_off.p, e->d_lbits.p,
                               e->d_batch_off.p, e->d_hrank.p, e->d_batches.p, e->nbatches, e->d_recs.p,
                               bounded ? (unsigned int *)nullptr : s_mx.as<unsigned int>());
            VIDC_HIP(hipGetLastError());
            VIDC_HIP(hipEventRecord(ctx->ev_chain[1], ctx->stream));
            if (bounded) {
                // (the records are published -- other contexts may use them -- when this call has synchronised; until then the
                // object's lock stays with this call)
                // (the bound is written to the object only when the records are published, so a call that fails
                // between here and its wait leaves the object as it found it; the lock is held for the ~20 us of a small object's
                // first decode -- only objects below EF_ENC_RECS_MAX batches come this way)
                pending_max_cnt = (uint32_t)std::min<uint64_t>(e->max_list, EF_BATCH_BITS);
                publish_after_sync = true;
            } else {
                VIDC_HIP(hipMemcpyAsync(h_mx.p, s_mx.p, 4, hipMemcpyDeviceToHost, ctx->stream));
                VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));  // another context may use the records next
                e->recs_max_cnt = *h_mx.as<unsigned int>();
                e->recs_ready = true;
            }
            recs_timed = true;
        }
        if (!publish_after_sync) g.unlock();
    }
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (e->nbatches) {
        const dim3 grid((uint32_t)std::min<uint64_t>(e->nbatches, (uint64_t)ctx->num_cu * 256));
        const uint32_t max_cnt = publish_after_sync ? pending_max_cnt : e->recs_max_cnt;
        const uint32_t lds = std::min<uint32_t>(EF_BATCH_BITS, (max_cnt + 63u) & ~63u) * 2;
        const bool small = max_cnt <= 256;
        if (e->narrow && small)
            hipLaunchKernelGGL((k_ef_decode_rec<uint32_t, 4>), grid, dim3(64), lds, ctx->stream, e->d_low.p, e->d_high.p,
                               e->d_recs.p, (uint32_t)e->nbatches, d_out);
        else if (e->narrow)
            hipLaunchKernelGGL((k_ef_decode_rec<uint32_t, 8>), grid, dim3(64), lds, ctx->stream, e->d_low.p, e->d_high.p,
                               e->d_recs.p, (uint32_t)e->nbatches, d_out);
        else if (small)
            hipLaunchKernelGGL((k_ef_decode_rec<uint64_t, 4>), grid, dim3(64), lds, ctx->stream, e->d_low.p, e->d_high.p,
                               e->d_recs.p, (uint32_t)e->nbatches, d_out);
        else
            hipLaunchKernelGGL((k_ef_decode_rec<uint64_t, 8>), grid, dim3(64), lds, ctx->stream, e->d_low.p, e->d_high.p,
                               e->d_recs.p, (uint32_t)e->nbatches, d_out);
    }
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    if (publish_after_sync) { e->recs_max_cnt = pending_max_cnt; e->recs_ready = true; g.unlock(); }
    if (recs_timed) {
        float rms = 0;
        if (hipEventElapsedTime(&rms, ctx->ev_chain[0], ctx->ev_chain[1]) == hipSuccess) recs_ms = rms;
    }
    ctx->last_kernel_ms = ms + recs_ms;
    return VIDC_OK;
}

static int ef_decode_some(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *list_nos,
                          const uint64_t *out_off_host, uint64_t *d_out, int32_t *d_rows, uint32_t K,
                          uint32_t *counts_host = nullptr) {
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_l, s_o;
    Pinned h_up;
    VIDC_TRY(h_up.get(ctx, m * 16));
    const uint64_t *d_l = nullptr;  // list_nos == NULL: lists 0..m-1, no index array
    if (list_nos) {
        VIDC_TRY(s_l.get(ctx, m * 8));
        std::memcpy(h_up.p, list_nos, m * 8);
        VIDC_HIP(hipMemcpyAsync(s_l.p, h_up.p, m * 8, hipMemcpyHostToDevice, ctx->stream));
        d_l = s_l.as<uint64_t>();
    }
    if (out_off_host) {
        VIDC_TRY(s_o.get(ctx, m * 8));
        std::memcpy(h_up.as<uint64_t>() + m, out_off_host, m * 8);
        VIDC_HIP(hipMemcpyAsync(s_o.p, h_up.as<uint64_t>() + m, m * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    {
        // few lists: several workgroups per list (by the longest requested list, ~2048 ids per workgroup; the host knows the sizes
        // whenever it knows the output offsets)
        uint32_t nsplit = 1;
        if (out_off_host && m < (uint64_t)ctx->num_cu * 8) {
            uint64_t longest = 0;
            for (uint64_t i = 0; i < m; i++) {
                const uint64_t l = list_nos ? list_nos[i] : i;
                longest = std::max(longest, e->offsets[l + 1] - e->offsets[l]);
            }
            nsplit = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, longest / 2048), std::max<uint64_t>(1, (uint64_t)ctx->num_cu * 16 / m));
        }
        hipLaunchKernelGGL(k_ef_decode, dim3((uint32_t)std::min<uint64_t>(m * nsplit, (uint64_t)ctx->num_cu * 64)), dim3(64), 0,
                           ctx->stream, e->d_low.p, e->d_high.p, e->d_offsets.p, e->d_low_off.p, e->d_high_off.p,
                           e->d_lbits.p, e->d_batch_off.p, e->d_hrank.p, (uint32_t)m, (const Chunk *)nullptr,
                           d_l, out_off_host ? s_o.as<uint64_t>() : nullptr, d_out, d_rows, K, nsplit);
    }
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    if (counts_host) VIDC_TRY(fetch_sizes<uint64_t>(ctx, e->d_offsets.p, d_l, m, counts_host));
    return VIDC_OK;
}

int vidc_ef_decode_lists(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *list_nos, uint64_t *d_out,
                         uint64_t *out_offsets) {
    if (!ctx || !e || (m && !list_nos) || !out_offsets) return VIDC_ERR_INVALID;
    VIDC_TRY(ef_ensure_csr(ctx, e));
    VIDC_TRY(ef_ensure_offsets(e));
    out_offsets[0] = 0;
    for (uint64_t i = 0; i < m; i++) {
        if (list_nos[i] >= e->nlist) { set_error("list number out of range"); return VIDC_ERR_INVALID; }
        out_offsets[i + 1] = out_offsets[i] + (e->offsets[list_nos[i] + 1] - e->offsets[list_nos[i]]);
    }
    if (!m) return VIDC_OK;
    return ef_decode_some(ctx, e, m, list_nos, out_offsets, d_out, nullptr, 0);
}

int vidc_ef_decode_gather(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                          const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out) {
    if (!ctx || !e) return VIDC_ERR_INVALID;
    VIDC_TRY(ef_ensure_offsets(e));
    return vidc_decode_gather_impl(ctx, e->nlist, m, list_nos, n_items, item_slot, item_off, ids_out,
                                   [&](uint64_t l) { return e->offsets[l + 1] - e->offsets[l]; },
                                   [&](uint64_t *d, uint64_t *lo) { return vidc_ef_decode_lists(ctx, e, m, list_nos, d, lo); });
}

// words per row of the arena of a graph object: see k_ef_rows_encode_tile
static void ef_rows_geometry(uint32_t K, uint32_t ubound, uint32_t *LW, uint32_t *HW) {
    uint32_t maxbits = 0;
    for (uint32_t n = 1; n <= K; n++) {
        const uint32_t q = ubound / n;
        maxbits = std::max(maxbits, n * (q ? (uint32_t)msb64(q) : 0u));
    }
    *LW = (maxbits + 63u) / 64u;
    *HW = (3u * K + 1u + 63u) / 64u;
}

// graph object, every requested row at most K wide: one row per lane from the arena
static int ef_decode_rows_arena(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *nodes, uint32_t K, int32_t *d_out,
                                uint32_t *counts) {
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_n, s_c;
    Pinned h_io;
    const uint64_t *d_nodes = nullptr;  // nodes == NULL: rows 0..m-1, no index array
    if (nodes || counts) VIDC_TRY(h_io.get(ctx, m * 8));
    if (nodes) {
        VIDC_TRY(s_n.get(ctx, m * 8));
        std::memcpy(h_io.p, nodes, m * 8);
        VIDC_HIP(hipMemcpyAsync(s_n.p, h_io.p, m * 8, hipMemcpyHostToDevice, ctx->stream));
        d_nodes = s_n.as<uint64_t>();
    }
    if (counts) VIDC_TRY(s_c.get(ctx, m * 4));
    const uint32_t S = e->a_lw + e->a_hw;
    const size_t dyn = ((size_t)64 * (S | 1u) + 1) * 8;
    const dim3 grid(tile_grid(ctx->num_cu, (m + 63) / 64, (uint32_t)std::min<size_t>(16, (150u << 10) / (dyn + 4608 + 256))));
    uint32_t *d_cnt = counts ? s_c.as<uint32_t>() : nullptr;
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    const bool quad = (K & 3u) == 0u && ((uintptr_t)d_out & 15u) == 0;
#define VIDC_EF_ROWS_DEC2(KP, HWC, Q)                                                                                                  \
    hipLaunchKernelGGL((k_ef_rows_decode_tile<KP, HWC, Q>), grid, dim3(64), dyn, ctx->stream, e->d_arena.p, e->d_rmeta.p, m, d_nodes, K, \
                       e->a_lw, tile_magic(S), d_out, d_cnt)
#define VIDC_EF_ROWS_DEC(KP, HWC)                        \
    do {                                                 \
        if (quad) VIDC_EF_ROWS_DEC2(KP, HWC, true);      \
        else VIDC_EF_ROWS_DEC2(KP, HWC, false);          \
    } while (0)
    if (e->K <= 32) {  // (3 K + 1 bits: K <= 21 / 32)
        if (e->a_hw == 1) VIDC_EF_ROWS_DEC(32, 1);
        else VIDC_EF_ROWS_DEC(32, 2);
    } else {           // (K <= 42 / 63 / 64)
        if (e->a_hw <= 2) VIDC_EF_ROWS_DEC(64, 2);
        else if (e->a_hw == 3) VIDC_EF_ROWS_DEC(64, 3);
        else VIDC_EF_ROWS_DEC(64, 4);
    }
#undef VIDC_EF_ROWS_DEC
#undef VIDC_EF_ROWS_DEC2
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

int vidc_ef_decode_rows(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *nodes, uint32_t K, int32_t *d_out,
                        uint32_t *counts) {
    if (!ctx || !e || (m && !d_out) || K == 0) return VIDC_ERR_INVALID;
    if (!nodes && m > e->nlist) { set_error("nodes == NULL selects nodes 0..m-1: m exceeds the node count"); return VIDC_ERR_INVALID; }
    // rows of a graph object never exceed its K
    const bool lean = e->arena && K >= e->K;
    if (!lean) {
        VIDC_TRY(ef_ensure_csr(ctx, e));
        VIDC_TRY(ef_ensure_offsets(e));
    }
    for (uint64_t i = 0; i < m && (nodes || !lean); i++) {
        const uint64_t node = nodes ? nodes[i] : i;
        if (node >= e->nlist) { set_error("node out of range"); return VIDC_ERR_INVALID; }
        if (lean) continue;
        uint64_t n = e->offsets[node + 1] - e->offsets[node];
        if (n > K) { set_error("node has more than K edges"); return VIDC_ERR_INVALID; }
        if (counts) counts[i] = (uint32_t)n;
    }
    if (!m) return VIDC_OK;
    if (lean) return ef_decode_rows_arena(ctx, e, m, nodes, K, d_out, counts);
    return ef_decode_some(ctx, e, m, nodes, nullptr, nullptr, d_out, K, nullptr);
}

int vidc_ef_encode_rows(vidc_ctx *ctx, uint64_t N, uint32_t K, const int32_t *d_rows, vidc_ef **out) {
    if (!ctx || !out || (N && !d_rows)) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (K == 0) { set_error("EF rows: K=0 unsupported"); return VIDC_ERR_UNSUPPORTED; }
    if (N >= 0xffffffffull) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(ctx->device));
    if (K > 64) {
        // wide rows (NSG128, NSG256, ...): the rows become CSR lists and take the per-list kernels (which sort each list
        // like altid_impl.cpp:76 and use its largest id as the universe, :75,77)
        std::vector<uint64_t> offsets;
        Scratch s_ids;
        VIDC_TRY(rows_to_csr(ctx, N, K, d_rows, offsets, s_ids));
        VIDC_TRY(vidc_ef_encode(ctx, N, offsets.data(), s_ids.as<uint64_t>(), 0, out));
        (*out)->K = K;
        return VIDC_OK;
    }
    std::unique_ptr<vidc_ef> e(new vidc_ef());
    e->device = ctx->device;
    e->nlist = N;
    e->rows = true;
    e->arena = true;
    e->csr_ready = false;
    e->K = K;
    e->narrow = true;
    e->max_list = K;
    Scratch s_tot;
    Pinned tail;
    const size_t tot_bytes = 64 * sizeof(EfRowsTot) + sizeof(EfRowsFlags);
    VIDC_TRY(tail.get(ctx, tot_bytes));
    VIDC_TRY(s_tot.get(ctx, tot_bytes));
    VIDC_TRY(e->d_rmeta.alloc(N ? N : 1, ctx->dpool));
    // neighbours of a graph are node numbers: ids < N bound the record size.  A row array that breaks this (a subgraph with global
    // ids) is seen by the kernel, which reports the largest id, and the call is repeated with records sized for it.
    uint32_t ubound = N ? (uint32_t)std::min<uint64_t>(N - 1, 0x7fffffffull) : 0u;
    const bool vec = ((uintptr_t)d_rows & 15u) == 0;
    for (int attempt = 0;; attempt++) {
        ef_rows_geometry(K, ubound, &e->a_lw, &e->a_hw);
        const uint32_t S = e->a_lw + e->a_hw;
        if (2u * S + 1u > (K <= 32 ? 33u : 65u)) { set_error("EF rows: record of %u words does not fit the tile", S); return VIDC_ERR_UNSUPPORTED; }
        VIDC_TRY(e->d_arena.alloc(N * S + 2, ctx->dpool));  // (+2: the decoder reads 16-byte pieces)
        VIDC_HIP(hipMemsetAsync(s_tot.p, 0, tot_bytes, ctx->stream));
        VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        if (N) {
            const dim3 grid((uint32_t)((N + 63) / 64));
            EfRowsTot *d_tot = s_tot.as<EfRowsTot>();
            EfRowsFlags *d_fl = (EfRowsFlags *)(d_tot + 64);
#define VIDC_EF_ROWS_ENC(KP, FULL)                                                                                                     \
    hipLaunchKernelGGL((k_ef_rows_encode_tile<KP, FULL>), grid, dim3(64),                                                                   \
                       4u * std::max<size_t>((FULL && vec) ? RowsTile<KP>::DWORDS_PF : RowsTile<KP>::DWORDS, 64u * (2u * S + 1u) + 64u), ctx->stream, \
                       d_rows, N, K, tile_magic(K), vec ? 1u : 0u, e->a_lw, \
                       e->a_hw, tile_magic(S), ubound, e->d_arena.p, e->d_rmeta.p, d_tot, d_fl)
            if (K == 32) VIDC_EF_ROWS_ENC(32, true);
            else if (K < 32) VIDC_EF_ROWS_ENC(32, false);
            else if (K == 64) VIDC_EF_ROWS_ENC(64, true);
            else VIDC_EF_ROWS_ENC(64, false);
#undef VIDC_EF_ROWS_ENC
            VIDC_HIP(hipGetLastError());
        }
        VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        VIDC_HIP(hipMemcpyAsync(tail.p, s_tot.p, tot_bytes, hipMemcpyDeviceToHost, ctx->stream));
        VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
        const EfRowsTot *t = tail.as<EfRowsTot>();
        const EfRowsFlags *fl = (const EfRowsFlags *)(t + 64);
        if (fl->bad) { set_error("EF rows: negative neighbour id before the -1 terminator"); return VIDC_ERR_DOMAIN; }
        if (fl->over && attempt == 0) { ubound = fl->maxid; continue; }
        if (fl->over) { set_error("EF rows: internal: record bound exceeded twice"); return VIDC_ERR_OVERFLOW; }
        e->ntotal = 0; e->total_bits = 0;
        for (int i = 0; i < 64; i++) { e->ntotal += t[i].edges; e->total_bits += t[i].bits; }
        break;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    *out = e.release();
    return VIDC_OK;
}

int vidc_ef_get(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *list_nos, const uint64_t *offs,
                int64_t *ids_out) {
    if (!ctx || !e || (m && (!list_nos || !offs || !ids_out))) return VIDC_ERR_INVALID;
    if (!m) return VIDC_OK;
    VIDC_TRY(ef_ensure_csr(ctx, e));
    VIDC_TRY(ef_ensure_offsets(e));
    for (uint64_t i = 0; i < m; i++) {
        if (list_nos[i] >= e->nlist || offs[i] >= e->offsets[list_nos[i] + 1] - e->offsets[list_nos[i]]) {
            set_error("ef get: (list %llu, offset %llu) out of range", (unsigned long long)list_nos[i],
                      (unsigned long long)offs[i]);
            return VIDC_ERR_INVALID;
        }
    }
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_l, s_o, s_r;
    VIDC_TRY(s_l.get(ctx, m * 8)); VIDC_TRY(s_o.get(ctx, m * 8)); VIDC_TRY(s_r.get(ctx, m * 8));
    VIDC_HIP(hipMemcpyAsync(s_l.p, list_nos, m * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(s_o.p, offs, m * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_ef_get, dim3((uint32_t)std::min<uint64_t>(m, 1u << 20)), dim3(64), 0, ctx->stream, e->d_low.p,
                       e->d_high.p, e->d_low_off.p, e->d_high_off.p, e->d_lbits.p, e->d_batch_off.p, e->d_hrank.p, m,
                       s_l.as<uint64_t>(), s_o.as<uint64_t>(), s_r.as<int64_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipMemcpyAsync(ids_out, s_r.p, m * 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    ctx->d2h_bytes += m * 8;
    return VIDC_OK;
}

int vidc_ef_perm(vidc_ctx *ctx, const vidc_ef *e, uint32_t *perm_host) {
    if (!ctx || !e || !perm_host) return VIDC_ERR_INVALID;
    if (!e->has_perm) { set_error("encode was called without VIDC_EF_WANT_PERM"); return VIDC_ERR_INVALID; }
    return vidc_copy_d2h(ctx, perm_host, e->d_perm.p, e->ntotal * 4);
}

int vidc_ef_export(vidc_ctx *ctx, const vidc_ef *e, uint64_t list_no, uint64_t *low, size_t low_cap, uint64_t *high,
                   size_t high_cap, uint64_t *low_nbits, uint64_t *high_nbits) {
    if (!ctx || !e || list_no >= e->nlist) return VIDC_ERR_INVALID;
    VIDC_TRY(ef_ensure_csr(ctx, e));
    VIDC_TRY(ef_ensure_meta(e));
    uint64_t m = e->offsets[list_no + 1] - e->offsets[list_no];
    uint64_t lb = m * e->lbits[list_no], hb = e->high_nbits[list_no];
    if (low_nbits) *low_nbits = lb;
    if (high_nbits) *high_nbits = hb;
    uint64_t lw = (lb + 63) / 64, hw = (hb + 63) / 64;
    if (low) {
        if (lw > low_cap) { set_error("low buffer too small"); return VIDC_ERR_INVALID; }
        VIDC_TRY(vidc_copy_d2h(ctx, low, e->d_low.p + e->low_off[list_no], lw * 8));
    }
    if (high) {
        if (hw > high_cap) { set_error("high buffer too small"); return VIDC_ERR_INVALID; }
        VIDC_TRY(vidc_copy_d2h(ctx, high, e->d_high.p + e->high_off[list_no], hw * 8));
    }
    return VIDC_OK;
}

// ---- flat image of the object (the reference keeps compressed lists in memory only):
// {offsets, l[], universe[], low[], high[]}; stream offsets follow from (count, l, universe) per list
int vidc_ef_stream_words(const vidc_ef *e, uint64_t *low_words, uint64_t *high_words) {
    if (!e) return VIDC_ERR_INVALID;
    if (e->arena) {  // (the CSR streams may not exist yet: their sizes follow from the per-row geometry)
        VIDC_TRY(ef_ensure_meta(e));
        if (low_words) *low_words = e->low_off[e->nlist] ? e->low_off[e->nlist] : 1;
        if (high_words) *high_words = e->high_off[e->nlist] ? e->high_off[e->nlist] : 1;
        return VIDC_OK;
    }
    if (low_words) *low_words = ef_low_count(e);
    if (high_words) *high_words = ef_high_count(e);
    return VIDC_OK;
}
int vidc_ef_export_all(vidc_ctx *ctx, const vidc_ef *e, uint64_t *low, size_t low_cap, uint64_t *high, size_t high_cap) {
    if (!ctx || !e || !low || !high) return VIDC_ERR_INVALID;
    VIDC_TRY(ef_ensure_csr(ctx, e));
    const uint64_t nl = ef_low_count(e), nh = ef_high_count(e);
    if (nl > low_cap || nh > high_cap) { set_error("export buffers too small"); return VIDC_ERR_INVALID; }
    VIDC_TRY(vidc_copy_d2h(ctx, low, e->d_low.p, nl * 8));
    return vidc_copy_d2h(ctx, high, e->d_high.p, nh * 8);
}
int vidc_ef_import(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint32_t *lbits,
                   const uint64_t *universe, const uint64_t *low, uint64_t n_low, const uint64_t *high, uint64_t n_high,
                   vidc_ef **out) {
    if (!ctx || !out || (nlist && (!offsets || !lbits || !universe))) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (nlist >= 0xffffffffull) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_ef> e(new vidc_ef());
    e->device = ctx->device;
    e->nlist = nlist;
    e->offsets.assign(nlist + 1, 0);
    if (nlist) e->offsets.assign(offsets, offsets + nlist + 1);
    e->ntotal = e->offsets[nlist];
    e->lbits.assign(lbits, lbits + nlist);
    e->universe.assign(universe, universe + nlist);
    e->high_nbits.assign(nlist, 0);
    e->low_off.assign(nlist + 1, 0);
    e->high_off.assign(nlist + 1, 0);
    std::vector<uint64_t> batch_off(nlist + 1, 0);
    for (uint64_t l = 0; l < nlist; l++) {
        if (e->offsets[l + 1] < e->offsets[l] || lbits[l] > 63) { set_error("ef import: bad geometry at list %llu", (unsigned long long)l); return VIDC_ERR_INVALID; }
        const uint64_t m = e->offsets[l + 1] - e->offsets[l];
        uint64_t lw = 0, hw = 0;
        if (m) {
            const uint64_t hb = (m + 1) + (universe[l] >> lbits[l]) + 1;
            e->high_nbits[l] = hb;
            e->total_bits += m * lbits[l] + hb;
            lw = (m * lbits[l] + 63) / 64 + 1;
            hw = (hb + 63) / 64;
        }
        e->low_off[l + 1] = e->low_off[l] + lw;
        e->high_off[l + 1] = e->high_off[l] + hw;
        batch_off[l + 1] = batch_off[l] + (hw + 63) / 64;
    }
    e->offsets_host = e->meta_host = true;
    const uint64_t need_low = e->low_off[nlist] ? e->low_off[nlist] : 1, need_high = e->high_off[nlist] ? e->high_off[nlist] : 1;
    if (n_low != need_low || n_high != need_high || !low || !high) {
        set_error("ef import: streams of %llu / %llu words given, the geometry needs %llu / %llu", (unsigned long long)n_low,
                  (unsigned long long)n_high, (unsigned long long)need_low, (unsigned long long)need_high);
        return VIDC_ERR_INVALID;
    }
    e->nbatches = batch_off[nlist];
    VIDC_TRY(e->d_offsets.alloc(nlist + 1, ctx->dpool)); VIDC_TRY(e->d_low_off.alloc(nlist + 1, ctx->dpool));
    VIDC_TRY(e->d_high_off.alloc(nlist + 1, ctx->dpool)); VIDC_TRY(e->d_batch_off.alloc(nlist + 1, ctx->dpool));
    VIDC_TRY(e->d_lbits.alloc(nlist ? nlist : 1, ctx->dpool)); VIDC_TRY(e->d_universe.alloc(nlist ? nlist : 1, ctx->dpool));
    VIDC_TRY(e->d_low.alloc(need_low, ctx->dpool)); VIDC_TRY(e->d_high.alloc(need_high, ctx->dpool));
    VIDC_TRY(e->d_chunks.alloc(1, ctx->dpool));  // encoder-only table
    VIDC_TRY(e->d_batches.alloc(e->nbatches ? e->nbatches : 1, ctx->dpool));
    VIDC_TRY(e->d_hrank.alloc(e->nbatches ? e->nbatches : 1, ctx->dpool));
    VIDC_TRY(vidc_copy_h2d(ctx, e->d_offsets.p, e->offsets.data(), (nlist + 1) * 8));
    VIDC_TRY(vidc_copy_h2d(ctx, e->d_low_off.p, e->low_off.data(), (nlist + 1) * 8));
    VIDC_TRY(vidc_copy_h2d(ctx, e->d_high_off.p, e->high_off.data(), (nlist + 1) * 8));
    VIDC_TRY(vidc_copy_h2d(ctx, e->d_batch_off.p, batch_off.data(), (nlist + 1) * 8));
    if (nlist) {
        VIDC_TRY(vidc_copy_h2d(ctx, e->d_lbits.p, e->lbits.data(), nlist * 4));
        VIDC_TRY(vidc_copy_h2d(ctx, e->d_universe.p, e->universe.data(), nlist * 8));
    }
    VIDC_TRY(vidc_copy_h2d(ctx, e->d_low.p, low, n_low * 8));
    VIDC_TRY(vidc_copy_h2d(ctx, e->d_high.p, high, n_high * 8));
    if (e->nbatches) {
        const uint32_t wgrid = (uint32_t)std::min<uint64_t>(nlist, (uint64_t)ctx->num_cu * 64);
        launch_fill_items(ctx->stream, e->d_batch_off.p, (uint32_t)nlist, 1u, e->d_batches.p, e->nbatches, (uint32_t)ctx->num_cu);
        hipLaunchKernelGGL(k_ef_hrank_from_high, dim3(wgrid), dim3(64), 0, ctx->stream, e->d_high.p, e->d_high_off.p,
                           e->d_batch_off.p, (uint32_t)nlist, e->d_hrank.p);
        VIDC_HIP(hipGetLastError());
    }
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    *out = e.release();
    return VIDC_OK;
}

}  // extern "C"
