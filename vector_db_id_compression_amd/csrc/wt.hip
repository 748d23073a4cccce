// wt.hip -- wavelet tree over the sequence list_nos[id] (CompressedIDInvertedListsWaveletTree,
// custom_invlists_impl.cpp:346-397): id = select(offset + 1, list_no)  (:377-379).
//
// The reference delegates to sdsl::wt_int (not in the reference tree, "parity unpinned" for its size and
// internal layout); what is reproduced here is the data structure itself and the select semantics:
//   L = bit_width(nlist - 1) levels; level l holds one bit per id (bit L-1-l of the symbol) in the order
//   obtained by stable-sorting the ids by the top l bits of their symbol (pointerless / levelwise layout);
//   every level has a rank directory (ones before each 512-bit block).
// select walks the levels bottom-up with one rank and one select per level.  Everything is built and
// queried on the GPU: thread-per-word bit gathering, a workgroup scan for the rank directory, and a
// rank-driven stable partition to derive the next level's order.
#include <algorithm>
#include <cmath>
#include <memory>

#include "bits.h"
#include "common.h"

using namespace vidc;
using namespace vidc::dev;

struct vidc_wt {
    int device = 0;
    uint64_t ntotal = 0, nlist = 0;
    uint32_t L = 0;
    int wt_type = 0;
    uint64_t words_per_level = 0, blocks_per_level = 0;
    uint64_t size_bytes = 0;
    std::vector<uint64_t> offsets;      // host copy of the symbol start positions C[s]
    DevBuf<uint64_t> d_bits;            // L * words_per_level
    DevBuf<uint32_t> d_rank;            // L * (blocks_per_level + 1): ones before each 512-bit block
    DevBuf<uint64_t> d_C;               // nlist + 1
};

namespace {

constexpr uint32_t BLK_WORDS = 8;  // 512-bit rank blocks

// list_nos[id] = list number; validates the reference's asserts (ids ascending inside a list, < ntotal)
__global__ void k_wt_scatter_syms(const uint64_t *ids, const uint64_t *offsets, uint32_t nlist, uint64_t ntotal,
                                  uint32_t *syms, uint32_t *err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ntotal; g += stride) {
        const uint32_t l = find_list(offsets, nlist, g);
        const uint64_t id = ids[g];
        bool bad = id >= ntotal;
        if (g > offsets[l]) bad |= ids[g - 1] >= id;  // assert(ids_data[i] > prev_id), :359
        if (bad) { atomicOr(err, 1u); continue; }
        if (atomicExch(&syms[id], l) != 0xffffffffu) atomicOr(err, 2u);  // id present twice
    }
}

// one thread per 64-bit word of the level bitvector
__global__ void k_wt_bits(const uint32_t *syms_in_order, uint64_t ntotal, uint32_t shift, uint64_t *bits,
                          uint64_t nwords) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
        uint64_t out = 0;
        const uint64_t base = w * 64;
        for (uint32_t b = 0; b < 64 && base + b < ntotal; b++)
            out |= (uint64_t)((syms_in_order[base + b] >> shift) & 1u) << b;
        bits[w] = out;
    }
}

// rank directory: rank[j] = ones in blocks [0, j); single workgroup, chunked scan
__global__ void __launch_bounds__(1024) k_wt_rankdir(const uint64_t *bits, uint64_t nwords, uint64_t nblocks,
                                                     uint32_t *rank) {
    __shared__ uint32_t sh[1024];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint64_t b0 = 0; b0 < nblocks; b0 += 1024) {
        const uint64_t b = b0 + threadIdx.x;
        uint32_t c = 0;
        if (b < nblocks)
            for (uint32_t j = 0; j < BLK_WORDS; j++) {
                uint64_t w = b * BLK_WORDS + j;
                if (w < nwords) c += (uint32_t)__builtin_popcountll(bits[w]);
            }
        sh[threadIdx.x] = c;
        __syncthreads();
        for (uint32_t o = 1; o < 1024; o <<= 1) {
            uint32_t v = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += v;
            __syncthreads();
        }
        if (b < nblocks) rank[b] = carry + sh[threadIdx.x] - c;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) rank[nblocks] = carry;
}

__device__ __forceinline__ uint64_t rank1(const uint64_t *bits, const uint32_t *rank, uint64_t i) {
    const uint64_t blk = i / (64 * BLK_WORDS);
    uint64_t r = rank[blk];
    const uint64_t w_end = i >> 6;
    for (uint64_t w = blk * BLK_WORDS; w < w_end; w++) r += (uint64_t)__builtin_popcountll(bits[w]);
    const uint32_t rem = (uint32_t)(i & 63);
    if (rem) r += (uint64_t)__builtin_popcountll(bits[w_end] & ((1ull << rem) - 1ull));
    return r;
}

// position of the (j+1)-th bit equal to `one` (j 0-based); nblocks = number of 512-bit blocks
__device__ __forceinline__ uint64_t select_bit(const uint64_t *bits, const uint32_t *rank, uint64_t nblocks,
                                               uint64_t nwords, uint64_t j, bool one) {
    // largest block b with count_before(b) <= j
    uint64_t lo = 0, hi = nblocks;
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        const uint64_t before = one ? rank[mid] : mid * 64 * BLK_WORDS - rank[mid];
        if (before <= j) lo = mid; else hi = mid;
    }
    uint64_t seen = one ? rank[lo] : lo * 64 * BLK_WORDS - rank[lo];
    for (uint64_t w = lo * BLK_WORDS; w < nwords; w++) {
        uint64_t word = one ? bits[w] : ~bits[w];
        const uint32_t c = (uint32_t)__builtin_popcountll(word);
        if (seen + c > j) {
            for (uint64_t k = j - seen; k; k--) word &= word - 1;
            return w * 64 + (uint64_t)__builtin_ctzll(word);
        }
        seen += c;
    }
    return ~0ull;
}

// stable partition of every node of the level by its bit -> order of the next level
__global__ void k_wt_partition(const uint32_t *syms_in, uint32_t *syms_out, const uint64_t *bits, const uint32_t *rank,
                               const uint64_t *C, uint64_t ntotal, uint32_t nlist, uint32_t L, uint32_t level) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t sh = L - level;  // symbols of one node share their top `level` bits
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntotal; i += stride) {
        const uint32_t s = syms_in[i];
        const uint64_t p = sh >= 32 ? 0 : (uint64_t)(s >> sh);
        uint64_t s_lo = sh >= 32 ? 0 : (p << sh), s_hi = sh >= 32 ? nlist : ((p + 1) << sh);
        if (s_hi > nlist) s_hi = nlist;
        const uint64_t ns = C[s_lo], ne = C[s_hi];
        const uint64_t r_ns = rank1(bits, rank, ns), r_i = rank1(bits, rank, i), r_ne = rank1(bits, rank, ne);
        const bool bit = (bits[i >> 6] >> (i & 63)) & 1ull;
        const uint64_t zeros_in_node = (ne - ns) - (r_ne - r_ns);
        const uint64_t dst = bit ? ns + zeros_in_node + (r_i - r_ns) : ns + ((i - ns) - (r_i - r_ns));
        syms_out[dst] = s;
    }
}

// one thread per query: id of the (k+1)-th element of list c
__global__ void k_wt_select(const uint64_t *bits, const uint32_t *rank, const uint64_t *C, uint64_t words_per_level,
                            uint64_t blocks_per_level, uint32_t nlist, uint32_t L, uint64_t m,
                            const uint64_t *list_nos, const uint64_t *offs, int64_t *out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += stride) {
        const uint32_t c = (uint32_t)list_nos[q];
        uint64_t pos = offs[q];
        for (int level = (int)L - 1; level >= 0; level--) {
            const uint32_t sh = L - (uint32_t)level;
            const uint64_t p = sh >= 32 ? 0 : (uint64_t)(c >> sh);
            const uint64_t s_lo = sh >= 32 ? 0 : (p << sh);
            const uint64_t ns = C[s_lo];
            const bool bit = (c >> (L - 1u - (uint32_t)level)) & 1u;
            const uint64_t *b = bits + (uint64_t)level * words_per_level;
            const uint32_t *r = rank + (uint64_t)level * (blocks_per_level + 1);
            const uint64_t r_ns = rank1(b, r, ns);
            const uint64_t before = bit ? r_ns : ns - r_ns;
            pos = select_bit(b, r, blocks_per_level, words_per_level, before + pos, bit) - ns;
        }
        out[q] = (int64_t)pos;
    }
}

// get_ids of the requested lists (custom_invlists_impl.cpp:381-392 loops get_single_id): one thread per output slot,
// its request item found in the m + 1 output offsets
__global__ void k_wt_decode_lists(const uint64_t *bits, const uint32_t *rank, const uint64_t *C, uint64_t words_per_level,
                                  uint64_t blocks_per_level, uint32_t L, uint64_t m, const uint64_t *list_nos,
                                  const uint64_t *out_off, uint64_t total, uint64_t *out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += stride) {
        uint64_t lo = 0, hi = m;  // largest i with out_off[i] <= g
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (out_off[mid] <= g) lo = mid; else hi = mid;
        }
        const uint32_t c = (uint32_t)list_nos[lo];
        uint64_t pos = g - out_off[lo];
        for (int level = (int)L - 1; level >= 0; level--) {
            const uint32_t sh = L - (uint32_t)level;
            const uint64_t p = sh >= 32 ? 0 : (uint64_t)(c >> sh);
            const uint64_t ns = C[sh >= 32 ? 0 : (p << sh)];
            const bool bit = (c >> (L - 1u - (uint32_t)level)) & 1u;
            const uint64_t *b = bits + (uint64_t)level * words_per_level;
            const uint32_t *r = rank + (uint64_t)level * (blocks_per_level + 1);
            const uint64_t r_ns = rank1(b, r, ns);
            pos = select_bit(b, r, blocks_per_level, words_per_level, (bit ? r_ns : ns - r_ns) + pos, bit) - ns;
        }
        out[g] = pos;
    }
}

// get_ids for every list (custom_invlists_impl.cpp:381-392 loops get_single_id)
__global__ void k_wt_decode_all(const uint64_t *bits, const uint32_t *rank, const uint64_t *C, uint64_t words_per_level,
                                uint64_t blocks_per_level, uint32_t nlist, uint32_t L, uint64_t ntotal,
                                uint64_t *out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ntotal; g += stride) {
        const uint32_t c = find_list(C, nlist, g);
        uint64_t pos = g - C[c];
        for (int level = (int)L - 1; level >= 0; level--) {
            const uint32_t sh = L - (uint32_t)level;
            const uint64_t p = sh >= 32 ? 0 : (uint64_t)(c >> sh);
            const uint64_t ns = C[sh >= 32 ? 0 : (p << sh)];
            const bool bit = (c >> (L - 1u - (uint32_t)level)) & 1u;
            const uint64_t *b = bits + (uint64_t)level * words_per_level;
            const uint32_t *r = rank + (uint64_t)level * (blocks_per_level + 1);
            const uint64_t r_ns = rank1(b, r, ns);
            pos = select_bit(b, r, blocks_per_level, words_per_level, (bit ? r_ns : ns - r_ns) + pos, bit) - ns;
        }
        out[g] = pos;
    }
}

// get_ids for every list as L stable partitions of the id sequence (the build's own passes, run on ids instead of
// symbols): at level l the element at position i of node p goes to the zeros / ones part of its node, one rank per
// element instead of the select (binary search over the rank directory + word scan) the per-id walk needs per level.
// in == nullptr: level 0 (position i holds id i, prefix 0).  LAST: the final order is the answer (u64 ids).
struct WtItem {
    uint32_t id, pref;
};
template <bool LAST>
__global__ void k_wt_decode_level(const WtItem *in, WtItem *out, uint64_t *out_ids, const uint64_t *bits,
                                  const uint32_t *rank, const uint64_t *C, uint64_t ntotal, uint32_t nlist, uint32_t L,
                                  uint32_t level) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t sh = L - level;  // symbols of one node share their top `level` bits
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntotal; i += stride) {
        const WtItem it = in ? in[i] : WtItem{(uint32_t)i, 0u};
        const uint64_t p = it.pref;
        uint64_t s_lo = sh >= 32 ? 0 : (p << sh), s_hi = sh >= 32 ? nlist : ((p + 1) << sh);
        if (s_lo > nlist) s_lo = nlist;
        if (s_hi > nlist) s_hi = nlist;
        const uint64_t ns = C[s_lo], ne = C[s_hi];
        const uint64_t r_ns = rank1(bits, rank, ns), r_i = rank1(bits, rank, i), r_ne = rank1(bits, rank, ne);
        const bool bit = (bits[i >> 6] >> (i & 63)) & 1ull;
        const uint64_t zeros_in_node = (ne - ns) - (r_ne - r_ns);
        const uint64_t dst = bit ? ns + zeros_in_node + (r_i - r_ns) : ns + ((i - ns) - (r_i - r_ns));
        if (LAST) out_ids[dst] = it.id;
        else out[dst] = WtItem{it.id, (uint32_t)((p << 1) | (bit ? 1u : 0u))};
    }
}

// size of an RRR-63 coded bitvector (sdsl::rrr_vector<63>): 6-bit class + ceil(log2 C(63, class)) offset bits
// per 63-bit block, plus a 64-bit pointer and rank sample every 32 blocks (the layout sdsl documents)
__global__ void k_wt_rrr_bits(const uint64_t *bits, uint64_t nbits, const uint8_t *offbits, unsigned long long *total) {
    const uint64_t nblk = (nbits + 62) / 63;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        const uint64_t pos = b * 63;
        uint64_t v = bits[pos >> 6] >> (pos & 63);
        if ((pos & 63) + 63 > 64) v |= bits[(pos >> 6) + 1] << (64 - (pos & 63));
        v &= (1ull << 63) - 1ull;
        acc += 6u + offbits[__builtin_popcountll(v)];
    }
    atomicAdd(total, acc);
}

}  // namespace

extern "C" {

int vidc_wt_build(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids, int wt_type,
                  vidc_wt **out) {
    if (!ctx || !out || (nlist && !offsets) || (wt_type != 0 && wt_type != 1)) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (nlist == 0 || nlist >= 0xffffffffull) { set_error("wavelet tree needs 1 <= nlist < 2^32"); return VIDC_ERR_INVALID; }
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_wt> w(new vidc_wt());
    w->device = ctx->device;
    w->nlist = nlist;
    w->wt_type = wt_type;
    w->offsets.assign(offsets, offsets + nlist + 1);
    w->ntotal = offsets[nlist];
    if (w->ntotal >= 0xffffffffull) { set_error("wavelet tree: ntotal must be < 2^32"); return VIDC_ERR_UNSUPPORTED; }
    uint32_t L = 1;
    while ((1ull << L) < nlist) L++;
    w->L = L;
    const uint64_t nt = w->ntotal;
    w->words_per_level = (nt + 63) / 64 + 1;
    w->blocks_per_level = (w->words_per_level + BLK_WORDS - 1) / BLK_WORDS;
    VIDC_TRY(w->d_C.alloc(nlist + 1));
    VIDC_HIP(hipMemcpyAsync(w->d_C.p, w->offsets.data(), (nlist + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_TRY(w->d_bits.alloc(L * w->words_per_level));
    VIDC_TRY(w->d_rank.alloc(L * (w->blocks_per_level + 1)));
    VIDC_HIP(hipMemsetAsync(w->d_bits.p, 0, L * w->words_per_level * 8, ctx->stream));
    Scratch s_a, s_b, s_err, s_tot, s_tab;
    VIDC_TRY(s_a.get(ctx, (nt ? nt : 1) * 4));
    VIDC_TRY(s_b.get(ctx, (nt ? nt : 1) * 4));
    VIDC_TRY(s_err.get(ctx, 4));
    VIDC_HIP(hipMemsetAsync(s_err.p, 0, 4, ctx->stream));
    VIDC_HIP(hipMemsetAsync(s_a.p, 0xff, (nt ? nt : 1) * 4, ctx->stream));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((nt + 255) / 256 + 1, (uint64_t)ctx->num_cu * 32);
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (nt) {
        hipLaunchKernelGGL(k_wt_scatter_syms, dim3(grid), dim3(256), 0, ctx->stream, d_ids, w->d_C.p, (uint32_t)nlist, nt,
                           s_a.as<uint32_t>(), s_err.as<uint32_t>());
        VIDC_HIP(hipGetLastError());
    }
    uint32_t err = 0;
    VIDC_HIP(hipMemcpyAsync(&err, s_err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    if (err) {
        set_error("wavelet tree: ids must be a permutation of 0..ntotal-1, ascending inside every list "
                  "(asserts at custom_invlists_impl.cpp:359-360)");
        return VIDC_ERR_DOMAIN;
    }
    uint32_t *cur = s_a.as<uint32_t>(), *nxt = s_b.as<uint32_t>();
    for (uint32_t level = 0; level < L && nt; level++) {
        uint64_t *bits = w->d_bits.p + (uint64_t)level * w->words_per_level;
        uint32_t *rank = w->d_rank.p + (uint64_t)level * (w->blocks_per_level + 1);
        hipLaunchKernelGGL(k_wt_bits, dim3(grid), dim3(256), 0, ctx->stream, cur, nt, L - 1 - level, bits,
                           (nt + 63) / 64);
        hipLaunchKernelGGL(k_wt_rankdir, dim3(1), dim3(1024), 0, ctx->stream, bits, w->words_per_level,
                           w->blocks_per_level, rank);
        if (level + 1 < L) {
            hipLaunchKernelGGL(k_wt_partition, dim3(grid), dim3(256), 0, ctx->stream, cur, nxt, bits, rank, w->d_C.p, nt,
                               (uint32_t)nlist, L, level);
            std::swap(cur, nxt);
        }
        VIDC_HIP(hipGetLastError());
    }
    // size accounting: plain = bits + rank directory; rrr = class/offset model of rrr_vector<63>
    uint64_t plain = (uint64_t)L * (((nt + 63) / 64) * 8 + (w->blocks_per_level + 1) * 4) + (nlist + 1) * 8;
    w->size_bytes = plain;
    if (wt_type == 1 && nt) {
        uint8_t tab[64];
        for (int c = 0; c <= 63; c++) {  // ceil(log2(binom(63, c)))
            long double lg = 0;
            for (int i = 1; i <= c; i++) lg += log2l((long double)(63 - c + i)) - log2l((long double)i);
            tab[c] = (uint8_t)ceill(lg - 1e-12L);
        }
        VIDC_TRY(s_tab.get(ctx, 64));
        VIDC_TRY(s_tot.get(ctx, 8));
        VIDC_HIP(hipMemcpyAsync(s_tab.p, tab, 64, hipMemcpyHostToDevice, ctx->stream));
        VIDC_HIP(hipMemsetAsync(s_tot.p, 0, 8, ctx->stream));
        for (uint32_t level = 0; level < L; level++)
            hipLaunchKernelGGL(k_wt_rrr_bits, dim3(grid), dim3(256), 0, ctx->stream,
                               w->d_bits.p + (uint64_t)level * w->words_per_level, nt, s_tab.as<uint8_t>(),
                               s_tot.as<unsigned long long>());
        unsigned long long tot = 0;
        VIDC_HIP(hipMemcpyAsync(&tot, s_tot.p, 8, hipMemcpyDeviceToHost, ctx->stream));
        VIDC_HIP(hipStreamSynchronize(ctx->stream));
        const uint64_t nblk = (nt + 62) / 63;
        w->size_bytes = (tot + 7) / 8 + (uint64_t)L * ((nblk + 31) / 32) * 16 + (nlist + 1) * 8;
    }
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    *out = w.release();
    return VIDC_OK;
}

void vidc_wt_destroy(vidc_wt *w) { delete w; }
uint64_t vidc_wt_size_in_bytes(const vidc_wt *w) { return w ? w->size_bytes : 0; }
uint32_t vidc_wt_levels(const vidc_wt *w) { return w ? w->L : 0; }

int vidc_wt_select(vidc_ctx *ctx, const vidc_wt *w, uint64_t m, const uint64_t *list_nos, const uint64_t *offs,
                   int64_t *ids_out) {
    if (!ctx || !w || (m && (!list_nos || !offs || !ids_out))) return VIDC_ERR_INVALID;
    if (!m) return VIDC_OK;
    for (uint64_t i = 0; i < m; i++)
        if (list_nos[i] >= w->nlist || offs[i] >= w->offsets[list_nos[i] + 1] - w->offsets[list_nos[i]]) {
            set_error("wt select: (list %llu, offset %llu) out of range", (unsigned long long)list_nos[i],
                      (unsigned long long)offs[i]);
            return VIDC_ERR_INVALID;
        }
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_l, s_o, s_r;
    VIDC_TRY(s_l.get(ctx, m * 8)); VIDC_TRY(s_o.get(ctx, m * 8)); VIDC_TRY(s_r.get(ctx, m * 8));
    VIDC_HIP(hipMemcpyAsync(s_l.p, list_nos, m * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(s_o.p, offs, m * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_wt_select, dim3((uint32_t)std::min<uint64_t>((m + 127) / 128, 1u << 16)), dim3(128), 0,
                       ctx->stream, w->d_bits.p, w->d_rank.p, w->d_C.p, w->words_per_level, w->blocks_per_level,
                       (uint32_t)w->nlist, w->L, m, s_l.as<uint64_t>(), s_o.as<uint64_t>(), s_r.as<int64_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipMemcpyAsync(ids_out, s_r.p, m * 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    return VIDC_OK;
}

int vidc_wt_decode_lists(vidc_ctx *ctx, const vidc_wt *w, uint64_t m, const uint64_t *list_nos, uint64_t *d_out,
                         uint64_t *out_offsets) {
    if (!ctx || !w || (m && !list_nos) || !out_offsets) return VIDC_ERR_INVALID;
    out_offsets[0] = 0;
    for (uint64_t i = 0; i < m; i++) {
        if (list_nos[i] >= w->nlist) { set_error("list number out of range"); return VIDC_ERR_INVALID; }
        out_offsets[i + 1] = out_offsets[i] + (w->offsets[list_nos[i] + 1] - w->offsets[list_nos[i]]);
    }
    const uint64_t total = out_offsets[m];
    ctx->last_kernel_ms = 0;
    if (!total) return VIDC_OK;
    if (!d_out) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_l, s_o;
    VIDC_TRY(s_l.get(ctx, m * 8)); VIDC_TRY(s_o.get(ctx, (m + 1) * 8));
    VIDC_HIP(hipMemcpyAsync(s_l.p, list_nos, m * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(s_o.p, out_offsets, (m + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_wt_decode_lists, dim3((uint32_t)std::min<uint64_t>((total + 127) / 128, 1u << 16)), dim3(128), 0,
                       ctx->stream, w->d_bits.p, w->d_rank.p, w->d_C.p, w->words_per_level, w->blocks_per_level, w->L, m,
                       s_l.as<uint64_t>(), s_o.as<uint64_t>(), total, d_out);
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    return VIDC_OK;
}

int vidc_wt_decode_all(vidc_ctx *ctx, const vidc_wt *w, uint64_t *d_out) {
    if (!ctx || !w || (w->ntotal && !d_out)) return VIDC_ERR_INVALID;
    if (!w->ntotal) return VIDC_OK;
    VIDC_HIP(hipSetDevice(ctx->device));
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((w->ntotal + 255) / 256, (uint64_t)ctx->num_cu * 32);
    if (w->L == 0) {  // one list: ids 0..ntotal-1 in order (the per-id kernel handles it)
        hipLaunchKernelGGL(k_wt_decode_all, dim3(grid), dim3(128), 0, ctx->stream, w->d_bits.p, w->d_rank.p, w->d_C.p,
                           w->words_per_level, w->blocks_per_level, (uint32_t)w->nlist, w->L, w->ntotal, d_out);
    } else {
        Scratch s_a, s_b;  // ping-pong (id, symbol prefix) arrays
        if (w->L > 1) {
            VIDC_TRY(s_a.get(ctx, w->ntotal * sizeof(WtItem)));
            if (w->L > 2) VIDC_TRY(s_b.get(ctx, w->ntotal * sizeof(WtItem)));
        }
        const WtItem *in = nullptr;
        for (uint32_t level = 0; level < w->L; level++) {
            const uint64_t *b = w->d_bits.p + (uint64_t)level * w->words_per_level;
            const uint32_t *r = w->d_rank.p + (uint64_t)level * (w->blocks_per_level + 1);
            if (level + 1 == w->L) {
                hipLaunchKernelGGL(k_wt_decode_level<true>, dim3(grid), dim3(256), 0, ctx->stream, in, (WtItem *)nullptr,
                                   d_out, b, r, w->d_C.p, w->ntotal, (uint32_t)w->nlist, w->L, level);
            } else {
                WtItem *out = (level & 1u) ? s_b.as<WtItem>() : s_a.as<WtItem>();
                hipLaunchKernelGGL(k_wt_decode_level<false>, dim3(grid), dim3(256), 0, ctx->stream, in, out,
                                   (uint64_t *)nullptr, b, r, w->d_C.p, w->ntotal, (uint32_t)w->nlist, w->L, level);
                in = out;
            }
        }
        VIDC_HIP(hipGetLastError());
        VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        VIDC_HIP(hipStreamSynchronize(ctx->stream));  // the scratch of this scope goes back to the pool below
        float ms2 = 0;
        (void)hipEventElapsedTime(&ms2, ctx->ev0, ctx->ev1);
        ctx->last_kernel_ms = ms2;
        return VIDC_OK;
    }
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    return VIDC_OK;
}

}  // extern "C"
