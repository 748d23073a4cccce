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
//
// wt_type = 1 (sdsl::wt_int<rrr_vector<63>>, custom_invlists_impl.h:104-113): every level is stored RRR-coded and the
// plain bits are dropped after the build -- 63-bit blocks, each a 6-bit class (its popcount c) and a
// ceil(log2 C(63, c))-bit offset (the block's index among the 63-bit words with c ones, combinatorial number
// system), plus one (offset-stream pointer, ones-before) sample per 32 blocks.  rank / access decode ONE block
// (sample -> add the classes of the blocks before it -> unrank the offset); select is a binary search over the
// samples, a scan over <= 32 classes and one block decode.  The bit order inside sdsl's structure is not pinned
// (sdsl is absent here); the structure, its parameters (block 63, sample rate 32) and what it answers are.
#include <algorithm>
#include <cmath>
#include <memory>

#include "bits.h"
#include "common.h"
#include "scan.h"

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
    DevBuf<uint64_t> d_bits;            // L * words_per_level                      (wt_type 0)
    DevBuf<uint32_t> d_rank;            // L * (blocks_per_level + 1): ones before each 512-bit block
    DevBuf<uint64_t> d_C;               // nlist + 1
    // wt_type 1: RRR-63 levels
    uint64_t rrr_nblk = 0, rrr_nsamp = 0, rrr_cls_wpl = 0;
    std::vector<uint64_t> rrr_off_base;  // [L + 1] first word of every level's offset stream
    DevBuf<uint32_t> d_cls;             // L * rrr_cls_wpl: packed 6-bit classes
    DevBuf<uint64_t> d_offs;            // offset streams, level after level
    DevBuf<uint32_t> d_ptr, d_rs;       // L * (rrr_nsamp + 1): bit position in the offset stream / ones before block 32 s
    // Query acceleration (both types; like the batch records of an Elias-Fano object it is not part of the reported size):
    // ones before the start of every node of every level -- level l has 2^l nodes, entry (l, p) at wt_nrank_base(l) + p,
    // entry (l, 2^l) = the level's total --, what every step of a select / decode walk asked the level's rank structure for.
    DevBuf<uint32_t> d_nrank;
    // two entries per node of every level (k_wt_node_ranks_from_C): what the build and the bulk decode add an element's rank to
    DevBuf<uint32_t> d_dstab;
    DevBuf<uint64_t> d_binom;           // C(n, k), n, k < 64 (row n at 64 n); behind it (d_binom + 4096) the 64 offset widths as bytes
};

namespace {

constexpr uint32_t BLK_WORDS = 8;  // 512-bit rank blocks
__host__ __device__ inline uint64_t wt_nrank_base(uint32_t level) { return ((uint64_t)1 << level) - 1u + level; }  // sum of (2^j + 1), j < level

// list_nos[id] = list number; validates the reference's asserts (ids ascending inside a list, < ntotal).
// A workgroup takes 4096 consecutive positions: two searches of the whole offset table find the lists of its first and last position,
// every position then searches only between them; the symbol goes out with a plain store and k_wt_check_syms looks for slots nobody
// wrote (ntotal ids below ntotal without a gap are a permutation).  Round 5: a 16-step search per id and an atomic exchange per
// id -- 0.69 ms for 16.8 M ids.
__global__ void __launch_bounds__(256) k_wt_scatter_syms(const uint64_t *__restrict__ ids, const uint64_t *__restrict__ offsets, uint32_t nlist,
                                                         uint64_t ntotal, uint32_t *__restrict__ syms, uint32_t *err) {
    __shared__ uint32_t lo_s, hi_s;
    bool bad = false;
    for (uint64_t base = (uint64_t)blockIdx.x * 4096u; base < ntotal; base += (uint64_t)gridDim.x * 4096u) {
        const uint64_t end = base + 4096u < ntotal ? base + 4096u : ntotal;
        if (threadIdx.x == 0) lo_s = find_list(offsets, nlist, base);
        if (threadIdx.x == 64) hi_s = find_list(offsets, nlist, end - 1);
        __syncthreads();
        const uint32_t llo = lo_s, lhi = hi_s;
        for (uint64_t g = base + threadIdx.x; g < end; g += 256u) {
            uint32_t lo = llo, hi = lhi + 1u;  // invariant: offsets[lo] <= g < offsets[hi]
            while (hi - lo > 1u) {
                const uint32_t mid = (lo + hi) >> 1;
                if (offsets[mid] <= g) lo = mid; else hi = mid;
            }
            const uint64_t id = ids[g];
            bool b = id >= ntotal;
            if (g > offsets[lo]) b |= ids[g - 1] >= id;  // assert(ids_data[i] > prev_id), :359
            if (!b) syms[id] = lo;
            bad |= b;
        }
        __syncthreads();
    }
    if (bad) atomicOr(err, 1u);
}
// a slot nobody wrote = an id that is missing = another one present twice
__global__ void k_wt_check_syms(const uint32_t *syms, uint64_t ntotal, uint32_t *err) {
    bool bad = false;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntotal; i += (uint64_t)gridDim.x * blockDim.x) bad |= syms[i] == 0xffffffffu;
    if (bad) atomicOr(err, 2u);
}

// ---- the same in two streaming passes (round 6, second half; 2^18 <= ntotal <= 2^24 ids, <= 2^18 lists).  The direct scatter writes 4 bytes
// at 16.8 M random places: 32-byte partial writes, 0.48 ms at 92 % of its wavefronts' cycles waiting (profiles/r06_pmc_wt.json).  Here the
// (id, list) pairs are first PARTITIONED by id >> 14 into <= 1024 buckets -- 256 workgroups over consecutive positions, per-workgroup
// histograms, no global atomic (rule 3) -- as one packed dword each (14 low id bits | list number; the list of a position from the list starts marked in a tile's LDS image and a scan,
// not from a search per position), then one workgroup
// per bucket places its pairs in a 64 KiB LDS image of the bucket's 16 384 ids and writes the image out with 16-byte stores; a slot nobody
// wrote is found on the way (k_wt_check_syms and the 0xff fill of the array are not needed).
#define VIDC_WTP_BSH 14u
#define VIDC_WTP_BUCKET (1u << VIDC_WTP_BSH)
#define VIDC_WTP_MAXB 1024u  // buckets
#define VIDC_WTP_NBLK 1024u  // workgroups of the two passes over the positions (16 384 positions each at 16.8 M ids)
__host__ __device__ inline uint64_t wtp_chunk(uint64_t ntotal) {  // positions per workgroup: whole 4096-position tiles
    return ((ntotal + VIDC_WTP_NBLK - 1u) / VIDC_WTP_NBLK + 4095u) & ~4095ull;
}
__global__ void __launch_bounds__(256) k_wt_part_hist(const uint64_t *__restrict__ ids, uint64_t ntotal, uint32_t *__restrict__ part) {
    __shared__ uint32_t h[VIDC_WTP_MAXB];
    for (uint32_t k = threadIdx.x; k < VIDC_WTP_MAXB; k += 256u) h[k] = 0u;
    __syncthreads();
    const uint64_t chunk = wtp_chunk(ntotal), lo = blockIdx.x * chunk, hi = lo + chunk < ntotal ? lo + chunk : ntotal;
    for (uint64_t g0 = lo; g0 < hi; g0 += 1024u) {
        uint64_t v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t g = g0 + (uint32_t)j * 256u + threadIdx.x;
            v[j] = g < hi ? ids[g] : ~0ull;
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (v[j] < ntotal) atomicAdd(&h[v[j] >> VIDC_WTP_BSH], 1u);
    }
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < VIDC_WTP_MAXB; k += 256u) part[blockIdx.x * VIDC_WTP_MAXB + k] = h[k];
}
// thread = bucket: its per-workgroup counts 128 at a time in registers -> exclusive prefix over the workgroups in place, the bucket's total to tot
// (64 threads per workgroup: 2048 memory instructions per wavefront go through ONE CU's address unit, so sixteen CUs share them)
__global__ void __launch_bounds__(64) k_wt_part_scan(uint32_t *part, uint32_t *tot) {
    const uint32_t k = blockIdx.x * 64u + threadIdx.x;
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < VIDC_WTP_NBLK; b0 += 128u) {
        uint32_t c[128];
#pragma unroll
        for (uint32_t b = 0; b < 128u; b++) c[b] = part[(b0 + b) * VIDC_WTP_MAXB + k];
        asm volatile("" ::: "memory");  // (all 128 loads in flight before the first add: scheduled in groups of ~32 they were 32 round trips)
#pragma unroll
        for (uint32_t b = 0; b < 128u; b++) {
            const uint32_t t = c[b];
            c[b] = run;
            run += t;
        }
#pragma unroll
        for (uint32_t b = 0; b < 128u; b++) part[(b0 + b) * VIDC_WTP_MAXB + k] = c[b];
    }
    tot[k] = run;
}
__global__ void __launch_bounds__(256) k_wt_part_scatter(const uint64_t *__restrict__ ids, const uint64_t *__restrict__ offsets, uint32_t nlist,
                                                         uint64_t ntotal, const uint32_t *__restrict__ part, const uint32_t *__restrict__ tot,
                                                         uint32_t *__restrict__ bstart, uint32_t *__restrict__ pairs, uint32_t *err) {
    __shared__ uint32_t cur[VIDC_WTP_MAXB];
    __shared__ uint32_t tsum[256];
    __shared__ uint32_t lidx[4096 + 256];       // per position of a tile: list starts marked, then the number of the position's list
    __shared__ uint32_t lo_s, hi_s;
    uint32_t bs4[4];  // starts of this thread's four buckets
    {   // bucket starts = exclusive scan of the 1024 totals: four per thread + a scan over the threads
        uint32_t t4[4], s = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { t4[j] = tot[threadIdx.x * 4u + (uint32_t)j]; s += t4[j]; }
        tsum[threadIdx.x] = s;
        __syncthreads();
        for (uint32_t o = 1; o < 256u; o <<= 1) {
            const uint32_t v = threadIdx.x >= o ? tsum[threadIdx.x - o] : 0u;
            __syncthreads();
            tsum[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t run = tsum[threadIdx.x] - s;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t k = threadIdx.x * 4u + (uint32_t)j;
            bs4[j] = run;
            if (blockIdx.x == 0) bstart[k] = run;
            run += t4[j];
        }
        if (blockIdx.x == 0 && threadIdx.x == 255u) bstart[VIDC_WTP_MAXB] = run;
    }
    bool bad = false;
    // chunks blockIdx.x, blockIdx.x + gridDim.x, ... (the launch has one workgroup per chunk; fewer, each walking several chunks -- so that
    // fewer of the 64-byte bucket runs are open at once and their lines fill up in the L2 -- measured no faster: 143 / 221 us with 512 / 256)
    for (uint32_t cb = blockIdx.x; cb < VIDC_WTP_NBLK; cb += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t k = threadIdx.x * 4u + (uint32_t)j;
        cur[k] = bs4[j] + part[cb * VIDC_WTP_MAXB + k];
    }
    __syncthreads();
    const uint64_t chunk = wtp_chunk(ntotal), c_lo = cb * chunk, c_hi = c_lo + chunk < ntotal ? c_lo + chunk : ntotal;
    for (uint64_t base = c_lo; base < c_hi; base += 4096u) {
        const uint64_t end = base + 4096u < c_hi ? base + 4096u : c_hi;
        // the tile's 16 ids per thread (and the id in front of each: the order check) are requested before anything waits
        uint64_t idv[16], prv[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint64_t g = base + (uint32_t)j * 256u + threadIdx.x;
            idv[j] = g < end ? ids[g] : ~0ull;
            prv[j] = (g < end && g) ? ids[g - 1] : 0ull;
        }
        if (threadIdx.x == 0) lo_s = find_list(offsets, nlist, base);
        if (threadIdx.x == 64) hi_s = find_list(offsets, nlist, end - 1);
#pragma unroll
        for (int j = 0; j < 17; j++) lidx[(uint32_t)j * 256u + threadIdx.x] = 0u;
        __syncthreads();
        const uint32_t llo = lo_s, lhi = hi_s, nl = lhi - llo + 1u;  // lists llo .. lhi
        const bool first_starts = offsets[llo] == base;  // (the tile's first position opens list llo: no id in front of it to compare with)
        // list of a position = llo + the number of lists llo + 1 .. lhi that start at or in front of it: every such list marks its start
        // (all of them lie inside the tile), a scan over the tile's 4096 positions does the rest -- no search per position (a binary search
        // in LDS per position, a chain of dependent reads sixteen times in a row per thread, was most of this kernel's 0.2 ms)
        for (uint32_t k = 1u + threadIdx.x; k < nl; k += 256u) {
            const uint32_t rel = (uint32_t)(offsets[llo + k] - base);
            atomicAdd(&lidx[rel + (rel >> 4)], 1u);  // (entry p at p + p / 16: thread t's sixteen entries of the scan below are conflict-free)
        }
        __syncthreads();
        {
            uint32_t f[16], sum = 0;
#pragma unroll
            for (int j = 0; j < 16; j++) { f[j] = lidx[threadIdx.x * 17u + (uint32_t)j]; sum += f[j]; }
            uint32_t incl = sum;
            const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t u = (uint32_t)__shfl_up((int)incl, o, 64);
                if (lane >= (uint32_t)o) incl += u;
            }
            if (lane == 63u) tsum[wave] = incl;
            __syncthreads();
            uint32_t run = incl - sum;
            for (uint32_t w2 = 0; w2 < wave; w2++) run += tsum[w2];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                run += f[j];
                lidx[threadIdx.x * 17u + (uint32_t)j] = run | (f[j] ? 0x80000000u : 0u);  // bit 31: a list starts here
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t rel = (uint32_t)j * 256u + threadIdx.x;
            const uint64_t g = base + rel;
            if (g >= end) continue;
            const uint32_t v = lidx[rel + (rel >> 4)];
            const uint64_t id = idv[j];
            if (id < ntotal) {  // (counted by k_wt_part_hist: takes its slot whatever else is wrong with it)
                if (!(v >> 31) && !(rel == 0u && first_starts)) bad |= prv[j] >= id;  // assert(ids_data[i] > prev_id), :359 (not for the first id of a list)
                const uint32_t slot = atomicAdd(&cur[id >> VIDC_WTP_BSH], 1u);
                pairs[slot] = ((uint32_t)id & (VIDC_WTP_BUCKET - 1u)) | ((llo + (v & 0x7fffffffu)) << VIDC_WTP_BSH);
            } else {
                bad = true;
            }
        }
        __syncthreads();
    }
    }
    if (bad) atomicOr(err, 1u);
}
__global__ void __launch_bounds__(256) k_wt_part_place(const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ bstart, uint64_t ntotal,
                                                       uint32_t *__restrict__ syms, uint32_t *err) {
    extern __shared__ __attribute__((aligned(16))) uint32_t img[];  // VIDC_WTP_BUCKET dwords
    for (uint32_t i = threadIdx.x; i < VIDC_WTP_BUCKET / 4u; i += 256u) ((uint4 *)img)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
    __syncthreads();
    const uint32_t j0 = bstart[blockIdx.x], j1 = bstart[blockIdx.x + 1u];
    for (uint32_t j = j0 + threadIdx.x; j < j1; j += 1024u) {
        uint32_t v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = j + (uint32_t)q * 256u < j1 ? pairs[j + (uint32_t)q * 256u] : ~0u;
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (j + (uint32_t)q * 256u < j1) img[v[q] & (VIDC_WTP_BUCKET - 1u)] = v[q] >> VIDC_WTP_BSH;
    }
    __syncthreads();
    const uint64_t first = (uint64_t)blockIdx.x << VIDC_WTP_BSH;
    const uint32_t cnt = ntotal - first < VIDC_WTP_BUCKET ? (uint32_t)(ntotal - first) : VIDC_WTP_BUCKET;
    bool bad = false;
    for (uint32_t i = threadIdx.x * 4u; i < cnt; i += 1024u) {
        const uint4 v = *(const uint4 *)(img + i);
        if (i + 4u <= cnt) {
            bad |= v.x == ~0u || v.y == ~0u || v.z == ~0u || v.w == ~0u;
            *(uint4 *)(syms + first + i) = v;  // (the array and the bucket's first id are 16-byte aligned)
        } else {
            const uint32_t e[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t q = 0; i + q < cnt; q++) { bad |= e[q] == ~0u; syms[first + i + q] = e[q]; }
        }
    }
    if (bad) atomicOr(err, 2u);  // a slot nobody wrote = an id that is missing = another one present twice
}

// ---- one level of the tree in ONE pass over its symbols (round 6).  Before: four launches per level -- a thread per 64-bit word gathering 64
// symbols 4 bytes at a time, a SINGLE workgroup scanning the whole rank directory, a node-rank kernel, and a partition whose every element
// asked the fresh rank directory three times -- 0.4 ms per level on 16.8 M ids, 6.4 ms for the 16 levels of a 65 536-list tree.
// Now a tile of 4096 consecutive positions per workgroup: the symbols come in coalesced, a wavefront's ballot IS the bit-vector word of its
// 64 positions, the ones before the tile come from a chained scan over the tiles (decoupled look-back: a tile publishes its own count, then
// its inclusive prefix; a successor adds counts until it meets a prefix; tiles take their numbers from a counter in start order, so every
// wait ends), and an element's slot in the next level follows from its rank and from tables that need only the symbol start table C: the
// ones before every node start (k_wt_node_ranks_from_C, all levels in one launch before the first pass).
#define VIDC_WT_EPT 64u                    // elements per thread
#define VIDC_WT_TILE (256u * VIDC_WT_EPT)  // 16 384 positions = 256 bit-vector words = 32 rank blocks per workgroup
struct WtScanState {  // per tile: bit 63 = inclusive prefix, bit 62 = the tile's own count; 0 = nothing yet.  [ntiles] | ticket
    unsigned long long v;
};
// nrank[(level, p)] = ones before the start of node p of the level (p = 0 .. 2^level; entry 2^level = the level's ones): the elements of the
// right children of the nodes before p, i.e. sums of symbol counts.  One workgroup per level.
// dstab[2 (level, p) + b] (build only): what an element of node p with bit b adds its zero-rank / one-rank to for its slot in the next level:
// slot = b ? (node start + zeros of the node - ones before the node) + ones before the element
//          : (ones before the node) + zeros before the element        -- ONE table entry per element instead of four (two C, two nrank).
// VIDC_WT_NR_G workgroups per level: workgroup g owns a run of nodes, adds up the right children of every node in FRONT of its run itself
// (coalesced, independent loads: redundant work, 128 loads per thread at most for a 65 536-list tree) and then scans its run 256 nodes
// per trip.  One workgroup per level -- 64 trips for the last level, or a run of 128 nodes per thread with every access 2 KiB from its
// neighbour's -- kept the whole last level on ONE CU's address unit: 92-98 us whatever the loop looked like.
#define VIDC_WT_NR_G 16u
__global__ void __launch_bounds__(256) k_wt_node_ranks_from_C(const uint64_t *__restrict__ C, uint32_t nlist, uint32_t L, uint32_t *__restrict__ nrank,
                                                              uint32_t *__restrict__ dstab) {
    __shared__ uint64_t wsum[4];
    const uint32_t level = blockIdx.x / VIDC_WT_NR_G, g = blockIdx.x % VIDC_WT_NR_G, shn = L - level;  // a node of this level spans 2^shn symbols
    uint32_t *out = nrank + wt_nrank_base(level);
    uint32_t *dst = dstab + 2 * wt_nrank_base(level);
    const uint64_t nodes = 1ull << level;
    const uint64_t B = ((nodes + VIDC_WT_NR_G - 1u) / VIDC_WT_NR_G + 255u) & ~255ull;
    const uint64_t first = g * B, last = first + B < nodes ? first + B : nodes;
    if (first >= nodes) return;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    auto csym = [&](uint64_t sym) -> uint64_t { return C[sym > nlist ? nlist : sym]; };
    auto block_sum = [&](uint64_t v) -> uint64_t {  // every thread gets the sum over the workgroup
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if (lane == 0) wsum[wave] = v;
        __syncthreads();
        return wsum[0] + wsum[1] + wsum[2] + wsum[3];
    };
    uint64_t acc = 0;  // right children of the nodes in front of the run
    for (uint64_t q0 = 0; q0 < first; q0 += 1024u) {
        uint64_t m[4], h[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint64_t q = q0 + (uint32_t)j * 256u + t;
            m[j] = q < first ? csym((2 * q + 1) << (shn - 1)) : 0;
            h[j] = q < first ? csym((2 * q + 2) << (shn - 1)) : 0;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc += h[j] - m[j];
    }
    uint64_t carry = block_sum(acc);
    for (uint64_t p0 = first; p0 < last; p0 += 256u) {
        const uint64_t p = p0 + t;
        const bool in = p < last;
        const uint64_t ns = in ? csym(p << shn) : 0, mid = in ? csym((2 * p + 1) << (shn - 1)) : 0, hi = in ? csym((p + 1) << shn) : 0;
        const uint64_t rs = hi - mid;  // size of the node's right child
        uint64_t incl = rs;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t u = __shfl_up(incl, o, 64);
            if (lane >= (uint32_t)o) incl += u;
        }
        __syncthreads();
        if (lane == 63u) wsum[wave] = incl;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w2 = 0; w2 < wave; w2++) before += wsum[w2];
        if (in) {
            const uint64_t r_ns = before + incl - rs, zeros = mid - ns;
            out[p] = (uint32_t)r_ns;
            dst[2 * p] = (uint32_t)r_ns;
            dst[2 * p + 1] = (uint32_t)(ns + zeros - r_ns);
        }
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
    if (last == nodes && t == 0) out[nodes] = (uint32_t)carry;
}

// syms_in: the level's order.  bits / rank: the level's bit vector (nwords words, zero padded) and its directory of ones before every
// 512-bit block (nblocks + 1 entries).  syms_out (nullptr on the last level): the next level's order.  state: zeroed, ntiles + 1 entries.
__global__ void __launch_bounds__(256) k_wt_level(const uint32_t *__restrict__ syms_in, uint32_t *__restrict__ syms_out, uint64_t ntotal,
                                                  uint32_t nlist, uint32_t L, uint32_t level,
                                                  const uint32_t *__restrict__ dstab_l, uint64_t *__restrict__ bits, uint64_t nwords,
                                                  uint32_t *__restrict__ rank, uint64_t nblocks, unsigned long long *state, uint32_t ntiles) {
    __shared__ uint64_t words[VIDC_WT_EPT * 4];
    __shared__ uint32_t wpre[VIDC_WT_EPT * 4 + 1];  // ones before every word of the tile
    __shared__ uint32_t wsum[4];
    __shared__ uint32_t tile_s;
    __shared__ unsigned long long before_s;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) tile_s = (uint32_t)atomicAdd(&state[ntiles], 1ull);
    __syncthreads();
    const uint32_t tile = tile_s;
    const uint64_t base = (uint64_t)tile * VIDC_WT_TILE;
    const uint32_t bsh = L - 1u - level;  // the symbol bit this level stores
    uint32_t sym[VIDC_WT_EPT];
    uint32_t inword[VIDC_WT_EPT / 4];  // ones before the element inside its word (< 64): four per register
    uint64_t mybits = 0;
    // (every load of the tile is issued before the first ballot: a predicated load inside the ballot loop made the compiler wait for each
    // of the 64 loads in turn -- 63 us of a level's 100 were this loop)
#pragma unroll
    for (int j = 0; j < (int)VIDC_WT_EPT; j++) {
        const uint64_t i = base + (uint64_t)j * 256u + t;
        sym[j] = syms_in[i < ntotal ? i : ntotal - 1u];
    }
#pragma unroll
    for (int j = 0; j < (int)VIDC_WT_EPT; j++) {
        const uint64_t i = base + (uint64_t)j * 256u + t;
        const bool b = i < ntotal && ((sym[j] >> bsh) & 1u);
        const uint64_t m = __ballot(b);
        const uint32_t iw = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if ((j & 3) == 0) inword[j >> 2] = iw; else inword[j >> 2] |= iw << (8 * (j & 3));
        mybits |= b ? 1ull << j : 0ull;
        if (lane == 0) words[j * 4 + wave] = m;
    }
    __syncthreads();
    // ones before every word of the tile (256 words: a thread each, scan inside the wavefronts, then across the four)
    {
        const uint32_t c = (uint32_t)__builtin_popcountll(words[t]);
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, o, 64);
            if (lane >= (uint32_t)o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t wbefore = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) wbefore += k < wave ? wsum[k] : 0u;
        wpre[t] = wbefore + incl - c;
        if (t == 255) wpre[256] = wbefore + incl;
        // the bit-vector words of the tile: one coalesced store
        const uint64_t w = (uint64_t)tile * (VIDC_WT_EPT * 4) + t;
        if (w < nwords) bits[w] = words[t];
    }
    __syncthreads();
    const uint32_t tile_ones = wpre[VIDC_WT_EPT * 4];
    // chained scan: publish the tile's count, look back for the ones before the tile, publish the inclusive prefix.  The look-back reads 256
    // predecessors per round trip (one per thread): the first tiles of a launch all start together and none of them holds a prefix
    // yet, so the tile numbered k makes k / 256 rounds -- with 64 per round the 2048 resident tiles of a 16.8 M-id level finished their
    // look-backs at 2 us x k / 64, and the level took 100 us for 30 us of traffic.
    {
        __shared__ unsigned long long w_all[4], w_upto[4];
        __shared__ uint32_t w_stop[4], done_s;
        if (t == 0) {
            atomicExch(&state[tile], (1ull << 62) | tile_ones);
            before_s = 0;
            done_s = tile == 0u ? 1u : 0u;
        }
        __syncthreads();
        int64_t k0 = (int64_t)tile - 1;
        while (!done_s) {
            const int64_t k = k0 - (int64_t)t;
            unsigned long long x = 0;
            if (k >= 0) {
                do { x = __hip_atomic_load(&state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(x >> 62));
            }
            const uint64_t pm = __ballot(k >= 0 && (x >> 63));  // predecessors that hold an inclusive prefix
            const uint32_t stop = pm ? (uint32_t)__builtin_ctzll(pm) : 64u;  // the nearest of this wavefront's
            const unsigned long long v = k >= 0 ? (x & ((1ull << 62) - 1ull)) : 0ull;
            unsigned long long all = v, upto = lane <= stop ? v : 0ull;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                all += __shfl_xor(all, o, 64);
                upto += __shfl_xor(upto, o, 64);
            }
            if (lane == 0) { w_all[wave] = all; w_upto[wave] = upto; w_stop[wave] = stop; }
            __syncthreads();
            if (t == 0) {
                unsigned long long acc = before_s;
                bool found = false;
                for (int w = 0; w < 4 && !found; w++) {  // wavefront w holds predecessors k0 - 64 w .. k0 - 64 w - 63
                    found = w_stop[w] < 64u;
                    acc += found ? w_upto[w] : w_all[w];
                }
                before_s = acc;
                if (found || k0 - 256 < 0) done_s = 1u;
            }
            __syncthreads();
            k0 -= 256;
        }
        if (t == 0) atomicExch(&state[tile], (1ull << 63) | (before_s + tile_ones));
    }
    const uint64_t before = before_s;
    // rank directory: ones before every 512-bit block (8 words) of the tile; the entry behind the last block = the level's ones
    if (t < VIDC_WT_EPT / 2) {
        const uint64_t blk = (uint64_t)tile * (VIDC_WT_EPT / 2) + t;
        if (blk < nblocks) rank[blk] = (uint32_t)(before + wpre[t * 8u]);
    }
    if (tile + 1u == ntiles && t == 0) rank[nblocks] = (uint32_t)(before + tile_ones);
    if (!syms_out) return;
    // the next level's order: zeros of a node keep their order in its left child, ones in its right child
    const uint32_t shn = L - level;  // a node of this level spans 2^shn symbols
#pragma unroll
    for (int j0 = 0; j0 < (int)VIDC_WT_EPT; j0 += 16) {
        uint32_t tb[16];  // sixteen table entries in flight, then sixteen stores
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int j = j0 + jj;
            const uint32_t p = shn >= 32u ? 0u : sym[j] >> shn;
            tb[jj] = dstab_l[2u * p + ((uint32_t)(mybits >> j) & 1u)];
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int j = j0 + jj;
            const uint64_t i = base + (uint64_t)j * 256u + t;
            const uint32_t bit = (uint32_t)(mybits >> j) & 1u;
            const uint64_t r_i = before + wpre[j * 4 + wave] + ((inword[j >> 2] >> (8 * (j & 3))) & 0xffu);  // ones before the element
            const uint64_t dst = (uint64_t)tb[jj] + (bit ? r_i : i - r_i);
            if (i < ntotal) syms_out[dst] = sym[j];
        }
    }
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


// ---------------------------------------------------------------------------------------------------------------
// bit-vector views: what the query kernels need from a level, over plain bits or over the RRR coding
struct BvPlain {
    const uint64_t *bits;
    const uint32_t *rank;
    uint64_t nblocks, nwords;
    __device__ __forceinline__ uint64_t rank_1(uint64_t i) const { return rank1(bits, rank, i); }
    __device__ __forceinline__ uint64_t rank_1_bit(uint64_t i, bool &bit) const {
        bit = (bits[i >> 6] >> (i & 63)) & 1ull;
        return rank1(bits, rank, i);
    }
    __device__ __forceinline__ uint64_t select(uint64_t j, bool one) const { return select_bit(bits, rank, nblocks, nwords, j, one); }
    // the same when the answer is known to lie in [p0, p1) (a node of the level): the directory search starts from the node's blocks
    __device__ __forceinline__ uint64_t select_in(uint64_t j, bool one, uint64_t p0, uint64_t p1) const {
        uint64_t lo = p0 / (64 * BLK_WORDS), hi = p1 / (64 * BLK_WORDS) + 1;
        if (hi > nblocks) hi = nblocks;
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
    static constexpr bool kRrr = false;
};
struct WtPlainView {
    static constexpr bool kRrrView = false;
    const uint64_t *bits;
    const uint32_t *rank;
    uint64_t wpl, bpl;
    __device__ __forceinline__ BvPlain level(uint32_t l) const {
        return BvPlain{bits + (uint64_t)l * wpl, rank + (uint64_t)l * (bpl + 1), bpl, wpl};
    }
};

constexpr uint32_t RRR_B = 63, RRR_K = 32;
struct RrrTab { uint8_t ow[64]; };  // offset width of class c: ceil(log2 C(63, c))

// index of the 63-bit word v (c ones) among the words with c ones: sum over its ones (ascending, i = 1..c) of C(p_i, i)
__device__ __forceinline__ uint64_t rrr_rank_word(uint64_t v, const uint64_t *binom) {
    uint64_t o = 0;
    uint32_t i = 1;
    while (v) {
        const uint32_t p = (uint32_t)__builtin_ctzll(v);
        o += binom[p * 64u + i];
        i++;
        v &= v - 1;
    }
    return o;
}
__device__ __forceinline__ uint64_t rrr_unrank_word(uint32_t c, uint64_t o, const uint64_t *binom) {
    if (c == 0) return 0;
    if (c == RRR_B) return (1ull << RRR_B) - 1ull;
    uint64_t v = 0;
    for (int p = (int)RRR_B - 1; c; p--) {
        const uint64_t b = binom[(uint32_t)p * 64u + c];  // words with c ones below position p
        if (o >= b) { v |= 1ull << p; o -= b; c--; }
    }
    return v;
}
// ones of the block (class c, offset o) below position rem, and the bit at rem: the unranking only has to walk the positions
// from the top down to rem (what is left of c afterwards sits below)
__device__ __forceinline__ uint32_t rrr_rank_below(uint32_t c, uint64_t o, uint32_t rem, bool &bit, const uint64_t *binom) {
    bit = false;
    if (c == 0) return 0;
    if (c == RRR_B) { bit = rem < RRR_B; return rem; }
    for (int p = (int)RRR_B - 1; p >= (int)rem && c; p--) {
        const uint64_t b = binom[(uint32_t)p * 64u + c];
        if (o >= b) {
            o -= b;
            c--;
            if ((uint32_t)p == rem) { bit = true; }
        }
    }
    return c;
}
struct BvRrr {
    const uint32_t *cls;
    const uint64_t *offs;
    const uint32_t *ptr, *rs;
    const uint64_t *binom;
    const uint8_t *ow;  // offset width by class (memory, not a by-value table: a dynamically indexed kernel argument goes to scratch)
    uint64_t nblk, nsamp, nbits;
    __device__ __forceinline__ uint32_t cls_at(uint64_t b) const {
        const uint64_t bp = 6 * b;
        uint32_t w = cls[bp >> 5] >> (bp & 31);
        if ((bp & 31) > 26) w |= cls[(bp >> 5) + 1] << (32 - (bp & 31));
        return w & 63u;
    }
    __device__ __forceinline__ uint64_t offset_at(uint64_t bp, uint32_t c) const {  // offset field of the block starting at bit bp
        const uint32_t wd = ow[c];
        uint64_t o = 0;
        if (wd) {
            o = offs[bp >> 6] >> (bp & 63);
            if ((bp & 63) + wd > 64) o |= offs[(bp >> 6) + 1] << (64 - (bp & 63));
            o &= (1ull << wd) - 1ull;
        }
        return o;
    }
    __device__ __forceinline__ uint64_t word_at(uint64_t bp, uint32_t c) const { return rrr_unrank_word(c, offset_at(bp, c), binom); }
    // position inside the block (class c, offset o, `len` existing bits) of its (k+1)-th bit equal to `one`: the unranking walks
    // the positions from the top and stops at the wanted bit (the k-th from below is the (count - 1 - k)-th from above) instead
    // of rebuilding the whole word and clearing k bits
    __device__ __forceinline__ uint32_t select_in_block(uint32_t c, uint64_t o, uint32_t k, bool one, uint32_t len) const {
        uint32_t t = (one ? c : len - c) - 1u - k;  // wanted bit, counted from the top among the existing bits of its kind
        for (int p = (int)RRR_B - 1; p >= 0; p--) {
            if (c == 0) return one ? 0u : (uint32_t)p - t - ((uint32_t)p >= len ? (uint32_t)p + 1u - len : 0u);  // only zeros below
            const uint64_t b = binom[(uint32_t)p * 64u + c];
            const bool bit = o >= b;
            if (bit) { o -= b; c--; }
            if ((uint32_t)p < len && bit == one) {
                if (t == 0) return (uint32_t)p;
                t--;
            }
        }
        return 0u;
    }
    __device__ __forceinline__ uint64_t rank_1_bit(uint64_t i, bool &bit) const {
        const uint64_t blk = i / RRR_B, s = blk / RRR_K;
        uint64_t r = rs[s], bp = ptr[s];
        for (uint64_t b = s * RRR_K; b < blk; b++) {
            const uint32_t c = cls_at(b);
            r += c;
            bp += ow[c];
        }
        const uint32_t rem = (uint32_t)(i - blk * RRR_B);
        bit = false;
        if (blk < nblk) {
            const uint32_t c = cls_at(blk);
            r += rrr_rank_below(c, offset_at(bp, c), rem, bit, binom);
        }
        return r;
    }
    __device__ __forceinline__ uint64_t rank_1(uint64_t i) const {
        bool b;
        return rank_1_bit(i, b);
    }
    __device__ __forceinline__ uint64_t before(uint64_t s, bool one) const {  // ones / zeros in blocks [0, 32 s)
        const uint64_t r = rs[s];
        if (one) return r;
        const uint64_t pos = s * RRR_K * RRR_B;
        return (pos < nbits ? pos : nbits) - r;
    }
    __device__ __forceinline__ uint64_t select(uint64_t j, bool one) const { return select_in(j, one, 0, nbits); }
    // (j+1)-th bit equal to `one`, known to lie in [p0, p1) (a node of the level).  The sample search starts from the node's
    // samples (a node of a deep level spans one or two), the 32 classes behind the sample -- 192 bits, word-aligned: sample s
    // starts at bit 6 * 32 s -- come with three independent 8-byte loads and are walked in registers, and the unranking stops at
    // the wanted bit.  The first version made up to 13 + 2 * 31 + 63 DEPENDENT loads per level (directory, one class and one
    // width per block, one binomial per position): 5 ms for a select on a 16-level tree, whatever the number of queries.
    __device__ __forceinline__ uint64_t select_in(uint64_t j, bool one, uint64_t p0, uint64_t p1) const {
        uint64_t lo = p0 / (RRR_K * RRR_B), hi = p1 / (RRR_K * RRR_B) + 1;
        if (hi > nsamp) hi = nsamp;
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (before(mid, one) <= j) lo = mid; else hi = mid;
        }
        uint64_t seen = before(lo, one), bp = ptr[lo];
        const uint2 *cw = (const uint2 *)(cls + 6 * lo);  // (cls_wpl has a pad word; the level's last sample may read into it)
        const uint2 q0 = cw[0], q1 = cw[1], q2 = cw[2];
        const uint32_t w[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, 0u};
        bool found = false;
        uint64_t fb = 0, fbp = 0, fseen = 0;
        uint32_t fc = 0, flen = 0;
#pragma unroll
        for (uint32_t k = 0; k < RRR_K; k++) {
            const uint64_t b = lo * RRR_K + k;
            const uint32_t bit0 = 6u * k, wi = bit0 >> 5, sh = bit0 & 31u;
            uint32_t c = w[wi] >> sh;
            if (sh > 26u) c |= w[wi + 1] << (32u - sh);
            c &= 63u;
            if (!found && b < nblk) {
                const uint64_t len = nbits - b * RRR_B < RRR_B ? nbits - b * RRR_B : RRR_B;  // bits of this block that exist
                const uint64_t cnt = one ? c : len - c;
                if (seen + cnt > j) { found = true; fb = b; fbp = bp; fseen = seen; fc = c; flen = (uint32_t)len; }
                else { seen += cnt; bp += ow[c]; }
            }
        }
        if (found) return fb * RRR_B + select_in_block(fc, offset_at(fbp, fc), (uint32_t)(j - fseen), one, flen);
        for (uint64_t b = (lo + 1) * RRR_K; b < nblk; b++) {  // (not reached while the samples are consistent)
            const uint32_t c = cls_at(b);
            const uint64_t len = nbits - b * RRR_B < RRR_B ? nbits - b * RRR_B : RRR_B;
            const uint64_t cnt = one ? c : len - c;
            if (seen + cnt > j) return b * RRR_B + select_in_block(c, offset_at(bp, c), (uint32_t)(j - seen), one, (uint32_t)len);
            seen += cnt;
            bp += ow[c];
        }
        return ~0ull;
    }
    static constexpr bool kRrr = true;
};
struct WtRrrView {
    static constexpr bool kRrrView = true;
    const uint32_t *cls;
    const uint64_t *offs;
    const uint32_t *ptr, *rs;
    const uint64_t *binom;
    uint64_t cls_wpl, nblk, nsamp, nbits;
    uint64_t off_base[32];
    __device__ __forceinline__ BvRrr level(uint32_t l) const {
        return BvRrr{cls + (uint64_t)l * cls_wpl, offs + off_base[l], ptr + (uint64_t)l * (nsamp + 1), rs + (uint64_t)l * (nsamp + 1),
                     binom, (const uint8_t *)(binom + 64 * 64), nblk, nsamp, nbits};
    }
};

// ---- RRR build kernels (one thread per 63-bit block of a level's plain bits)
__device__ __forceinline__ uint64_t rrr_get63(const uint64_t *bits, uint64_t b) {
    const uint64_t pos = b * RRR_B;
    uint64_t v = bits[pos >> 6] >> (pos & 63);
    if ((pos & 63) + RRR_B > 64) v |= bits[(pos >> 6) + 1] << (64 - (pos & 63));
    return v & ((1ull << RRR_B) - 1ull);
}
__global__ void k_rrr_classes(const uint64_t *bits, uint64_t nblk, RrrTab tab, uint32_t *cls, uint32_t *width) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        const uint32_t c = (uint32_t)__builtin_popcountll(rrr_get63(bits, b));
        width[b] = tab.ow[c];
        const uint64_t bp = 6 * b;
        atomicOr(&cls[bp >> 5], c << (bp & 31));
        if ((bp & 31) > 26) atomicOr(&cls[(bp >> 5) + 1], c >> (32 - (bp & 31)));
    }
}
__global__ void k_rrr_offsets(const uint64_t *bits, uint64_t nblk, RrrTab tab, const uint64_t *binom, const uint64_t *bitpos,
                              unsigned long long *offs) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk; b += stride) {
        const uint64_t v = rrr_get63(bits, b);
        const uint32_t wd = tab.ow[__builtin_popcountll(v)];
        if (!wd) continue;
        const uint64_t o = rrr_rank_word(v, binom), bp = bitpos[b];
        atomicOr(&offs[bp >> 6], (unsigned long long)(o << (bp & 63)));
        if ((bp & 63) + wd > 64) atomicOr(&offs[(bp >> 6) + 1], (unsigned long long)(o >> (64 - (bp & 63))));
    }
}
__global__ void k_rrr_samples(const uint64_t *bits, const uint32_t *rank, uint64_t nbits, uint64_t nblk, uint64_t nsamp,
                              const uint64_t *bitpos, uint32_t *ptr, uint32_t *rs) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s <= nsamp; s += stride) {
        const uint64_t b = s * RRR_K < nblk ? s * RRR_K : nblk;
        const uint64_t pos = b * RRR_B < nbits ? b * RRR_B : nbits;
        ptr[s] = (uint32_t)bitpos[b];
        rs[s] = (uint32_t)rank1(bits, rank, pos);
    }
}

// Bulk decode of a compressed tree: a level is turned back into plain bits ONCE -- a thread per 63-bit block: the sample before
// it, the classes up to it, one unranking -- with the 512-bit rank directory the plain kernels use (a thread per entry, from
// the samples and classes), and the level's partition then runs on the plain view.  The per-element form unranked a block for
// each of the three ranks every element asks its level for: 16 M ids decoded in 23 ms, 11 x the plain tree.
__global__ void k_rrr_expand(BvRrr bv, unsigned long long *bits) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < bv.nblk; b += stride) {
        const uint64_t s = b / RRR_K;
        uint64_t bp = bv.ptr[s];
        for (uint64_t k = s * RRR_K; k < b; k++) bp += bv.ow[bv.cls_at(k)];
        const uint32_t c = bv.cls_at(b);
        if (!c) continue;  // (the scratch is zeroed)
        const uint64_t v = bv.word_at(bp, c), pos = b * RRR_B;
        atomicOr(&bits[pos >> 6], (unsigned long long)(v << (pos & 63)));
        if ((pos & 63) + RRR_B > 64) atomicOr(&bits[(pos >> 6) + 1], (unsigned long long)(v >> (64 - (pos & 63))));
    }
}
__global__ void k_rrr_rankdir(BvRrr bv, uint64_t nblocks, uint32_t *rank) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j <= nblocks; j += stride) {
        const uint64_t pos = j * 64 * BLK_WORDS;
        rank[j] = (uint32_t)bv.rank_1(pos < bv.nbits ? pos : bv.nbits);
    }
}

// the tables an RRR view looks up once per position / per block, in LDS for the kernels that walk levels per query (the binomials
// are a dependent load per unranked position: ~30 of them per level and query)
template <class View>
__device__ __forceinline__ void wt_stage_tables(const View &vw, uint64_t *s_binom, uint8_t *s_ow) {
    if constexpr (View::kRrrView) {
        for (uint32_t i = threadIdx.x; i < 64u * 64u; i += blockDim.x) s_binom[i] = vw.binom[i];
        const uint8_t *ow = (const uint8_t *)(vw.binom + 64 * 64);
        for (uint32_t i = threadIdx.x; i < 64u; i += blockDim.x) s_ow[i] = ow[i];
        __syncthreads();
    }
}
template <class Bv>
__device__ __forceinline__ void wt_use_tables(Bv &bv, const uint64_t *s_binom, const uint8_t *s_ow) {
    if constexpr (Bv::kRrr) { bv.binom = s_binom; bv.ow = s_ow; }
}

// one thread per query: id of the (k+1)-th element of list c
template <class View>
__global__ void k_wt_select(View vw, const uint64_t *C, const uint32_t *__restrict__ nrank, uint32_t nlist, uint32_t L, uint64_t m,
                            const uint64_t *list_nos, const uint64_t *offs, int64_t *out) {
    __shared__ uint64_t s_binom[View::kRrrView ? 64 * 64 : 1];
    __shared__ uint8_t s_ow[64];
    wt_stage_tables(vw, s_binom, s_ow);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; q < m; q += stride) {
        const uint32_t c = (uint32_t)list_nos[q];
        uint64_t pos = offs[q];
        for (int level = (int)L - 1; level >= 0; level--) {
            const uint32_t sh = L - (uint32_t)level;
            const uint64_t p = sh >= 32 ? 0 : (uint64_t)(c >> sh);
            const uint64_t s_lo = sh >= 32 ? 0 : (p << sh);
            uint64_t s_hi = sh >= 32 ? nlist : ((p + 1) << sh);
            if (s_hi > nlist) s_hi = nlist;
            const uint64_t ns = C[s_lo], ne = C[s_hi];
            const bool bit = (c >> (L - 1u - (uint32_t)level)) & 1u;
            auto bv = vw.level((uint32_t)level);
            wt_use_tables(bv, s_binom, s_ow);
            const uint64_t r_ns = nrank[wt_nrank_base((uint32_t)level) + p];  // (== bv.rank_1(ns), from the build)
            const uint64_t before = bit ? r_ns : ns - r_ns;
            pos = bv.select_in(before + pos, bit, ns, ne) - ns;
        }
        out[q] = (int64_t)pos;
    }
}

// get_ids of the requested lists (custom_invlists_impl.cpp:381-392 loops get_single_id): one thread per output slot,
// its request item found in the m + 1 output offsets
template <class View>
__global__ void k_wt_decode_lists(View vw, const uint64_t *C, const uint32_t *__restrict__ nrank, uint32_t nlist_, uint32_t L, uint64_t m,
                                  const uint64_t *list_nos, const uint64_t *out_off, uint64_t total, uint64_t *out) {
    __shared__ uint64_t s_binom[View::kRrrView ? 64 * 64 : 1];
    __shared__ uint8_t s_ow[64];
    wt_stage_tables(vw, s_binom, s_ow);
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
            uint64_t s_hi = sh >= 32 ? nlist_ : ((p + 1) << sh);
            if (s_hi > nlist_) s_hi = nlist_;
            const uint64_t ne = C[s_hi];
            const bool bit = (c >> (L - 1u - (uint32_t)level)) & 1u;
            auto bv = vw.level((uint32_t)level);
            wt_use_tables(bv, s_binom, s_ow);
            const uint64_t r_ns = nrank[wt_nrank_base((uint32_t)level) + p];
            pos = bv.select_in((bit ? r_ns : ns - r_ns) + pos, bit, ns, ne) - ns;
        }
        out[g] = pos;
    }
}

// get_ids for every list (custom_invlists_impl.cpp:381-392 loops get_single_id)
template <class View>
__global__ void k_wt_decode_all(View vw, const uint64_t *C, const uint32_t *__restrict__ nrank, uint32_t nlist, uint32_t L, uint64_t ntotal,
                                uint64_t *out) {
    __shared__ uint64_t s_binom[View::kRrrView ? 64 * 64 : 1];
    __shared__ uint8_t s_ow[64];
    wt_stage_tables(vw, s_binom, s_ow);
    const uint32_t nlist_ = nlist;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ntotal; g += stride) {
        const uint32_t c = find_list(C, nlist, g);
        uint64_t pos = g - C[c];
        for (int level = (int)L - 1; level >= 0; level--) {
            const uint32_t sh = L - (uint32_t)level;
            const uint64_t p = sh >= 32 ? 0 : (uint64_t)(c >> sh);
            const uint64_t ns = C[sh >= 32 ? 0 : (p << sh)];
            uint64_t s_hi = sh >= 32 ? nlist_ : ((p + 1) << sh);
            if (s_hi > nlist_) s_hi = nlist_;
            const uint64_t ne = C[s_hi];
            const bool bit = (c >> (L - 1u - (uint32_t)level)) & 1u;
            auto bv = vw.level((uint32_t)level);
            wt_use_tables(bv, s_binom, s_ow);
            const uint64_t r_ns = nrank[wt_nrank_base((uint32_t)level) + p];
            pos = bv.select_in((bit ? r_ns : ns - r_ns) + pos, bit, ns, ne) - ns;
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
template <bool LAST, class Bv>
__global__ void k_wt_decode_level(const WtItem *in, WtItem *out, uint64_t *out_ids, Bv bv, const uint64_t *C,
                                  const uint32_t *__restrict__ nrank_level, uint64_t ntotal, uint32_t nlist, uint32_t L, uint32_t level) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint32_t sh = L - level;  // symbols of one node share their top `level` bits
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ntotal; i += stride) {
        const WtItem it = in ? in[i] : WtItem{(uint32_t)i, 0u};
        const uint64_t p = it.pref;
        uint64_t s_lo = sh >= 32 ? 0 : (p << sh), s_hi = sh >= 32 ? nlist : ((p + 1) << sh);
        if (s_lo > nlist) s_lo = nlist;
        if (s_hi > nlist) s_hi = nlist;
        const uint64_t ns = C[s_lo], ne = C[s_hi];
        bool bit;
        // (ones before the node and before the next one: from the build's table -- one rank per element instead of three)
        const uint64_t r_ns = nrank_level[p], r_ne = nrank_level[p + 1], r_i = bv.rank_1_bit(i, bit);
        const uint64_t zeros_in_node = (ne - ns) - (r_ne - r_ns);
        const uint64_t dst = bit ? ns + zeros_in_node + (r_i - r_ns) : ns + ((i - ns) - (r_i - r_ns));
        if (LAST) out_ids[dst] = it.id;
        else out[dst] = WtItem{it.id, (uint32_t)((p << 1) | (bit ? 1u : 0u))};
    }
}

// The same pass over tiles of 8192 consecutive positions (round 6): the tile's 128 bit-vector words and 16 rank-directory entries give
// every position its bit and rank (v_mbcnt inside the word) -- the per-element form above asked the rank directory and up to eight words
// per element --, the destination is one table entry (d_dstab) plus that rank, and all of a thread's 32 items are requested before any
// of them is used.
#define VIDC_WTD_EPT 32u
template <bool LAST>
__global__ void __launch_bounds__(256) k_wt_decode_tile(const WtItem *__restrict__ in, WtItem *__restrict__ out, uint64_t *__restrict__ out_ids,
                                                        const uint64_t *__restrict__ bits, uint64_t nwords, const uint32_t *__restrict__ rank,
                                                        const uint32_t *__restrict__ dstab_l, uint64_t ntotal) {
    __shared__ uint64_t words[VIDC_WTD_EPT * 4];
    __shared__ uint32_t wpre[VIDC_WTD_EPT * 4];  // ones before every word of the tile, from the start of the level
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * (256u * VIDC_WTD_EPT);
    WtItem it[VIDC_WTD_EPT];
#pragma unroll
    for (int j = 0; j < (int)VIDC_WTD_EPT; j++) {
        const uint64_t i = base + (uint64_t)j * 256u + t;
        if (in) it[j] = in[i < ntotal ? i : ntotal - 1u];
        else it[j] = WtItem{(uint32_t)i, 0u};  // level 0: position i holds id i, prefix 0
    }
    if (t < VIDC_WTD_EPT * 4) {
        const uint64_t w = (uint64_t)blockIdx.x * (VIDC_WTD_EPT * 4) + t;
        words[t] = w < nwords ? bits[w] : 0ull;
    }
    __syncthreads();
    if (t < VIDC_WTD_EPT * 4) {
        const uint64_t blk = (uint64_t)blockIdx.x * (VIDC_WTD_EPT / 2) + (t >> 3);
        uint32_t c = blk * 8u < nwords ? rank[blk] : 0u;
        for (uint32_t k = t & ~7u; k < t; k++) c += (uint32_t)__builtin_popcountll(words[k]);
        wpre[t] = c;
    }
    __syncthreads();
#pragma unroll
    for (int j0 = 0; j0 < (int)VIDC_WTD_EPT; j0 += 16) {
        uint32_t tb[16];
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int j = j0 + jj;
            const uint64_t wd = words[j * 4 + wave];
            const uint32_t bit = (uint32_t)(wd >> lane) & 1u;
            tb[jj] = dstab_l[2u * it[j].pref + bit];
        }
#pragma unroll
        for (int jj = 0; jj < 16; jj++) {
            const int j = j0 + jj;
            const uint64_t i = base + (uint64_t)j * 256u + t;
            const uint64_t wd = words[j * 4 + wave];
            const uint32_t bit = (uint32_t)(wd >> lane) & 1u;
            const uint64_t r_i = (uint64_t)wpre[j * 4 + wave] +
                                 __builtin_amdgcn_mbcnt_hi((uint32_t)(wd >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)wd, 0u));
            const uint64_t dst = (uint64_t)tb[jj] + (bit ? r_i : i - r_i);
            if (i < ntotal) {
                if (LAST) out_ids[dst] = it[j].id;
                else out[dst] = WtItem{it[j].id, (it[j].pref << 1) | bit};
            }
        }
    }
}

WtPlainView plain_view(const vidc_wt *w) { return WtPlainView{w->d_bits.p, w->d_rank.p, w->words_per_level, w->blocks_per_level}; }
RrrTab rrr_tab() {
    RrrTab t;
    for (int c = 0; c <= 63; c++) {  // ceil(log2(binom(63, c)))
        long double lg = 0;
        for (int i = 1; i <= c; i++) lg += log2l((long double)(63 - c + i)) - log2l((long double)i);
        t.ow[c] = (uint8_t)ceill(lg - 1e-12L);
    }
    return t;
}
BvRrr rrr_view_level(const vidc_wt *w, uint32_t l) {
    return BvRrr{w->d_cls.p + (uint64_t)l * w->rrr_cls_wpl, w->d_offs.p + w->rrr_off_base[l], w->d_ptr.p + (uint64_t)l * (w->rrr_nsamp + 1),
                 w->d_rs.p + (uint64_t)l * (w->rrr_nsamp + 1), w->d_binom.p, (const uint8_t *)(w->d_binom.p + 64 * 64), w->rrr_nblk,
                 w->rrr_nsamp, w->ntotal};
}
WtRrrView rrr_view(const vidc_wt *w) {
    WtRrrView v{};
    v.cls = w->d_cls.p; v.offs = w->d_offs.p; v.ptr = w->d_ptr.p; v.rs = w->d_rs.p; v.binom = w->d_binom.p;
    v.cls_wpl = w->rrr_cls_wpl; v.nblk = w->rrr_nblk; v.nsamp = w->rrr_nsamp; v.nbits = w->ntotal;
    for (uint32_t l = 0; l < w->L && l < 32; l++) v.off_base[l] = w->rrr_off_base[l];
    return v;
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
    // wt_type 1 keeps one level of plain bits at a time (scratch) and the RRR coding of every level
    const bool rrr = wt_type == 1;
    if (rrr && L > 32) { set_error("wavelet tree: more than 32 levels"); return VIDC_ERR_UNSUPPORTED; }
    Scratch s_a, s_b, s_err, s_lvl_bits, s_lvl_rank, s_width, s_bitpos, s_scan, s_offs_tmp;
    const uint64_t nblk = (nt + RRR_B - 1) / RRR_B, nsamp = (nblk + RRR_K - 1) / RRR_K;
    const RrrTab tab = rrr_tab();
    if (!rrr) {
        VIDC_TRY(w->d_bits.alloc(L * w->words_per_level));
        VIDC_TRY(w->d_rank.alloc(L * (w->blocks_per_level + 1)));
        VIDC_HIP(hipMemsetAsync(w->d_bits.p, 0, L * w->words_per_level * 8, ctx->stream));
    } else {
        w->rrr_nblk = nblk; w->rrr_nsamp = nsamp;
        w->rrr_cls_wpl = 6 * nsamp + 2;  // 192 bits per sample of 32 blocks (whole samples: select_in reads a sample's six words), even: 8-byte loads
        VIDC_TRY(w->d_cls.alloc(L * w->rrr_cls_wpl));
        VIDC_TRY(w->d_ptr.alloc(L * (nsamp + 1)));
        VIDC_TRY(w->d_rs.alloc(L * (nsamp + 1)));
        VIDC_HIP(hipMemsetAsync(w->d_cls.p, 0, L * w->rrr_cls_wpl * 4, ctx->stream));
        VIDC_TRY(s_lvl_bits.get(ctx, w->words_per_level * 8));
        VIDC_TRY(s_lvl_rank.get(ctx, (w->blocks_per_level + 1) * 4));
        VIDC_TRY(s_width.get(ctx, (nblk + 1) * 4));
        VIDC_TRY(s_bitpos.get(ctx, (nblk + 1) * 8));
        // worst case of a level's offset stream: 61 bits per block (transient; the object keeps the exact size)
        VIDC_TRY(s_offs_tmp.get(ctx, (size_t)L * (w->words_per_level + 1) * 8));
        VIDC_HIP(hipMemsetAsync(s_offs_tmp.p, 0, (size_t)L * (w->words_per_level + 1) * 8, ctx->stream));
        std::vector<uint64_t> bn(64 * 64, 0);
        for (int n = 0; n < 64; n++) {
            bn[n * 64] = 1;
            for (int k = 1; k <= n; k++) bn[n * 64 + k] = (n ? bn[(n - 1) * 64 + k - 1] : 0) + (k <= n - 1 ? bn[(n - 1) * 64 + k] : 0);
        }
        VIDC_TRY(w->d_binom.alloc(64 * 64 + 8));
        VIDC_HIP(hipMemcpyAsync(w->d_binom.p, bn.data(), 64 * 64 * 8, hipMemcpyHostToDevice, ctx->stream));
        VIDC_HIP(hipMemcpyAsync(w->d_binom.p + 64 * 64, tab.ow, 64, hipMemcpyHostToDevice, ctx->stream));
        VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));  // (bn leaves scope)
        w->rrr_off_base.assign(L + 1, 0);
    }
    VIDC_TRY(w->d_nrank.alloc(wt_nrank_base(L) + 1));
    VIDC_HIP(hipMemsetAsync(w->d_nrank.p, 0, (wt_nrank_base(L) + 1) * 4, ctx->stream));
    VIDC_TRY(s_a.get(ctx, (nt ? nt : 1) * 4));
    VIDC_TRY(s_b.get(ctx, (nt ? nt : 1) * 4));
    VIDC_TRY(s_err.get(ctx, 4));
    VIDC_HIP(hipMemsetAsync(s_err.p, 0, 4, ctx->stream));
    // list_nos[id]: partition + place (two streaming passes) for the sizes it is built for, the direct scatter otherwise
    const bool part2 = nt >= (1ull << 18) && nt <= (1ull << 24) && nlist <= (1ull << 18) && !std::getenv("VIDC_WT_SCATTER");
    Scratch s_part;
    if (!part2) VIDC_HIP(hipMemsetAsync(s_a.p, 0xff, (nt ? nt : 1) * 4, ctx->stream));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((nt + 255) / 256 + 1, (uint64_t)ctx->num_cu * 32);
    if (part2) VIDC_TRY(s_part.get(ctx, ((size_t)VIDC_WTP_NBLK * VIDC_WTP_MAXB + 2 * VIDC_WTP_MAXB + 8) * 4));
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    if (part2) {
        uint32_t *part = s_part.as<uint32_t>(), *tot = part + (size_t)VIDC_WTP_NBLK * VIDC_WTP_MAXB, *bstart = tot + VIDC_WTP_MAXB;
        uint32_t *pairs = s_b.as<uint32_t>();  // (the second symbol array is free until the first level has run)
        const uint32_t nbuckets = (uint32_t)((nt + VIDC_WTP_BUCKET - 1) >> VIDC_WTP_BSH);
        hipLaunchKernelGGL(k_wt_part_hist, dim3(VIDC_WTP_NBLK), dim3(256), 0, ctx->stream, d_ids, nt, part);
        hipLaunchKernelGGL(k_wt_part_scan, dim3(VIDC_WTP_MAXB / 64), dim3(64), 0, ctx->stream, part, tot);
        hipLaunchKernelGGL(k_wt_part_scatter, dim3(VIDC_WTP_NBLK), dim3(256), 0, ctx->stream, d_ids, w->d_C.p, (uint32_t)nlist, nt, part, tot,
                           bstart, pairs, s_err.as<uint32_t>());
        hipLaunchKernelGGL(k_wt_part_place, dim3(nbuckets), dim3(256), VIDC_WTP_BUCKET * 4, ctx->stream, pairs, bstart, nt, s_a.as<uint32_t>(),
                           s_err.as<uint32_t>());
        VIDC_HIP(hipGetLastError());
    } else if (nt) {
        hipLaunchKernelGGL(k_wt_scatter_syms, dim3((uint32_t)std::min<uint64_t>((nt + 4095) / 4096, (uint64_t)ctx->num_cu * 32)), dim3(256), 0,
                           ctx->stream, d_ids, w->d_C.p, (uint32_t)nlist, nt, s_a.as<uint32_t>(), s_err.as<uint32_t>());
        hipLaunchKernelGGL(k_wt_check_syms, dim3(grid), dim3(256), 0, ctx->stream, s_a.as<uint32_t>(), nt, s_err.as<uint32_t>());
        VIDC_HIP(hipGetLastError());
    }
    // scan states of every level's pass (zeroed once) and the ones before every node start of every level (from C alone): queued in front
    // of the wait for the verdict on the ids, which they do not depend on
    Scratch s_state;
    {
        VIDC_TRY(w->d_dstab.alloc(2 * (wt_nrank_base(L) + 1)));
        const size_t per_level = (size_t)((w->words_per_level + VIDC_WT_EPT * 4 - 1) / (VIDC_WT_EPT * 4)) + 1;
        VIDC_TRY(s_state.get(ctx, (size_t)L * per_level * 8));
        VIDC_HIP(hipMemsetAsync(s_state.p, 0, (size_t)L * per_level * 8, ctx->stream));
        hipLaunchKernelGGL(k_wt_node_ranks_from_C, dim3(L * VIDC_WT_NR_G), dim3(256), 0, ctx->stream, w->d_C.p, (uint32_t)nlist, L, w->d_nrank.p,
                           w->d_dstab.p);
        VIDC_HIP(hipGetLastError());
    }
    uint32_t err = 0;
    VIDC_HIP(hipMemcpyAsync(&err, s_err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    if (err) {
        set_error("wavelet tree: ids must be a permutation of 0..ntotal-1, ascending inside every list "
                  "(asserts at custom_invlists_impl.cpp:359-360)");
        return VIDC_ERR_DOMAIN;
    }
    uint32_t *cur = s_a.as<uint32_t>(), *nxt = s_b.as<uint32_t>();
    std::vector<uint64_t> lvl_bits(L, 0);  // wt_type 1: bits of every level's offset stream
    for (uint32_t level = 0; level < L && nt; level++) {
        uint64_t *bits = rrr ? s_lvl_bits.as<uint64_t>() : w->d_bits.p + (uint64_t)level * w->words_per_level;
        uint32_t *rank = rrr ? s_lvl_rank.as<uint32_t>() : w->d_rank.p + (uint64_t)level * (w->blocks_per_level + 1);
        if (rrr) VIDC_HIP(hipMemsetAsync(bits, 0, w->words_per_level * 8, ctx->stream));
        const uint32_t ntiles = (uint32_t)((w->words_per_level + VIDC_WT_EPT * 4 - 1) / (VIDC_WT_EPT * 4));
        unsigned long long *st = s_state.as<unsigned long long>() + (size_t)level * (ntiles + 1);
        hipLaunchKernelGGL(k_wt_level, dim3(ntiles), dim3(256), 0, ctx->stream, cur, level + 1 < L ? nxt : (uint32_t *)nullptr, nt,
                           (uint32_t)nlist, L, level, w->d_dstab.p + 2 * wt_nrank_base(level), bits, w->words_per_level, rank,
                           w->blocks_per_level, st, ntiles);
        if (level + 1 < L) std::swap(cur, nxt);
        VIDC_HIP(hipGetLastError());
        if (rrr) {  // code this level: classes -> offset widths -> bit positions (scan) -> offsets, samples
            const uint32_t bgrid = (uint32_t)std::min<uint64_t>((nblk + 255) / 256 + 1, (uint64_t)ctx->num_cu * 32);
            uint64_t *otmp = s_offs_tmp.as<uint64_t>() + (size_t)level * (w->words_per_level + 1);
            hipLaunchKernelGGL(k_rrr_classes, dim3(bgrid), dim3(256), 0, ctx->stream, bits, nblk, tab,
                               w->d_cls.p + (uint64_t)level * w->rrr_cls_wpl, s_width.as<uint32_t>());
            VIDC_TRY(device_exscan(ctx, s_width.as<uint32_t>(), (uint32_t)nblk, s_bitpos.as<uint64_t>(), s_scan));
            hipLaunchKernelGGL(k_rrr_offsets, dim3(bgrid), dim3(256), 0, ctx->stream, bits, nblk, tab, w->d_binom.p,
                               s_bitpos.as<uint64_t>(), (unsigned long long *)otmp);
            hipLaunchKernelGGL(k_rrr_samples, dim3((uint32_t)std::min<uint64_t>(nsamp / 256 + 1, 4096)), dim3(256), 0, ctx->stream,
                               bits, rank, nt, nblk, nsamp, s_bitpos.as<uint64_t>(), w->d_ptr.p + (uint64_t)level * (nsamp + 1),
                               w->d_rs.p + (uint64_t)level * (nsamp + 1));
            VIDC_HIP(hipGetLastError());
            VIDC_HIP(hipMemcpyAsync(&lvl_bits[level], s_bitpos.as<uint64_t>() + nblk, 8, hipMemcpyDeviceToHost, ctx->stream));
            VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));  // (the scratch of the level is reused by the next one)
        }
    }
    if (rrr) {  // the exact-size offset streams, level after level (+ one pad word each: two-word reads)
        for (uint32_t level = 0; level < L; level++) w->rrr_off_base[level + 1] = w->rrr_off_base[level] + (lvl_bits[level] + 63) / 64 + 1;
        VIDC_TRY(w->d_offs.alloc(w->rrr_off_base[L] ? w->rrr_off_base[L] : 1));
        for (uint32_t level = 0; level < L && nt; level++)
            VIDC_HIP(hipMemcpyAsync(w->d_offs.p + w->rrr_off_base[level], s_offs_tmp.as<uint64_t>() + (size_t)level * (w->words_per_level + 1),
                                    (w->rrr_off_base[level + 1] - w->rrr_off_base[level]) * 8, hipMemcpyDeviceToDevice, ctx->stream));
    }
    // size accounting = what the object holds: plain bits + rank directory, or the RRR arrays (packed classes,
    // offset streams, samples); + the symbol start table
    if (!rrr) {
        w->size_bytes = (uint64_t)L * (((nt + 63) / 64) * 8 + (w->blocks_per_level + 1) * 4) + (nlist + 1) * 8;
    } else {
        uint64_t off_bits = 0;
        for (uint32_t level = 0; level < L; level++) off_bits += lvl_bits[level];
        w->size_bytes = (off_bits + 7) / 8 + (uint64_t)L * ((6 * nblk + 7) / 8) + (uint64_t)L * (nsamp + 1) * 8 + (nlist + 1) * 8;
    }
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
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
    const dim3 sgrid((uint32_t)std::min<uint64_t>((m + 127) / 128, 1u << 16));
    if (w->wt_type == 1)
        hipLaunchKernelGGL(k_wt_select<WtRrrView>, sgrid, dim3(128), 0, ctx->stream, rrr_view(w), w->d_C.p, w->d_nrank.p, (uint32_t)w->nlist,
                           w->L, m, s_l.as<uint64_t>(), s_o.as<uint64_t>(), s_r.as<int64_t>());
    else
        hipLaunchKernelGGL(k_wt_select<WtPlainView>, sgrid, dim3(128), 0, ctx->stream, plain_view(w), w->d_C.p, w->d_nrank.p,
                           (uint32_t)w->nlist, w->L, m, s_l.as<uint64_t>(), s_o.as<uint64_t>(), s_r.as<int64_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipMemcpyAsync(ids_out, s_r.p, m * 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    ctx->d2h_bytes += m * 8;
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
    const dim3 dgrid((uint32_t)std::min<uint64_t>((total + 127) / 128, 1u << 16));
    if (w->wt_type == 1)
        hipLaunchKernelGGL(k_wt_decode_lists<WtRrrView>, dgrid, dim3(128), 0, ctx->stream, rrr_view(w), w->d_C.p, w->d_nrank.p, (uint32_t)w->nlist, w->L, m,
                           s_l.as<uint64_t>(), s_o.as<uint64_t>(), total, d_out);
    else
        hipLaunchKernelGGL(k_wt_decode_lists<WtPlainView>, dgrid, dim3(128), 0, ctx->stream, plain_view(w), w->d_C.p, w->d_nrank.p, (uint32_t)w->nlist, w->L, m,
                           s_l.as<uint64_t>(), s_o.as<uint64_t>(), total, d_out);
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    return VIDC_OK;
}

int vidc_wt_decode_gather(vidc_ctx *ctx, const vidc_wt *w, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                          const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out) {
    if (!ctx || !w) return VIDC_ERR_INVALID;
    return vidc_decode_gather_impl(ctx, w->nlist, m, list_nos, n_items, item_slot, item_off, ids_out,
                                   [&](uint64_t l) { return w->offsets[l + 1] - w->offsets[l]; },
                                   [&](uint64_t *d, uint64_t *lo) { return vidc_wt_decode_lists(ctx, w, m, list_nos, d, lo); });
}

int vidc_wt_decode_all(vidc_ctx *ctx, const vidc_wt *w, uint64_t *d_out) {
    if (!ctx || !w || (w->ntotal && !d_out)) return VIDC_ERR_INVALID;
    if (!w->ntotal) return VIDC_OK;
    VIDC_HIP(hipSetDevice(ctx->device));
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((w->ntotal + 255) / 256, (uint64_t)ctx->num_cu * 32);
    if (w->L == 0) {  // one list: ids 0..ntotal-1 in order (the per-id kernel handles it)
        if (w->wt_type == 1)
            hipLaunchKernelGGL(k_wt_decode_all<WtRrrView>, dim3(grid), dim3(128), 0, ctx->stream, rrr_view(w), w->d_C.p, w->d_nrank.p,
                               (uint32_t)w->nlist, w->L, w->ntotal, d_out);
        else
            hipLaunchKernelGGL(k_wt_decode_all<WtPlainView>, dim3(grid), dim3(128), 0, ctx->stream, plain_view(w), w->d_C.p, w->d_nrank.p,
                               (uint32_t)w->nlist, w->L, w->ntotal, d_out);
    } else {
        Scratch s_a, s_b;  // ping-pong (id, symbol prefix) arrays
        Scratch s_bits, s_rank;  // wt_type 1: one level as plain bits + rank directory
        if (w->wt_type == 1) {
            VIDC_TRY(s_bits.get(ctx, w->words_per_level * 8));
            VIDC_TRY(s_rank.get(ctx, (w->blocks_per_level + 1) * 4));
        }
        if (w->L > 1) {
            VIDC_TRY(s_a.get(ctx, w->ntotal * sizeof(WtItem)));
            if (w->L > 2) VIDC_TRY(s_b.get(ctx, w->ntotal * sizeof(WtItem)));
        }
        const WtItem *in = nullptr;
        for (uint32_t level = 0; level < w->L; level++) {
            const bool last = level + 1 == w->L;
            WtItem *out = last ? nullptr : ((level & 1u) ? s_b.as<WtItem>() : s_a.as<WtItem>());
            uint64_t *oid = last ? d_out : nullptr;
            const uint32_t *nr = w->d_nrank.p + wt_nrank_base(level);
            BvPlain bv{w->d_bits.p + (uint64_t)level * w->words_per_level,
                       w->d_rank.p + (uint64_t)level * (w->blocks_per_level + 1), w->blocks_per_level, w->words_per_level};
            if (w->wt_type == 1) {  // the level back as plain bits + rank directory (scratch), each block unranked once
                const BvRrr rv = rrr_view_level(w, level);
                VIDC_HIP(hipMemsetAsync(s_bits.p, 0, w->words_per_level * 8, ctx->stream));
                hipLaunchKernelGGL(k_rrr_expand, dim3((uint32_t)std::min<uint64_t>((w->rrr_nblk + 255) / 256, (uint64_t)ctx->num_cu * 32)),
                                   dim3(256), 0, ctx->stream, rv, (unsigned long long *)s_bits.p);
                hipLaunchKernelGGL(k_rrr_rankdir, dim3((uint32_t)std::min<uint64_t>((w->blocks_per_level + 256) / 256, 1024)), dim3(256),
                                   0, ctx->stream, rv, w->blocks_per_level, s_rank.as<uint32_t>());
                bv = BvPlain{s_bits.as<uint64_t>(), s_rank.as<uint32_t>(), w->blocks_per_level, w->words_per_level};
            }
            const dim3 tgrid((uint32_t)((w->ntotal + 256u * VIDC_WTD_EPT - 1) / (256u * VIDC_WTD_EPT)));
            const uint32_t *dt = w->d_dstab.p + 2 * wt_nrank_base(level);
            if (last) hipLaunchKernelGGL(k_wt_decode_tile<true>, tgrid, dim3(256), 0, ctx->stream, in, out, oid, bv.bits, w->words_per_level, bv.rank,
                                         dt, w->ntotal);
            else hipLaunchKernelGGL(k_wt_decode_tile<false>, tgrid, dim3(256), 0, ctx->stream, in, out, oid, bv.bits, w->words_per_level, bv.rank,
                                    dt, w->ntotal);
            (void)nr;
            in = out;
        }
        VIDC_HIP(hipGetLastError());
        VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
        VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));  // the scratch of this scope goes back to the pool below
        float ms2 = 0;
        (void)hipEventElapsedTime(&ms2, ctx->ev0, ctx->ev1);
        ctx->last_kernel_ms = ms2;
        return VIDC_OK;
    }
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    return VIDC_OK;
}

}  // extern "C"
