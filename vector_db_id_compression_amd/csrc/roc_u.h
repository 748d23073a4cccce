// roc_u.h -- "universe bitmap" ROC kernels: the latency-optimised path for ids below 2^UB (UB <= 20),
// i.e. every 1M-vector configuration of the reference's benchmarks (bench_invlists.py on sift1M /
// deep1M: ids < 10^6 < 2^20).
//
// One list per wavefront.  The order-statistic structure over the set is a three-level counting
// hierarchy over a 2^UB-bit membership bitmap:
//     level 1   64 per-lane inclusive prefix counters in ONE VGPR          (ballot + s_ff1 / v_readlane)
//     level 2   64 rows x 64 inclusive prefix counters in 64 VGPRs, row c selected with a wave-uniform
//               register index (s_set_gpr_idx / v_movrel), i.e. no memory access
//     level 3   the bitmap words in LDS (2^UB / 8 bytes: 128 KiB for UB = 20, one wave per CU)
// select(k) (encoder) and rank(x) (decoder) therefore cost ONE LDS round trip per codec step; no sort of
// the input is needed (ids index the bitmap directly) and the id itself falls out of the select.
// Duplicate ids cannot be represented: the encoder reports such a list back (it is then encoded by the
// general kernels), the decoder keeps exact multiset semantics with a side list of duplicates.
#pragma once
#include "roc_kernels.h"

namespace vidc {
namespace dev {

// Level-2 rows: 64 rows x 64 inclusive prefix counters, one counter per lane, TWO rows per VGPR (row c in the
// low/high 16 bits of register c >> 1; a counter never exceeds 2^14).  32 VGPRs, indexed with a wave-uniform
// register index (s_set_gpr_idx_on + v_mov): no memory access, no branches.
typedef uint32_t v32u __attribute__((ext_vector_type(32)));
__device__ __forceinline__ uint32_t rows_get(const v32u &r, uint32_t c) {
    return (r[c >> 1] >> ((c & 1u) << 4)) & 0xffffu;
}
// per-lane add / subtract on row c; the caller guarantees the 16-bit counter neither overflows nor borrows
__device__ __forceinline__ void rows_add(v32u &r, uint32_t c, uint32_t v16) { r[c >> 1] += v16 << ((c & 1u) << 4); }
__device__ __forceinline__ void rows_sub(v32u &r, uint32_t c, uint32_t v16) { r[c >> 1] -= v16 << ((c & 1u) << 4); }

template <int UB>
struct UGeom {
    static constexpr uint32_t NW = 1u << (UB - 6);           // 64-bit bitmap words
    static constexpr uint32_t WPL = NW / 64u;                // words per level-1 block (per lane)
    static constexpr uint32_t ENT = WPL < 64u ? WPL : 64u;   // level-2 entries per row
    static constexpr uint32_t G = WPL / ENT;                 // bitmap words per level-2 entry (1 or 4)
    static constexpr uint32_t LDS_BYTES = NW * 8u;
    static_assert(UB >= 12 && UB <= 20, "universe bits");
    static_assert(G == 1 || G == 4, "entry width");
};

// rows[c] lane t = number of set bits in entries 0..t of level-1 block c; P1 lane c = set bits in blocks 0..c
template <int UB>
__device__ __forceinline__ void u_build_counts(const uint64_t *bm, v32u &rows, uint32_t &P1, uint32_t &P1x) {
    using U = UGeom<UB>;
    const uint32_t lane = lane_id();
    uint32_t running = 0;
    P1 = 0;
    P1x = 0;
#pragma unroll
    for (int c = 0; c < 32; c++) rows[c] = 0u;
    for (uint32_t c = 0; c < 64; c++) {
        uint32_t cnt = 0;
        if (lane < U::ENT) {
#pragma unroll
            for (uint32_t g = 0; g < U::G; g++) cnt += popc64(bm[c * U::WPL + lane * U::G + g]);
        }
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t v = (uint32_t)__shfl_up((int)cnt, o, 64);
            if (lane >= (uint32_t)o) cnt += v;
        }
        rows_add(rows, c, lane < U::ENT ? cnt : 0xffffu);  // padding lanes never win a "> k" vote before a real one
        P1x = lane == c ? running : P1x;
        running += rl(cnt, 63);
        P1 = lane == c ? running : P1;
    }
}

// -------------------------------------------------------------------------------------------------
#ifdef VIDC_PROF
#define VIDC_TICK(i) { uint64_t _t = __builtin_readcyclecounter(); prof[i] += _t - tprev; tprev = _t; }
#else
#define VIDC_TICK(i)
#endif

template <int UB, bool WANT_ORDER>
__global__ void __launch_bounds__(64) k_roc_encode_u(RocEncArgs a) {
    using U = UGeom<UB>;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *bm = (uint64_t *)smem;
    uint32_t *bm32 = (uint32_t *)smem;
    const uint32_t lane = lane_id();
    const uint32_t wi = blockIdx.x;
    if (wi >= a.nwork) return;
    // per-list scalars are read with vector loads: pin them to SGPRs so the whole ANS chain stays scalar
    const uint32_t l = rfl(a.worklist[wi]);
    const uint64_t off = rfl64(a.offsets[l]);
    const uint32_t n = rfl((uint32_t)(a.offsets[l + 1] - off));
    // ---- membership bitmap
    {
        uint4 *z = (uint4 *)smem;
        for (uint32_t w = lane; w < U::LDS_BYTES / 16u; w += 64) z[w] = make_uint4(0, 0, 0, 0);
    }
    wave_sync();
    bool dup = false;
    for (uint32_t j = lane; j < n; j += 64) {
        const uint32_t x = (uint32_t)a.ids[off + j];  // < 2^UB: guaranteed by the host's classification
        const uint32_t bit = 1u << (x & 31u);
        const uint32_t old = atomicOr(&bm32[x >> 5], bit);
        dup |= (old & bit) != 0;
    }
    wave_sync();
    if (ballot(dup)) {  // multiset input: needs the sorted-position kernels
        if (lane == 0) a.status[l] = VIDC_ST_PENDING_SORT;
        return;
    }
    v32u rows;
    uint32_t P1, P1x;  // lane c: set bits in blocks 0..c (inclusive) / 0..c-1 (exclusive)
    u_build_counts<UB>(bm, rows, P1, P1x);

    const uint32_t P = rfl(a.prec[l]);  // written by the prepass
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    WStack st;
    {
        const uint64_t ao = rfl64(arena_at(a, l));
        ws_init_empty(st, a.arena + ao, rfl((uint32_t)(arena_at(a, l + 1) - ao)), a.mt, VIDC_MT_TABLE);
    }
    uint64_t head = VIDC_RANS_L;
    Recip rc;
    uint32_t obuf = 0;
    uint32_t *order = WANT_ORDER ? a.perm + off : nullptr;  // sampled ids; k_perm_from_order turns them into positions

#ifdef VIDC_PROF
    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = __builtin_readcyclecounter();
#endif
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        recip_block(rc, n - i0);  // lane t owns the divisor of step i0 + t
        const uint32_t steps = n - i0 < 64u ? n - i0 : 64u;
        VIDC_TICK(0)
        for (uint32_t t64 = 0; t64 < steps; t64++) {
            ws_prepare(st);
            VIDC_TICK(1)
            uint32_t k = ans_idx_pop_v(head, st, n - i0 - t64, rl(rc.thr, t64) - 1u, rc.d, rc.m_lo, rc.m_hi, t64);
            VIDC_TICK(2)
            // k is a uniform VGPR value: the select arithmetic stays on the VALU (see to_v())
            // level 1
            const uint32_t c = ff1(ballot(P1 > k));
            k -= rl(P1x, c);
            // level 2 (register row)
            const uint32_t row = rows_get(rows, c);
            const uint32_t t = ff1(ballot(row > k));
            k -= rl(lane_shr1(row), t);
            uint32_t wb = c * U::WPL + t * U::G;
            VIDC_TICK(3)
            // level 3 (LDS words)
            uint32_t w_lo, w_hi;
            if (U::G == 1) {
                const uint64_t wv = bm[wb];
                w_lo = rfl((uint32_t)wv);
                w_hi = rfl((uint32_t)(wv >> 32));
            } else {
                const uint64_t wv = bm[wb + (lane & 3u)];
                // counter updates that only need (c, t) run in the shadow of the LDS round trip (-1.2 % per step)
                __builtin_amdgcn_sched_barrier(0);
                P1 -= lane >= c ? 1u : 0u;
                P1x -= lane > c ? 1u : 0u;
                rows_sub(rows, c, (lane >= t && lane < U::ENT) ? 1u : 0u);  // those counters include x: no borrow
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t pc = popc64(wv);
                const uint32_t incl = prefix4(pc);                       // lanes 0..3
                const uint32_t g = ff1((ballot(incl > k) & 0xfull) | 8ull);  // first of the 4 words holding bit k
                k -= rl(incl - pc, g);
                w_lo = rl((uint32_t)wv, g);
                w_hi = rl((uint32_t)(wv >> 32), g);
                wb += g;
            }
            const uint64_t W = ((uint64_t)w_hi << 32) | w_lo;
            const uint32_t b = ff1(ballot(mbcnt(W) == k) & W);
            const uint32_t x = (wb << 6) | b;
            VIDC_TICK(4)
            // remove x (every lane stores the same word: no exec-mask juggling)
            if (U::G == 1) {
                P1 -= lane >= c ? 1u : 0u;
                P1x -= lane > c ? 1u : 0u;
                rows_sub(rows, c, (lane >= t && lane < U::ENT) ? 1u : 0u);  // those counters include x: no borrow
            }
            bm[wb] = W & ~(1ull << b);
            VIDC_TICK(5)
            ans_id_push(head, st, x, p0, p1);
            if (WANT_ORDER) obuf = wl(x, t64, obuf);
            VIDC_TICK(6)
        }
        if (WANT_ORDER) {
            if (lane < steps) order[i0 + lane] = obuf;
        }
    }
    ws_flush(st);
#ifdef VIDC_PROF
    if (lane == 0)
        for (int q = 0; q < 8; q++) ((uint64_t *)a.sid)[q] = prof[q];
#endif
    if (lane == 0) {
        a.heads[l] = head;
        a.nwords[l] = st.sp;
        a.draws[l] = st.draws;
        a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
    }
}

// -------------------------------------------------------------------------------------------------
// (a device function: the round-2 kernel of roc_u2.h falls back to it for streams that decode to a multiset)
template <int UB>
__device__ __forceinline__ void roc_decode_u_body(const RocDecArgs &a, unsigned char *smem) {
    using U = UGeom<UB>;
    constexpr uint32_t GSH = U::G == 4 ? 2u : 0u;
    constexpr uint32_t ESH = 6u + GSH;                       // bits of x inside one level-2 entry
    constexpr uint32_t ENT_SH = U::ENT == 64u ? 6u : (U::ENT == 16u ? 4u : (U::ENT == 32u ? 5u : 0u));
    static_assert((1u << ENT_SH) == U::ENT, "entries per row must be 16, 32 or 64");
    uint64_t *bm = (uint64_t *)smem;
    const uint32_t lane = lane_id();
    const uint32_t wi = blockIdx.x;
    if (wi >= a.nwork) return;
    const uint32_t l = rfl(a.worklist[wi]);
    const uint32_t n = rfl((uint32_t)(a.offsets[l + 1] - a.offsets[l]));
    const uint64_t ooff = rfl64(a.out_off ? a.out_off[wi] : a.offsets[l]);
    {
        uint4 *z = (uint4 *)smem;
        for (uint32_t w = lane; w < U::LDS_BYTES / 16u; w += 64) z[w] = make_uint4(0, 0, 0, 0);
    }
    const uint32_t P = rfl(a.prec[l]);
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    const uint32_t W0 = rfl(a.nwords[l]);
    WStack st;
    ws_init_loaded(st, a.words + rfl64(a.word_off[l]), W0, a.scratch_words + rfl64(a.scratch_off[wi]), roc_dec_stack_cap(n, W0),
                   rfl(a.draws[l]), a.mt, VIDC_MT_TABLE);
    const uint32_t draws0 = st.draws;
    uint64_t head = rfl64(a.heads[l]);
    v32u rows;
#pragma unroll
    for (int c = 0; c < 32; c++) rows[c] = 0u;
    uint32_t P1x = 0;  // lane c: decoded elements in blocks 0..c-1 (exclusive prefix)
    uint32_t *dups = a.slots + rfl64(a.slots_off[wi]);  // capacity n
    uint32_t ndup = 0;
    uint32_t ring = 0, lq = 0;
    wave_sync();

    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        lq = 0x80000000u / (i0 + 1u + lane);  // lane t owns floor(2^31 / nmax) of step i0 + t
        const uint32_t steps = n - i0 < 64u ? n - i0 : 64u;
        for (uint32_t t64 = 0; t64 < steps; t64++) {
            ws_prepare(st);
            const uint32_t x = ans_id_pop(head, st, p0, p1);
            const uint32_t e = x >> ESH;
            const uint32_t c = e >> ENT_SH, t = e & (U::ENT - 1u);
            const uint32_t bit = x & 63u;
            const uint32_t row = rows_get(rows, c);
            // the partial ranks are summed on the VALU (uniform VGPR value, see to_v())
            uint32_t r = to_v(rl(P1x, c)) + rl(lane_shr1(row), t);
            const uint32_t wb = x >> 6;
            uint64_t W;
            if (U::G == 1) {
                W = rfl64(bm[wb]);
                r += popc64(W & ((1ull << bit) - 1ull));
            } else {
                const uint32_t wsel = wb & 3u;
                const uint32_t l4 = lane & 3u;
                const uint64_t wv = bm[(wb & ~3u) + l4];
                const uint64_t below = l4 < wsel ? ~0ull : (l4 == wsel ? ((1ull << bit) - 1ull) : 0ull);
                const uint32_t pc = prefix4(popc64(wv & below));  // lane 3: bits below x in the 4 words
                r += rl(pc, 3);
                W = rl64((uint32_t)wv, (uint32_t)(wv >> 32), wsel);
            }
            const bool isdup = (W >> bit) & 1ull;
            if (__builtin_expect(ndup != 0u, 0)) {  // multiset streams only (reference quirk cases)
                for (uint32_t j0 = 0; j0 < ndup; j0 += 64) {
                    const uint32_t jj = j0 + lane;
                    const uint32_t z = jj < ndup ? dups[jj] : 0xffffffffu;
                    r += popc64(ballot(jj < ndup && (z >> ESH) == e && z < x));
                }
            }
            ans_idx_push(head, st, rfl(r), i0 + t64 + 1u, rl(lq, t64));
            // insert x
            P1x += lane > c ? 1u : 0u;
            rows_add(rows, c, (lane >= t && lane < U::ENT) ? 1u : 0u);
            if (__builtin_expect(isdup, 0)) {
                if (lane == 0) dups[ndup] = x;
                ndup++;
                wave_sync();
            } else {
                bm[wb] = W | (1ull << bit);  // every lane stores the same word
            }
            ring = wl(x, t64, ring);
        }
        if (lane < steps) a.out[ooff + (n - 1u - (i0 + lane))] = (uint64_t)ring;
    }
    if (lane == 0) {
        const bool clean = (head == VIDC_RANS_L) && (st.sp == st.draws - draws0);
        a.end_state[l] = clean ? 0u : 1u;
        a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
    }
}

template <int UB>
__global__ void __launch_bounds__(64) k_roc_decode_u(RocDecArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    roc_decode_u_body<UB>(a, smem);
}

// -------------------------------------------------------------------------------------------------
// classification prepass: one wavefront per list -> max id, precision, flags
#define VIDC_PF_UNSORTED 1u  // not strictly ascending (unsorted or duplicates)
#define VIDC_PF_DOMAIN 2u    // an id >= 2^31
__global__ void __launch_bounds__(256) k_roc_prepass(const uint64_t *ids, const uint64_t *offsets, uint32_t nlist,
                                                     int precision_mode, uint32_t *maxid, uint32_t *flags,
                                                     uint32_t *prec) {
    __shared__ uint32_t s_max[4], s_flag[4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t l = blockIdx.x; l < nlist; l += gridDim.x) {
        const uint64_t off = offsets[l];
        const uint32_t n = (uint32_t)(offsets[l + 1] - off);
        uint32_t mx = 0;
        bool bad = false, uns = false;
        for (uint32_t j = threadIdx.x; j < n; j += 256) {
            const uint64_t id = ids[off + j];
            bad |= id >= (1ull << 31);
            if (j) uns |= ids[off + j - 1] >= id;
            const uint32_t v = (uint32_t)id;
            mx = v > mx ? v : mx;
        }
        const uint32_t m = wave_max_u32(mx);
        const uint32_t f = (ballot(uns) ? VIDC_PF_UNSORTED : 0u) | (ballot(bad) ? VIDC_PF_DOMAIN : 0u);
        if (lane == 0) { s_max[wave] = m; s_flag[wave] = f; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t mm = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));
            maxid[l] = mm;
            flags[l] = s_flag[0] | s_flag[1] | s_flag[2] | s_flag[3];
            prec[l] = n ? precision_for(mm, precision_mode) : 0u;
        }
        __syncthreads();
    }
}

// the same from the last id of each list alone (one thread per list): exact for ascending lists; the encode kernels
// verify what this one assumes (roc.hip, light_prepass)
__global__ void __launch_bounds__(256) k_roc_prepass_last(const uint64_t *ids, const uint64_t *offsets, uint32_t nlist,
                                                          int precision_mode, uint32_t *maxid, uint32_t *flags, uint32_t *prec) {
    const uint32_t l = blockIdx.x * 256u + threadIdx.x;
    if (l >= nlist) return;
    const uint64_t off = offsets[l], n = offsets[l + 1] - off;
    const uint64_t id = n ? ids[off + n - 1] : 0ull;
    maxid[l] = (uint32_t)id;
    flags[l] = id >= (1ull << 31) ? VIDC_PF_DOMAIN : 0u;
    prec[l] = n ? precision_for((uint32_t)id, precision_mode) : 0u;
}

// order[] (sampled ids, written by the U encoder into the perm buffer) -> input positions, for lists whose
// input is strictly ascending: position = index of the id in the list (binary search).
// One workgroup per (list, chunk of VIDC_PERM_CHUNK ids), with all chunks of a list on ONE XCD: workgroups go to the 8 XCDs
// round-robin by index, so item k of "lane" x sits at workgroup 8 k + x and the host fills each lane with whole lists
// (roc.hip).  A workgroup per LIST kept one CU busy for 0.53 ms behind the S1 chain (2 % of the step); chunks spread over
// all XCDs took 0.03 ms but every XCD's L2 fetched the whole list again (7.7 instead of 1.8 MiB per step).
#define VIDC_PERM_CHUNK 2048u
__global__ void k_perm_from_order(const uint64_t *ids, const uint64_t *offsets, const uint2 *items, uint32_t nitems,
                                  uint32_t *perm) {
    for (uint32_t wi = blockIdx.x; wi < nitems; wi += gridDim.x) {
        const uint32_t l = items[wi].x, start = items[wi].y;
        if (l == 0xffffffffu) continue;  // padding of a shorter lane
        const uint64_t off = offsets[l];
        const uint32_t n = (uint32_t)(offsets[l + 1] - off);
        const uint32_t end = start + VIDC_PERM_CHUNK < n ? start + VIDC_PERM_CHUNK : n;
        for (uint32_t j = start + threadIdx.x; j < end; j += blockDim.x) {
            const uint64_t x = perm[off + j];
            uint32_t lo = 0, hi = n;
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (ids[off + mid] <= x) lo = mid; else hi = mid;
            }
            perm[off + j] = lo;
        }
    }
}

}  // namespace dev
}  // namespace vidc
