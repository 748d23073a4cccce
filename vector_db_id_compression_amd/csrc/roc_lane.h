// roc_lane.h -- "one list per LANE" Random Order Coding kernels for short lists (65 .. VIDC_LANE_MAX64 ids).
//
// The wave-per-list kernels of roc_kernels.h spend ~130 wave-wide instructions per codec step on what is almost
// entirely wave-uniform arithmetic; with tens of thousands of short lists that is the bound (one step of ONE list
// per ~300 SIMD cycles).  Here every lane runs the serial chain of its own list, so one wave-wide instruction
// advances 64 lists: the ANS head, stack pointer and divisor are per-lane VGPR values, the order-statistic
// structure of a list is a private strip of LDS (interleaved by lane: any per-lane index is bank-conflict free),
// and ANS words go straight to the list's arena / come straight from its stream with per-lane addresses.
//
//   encode step (codec.cpp:131-137): k = IDX_pop(n - i) with a table of reciprocals indexed by the divisor,
//        select+remove the k-th alive position in a 3-level counted bitmap (<= 4 groups x 4 words x 64 bits),
//        x = ids[position] (the input list is strictly ascending: checked here, id by id, as positions are sampled), ID_push(x, P)
//   decode step (codec.cpp:140-152): x = ID_pop(P); rank of x among the decoded ids through bucket counters
//        (64 buckets over the top bits, two-level byte counters in LDS) + the members of x's bucket (global
//        memory, one 64/128-byte row); IDX_push(rank, i + 1)
//
// Anything outside the clean domain of these kernels (unsorted or multiset input, bucket overflow on skewed ids,
// stack underflow in the decoder) is handed back with VIDC_ST_PENDING_SORT / VIDC_ST_RETRY and re-done by the
// wave-per-list kernels, which handle every case.
#pragma once
#include "roc_kernels.h"
#include "roc_lane_reg_asm.h"
#include "rows_tile.h"

namespace vidc {
namespace dev {

#define VIDC_LANE_MAX 1024u    // NW = 16 encoder / 64-bucket decoder
#define VIDC_LANE_MAX64 4096u  // NW = 64 encoder / 256-bucket decoder (the divisor table ends here)
#define VIDC_LANE_MAX128 2048u // 128-bucket decoder (round 4): lists of 1025 .. 2048 ids, 18.5 instead of 26.5 KiB of LDS per wavefront
#define VIDC_ST_RETRY 5u  // lane decoder: redo this list with the wave-per-list kernel

// entry d of the divisor table (d = 1 .. VIDC_LANE_MAX64): x = m_lo, y = m_hi of floor((2^64-1)/d),
// z = thr = d * floor(2^31/d), w = lq = floor(2^31/d)
typedef uint4 LaneDiv;

// r-th (0-based) set bit of w; r < popcount(w)
__device__ __forceinline__ uint32_t lane_select64(uint64_t w, uint32_t r) {
    uint32_t v = (uint32_t)w;
    uint32_t c = (uint32_t)__popc(v);
    uint32_t pos = 0;
    if (r >= c) { r -= c; v = (uint32_t)(w >> 32); pos = 32; }
    c = (uint32_t)__popc(v & 0xffffu);
    if (r >= c) { r -= c; v >>= 16; pos += 16; }
    c = (uint32_t)__popc(v & 0xffu);
    if (r >= c) { r -= c; v >>= 8; pos += 8; }
    c = (uint32_t)__popc(v & 0xfu);
    if (r >= c) { r -= c; v >>= 4; pos += 4; }
    c = (uint32_t)__popc(v & 0x3u);
    if (r >= c) { r -= c; v >>= 2; pos += 2; }
    c = v & 1u;
    if (r >= c) pos += 1;
    return pos;
}

// per-lane ANS stack in global memory
struct LStack {
    uint32_t *mem;       // encoder: the list's arena.  decoder: private scratch for pushed words
    const uint32_t *orig;  // decoder: the stored stream
    uint32_t sp, cap, dirty, draws, err;
    const uint32_t *mt;
};
__device__ __forceinline__ void ls_push(LStack &s, uint32_t w) {
    if (s.sp < s.cap) s.mem[s.sp] = w; else s.err |= 1u;
    if (s.sp < s.dirty) s.dirty = s.sp;
    s.sp++;
}
__device__ __forceinline__ uint32_t ls_pop(LStack &s) {  // codec.h:32-40
    if (__builtin_expect(s.sp == 0u, 0)) {
        uint32_t w = 0;
        if (s.draws < VIDC_MT_TABLE) w = s.mt[s.draws]; else s.err |= 2u;
        s.draws++;
        return w;
    }
    s.sp--;
    return s.sp >= s.dirty ? s.mem[s.sp] : s.orig[s.sp];
}
__device__ __forceinline__ bool l_lt_2p31(uint64_t v) { return (v >> 31) == 0; }

// Encoder stack of the mid-size lane kernels: pushed words wait in a 32-deep LDS ring per lane and are stored 16 at
// a time, back to back.  Single 4-byte stores, one every step or two, left every 64-byte arena
// line partially written for so long that it went to HBM several times (WRITE_SIZE 7x the stream size).
// (round 4: a 16-deep ring with 32-byte runs -- 15.5 instead of 21.5 KiB of LDS per wavefront of the 1024-id class -- measured
// slower: 64 M ids in lists of 1024 encode in 1.95 instead of 1.69 ms, S2 55-56 instead of 54-55 ms; -DVIDC_PRING=16)
#ifndef VIDC_PRING
#define VIDC_PRING 32u
#endif
#define VIDC_PDRAIN (VIDC_PRING / 2u)  // words stored per drain: a 64-byte run
struct LEStack {
    uint32_t *mem;   // the list's arena
    uint32_t *ring;  // LDS: entry e of this lane at ring[e * 64]
    uint32_t sp, sp_mem, cap, draws, err;  // words [sp_mem, sp) are in the ring
    const uint32_t *mt;
};
__device__ __forceinline__ void ls_push(LEStack &s, uint32_t w) {
    s.ring[(s.sp & (VIDC_PRING - 1u)) * 64u] = w;
    s.sp++;
}
__device__ __forceinline__ uint32_t ls_pop(LEStack &s) {  // codec.h:32-40 (rare in the encoder)
    if (s.sp > s.sp_mem) {
        s.sp--;
        return s.ring[(s.sp & (VIDC_PRING - 1u)) * 64u];
    }
    if (s.sp == 0u) {
        uint32_t w = 0;
        if (s.draws < VIDC_MT_TABLE) w = s.mt[s.draws]; else s.err |= 2u;
        s.draws++;
        return w;
    }
    s.sp--;
    s.sp_mem = s.sp;
    return s.mem[s.sp];
}
// per lane: once half a ring of words is pending, store them as one run (a full 64-byte line, or the tails of two)
__device__ __forceinline__ void le_drain16(LEStack &s) {
    if (s.sp - s.sp_mem >= VIDC_PDRAIN) {
        if (s.sp_mem + VIDC_PDRAIN <= s.cap) {
#pragma unroll
            for (uint32_t j = 0; j < VIDC_PDRAIN; j++) s.mem[s.sp_mem + j] = s.ring[((s.sp_mem + j) & (VIDC_PRING - 1u)) * 64u];
        } else {
            s.err |= 1u;
        }
        s.sp_mem += VIDC_PDRAIN;
    }
}
// wave-uniform call: every lane stores its pending words
__device__ __forceinline__ void le_flush(LEStack &s) {
    const uint32_t cnt = s.sp - s.sp_mem;
    const uint32_t maxc = wave_max_u32(cnt);
    for (uint32_t j = 0; j < maxc; j++) {
        if (j < cnt) {
            const uint32_t idx = s.sp_mem + j;
            if (idx < s.cap) s.mem[idx] = s.ring[(idx & (VIDC_PRING - 1u)) * 64u]; else s.err |= 1u;
        }
    }
    s.sp_mem = s.sp;
}

// Decoder stack of the mid-size lane kernels: no global access of it sits on the serial chain of a list.
//   * the stored stream is consumed from its top: a 16-word window of it lives in an LDS ring per lane, refilled four
//     words (one aligned 16-byte load) at a time from a uniform point of the loop, a step or more before the words
//     are needed (a pop that outruns the window reads the stream directly);
//   * the words the decoder pushes itself (IDX_push renormalisations, codec.cpp:44-63) are popped again within a
//     step or two: they wait in an 8-deep LDS stack (ids of a list are distinct, so a step never pushes more bits
//     than it popped: the depth stays at 1-2; deeper -> VIDC_ST_RETRY, the wave-per-list kernel redoes the list).
// Before: every pop was a global load on the chain (stream word or a word pushed a step earlier), 3.9 us per step on
// 65 536 lists of 256 ids; with the bucket row as the only global access of a step see DESIGN.
#define VIDC_DWIN 16u
#define VIDC_DPST 8u
struct LWStack {
    const uint32_t *wbase;  // stream of the list minus `al` words: 16-byte aligned
    uint32_t *ring;         // LDS (lane offset included): shifted index s at ring[(s & 15) * 64]
    uint32_t *pst;          // LDS (lane offset included): pushed word e at pst[e * 64]
    uint32_t al;            // word_off & 3: shifted index = index in the list's stream + al
    uint32_t otop;          // shifted index one past the next stored word to pop (== al: none left)
    uint32_t wlo;           // lowest shifted index the window holds (multiple of 4)
    uint32_t d;             // pushed words
    uint32_t draws, err;
    const uint32_t *mt;
};
__device__ __forceinline__ void ls_push(LWStack &s, uint32_t w) {
    if (s.d < VIDC_DPST) { s.pst[s.d * 64u] = w; s.d++; } else s.err |= 4u;
}
__device__ __forceinline__ uint32_t ls_pop(LWStack &s) {  // codec.h:32-40
    if (s.d) {
        s.d--;
        return s.pst[s.d * 64u];
    }
    if (__builtin_expect(s.otop == s.al, 0)) {
        uint32_t w = 0;
        if (s.draws < VIDC_MT_TABLE) w = s.mt[s.draws]; else s.err |= 2u;
        s.draws++;
        return w;
    }
    s.otop--;
    return s.otop >= s.wlo ? s.ring[(s.otop & (VIDC_DWIN - 1u)) * 64u] : s.wbase[s.otop];
}
__device__ __forceinline__ void lw_init(LWStack &s, bool have, const uint32_t *words, uint64_t word_off, uint32_t W) {
    s.al = (uint32_t)word_off & 3u;
    s.wbase = words + (word_off - s.al);
    s.otop = W + s.al;
    s.wlo = s.otop > VIDC_DWIN ? ((s.otop - VIDC_DWIN + 3u) & ~3u) : 0u;
    s.d = 0; s.err = 0;
#pragma unroll
    for (uint32_t k = 0; k < VIDC_DWIN / 4u; k++) {
        const uint32_t q = s.wlo + 4u * k;  // (the last chunk may read up to 3 words past the stream: padded allocation)
        if (have && q < s.otop) {
            const uint4 v = *(const uint4 *)(s.wbase + q);
            uint32_t *r = s.ring + (q & (VIDC_DWIN - 1u)) * 64u;
            r[0] = v.x; r[64] = v.y; r[128] = v.z; r[192] = v.w;
        }
    }
}
// uniform call sites: issue the load of the next four words below the window once four consumed slots are free ...
__device__ __forceinline__ bool lw_issue(const LWStack &s, uint4 &pf) {
    const bool go = s.wlo >= 4u && s.otop <= s.wlo + (VIDC_DWIN - 4u);
    if (go) pf = *(const uint4 *)(s.wbase + (s.wlo - 4u));
    return go;
}
// ... and move them into the ring later in the step (behind the step's own global access: no extra wait)
__device__ __forceinline__ void lw_land(LWStack &s, bool go, const uint4 &pf) {
    if (go) {
        s.wlo -= 4u;
        uint32_t *r = s.ring + (s.wlo & (VIDC_DWIN - 1u)) * 64u;
        r[0] = pf.x; r[64] = pf.y; r[128] = pf.z; r[192] = pf.w;
    }
}

// codec.cpp:65-76
template <typename Stack>
__device__ __forceinline__ void l_u_push(uint64_t &head, Stack &s, uint32_t start, uint32_t p) {
    if ((uint32_t)(head >> 32) >= (0x80000000u >> p)) {
        ls_push(s, (uint32_t)head);
        head >>= 32;
    }
    head = (head << p) + start;
}
// codec.cpp:78-90
template <typename Stack>
__device__ __forceinline__ uint32_t l_u_pop(uint64_t &head, Stack &s, uint32_t p) {
    uint32_t sym = (uint32_t)head & ((1u << p) - 1u);
    head >>= p;
    if (l_lt_2p31(head)) head = (head << 32) | ls_pop(s);
    return sym;
}

// LDS strip of one lane: u64 bm[NW] | u32 wc[NW/4] (4 byte counters each) | u64 gc[NW/16] (4 u16 group counters each,
// NW >= 16) | u64 sc (4 u16 super-group counters, NW == 64)
// element e of lane t lives at base + (e * 64 + t) * size: conflict-free for any per-lane e
template <int NW>
struct LaneEncGeom {
    static_assert(NW == 4 || NW == 16 || NW == 32 || NW == 64, "1, 2 or 3 counter levels above the words");
    static constexpr uint32_t BM_BYTES = NW * 64 * 8;
    static constexpr uint32_t WC_BYTES = (NW / 4) * 64 * 4;
    static constexpr uint32_t GC_BYTES = NW > 4 ? (NW / 16) * 64 * 8 : 0;
    static constexpr uint32_t SC_BYTES = NW > 16 ? 64 * 8 : 0;
    static constexpr uint32_t RING_BYTES = VIDC_PRING * 64 * 4;
    static constexpr uint32_t PERM_BYTES = VIDC_PDRAIN * 64 * 4;  // sampled positions of the last <= VIDC_PDRAIN steps
    static constexpr uint32_t LDS_BYTES = BM_BYTES + WC_BYTES + GC_BYTES + SC_BYTES + RING_BYTES + PERM_BYTES;
};

// 4 u16 counters in v: the field f holding the k-th element (k < total), k reduced to its rank inside the field
__device__ __forceinline__ uint32_t lane_pick4_u16(uint64_t v, uint32_t &k) {
    const uint32_t c0 = (uint32_t)v & 0xffffu, c1 = (uint32_t)(v >> 16) & 0xffffu, c2 = (uint32_t)(v >> 32) & 0xffffu;
    const uint32_t q0 = c0, q1 = c0 + c1, q2 = q1 + c2;
    uint32_t f = 0, base = 0;
    if (k >= q0) { f = 1; base = q0; }
    if (k >= q1) { f = 2; base = q1; }
    if (k >= q2) { f = 3; base = q2; }
    k -= base;
    return f;
}

template <int NW, bool WANT_PERM>
__global__ void __launch_bounds__(64) k_roc_encode_lane(RocEncArgs a, const LaneDiv *__restrict__ dtab) {
    __shared__ __align__(16) unsigned char smem[LaneEncGeom<NW>::LDS_BYTES];
    uint64_t *bm = (uint64_t *)smem;
    uint32_t *wc = (uint32_t *)(smem + LaneEncGeom<NW>::BM_BYTES);
    uint64_t *gc = (uint64_t *)(smem + LaneEncGeom<NW>::BM_BYTES + LaneEncGeom<NW>::WC_BYTES);
    uint64_t *sc = (uint64_t *)(smem + LaneEncGeom<NW>::BM_BYTES + LaneEncGeom<NW>::WC_BYTES + LaneEncGeom<NW>::GC_BYTES);
    const uint32_t lane = lane_id();
    if (a.lpw >> 31) __builtin_amdgcn_s_setprio(3);  // (VIDC_ENC_PRIO: this class first in its SIMD's arbitration, roc.hip)
    const uint32_t lpw = (a.lpw & 0xffffu) ? (a.lpw & 0xffffu) : 64u;  // few lists in the launch: fewer per wavefront, more wavefronts (host)
    const uint32_t wi = blockIdx.x * lpw + lane;
    const bool have = lane < lpw && wi < a.nwork;
    const uint32_t l = have ? a.worklist[wi] : 0u;
    const uint64_t off = have ? a.offsets[l] : 0ull;
    const uint32_t n = have ? (uint32_t)(a.offsets[l + 1] - off) : 0u;
    const uint32_t P = have ? a.prec[l] : 0u;  // written by k_roc_prepass
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;

    // alive bitmap over the (ascending) input positions + counters
#pragma unroll
    for (int g = 0; g < NW / 4; g++) {
        uint32_t packed = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t w = g * 4 + j;
            const uint32_t lo_e = w << 6;
            const uint32_t in_w = lo_e >= n ? 0u : (n - lo_e >= 64u ? 64u : n - lo_e);
            bm[w * 64 + lane] = in_w == 64u ? ~0ull : ((1ull << in_w) - 1ull);
            packed |= in_w << (8 * j);
        }
        wc[g * 64 + lane] = packed;
    }
    if (NW > 4) {
#pragma unroll
        for (int q = 0; q < NW / 16; q++) {
            uint64_t g4 = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t lo_e = (uint32_t)(q * 4 + g) << 8;
                const uint32_t in_g = lo_e >= n ? 0u : (n - lo_e >= 256u ? 256u : n - lo_e);
                g4 |= (uint64_t)in_g << (16 * g);
            }
            gc[q * 64 + lane] = g4;
        }
    }
    if (NW > 16) {
        uint64_t s4 = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t lo_e = (uint32_t)q << 10;
            const uint32_t in_s = lo_e >= n ? 0u : (n - lo_e >= 1024u ? 1024u : n - lo_e);
            s4 |= (uint64_t)in_s << (16 * q);
        }
        sc[lane] = s4;
    }

    LEStack st;
    {
        const uint64_t ao = have ? arena_at(a, l) : 0ull;
        st.mem = a.arena + ao;
        st.ring = (uint32_t *)(smem + LaneEncGeom<NW>::BM_BYTES + LaneEncGeom<NW>::WC_BYTES + LaneEncGeom<NW>::GC_BYTES + LaneEncGeom<NW>::SC_BYTES) + lane;
        st.cap = have ? (uint32_t)(arena_at(a, l + 1) - ao) : 0u;
        st.sp = 0; st.sp_mem = 0; st.draws = 0; st.err = 0; st.mt = a.mt;
    }
    uint64_t head = VIDC_RANS_L;
    const uint64_t *ids = a.ids + off;
    const uint32_t nsteps = wave_max_u32(n);
    // the sampled positions leave 16 at a time (one 64-byte run per lane): a 4-byte store per step to a line that is
    // written back before the next one arrives cost a 32-byte write each (15 GiB of S2's encode writes)
    uint32_t *pring = (uint32_t *)(smem + LaneEncGeom<NW>::BM_BYTES + LaneEncGeom<NW>::WC_BYTES + LaneEncGeom<NW>::GC_BYTES +
                                   LaneEncGeom<NW>::SC_BYTES + LaneEncGeom<NW>::RING_BYTES) + lane;

    // the divisor entry of step i + 1 is requested during step i: a per-lane load at the head of the step cost every step a
    // round trip to the table before its first instruction (65 536 lists of 256 ids: 326 -> 297 us)
    LaneDiv dv_next = dtab[n ? n : 1u];
    bool disorder = false;
    for (uint32_t i = 0; i < nsteps; i++) {
        if (i < n) {
            // ---- k = IDX_pop(n - i), codec.cpp:21-42
            const uint32_t d = n - i;
            const LaneDiv dv = dv_next;
            dv_next = dtab[d > 1u ? d - 1u : 1u];
            uint64_t h0 = head;
            if ((uint32_t)(h0 >> 32) >= dv.z) {  // h0 >= nmax * ((L / nmax) << 32)
                ls_push(st, (uint32_t)h0);
                h0 >>= 32;
            }
            uint64_t q = __umul64hi(h0, ((uint64_t)dv.y << 32) | dv.x);
            uint32_t k = (uint32_t)h0 - (uint32_t)q * d;
            if (k >= d) { k -= d; q++; }
            if (__builtin_expect(l_lt_2p31(h0), 0)) q = (uint64_t)ls_pop(st) | (q << 32);  // test on h0 (codec.cpp:35)
            head = q;

            // ---- select + remove the k-th alive position
            uint32_t g = 0;  // group of 4 words (256 positions) holding the k-th alive position
            if (NW > 4) {
                uint32_t q = 0;
                if (NW > 16) {
                    const uint64_t sv = sc[lane];
                    q = lane_pick4_u16(sv, k);
                    sc[lane] = sv - (1ull << (16u * q));
                }
                const uint64_t gv = gc[q * 64 + lane];
                const uint32_t f = lane_pick4_u16(gv, k);
                gc[q * 64 + lane] = gv - (1ull << (16u * f));
                g = q * 4u + f;
            }
            const uint32_t wv = wc[g * 64 + lane];
            uint32_t j = 0;
            {
                const uint32_t c0 = wv & 0xffu, c1 = (wv >> 8) & 0xffu, c2 = (wv >> 16) & 0xffu;
                const uint32_t q0 = c0, q1 = c0 + c1, q2 = q1 + c2;
                uint32_t base = 0;
                if (k >= q0) { j = 1; base = q0; }
                if (k >= q1) { j = 2; base = q1; }
                if (k >= q2) { j = 3; base = q2; }
                k -= base;
                wc[g * 64 + lane] = wv - (1u << (8u * j));
            }
            const uint32_t w = g * 4u + j;
            const uint64_t word = bm[w * 64 + lane];
            const uint32_t b = lane_select64(word, k);
            bm[w * 64 + lane] = word & ~(1ull << b);
            const uint32_t pos = (w << 6) + b;
            // the sampled id and its left neighbour (the same 64-byte line seven times out of eight): every position is
            // sampled exactly once, so the list is checked to be strictly ascending -- what select over POSITIONS relies on
            // -- and to lie below 2^31; the classification prepass then only needs the last id of each list (roc.hip)
            // (one 16-byte load for both: a lane-class list has more than 64 ids, so ids[1] exists when pos is 0)
            struct __attribute__((aligned(8))) IdPair { uint64_t a, b; };
            const IdPair pr = *(const IdPair *)(ids + (pos ? pos - 1u : 0u));
            const uint64_t xid = pos ? pr.b : pr.a;
            const uint64_t xprev = pos ? pr.a : 0ull;
            disorder |= (pos && xprev >= xid) || (xid >> 31) != 0ull;
            const uint32_t x = (uint32_t)xid;
            if (WANT_PERM) pring[(i & (VIDC_PDRAIN - 1u)) * 64u] = pos;

            // ---- ID_push(x, P), codec.cpp:92-105
            l_u_push(head, st, x & 0xffffu, p0);
            l_u_push(head, st, x >> 16, p1);
            if (__builtin_expect((uint32_t)(head >> 63) != 0u, 0)) {  // slices 2, 3: precision 0, symbol 0
                l_u_push(head, st, 0u, 0u);
                l_u_push(head, st, 0u, 0u);
            }
        }
        le_drain16(st);  // at most 5 words per step: fewer than half a ring are pending afterwards, the ring never overflows
        if (WANT_PERM && ((i & (VIDC_PDRAIN - 1u)) == VIDC_PDRAIN - 1u || i + 1u == nsteps)) {  // uniform
            const uint32_t s0 = i & ~(VIDC_PDRAIN - 1u);
#pragma unroll
            for (uint32_t k = 0; k < VIDC_PDRAIN; k++)
                if (s0 + k <= i && s0 + k < n) a.perm[off + s0 + k] = pring[k * 64u];
        }
    }
    le_flush(st);
    if (have) {
        a.heads[l] = head;
        a.nwords[l] = st.sp;
        a.draws[l] = st.draws;
        // (not ascending / outside the domain: the wave-per-list kernel redoes the list and reports what is wrong with it)
        a.status[l] = disorder ? VIDC_ST_PENDING_SORT : (st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// decoder: rank of x among the ids decoded so far.  NB = 64 (lists up to 1024 ids) or 256 (up to 4096) buckets over
// the top 6 / 8 bits of the P-bit universe; LDS per lane: NB byte counters (groups of 16) + u16 group counters (4 per
// u64) + for NB = 256 4 u16 super-group counters; bucket members (u32, unsorted) in a row of `cap` slots in global
// memory, written in chunks of 4 so that every chunk below ceil(count/4) is fully defined.
template <int NB>
__host__ __device__ inline uint32_t roc_lane_cap_nb(uint32_t n) {
    // twice the mean + 16 (round 4; + 12 before: four of S2's 900 000 bucket-row lists -- 1089 .. 3291 ids -- overflowed a bucket,
    // and the pass that redoes them on the wave-per-list kernels waited for the whole decode first: 2.5 ms behind a 71 ms call)
    return (((n / (uint32_t)NB) * 2u + 16u) + 3u) & ~3u;
}
// align = 16 (round 5, RocDecArgs::row_align): rows start on 64-byte boundaries and are a multiple of 64 bytes long, so the four
// 16-byte chunks a step requests together lie in ONE 64-byte sector -- at align = 4 a row starts anywhere and they straddle two
// sectors three times out of four (S2: 1.45 sectors fetched per decoded id of the lane classes)
template <int NB>
__host__ __device__ inline uint32_t roc_lane_cap_nb(uint32_t n, uint32_t align) {
    return align > 4u ? (roc_lane_cap_nb<NB>(n) + align - 1u) & ~(align - 1u) : roc_lane_cap_nb<NB>(n);
}
__host__ __device__ inline uint32_t roc_lane_cap(uint32_t n) { return roc_lane_cap_nb<64>(n); }

// sum of the bytes j < t (t <= 8) of the 8-byte value v
__device__ __forceinline__ uint32_t lane_sum_bytes_below(uint64_t v, uint32_t t) {
    const uint64_t m = (v << 8) << (56u - 8u * (t > 7u ? 7u : t));  // bytes j < min(t, 7) moved to the top, rest dropped
    uint32_t s = __builtin_amdgcn_sad_u8((uint32_t)m, 0u, 0u);
    s = __builtin_amdgcn_sad_u8((uint32_t)(m >> 32), 0u, s);
    if (t > 7u) s += (uint32_t)(v >> 56);
    return s;
}
// sum of the u16 fields j < t (t <= 3) of v
__device__ __forceinline__ uint32_t lane_sum_u16_below(uint64_t v, uint32_t t) {
    const uint64_t m = (v << 16) << (48u - 16u * t);
    uint32_t s = __builtin_amdgcn_sad_u16((uint32_t)m, 0u, 0u);
    return __builtin_amdgcn_sad_u16((uint32_t)(m >> 32), 0u, s);
}

template <int NB>
struct LaneDecGeom {
    static_assert(NB == 64 || NB == 128 || NB == 256, "2 or 3 counter levels");
    static constexpr uint32_t NG = NB / 16;                   // groups of 16 byte counters
    static constexpr uint32_t CNT_BYTES = NG * 64 * 16;       // uint4 cnt[NG][64]
    static constexpr uint32_t GRP_BYTES = (NG / 4) * 64 * 8;  // u64 grp[NG/4][64]
    static constexpr uint32_t SUP_BYTES = NB > 64 ? 64 * 8 : 0;
    static constexpr uint32_t RING_BYTES = 8 * 64 * 4;        // u32 oring[8][64]
    static constexpr uint32_t WIN_BYTES = VIDC_DWIN * 64 * 4; // stream window
    static constexpr uint32_t PST_BYTES = VIDC_DPST * 64 * 4; // pushed words
    static constexpr uint32_t LDS_BYTES = CNT_BYTES + GRP_BYTES + SUP_BYTES + RING_BYTES + WIN_BYTES + PST_BYTES;
    static constexpr uint32_t BITS = NB == 64 ? 6u : (NB == 128 ? 7u : 8u);
};

// BATCH (default since round 4; VIDC_LANE_LOOP=1: the loop form): the first four 16-byte chunks of the bucket's row are
// requested together and counted behind ONE wait; the loop form waits for every chunk in turn, and the wavefront makes as many
// round trips as its fullest bucket has chunks (3-5 at ~16 members for the longest lists of a class).  65 536 lists of 1024
// ids: 3.9 -> 3.1-3.5 ms, S2 decode 78 -> 76 ms.  Measured on top and dropped: the row store of a step
// issued BEHIND the next step's loads with the id forwarded from a register (one asm statement, wait = vmcnt(stores behind
// the loads)) -- bit-exact, and no faster than this form whether it waited for vmcnt(2) or vmcnt(0): the kernel is bound by
// the number of scattered L2 transactions per step (~5 per lane), not by the store acknowledgements a load waits behind.
template <int NB, bool BATCH = true>
__global__ void __launch_bounds__(64) k_roc_decode_lane(RocDecArgs a, const LaneDiv *__restrict__ dtab) {
    using G = LaneDecGeom<NB>;
    __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];
    uint4 *cnt4 = (uint4 *)smem;
    unsigned char *cnt1 = smem;
    uint64_t *grp = (uint64_t *)(smem + G::CNT_BYTES);
    uint64_t *sup = (uint64_t *)(smem + G::CNT_BYTES + G::GRP_BYTES);
    // decoded ids wait in an 8-deep LDS ring and leave 8 at a time: the 8 stores of a lane hit one or two 64-byte
    // lines back to back, so a line is completed in L2 instead of being written back to HBM partially up to 8 times
    uint32_t *oring = (uint32_t *)(smem + G::CNT_BYTES + G::GRP_BYTES + G::SUP_BYTES);
    const uint32_t lane = lane_id();
    const uint32_t lpw = a.lpw ? a.lpw : 64u;
    // (the host may launch fewer wavefronts than the work list has groups of lpw lists: a wavefront then takes every
    // gridDim.x-th group -- fewer lists in flight keep their bucket rows in the caches)
    for (uint32_t blk = blockIdx.x; blk * lpw < a.nwork; blk += gridDim.x) {
    const uint32_t wi = blk * lpw + lane;
    const bool have = lane < lpw && wi < a.nwork;
    const uint32_t l = have ? a.worklist[wi] : 0u;
    const uint32_t n = have ? (uint32_t)(a.offsets[l + 1] - a.offsets[l]) : 0u;
    const uint64_t ooff = have ? (a.out_off ? a.out_off[wi] : a.offsets[l]) : 0ull;
    const uint32_t P = have ? a.prec[l] : 0u;
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    const uint32_t bsh = P > G::BITS ? P - G::BITS : 0u;
    const uint32_t cap = roc_lane_cap_nb<NB>(n, a.row_align);
#pragma unroll
    for (int g = 0; g < (int)G::NG; g++) cnt4[g * 64 + lane] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < (int)G::NG / 4; q++) grp[q * 64 + lane] = 0;
    if (NB > 64) sup[lane] = 0;

    LWStack st;
    st.ring = (uint32_t *)(smem + G::CNT_BYTES + G::GRP_BYTES + G::SUP_BYTES + G::RING_BYTES) + lane;
    st.pst = (uint32_t *)(smem + G::CNT_BYTES + G::GRP_BYTES + G::SUP_BYTES + G::RING_BYTES + G::WIN_BYTES) + lane;
    lw_init(st, have, a.words, have ? a.word_off[l] : 0ull, have ? a.nwords[l] : 0u);
    st.mt = a.mt;
    st.draws = have ? a.draws[l] : 0u;
    const uint32_t draws0 = st.draws;
    uint64_t head = have ? a.heads[l] : VIDC_RANS_L;
    uint32_t *slots = a.slots + (have ? a.slots_off[wi] : 0ull);
    uint32_t n_eff = n;
    bool retry = false;
    const uint32_t nsteps = wave_max_u32(n);

    for (uint32_t i = 0; i < nsteps; i++) {
        const uint32_t lq = dtab[i + 1u].w;  // uniform: floor(2^31 / (i + 1))
        uint4 pf = make_uint4(0, 0, 0, 0);
        const bool pf_go = (i & 1u) == 0u && i < n_eff && lw_issue(st, pf);  // a step pops at most two words
        if (i < n_eff) {
            // ---- x = ID_pop(P), codec.cpp:107-121 (slices 3, 2 have precision 0: refill test only)
            if (__builtin_expect(l_lt_2p31(head), 0)) {
                (void)l_u_pop(head, st, 0u);
                (void)l_u_pop(head, st, 0u);
            }
            const uint32_t hi = l_u_pop(head, st, p1);
            const uint32_t lo = l_u_pop(head, st, p0);
            const uint32_t x = (hi << 16) | lo;
            // ---- rank of x among the decoded ids
            const uint32_t b = (bsh >= 32u ? 0u : (x >> bsh)) & (uint32_t)(NB - 1);
            const uint32_t g = b >> 4, j = b & 15u;  // group of 16 buckets, bucket inside the group
            const uint32_t q = g >> 2;               // super-group (0 for NB = 64)
            uint32_t r = 0;
            uint64_t sv = 0;
            if (NB > 64) {
                sv = sup[lane];
                r = lane_sum_u16_below(sv, q);
            }
            const uint64_t gv = grp[q * 64 + lane];
            r += lane_sum_u16_below(gv, g & 3u);
            const uint4 cv = cnt4[g * 64 + lane];
            const uint64_t c_lo = ((uint64_t)cv.y << 32) | cv.x, c_hi = ((uint64_t)cv.w << 32) | cv.z;
            r += lane_sum_bytes_below(c_lo, j < 8u ? j : 8u);
            r += j > 8u ? lane_sum_bytes_below(c_hi, j - 8u) : 0u;
            const uint32_t cb = cnt1[(g * 64 + lane) * 16 + j];
            const uint4 *row4 = (const uint4 *)(slots + (size_t)b * cap);
            if (BATCH) {
                // (unconditional loads at clamped chunk numbers, counts masked afterwards: a load under a lane predicate is
                // followed by its own wait -- the compiler sinks the compares into the predicated block)
                const uint32_t nch = (cb + 3u) >> 2;
                const uint32_t last = nch ? nch - 1u : 0u;
                const uint4 v0 = row4[0];
                const uint4 v1 = row4[last < 1u ? last : 1u];
                const uint4 v2 = row4[last < 2u ? last : 2u];
                const uint4 v3 = row4[last < 3u ? last : 3u];
                const uint32_t c0 = (v0.x < x) + (v0.y < x) + (v0.z < x) + (v0.w < x);
                const uint32_t c1 = (v1.x < x) + (v1.y < x) + (v1.z < x) + (v1.w < x);
                const uint32_t c2 = (v2.x < x) + (v2.y < x) + (v2.z < x) + (v2.w < x);
                const uint32_t c3 = (v3.x < x) + (v3.y < x) + (v3.z < x) + (v3.w < x);
                r += (nch > 0u ? c0 : 0u) + (nch > 1u ? c1 : 0u) + (nch > 2u ? c2 : 0u) + (nch > 3u ? c3 : 0u);
                for (uint32_t c = 4u; c < nch; c++) {
                    const uint4 v = row4[c];
                    r += (v.x < x) + (v.y < x) + (v.z < x) + (v.w < x);
                }
            } else
            for (uint32_t c = 0; c * 4u < cb; c++) {
                const uint4 v = row4[c];
                r += (v.x < x) + (v.y < x) + (v.z < x) + (v.w < x);
            }
            // ---- IDX_push(r, i + 1), codec.cpp:44-63
            {
                uint64_t h0 = head;
                if ((uint32_t)(h0 >> 32) >= lq) {
                    ls_push(st, (uint32_t)h0);
                    h0 >>= 32;
                }
                uint64_t h = h0 * (uint64_t)(i + 1u) + r;
                if (__builtin_expect(l_lt_2p31(h), 0)) h = (uint64_t)ls_pop(st) | (h << 32);
                head = h;
            }
            // ---- insert x
            if (__builtin_expect(cb >= cap, 0)) {
                retry = true;  // skewed ids: this bucket is full -> the wave-per-list kernel redoes the list
                n_eff = 0;
            } else {
                uint32_t *row = slots + (size_t)b * cap;
                if ((cb & 3u) == 0u) *(uint4 *)(row + cb) = make_uint4(x, 0xffffffffu, 0xffffffffu, 0xffffffffu);
                else row[cb] = x;
                cnt1[(g * 64 + lane) * 16 + j] = (unsigned char)(cb + 1u);
                grp[q * 64 + lane] = gv + (1ull << (16u * (g & 3u)));
                if (NB > 64) sup[lane] = sv + (1ull << (16u * q));
                oring[(i & 7u) * 64u + lane] = x;
            }
        }
        lw_land(st, pf_go, pf);
        if ((i & 7u) == 7u || i + 1u == nsteps) {  // uniform: flush the ring (steps s0 .. i)
            const uint32_t s0 = i & ~7u;
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t sstep = s0 + k;
                // (lists handed back for a retry are rewritten as a whole: whatever they store here is harmless)
                if (sstep <= i && sstep < n) a.out[ooff + (n - 1u - sstep)] = (uint64_t)oring[k * 64u + lane];
            }
        }
    }
    if (have) {
        retry |= (st.err & 4u) != 0u;
        const bool clean = (head == VIDC_RANS_L) && (st.otop - st.al + st.d == st.draws - draws0);
        a.end_state[l] = (clean || retry) ? 0u : 1u;
        a.status[l] = retry ? VIDC_ST_RETRY : ((st.err & 2u) ? VIDC_ST_MT : VIDC_ST_OK);
    }
    }  // next group of lists
}

// ---------------------------------------------------------------------------------------------------------------
// decoder for lists of 65 .. 256 ids per lane with NOTHING in global memory but the stored stream and the output: the
// ids decoded so far sit in the lane's own registers (slot i = the id of step i, 0xffffffff while empty) and the rank
// of x is the number of slots below x -- two VALU instructions per slot (the sign of slot - x is shifted into a bit
// string, one popcount per 16 slots), in blocks of 16 entered by a computed jump at the highest block that holds an id (roc_lane_reg_asm.h; the step
// counter is wave-uniform, so slot i is one register for the whole wavefront, written with s_set_gpr_idx).  The 192
// slots are pinned to v64..v255; slots from 192 on live in an LDS strip.  The bucket kernel above paid one random 64-byte row read and
// a row write-back per step (881 MB fetched + 657 MB written for 134 MB of ids on 65 536 x 256; a step took 3.5 us with
// every list in flight at once); here a step is issue-bound: ~2 i + 120 instructions.
#define VIDC_LANE_REG_MAX 256u
#ifndef VIDC_LANE_REG_EL
#define VIDC_LANE_REG_EL 192  // slots in registers (the rest of the 256 in LDS)
#endif
// PAIR: a list of 257 .. 512 ids per PAIR of lanes (32 lists per wavefront).  Both lanes run the list's ANS chain (the state
// is replicated, like the 16 lanes of a row in roc_grp.h); the id of step i goes to slot i >> 1 of the lane with the parity
// of i, so each lane still holds at most 256 slots, and the rank is the sum of the two lanes' counts (one DPP swap).  The
// bucket-row decoder these lists used before read and wrote a random 64-byte line per step (S2: 114 bytes of HBM traffic
// per id on 217 M ids); here a step touches no memory but the stream window and the output ring.
// LPL = 4: the same on QUADS of lanes (16 lists per wavefront) for lists of 513 .. 1024 ids: slot i >> 2 of lane i & 3, rank = sum of
// the four lanes (two DPP swaps).  65 536 lists of 1024 ids decode in the time of 1024 steps at ~400 instructions each instead
// of 1024 steps of 3.6 us with a bucket row read and written back per step.
#define VIDC_LANE_PAIR_MAX (2u * VIDC_LANE_REG_MAX)
#define VIDC_LANE_QUAD_MAX (4u * VIDC_LANE_REG_MAX)
template <int EL, int LPL = 1>
struct LaneRegGeom {
    static constexpr uint32_t TAIL = VIDC_LANE_REG_MAX - EL;          // slots kept in LDS
    static constexpr uint32_t TAIL_BYTES = TAIL * 64 * 4;             // uint4 tail4[TAIL / 4][64]
    static constexpr uint32_t RING_BYTES = 8 * 64 * 4;
    static constexpr uint32_t WIN_BYTES = VIDC_DWIN * 64 * 4;
    static constexpr uint32_t PST_BYTES = VIDC_DPST * 64 * 4;
    static constexpr uint32_t NMAX = (uint32_t)LPL * VIDC_LANE_REG_MAX;
    static constexpr uint32_t LQ_BYTES = (NMAX + 4) * 4;  // floor(2^31 / d), d = 0 .. NMAX (a load per step otherwise)
    static constexpr uint32_t LDS_BYTES = TAIL_BYTES + RING_BYTES + WIN_BYTES + PST_BYTES + LQ_BYTES;
};
template <int EL, int LPL = 1>  // LPL lanes per list: 1, 2 (pairs) or 4 (quads)
__global__ void __launch_bounds__(64) k_roc_decode_lane_reg(RocDecArgs a, const LaneDiv *__restrict__ dtab) {
    static_assert(EL == 192, "the rank asm is generated for 192 slots in v64..v255");
    static_assert(LPL == 1 || LPL == 2 || LPL == 4, "lanes per list");
    constexpr bool PAIR = LPL > 1;
    using G = LaneRegGeom<EL, LPL>;
    __shared__ __align__(16) unsigned char smem[G::LDS_BYTES];
    const uint32_t lane = lane_id();
    uint4 *tail4 = (uint4 *)smem + lane;                       // chunk c of this lane at tail4[c * 64]
    uint32_t *tail1 = (uint32_t *)smem + lane * 4u;            // slot s: tail1[(s >> 2) * 256 + (s & 3)]
    uint32_t *oring = (uint32_t *)(smem + G::TAIL_BYTES);
    const uint32_t lpw = a.lpw ? a.lpw : 64u / (uint32_t)LPL;
    const uint32_t lslot = lane / (uint32_t)LPL;  // list of the wavefront this lane works on
    const uint32_t wi = blockIdx.x * lpw + lslot;
    const bool have = lslot < lpw && wi < a.nwork;
    const bool writer = (lane & ((uint32_t)LPL - 1u)) == 0u;  // the lane of a pair that stores the list's results
    const uint32_t l = have ? a.worklist[wi] : 0u;
    const uint32_t n = have ? (uint32_t)(a.offsets[l + 1] - a.offsets[l]) : 0u;
    const uint64_t ooff = have ? (a.out_off ? a.out_off[wi] : a.offsets[l]) : 0ull;
    const uint32_t P = have ? a.prec[l] : 0u;
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
#pragma unroll
    for (uint32_t c = 0; c < G::TAIL / 4u; c++) tail4[c * 64u] = make_uint4(~0u, ~0u, ~0u, ~0u);
    uint32_t *lqs = (uint32_t *)(smem + G::TAIL_BYTES + G::RING_BYTES + G::WIN_BYTES + G::PST_BYTES);
    for (uint32_t d = lane; d <= G::NMAX; d += 64u) lqs[d] = dtab[d].w;
    v32u e0, e1, e2, e3, e4, e5;  // slots 0..191, pinned to v64..v255 by the asm statements below
#pragma unroll
    for (int k = 0; k < 32; k++) e0[k] = e1[k] = e2[k] = e3[k] = e4[k] = e5[k] = 0x7fffffffu;  // empty: slot - x >= 0

    LWStack st;
    st.ring = (uint32_t *)(smem + G::TAIL_BYTES + G::RING_BYTES) + lane;
    st.pst = (uint32_t *)(smem + G::TAIL_BYTES + G::RING_BYTES + G::WIN_BYTES) + lane;
    lw_init(st, have, a.words, have ? a.word_off[l] : 0ull, have ? a.nwords[l] : 0u);
    st.mt = a.mt;
    st.draws = have ? a.draws[l] : 0u;
    const uint32_t draws0 = st.draws;
    uint64_t head = have ? a.heads[l] : VIDC_RANS_L;
    const uint32_t nsteps = wave_max_u32(n);

    wave_sync();
    // the look-ahead load of the stream window is issued at the END of a step, ahead of the step's output stores, and
    // lands at the end of the next one: a load issued behind the stores would wait for their write acknowledgements
    uint4 pf = make_uint4(0, 0, 0, 0);
    bool pf_go = false;
    // rank of xx among the ids in slots 0 .. i-1 of this lane
    auto rank_of = [&](uint32_t xx, uint32_t i) __attribute__((always_inline)) -> uint32_t {
        uint32_t r = 0;
        const uint32_t nb = i >= (uint32_t)EL ? (uint32_t)EL / 16u : (i + 15u) >> 4;  // register blocks holding an id
        const uint32_t jmp = 12u + VIDC_LREG_BLOCK_BYTES * ((uint32_t)EL / 16u - nb) + (nb <= 8u ? VIDC_LREG_FLUSH_BYTES : 0u);
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, t0, t1, t2, t3;
        asm volatile(VIDC_LREG_RANK_ASM
                     : [r] "+v"(r), [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t0] "=&v"(t0),
                       [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1),
                       "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4), "+{v[224:255]}"(e5)
                     : [x] "v"(xx), [jmp] "s"(jmp)
                     : "s28", "s29", "scc");
        if (i > (uint32_t)EL) {  // uniform: the LDS strip
            const uint32_t nch = (i - (uint32_t)EL + 3u) >> 2;
            for (uint32_t c = 0; c < nch; c++) {
                const uint4 v = tail4[c * 64u];
                r += (v.x < xx) + (v.y < xx) + (v.z < xx) + (v.w < xx);
            }
        }
        return r;
    };
    for (uint32_t i = 0; i < nsteps; i++) {
        const uint32_t lq = lqs[i + 1u];  // uniform: floor(2^31 / (i + 1))
        const bool act = i < n;
        uint32_t x = 0;
        if (act) {
            // ---- x = ID_pop(P), codec.cpp:107-121
            if (__builtin_expect(l_lt_2p31(head), 0)) {
                (void)l_u_pop(head, st, 0u);
                (void)l_u_pop(head, st, 0u);
            }
            const uint32_t hi = l_u_pop(head, st, p1);
            const uint32_t lo = l_u_pop(head, st, p0);
            x = (hi << 16) | lo;
        }
        // (PAIR: the lane with the parity of a step holds its id: after i steps at most (i + 1) / 2 slots of a lane are in use)
        uint32_t r = rank_of(x, (i + (uint32_t)LPL - 1u) / (uint32_t)LPL);
        if (LPL >= 2) r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0xb1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]: the partner's count
        if (LPL == 4) r += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x4e, 0xf, 0xf, false);  // quad_perm [2, 3, 0, 1]: the other pair's
        if (act) {
            // ---- IDX_push(r, i + 1), codec.cpp:44-63
            uint64_t h0 = head;
            if ((uint32_t)(h0 >> 32) >= lq) {
                ls_push(st, (uint32_t)h0);
                h0 >>= 32;
            }
            uint64_t h = h0 * (uint64_t)(i + 1u) + r;
            if (__builtin_expect(l_lt_2p31(h), 0)) h = (uint64_t)ls_pop(st) | (h << 32);
            head = h;
        }
        if (act) oring[(i & 7u) * 64u + lane] = x;
        // ---- slot i = x (lanes past their list only overwrite an empty slot they never read again)
        if (PAIR) {
            const uint32_t sl = i / (uint32_t)LPL;            // slot, in the lane with the parity (LPL = 4: residue mod 4) of the step
            const bool mine = ((lane ^ i) & ((uint32_t)LPL - 1u)) == 0u;
            if (sl < (uint32_t)EL) {
                // The lanes of the other parity are masked off by exec, as in the one-lane-per-list kernel.  The first version wrote
                // the slot with "v_cndmask_b32_e64 v64, v64, x, mask" right behind "s_set_gpr_idx_on i, gpr_idx(SRC0,DST)": bit-exact by
                // itself, but S2-sized decodes came back with wrong lists of OTHER kernel classes in ~2 % of the calls (16 / 32 / 48
                // lists of a lane<64> wavefront, single general-decoder lists, once a memory fault).  Round 4 took the construct apart
                // (the forms below, tools/pair_forms.sh, 600 S2 decodes each, profiles/r04c_pair_forms.txt): every form that issues a
                // VOP3 instruction with an SRC0-relative register IMMEDIATELY behind s_set_gpr_idx_on fails (9-17 of 600, whatever
                // produces the mask, also with wait states in front of the switch or behind the switch back); the same instruction
                // with s_nop 4 between the switch and itself: 0 of 1400; the compiler's own pattern -- a VOP1 v_mov under the index
                // mode -- with or without the exec mask: 0 of 400 (and of the thousands of stress decodes of the shipped kernels).
                // So gfx950 wants wait states between s_set_gpr_idx_on and a VOP3 instruction that reads a register through the
                // index; without them the instruction can address registers outside its wavefront (the victims are wavefronts of
                // other kernels on the same SIMD, hit in quarter-wave groups of lanes).  Not reproducible with ALU-only victims
                // outside the library (tools/hw_gpr_idx_probe.hip).  The shipped statements keep to the VOP1 pattern and put a wait
                // state behind the switch all the same.  DESIGN sections 10 and 11.
#ifdef VIDC_PAIR_OLD_FORM  // (investigations only, tools/pair_forms.sh: 1 = the round-3 form, 2.. = variants that take it apart; failures of 600 S2 decodes
                          //  1: 10, 4: 15, 7: 9, 8: 17, 10: memory fault, 11: 11 | 5: 0, 9: 0 | of 200: 2: 0, 3: 0, 6: 0)
                const uint64_t m = __ballot(mine);
#if VIDC_PAIR_OLD_FORM == 1
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
#elif VIDC_PAIR_OLD_FORM == 2  // no VOP3 under the index mode: indexed read, plain select, indexed write
                uint32_t t_;
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0)\n\tv_mov_b32 %[t], v64\n\ts_set_gpr_idx_off"
                             : [t] "=v"(t_), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5) : [i] "s"(sl));
                t_ = mine ? x : t_;
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(DST)\n\tv_mov_b32 v64, %[t]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5) : [t] "v"(t_), [i] "s"(sl));
#elif VIDC_PAIR_OLD_FORM == 3  // VOP3 under the index mode, DST relative only
                uint32_t t_;
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0)\n\tv_mov_b32 %[t], v64\n\ts_set_gpr_idx_off"
                             : [t] "=v"(t_), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5) : [i] "s"(sl));
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(DST)\n\tv_cndmask_b32_e64 v64, %[t], %[x], %[m]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5) : [t] "v"(t_), [x] "v"(x), [i] "s"(sl), [m] "s"(m));
#elif VIDC_PAIR_OLD_FORM == 4  // VOP3 under the index mode, SRC0 relative only
                uint32_t t_;
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0)\n\tv_cndmask_b32_e64 %[t], v64, %[x], %[m]\n\ts_set_gpr_idx_off"
                             : [t] "=v"(t_), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5) : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(DST)\n\tv_mov_b32 v64, %[t]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5) : [t] "v"(t_), [i] "s"(sl));
#elif VIDC_PAIR_OLD_FORM == 5  // the round-3 form between wait states
                asm volatile("s_nop 4\n\ts_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\ts_nop 4\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_nop 4\n\ts_set_gpr_idx_off\n\ts_nop 4"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
#elif VIDC_PAIR_OLD_FORM == 6  // the shipped instruction (v_mov, DST relative) without the exec mask: the select is done on plain registers first
                uint32_t t_;
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0)\n\tv_mov_b32 %[t], v64\n\ts_set_gpr_idx_off"
                             : [t] "=v"(t_), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5) : [i] "s"(sl));
                asm volatile("v_cndmask_b32_e64 %[t], %[t], %[x], %[m]\n\ts_set_gpr_idx_on %[i], gpr_idx(DST)\n\tv_mov_b32 v64, %[t]\n\ts_set_gpr_idx_off"
                             : [t] "+v"(t_), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5) : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
#elif VIDC_PAIR_OLD_FORM == 7  // the round-3 form with the two wait states gfx940+ want between a VALU that writes an SGPR pair and a VALU that reads it as a mask
                asm volatile("s_nop 1\n\ts_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
#elif VIDC_PAIR_OLD_FORM == 8  // the round-3 form with a mask that comes from the scalar unit (lane pairs: a constant per parity of the step)
                const uint64_t m8 = LPL == 2 ? ((i & 1u) ? 0xaaaaaaaaaaaaaaaaull : 0x5555555555555555ull) : m;
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m8));
#elif VIDC_PAIR_OLD_FORM == 9  // wait states between the mode switch and the VOP3 instruction only
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\ts_nop 4\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
#elif VIDC_PAIR_OLD_FORM == 10  // wait states between the VOP3 instruction and the switch back only
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_nop 4\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
#elif VIDC_PAIR_OLD_FORM == 11  // wait states behind the switch back only
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_set_gpr_idx_off\n\ts_nop 4"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
#endif
#else
                if (mine) {
                    asm volatile("s_set_gpr_idx_on %[i], gpr_idx(DST)\n\ts_nop 0\n\tv_mov_b32 v64, %[x]\n\ts_set_gpr_idx_off"
                                 : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                                   "+{v[224:255]}"(e5)
                                 : [x] "v"(x), [i] "s"(sl));
                }
#endif
            } else if (act && mine) {
                const uint32_t sidx = sl - (uint32_t)EL;
                tail1[(sidx >> 2) * 256u + (sidx & 3u)] = x;
            }
        } else if (i < (uint32_t)EL) {
            asm volatile("s_set_gpr_idx_on %[i], gpr_idx(DST)\n\ts_nop 0\n\tv_mov_b32 v64, %[x]\n\ts_set_gpr_idx_off"
                         : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                           "+{v[224:255]}"(e5)
                         : [x] "v"(x), [i] "s"(i));
        } else if (act) {
            const uint32_t sidx = i - (uint32_t)EL;
            tail1[(sidx >> 2) * 256u + (sidx & 3u)] = x;
        }
        lw_land(st, pf_go, pf);
        pf_go = act && lw_issue(st, pf);
        if ((i & 7u) == 7u || i + 1u == nsteps) {  // uniform: flush the ring (steps s0 .. i)
            const uint32_t s0 = i & ~7u;
#pragma unroll
            for (uint32_t k = 0; k < 8u; k++) {
                const uint32_t sstep = s0 + k;
                if (writer && sstep <= i && sstep < n) a.out[ooff + (n - 1u - sstep)] = (uint64_t)oring[k * 64u + lane];
            }
        }
    }
    if (have && writer) {
        const bool retry = (st.err & 4u) != 0u;
        const bool clean = (head == VIDC_RANS_L) && (st.otop - st.al + st.d == st.draws - draws0);
        a.end_state[l] = (clean || retry) ? 0u : 1u;
        a.status[l] = retry ? VIDC_ST_RETRY : ((st.err & 2u) ? VIDC_ST_MT : VIDC_ST_OK);
    }
}

// ===============================================================================================================
// TINY lists / graph rows (n <= 64) per lane: the whole codec state of a list is on chip.
//   encoder: the row is sorted by a compile-time bitonic network over 64 (or 32) registers of the lane (672
//            min/max pairs for 64 rows at once), the sorted ids go to an LDS strip, the alive set is one u64 mask
//   decoder: the stream words are copied to an LDS strip (the ANS stack shrinks from its top while the decoded
//            ids grow down from the end of the same strip), rank = linear scan over the decoded ids
// No global load sits on the serial chain (the divisor table is copied to LDS), so a step never waits on memory.
// ===============================================================================================================
// Graph rows arrive as a TILE (rows_tile.h): the wavefront's 64 rows are one contiguous block, read with coalesced 16-byte loads and
// transposed through the strip's LDS (FULL: K == KP).  Every lane reading its own row 256 bytes from its neighbour's -- 16 KiB of cache
// lines per wavefront, nine wavefronts on a 32 KiB L1 -- made everything in front of the step loop 98 us on 10^6 x 64 rows; as a tile
// 79 (44 load + edges + strip, 27 sort, 8 results).  The whole kernel stays at ~305 us: one wavefront's prologue runs under the
// loops of the others, and the loop is the vector-issue bound of DESIGN section 5.
template <int KP, bool ROWS, bool FULL = false>
__global__ void __launch_bounds__(64) k_roc_encode_tiny_lane(RocEncArgs a, const LaneDiv *__restrict__ dtab) {
    __shared__ __attribute__((aligned(16))) uint32_t sid[KP * 64 + (ROWS ? 64 : 0)];  // (+ 64: the tile image of rows narrower than KP has a leading dimension of KP + 1)
    __shared__ LaneDiv dt[KP + 1];
    const uint32_t lane = lane_id();
    const uint32_t wi = blockIdx.x * 64u + lane;
    const bool have = wi < a.nwork;
    const uint32_t l = have ? (a.worklist ? a.worklist[wi] : wi) : 0u;
    for (uint32_t t = lane; t <= (uint32_t)KP; t += 64) dt[t] = dtab[t];

    uint32_t r[KP];
    uint32_t n = 0;
    uint64_t off = 0;
    bool bad = false, unsorted = false;
    if (ROWS) {
        // altid_impl.cpp:110-117: edges up to the first -1 (rows are never on a work list: lane t = row blockIdx.x * 64 + t)
        off = (uint64_t)l * a.K;
        const uint64_t row0 = (uint64_t)blockIdx.x * 64u;
        const uint32_t nrows = a.nwork - (uint32_t)row0 < 64u ? a.nwork - (uint32_t)row0 : 64u;
        const bool pf = FULL && (((uintptr_t)a.rows) & 15u) == 0u;
        {
            TileRegs<KP> pre;
            if (pf) tile_issue_rows<KP>(a.rows, row0, nrows, pre);
            tile_commit_rows<KP, FULL>(a.rows, row0, nrows, a.K, tile_magic(a.K), pf, pre, sid, r);
        }
        uint32_t row_max;
        n = tile_row_edges<KP>(r, bad, row_max);
        if (!have) { n = 0; bad = false; }  // (lanes behind the last row of the last tile hold whatever the LDS held)
    } else {
        off = have ? a.offsets[l] : 0ull;
        n = have ? (uint32_t)(a.offsets[l + 1] - off) : 0u;
        uint32_t prev = 0;
#pragma unroll
        for (int e = 0; e < KP; e++) {
            uint64_t id = 0;
            if ((uint32_t)e < n) id = a.ids[off + e];
            bad |= id >= (1ull << 31);  // reference: int max_id (custom_invlists_impl.cpp:163)
            const uint32_t v = (uint32_t)id;
            unsorted |= (uint32_t)e < n && e > 0 && prev >= v;
            prev = v;
            r[e] = (uint32_t)e < n ? v : 0xffffffffu;
        }
    }
    uint32_t mx = 0;
#pragma unroll
    for (int e = 0; e < KP; e++) mx = ((uint32_t)e < n && r[e] > mx) ? r[e] : mx;
    const uint32_t P = precision_for(mx, a.precision_mode);
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    // graph rows are in no particular order; IVF lists normally ascend (Faiss add order)
    const bool want_perm = !ROWS && a.perm != nullptr;
    const bool pending = want_perm && unsorted;  // input positions of an unsorted list: wave-per-list kernel
    if (ROWS || ballot(unsorted && !pending)) lane_sort<KP>(r);
#pragma unroll
    for (int e = 0; e < KP; e++) sid[e * 64 + lane] = r[e];
    if (ROWS && have) a.sizes[l] = n;
    __syncthreads();

    LStack st;
    {
        const uint64_t ao = have ? arena_at(a, l) : 0ull;
        st.mem = a.arena + ao;
        st.orig = st.mem;
        st.cap = have ? (uint32_t)(arena_at(a, l + 1) - ao) : 0u;
        st.sp = 0; st.dirty = 0; st.draws = 0; st.err = 0; st.mt = a.mt;
    }
    uint64_t head = VIDC_RANS_L;
    uint64_t alive = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
    const uint32_t n_eff = (bad || pending) ? 0u : n;
    const uint32_t nsteps = wave_max_u32(n_eff);
    for (uint32_t i = 0; i < nsteps; i++) {
        if (i < n_eff) {
            const uint32_t d = n - i;
            const LaneDiv dv = dt[d];
            uint64_t h0 = head;
            if ((uint32_t)(h0 >> 32) >= dv.z) {
                ls_push(st, (uint32_t)h0);
                h0 >>= 32;
            }
            uint64_t q = __umul64hi(h0, ((uint64_t)dv.y << 32) | dv.x);
            uint32_t k = (uint32_t)h0 - (uint32_t)q * d;
            if (k >= d) { k -= d; q++; }
            if (__builtin_expect(l_lt_2p31(h0), 0)) q = (uint64_t)ls_pop(st) | (q << 32);
            head = q;
            const uint32_t pos = lane_select64(alive, k);
            alive &= ~(1ull << pos);
            const uint32_t x = sid[pos * 64 + lane];
            if (want_perm) a.perm[off + i] = pos;
            l_u_push(head, st, x & 0xffffu, p0);
            l_u_push(head, st, x >> 16, p1);
            if (__builtin_expect((uint32_t)(head >> 63) != 0u, 0)) {
                l_u_push(head, st, 0u, 0u);
                l_u_push(head, st, 0u, 0u);
            }
        }
    }
    if (have) {
        if (bad) {
            a.status[l] = VIDC_ST_DOMAIN;
        } else if (pending) {
            a.status[l] = VIDC_ST_PENDING_SORT;
        } else {
            a.heads[l] = head;
            a.prec[l] = n ? P : 0u;
            a.nwords[l] = st.sp;
            a.draws[l] = st.draws;
            a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
        }
    }
}

#define VIDC_TINY_STRIP 48u  // LDS words per lane of the tiny decoder: a window over the top of the list's stream (word w at slot w mod 48)
#define VIDC_TINY_LD 65u     // word w of lane t at buf[w * 65 + t]: conflict-free per lane AND per row (output phase)

// Round 6: the stream words come in eight loads at a time (the loop that loaded one word, waited and stored it made ~40 memory round
// trips in a row per wavefront -- most of the kernel's 53 us of residence on 10^6 graph rows), the step's divisor constant comes from a
// register (it was a scalar load per step), the decoded ids live ONLY in the lane's registers during the loop (the strip holds nothing
// but the stream: 12.4 instead of 21.3 KiB per wavefront, 12 instead of 7 wavefronts per CU) and go through the strip's LDS 32
// positions at a time on the way out, where a lane stores 16 bytes of a row (graph flavour, K % 4 == 0).
// The strip is a WINDOW of 48 words over the top of the stream (a list of 64 ids at P <= 20 has ~41 words; at P = 32 up to 74): the
// decoder consumes the stream from its top, and a lane that reaches the bottom of its window with words left below it loads them
// (all of them fit: <= 74 - 48) in the branch that also serves the empty stack.
template <bool ROWS>
__global__ void __launch_bounds__(64) k_roc_decode_tiny_lane(RocDecArgs a, const LaneDiv *__restrict__ dtab) {
    __shared__ __attribute__((aligned(16))) uint32_t buf[VIDC_TINY_STRIP * VIDC_TINY_LD];
    __shared__ uint32_t rown[64], rowix[64];
    const uint32_t lane = lane_id();
    const uint32_t wi = blockIdx.x * 64u + lane;
    const bool have = wi < a.nwork;
    const uint32_t l = have ? (a.worklist ? a.worklist[wi] : wi) : 0u;
    const uint32_t n = have ? (uint32_t)(a.offsets[l + 1] - a.offsets[l]) : 0u;
    const uint32_t orow = (ROWS && a.out_by_list) ? l : wi;  // output row of a graph decode: the item's number, or the row's own (ordered work list)
    const uint64_t ooff = have ? (a.out_off ? a.out_off[wi] : (ROWS ? (uint64_t)orow * a.K : a.offsets[l])) : 0ull;
    const uint32_t P = have ? a.prec[l] : 0u;
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    const uint32_t W = have ? a.nwords[l] : 0u;
    const uint32_t *orig = a.words + (have ? a.word_off[l] : 0ull);
    const uint32_t lqv = dtab[lane + 1u].w;  // lane i: floor(2^31 / (i + 1)), the constant of step i
    uint32_t err = (W > 2u * VIDC_TINY_STRIP - 2u) ? 1u : 0u;
    const uint32_t Wc = err ? 0u : W;
    uint32_t base = Wc > VIDC_TINY_STRIP ? Wc - VIDC_TINY_STRIP : 0u;  // words [base, sp) are in the window
    auto slot = [&](uint32_t w) -> uint32_t {  // w mod 48 for w < 96
        const uint32_t v = w - VIDC_TINY_STRIP;
        return (w < v ? w : v) * VIDC_TINY_LD + lane;
    };
    {
        const uint32_t wmax = wave_max_u32(Wc - base);
        for (uint32_t w0 = 0; w0 < wmax; w0 += 8u) {
            uint32_t t[8];
#pragma unroll
            for (int j = 0; j < 8; j++) t[j] = base + w0 + (uint32_t)j < Wc ? orig[base + w0 + (uint32_t)j] : 0u;
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (base + w0 + (uint32_t)j < Wc) buf[slot(base + w0 + (uint32_t)j)] = t[j];
        }
    }
    uint32_t sp = Wc;
    uint32_t draws = have ? a.draws[l] : 0u;
    const uint32_t draws0 = draws;
    uint64_t head = have ? a.heads[l] : VIDC_RANS_L;
    const uint32_t n_eff = err ? 0u : n;
    const uint32_t nsteps = wave_max_u32(n_eff);

    auto pop = [&]() -> uint32_t {  // codec.h:32-40
        if (__builtin_expect(sp == base, 0)) {
            if (base == 0u) {
                uint32_t w = 0;
                if (draws < VIDC_MT_TABLE) w = a.mt[draws]; else err |= 2u;
                draws++;
                return w;
            }
            for (uint32_t w = 0; w < base; w++) buf[slot(w)] = orig[w];  // (once per list at most: base <= 46 after this)
            base = 0u;
        }
        sp--;
        return buf[slot(sp)];
    };
    auto u_pop = [&](uint32_t p) -> uint32_t {  // codec.cpp:78-90
        const uint32_t sym = (uint32_t)head & ((1u << p) - 1u);
        head >>= p;
        if (l_lt_2p31(head)) head = (head << 32) | pop();
        return sym;
    };

    // the ids decoded so far sit in 64 registers of the lane (slot i = step i, 0x7fffffff while empty): the rank
    // is roc_lane_reg_asm.h's sign-bit count over the blocks of 16 slots in use (two instructions per id)
    v32u e0, e1;
#pragma unroll
    for (int k = 0; k < 32; k++) e0[k] = e1[k] = 0x7fffffffu;
    for (uint32_t ii = 0; ii < nsteps; ii++) {
        const uint32_t i = rfl(ii);      // (kept in a scalar register: it indexes registers below)
        const uint32_t lq = rl(lqv, i);  // uniform: floor(2^31 / (i + 1))
        uint32_t xs = 0x7fffffffu;
        if (i < n_eff) {
            if (__builtin_expect(l_lt_2p31(head), 0)) {
                (void)u_pop(0u);
                (void)u_pop(0u);
            }
            const uint32_t hi = u_pop(p1);
            const uint32_t lo = u_pop(p0);
            const uint32_t x = (hi << 16) | lo;
            xs = x;
            // rank among the i ids decoded so far (strictly smaller, fenwick_tree.h:42-94)
            uint32_t r = 0;
            {
                const uint32_t nb = (i + 15u) >> 4;
                const uint32_t jmp = 12u + VIDC_LREG_BLOCK_BYTES * (4u - nb);
                uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, t0, t1, t2, t3;
                asm volatile(VIDC_LREG64_RANK_ASM
                             : [r] "+v"(r), [a0] "+v"(a0), [a1] "+v"(a1), [a2] "+v"(a2), [a3] "+v"(a3), [t0] "=&v"(t0),
                               [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1)
                             : [x] "v"(x), [jmp] "s"(jmp)
                             : "s28", "s29", "scc");
            }
            // IDX_push(r, i + 1), codec.cpp:44-63
            {
                uint64_t h0 = head;
                if (__builtin_expect((uint32_t)(h0 >> 32) >= lq, 0)) {
                    if (sp - base < VIDC_TINY_STRIP) buf[slot(sp)] = (uint32_t)h0; else err |= 1u;
                    sp++;
                    h0 >>= 32;
                }
                uint64_t h = h0 * (uint64_t)(i + 1u) + r;
                if (__builtin_expect(l_lt_2p31(h), 0)) h = (uint64_t)pop() | (h << 32);
                head = h;
            }
        }
        // slot i = x (uniform register index; lanes past their list keep the empty value)
        asm volatile("s_set_gpr_idx_on %[i], gpr_idx(DST)\n\ts_nop 0\n\tv_mov_b32 v64, %[x]\n\ts_set_gpr_idx_off"
                     : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1)
                     : [x] "v"(xs), [i] "s"(i));
    }
    if (have) {
        const bool clean = (head == VIDC_RANS_L) && (sp == draws - draws0);
        a.end_state[l] = (clean || n == 0u) ? 0u : 1u;
        a.status[l] = err ? ((err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
    }
    // output: decoded order == sampling order, the id of step i goes to position n-1-i (codec.cpp:150); graph rows are padded
    // with -1 (the reference leaves slots >= n untouched, altid_impl.cpp:153-165).  32 positions at a time through the strip's
    // LDS (position p of lane t at buf[(p & 31) * 65 + t]), then row-wise stores.
    rown[lane] = n_eff | (have ? 0x100u : 0u);
    rowix[lane] = orow;
    const bool quad = ROWS && (a.K & 3u) == 0u && (((uintptr_t)a.out_rows | (ooff * 4u)) & 15u) == 0u && !a.out_off;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        if (rfl((uint32_t)h * 32u) >= nsteps && !ROWS) break;  // (nothing in the upper half)
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 64; i++) {
            const uint32_t p = n_eff - 1u - (uint32_t)i;  // (wraps for i >= n: fails the range test)
            if ((p >> 5) == (uint32_t)h && p < n_eff) buf[(p & 31u) * VIDC_TINY_LD + lane] = i < 32 ? e0[i & 31] : e1[i & 31];
        }
        __syncthreads();
        if (ROWS && ballot(quad) == ~0ull) {
            // lane = (row of a group of eight, four consecutive positions of this half): 16-byte stores
            const uint32_t g = lane >> 3, q4 = (lane & 7u) * 4u, c0 = (uint32_t)h * 32u + q4;
            for (uint32_t r0 = 0; r0 < 64u; r0 += 8u) {
                const uint32_t rr = r0 + g;
                const uint32_t rm = rown[rr];
                if (!(rm & 0x100u) || c0 >= a.K) continue;
                const uint32_t n_r = rm & 0xffu;
                int4 o;
                o.x = c0 + 0u < n_r ? (int32_t)buf[(q4 + 0u) * VIDC_TINY_LD + rr] : -1;
                o.y = c0 + 1u < n_r ? (int32_t)buf[(q4 + 1u) * VIDC_TINY_LD + rr] : -1;
                o.z = c0 + 2u < n_r ? (int32_t)buf[(q4 + 2u) * VIDC_TINY_LD + rr] : -1;
                o.w = c0 + 3u < n_r ? (int32_t)buf[(q4 + 3u) * VIDC_TINY_LD + rr] : -1;
                *(int4 *)(a.out_rows + (uint64_t)rowix[rr] * a.K + c0) = o;
            }
        } else {
            for (uint32_t rr = 0; rr < 64u; rr++) {
                const uint32_t rm = rl(rown[lane], rr);
                if (!(rm & 0x100u)) break;  // work items fill the lanes from 0
                const uint32_t n_r = rm & 0xffu;
                const uint64_t o_r = rl64((uint32_t)ooff, (uint32_t)(ooff >> 32), rr);
                const uint32_t c = (uint32_t)h * 32u + lane;
                if (lane < 32u) {
                    if (ROWS) {
                        if (c < a.K) a.out_rows[o_r + c] = c < n_r ? (int32_t)buf[lane * VIDC_TINY_LD + rr] : -1;
                    } else if (c < n_r) {
                        a.out[o_r + c] = (uint64_t)buf[lane * VIDC_TINY_LD + rr];
                    }
                }
            }
        }
    }
}

}  // namespace dev
}  // namespace vidc
