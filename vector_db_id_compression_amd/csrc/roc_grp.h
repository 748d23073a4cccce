// roc_grp.h -- "one list per 16-lane ROW" Random Order Coding kernels for mid-size and long lists (4 097 .. 131 072
// ids): four lists per wavefront.
//
// Between the lane-per-list kernels (roc_lane.h: 64 chains per wavefront, order statistics in a private LDS strip per
// lane -- up to 4096 ids) and the wave-per-list kernels (roc_kernels.h / roc_u2.h: ONE chain per wavefront, ~120
// wave-wide instructions per codec step spent on wave-uniform arithmetic) sits the bulk of a large index: S2 has
// 26 316 lists of 4 097..32 768 ids (230 M ids) that ran 43 ms per direction on the general wave-per-list kernels
// (5 G steps/s: issue-bound, one step of one list per ~120 issue slots).  Here a list owns one DPP row of 16 lanes:
//   * the ANS state (head, stack pointers) is replicated in the 16 lanes of the row (VGPRs; redundant arithmetic is
//     free on a SIMD), LDS / memory side effects are done by one lane of the row;
//   * the order-statistic search is 16-ary: every level is 16 counters in inclusive-prefix form, one per lane -- the
//     top level in registers, the levels below in LDS -- "counter <= k" over the row gives the child (ballot, 16-bit
//     field, popcount), a row shuffle gives the prefix to subtract, the lanes at and behind the child update their
//     own counter with one masked store;
//   * every access of the row to the list's memory is one contiguous 64-byte piece: stream words leave / arrive 16 at
//     a time, sampled positions and decoded ids leave 16 at a time, a bucket's member row is read by the 16 lanes
//     with one load instruction.
//
//   encode step (codec.cpp:131-137): k = IDX_pop(n - i) (reciprocals from the per-context divisor table);
//        select+remove the k-th alive POSITION: top level (16 x 512 or 16 x 8192 positions, registers), [mid level:
//        16 x 512 positions, u16 in LDS,] leaf: 16 bitmap words of 32 positions (popcount + DPP row scan + in-word
//        select); x = ids[position] (one global load per step; its left neighbour is loaded with it and checks the
//        ascending order the position arithmetic relies on); ID_push(x, P)
//   decode step (codec.cpp:140-152): x = ID_pop(P); rank of x = ids in smaller value buckets (2^B buckets over the
//        top bits of the P-bit universe: three 16-ary prefix levels, top in registers, mid / leaf u16 in LDS) + members
//        of x's bucket below x (one row of global memory, read by the 16 lanes at once); IDX_push(rank, i + 1)
//
// Outside the clean domain (input not strictly ascending, a full bucket row on skewed ids, deep decoder stacks) a list
// is handed back with VIDC_ST_PENDING_SORT / VIDC_ST_RETRY and redone by the wave-per-list kernels.
#pragma once
#include "roc_lane.h"
#include "roc_u2.h"

namespace vidc {
namespace dev {

#define VIDC_GRP_MIN_LIST 65u
#define VIDC_GRP_MAX_LIST 131072u   // 16 top groups x 16 blocks x 512 positions
#define VIDC_GRP_LEV2_MAX 8192u     // two levels: 16 blocks x 512 positions
#define VIDC_GRP_RING 32u           // encoder: pushed words waiting for their 16-word store
#define VIDC_GRP_DWIN 32u           // decoder: window of the stored stream
#define VIDC_GRP_DPST 8u            // decoder: words pushed by the decoder itself

__device__ __forceinline__ uint32_t grp_sub() { return threadIdx.x & 15u; }
// number of lanes of this lane's 16-lane row for which p holds
__device__ __forceinline__ uint32_t grp_count(bool p) {
    const uint64_t m = __ballot(p);
    const uint32_t half = (threadIdx.x & 32u) ? (uint32_t)(m >> 32) : (uint32_t)m;
    return (uint32_t)__popc((half >> (threadIdx.x & 16u)) & 0xffffu);
}
// value of lane `src` (0..15) of this lane's row
__device__ __forceinline__ uint32_t grp_get(uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 48u) | (src & 15u)) << 2), (int)v);
}
// inclusive prefix sum over the row (DPP row_shr 1, 2, 4, 8; lanes without a source add 0)
__device__ __forceinline__ uint32_t grp_scan(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}
// r-th (0-based) set bit of v; r < popcount(v)
__device__ __forceinline__ uint32_t grp_select32(uint32_t v, uint32_t r) {
    uint32_t pos = 0;
    uint32_t c = (uint32_t)__popc(v & 0xffffu);
    if (r >= c) { r -= c; v >>= 16; pos = 16; }
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

// ---------------------------------------------------------------------------------------------------------------
// encoder stack: the state is replicated over the row, lane 0 of the row writes; pushed words wait in a 32-word LDS
// ring and go to the list's arena 16 at a time (one 64-byte piece stored by the 16 lanes)
struct GEStack {
    uint32_t *mem;   // the list's arena
    uint32_t *ring;  // LDS, VIDC_GRP_RING words of this list
    uint32_t sp, sp_mem, cap, draws, err;  // words [sp_mem, sp) are in the ring
    const uint32_t *mt;
    bool w0;         // this lane does the row's LDS / memory writes
};
__device__ __forceinline__ void ls_push(GEStack &s, uint32_t w) {
    if (s.w0) s.ring[s.sp & (VIDC_GRP_RING - 1u)] = w;
    s.sp++;
}
__device__ __forceinline__ uint32_t ls_pop(GEStack &s) {  // codec.h:32-40 (rare in the encoder)
    if (s.sp > s.sp_mem) {
        s.sp--;
        return s.ring[s.sp & (VIDC_GRP_RING - 1u)];
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
__device__ __forceinline__ void ge_drain16(GEStack &s, uint32_t sub) {
    if (s.sp - s.sp_mem >= 16u) {
        if (s.sp_mem + 16u <= s.cap) s.mem[s.sp_mem + sub] = s.ring[(s.sp_mem + sub) & (VIDC_GRP_RING - 1u)];
        else s.err |= 1u;
        s.sp_mem += 16u;
    }
}
__device__ __forceinline__ void ge_flush(GEStack &s, uint32_t sub) {
    for (uint32_t j = s.sp_mem + sub; j < s.sp; j += 16u) {
        if (j < s.cap) s.mem[j] = s.ring[j & (VIDC_GRP_RING - 1u)]; else s.err |= 1u;
    }
    if (s.sp > s.cap) s.err |= 1u;
    s.sp_mem = s.sp;
}

// LDS words of one list of the encoder: bitmap (16 words per 512-position block), mid counters (LEV 3: 16 u16 per top
// group = 8 words), stack ring, position ring
template <int LEV>
__host__ __device__ inline uint32_t roc_grp_enc_lds_words(uint32_t nblk) {
    const uint32_t ntop = (nblk + 15u) >> 4;
    return nblk * 16u + (LEV == 3 ? ntop * 8u : 0u) + VIDC_GRP_RING + 16u;
}

template <int LEV, bool WANT_PERM>
__global__ void __launch_bounds__(64) k_roc_encode_grp(RocEncArgs a, const U2Div *__restrict__ dtab, uint32_t nblk_max) {
    extern __shared__ __align__(16) uint32_t gsm[];
    const uint32_t lane = lane_id(), grp = lane >> 4, sub = lane & 15u;
    const uint32_t wi = blockIdx.x * 4u + grp;
    const bool have = wi < a.nwork;
    const uint32_t l = have ? a.worklist[wi] : 0u;
    const uint64_t off = have ? a.offsets[l] : 0ull;
    const uint32_t n = have ? (uint32_t)(a.offsets[l + 1] - off) : 0u;
    const uint32_t P = have ? a.prec[l] : 0u;  // written by the prepass (the maximum of an ascending list is its last id)
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    const uint32_t stride = roc_grp_enc_lds_words<LEV>(nblk_max);
    uint32_t *bm = gsm + grp * stride;
    uint16_t *mid = (uint16_t *)(bm + nblk_max * 16u);
    uint32_t *ring = bm + nblk_max * 16u + (LEV == 3 ? ((nblk_max + 15u) >> 4) * 8u : 0u);
    uint32_t *pring = ring + VIDC_GRP_RING;

    // alive bitmap over the (ascending) input positions + prefix counters
    const uint32_t nblk = (n + 511u) >> 9;
    for (uint32_t w = sub; w < nblk * 16u; w += 16u) {
        const uint32_t lo_e = w << 5;
        const uint32_t in_w = lo_e >= n ? 0u : (n - lo_e >= 32u ? 32u : n - lo_e);
        bm[w] = in_w == 32u ? ~0u : ((1u << in_w) - 1u);
    }
    uint32_t T;  // lane j: alive positions in top groups 0..j
    if (LEV == 3) {
        const uint32_t ntop = (nblk + 15u) >> 4;
        for (uint32_t e = sub; e < ntop * 16u; e += 16u) {  // entry e: positions below (e + 1) * 512 inside its top group
            const uint32_t t = e >> 4;
            const uint32_t base = t << 13, hi_e = (e + 1u) << 9;
            const uint32_t v = (hi_e < n ? hi_e : n);
            mid[e] = (uint16_t)(v > base ? v - base : 0u);  // (8192 for a full group: fits u16)
        }
        const uint32_t e = (sub + 1u) << 13;
        T = e < n ? e : n;
    } else {
        const uint32_t e = (sub + 1u) << 9;
        T = e < n ? e : n;
    }

    GEStack st;
    {
        const uint64_t ao = have ? arena_at(a, l) : 0ull;
        st.mem = a.arena + ao;
        st.ring = ring;
        st.cap = have ? (uint32_t)(arena_at(a, l + 1) - ao) : 0u;
        st.sp = 0; st.sp_mem = 0; st.draws = 0; st.err = 0; st.mt = a.mt;
        st.w0 = sub == 0u;
    }
    uint64_t head = VIDC_RANS_L;
    const uint64_t *ids = a.ids + off;
    const uint32_t nsteps = wave_max_u32(n);
    bool disorder = false;
    __syncthreads();

    U2Div dv = dtab[n];  // divisor of step 0; the entry of step i + 1 is loaded during step i
    for (uint32_t i = 0; i < nsteps; i++) {
        const bool act = i < n;
        const uint32_t d = act ? n - i : 1u;
        const U2Div dv_next = dtab[d > 1u ? d - 1u : 1u];
        if (act) {
            // ---- k = IDX_pop(n - i), codec.cpp:21-42
            uint64_t h0 = head;
            if ((uint32_t)(h0 >> 32) >= dv.w * d) {  // h0 >= nmax * ((L / nmax) << 32)
                ls_push(st, (uint32_t)h0);
                h0 >>= 32;
            }
            uint64_t q = __umul64hi(h0, ((uint64_t)dv.y << 32) | dv.x);
            uint32_t k = (uint32_t)h0 - (uint32_t)q * d;
            if (k >= d) { k -= d; q++; }
            if (__builtin_expect(l_lt_2p31(h0), 0)) q = (uint64_t)ls_pop(st) | (q << 32);  // test on h0 (codec.cpp:35)
            head = q;

            // ---- select + remove the k-th alive position
            uint32_t blk;
            {
                const uint32_t c = grp_count(T <= k);          // top group / block holding it
                const uint32_t prev = grp_get(T, c - 1u);
                k -= c ? prev : 0u;
                T -= sub >= c ? 1u : 0u;
                blk = c;
            }
            if (LEV == 3) {
                const uint32_t mv = mid[blk * 16u + sub];
                const uint32_t c = grp_count(mv <= k);
                const uint32_t prev = grp_get(mv, c - 1u);
                k -= c ? prev : 0u;
                if (sub >= c) mid[blk * 16u + sub] = (uint16_t)(mv - 1u);
                blk = blk * 16u + c;
            }
            const uint32_t wv = bm[blk * 16u + sub];
            const uint32_t incl = grp_scan((uint32_t)__popc(wv));
            const uint32_t c = grp_count(incl <= k);            // word of the block
            const uint32_t prev = grp_get(incl, c - 1u);
            const uint32_t word = grp_get(wv, c);
            k -= c ? prev : 0u;
            const uint32_t b = grp_select32(word, k);
            if (sub == c) bm[blk * 16u + sub] = wv & ~(1u << b);
            const uint32_t pos = ((blk * 16u + c) << 5) + b;
            // the sampled id and its left neighbour (same 64-byte line seven times out of eight): every position is sampled
            // exactly once, so the list is checked to be strictly ascending -- what select over POSITIONS relies on -- and,
            // with its last id below 2^31, to lie inside the domain
            // (one 16-byte load for both: a list of these kernels has at least VIDC_GRP_MIN_LIST ids, so ids[1] exists when pos is 0)
            struct __attribute__((aligned(8))) IdPair { uint64_t a, b; };
            const IdPair pr = *(const IdPair *)(ids + (pos ? pos - 1u : 0u));
            const uint64_t xid = pos ? pr.b : pr.a;
            const uint64_t xprev = pos ? pr.a : 0ull;
            disorder |= (pos && xprev >= xid) || (xid >> 31) != 0ull;
            const uint32_t x = (uint32_t)xid;
            if (WANT_PERM && st.w0) pring[i & 15u] = pos;

            // ---- ID_push(x, P), codec.cpp:92-105
            l_u_push(head, st, x & 0xffffu, p0);
            l_u_push(head, st, x >> 16, p1);
            if (__builtin_expect((uint32_t)(head >> 63) != 0u, 0)) {  // slices 2, 3: precision 0, symbol 0
                l_u_push(head, st, 0u, 0u);
                l_u_push(head, st, 0u, 0u);
            }
            ge_drain16(st, sub);  // at most 5 words per step: the ring (32) never overflows
        }
        dv = dv_next;
        if (WANT_PERM && ((i & 15u) == 15u || i + 1u == nsteps)) {  // uniform
            const uint32_t s0 = i & ~15u;
            if (s0 + sub <= i && s0 + sub < n) a.perm[off + s0 + sub] = pring[sub];
        }
    }
    ge_flush(st, sub);
    if (have && sub == 0u) {
        a.heads[l] = head;
        a.nwords[l] = st.sp;
        a.draws[l] = st.draws;
        a.status[l] = disorder ? VIDC_ST_PENDING_SORT
                               : (st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// decoder stack: the stored stream is consumed from its top through a 32-word LDS window refilled 16 words (one aligned
// 64-byte piece, a word per lane) at a time a step or more before the words are needed; the words the decoder pushes
// itself wait in an 8-deep LDS stack (deeper -> VIDC_ST_RETRY)
struct GWStack {
    const uint32_t *wbase;  // stream of the list minus `al` words: 64-byte aligned
    uint32_t *ring;         // LDS: shifted index s at ring[s & 31]
    uint32_t *pst;          // LDS: pushed word e at pst[e]
    uint32_t al, otop, wlo, d, draws, err;
    const uint32_t *mt;
    bool w0;
};
__device__ __forceinline__ void ls_push(GWStack &s, uint32_t w) {
    if (s.d < VIDC_GRP_DPST) { if (s.w0) s.pst[s.d] = w; s.d++; } else s.err |= 4u;
}
__device__ __forceinline__ uint32_t ls_pop(GWStack &s) {  // codec.h:32-40
    if (s.d) {
        s.d--;
        return s.pst[s.d];
    }
    if (__builtin_expect(s.otop == s.al, 0)) {
        uint32_t w = 0;
        if (s.draws < VIDC_MT_TABLE) w = s.mt[s.draws]; else s.err |= 2u;
        s.draws++;
        return w;
    }
    s.otop--;
    return s.otop >= s.wlo ? s.ring[s.otop & (VIDC_GRP_DWIN - 1u)] : s.wbase[s.otop];
}
__device__ __forceinline__ void gw_init(GWStack &s, bool have, const uint32_t *words, uint64_t word_off, uint32_t W, uint32_t sub) {
    s.al = (uint32_t)word_off & 15u;
    s.wbase = words + (word_off - s.al);
    s.otop = W + s.al;
    s.wlo = s.otop > VIDC_GRP_DWIN ? ((s.otop - VIDC_GRP_DWIN + 15u) & ~15u) : 0u;
    s.d = 0; s.err = 0;
#pragma unroll
    for (uint32_t k = 0; k < VIDC_GRP_DWIN / 16u; k++) {
        const uint32_t q = s.wlo + 16u * k;  // (the last piece may read up to 15 words past the stream: padded allocation)
        if (have && q < s.otop) s.ring[(q + sub) & (VIDC_GRP_DWIN - 1u)] = s.wbase[q + sub];
    }
}
__device__ __forceinline__ bool gw_issue(const GWStack &s, uint32_t sub, uint32_t &pf) {
    const bool go = s.wlo >= 16u && s.otop <= s.wlo + (VIDC_GRP_DWIN - 16u);
    if (go) pf = s.wbase[s.wlo - 16u + sub];
    return go;
}
__device__ __forceinline__ void gw_land(GWStack &s, bool go, uint32_t sub, uint32_t pf) {
    if (go) {
        s.wlo -= 16u;
        s.ring[(s.wlo + sub) & (VIDC_GRP_DWIN - 1u)] = pf;
    }
}

// bucket geometry of the decoder: 2^B value buckets, B = 8 + F (top 4 bits: registers, next 4: `mid`, last F: `leaf`)
__host__ __device__ inline uint32_t roc_grp_dec_fbits(uint32_t n) {  // <= 8 members per bucket on average up to 32 768 ids
    return n <= 2048u ? 0u : (n <= 8192u ? 2u : (n <= 16384u ? 3u : 4u));
}
__host__ __device__ inline uint32_t roc_grp_dec_cap(uint32_t n) { return n <= 32768u ? 32u : (n <= 65536u ? 64u : 96u); }
__host__ __device__ inline uint32_t roc_grp_dec_lds_words(uint32_t F) {
    return 128u + (F ? (128u << F) : 0u) + VIDC_GRP_DWIN + VIDC_GRP_DPST;  // mid 256 u16 | leaf 256 << F u16 | window | pst
}
// member rows of one list: 2^(8 + F) rows of `cap` u32
__host__ __device__ inline uint64_t roc_grp_dec_slots(uint32_t n) {
    return ((uint64_t)256u << roc_grp_dec_fbits(n)) * roc_grp_dec_cap(n);
}

template <int F>
__global__ void __launch_bounds__(64) k_roc_decode_grp(RocDecArgs a, const U2Div *__restrict__ dtab) {
    extern __shared__ __align__(16) uint32_t gsm[];
    constexpr uint32_t B = 8u + (uint32_t)F;
    const uint32_t lane = lane_id(), grp = lane >> 4, sub = lane & 15u;
    const uint32_t wi = blockIdx.x * 4u + grp;
    const bool have = wi < a.nwork;
    const uint32_t l = have ? a.worklist[wi] : 0u;
    const uint32_t n = have ? (uint32_t)(a.offsets[l + 1] - a.offsets[l]) : 0u;
    const uint64_t ooff = have ? (a.out_off ? a.out_off[wi] : a.offsets[l]) : 0ull;
    const uint32_t P = have ? a.prec[l] : 0u;
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    const uint32_t bsh = P > B ? P - B : 0u;
    const uint32_t cap = roc_grp_dec_cap(n);
    uint32_t *base = gsm + grp * roc_grp_dec_lds_words(F);
    uint16_t *mid = (uint16_t *)base;
    uint16_t *leaf = (uint16_t *)(base + 128u);
    uint32_t *win = base + 128u + (F ? (128u << F) : 0u);
    uint32_t *pst = win + VIDC_GRP_DWIN;
    for (uint32_t e = sub; e < 128u + (F ? (128u << F) : 0u); e += 16u) base[e] = 0u;
    uint32_t P0 = 0;  // lane j: decoded ids in top groups 0..j
    // The ids of the current block of 16 steps stay in registers (lane j: step i0 + j -- value, bucket, slot in the bucket's
    // row) and go to their rows, and to the output, at the end of the block: a store per step kept every step waiting for its
    // write acknowledgement (the loop's s_waitcnt vmcnt(0) counts stores too on gfx9: 1.9 us per step instead of 1.1).  The
    // rank adds the block's members of x's bucket from these registers; memory rows are read up to their size at the start
    // of the block.
    uint32_t rx = 0, rb = 0, rs = 0;

    GWStack st;
    st.ring = win; st.pst = pst; st.w0 = sub == 0u;
    gw_init(st, have, a.words, have ? a.word_off[l] : 0ull, have ? a.nwords[l] : 0u, sub);
    st.mt = a.mt;
    st.draws = have ? a.draws[l] : 0u;
    const uint32_t draws0 = st.draws;
    uint64_t head = have ? a.heads[l] : VIDC_RANS_L;
    uint32_t *slots = a.slots + (have ? a.slots_off[wi] : 0ull);
    uint32_t n_eff = n;
    bool retry = false;
    const uint32_t nsteps = wave_max_u32(n);
    __syncthreads();

    for (uint32_t i = 0; i < nsteps; i++) {
        const uint32_t lq = dtab[i + 1u].w;  // uniform: floor(2^31 / (i + 1))
        uint32_t pf = 0;
        const bool pf_go = i < n_eff && gw_issue(st, sub, pf);
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
            const uint32_t b = (bsh >= 32u ? 0u : (x >> bsh)) & ((1u << B) - 1u);
            const uint32_t t = b >> (B - 4u), tm = b >> F, m = tm & 15u, f = b & ((1u << F) - 1u);
            uint32_t *row = slots + (size_t)b * cap;
            const uint32_t y0 = row[sub];  // first 16 members (whatever the row holds beyond its count is ignored)
            const uint32_t prev0 = grp_get(P0, t - 1u);
            const uint32_t mown = mid[t * 16u + sub];
            const uint32_t mprev = m ? (uint32_t)mid[tm - 1u] : 0u;
            uint32_t r = (t ? prev0 : 0u) + mprev;
            uint32_t cb;
            uint32_t lown = 0;
            if (F) {
                lown = sub < (1u << F) ? (uint32_t)leaf[(tm << F) + sub] : 0u;
                const uint32_t lprev = f ? (uint32_t)leaf[b - 1u] : 0u;
                cb = (uint32_t)leaf[b] - lprev;
                r += lprev;
            } else {
                cb = (uint32_t)mid[tm] - mprev;
            }
            const uint32_t ib = i & 15u;
            const bool rsame = sub < ib && rb == b;         // members of this bucket decoded in the current block
            const uint32_t cmem = cb - grp_count(rsame);   // members the row holds in memory
            r += grp_count(sub < cmem && y0 < x) + grp_count(rsame && rx < x);
            for (uint32_t c = 16u; c < cmem; c += 16u) {
                const uint32_t y = row[c + sub];
                r += grp_count(c + sub < cmem && y < x);
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
                if (sub == ib) { rx = x; rb = b; rs = cb; }
                P0 += sub >= t ? 1u : 0u;
                if (sub >= m) mid[t * 16u + sub] = (uint16_t)(mown + 1u);
                if (F && sub >= f && sub < (1u << F)) leaf[(tm << F) + sub] = (uint16_t)(lown + 1u);
            }
        }
        gw_land(st, pf_go, sub, pf);
        if ((i & 15u) == 15u || i + 1u == nsteps) {  // uniform: the ids of steps s0 .. i go to their rows and to the output (one 128-byte piece)
            const uint32_t sstep = (i & ~15u) + sub;
            // (lists handed back for a retry are rewritten as a whole: whatever they store here is harmless)
            if (sstep <= i && sstep < n) {
                slots[(size_t)rb * cap + rs] = rx;
                a.out[ooff + (n - 1u - sstep)] = (uint64_t)rx;
            }
        }
    }
    if (have && sub == 0u) {
        retry |= (st.err & 4u) != 0u;
        const bool clean = (head == VIDC_RANS_L) && (st.otop - st.al + st.d == st.draws - draws0);
        a.end_state[l] = (clean || retry) ? 0u : 1u;
        a.status[l] = retry ? VIDC_ST_RETRY : ((st.err & 2u) ? VIDC_ST_MT : VIDC_ST_OK);
    }
}

}  // namespace dev
}  // namespace vidc
