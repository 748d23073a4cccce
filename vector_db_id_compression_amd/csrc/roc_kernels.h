// roc_kernels.h -- Random Order Coding (bits-back ANS over a set) encode / decode kernels for gfx950.
//
// One inverted list (or one graph row) per 64-lane wavefront; lists are strictly serial inside
// (every step depends on the 64-bit ANS head) and independent across wavefronts.
//
//   encode step i (custom_invlists_impl.cpp:178-192 == codec.cpp:131-137):
//        k = IDX_pop(n - i)            u64 div/mod by a runtime divisor -> mulhi with per-lane
//                                      precomputed reciprocals (one 64-bit division per lane per 64 steps)
//        x = select+remove k-th alive  alive bitmap over the SORTED positions; per-lane inclusive
//                                      prefix counters (ballot + s_ff1) -> LDS row of prefix counters ->
//                                      in-word select with v_mbcnt
//        ID_push(x, P)                 four 16-bit uniform slices
//   decode step i (codec.cpp:140-152):
//        x = ID_pop(P); r = #decoded elements < x; IDX_push(r, i + 1); out[n-1-i] = x
//                                      rank = per-lane coarse prefix counters + LDS fine prefix row +
//                                      compare against the (<= 16) members of x's fine bucket
//
// The order-statistic structures replace the reference's pointer BST (fenwick_tree.h:20-167);
// the bitstream only depends on rank/select over the set.
#pragma once
#include "wave.h"

namespace vidc {
namespace dev {

// status codes written per list
#define VIDC_ST_OK 0u
#define VIDC_ST_PENDING_SORT 1u
#define VIDC_ST_DOMAIN 2u
#define VIDC_ST_OVERFLOW 3u
#define VIDC_ST_MT 4u

struct RocEncArgs {
    const uint64_t *ids;       // [ntotal] (IVF flavour) or nullptr
    const int32_t *rows;       // [N*K]   (graph flavour) or nullptr
    uint32_t K;
    const uint64_t *offsets;   // [nlist+1] CSR offsets of the OUTPUT numbering (IVF: == input offsets)
    const uint32_t *worklist;  // list numbers handled by this launch
    uint32_t nwork;
    int precision_mode;
    uint64_t *heads;           // [nlist]
    uint32_t *prec;            // [nlist]
    uint32_t *nwords;          // [nlist]
    uint32_t *draws;           // [nlist]
    uint32_t *sizes;           // [nlist] (graph flavour: edge counts, altid_impl.h:61)
    uint32_t *status;          // [nlist]
    uint32_t *arena;           // worst-case word arena; list l owns [arena_at(l), arena_at(l + 1))
    uint32_t arena_stride;     // graph flavour: words per row
    uint32_t *perm;            // [ntotal] or nullptr
    uint32_t *sid;             // [ntotal] scratch: ids of each list in ascending order (u32)
    uint32_t *spos;            // [ntotal] scratch: input positions in that order (unsorted lists only)
    uint64_t *skey;            // sort scratch for unsorted lists (pow2-padded) or nullptr
    const uint64_t *skey_off;  // [nlist+1] or nullptr
    const uint32_t *mt;
    uint32_t lpw;              // lane-per-list kernels: lists per wavefront (0 = 64); the remaining lanes idle
};

// Worst-case arena layout in closed form (no per-list offset array to build, upload or read): a list of n ids
// needs at most n*37/32 + 8 words (<= P + 4 bits of growth per step, P <= 32).
//   graph rows : l * stride
//   IVF lists  : floor(offsets[l] * 37 / 32) + 9 * l   (consecutive differences >= floor(n*37/32) + 8)
__host__ __device__ inline uint64_t roc_arena_at(const uint64_t *offsets, uint32_t stride, uint64_t l) {
    return offsets ? ((offsets[l] * 37ull) >> 5) + 9ull * l : l * (uint64_t)stride;
}
// capacity of a decoder's private stack scratch (plan_decode allots exactly this): the decoder pushes at most
// ~log2(n) bits per step, so streams of the encoder never outgrow the encoder's own arena bound; a stream whose
// precision is far below log2(n) (reference quirk domain) GROWS while it is decoded and needs that room
__host__ __device__ inline uint32_t roc_dec_stack_cap(uint32_t n, uint32_t W) {
    const uint32_t a = (uint32_t)(((uint64_t)n * 37ull) >> 5) + 8u;
    return (a > W ? a : W) + 64u;
}
__device__ __forceinline__ uint64_t arena_at(const RocEncArgs &a, uint64_t l) {
    return roc_arena_at(a.rows ? nullptr : a.offsets, a.arena_stride, l);
}

struct RocDecArgs {
    const uint64_t *offsets;    // [nlist+1] CSR offsets of the stored lists
    const uint32_t *worklist;   // list numbers handled by this launch
    const uint64_t *out_off;    // [nwork] output offset per work item (or nullptr: use offsets[list])
    uint32_t nwork;
    const uint64_t *heads;
    const uint32_t *prec;
    const uint32_t *nwords;
    const uint32_t *draws;
    const uint32_t *words;      // compact stream
    const uint64_t *word_off;   // [nlist+1]
    uint64_t *out;              // u64 output (IVF flavour) or nullptr
    int32_t *out_rows;          // int32 rows (graph flavour) or nullptr; row stride K, -1 padded
    uint32_t K;
    uint32_t *scratch_words;    // decoder stack re-spill scratch
    const uint64_t *scratch_off;// [nwork] per work item (relative to work_base)
    uint32_t *slots;            // fine-bucket member storage
    const uint64_t *slots_off;  // [nwork] per work item: start of (NF*CAP + n) words
    uint32_t *end_state;        // [nlist] 0 = clean end state (head == 2^31 and stack == drawn words)
    uint32_t *status;           // [nlist]
    const uint32_t *mt;
    uint32_t lpw;               // lane-per-list kernels: lists per wavefront (0 = 64)
    uint32_t row_align;         // bucket-row lane decoders: rows are aligned to / padded to this many slots (0 / 4 or 16: the plan's)
    uint32_t out_by_list;       // graph flavour: the work list is an ORDER of the rows (k_rows_order_*): row l goes to output row l, not to its item number
};

#define VIDC_DEC_CAP 16u       // members per fine bucket before spilling to the overflow list (lists <= 32768)
#define VIDC_DEC_CAP_BIG 64u   // same for longer lists (average bucket load up to 64)
__host__ __device__ inline uint32_t roc_dec_cap(uint32_t n) { return n > 32768u ? VIDC_DEC_CAP_BIG : VIDC_DEC_CAP; }
#define VIDC_DEC_MAX_FB 12u    // <= 64 coarse x 64 fine buckets

// fine-bucket bits used by the decoder for a list of n elements with precision P (host + device)
__host__ __device__ inline uint32_t roc_dec_fine_bits(uint32_t n, uint32_t P) {
    uint32_t lg = 0;
    while ((1u << lg) < n) lg++;
    uint32_t fb = lg > 3u ? lg - 3u : 0u;
    if (fb > VIDC_DEC_MAX_FB) fb = VIDC_DEC_MAX_FB;
    if (fb > P) fb = P;
    return fb;
}

// ---------------------------------------------------------------------------------------------
// Per-64-step reciprocals: lane t prepares the divisor d = dmax - t.
struct Recip {
    uint32_t m_lo, m_hi;  // floor((2^64 - 1) / d)
    uint32_t thr;         // d * floor(2^31 / d)
    uint32_t lq;          // floor(2^31 / d)
    uint32_t d;           // the divisor itself
};
__device__ __forceinline__ void recip_block(Recip &r, uint32_t dmax) {
    uint32_t t = lane_id();
    uint32_t d = dmax > t ? dmax - t : 1u;
    uint64_t m = ~0ull / (uint64_t)d;
    r.m_lo = (uint32_t)m;
    r.m_hi = (uint32_t)(m >> 32);
    r.lq = 0x80000000u / d;
    r.thr = r.lq * d;
    r.d = d;
}

// =============================================================================================
// TINY lists (n <= 64): the whole set lives in one VGPR, sorted with an in-register bitonic network.
// =============================================================================================
template <bool ROWS>
__global__ void __launch_bounds__(64) k_roc_encode_tiny(RocEncArgs a) {
    const uint32_t lane = lane_id();
    for (uint32_t wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
        const uint32_t l = a.worklist ? a.worklist[wi] : wi;
        uint32_t n;
        uint64_t off;
        uint32_t v = 0xffffffffu;
        bool bad = false;
        if (ROWS) {
            // altid_impl.cpp:110-117: edges up to the first -1
            int32_t e = lane < a.K ? a.rows[(uint64_t)l * a.K + lane] : -1;
            uint64_t neg = ballot(e == -1);
            n = neg ? ff1(neg) : 64u;
            if (n > a.K) n = a.K;
            off = (uint64_t)l * a.K;
            if (lane < n) {
                v = (uint32_t)e;
                bad = e < 0;
            }
        } else {
            off = a.offsets[l];
            n = (uint32_t)(a.offsets[l + 1] - off);
            if (lane < n) {
                uint64_t id = a.ids[off + lane];
                bad = id >= (1ull << 31);  // reference: int max_id (custom_invlists_impl.cpp:163)
                v = (uint32_t)id;
            }
        }
        if (ROWS) a.sizes[l] = n;
        if (n == 0) {
            a.heads[l] = VIDC_RANS_L; a.prec[l] = 0; a.nwords[l] = 0; a.draws[l] = 0; a.status[l] = VIDC_ST_OK;
            continue;
        }
        if (ballot(bad)) {
            if (lane == 0) a.status[l] = VIDC_ST_DOMAIN;
            continue;
        }
        const uint32_t maxid = wave_max_u32(lane < n ? v : 0u);
        const uint32_t P = precision_for(maxid, a.precision_mode);
        const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;

        // sort (id, position) ascending: key = id * 64 + position, padding lanes carry ~0
        uint64_t key = lane < n ? (((uint64_t)v << 6) | lane) : ~0ull;
#pragma unroll
        for (uint32_t k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                uint32_t plo = (uint32_t)__shfl_xor((int)(uint32_t)key, (int)j, 64);
                uint32_t phi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), (int)j, 64);
                uint64_t other = ((uint64_t)phi << 32) | plo;
                bool up = ((lane & k) == 0);
                bool lower = ((lane & j) == 0);
                bool take_min = (up == lower);
                uint64_t mn = key < other ? key : other, mx = key < other ? other : key;
                key = take_min ? mn : mx;
            }
        }
        const uint32_t sid = (uint32_t)(key >> 6);
        const uint32_t spos = (uint32_t)key & 63u;

        WStack st;
        const uint64_t ao = arena_at(a, l);
        uint32_t *arena = a.arena + ao;
        ws_init_empty(st, arena, (uint32_t)(arena_at(a, l + 1) - ao), a.mt, VIDC_MT_TABLE);
        uint64_t head = VIDC_RANS_L;
        uint64_t alive = n == 64u ? ~0ull : ((1ull << n) - 1ull);
        Recip rc;
        recip_block(rc, n);
        uint32_t pbuf = 0;
        for (uint32_t i = 0; i < n; i++) {
            ws_prepare(st);
            const uint32_t nmax = n - i;
            const uint64_t magic = rl64(rc.m_lo, rc.m_hi, i);
            const uint32_t k = ans_idx_pop(head, st, nmax, rl(rc.thr, i), magic);
            const bool mine = ((alive >> lane) & 1ull) && (mbcnt(alive) == k);
            const uint32_t b = ff1(ballot(mine));
            alive &= ~(1ull << b);
            const uint32_t x = rl(sid, b);
            ans_id_push(head, st, x, p0, p1);
            pbuf = wl(rl(spos, b), i, pbuf);
        }
        ws_flush(st);
        if (!ROWS && a.perm && lane < n) a.perm[off + lane] = pbuf;
        if (lane == 0) {
            a.heads[l] = head;
            a.prec[l] = P;
            a.nwords[l] = st.sp;
            a.draws[l] = st.draws;
            a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
        }
    }
}

template <bool ROWS>
__global__ void __launch_bounds__(64) k_roc_decode_tiny(RocDecArgs a) {
    const uint32_t lane = lane_id();
    for (uint32_t wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
        const uint32_t l = a.worklist[wi];
        const uint32_t n = (uint32_t)(a.offsets[l + 1] - a.offsets[l]);
        const uint64_t ooff = a.out_off ? a.out_off[wi] : (ROWS ? (uint64_t)wi * a.K : a.offsets[l]);
        if (ROWS) {
            // pad the row with -1 first (the reference leaves slots >= n untouched, altid_impl.cpp:153-165)
            if (lane < a.K) a.out_rows[ooff + lane] = -1;
        }
        if (n == 0) {
            if (lane == 0) { a.end_state[l] = 0; a.status[l] = VIDC_ST_OK; }
            continue;
        }
        const uint32_t P = a.prec[l];
        const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
        const uint32_t W = a.nwords[l];
        WStack st;
        ws_init_loaded(st, a.words + a.word_off[l], W, a.scratch_words + a.scratch_off[wi], roc_dec_stack_cap(n, W), a.draws[l], a.mt,
                       VIDC_MT_TABLE);
        const uint32_t draws0 = st.draws;
        uint64_t head = a.heads[l];
        Recip rc;
        // decoder divisors grow: nmax = i + 1; lane t prepares d = t + 1  (recip_block counts down from dmax)
        {
            uint32_t d = lane + 1u;
            rc.lq = 0x80000000u / d;
            rc.thr = 0; rc.m_lo = 0; rc.m_hi = 0;
        }
        uint32_t val = 0xffffffffu;  // lane i holds the element decoded at step i
        for (uint32_t i = 0; i < n; i++) {
            ws_prepare(st);
            const uint32_t x = ans_id_pop(head, st, p0, p1);
            const uint32_t r = popc64(ballot(lane < i && val < x));  // strictly smaller (fenwick_tree.h:42-94)
            ans_idx_push(head, st, r, i + 1u, rl(rc.lq, i));
            val = wl(x, i, val);
        }
        if (lane < n) {
            if (ROWS) a.out_rows[ooff + (n - 1u - lane)] = (int32_t)val;
            else a.out[ooff + (n - 1u - lane)] = (uint64_t)val;
        }
        if (lane == 0) {
            // clean end: head back at 2^31 and the stack holds exactly the words drawn from mt19937
            bool clean = (head == VIDC_RANS_L) && (st.sp == st.draws - draws0);
            a.end_state[l] = clean ? 0u : 1u;
            a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
        }
    }
}

// =============================================================================================
// GENERAL lists (n <= 262144)
// =============================================================================================

// wave-level bitonic sort of npow2 u64 keys in global memory (only taken for unsorted input lists)
__device__ inline void wave_bitonic_global(uint64_t *keys, uint32_t npow2) {
    const uint32_t lane = lane_id();
    for (uint32_t k = 2; k <= npow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (npow2 >> 1); t += 64) {
                uint32_t i = ((t & ~(j - 1u)) << 1) | (t & (j - 1u));  // index with bit j clear
                uint32_t p = i | j;
                uint64_t x = keys[i], y = keys[p];
                bool up = ((i & k) == 0);
                if ((x > y) == up) {
                    keys[i] = y;
                    keys[p] = x;
                }
            }
            wave_sync();
        }
    }
}

// LDS layout (dynamic): u64 words[64*RL] | u32 rowpref[64*RL]
__global__ void __launch_bounds__(64) k_roc_encode_gen(RocEncArgs a, uint32_t rl_max) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *words = (uint64_t *)smem;
    uint32_t *rowpref = (uint32_t *)(smem + (size_t)64u * rl_max * 8u);
    const uint32_t lane = lane_id();
    if (a.lpw >> 31) __builtin_amdgcn_s_setprio(3);  // (VIDC_ENC_PRIO, roc.hip)

    for (uint32_t wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
        // per-list scalars arrive through vector loads: pin them to SGPRs so the ANS chain stays scalar
        const uint32_t l = rfl(a.worklist[wi]);
        const uint64_t off = rfl64(a.offsets[l]);
        const uint32_t n = rfl((uint32_t)(a.offsets[l + 1] - off));
        if (n == 0) {
            if (lane == 0) {
                a.heads[l] = VIDC_RANS_L; a.prec[l] = 0; a.nwords[l] = 0; a.draws[l] = 0; a.status[l] = VIDC_ST_OK;
            }
            continue;
        }
        // ---- phase 0: stream the list once: domain check, max id, sortedness.  An ascending list (the usual case:
        // Faiss lists are in add order) is sampled straight from the input; only a list that has to be sorted gets a
        // u32 copy in `sid` (that copy used to be written and read back for every list: 4 + 4 bytes per id)
        uint32_t mx = 0;
        bool bad = false, unsorted = false;
        for (uint32_t j = lane; j < n; j += 64) {
            uint64_t id = a.ids[off + j];
            bad |= id >= (1ull << 31);
            if (j) unsorted |= a.ids[off + j - 1] >= id;
            uint32_t v = (uint32_t)id;
            mx = v > mx ? v : mx;
        }
        if (ballot(bad)) {
            if (lane == 0) a.status[l] = VIDC_ST_DOMAIN;
            continue;
        }
        const bool need_sort = ballot(unsorted) != 0;
        if (need_sort) {
            if (!a.skey) {
                if (lane == 0) a.status[l] = VIDC_ST_PENDING_SORT;
                continue;
            }
            uint64_t *keys = a.skey + a.skey_off[l];
            const uint32_t np2 = (uint32_t)(a.skey_off[l + 1] - a.skey_off[l]);
            for (uint32_t j = lane; j < np2; j += 64)
                keys[j] = j < n ? ((a.ids[off + j] << 32) | j) : ~0ull;  // order (id, position), cf. tuple<id, code ptr>
            wave_sync();
            wave_bitonic_global(keys, np2);
            for (uint32_t j = lane; j < n; j += 64) {
                uint64_t kk = keys[j];
                a.sid[off + j] = (uint32_t)(kk >> 32);
                a.spos[off + j] = (uint32_t)kk;
            }
        }
        const uint32_t maxid = wave_max_u32(mx);
        const uint32_t P = precision_for(maxid, a.precision_mode);
        const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;

        // ---- phase 1: alive bitmap over sorted positions + two levels of inclusive prefix counters
        const uint32_t nw = (n + 63u) >> 6;
        uint32_t RL = 1;
        while (64u * RL < nw) RL <<= 1;  // words per lane-block
        const uint32_t rlsh = (uint32_t)__builtin_ctz(RL);
        for (uint32_t w = lane; w < 64u * RL; w += 64) {
            uint32_t lo_e = w << 6, hi_e = lo_e + 64u;
            uint32_t c = w >> rlsh;
            uint32_t blk_lo = (c << rlsh) << 6;
            uint32_t cnt_hi = hi_e < n ? hi_e : n;
            uint32_t cnt_blk = blk_lo < n ? blk_lo : n;
            rowpref[w] = cnt_hi - cnt_blk;
            uint32_t in_w = lo_e >= n ? 0u : (n - lo_e >= 64u ? 64u : n - lo_e);
            words[w] = in_w == 64u ? ~0ull : ((1ull << in_w) - 1ull);
        }
        uint32_t P1;  // lane c: alive elements in lane-blocks 0..c
        {
            uint64_t e = ((uint64_t)(lane + 1u) << rlsh) << 6;
            P1 = e < n ? (uint32_t)e : n;
        }
        wave_sync();

        // ---- phase 2: the serial chain
        WStack st;
        {
            const uint64_t ao = rfl64(arena_at(a, l));
            ws_init_empty(st, a.arena + ao, rfl((uint32_t)(arena_at(a, l + 1) - ao)), a.mt, VIDC_MT_TABLE);
        }
        uint64_t head = VIDC_RANS_L;
        Recip rc;
        uint32_t pbuf = 0;
        // id of sorted position j: sid[j] after a sort, the low dword of ids[j] otherwise
        const uint32_t *sid = need_sort ? a.sid + off : (const uint32_t *)(a.ids + off);
        const uint32_t *spos = a.spos + off;
        const bool want_perm = a.perm != nullptr;
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            recip_block(rc, n - i0);  // lane t owns the divisor of step i0 + t
            const uint32_t steps = n - i0 < 64u ? n - i0 : 64u;
            for (uint32_t t64 = 0; t64 < steps; t64++) {
                ws_prepare(st);
                uint32_t k = ans_idx_pop_v(head, st, n - i0 - t64, rl(rc.thr, t64) - 1u, rc.d, rc.m_lo, rc.m_hi, t64);
                // level 1: lane-block
                const uint32_t c = ff1(ballot(P1 > k));
                const uint32_t prev1 = rl(P1, (c - 1u) & 63u);
                k -= c ? prev1 : 0u;
                // level 2: word inside the lane-block
                uint32_t w = c;
                uint32_t tsel = 0;
                uint32_t rowv = 0;
                if (RL > 1u) {
                    rowv = lane < RL ? rowpref[(c << rlsh) + lane] : 0xffffffffu;
                    tsel = ff1(ballot(rowv > k));
                    const uint32_t prev2 = rl(rowv, (tsel - 1u) & 63u);
                    k -= tsel ? prev2 : 0u;
                    w = (c << rlsh) + tsel;
                }
                // level 3: bit inside the word
                const uint64_t W = rfl64(words[w]);
                const uint32_t b = ff1(ballot(mbcnt(W) == k) & W);
                const uint32_t j = (w << 6) + b;
                // the only global load on the chain: issued first, consumed last.  An ascending list is read from the
                // input array (never written by this kernel): a scalar load (the address is wave-uniform), no v_readfirstlane
                uint32_t xv = 0, xs = 0;
                if (need_sort) xv = sid[j];
                else asm volatile("s_load_dword %0, %1, %2" : "=s"(xs) : "s"(sid), "s"(j << 3) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                // remove (every lane stores the same word): runs in the shadow of the load
                words[w] = W & ~(1ull << b);
                if (RL > 1u && lane >= tsel && lane < RL) rowpref[(c << rlsh) + lane] = rowv - 1u;
                P1 -= (lane >= c) ? 1u : 0u;
                __builtin_amdgcn_sched_barrier(0);
                if (!need_sort) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(xs) : : "memory");
                const uint32_t x = need_sort ? rfl(xv) : xs;
                ans_id_push(head, st, x, p0, p1);
                if (want_perm) {
                    const uint32_t pos = need_sort ? rfl(spos[j]) : j;
                    pbuf = wl(pos, t64, pbuf);
                }
            }
            if (want_perm) {
                if (lane < steps) a.perm[off + i0 + lane] = pbuf;
            }
        }
        ws_flush(st);
        if (lane == 0) {
            a.heads[l] = head;
            a.prec[l] = P;
            a.nwords[l] = st.sp;
            a.draws[l] = st.draws;
            a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
        }
        wave_sync();
    }
}

// LDS layout (dynamic): u32 rowpref[2^fb_max]
// CT = counter type of the fine prefix rows: uint16_t for lists up to 65 536 ids (a counter is only read while
// fewer than n ids are decoded, so it never exceeds 65 535 when it matters; the wrap at the last insert is never
// read), uint32_t beyond.  Halves the LDS footprint: with 16 KiB per list the two deepest classes of S2 (7 + 10
// lists per CU) did not fit a CU's LDS together and the second one ran at a third of its occupancy.
// LROWS: the member rows of the fine buckets live in LDS behind the prefix rows (lists up to 4096 ids: 512 buckets
// x 16 members = 32 KiB) instead of global memory -- no scattered 4-byte stores (each one cost a 32-byte
// write-back: 11.5 MB written per S1 step for 0.7 M ids) and no global load on the chain; only the overflow list of
// a skewed bucket stays in memory.
template <typename CT, bool LROWS = false>
__global__ void __launch_bounds__(64) k_roc_decode_gen(RocDecArgs a, uint32_t lds_entries, uint32_t cap) {
    extern __shared__ __align__(16) unsigned char smem[];
    CT *rowpref = (CT *)smem;
    const uint32_t lane = lane_id();

    for (uint32_t wi = blockIdx.x; wi < a.nwork; wi += gridDim.x) {
        const uint32_t l = rfl(a.worklist[wi]);
        const uint32_t n = rfl((uint32_t)(a.offsets[l + 1] - a.offsets[l]));
        const uint64_t ooff = rfl64(a.out_off ? a.out_off[wi] : a.offsets[l]);
        if (n == 0) {
            if (lane == 0) { a.end_state[l] = 0; a.status[l] = VIDC_ST_OK; }
            continue;
        }
        const uint32_t P = rfl(a.prec[l]);
        const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
        const uint32_t W = rfl(a.nwords[l]);
        WStack st;
        ws_init_loaded(st, a.words + rfl64(a.word_off[l]), W, a.scratch_words + rfl64(a.scratch_off[wi]), roc_dec_stack_cap(n, W),
                       rfl(a.draws[l]), a.mt, VIDC_MT_TABLE);
        const uint32_t draws0 = st.draws;
        uint64_t head = rfl64(a.heads[l]);

        // bucket geometry: fine bucket f = x >> s, coarse c = f >> rb (<= 64 of them), row slot t = f & (RL-1)
        const uint32_t fb = roc_dec_fine_bits(n, P > 32u ? 32u : P);
        const uint32_t cb = fb < 6u ? fb : 6u;
        const uint32_t rb = fb - cb;
        const uint32_t RL = 1u << rb;
        const uint32_t s = (P > 32u ? 32u : P) - fb;
        const uint32_t NF = 1u << fb;
        for (uint32_t t = lane; t < NF && t < lds_entries; t += 64) rowpref[t] = 0;
        uint32_t C1 = 0;  // lane c: decoded elements in coarse buckets 0..c
        uint32_t *slots = LROWS ? (uint32_t *)(smem + (size_t)lds_entries * sizeof(CT)) : a.slots + rfl64(a.slots_off[wi]);
        uint32_t *ovf = LROWS ? a.slots + rfl64(a.slots_off[wi]) : slots + (size_t)NF * cap;
        uint32_t novf = 0, novf_vis = 0;
        uint32_t ring = 0;  // lane t: element decoded at step (block start + t)
        Recip rc;
        wave_sync();

        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            // new 64-step block: earlier slot / overflow stores become visible to loads
            wave_sync();
            novf_vis = novf;
            rc.lq = 0x80000000u / (i0 + 1u + lane);  // lane t owns floor(2^31 / nmax) of step i0 + t
            const uint32_t steps = n - i0 < 64u ? n - i0 : 64u;
            for (uint32_t t64 = 0; t64 < steps; t64++) {
                ws_prepare(st);
                const uint32_t x = ans_id_pop(head, st, p0, p1);
                const uint32_t f = s >= 32u ? 0u : (x >> s);
                const uint32_t c = f >> rb;
                const uint32_t t = f & (RL - 1u);
                // prefix counts of the buckets below f
                const uint32_t prev1 = rl(C1, (c - 1u) & 63u);
                const uint32_t base1 = c ? prev1 : 0u;
                uint32_t base2 = 0, cnt, rowv = 0;
                if (rb) {
                    rowv = lane < RL ? (uint32_t)rowpref[(c << rb) + lane] : 0u;
                    const uint32_t prev2 = rl(rowv, (t - 1u) & 63u);
                    base2 = t ? prev2 : 0u;
                    cnt = rl(rowv, t) - base2;
                } else {
                    cnt = rl(C1, c) - base1;
                }
                // members of bucket f: those stored before this block (visible in memory) ...
                const bool ring_valid = lane < t64;
                const bool ring_same = ring_valid && ((s >= 32u ? 0u : (ring >> s)) == f);
                const uint32_t in_ring = popc64(ballot(ring_same));
                const uint32_t cnt_vis = cnt - in_ring;
                const uint32_t m = cnt_vis < cap ? cnt_vis : cap;
                uint32_t y = 0xffffffffu;
                if (lane < m) y = slots[(size_t)f * cap + lane];
                uint32_t within = popc64(ballot(lane < m && y < x));
                if (__builtin_expect(cnt_vis > cap, 0)) {  // skewed data: members that did not fit the row
                    for (uint32_t j0 = 0; j0 < novf_vis; j0 += 64) {
                        uint32_t jj = j0 + lane;
                        uint32_t z = jj < novf_vis ? ovf[jj] : 0xffffffffu;
                        bool hit = jj < novf_vis && ((s >= 32u ? 0u : (z >> s)) == f) && z < x;
                        within += popc64(ballot(hit));
                    }
                }
                // ... plus those decoded inside the current block (still in flight to memory)
                within += popc64(ballot(ring_same && ring < x));
                const uint32_t r = base1 + base2 + within;
                ans_idx_push(head, st, r, i0 + t64 + 1u, rl(rc.lq, t64));
                // insert x
                if (__builtin_expect(cnt < cap, 1)) {
                    if (lane == 0) slots[(size_t)f * cap + cnt] = x;
                } else {
                    if (lane == 0) ovf[novf] = x;
                    novf++;
                }
                C1 += (lane >= c) ? 1u : 0u;
                if (rb && lane >= t && lane < RL) rowpref[(c << rb) + lane] = (CT)(rowv + 1u);
                ring = wl(x, t64, ring);
            }
            if (lane < steps) {
                const uint32_t step = i0 + lane;
                if (a.out) a.out[ooff + (n - 1u - step)] = (uint64_t)ring;
                else a.out_rows[ooff + (n - 1u - step)] = (int32_t)ring;
            }
        }
        if (lane == 0) {
            bool clean = (head == VIDC_RANS_L) && (st.sp == st.draws - draws0);
            a.end_state[l] = clean ? 0u : 1u;
            a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
        }
        wave_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// status summary: out[0] = smallest list number with an error status (~0 = none), out[1] = number of lists
// whose decode did not end in the initial ANS state, out[2] = number of lists handed back for a retry (status 5),
// out[3] = number of lists waiting for the sorting second pass (status 1; also counted as an error in out[0])
__global__ void k_roc_status_summary(const uint32_t *status, const uint32_t *end_state, uint32_t nlist,
                                     unsigned long long *out) {
    unsigned long long bad = ~0ull, nonclean = 0, retry = 0, pending = 0;
    for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < nlist; l += gridDim.x * blockDim.x) {
        const uint32_t s = status[l];
        if (s == 5u) retry++;  // VIDC_ST_RETRY (roc_lane.h)
        else if (s != VIDC_ST_OK && (unsigned long long)l < bad) bad = l;
        if (s == VIDC_ST_PENDING_SORT) pending++;
        if (end_state) nonclean += end_state[l];
    }
    if (bad != ~0ull) atomicMin(&out[0], bad);
    if (nonclean) atomicAdd(&out[1], nonclean);
    if (retry) atomicAdd(&out[2], retry);
    if (pending) atomicAdd(&out[3], pending);
}

// The same summary for the end of a decode call, delivered by the kernel itself: `acc` (device, zeroed with the call's status block:
// [0] = max over bad lists of ~list, [1..3] as above, [4] = workgroups done) is summed up by atomics, and the LAST workgroup to
// finish stores the four values into `host` -- pinned host memory, read by the host behind the one synchronisation of the call.
// A copy engine between the last kernel and the host's wake-up (init copy up, result copy down) cost more than the kernel.
__global__ void k_roc_status_summary_host(const uint32_t *status, const uint32_t *end_state, uint32_t nlist,
                                          unsigned long long *acc, unsigned long long *host, uint32_t retry_cap) {
    unsigned long long bad = ~0ull, nonclean = 0, retry = 0, pending = 0;
    // (four lists per thread and trip, every load ahead of the first use; at most 256 workgroups: each of them ends with an atomic on the
    // one "workgroups done" counter, and same-address atomics of different workgroups take ~10 ns each, one after the other)
    for (uint32_t l0 = blockIdx.x * blockDim.x * 4u; l0 < nlist; l0 += gridDim.x * blockDim.x * 4u) {
        uint32_t sv[4], ev[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t l = l0 + (uint32_t)j * blockDim.x + threadIdx.x;
            sv[j] = l < nlist ? status[l] : (uint32_t)VIDC_ST_OK;
            ev[j] = (end_state && l < nlist) ? end_state[l] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const unsigned long long l = l0 + (uint32_t)j * blockDim.x + threadIdx.x;
            const uint32_t s = sv[j];
            if (s == 5u) retry++;  // VIDC_ST_RETRY (roc_lane.h)
            else if (s != VIDC_ST_OK && l < bad) bad = l;
            if (s == VIDC_ST_PENDING_SORT) pending++;
            nonclean += ev[j];
        }
    }
    if (bad != ~0ull) atomicMax(&acc[0], ~bad);
    if (nonclean) atomicAdd(&acc[1], nonclean);
    if (retry) atomicAdd(&acc[2], retry);
    if (pending) atomicAdd(&acc[3], pending);
    if (retry && retry_cap) {  // the handed-back lists themselves (a handful per 10^6): the host redoes them without fetching nlist statuses
        for (uint32_t l0 = blockIdx.x * blockDim.x * 4u; l0 < nlist; l0 += gridDim.x * blockDim.x * 4u)  // (this thread's lists again)
            for (uint32_t j = 0; j < 4u; j++) {
                const uint32_t l = l0 + j * blockDim.x + threadIdx.x;
                if (l < nlist && status[l] == 5u) {
                    const unsigned long long k = atomicAdd(&acc[5], 1ull);
                    if (k < retry_cap) host[8 + k] = l;  // (pinned memory: each slot has one writer)
                }
            }
    }
    // (Everything the last workgroup reads was written by device-scope atomics, which are performed at the memory side of the XCDs' L2s:
    // waiting for this wavefront's own outstanding operations orders them before the counter.  A __threadfence() here is a release
    // at device scope -- a write-back of the XCD's L2, full of the decode kernels' output: 67 us for the 4096 wavefronts of a 10^6-row call.)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&acc[4], 1ull) == (unsigned long long)gridDim.x - 1ull) {
        const unsigned long long b = atomicAdd(&acc[0], 0ull);
        host[0] = b ? ~b : ~0ull;
        host[1] = atomicAdd(&acc[1], 0ull);
        host[2] = atomicAdd(&acc[2], 0ull);
        host[3] = atomicAdd(&acc[3], 0ull);
    }
}

// ---------------------------------------------------------------------------------------------
// The tail of an encode call in ONE launch (lists of an IVF object): status summary (as k_roc_status_summary) + exclusive scan
// of the word counts into word_off[0..n] + the total to sum[4].  A single-pass chained scan: tile t (4096 lists) publishes its
// sum in state[t] (bit 63 = valid; the array is zeroed before the launch) and adds up the sums of the tiles before it.  Tile
// numbers are handed out by a counter (state[gridDim.x]) in the order the workgroups START, not taken from blockIdx: a
// workgroup only waits for tiles whose workgroups are already running -- so the wait always ends, whatever the dispatch order.  The four launches this replaces (summary, tile sums, scan of the
// tiles, apply) were ~25 us between the end of a 65 536-list call's encode kernel and the host's wake-up.
// state: [gridDim.x tile sums | tile counter | tiles done | sum[8]], all zero at the launch; sum[0] = max over bad lists of ~list,
// sum[2] / [3] = retries / pending sorts, sum[4] = total words.  The last tile to finish stores {first bad list or ~0, 0, retries,
// pending, total} into `host` (pinned host memory: no copy engine between this kernel and the host's wake-up).
#define VIDC_TAIL_TILE 4096u
// (Graph objects have their own end of call: k_roc_rows_totals / k_roc_rows_compact below.)
__global__ void __launch_bounds__(256) k_roc_tail(const uint32_t *__restrict__ nwords, uint32_t n, uint64_t *__restrict__ word_off,
                                                  const uint32_t *__restrict__ status, unsigned long long *state,
                                                  unsigned long long *host) {
    unsigned long long *const sum = state + gridDim.x + 2;
    __shared__ uint64_t sh[256];
    __shared__ uint64_t tile_off_s;
    __shared__ uint32_t tile_s;
    const uint32_t t = threadIdx.x;
    if (t == 0) tile_s = (uint32_t)atomicAdd(&state[gridDim.x], 1ull);
    __syncthreads();
    const uint32_t tile = tile_s;
    const uint32_t base = tile * VIDC_TAIL_TILE + t * 16u;
    uint32_t v[16];
    uint64_t s = 0;
    unsigned long long bad = ~0ull, retry = 0, pending = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const uint32_t l = base + j;
        v[j] = l < n ? nwords[l] : 0u;
        s += v[j];
        if (l < n) {
            const uint32_t st = status[l];
            if (st == 5u) retry++;  // VIDC_ST_RETRY (roc_lane.h)
            else if (st != VIDC_ST_OK && (unsigned long long)l < bad) bad = l;
            if (st == VIDC_ST_PENDING_SORT) pending++;
        }
    }
    sh[t] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint64_t x = t >= o ? sh[t - o] : 0;
        __syncthreads();
        sh[t] += x;
        __syncthreads();
    }
    const uint64_t incl = sh[t], tile_sum = sh[255];
    // (no fence in front: the published value is the exchange's own operand, and a device-scope release writes back the XCD's whole L2)
    if (t == 0) atomicExch(&state[tile], (1ull << 63) | tile_sum);
    // sums of the tiles before this one (they were dispatched earlier: every wait ends)
    uint64_t before = 0;
    for (uint32_t k = t; k < tile; k += 256u) {
        unsigned long long x;
        // (device-scope atomic LOADS: read-modify-write polls of 245 tiles on the same few cache lines serialise in the memory system)
        do { x = __hip_atomic_load(&state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (!(x >> 63));
        before += x & ~(1ull << 63);
    }
    __syncthreads();  // (sh is read above by every thread before it is reused)
    sh[t] = before;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) {
        if (t < o) sh[t] += sh[t + o];
        __syncthreads();
    }
    if (t == 0) tile_off_s = sh[0];
    __syncthreads();
    uint64_t acc = tile_off_s + incl - s;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (base + j <= n) word_off[base + j] = acc;  // (index n receives the total)
        if (base + j == n) atomicExch(&sum[4], acc);
        acc += v[j];
    }
    // status summary: few lists ever report anything
    if (bad != ~0ull) atomicMax(&sum[0], ~bad);
    if (retry) atomicAdd(&sum[2], retry);
    if (pending) atomicAdd(&sum[3], pending);
    __builtin_amdgcn_s_waitcnt(0);  // (see k_roc_status_summary_host: atomics only, no device-scope release)
    __syncthreads();
    if (t == 0 && atomicAdd(&state[gridDim.x + 1], 1ull) == (unsigned long long)gridDim.x - 1ull) {
        const unsigned long long b = atomicAdd(&sum[0], 0ull);
        host[0] = b ? ~b : ~0ull;
        host[1] = 0ull;
        host[2] = atomicAdd(&sum[2], 0ull);
        host[3] = atomicAdd(&sum[3], 0ull);
        host[4] = atomicAdd(&sum[4], 0ull);
    }
}

// ---------------------------------------------------------------------------------------------
// compaction of the worst-case arena into the CSR stream
__global__ void k_roc_compact(const uint32_t *arena, const uint64_t *offsets, uint32_t stride, const uint64_t *word_off,
                              uint32_t *words, uint32_t nlist) {
    for (uint32_t l = blockIdx.x; l < nlist; l += gridDim.x) {
        const uint32_t *src = arena + roc_arena_at(offsets, stride, l);
        uint32_t *dst = words + word_off[l];
        uint32_t nw = (uint32_t)(word_off[l + 1] - word_off[l]);
        for (uint32_t j = threadIdx.x; j < nw; j += blockDim.x) dst[j] = src[j];
    }
}

// The same for lists of ~64 .. 256 words (65 536 lists of 256 ids: 84 words each): a wavefront per FOUR consecutive lists, their
// headers loaded together and the first 128 words of each requested before anything is stored.  A workgroup per list spent a
// round trip on three header words and another on 84 words of payload, sixteen times in a row: 33 us for 22 MB.
__global__ void __launch_bounds__(256) k_roc_compact_waves(const uint32_t *__restrict__ arena, const uint64_t *__restrict__ offsets,
                                                           uint32_t stride, const uint64_t *__restrict__ word_off,
                                                           uint32_t *__restrict__ words, uint32_t nlist) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, nwaves = gridDim.x * 4u;
    for (uint64_t l0 = (uint64_t)wave * 4u; l0 < nlist; l0 += (uint64_t)nwaves * 4u) {
        uint64_t so[4], dof[5];
#pragma unroll
        for (uint32_t k = 0; k < 5; k++) dof[k] = word_off[l0 + k <= nlist ? l0 + k : nlist];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) so[k] = roc_arena_at(offsets, stride, l0 + k < nlist ? l0 + k : nlist - 1u);
        uint32_t v[4][2];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t nw = (uint32_t)(dof[k + 1] - dof[k]);
#pragma unroll
            for (uint32_t j = 0; j < 2; j++) {
                const uint32_t i = lane + 64u * j;
                v[k][j] = i < nw ? arena[so[k] + i] : 0u;
            }
        }
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
            const uint32_t nw = (uint32_t)(dof[k + 1] - dof[k]);
#pragma unroll
            for (uint32_t j = 0; j < 2; j++) {
                const uint32_t i = lane + 64u * j;
                if (i < nw) words[dof[k] + i] = v[k][j];
            }
            for (uint32_t i = lane + 128u; i < nw; i += 64u) words[dof[k] + i] = arena[so[k] + i];
        }
    }
}

// The same for many short lists / graph rows (a few dozen words each): one wavefront per 64 consecutive lists, lane
// per OUTPUT word (the group's words are contiguous in `words`), the owning list found by a binary search over the
// group's 65 word offsets in LDS.  A workgroup per 34-word row spent its time on launch and header loads: the
// compaction of 10^6 graph rows took as long as their encode kernel.
__global__ void __launch_bounds__(64) k_roc_compact_groups(const uint32_t *arena, const uint64_t *offsets, uint32_t stride,
                                                           const uint64_t *word_off, uint32_t *words, uint32_t nlist) {
    __shared__ uint64_t wo[65];
    __shared__ uint64_t ao[64];
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t g0 = blockIdx.x * 64u; g0 < nlist; g0 += gridDim.x * 64u) {
        const uint32_t cnt = nlist - g0 < 64u ? nlist - g0 : 64u;
        if (lane < cnt) {
            wo[lane] = word_off[g0 + lane];
            ao[lane] = roc_arena_at(offsets, stride, g0 + lane);
        }
        if (lane == 0) wo[cnt] = word_off[g0 + cnt];
        __syncthreads();
        const uint64_t w_begin = wo[0], total = wo[cnt] - w_begin;
        // eight words per lane and trip, every load issued before the first store (round 6: one load per trip -- load, wait, store --
        // was ~25 memory round trips in a row per group: 78 us for 10^6 graph rows, the kernel's wavefronts waiting 83 % of the time)
        for (uint64_t k0 = 0; k0 < total; k0 += 512u) {
            uint32_t v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint64_t k = k0 + (uint64_t)j * 64u + lane;
                v[j] = 0u;
                if (k < total) {
                    const uint64_t g = w_begin + k;
                    uint32_t lo = 0, hi = cnt;  // largest r with wo[r] <= g
                    while (hi - lo > 1u) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (wo[mid] <= g) lo = mid; else hi = mid;
                    }
                    v[j] = arena[ao[lo] + (g - wo[lo])];
                }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint64_t k = k0 + (uint64_t)j * 64u + lane;
                if (k < total) words[w_begin + k] = v[j];
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// End of a GRAPH encode (round 6, second half): two launches instead of k_roc_tail<true> + k_roc_compact_groups (36 + 67 us behind a
// 310 us encode kernel on 10^6 rows, both of them chains of memory round trips):
//   k_roc_rows_totals   workgroup c adds up CHUNK c (4096 rows); the last workgroup to finish adds the chunks up -> {first bad row,
//                       retries, pending, words, edges, non-empty rows} in pinned host memory (all the host needs to size the stream)
//                       and leaves the exclusive scan of the chunks' word and edge counts in chunk_pref
//   k_roc_rows_compact  one wavefront per TILE of 64 rows: word offset of the tile = chunk_pref of its chunk + the sums of the tiles
//                       in front of it in the SAME chunk (at most 63: one look at their published sums); the wavefront requests its
//                       rows' words from the arena first (the addresses need only its own 64 counts) and looks at its neighbours while
//                       they are in flight.  The same for the edge counts -> CSR offsets.
// Two designs measured and dropped: a counter handing out tile numbers (one same-address atomic per tile: 15 625 of them = 160 us;
// atomics of many workgroups on ONE address are executed one after the other on the memory side of the eight L2s, ~10 ns each -- also
// why the totals are not atomicAdds of every wavefront: 114 us); a decoupled look-back over all tiles (4096 resident tiles that start
// together find no finished prefix in front of them: the frontier advances one 64-tile window per round trip, 170 us).
// state: [tile sums, words: ntiles | tile sums, edges: ntiles | chunk sums: nchunks x 8 | counter | chunk_pref: nchunks x 2], zero at the launch.
#define VIDC_ROWS_CHUNK 4096u
__global__ void __launch_bounds__(256) k_roc_rows_totals(const uint32_t *__restrict__ nwords, const uint32_t *__restrict__ sizes,
                                                         const uint32_t *__restrict__ status, uint32_t n, unsigned long long *part,
                                                         unsigned long long *host) {
    __shared__ unsigned long long red[4][6];
    __shared__ unsigned long long sc[256][2];
    __shared__ bool last;
    unsigned long long *const chunk_pref = part + (size_t)gridDim.x * 8u + 1u;
    unsigned long long w = 0, e = 0, nz = 0, retry = 0, pending = 0, bad = ~0ull;
    const uint32_t t = threadIdx.x;
    {
        const uint32_t l0 = blockIdx.x * VIDC_ROWS_CHUNK;
        uint32_t a[16], b[16], c[16];
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t l = l0 + (uint32_t)j * 256u + t;
            const bool in = l < n;
            a[j] = in ? nwords[l] : 0u;
            b[j] = in ? sizes[l] : 0u;
            c[j] = in ? status[l] : (uint32_t)VIDC_ST_OK;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const unsigned long long l = l0 + (uint32_t)j * 256u + t;
            w += a[j];
            e += b[j];
            nz += b[j] != 0u;
            if (c[j] == 5u) retry++;  // VIDC_ST_RETRY (roc_lane.h)
            else if (c[j] != VIDC_ST_OK && l < bad) bad = l;
            if (c[j] == VIDC_ST_PENDING_SORT) pending++;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        w += __shfl_xor(w, o, 64);
        e += __shfl_xor(e, o, 64);
        nz += __shfl_xor(nz, o, 64);
        retry += __shfl_xor(retry, o, 64);
        pending += __shfl_xor(pending, o, 64);
        const unsigned long long ob = __shfl_xor(bad, o, 64);
        bad = ob < bad ? ob : bad;
    }
    if ((t & 63u) == 0u) {
        unsigned long long *r = red[t >> 6];
        r[0] = bad; r[1] = retry; r[2] = pending; r[3] = w; r[4] = e; r[5] = nz;
    }
    __syncthreads();
    if (t < 6u) {
        unsigned long long v = red[0][t];
        for (int k = 1; k < 4; k++) v = t == 0u ? (red[k][0] < v ? red[k][0] : v) : v + red[k][t];
        atomicExch(&part[(size_t)blockIdx.x * 8u + t], v);  // (its own address: nothing queues behind it)
    }
    __builtin_amdgcn_s_waitcnt(0);  // (atomics only, no device-scope release: k_roc_status_summary_host)
    __syncthreads();
    if (t == 0) last = atomicAdd(&part[(size_t)gridDim.x * 8u], 1ull) == (unsigned long long)gridDim.x - 1ull;
    __syncthreads();
    if (!last) return;
    // thread t: chunks [t * cpt, (t + 1) * cpt)
    const uint32_t nch = gridDim.x, cpt = (nch + 255u) / 256u;
    unsigned long long acc[6] = {~0ull, 0, 0, 0, 0, 0};
    for (uint32_t q = 0; q < cpt; q++) {
        const uint32_t ch = t * cpt + q;
        if (ch < nch) {
#pragma unroll
            for (int k = 0; k < 6; k++) {
                const unsigned long long v = __hip_atomic_load(&part[(size_t)ch * 8u + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc[k] = k == 0 ? (v < acc[0] ? v : acc[0]) : acc[k] + v;
            }
        }
    }
    sc[t][0] = acc[3];
    sc[t][1] = acc[4];
    __syncthreads();
    for (uint32_t o = 1; o < 256u; o <<= 1) {  // inclusive scan over the threads' runs of chunks
        const unsigned long long x0 = t >= o ? sc[t - o][0] : 0ull, x1 = t >= o ? sc[t - o][1] : 0ull;
        __syncthreads();
        sc[t][0] += x0;
        sc[t][1] += x1;
        __syncthreads();
    }
    {
        unsigned long long pw = sc[t][0] - acc[3], pe = sc[t][1] - acc[4];
        for (uint32_t q = 0; q < cpt; q++) {
            const uint32_t ch = t * cpt + q;
            if (ch < nch) {
                chunk_pref[2u * ch] = pw;
                chunk_pref[2u * ch + 1u] = pe;
                pw += __hip_atomic_load(&part[(size_t)ch * 8u + 3u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pe += __hip_atomic_load(&part[(size_t)ch * 8u + 4u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long ov = __shfl_xor(acc[k], o, 64);
            acc[k] = k == 0 ? (ov < acc[0] ? ov : acc[0]) : acc[k] + ov;
        }
    }
    if ((t & 63u) == 0u)
        for (int k = 0; k < 6; k++) red[t >> 6][k] = acc[k];
    __syncthreads();
    if (t == 0) {
        unsigned long long f[6];
        for (int k = 0; k < 6; k++) {
            f[k] = red[0][k];
            for (int q = 1; q < 4; q++) f[k] = k == 0 ? (red[q][0] < f[0] ? red[q][0] : f[0]) : f[k] + red[q][k];
        }
        host[0] = f[0];  // first bad row or ~0
        host[1] = 0ull;
        host[2] = f[1];
        host[3] = f[2];
        host[4] = f[3];
        host[5] = f[4];
        host[6] = f[5];
    }
}

#define VIDC_RC_REGS 32  // words per lane requested before anything is stored (2048 per tile and pass; 10^6 K = 64 rows: ~1600 per tile)
__global__ void __launch_bounds__(64) k_roc_rows_compact(const uint32_t *__restrict__ arena, uint32_t stride,
                                                         const uint32_t *__restrict__ nwords, const uint32_t *__restrict__ sizes,
                                                         uint32_t n, unsigned long long *state, uint32_t ntiles,
                                                         const unsigned long long *__restrict__ chunk_pref,
                                                         uint64_t *__restrict__ word_off, uint64_t *__restrict__ offsets,
                                                         uint32_t *__restrict__ words) {
    __shared__ uint32_t lp[65];
    const uint32_t lane = threadIdx.x;
    unsigned long long *const state_w = state, *const state_e = state + ntiles;
    // Tiles blockIdx.x, blockIdx.x + gridDim.x, ...: every wavefront of the launch is resident at once (the host sizes the grid by the
    // runtime's occupancy figure) and gridDim.x is a multiple of 64, so a tile waits only for running wavefronts with smaller numbers.
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t l = tile * 64u + lane;
        const bool in = l < n;
        const uint32_t nw = in ? nwords[l] : 0u, ne = in ? sizes[l] : 0u;
        const uint32_t first = tile & ~63u, j = tile & 63u;  // the chunk's first tile; tiles in front of this one in the chunk
        const unsigned long long cw = chunk_pref[2u * (tile >> 6)], ce = chunk_pref[2u * (tile >> 6) + 1u];
        uint32_t iw = nw, ie = ne;  // inclusive scans over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t a = __shfl_up(iw, o, 64), b = __shfl_up(ie, o, 64);
            if (lane >= (uint32_t)o) { iw += a; ie += b; }
        }
        const uint32_t total = rl(iw, 63u), total_e = rl(ie, 63u);
        if (lane == 0 && j != 63u) {  // (bit 62: published)
            atomicExch(&state_w[tile], (1ull << 62) | total);
            atomicExch(&state_e[tile], (1ull << 62) | total_e);
        }
        lp[lane] = iw - nw;
        wave_lds_sync();
        // the tile's words: lane per OUTPUT word (they are contiguous in `words`), the row it belongs to found in the 64 local offsets
        uint32_t v[VIDC_RC_REGS];
        const uint64_t abase = (uint64_t)tile * 64u * stride;
#pragma unroll
        for (int q = 0; q < VIDC_RC_REGS; q++) {
            v[q] = 0u;
            if ((uint32_t)q * 64u < total) {  // (uniform)
                const uint32_t k = (uint32_t)q * 64u + lane;
                const uint32_t kk = k < total ? k : total - 1u;
                uint32_t lo = 0, hi = 64;  // largest r with lp[r] <= kk
#pragma unroll
                for (int it = 0; it < 6; it++) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const bool ge = lp[mid] <= kk;
                    lo = ge ? mid : lo;
                    hi = ge ? hi : mid;
                }
                v[q] = arena[abase + (uint64_t)lo * stride + (kk - lp[lo])];
            }
        }
        // sums of the tiles in front of this one in its chunk: lane k < j looks at tile first + k
        unsigned long long xw = 1ull << 62, xe = 1ull << 62;
        for (;;) {
            if (lane < j) {
                xw = __hip_atomic_load(&state_w[first + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                xe = __hip_atomic_load(&state_e[first + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!ballot(!(xw >> 62) || !(xe >> 62))) break;
            __builtin_amdgcn_s_sleep(1);
        }
        unsigned long long bw = lane < j ? (xw & ~(1ull << 62)) : 0ull, be = lane < j ? (xe & ~(1ull << 62)) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            bw += __shfl_xor(bw, o, 64);
            be += __shfl_xor(be, o, 64);
        }
        const uint64_t before_w = cw + bw, before_e = ce + be;
        if (in) {
            word_off[l] = before_w + (iw - nw);
            offsets[l] = before_e + (ie - ne);
        }
        if (l + 1u == n) {  // (index n receives the totals)
            word_off[n] = before_w + iw;
            offsets[n] = before_e + ie;
        }
        uint32_t *dst = words + before_w;
#pragma unroll
        for (int q = 0; q < VIDC_RC_REGS; q++) {
            const uint32_t k = (uint32_t)q * 64u + lane;
            if (k < total) dst[k] = v[q];
        }
        for (uint32_t k = (uint32_t)VIDC_RC_REGS * 64u + lane; k < total; k += 64u) {  // (rows of wide ids: up to 82 words each)
            uint32_t lo = 0, hi = 64;
            for (int it = 0; it < 6; it++) {
                const uint32_t mid = (lo + hi) >> 1;
                const bool ge = lp[mid] <= k;
                lo = ge ? mid : lo;
                hi = ge ? hi : mid;
            }
            dst[k] = arena[abase + (uint64_t)lo * stride + (k - lp[lo])];
        }
        wave_lds_sync();  // (lp is rewritten by the next tile)
    }
}

// ---------------------------------------------------------------------------------------------
// Rows of a graph object ordered by edge count, most edges first (whole-graph decodes, round 6): the lane-per-row decoder runs as many
// steps as the longest row of its 64, and the rank of a step costs its index -- with the rows of a wavefront equally long a graph of
// degrees 32..64 takes 25 % fewer steps and ~40 % fewer rank comparisons than in node order.  Counting sort: histogram, 65 starts, scatter.
// No global atomics: workgroup b counts ITS rows (LDS), one workgroup lays the (count descending, workgroup ascending) segments out,
// workgroup b scatters its rows into its segments.  (Global counters shared by all workgroups: 25 + 92 us for 10^6 rows.)
#define VIDC_ORDER_BLOCKS 256u
__global__ void __launch_bounds__(256) k_rows_order_hist(const uint64_t *__restrict__ offsets, uint32_t n, uint32_t *part) {
    __shared__ uint32_t h[65];
    if (threadIdx.x < 65u) h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t chunk = (n + gridDim.x - 1u) / gridDim.x;
    const uint32_t lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (uint32_t l = lo + threadIdx.x; l < hi; l += 256u) {
        const uint32_t c = (uint32_t)(offsets[l + 1] - offsets[l]);
        atomicAdd(&h[c < 64u ? c : 64u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 65u) part[blockIdx.x * 65u + threadIdx.x] = h[threadIdx.x];
}
// ONE workgroup, thread = edge count: its 256 per-workgroup counts in registers (every load requested before the first add; a loop
// that loaded, added and stored one count per trip was 256 memory round trips: 77 us), part -> segment starts, in place
__global__ void __launch_bounds__(128) k_rows_order_starts(uint32_t *part) {
    __shared__ uint32_t tot[65], first[65];
    const uint32_t bin = threadIdx.x;
    uint32_t c[VIDC_ORDER_BLOCKS];
    if (bin < 65u) {
#pragma unroll
        for (uint32_t b = 0; b < VIDC_ORDER_BLOCKS; b++) c[b] = part[b * 65u + bin];
        uint32_t run = 0;
#pragma unroll
        for (uint32_t b = 0; b < VIDC_ORDER_BLOCKS; b++) {
            const uint32_t t = c[b];
            c[b] = run;
            run += t;
        }
        tot[bin] = run;
    }
    __syncthreads();
    if (bin == 0) {
        uint32_t run = 0;
        for (int k = 64; k >= 0; k--) { first[k] = run; run += tot[k]; }  // most edges first
    }
    __syncthreads();
    if (bin < 65u) {
        const uint32_t f = first[bin];
#pragma unroll
        for (uint32_t b = 0; b < VIDC_ORDER_BLOCKS; b++) part[b * 65u + bin] = c[b] + f;
    }
}
__global__ void __launch_bounds__(256) k_rows_order_scatter(const uint64_t *__restrict__ offsets, uint32_t n, const uint32_t *__restrict__ part,
                                                            uint32_t *__restrict__ order) {
    __shared__ uint32_t cur[65];
    if (threadIdx.x < 65u) cur[threadIdx.x] = part[blockIdx.x * 65u + threadIdx.x];
    __syncthreads();
    const uint32_t chunk = (n + gridDim.x - 1u) / gridDim.x;
    const uint32_t lo = blockIdx.x * chunk, hi = lo + chunk < n ? lo + chunk : n;
    for (uint32_t l = lo + threadIdx.x; l < hi; l += 256u) {
        const uint32_t c = (uint32_t)(offsets[l + 1] - offsets[l]);
        order[atomicAdd(&cur[c < 64u ? c : 64u], 1u)] = l;
    }
}

}  // namespace dev
}  // namespace vidc
