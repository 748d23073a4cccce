// wave.h -- device-side building blocks for "one list per wavefront" bitstream kernels (gfx950).
//
// A workgroup is exactly one 64-lane wavefront.  The ANS state (head, stack pointers) is
// wave-uniform and lives in SGPRs; the top of the ANS stack is a 64-word ring held in ONE VGPR
// (lane = word index & 63) so that push/pop are v_writelane/v_readlane, with 32-word coalesced
// spills/refills to HBM.  Order statistics use ballots + s_ff1/s_bcnt1 over per-lane counters.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define VIDC_WAVE 64
#define VIDC_RANS_L (1ull << 31)  // rans_l, codec.cpp:19

namespace vidc {
namespace dev {

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t rl(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
// write a wave-uniform value into one lane of a per-lane register (this clang has no writelane
// builtin; v_cmp_eq + v_cndmask with SGPR operands does the same in two VALU issues)
__device__ __forceinline__ uint32_t wl(uint32_t val, uint32_t lane, uint32_t old) {
    return lane_id() == lane ? val : old;
}
__device__ __forceinline__ uint64_t rfl64(uint64_t v) {
    return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v);
}
__device__ __forceinline__ uint64_t rl64(uint32_t lo, uint32_t hi, uint32_t lane) {
    return ((uint64_t)rl(hi, lane) << 32) | rl(lo, lane);
}
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
// Keep a wave-uniform value in a VGPR (all lanes equal) and make the compiler treat it as a vector value.
// A VALU-written SGPR (v_readlane, v_cmp mask) costs ~16 extra cycles when the SCALAR unit consumes it but nothing
// when a VALU instruction does (tools/ubench_issue.hip), so the select/rank arithmetic is kept on the VALU.
__device__ __forceinline__ uint32_t to_v(uint32_t x) {
    uint32_t y;
    asm volatile("" : "=v"(y) : "0"(x));
    return y;
}
// number of set bits of `mask` strictly below this lane's position
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint32_t ff1(uint64_t m) { return (uint32_t)__builtin_ctzll(m); }
__device__ __forceinline__ uint32_t popc64(uint64_t m) { return (uint32_t)__builtin_popcountll(m); }
// value of the previous lane (lane 0 gets 0): DPP wave_shr:1
__device__ __forceinline__ uint32_t lane_shr1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
// inclusive prefix sum over lanes 0..3 of each 16-lane row (DPP row_shr:1, row_shr:2); other lanes: don't care
__device__ __forceinline__ uint32_t prefix4(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    return v;
}
// ascending sort of KP values held in KP registers OF ONE LANE (every lane sorts its own row, 64 rows per wavefront): Batcher's
// merge-exchange network, unrolled at compile time -- 543 min/max pairs for KP = 64 (191 for 32) against the 672 (240) of the
// bitonic network used up to round 5
template <int KP>
__device__ __forceinline__ void lane_sort(uint32_t (&r)[KP]) {
#pragma unroll
    for (int p = 1; p < KP; p <<= 1) {
#pragma unroll
        for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
            for (int j = k % p; j <= KP - 1 - k; j += 2 * k) {
#pragma unroll
                for (int i = 0; i < k; i++) {
                    const int a = i + j, b = i + j + k;
                    if (b < KP && a / (2 * p) == b / (2 * p)) {
                        const uint32_t lo = r[a] < r[b] ? r[a] : r[b], hi = r[a] < r[b] ? r[b] : r[a];
                        r[a] = lo;
                        r[b] = hi;
                    }
                }
            }
        }
    }
}

// single-wave workgroup: orders LDS / global accesses between lanes of the wave
__device__ __forceinline__ void wave_sync() { __syncthreads(); }
// single-wave workgroup, LDS only: the wavefront's LDS operations execute in order, so lanes see each other's LDS writes without a
// barrier; what has to be stopped is the COMPILER moving LDS accesses across this point.  __syncthreads() also waits for every
// outstanding global load and store of the wavefront (s_waitcnt vmcnt(0)): a block requested for the NEXT tile would be waited for at the
// first barrier of the current one.
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t w = (uint32_t)__shfl_xor((int)v, o, 64);
        v = v > w ? v : w;
    }
    return rfl(v);
}

// ---------------------------------------------------------------------------------------------
// ANS stack: words [0, lo) are in memory, words [lo, sp) in the register ring.
// Encoder: mem == orig == the list's output arena.  Decoder: orig = stored stream (read-only),
// mem = private scratch for the (practically never taken) re-spill of pushed words.
//
// Hot-path contract: ws_prepare() once per codec step guarantees room for >= 8 pushes and (when the
// stack holds them) >= 8 pops inside the ring, so ws_push / ws_pop are branch-free ring accesses
// (a codec step performs at most 5 of either).
struct WStack {
    uint32_t win;          // per-lane: word (idx) with idx & 63 == lane, lo <= idx < sp
    uint32_t lo, sp;       // wave-uniform, lo multiple of 32, sp - lo <= 64
    uint32_t sp_min, sp_span;  // hot-path window: no spill/refill needed while sp - sp_min < sp_span
    uint32_t cap;          // capacity of mem in words
    uint32_t dirty;        // decoder: words in [dirty, lo) live in mem, words below in orig
    uint32_t draws;        // mt19937(1234) words consumed (codec.h:32-40)
    uint32_t err;          // sticky: 1 = arena overflow, 2 = mt table exhausted
    uint32_t *mem;
    const uint32_t *orig;
    const uint32_t *mt;
    uint32_t mt_n;
};

// window of stack pointers for which the ring needs no maintenance: resident count in [8, 56], or anything up
// to 56 while the whole stack is still resident (lo == 0)
__device__ __forceinline__ void ws_window(WStack &s) {
    s.sp_min = s.lo ? s.lo + 8u : 0u;
    s.sp_span = s.lo ? 49u : 57u;
}

__device__ __forceinline__ void ws_init_empty(WStack &s, uint32_t *arena, uint32_t cap, const uint32_t *mt,
                                              uint32_t mt_n) {
    s.win = 0; s.lo = 0; s.sp = 0; s.cap = cap; s.dirty = 0; s.draws = 0; s.err = 0;
    s.mem = arena; s.orig = arena; s.mt = mt; s.mt_n = mt_n;
    ws_window(s);
}

__device__ __forceinline__ void ws_init_loaded(WStack &s, const uint32_t *orig, uint32_t nwords, uint32_t *scratch,
                                               uint32_t cap, uint32_t draws, const uint32_t *mt, uint32_t mt_n) {
    s.cap = cap; s.dirty = 0xffffffffu; s.draws = draws; s.err = 0;
    s.mem = scratch; s.orig = orig; s.mt = mt; s.mt_n = mt_n;
    s.sp = nwords;
    s.lo = nwords ? ((nwords - 1u) & ~31u) : 0u;
    uint32_t idx = s.lo + ((lane_id() - s.lo) & 63u);
    s.win = idx < nwords ? orig[idx] : 0u;
    ws_window(s);
}

// cold: write the 32 oldest resident words to memory
__device__ __forceinline__ void ws_spill32(WStack &s) {
    uint32_t rel = (lane_id() - s.lo) & 63u;
    uint32_t idx = s.lo + rel;
    if (rel < 32u) {
        if (idx < s.cap) s.mem[idx] = s.win;
    }
    if (s.lo + 32u > s.cap) s.err |= 1u;
    if (s.dirty > s.lo) s.dirty = s.lo;
    s.lo += 32u;
    ws_window(s);
}
// cold: bring the 32 words below the ring back in (their ring slots are free: resident <= 32 here)
__device__ __forceinline__ void ws_refill32(WStack &s) {
    uint32_t base = s.lo - 32u;
    uint32_t rel = (lane_id() - base) & 63u;
    uint32_t idx = base + rel;
    if (rel < 32u) s.win = (idx >= s.dirty) ? s.mem[idx] : s.orig[idx];
    s.lo = base;
    ws_window(s);
}

__device__ __forceinline__ void ws_prepare(WStack &s) {
    // one subtract + one unsigned compare on the hot path
    if (__builtin_expect(s.sp - s.sp_min >= s.sp_span, 0)) {
        const uint32_t res = s.sp - s.lo;
        if (res > 56u) ws_spill32(s);
        else if (res < 8u && s.lo != 0u) ws_refill32(s);
    }
}

__device__ __forceinline__ void ws_push(WStack &s, uint32_t w) {  // codec.h:20-22
    const uint32_t sp = rfl(s.sp);  // keep the stack pointer chain in SGPRs
    s.win = wl(w, sp & 63u, s.win);
    s.sp = sp + 1u;
}

__device__ __forceinline__ uint32_t ws_pop(WStack &s) {  // codec.h:32-40
    if (__builtin_expect(s.sp == 0u, 0)) {
        uint32_t w = 0u;
        if (s.draws < s.mt_n) w = s.mt[s.draws]; else s.err |= 2u;
        s.draws += 1u;
        return rfl(w);
    }
    const uint32_t sp = rfl(s.sp) - 1u;
    s.sp = sp;
    return rl(s.win, sp & 63u);
}

// write every resident word to mem (encoder epilogue)
__device__ __forceinline__ void ws_flush(WStack &s) {
    uint32_t rel = (lane_id() - s.lo) & 63u;
    uint32_t idx = s.lo + rel;
    if (idx < s.sp) {
        if (idx < s.cap) s.mem[idx] = s.win;
    }
    if (s.sp > s.cap) s.err |= 1u;
}

// ------------------------------------------------------------------------------- ANS primitives
// All arguments are wave-uniform.

// v < 2^31 (rans_l) with 32-bit scalar ops: the scalar unit has no ordered 64-bit compare
__device__ __forceinline__ bool lt_2p31(uint64_t v) {
    return (((uint32_t)(v >> 32)) | ((uint32_t)v >> 31)) == 0u;
}

// codec.cpp:65-76 : uniform 2^p push, 0 <= p <= 16.  '+' (not '|'): carries if start >= 2^p.
__device__ __forceinline__ void ans_u_push(uint64_t &head, WStack &s, uint32_t start, uint32_t p) {
    // (a branch-free, predicated form of this renormalisation measured 9 % slower per step: tools/ab_chain.sh)
    if ((uint32_t)(head >> 32) >= (0x80000000u >> p)) {
        ws_push(s, (uint32_t)head);
        head >>= 32;
    }
    head = (head << p) + start;
}
// codec.cpp:78-90
__device__ __forceinline__ uint32_t ans_u_pop(uint64_t &head, WStack &s, uint32_t p) {
    uint32_t sym = (uint32_t)head & ((1u << p) - 1u);
    head >>= p;
    if (lt_2p31(head)) head = (head << 32) | ws_pop(s);
    return sym;
}
// codec.cpp:92-105 : four 16-bit slices, low -> high; slice precisions clamp(P - lower, 0, 16).
// ids are < 2^32 here, so slices 2 and 3 carry symbol 0 (their renormalisation checks still run).
__device__ __forceinline__ void ans_id_push(uint64_t &head, WStack &s, uint32_t x, uint32_t p0, uint32_t p1) {
    ans_u_push(head, s, x & 0xffffu, p0);
    ans_u_push(head, s, x >> 16, p1);
    // slices 2 and 3 (precision 0, symbol 0): "if (head >= 2^63) push", twice; after one push head < 2^31, so the
    // second test can only fire when the first did not.  Only reachable through the carry quirk (x >= 2^P).
    uint32_t top;
    asm volatile("s_lshr_b32 %0, %1, 31" : "=s"(top) : "s"((uint32_t)(head >> 32)));
    if (__builtin_expect(top != 0u, 0)) {
        ans_u_push(head, s, 0u, 0u);
        ans_u_push(head, s, 0u, 0u);
    }
}
// codec.cpp:107-121 : slices high -> low
__device__ __forceinline__ uint32_t ans_id_pop(uint64_t &head, WStack &s, uint32_t p0, uint32_t p1) {
    // slices 3 and 2 (precision 0) only test "head < 2^31 -> refill"; a refill makes head >= 2^31, so one test
    // covers both unless the first refill happened (then the second slice is evaluated normally)
    if (__builtin_expect(lt_2p31(head), 0)) {
        (void)ans_u_pop(head, s, 0u);
        (void)ans_u_pop(head, s, 0u);
    }
    uint32_t hi = ans_u_pop(head, s, p1);
    uint32_t lo = ans_u_pop(head, s, p0);
    return (hi << 16) | lo;
}
// codec.cpp:21-42 : k = h0 mod nmax, head = h0 div nmax.
//   thr = nmax * floor(2^31 / nmax)  (<= 2^31), magic = floor((2^64-1) / nmax): q = mulhi(h0, magic)
//   is floor(h0/nmax) or one less (0 <= h0/nmax - h0*magic/2^64 < 1), fixed by one correction.
__device__ __forceinline__ uint32_t ans_idx_pop(uint64_t &head, WStack &s, uint32_t nmax, uint32_t thr,
                                                uint64_t magic) {
    uint64_t h0 = head;
    if (__builtin_expect((uint32_t)(h0 >> 32) >= thr, 0)) {  // h0 >= nmax * ((L / nmax) << 32)
        ws_push(s, (uint32_t)h0);
        h0 >>= 32;
    }
    uint64_t q = __umul64hi(h0, magic);
    uint32_t r = (uint32_t)h0 - (uint32_t)q * nmax;
    const uint32_t ge = (uint32_t)(((uint64_t)r - (uint64_t)nmax) >> 63) ^ 1u;  // 1 iff r >= nmax
    r -= nmax & (0u - ge);
    q += ge;
    if (__builtin_expect(lt_2p31(h0), 0)) q = (uint64_t)ws_pop(s) | (q << 32);  // the test is on h0 (codec.cpp:35)
    head = q;
    return r;
}
// Same, with the reciprocal kept per lane (lane t64 owns the divisor of this step): every lane divides the
// wave-uniform h0 by ITS divisor d (multiply by its reciprocal with v_mad_u64_u32, remainder and correction on
// the VALU); only the owner's quotient / remainder are read back.  Returns k as a uniform VGPR value.
__device__ __forceinline__ uint32_t ans_idx_pop_v(uint64_t &head, WStack &s, uint32_t nmax, uint32_t thrm1,
                                                  uint32_t d_lane, uint32_t m_lo, uint32_t m_hi, uint32_t t64) {
    uint64_t h0 = head;
    const uint32_t h_hi = (uint32_t)(h0 >> 32);
    // rare renormalisation cases in ONE unsigned compare: h_hi >= thr, or h_hi == 0 (needed for h0 < 2^31)
    if (__builtin_expect(h_hi - 1u >= thrm1, 0)) {
        const uint32_t thr = thrm1 + 1u;
        if (h_hi >= thr) {  // h0 >= nmax * ((L / nmax) << 32)
            ws_push(s, (uint32_t)h0);
            h0 >>= 32;
        }
        uint64_t q = h0 / nmax;
        uint32_t r = (uint32_t)(h0 - q * nmax);
        if (lt_2p31(h0)) q = (uint64_t)ws_pop(s) | (q << 32);  // the test is on h0 (codec.cpp:35)
        head = q;
        return to_v(rfl(r));
    }
    const uint32_t a0 = (uint32_t)h0, a1 = h_hi;
    // mulhi64(h0, m) = a1*m1 + hi(a1*m0) + hi(a0*m1) + carry(lo(a1*m0) + lo(a0*m1) + hi(a0*m0))
    const uint64_t t0 = (uint64_t)a0 * m_lo;
    const uint64_t t1 = (uint64_t)a0 * m_hi + (t0 >> 32);
    const uint64_t t2 = (uint64_t)a1 * m_lo + (uint32_t)t1;
    uint64_t qv = (uint64_t)a1 * m_hi + (t1 >> 32) + (t2 >> 32);
    uint32_t rv = a0 - (uint32_t)qv * d_lane;
    const bool fix = rv >= d_lane;
    rv = fix ? rv - d_lane : rv;
    qv = fix ? qv + 1 : qv;
    head = rl64((uint32_t)qv, (uint32_t)(qv >> 32), t64);
    return to_v(rl(rv, t64));
}
// codec.cpp:44-63 : lq = floor(2^31 / nmax)
__device__ __forceinline__ void ans_idx_push(uint64_t &head, WStack &s, uint32_t sym, uint32_t nmax, uint32_t lq) {
    uint64_t h0 = head;
    if ((uint32_t)(h0 >> 32) >= lq) {
        ws_push(s, (uint32_t)h0);
        h0 >>= 32;
    }
    uint64_t h = h0 * (uint64_t)nmax + sym;
    if (__builtin_expect(lt_2p31(h), 0)) h = (uint64_t)ws_pop(s) | (h << 32);
    head = h;
}

// precision rules.  Reference: (uint64_t)ceil(log2((int)max_id)) (custom_invlists_impl.cpp:163-164,
// altid_impl.cpp:124-125).  For 1 <= m < 2^31 that is bit_width(m - 1) exactly (log2 of a double is
// exact at powers of two and monotone in between); m == 0 gives precision 0 on x86-64.
__device__ __forceinline__ uint32_t precision_for(uint32_t max_id, int mode) {
    if (mode >= 0) return (uint32_t)mode;
    if (mode == -2) return max_id ? 32u - (uint32_t)__builtin_clz(max_id) : 0u;  // VIDC_PREC_EXACT
    return max_id > 1u ? 32u - (uint32_t)__builtin_clz(max_id - 1u) : 0u;        // VIDC_PREC_REFERENCE
}

}  // namespace dev
}  // namespace vidc
