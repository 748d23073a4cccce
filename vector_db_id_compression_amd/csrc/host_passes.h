// host_passes.h -- the host's passes over the CSR offsets of an encode call (no HIP in here: tests/host_passes_test.cpp builds it with g++).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "../../include/vidc.h"

#if !defined(__HIP_DEVICE_COMPILE__) && (defined(__x86_64__) || defined(_M_X64))
#include <immintrin.h>
#define VIDC_X86_HOST_COMMON 1
#endif

namespace vidc {

// One pass over the CSR offsets of an Elias-Fano / packed-bits encode call: longest list, chunks of 2^unit_shift ids, and for packed
// bits the byte and word counts of every list's `bits`-wide fields.  The scalar loops were 1-2 ns per list -- more than the kernels of a
// 65 536-list call, 1-2 ms of a 10^6-list one; with AVX2 four lists per instruction.  `wide` = some length of 2^32 or more (or offsets
// that decrease): the caller then takes its own scalar loop, which validates and reports list by list.
struct LengthsPass {
    uint64_t max_n = 0, nchunks = 0, bytes = 0, words = 0;  // bytes = sum (n*bits+7)/8, words = sum ((n*bits+63)/64 + 1)
    bool wide = false;
};
inline void lengths_pass_scalar(const uint64_t *off, uint64_t la, uint64_t lb, uint32_t unit_shift, uint32_t bits, LengthsPass &x) {
    const uint64_t unit_m1 = (1ull << unit_shift) - 1ull;
    for (uint64_t l = la; l < lb; l++) {
        const uint64_t n = off[l + 1] - off[l];
        if (n >> 32) { x.wide = true; continue; }
        x.max_n = n > x.max_n ? n : x.max_n;
        x.nchunks += (n + unit_m1) >> unit_shift;
        const uint64_t nb = n * bits;
        x.bytes += (nb + 7) >> 3;
        x.words += ((nb + 63) >> 6) + 1;
    }
}
#ifdef VIDC_X86_HOST_COMMON
__attribute__((target("avx2"))) inline void lengths_pass_avx2(const uint64_t *off, uint64_t nlist, uint32_t unit_shift, uint32_t bits,
                                                               LengthsPass &x) {
    uint64_t l = 0;
    __m256i vor = _mm256_setzero_si256(), vmax = _mm256_setzero_si256(), vch = _mm256_setzero_si256(), vby = _mm256_setzero_si256(),
            vwo = _mm256_setzero_si256();
    const __m256i um1 = _mm256_set1_epi64x((long long)((1ull << unit_shift) - 1ull)), vbits = _mm256_set1_epi64x((long long)bits);
    const __m256i c7 = _mm256_set1_epi64x(7), c63 = _mm256_set1_epi64x(63);
    const __m128i sh_unit = _mm_cvtsi32_si128((int)unit_shift);
    for (; l + 4 <= nlist; l += 4) {
        const __m256i o0 = _mm256_loadu_si256((const __m256i *)(off + l)), o1 = _mm256_loadu_si256((const __m256i *)(off + l + 1));
        const __m256i n = _mm256_sub_epi64(o1, o0);
        vor = _mm256_or_si256(vor, n);
        vmax = _mm256_blendv_epi8(vmax, n, _mm256_cmpgt_epi64(n, vmax));  // (only used when no length has bits above 2^32)
        vch = _mm256_add_epi64(vch, _mm256_srl_epi64(_mm256_add_epi64(n, um1), sh_unit));
        const __m256i nb = _mm256_mul_epu32(n, vbits);  // n < 2^32, bits <= 64: the product fits 64 bits
        vby = _mm256_add_epi64(vby, _mm256_srli_epi64(_mm256_add_epi64(nb, c7), 3));
        vwo = _mm256_add_epi64(vwo, _mm256_srli_epi64(_mm256_add_epi64(nb, c63), 6));
    }
    alignas(32) uint64_t t[4];
    _mm256_store_si256((__m256i *)t, vor);
    if ((t[0] | t[1] | t[2] | t[3]) >> 32) { x.wide = true; return; }
    _mm256_store_si256((__m256i *)t, vmax);
    x.max_n = std::max(std::max(t[0], t[1]), std::max(t[2], t[3]));
    _mm256_store_si256((__m256i *)t, vch);
    x.nchunks = t[0] + t[1] + t[2] + t[3];
    _mm256_store_si256((__m256i *)t, vby);
    x.bytes = t[0] + t[1] + t[2] + t[3];
    _mm256_store_si256((__m256i *)t, vwo);
    x.words = t[0] + t[1] + t[2] + t[3] + l;  // (+1 padding word per list)
    lengths_pass_scalar(off, l, nlist, unit_shift, bits, x);
}
#endif
inline LengthsPass lengths_pass(const uint64_t *off, uint64_t nlist, uint32_t unit_shift, uint32_t bits) {
    LengthsPass x;
#ifdef VIDC_X86_HOST_COMMON
    static const bool avx2 = __builtin_cpu_supports("avx2") && !std::getenv("VIDC_NO_AVX2");
    if (avx2) { lengths_pass_avx2(off, nlist, unit_shift, bits, x); return x; }
#endif
    lengths_pass_scalar(off, 0, nlist, unit_shift, bits, x);
    return x;
}

// The one pass over the offsets of a call that is not worth threads (below PAR_MIN_LISTS lists): lengths' extremes, non-empty lists,
// "longest first", "some length beyond the limit or offsets that decrease" as ONE flag, and the staging copy for the upload.  Nothing
// of a 65 536-list call can be launched before this is done, and the scalar loop is ~0.9 ns per list (60 us); four lists per AVX2
// instruction when the host has them.
struct OffsetsPass { uint64_t nonempty = 0, max_n = 0, min_n = ~0ull, prev = ~0ull; bool desc = true, bad = false; };
inline void offsets_pass_scalar(const uint64_t *offsets, uint64_t la, uint64_t lb, uint64_t *hp, OffsetsPass &x) {
    uint64_t nonempty_ = x.nonempty, mx = x.max_n, mn = x.min_n, prev = x.prev, o0 = offsets[la];
    bool desc = x.desc, bad_ = x.bad;
    hp[la] = o0;
    for (uint64_t l = la; l < lb; l++) {
        const uint64_t o1 = offsets[l + 1];
        hp[l + 1] = o1;
        const uint64_t n = o1 - o0;  // (wraps when the offsets decrease: caught as "too long")
        o0 = o1;
        bad_ |= n > VIDC_ROC_MAX_LIST;
        nonempty_ += n != 0;
        mx = n > mx ? n : mx;
        mn = n < mn ? n : mn;
        desc &= n <= prev;  // (equal-sized lists, or an index stored longest list first)
        prev = n;
    }
    x.bad = bad_; x.nonempty = nonempty_; x.max_n = mx; x.min_n = mn; x.desc = desc; x.prev = prev;
}
#ifdef VIDC_X86_HOST_COMMON
__attribute__((target("avx2"))) inline void offsets_pass_avx2(const uint64_t *offsets, uint64_t nlist, uint64_t *hp, OffsetsPass &x) {
    // lengths n[l] = o[l+1] - o[l], four at a time; a length of 2^32 or more (a wrapped difference included) only raises `bad` --
    // the caller then reports the first offending list from a scalar pass -- so the signed 64-bit compares below see small values
    // whenever their results are used
    uint64_t l = 0;
    if (nlist >= 9) {
        offsets_pass_scalar(offsets, 0, 1, hp, x);  // list 0 (gives the vector loop a predecessor)
        l = 1;
        __m256i vor = _mm256_setzero_si256(), vmax = _mm256_set1_epi64x((long long)x.max_n), vmin = _mm256_set1_epi64x((long long)x.min_n);
        __m256i vzero = _mm256_setzero_si256(), vasc = _mm256_setzero_si256();
        const __m256i zero = _mm256_setzero_si256();
        for (; l + 4 <= nlist; l += 4) {
            const __m256i om = _mm256_loadu_si256((const __m256i *)(offsets + l - 1));
            const __m256i o0 = _mm256_loadu_si256((const __m256i *)(offsets + l));
            const __m256i o1 = _mm256_loadu_si256((const __m256i *)(offsets + l + 1));
            _mm256_storeu_si256((__m256i *)(hp + l + 1), o1);
            const __m256i n = _mm256_sub_epi64(o1, o0), np = _mm256_sub_epi64(o0, om);
            vor = _mm256_or_si256(vor, n);
            vmax = _mm256_blendv_epi8(vmax, n, _mm256_cmpgt_epi64(n, vmax));
            vmin = _mm256_blendv_epi8(vmin, n, _mm256_cmpgt_epi64(vmin, n));
            vzero = _mm256_sub_epi64(vzero, _mm256_cmpeq_epi64(n, zero));   // (+1 per empty list)
            vasc = _mm256_or_si256(vasc, _mm256_cmpgt_epi64(n, np));         // some list longer than its predecessor
        }
        alignas(32) uint64_t t[4];
        _mm256_store_si256((__m256i *)t, vor);
        const uint64_t orall = t[0] | t[1] | t[2] | t[3];
        _mm256_store_si256((__m256i *)t, vmax);
        uint64_t mx = std::max(std::max(t[0], t[1]), std::max(t[2], t[3]));
        _mm256_store_si256((__m256i *)t, vmin);
        uint64_t mn = std::min(std::min(t[0], t[1]), std::min(t[2], t[3]));
        _mm256_store_si256((__m256i *)t, vzero);
        const uint64_t zeros = t[0] + t[1] + t[2] + t[3];
        _mm256_store_si256((__m256i *)t, vasc);
        const bool asc = (t[0] | t[1] | t[2] | t[3]) != 0;
        const bool wide = (orall >> 32) != 0;
        x.bad |= wide || mx > VIDC_ROC_MAX_LIST;
        x.nonempty += (l - 1) - zeros;
        x.max_n = wide ? ~0ull : mx;  // (only read when the call is not rejected)
        x.min_n = mn;
        x.desc = x.desc && !asc;
        x.prev = offsets[l] - offsets[l - 1];
    }
    if (l < nlist || nlist == 0) {
        if (l) {  // the tail continues from the state of the vector loop (hp[l] is written already: the scalar pass rewrites it)
            OffsetsPass y = x;
            offsets_pass_scalar(offsets, l, nlist, hp, y);
            x = y;
        } else {
            offsets_pass_scalar(offsets, 0, nlist, hp, x);
        }
    }
}
#endif
inline void offsets_pass(const uint64_t *offsets, uint64_t nlist, uint64_t *hp, OffsetsPass &x) {
#ifdef VIDC_X86_HOST_COMMON
    static const bool avx2 = __builtin_cpu_supports("avx2") && !std::getenv("VIDC_NO_AVX2");
    if (avx2) { offsets_pass_avx2(offsets, nlist, hp, x); return; }
#endif
    offsets_pass_scalar(offsets, 0, nlist, hp, x);
}

}  // namespace vidc
