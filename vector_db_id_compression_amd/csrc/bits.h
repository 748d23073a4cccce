// bits.h -- device helpers shared by the bandwidth-bound bit-packing kernels (packed bits, Elias-Fano).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vidc {
namespace dev {

// largest l with off[l] <= g  (off non-decreasing, off[0] == 0, g < off[n])
__device__ __forceinline__ uint32_t find_list(const uint64_t *off, uint32_t n, uint64_t g) {
    uint32_t lo = 0, hi = n;  // invariant: off[lo] <= g < off[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// `bits`-wide field at bit position pos of an LSB-first 64-bit word stream (one padding word must follow)
__device__ __forceinline__ uint64_t read_bits(const uint64_t *words, uint64_t pos, uint32_t bits) {
    if (!bits) return 0;
    const uint64_t w = pos >> 6;
    const uint32_t sh = (uint32_t)(pos & 63);
    uint64_t v = words[w] >> sh;
    if (sh + bits > 64) v |= words[w + 1] << (64 - sh);
    return bits == 64 ? v : (v & ((1ull << bits) - 1ull));
}

// the 64-bit word `w_in_list` of a stream that packs src[i] (i < n) at bit i*bits: gathered by ONE owner
// thread, so no atomics are needed.  flags bit 0 is raised when a value does not fit / exceeds id_limit.
template <bool CHECK>
__device__ __forceinline__ uint64_t gather_word(const uint64_t *src, uint64_t n, uint64_t w_in_list, uint32_t bits,
                                                uint64_t keep_mask, uint64_t id_limit, uint32_t *err) {
    uint64_t out = 0;
    if (!bits) return 0;
    const uint64_t bit0 = w_in_list * 64;
    for (uint64_t i = bit0 / bits; i < n; i++) {
        const uint64_t pos = i * bits;
        if (pos >= bit0 + 64) break;
        uint64_t v = src[i];
        if (CHECK) {
            if (v >= id_limit || (bits < 64 && (v >> bits))) atomicOr(err, 1u);
        }
        v &= keep_mask;
        if (pos >= bit0) out |= v << (pos - bit0);
        else out |= v >> (bit0 - pos);
    }
    return out;
}

}  // namespace dev
}  // namespace vidc
