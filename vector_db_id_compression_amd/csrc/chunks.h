// chunks.h -- work decomposition for the bandwidth-bound kernels: every list is cut into chunks of CHUNK_IDS ids;
// one wavefront per chunk knows its list without searching the CSR offsets.  CHUNK_IDS * bits is a multiple of
// 64 for every bit width, so a chunk owns whole 64-bit words of its list's bit stream (no atomics).
#pragma once
#include <cstdint>
#include <vector>

namespace vidc {

constexpr uint32_t CHUNK_IDS = 512;

struct Chunk {
    uint32_t list;
    uint32_t start;  // first id of the chunk inside its list (multiple of CHUNK_IDS)
};

inline std::vector<Chunk> build_chunks(const std::vector<uint64_t> &offsets) {
    std::vector<Chunk> c;
    const size_t nlist = offsets.size() - 1;
    c.reserve(nlist + offsets[nlist] / CHUNK_IDS);
    for (size_t l = 0; l < nlist; l++) {
        const uint64_t n = offsets[l + 1] - offsets[l];
        for (uint64_t s = 0; s < n; s += CHUNK_IDS) c.push_back(Chunk{(uint32_t)l, (uint32_t)s});
    }
    return c;
}

}  // namespace vidc
