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

#ifdef __HIPCC__
#include <hip/hip_runtime.h>
namespace {
// the same table built on the device (no host array, no PCIe crossing): chunks per list -> exclusive scan (scan.h)
// -> one wavefront per list fills its items
__global__ void k_count_chunks(const uint64_t *offsets, uint32_t nlist, uint32_t *cnt) {
    for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < nlist; l += gridDim.x * blockDim.x)
        cnt[l] = (uint32_t)((offsets[l + 1] - offsets[l] + vidc::CHUNK_IDS - 1) / vidc::CHUNK_IDS);
}
// items[item_off[l] + c] = (l, c * unit): one wavefront per list
__global__ void __launch_bounds__(64) k_fill_items(const uint64_t *item_off, uint32_t nlist, uint32_t unit, vidc::Chunk *out) {
    const uint32_t lane = (threadIdx.x & 63u);
    for (uint32_t l = blockIdx.x; l < nlist; l += gridDim.x) {
        const uint64_t o = item_off[l], n = item_off[l + 1] - o;
        for (uint64_t c = lane; c < n; c += 64) out[o + c] = vidc::Chunk{l, (uint32_t)(c * unit)};
    }
}
// the same with one THREAD per list: for objects whose lists hold one or two items each (graph rows: one batch per
// row) a wavefront per list is 10^6 workgroups writing 8 bytes
__global__ void k_fill_items_flat(const uint64_t *item_off, uint32_t nlist, uint32_t unit, vidc::Chunk *out) {
    for (uint32_t l = blockIdx.x * blockDim.x + threadIdx.x; l < nlist; l += gridDim.x * blockDim.x) {
        const uint64_t o = item_off[l], n = item_off[l + 1] - o;
        for (uint64_t c = 0; c < n; c++) out[o + c] = vidc::Chunk{l, (uint32_t)(c * unit)};
    }
}
// items of every list, by whichever kernel fits: lists with at most ~2 items each -> thread per list
inline void launch_fill_items(hipStream_t st, const uint64_t *item_off, uint32_t nlist, uint32_t unit, vidc::Chunk *out,
                              uint64_t total_items, uint32_t num_cu) {
    if (!nlist) return;
    if (total_items <= 2ull * nlist) {
        const uint32_t grid = (nlist + 255u) / 256u < num_cu * 16u ? (nlist + 255u) / 256u : num_cu * 16u;
        hipLaunchKernelGGL(k_fill_items_flat, dim3(grid), dim3(256), 0, st, item_off, nlist, unit, out);
    } else {
        const uint32_t grid = nlist < num_cu * 64u ? nlist : num_cu * 64u;
        hipLaunchKernelGGL(k_fill_items, dim3(grid), dim3(64), 0, st, item_off, nlist, unit, out);
    }
}
}  // namespace
#endif

