// rows_csr.h -- graph rows wider than the lane-per-row kernels take (K > 64: NSG128 / NSG256 graphs): the int32 [N, K]
// rows (-1 terminated, altid_impl.cpp:61-68,110-117) become a CSR set of lists and go through the per-list codecs,
// which accept unsorted lists of any length.  The reference's graph containers take any K.
#pragma once
#include "common.h"
#include "scan.h"

namespace vidc {
namespace {

__global__ void k_rows_count(const int32_t *rows, uint64_t N, uint32_t K, uint32_t *cnt) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t n = 0;
        while (n < K && rows[i * K + n] != -1) n++;  // edges before the first -1
        cnt[i] = n;
    }
}
// one wavefront per row: ids[off[i] + j] = rows[i][j]  (negative ids other than the terminator become >= 2^31: a domain
// error in the ROC encoder, like the rows flavour)
__global__ void __launch_bounds__(64) k_rows_to_ids(const int32_t *rows, uint64_t N, uint32_t K, const uint64_t *off,
                                                    uint64_t *ids) {
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t i = blockIdx.x; i < N; i += gridDim.x) {
        const uint64_t o = off[i], n = off[i + 1] - o;
        for (uint64_t j = lane; j < n; j += 64) ids[o + j] = (uint64_t)(uint32_t)rows[i * K + j];
    }
}

// -> host offsets[N + 1] and a device array of the ids in CSR order (s_ids)
inline int rows_to_csr(::vidc_ctx *ctx, uint64_t N, uint32_t K, const int32_t *d_rows, std::vector<uint64_t> &offsets,
                       Scratch &s_ids) {
    Scratch s_cnt, s_off, s_tmp;
    offsets.assign(N + 1, 0);
    if (!N) return VIDC_OK;
    VIDC_TRY(s_cnt.get(ctx, N * 4));
    VIDC_TRY(s_off.get(ctx, (N + 1) * 8));
    const uint32_t grid = (uint32_t)std::min<uint64_t>((N + 255) / 256, (uint64_t)ctx->num_cu * 16);
    hipLaunchKernelGGL(k_rows_count, dim3(grid), dim3(256), 0, ctx->stream, d_rows, N, K, s_cnt.as<uint32_t>());
    VIDC_TRY(device_exscan(ctx, s_cnt.as<uint32_t>(), (uint32_t)N, s_off.as<uint64_t>(), s_tmp));
    VIDC_HIP(hipMemcpyAsync(offsets.data(), s_off.p, (N + 1) * 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    VIDC_TRY(s_ids.get(ctx, (offsets[N] ? offsets[N] : 1) * 8));
    hipLaunchKernelGGL(k_rows_to_ids, dim3((uint32_t)std::min<uint64_t>(N, (uint64_t)ctx->num_cu * 64)), dim3(64), 0,
                       ctx->stream, d_rows, N, K, s_off.as<uint64_t>(), s_ids.as<uint64_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipStreamSynchronize(ctx->stream));  // s_off goes back to the pool
    return VIDC_OK;
}

}  // namespace
}  // namespace vidc
