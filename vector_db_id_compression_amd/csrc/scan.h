// scan.h -- device exclusive scan of u32 counts into u64 offsets (out has n + 1 entries): one launch up to 8192
// elements, three small launches beyond.
// Keeps per-list counts (ANS words, edges, chunk / batch counts) on the device: with 10^6 lists the host prefix sum
// plus two PCIe crossings of the arrays cost more than the codec kernels.  Included by several translation units:
// everything here has internal linkage.
#pragma once
#include <algorithm>

#include "common.h"

namespace vidc {
namespace {

#define VIDC_SCAN_TILE 4096u  // elements per block = 256 threads x 16
__global__ void __launch_bounds__(256) k_scan_tile_sums(const uint32_t *in, uint32_t n, uint64_t *tile_sums) {
    __shared__ uint64_t part[4];
    const uint32_t base = blockIdx.x * VIDC_SCAN_TILE;
    uint64_t s = 0;
    for (uint32_t j = threadIdx.x; j < VIDC_SCAN_TILE; j += 256) s += base + j < n ? in[base + j] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((unsigned long long)s, o, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ void __launch_bounds__(256) k_scan_tiles(uint64_t *tile_sums, uint32_t ntiles) {  // one block, in place
    __shared__ uint64_t sh[256];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < ntiles; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint64_t v = i < ntiles ? tile_sums[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 256; o <<= 1) {
            const uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < ntiles) tile_sums[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_scan_apply(const uint32_t *in, uint32_t n, const uint64_t *tile_off, uint64_t *out) {
    __shared__ uint64_t sh[256];
    const uint32_t base = blockIdx.x * VIDC_SCAN_TILE + threadIdx.x * 16u;
    uint32_t v[16];
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        v[j] = base + j < n ? in[base + j] : 0u;
        s += v[j];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t acc = tile_off[blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (base + j <= n) out[base + j] = acc;  // (index n receives the total)
        acc += v[j];
    }
}
// up to 8192 elements: one block, one launch (the three-launch version costs ~17 us whatever n is and an Elias-Fano
// encode runs four scans; beyond a few elements per thread the single block is latency-bound and loses)
#define VIDC_SCAN_SMALL_MAX 8192u
__global__ void __launch_bounds__(1024) k_scan_small(const uint32_t *in, uint32_t n, uint64_t *out) {
    __shared__ uint64_t wsum[16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t E = (n + 1023u) / 1024u;  // consecutive elements per thread
    const uint32_t base = t * E;
    uint64_t s = 0;
    for (uint32_t j = 0; j < E; j++) s += base + j < n ? in[base + j] : 0u;
    uint64_t incl = s;  // inclusive scan inside the wavefront
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t v = __shfl_up((unsigned long long)incl, o, 64);
        if (lane >= (uint32_t)o) incl += v;
    }
    if (lane == 63u) wsum[wave] = incl;
    __syncthreads();
    uint64_t before = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 16; w++) {
        before += w < wave ? wsum[w] : 0;
        total += wsum[w];
    }
    uint64_t acc = before + incl - s;
    for (uint32_t j = 0; j < E; j++) {
        if (base + j < n) {
            out[base + j] = acc;
            acc += in[base + j];
        }
    }
    if (t == 0) out[n] = total;
}

// number of non-zero entries
__global__ void k_count_nonzero(const uint32_t *in, uint32_t n, unsigned long long *out) {
    unsigned long long c = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += in[i] != 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if ((threadIdx.x & 63u) == 0 && c) atomicAdd(out, c);
}

// sizes[i] = offsets[idx[i] + 1] - offsets[idx[i]]
template <typename I>
__global__ void k_gather_sizes(const uint64_t *offsets, const I *idx, uint64_t m, uint32_t *sizes) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint64_t l = idx ? (uint64_t)idx[i] : i;  // no index array: lists 0..m-1
        sizes[i] = (uint32_t)(offsets[l + 1] - offsets[l]);
    }
}
// host copy of the sizes of the requested lists, computed on the device (no nlist-sized array crosses PCIe)
template <typename I>
inline int fetch_sizes(::vidc_ctx *ctx, const uint64_t *d_offsets, const I *d_idx, uint64_t m, uint32_t *host_out) {
    if (!m) return VIDC_OK;
    Scratch s_sz;
    Pinned h_sz;
    VIDC_TRY(s_sz.get(ctx, m * 4));
    VIDC_TRY(h_sz.get(ctx, m * 4));
    hipLaunchKernelGGL(k_gather_sizes<I>, dim3((uint32_t)std::min<uint64_t>((m + 255) / 256, 4096)), dim3(256), 0,
                       ctx->stream, d_offsets, d_idx, m, s_sz.as<uint32_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipMemcpyAsync(h_sz.p, s_sz.p, m * 4, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    std::memcpy(host_out, h_sz.p, m * 4);
    return VIDC_OK;
}

// out[0..n] = exclusive prefix sums of in[0..n) on the context's stream (tmp: scratch for the tile sums)
inline int device_exscan(::vidc_ctx *ctx, const uint32_t *d_in, uint32_t n, uint64_t *d_out, Scratch &tmp) {
    if (n <= VIDC_SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, d_in, n, d_out);
        VIDC_HIP(hipGetLastError());
        return VIDC_OK;
    }
    const uint32_t ntiles = n / VIDC_SCAN_TILE + 1u;
    VIDC_TRY(tmp.get(ctx, (size_t)ntiles * 8));
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(ntiles), dim3(256), 0, ctx->stream, d_in, n, tmp.as<uint64_t>());
    hipLaunchKernelGGL(k_scan_tiles, dim3(1), dim3(256), 0, ctx->stream, tmp.as<uint64_t>(), ntiles);
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(256), 0, ctx->stream, d_in, n, tmp.as<uint64_t>(), d_out);
    VIDC_HIP(hipGetLastError());
    return VIDC_OK;
}


// up to four scans of the same length in the launches of one (blockIdx.y = which): an Elias-Fano object built from graph
// rows needs the prefix sums of four per-row counts, and at 10^6 rows a three-launch scan is mostly launch latency
struct Scan4 {
    const uint32_t *in[4];
    uint64_t *out[4];
};
__global__ void __launch_bounds__(256) k_scan4_tile_sums(Scan4 a, uint32_t n, uint64_t *tile_sums, uint32_t ntiles) {
    __shared__ uint64_t part[4];
    const uint32_t *in = a.in[blockIdx.y];
    const uint32_t base = blockIdx.x * VIDC_SCAN_TILE;
    uint64_t s = 0;
    for (uint32_t j = threadIdx.x; j < VIDC_SCAN_TILE; j += 256) s += base + j < n ? in[base + j] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor((unsigned long long)s, o, 64);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[(size_t)blockIdx.y * ntiles + blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ void __launch_bounds__(256) k_scan4_tiles(uint64_t *tile_sums, uint32_t ntiles) {  // one block per scan
    __shared__ uint64_t sh[256];
    __shared__ uint64_t carry;
    uint64_t *ts = tile_sums + (size_t)blockIdx.x * ntiles;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < ntiles; b0 += 256) {
        const uint32_t i = b0 + threadIdx.x;
        const uint64_t v = i < ntiles ? ts[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t o = 1; o < 256; o <<= 1) {
            const uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < ntiles) ts[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry += sh[255];
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_scan4_apply(Scan4 a, uint32_t n, const uint64_t *tile_off, uint32_t ntiles) {
    __shared__ uint64_t sh[256];
    const uint32_t *in = a.in[blockIdx.y];
    uint64_t *out = a.out[blockIdx.y];
    const uint32_t base = blockIdx.x * VIDC_SCAN_TILE + threadIdx.x * 16u;
    uint32_t v[16];
    uint64_t s = 0;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        v[j] = base + j < n ? in[base + j] : 0u;
        s += v[j];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t o = 1; o < 256; o <<= 1) {
        const uint64_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint64_t acc = tile_off[(size_t)blockIdx.y * ntiles + blockIdx.x] + sh[threadIdx.x] - s;
#pragma unroll
    for (int j = 0; j < 16; j++) {
        if (base + j <= n) out[base + j] = acc;  // (index n receives the total)
        acc += v[j];
    }
}
// out[k][0..n] = exclusive prefix sums of in[k][0..n) for k < count (count <= 4) on the context's stream
inline int device_exscan4(::vidc_ctx *ctx, const Scan4 &a, uint32_t count, uint32_t n, Scratch &tmp) {
    if (n <= VIDC_SCAN_SMALL_MAX) {
        for (uint32_t k = 0; k < count; k++)
            hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, ctx->stream, a.in[k], n, a.out[k]);
        VIDC_HIP(hipGetLastError());
        return VIDC_OK;
    }
    const uint32_t ntiles = n / VIDC_SCAN_TILE + 1u;
    VIDC_TRY(tmp.get(ctx, (size_t)ntiles * count * 8));
    hipLaunchKernelGGL(k_scan4_tile_sums, dim3(ntiles, count), dim3(256), 0, ctx->stream, a, n, tmp.as<uint64_t>(), ntiles);
    hipLaunchKernelGGL(k_scan4_tiles, dim3(count), dim3(256), 0, ctx->stream, tmp.as<uint64_t>(), ntiles);
    hipLaunchKernelGGL(k_scan4_apply, dim3(ntiles, count), dim3(256), 0, ctx->stream, a, n, tmp.as<const uint64_t>(), ntiles);
    VIDC_HIP(hipGetLastError());
    return VIDC_OK;
}

}  // namespace
}  // namespace vidc
