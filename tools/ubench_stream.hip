// ubench_stream.hip -- which launch shape streams u64 arrays fastest on MI355X?  (dev tool; decides the work
// decomposition of the bandwidth-bound packed / Elias-Fano kernels)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_stream tools/ubench_stream.hip && /tmp/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Chunk { uint32_t list, start; };

__global__ void v0_copy(const uint64_t *in, uint64_t *out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) out[i] = in[i];
}
// one wave per 512-element chunk, direct indexing
__global__ void __launch_bounds__(64) v1_wave_chunk(const uint64_t *in, uint64_t *out, uint64_t nchunks) {
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint64_t base = c * 512;
#pragma unroll
        for (int r = 0; r < 8; r++) out[base + threadIdx.x + 64 * r] = in[base + threadIdx.x + 64 * r];
    }
}
// + chunk table and CSR offsets (dependent header loads), like the codec kernels
__global__ void __launch_bounds__(64) v2_wave_chunk_tab(const uint64_t *in, uint64_t *out, const Chunk *chunks,
                                                        const uint64_t *offsets, uint64_t nchunks) {
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const Chunk ch = chunks[c];
        const uint64_t base = offsets[ch.list] + ch.start;
        const uint64_t n = offsets[ch.list + 1] - offsets[ch.list];
        const uint32_t nc = (uint32_t)(n - ch.start < 512 ? n - ch.start : 512);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t i = threadIdx.x + 64 * r;
            if (i < nc) out[base + i] = in[base + i];
        }
    }
}
// 256-thread block, one chunk per wave of the block
__global__ void __launch_bounds__(256) v4_block4(const uint64_t *in, uint64_t *out, const Chunk *chunks,
                                                 const uint64_t *offsets, uint64_t nchunks) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (uint64_t c = (uint64_t)blockIdx.x * 4 + wv; c < nchunks; c += (uint64_t)gridDim.x * 4) {
        const Chunk ch = chunks[c];
        const uint64_t base = offsets[ch.list] + ch.start;
        const uint64_t n = offsets[ch.list + 1] - offsets[ch.list];
        const uint32_t nc = (uint32_t)(n - ch.start < 512 ? n - ch.start : 512);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t i = lane + 64 * r;
            if (i < nc) out[base + i] = in[base + i];
        }
    }
}
// two chunks in flight per wave (loads of both issued before the stores)
__global__ void __launch_bounds__(64) v5_two_inflight(const uint64_t *in, uint64_t *out, const Chunk *chunks,
                                                      const uint64_t *offsets, uint64_t nchunks) {
    for (uint64_t c = (uint64_t)blockIdx.x * 2; c < nchunks; c += (uint64_t)gridDim.x * 2) {
        uint64_t v[16];
        uint64_t base[2];
        uint32_t nc[2];
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint64_t cc = c + k < nchunks ? c + k : c;
            const Chunk ch = chunks[cc];
            base[k] = offsets[ch.list] + ch.start;
            const uint64_t n = offsets[ch.list + 1] - offsets[ch.list];
            nc[k] = c + k < nchunks ? (uint32_t)(n - ch.start < 512 ? n - ch.start : 512) : 0u;
        }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t i = threadIdx.x + 64 * r;
                v[k * 8 + r] = i < nc[k] ? in[base[k] + i] : 0;
            }
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint32_t i = threadIdx.x + 64 * r;
                if (i < nc[k]) out[base[k] + i] = v[k * 8 + r];
            }
    }
}

int main() {
    const uint64_t nlist = 65536, per = 1024, n = nlist * per;
    uint64_t *in, *out, *d_off;
    Chunk *d_ch;
    CK(hipMalloc(&in, n * 8)); CK(hipMalloc(&out, n * 8));
    CK(hipMemset(in, 1, n * 8)); CK(hipMemset(out, 0, n * 8));
    std::vector<uint64_t> off(nlist + 1);
    for (uint64_t l = 0; l <= nlist; l++) off[l] = l * per;
    std::vector<Chunk> ch;
    for (uint64_t l = 0; l < nlist; l++) for (uint64_t s = 0; s < per; s += 512) ch.push_back(Chunk{(uint32_t)l, (uint32_t)s});
    const uint64_t nchunks = ch.size();
    CK(hipMalloc(&d_off, off.size() * 8)); CK(hipMalloc(&d_ch, ch.size() * 8));
    CK(hipMemcpy(d_off, off.data(), off.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ch, ch.data(), ch.size() * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto &&launch) {
        for (int i = 0; i < 3; i++) launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; i++) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.3f ms  %6.2f TB/s (read+write)\n", name, ms / 10, 2.0 * n * 8 / (ms / 10 * 1e-3) / 1e12);
    };
    run("v0 grid-stride copy 256x(256*CU*8)", [&] { hipLaunchKernelGGL(v0_copy, dim3(256 * 8 * 8), dim3(256), 0, 0, in, out, n); });
    run("v1 wave/chunk grid=nchunks", [&] { hipLaunchKernelGGL(v1_wave_chunk, dim3(nchunks), dim3(64), 0, 0, in, out, nchunks); });
    run("v1 wave/chunk grid=CU*256", [&] { hipLaunchKernelGGL(v1_wave_chunk, dim3(65536), dim3(64), 0, 0, in, out, nchunks); });
    run("v1 wave/chunk grid=CU*32", [&] { hipLaunchKernelGGL(v1_wave_chunk, dim3(8192), dim3(64), 0, 0, in, out, nchunks); });
    run("v2 + chunk table grid=nchunks", [&] { hipLaunchKernelGGL(v2_wave_chunk_tab, dim3(nchunks), dim3(64), 0, 0, in, out, d_ch, d_off, nchunks); });
    run("v2 + chunk table grid=CU*256", [&] { hipLaunchKernelGGL(v2_wave_chunk_tab, dim3(65536), dim3(64), 0, 0, in, out, d_ch, d_off, nchunks); });
    run("v2 + chunk table grid=CU*32", [&] { hipLaunchKernelGGL(v2_wave_chunk_tab, dim3(8192), dim3(64), 0, 0, in, out, d_ch, d_off, nchunks); });
    run("v4 256-thr blocks grid=nchunks/4", [&] { hipLaunchKernelGGL(v4_block4, dim3(nchunks / 4), dim3(256), 0, 0, in, out, d_ch, d_off, nchunks); });
    run("v4 256-thr blocks grid=CU*8", [&] { hipLaunchKernelGGL(v4_block4, dim3(2048), dim3(256), 0, 0, in, out, d_ch, d_off, nchunks); });
    run("v5 two chunks in flight grid=CU*32", [&] { hipLaunchKernelGGL(v5_two_inflight, dim3(8192), dim3(64), 0, 0, in, out, d_ch, d_off, nchunks); });
    run("v5 two chunks in flight grid=n/2", [&] { hipLaunchKernelGGL(v5_two_inflight, dim3(nchunks / 2), dim3(64), 0, 0, in, out, d_ch, d_off, nchunks); });
    return 0;
}
