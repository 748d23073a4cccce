// ubench_memlat.hip -- latency of a dependent global load on a lone wavefront: vector (global_load_dword) pointer
// chases over buffers that fit L2 (1 MiB), the MALL (64 MiB) or neither (1 GiB) (dev tool).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_memlat.hip -o tools/_bin/ubench_memlat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
#include <numeric>
#include <algorithm>

__global__ void __launch_bounds__(64) k_vec(const uint32_t *buf, uint32_t steps, uint64_t *out) {
    uint32_t p = 0;
    uint64_t t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (uint32_t i = 0; i < steps; i++) p = __builtin_nontemporal_load(&buf[p]) ;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p; }
}
__global__ void __launch_bounds__(64) k_vec_plain(const uint32_t *buf, uint32_t steps, uint64_t *out) {
    uint32_t p = 0;
    uint64_t t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (uint32_t i = 0; i < steps; i++) p = buf[p];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p; }
}
int main() {
    uint64_t *d_out;
    hipMalloc(&d_out, 64);
    for (size_t words : {size_t(1) << 16, size_t(1) << 18, size_t(1) << 24, size_t(1) << 28}) {
        // one random cycle through the buffer (stride >= a cache line on average)
        std::vector<uint32_t> perm(words), buf(words);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937_64 g(7);
        std::shuffle(perm.begin() + 1, perm.end(), g);
        for (size_t i = 0; i + 1 < words; i++) buf[perm[i]] = perm[i + 1];
        buf[perm[words - 1]] = perm[0];
        uint32_t *d;
        hipMalloc(&d, words * 4);
        hipMemcpy(d, buf.data(), words * 4, hipMemcpyHostToDevice);
        const uint32_t steps = 20000;
        uint64_t h[2];
        const char *names[2] = {"vector, nontemporal", "vector"};
        for (int k = 0; k < 2; k++) {
            for (int rep = 0; rep < 2; rep++) {
                if (k == 0) hipLaunchKernelGGL(k_vec, dim3(1), dim3(64), 0, 0, d, steps, d_out);
                else hipLaunchKernelGGL(k_vec_plain, dim3(1), dim3(64), 0, 0, d, steps, d_out);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
            printf("%8.2f MiB buffer, %-20s %7.1f ticks per dependent load (%.0f ns at 2.4 GHz)\n", words * 4 / 1048576.0, names[k],
                   (double)h[0] / steps, (double)h[0] / steps / 2.4);
        }
        hipFree(d);
    }
    return 0;
}
