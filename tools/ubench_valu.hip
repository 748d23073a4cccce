// ubench_valu.hip -- VALU issue rate of straight-line code by encoding size / dependency / register index, for one
// wavefront and for one wavefront per SIMD on the whole chip (dev tool; the rank loop of k_roc_decode_lane_reg).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_valu.hip -o /tmp/ubench_valu && /tmp/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define TIMED(name, body)                                                                                  \
    __global__ void __launch_bounds__(64) name(uint64_t *out, uint32_t *buf) {                             \
        uint32_t x = buf[threadIdx.x], a0 = 0, a1 = 0, a2 = 0, a3 = 0, t0 = x, t1 = x + 1, t2 = x + 2, t3 = x + 3; \
        uint64_t T0, T1;                                                                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(T0));     \
        for (int it = 0; it < 16; it++) { body }                                                           \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(T1));     \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = T1 - T0;                                         \
        if (a0 + a1 + a2 + a3 + t0 + t1 + t2 + t3 == 0x12345) out[1] = 1;                                  \
    }
#define OPS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3) : "v"(x)

// 8 instructions per repetition, 256 repetitions, 16 iterations = 32768 instructions
TIMED(k_sub_e32_indep, asm volatile(".rept 256\n v_sub_u32_e32 %4, %0, %8\n v_sub_u32_e32 %5, %1, %8\n v_sub_u32_e32 %6, %2, %8\n v_sub_u32_e32 %7, %3, %8\n"
                                    "v_sub_u32_e32 %4, %0, %8\n v_sub_u32_e32 %5, %1, %8\n v_sub_u32_e32 %6, %2, %8\n v_sub_u32_e32 %7, %3, %8\n .endr" : OPS);)
TIMED(k_sub_e64_indep, asm volatile(".rept 256\n v_sub_u32_e64 %4, %0, %8\n v_sub_u32_e64 %5, %1, %8\n v_sub_u32_e64 %6, %2, %8\n v_sub_u32_e64 %7, %3, %8\n"
                                    "v_sub_u32_e64 %4, %0, %8\n v_sub_u32_e64 %5, %1, %8\n v_sub_u32_e64 %6, %2, %8\n v_sub_u32_e64 %7, %3, %8\n .endr" : OPS);)
TIMED(k_alignbit_4chains, asm volatile(".rept 256\n v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %6, 31\n v_alignbit_b32 %3, %3, %7, 31\n"
                                       "v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %6, 31\n v_alignbit_b32 %3, %3, %7, 31\n .endr" : OPS);)
TIMED(k_alignbit_1chain, asm volatile(".rept 256\n v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %0, %0, %5, 31\n v_alignbit_b32 %0, %0, %6, 31\n v_alignbit_b32 %0, %0, %7, 31\n"
                                      "v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %0, %0, %5, 31\n v_alignbit_b32 %0, %0, %6, 31\n v_alignbit_b32 %0, %0, %7, 31\n .endr" : OPS);)
TIMED(k_rank_pattern, asm volatile(".rept 256\n v_sub_u32_e32 %4, %0, %8\n v_sub_u32_e32 %5, %1, %8\n v_sub_u32_e32 %6, %2, %8\n v_sub_u32_e32 %7, %3, %8\n"
                                   "v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %6, 31\n v_alignbit_b32 %3, %3, %7, 31\n .endr" : OPS);)
// the same 8 instructions as a LOOP body (no .rept): code that stays in the instruction buffer / cache
TIMED(k_rank_pattern_loop, for (int j = 0; j < 256; j++) asm volatile("v_sub_u32_e32 %4, %0, %8\n v_sub_u32_e32 %5, %1, %8\n v_sub_u32_e32 %6, %2, %8\n v_sub_u32_e32 %7, %3, %8\n"
                                   "v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %1, %1, %5, 31\n v_alignbit_b32 %2, %2, %6, 31\n v_alignbit_b32 %3, %3, %7, 31\n" : OPS);)
TIMED(k_cmp_addc_sgpr, asm volatile(".rept 256\n v_cmp_lt_u32_e64 s[20:21], %0, %8\n v_cmp_lt_u32_e64 s[22:23], %1, %8\n v_cmp_lt_u32_e64 s[24:25], %2, %8\n v_cmp_lt_u32_e64 s[26:27], %3, %8\n"
                                    "v_addc_co_u32_e64 %4, s[20:21], %4, 0, s[20:21]\n v_addc_co_u32_e64 %4, s[22:23], %4, 0, s[22:23]\n v_addc_co_u32_e64 %4, s[24:25], %4, 0, s[24:25]\n v_addc_co_u32_e64 %4, s[26:27], %4, 0, s[26:27]\n .endr"
                                    : OPS : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
TIMED(k_cmp_vcc_addc_nop, asm volatile(".rept 512\n v_cmp_lt_u32_e32 vcc, %0, %8\n s_nop 1\n v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n v_cmp_lt_u32_e32 vcc, %1, %8\n s_nop 1\n v_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n .endr" : OPS : "vcc");)

typedef void (*kern_t)(uint64_t *, uint32_t *);
struct Item { const char *name; kern_t k; double n; };

int main() {
    uint64_t *d_out; uint32_t *d_buf;
    hipMalloc(&d_out, 64); hipMalloc(&d_buf, 4096);
    hipMemset(d_buf, 1, 4096);
    Item items[] = {{"v_sub e32 x4 independent", k_sub_e32_indep, 32768}, {"v_sub e64 x4 independent", k_sub_e64_indep, 32768},
                    {"v_alignbit 4 chains", k_alignbit_4chains, 32768}, {"v_alignbit 1 chain", k_alignbit_1chain, 32768},
                    {"rank pattern 4 sub + 4 alignbit (straight line)", k_rank_pattern, 32768},
                    {"rank pattern as a loop body", k_rank_pattern_loop, 32768},
                    {"v_cmp->sgpr x4 + v_addc x4", k_cmp_addc_sgpr, 32768},
                    {"v_cmp vcc, nop, v_addc (compiler's form, per 3 instr)", k_cmp_vcc_addc_nop, 16 * 512 * 2}};
    for (int grid : {1, 1024, 2048}) {
        printf("---- %d wavefront(s)\n", grid);
        for (auto &it : items) {
            uint64_t h[2] = {0, 0};
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                for (int q = 0; q < 20; q++) hipLaunchKernelGGL(it.k, dim3(grid), dim3(64), 0, 0, d_out, d_buf);
                hipEventRecord(e1, 0);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
            // wall time of 20 back-to-back launches: ns per instruction of ONE wavefront (all run concurrently up to 1/SIMD)
            const double waves_per_simd = grid <= 1024 ? 1.0 : grid / 1024.0;
            printf("%-58s %6.2f ticks per instruction, %6.2f ns wall per instruction per SIMD slot\n", it.name, (double)h[0] / it.n,
                   1e6 * ms / 20.0 / it.n / waves_per_simd);
        }
    }
    return 0;
}
