// ubench_sel5.hip -- what delays v_readlane on a lone wavefront: preceding SALU batches, outstanding LDS operations (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define TIMED(name, ...)                                                                  \
    __global__ void __launch_bounds__(64) name(uint64_t *out, uint32_t *buf) {             \
        __shared__ uint32_t lds[2048];                                                    \
        for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = buf[i & 1023];              \
        __syncthreads();                                                                  \
        uint32_t s = buf[0], v = buf[threadIdx.x], lane = threadIdx.x;                    \
        (void)lane;                                                                       \
        uint64_t t0, t1;                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)); \
        __VA_ARGS__                                                                        \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)); \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s + v; }                       \
    }
#define R ".rept 256\n"
#define CL : "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29"
#define V(n) " v_add_u32 v" #n ", 1, v" #n "\n"
#define V4 V(104) V(105) V(106) V(107)
#define V8 V4 V(108) V(109) V(110) V(111)
#define S(n) " s_add_u32 s" #n ", s" #n ", 1\n"
#define S2 S(24) S(25)
#define S5 S2 S(26) S(27) S(28)
#define S9 S5 S(24) S(25) S(26) S(27)
#define T(name, BODY) TIMED(name, asm volatile(R BODY ".endr" : "+v"(v) : "v"(lane) CL);)
T(base_v8, V8)
T(base_v8_s5, V8 S5)
T(base_v8_s9, V8 S9)
// SALU batch then readlane (immediate lane), then 8 valu
T(s0_rl, " v_readlane_b32 s20, %0, 3\n" V8)
T(s2_rl, S2 " v_readlane_b32 s20, %0, 3\n" V8)
T(s5_rl, S5 " v_readlane_b32 s20, %0, 3\n" V8)
T(s9_rl, S9 " v_readlane_b32 s20, %0, 3\n" V8)
T(s5_rl3, S5 " v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 4\n v_readlane_b32 s22, %0, 5\n" V8)
// SALU batch then plain VALU
T(s5_v, S5 V(112) V8)
// SALU batch then readlane with SALU lane select
T(s5_rls, " s_and_b32 s21, s24, 63\n" S5 " v_readlane_b32 s20, %0, s21\n" V8)
// ds_write then k valu then readlane
T(w_0_rl, " ds_write_b32 %1, %0\n v_readlane_b32 s20, %0, 3\n" V8)
T(w_4_rl, " ds_write_b32 %1, %0\n" V4 " v_readlane_b32 s20, %0, 3\n" V4)
T(w_8_rl, " ds_write_b32 %1, %0\n" V8 " v_readlane_b32 s20, %0, 3\n")
T(w_8_v, " ds_write_b32 %1, %0\n" V8 V(112))
T(r_4_rl, " ds_read_b32 v113, %1\n" V4 " v_readlane_b32 s20, %0, 3\n" V4 " s_waitcnt lgkmcnt(0)\n")
T(r_4_v, " ds_read_b32 v113, %1\n" V4 V(112) V4 " s_waitcnt lgkmcnt(0)\n")
T(r_4_s, " ds_read_b32 v113, %1\n" V4 S(24) V4 " s_waitcnt lgkmcnt(0)\n")
// readlane -> writelane etc
T(wl_after_s, S5 " s_mov_b32 m0, 3\n s_nop 0\n v_writelane_b32 v100, s24, m0\n" V8)
// v_cmp then readlane (independent)
T(cmp_rl, " v_cmp_le_u32 vcc, %1, %0\n v_readlane_b32 s20, %0, 3\n" V8)
// readlane followed by SALU batch after 3 valu
T(rl_v3_s5, " v_readlane_b32 s20, %0, 3\n" V(112) V(113) V(104) S5 V4)
typedef void (*kern_t)(uint64_t *, uint32_t *);
struct Item { const char *name; kern_t k; int instr; };
#define IT(n, i) {#n, n, i}
int main() {
    uint64_t *d_out; uint32_t *d_buf;
    hipMalloc(&d_out, 64); hipMalloc(&d_buf, 4096); hipMemset(d_buf, 0, 4096);
    Item items[] = { IT(base_v8, 8), IT(base_v8_s5, 13), IT(base_v8_s9, 17), IT(s0_rl, 9), IT(s2_rl, 11), IT(s5_rl, 14), IT(s9_rl, 18), IT(s5_rl3, 16), IT(s5_v, 14), IT(s5_rls, 15),
        IT(w_0_rl, 10), IT(w_4_rl, 10), IT(w_8_rl, 10), IT(w_8_v, 10), IT(r_4_rl, 11), IT(r_4_v, 11), IT(r_4_s, 11), IT(wl_after_s, 16), IT(cmp_rl, 10), IT(rl_v3_s5, 13) };
    for (auto &it : items) {
        uint64_t h[2];
        for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(it.k, dim3(1), dim3(64), 0, 0, d_out, d_buf); hipDeviceSynchronize(); }
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-14s %7.2f cycles = %2d instr x 4 + %6.2f\n", it.name, (double)h[0] / 256, it.instr, (double)h[0] / 256 - 4.0 * it.instr);
    }
    return 0;
}
