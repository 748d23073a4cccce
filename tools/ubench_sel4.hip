// ubench_sel4.hip -- distance rules between dependent instruction pairs on a lone wavefront (dev tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define TIMED(name, ...)                                                                  \
    __global__ void __launch_bounds__(64) name(uint64_t *out, uint32_t *buf) {             \
        uint32_t s = buf[0], v = buf[threadIdx.x], lane = threadIdx.x;                    \
        (void)lane;                                                                       \
        uint64_t t0, t1;                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)); \
        __VA_ARGS__                                                                        \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)); \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s + v; }                       \
    }
#define R ".rept 256\n"
#define CL : "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"
#define F1 " v_add_u32 v104, v104, 1\n"
#define F2 F1 " v_add_u32 v105, v105, 1\n"
#define F3 F2 " v_add_u32 v106, v106, 1\n"
#define F4 F3 " v_add_u32 v107, v107, 1\n"
#define F5 F4 " v_add_u32 v108, v108, 1\n"
#define F6 F5 " v_add_u32 v109, v109, 1\n"
#define F8 F6 " v_add_u32 v110, v110, 1\n v_add_u32 v111, v111, 1\n"
// producer P, k fillers, consumer C, then 8 fillers to isolate iterations
#define PAIR(name, P, FILL, C) TIMED(name, asm volatile(R P FILL C F8 ".endr" : "+v"(v) : "v"(lane) CL);)
// R1: VALU sgpr write -> independent SALU
PAIR(r1_rl_0, " v_readlane_b32 s20, %0, 3\n", "", " s_add_u32 s24, s24, 1\n")
PAIR(r1_rl_2, " v_readlane_b32 s20, %0, 3\n", F2, " s_add_u32 s24, s24, 1\n")
PAIR(r1_rl_4, " v_readlane_b32 s20, %0, 3\n", F4, " s_add_u32 s24, s24, 1\n")
PAIR(r1_rl_6, " v_readlane_b32 s20, %0, 3\n", F6, " s_add_u32 s24, s24, 1\n")
PAIR(r1_cmp_0, " v_cmp_le_u32 vcc, %1, %0\n", "", " s_add_u32 s24, s24, 1\n")
PAIR(r1_cmp_4, " v_cmp_le_u32 vcc, %1, %0\n", F4, " s_add_u32 s24, s24, 1\n")
PAIR(r1_cmpdep_0, " v_cmp_le_u32 vcc, %1, %0\n", "", " s_ff1_i32_b64 s24, vcc\n")
PAIR(r1_cmpdep_3, " v_cmp_le_u32 vcc, %1, %0\n", F3, " s_ff1_i32_b64 s24, vcc\n")
PAIR(r1_cmpdep_4, " v_cmp_le_u32 vcc, %1, %0\n", F4, " s_ff1_i32_b64 s24, vcc\n")
PAIR(r1_cmpdep_5, " v_cmp_le_u32 vcc, %1, %0\n", F5, " s_ff1_i32_b64 s24, vcc\n")
PAIR(r1_cmpdep_6, " v_cmp_le_u32 vcc, %1, %0\n", F6, " s_ff1_i32_b64 s24, vcc\n")
PAIR(r1_mad_0, " v_mad_u64_u32 v[100:101], s[26:27], %1, %1, v[102:103]\n", "", " s_add_u32 s24, s24, 1\n")
// R2: VALU vgpr write -> readlane of it
PAIR(r2_add_1, " v_add_u32 v100, %1, %0\n", F1, " v_readlane_b32 s20, v100, 3\n")
PAIR(r2_add_2, " v_add_u32 v100, %1, %0\n", F2, " v_readlane_b32 s20, v100, 3\n")
PAIR(r2_add_4, " v_add_u32 v100, %1, %0\n", F4, " v_readlane_b32 s20, v100, 3\n")
PAIR(r2_add64_1, " v_lshl_add_u64 v[100:101], v[102:103], 0, v[100:101]\n", F1, " v_readlane_b32 s20, v100, 3\n")
PAIR(r2_add64_4, " v_lshl_add_u64 v[100:101], v[102:103], 0, v[100:101]\n", F4, " v_readlane_b32 s20, v100, 3\n")
PAIR(r2_min_1, " v_min_u32 v100, %1, %0\n", F1, " v_readlane_b32 s20, v100, 3\n")
// R3: readlane sgpr -> VALU use
PAIR(r3_0, " v_readlane_b32 s20, %0, 3\n", " s_nop 0\n", " v_add_u32 v100, s20, v100\n")
PAIR(r3_1, " v_readlane_b32 s20, %0, 3\n", F1, " v_add_u32 v100, s20, v100\n")
PAIR(r3_3, " v_readlane_b32 s20, %0, 3\n", F3, " v_add_u32 v100, s20, v100\n")
// R4: SALU sgpr -> readlane lane select
PAIR(r4_0, " s_add_u32 s21, s21, 1\n s_and_b32 s20, s21, 63\n", "", " v_readlane_b32 s22, %0, s20\n")
PAIR(r4_2, " s_add_u32 s21, s21, 1\n s_and_b32 s20, s21, 63\n", F2, " v_readlane_b32 s22, %0, s20\n")
// R5: dpp after valu write
PAIR(r5_2, " v_add_u32 v100, %1, %0\n", F2, " v_add_u32_dpp v101, v100, v100 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
PAIR(r5_4, " v_add_u32 v100, %1, %0\n", F4, " v_add_u32_dpp v101, v100, v100 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
// R6: salu -> valu operand
PAIR(r6_0, " s_add_u32 s21, s21, 1\n", "", " v_add_u32 v100, s21, v100\n")
// R7: gpr idx: ff1-produced index
PAIR(r7_0, " s_add_u32 s21, s21, 1\n s_and_b32 s20, s21, 3\n", "", " s_set_gpr_idx_on s20, gpr_idx(SRC0)\n v_mov_b32 v101, v100\n s_set_gpr_idx_off\n")
// R8: writelane after m0 write
PAIR(r8_0, " s_and_b32 m0, s21, 63\n", " s_nop 0\n", " v_writelane_b32 v100, s21, m0\n")
// R9: v_cmp -> v_cndmask (vcc use by VALU)
PAIR(r9_0, " v_cmp_le_u32 vcc, %1, %0\n", "", " v_cndmask_b32 v100, v100, v101, vcc\n")
// R10: two valu sgpr writes then salu 4 later
PAIR(r10, " v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 4\n v_readlane_b32 s22, %0, 5\n", F4, " s_add_u32 s24, s24, 1\n")
// baseline: 8 fillers + nothing
TIMED(base8, asm volatile(R F8 ".endr" : "+v"(v) : "v"(lane) CL);)
typedef void (*kern_t)(uint64_t *, uint32_t *);
struct Item { const char *name; kern_t k; int instr; };
#define IT(n, i) {#n, n, i}
int main() {
    uint64_t *d_out; uint32_t *d_buf;
    hipMalloc(&d_out, 64); hipMalloc(&d_buf, 4096); hipMemset(d_buf, 0, 4096);
    Item items[] = { IT(base8, 8), IT(r1_rl_0, 10), IT(r1_rl_2, 12), IT(r1_rl_4, 14), IT(r1_rl_6, 16), IT(r1_cmp_0, 10), IT(r1_cmp_4, 14),
        IT(r1_cmpdep_0, 10), IT(r1_cmpdep_3, 13), IT(r1_cmpdep_4, 14), IT(r1_cmpdep_5, 15), IT(r1_cmpdep_6, 16), IT(r1_mad_0, 10),
        IT(r2_add_1, 11), IT(r2_add_2, 12), IT(r2_add_4, 14), IT(r2_add64_1, 11), IT(r2_add64_4, 14), IT(r2_min_1, 11),
        IT(r3_0, 11), IT(r3_1, 11), IT(r3_3, 13), IT(r4_0, 11), IT(r4_2, 13), IT(r5_2, 12), IT(r5_4, 14), IT(r6_0, 10), IT(r7_0, 13), IT(r8_0, 11), IT(r9_0, 10), IT(r10, 16) };
    for (auto &it : items) {
        uint64_t h[2];
        for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(it.k, dim3(1), dim3(64), 0, 0, d_out, d_buf); hipDeviceSynchronize(); }
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-14s %7.2f cycles = %2d instr x 4 + %6.2f\n", it.name, (double)h[0] / 256, it.instr, (double)h[0] / 256 - 4.0 * it.instr);
    }
    return 0;
}
