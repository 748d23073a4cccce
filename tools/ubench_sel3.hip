// ubench_sel3.hip -- cost of the instruction pairs used by the round-2 chain loop on a lone wavefront (dev tool)
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
#define CL : "vcc", "scc", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"
// A: three readlanes then a VALU consumer of the last
TIMED(kA, asm volatile(R "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %0, 4\n v_readlane_b32 s22, %0, 5\n v_nop\n v_add_u32 %0, s22, %0\n .endr" : "+v"(v) : CL);)
// B: v_cmp, 4 plain valu, s_ff1, readlane(lane from ff1), 1 valu, valu use
TIMED(kB, asm volatile(R "v_cmp_le_u32 vcc, %1, %0\n v_add_u32 v100, v100, 1\n v_add_u32 v101, v101, 1\n v_add_u32 v102, v102, 1\n v_add_u32 v103, v103, 1\n s_ff1_i32_b64 s20, vcc\n v_readlane_b32 s21, %0, s20\n v_add_u32 v100, v100, 1\n v_add_u32 %0, s21, %0\n .endr" : "+v"(v) : "v"(0u) CL);)
// B0: same with the s_ff1 right after v_cmp (fillers after)
TIMED(kB0, asm volatile(R "v_cmp_le_u32 vcc, %1, %0\n s_ff1_i32_b64 s20, vcc\n v_add_u32 v100, v100, 1\n v_add_u32 v101, v101, 1\n v_add_u32 v102, v102, 1\n v_add_u32 v103, v103, 1\n v_readlane_b32 s21, %0, s20\n v_add_u32 v100, v100, 1\n v_add_u32 %0, s21, %0\n .endr" : "+v"(v) : "v"(0u) CL);)
// B5/B6: 5 / 6 fillers
TIMED(kB5, asm volatile(R "v_cmp_le_u32 vcc, %1, %0\n v_add_u32 v100, v100, 1\n v_add_u32 v101, v101, 1\n v_add_u32 v102, v102, 1\n v_add_u32 v103, v103, 1\n v_add_u32 v104, v104, 1\n s_ff1_i32_b64 s20, vcc\n v_readlane_b32 s21, %0, s20\n v_add_u32 v100, v100, 1\n v_add_u32 %0, s21, %0\n .endr" : "+v"(v) : "v"(0u) CL);)
// C: salu -> valu forwarding chain: s_add ; v_add using it ; readfirstlane back
TIMED(kC, asm volatile(R "s_add_u32 s20, s20, 1\n v_add_u32 %0, s20, %0\n .endr" : "+v"(v) : CL);)
// E: s_ff1 (of a constant sgpr pair) -> readlane -> valu
TIMED(kE, asm volatile(R "s_ff1_i32_b64 s20, s[22:23]\n v_readlane_b32 s21, %0, s20\n v_nop\n v_add_u32 %0, s21, %0\n .endr" : "+v"(v) : CL);)
// F: s_ff1 -> gpr idx read
TIMED(kF, asm volatile(R "s_ff1_i32_b64 s20, s[22:23]\n s_and_b32 s20, s20, 3\n s_set_gpr_idx_on s20, gpr_idx(SRC0)\n v_mov_b32 %0, v100\n s_set_gpr_idx_off\n .endr" : "+v"(v) : CL);)
// G: the 64-bit division block + s_waitcnt
TIMED(kG, asm volatile("v_mov_b32 v101, 0\n v_mov_b32 v105, 0\n" R "v_mul_hi_u32 v100, s20, %0\n v_mad_u64_u32 v[102:103], s[24:25], s20, %0, v[100:101]\n v_mov_b32 v104, v102\n v_mad_u64_u32 v[106:107], s[24:25], s21, %0, v[104:105]\n v_add_co_u32_e64 v108, s[24:25], v103, v107\n v_addc_co_u32_e64 v109, s[26:27], 0, 0, s[24:25]\n v_mad_u64_u32 v[102:103], s[24:25], s21, %0, v[108:109]\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(v) : CL);)
// G2: same without the waitcnt
TIMED(kG2, asm volatile("v_mov_b32 v101, 0\n v_mov_b32 v105, 0\n" R "v_mul_hi_u32 v100, s20, %0\n v_mad_u64_u32 v[102:103], s[24:25], s20, %0, v[100:101]\n v_mov_b32 v104, v102\n v_mad_u64_u32 v[106:107], s[24:25], s21, %0, v[104:105]\n v_add_co_u32_e64 v108, s[24:25], v103, v107\n v_addc_co_u32_e64 v109, s[26:27], 0, 0, s[24:25]\n v_mad_u64_u32 v[102:103], s[24:25], s21, %0, v[108:109]\n .endr" : "+v"(v) : CL);)
// H: 64-bit shift / add
TIMED(kH1, asm volatile(R "v_lshlrev_b64 v[100:101], s20, 1\n .endr" : "+v"(v) : CL);)
TIMED(kH2, asm volatile(R "v_lshl_add_u64 v[100:101], v[102:103], 0, v[104:105]\n .endr" : "+v"(v) : CL);)
TIMED(kH3, asm volatile(R "v_bfi_b32 v100, v101, 0, s20\n .endr" : "+v"(v) : CL);)
TIMED(kH4, asm volatile(R "v_mul_lo_u32 v100, v101, %0\n .endr" : "+v"(v) : CL);)
TIMED(kH5, asm volatile(R "v_mul_hi_u32 v100, v101, %0\n .endr" : "+v"(v) : CL);)
TIMED(kH6, asm volatile(R "v_mad_u32_u24 v100, v101, s20, %0\n .endr" : "+v"(v) : CL);)
TIMED(kH7, asm volatile(R "v_bcnt_u32_b32 v100, v101, 0\n v_mbcnt_lo_u32_b32 v102, s20, 0\n .endr" : "+v"(v) : CL);)
// K: fix-up chain (8 dependent valu) + readlane + mov
TIMED(kK, asm volatile(R "v_lshrrev_b32_e64 v100, 16, s20\n v_bfe_u32 v101, s20, 0, 16\n v_add_u32 v102, %0, v100\n v_mad_u32_u24 v102, v101, s21, v102\n v_mul_hi_u32 v103, v102, %0\n v_mad_i32_i24 v104, v103, %0, v102\n v_sub_u32 v105, v104, %0\n v_min_u32 v106, v104, v105\n v_ashrrev_i32 v107, 31, v105\n v_readlane_b32 s22, v106, 3\n v_add3_u32 v108, v103, v107, 1\n v_mov_b32 %0, s22\n .endr" : "+v"(v) : CL);)
// L: salu batch A
TIMED(kL, asm volatile(R "s_and_b32 m0, s20, 63\n s_cmp_ge_u32 s21, s22\n v_writelane_b32 %0, s20, m0\n s_cselect_b32 s24, s21, s20\n s_cselect_b32 s25, 0, s21\n s_addc_u32 s20, s20, 0\n s_lshl_b64 s[26:27], s[24:25], 3\n .endr" : "+v"(v) : CL);)
// M: readlane x3 right after a s_ff1 + 5 salu
TIMED(kM, asm volatile(R "s_ff1_i32_b64 s20, s[22:23]\n s_lshl_b32 s24, s20, 6\n s_or_b32 s25, s25, s24\n s_lshl_b32 s26, s25, 2\n s_add_u32 s26, s26, s20\n s_lshl_b32 s26, s26, 3\n v_readlane_b32 s21, %0, s20\n v_readlane_b32 s24, %0, s20\n v_readlane_b32 s27, %0, s20\n v_subrev_u32 %0, s21, %0\n v_mbcnt_lo_u32_b32 v100, s24, 0\n v_mbcnt_hi_u32_b32 v100, s27, v100\n .endr" : "+v"(v) : CL);)
// N: loop-bottom salu + taken branch (in a real loop of 256 iterations)
TIMED(kN, uint32_t c = 0; asm volatile("1:\n s_add_u32 s24, s21, -1\n s_cmp_ge_u32 s24, s22\n s_cselect_b32 s25, 0, 256\n s_sub_u32 s24, s20, s21\n s_cmp_ge_u32 s24, 0x7fffffff\n s_cselect_b32 s25, 0, s25\n s_add_u32 %1, %1, 1\n s_cmp_lt_u32 %1, s25\n s_cbranch_scc1 1b\n" : "+v"(v), "+s"(c) : CL);)
// O: dpp pair with the real spacing
TIMED(kO, asm volatile(R "v_bcnt_u32_b32 v100, %0, 0\n v_bcnt_u32_b32 v100, %0, v100\n v_ashrrev_i32 v104, 31, v104\n v_mul_lo_u32 v105, v106, %0\n v_add_u32_dpp v101, v100, v100 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_add_u32 v107, v107, v104\n v_sub_u32 v105, s20, v105\n v_add_u32_dpp v102, v101, v101 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_cmp_gt_u32 vcc, v102, %0\n v_sub_u32 v103, v102, v100\n .endr" : "+v"(v) : CL);)
typedef void (*kern_t)(uint64_t *, uint32_t *);
struct Item { const char *name; kern_t k; int n; int instr; };
int main() {
    uint64_t *d_out; uint32_t *d_buf;
    hipMalloc(&d_out, 64); hipMalloc(&d_buf, 4096); hipMemset(d_buf, 0, 4096);
    Item items[] = {
        {"A readlane x3, nop, valu", kA, 256, 5}, {"B cmp,4 valu,ff1,readlane,valu,valu", kB, 256, 9}, {"B0 cmp,ff1,4 valu,readlane,valu,valu", kB0, 256, 9},
        {"B5 cmp,5 valu,ff1,readlane,valu,valu", kB5, 256, 10}, {"C s_add -> v_add", kC, 256, 2}, {"E ff1,readlane,nop,valu", kE, 256, 4},
        {"F ff1,and,idx_on,mov,idx_off", kF, 256, 5}, {"G division block + waitcnt", kG, 256, 8}, {"G2 division block", kG2, 256, 7},
        {"H1 v_lshlrev_b64", kH1, 256, 1}, {"H2 v_lshl_add_u64", kH2, 256, 1}, {"H3 v_bfi_b32", kH3, 256, 1}, {"H4 v_mul_lo_u32", kH4, 256, 1},
        {"H5 v_mul_hi_u32", kH5, 256, 1}, {"H6 v_mad_u32_u24", kH6, 256, 1}, {"H7 bcnt + mbcnt", kH7, 256, 2}, {"K fix-up chain", kK, 256, 12},
        {"L salu batch A", kL, 256, 7}, {"M ff1,5 salu,3 readlane,sub,mbcnt x2", kM, 256, 12}, {"N loop bottom + taken branch", kN, 256, 9},
        {"O bcnt/dpp block", kO, 256, 10},
    };
    for (auto &it : items) {
        uint64_t h[2];
        for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(it.k, dim3(1), dim3(64), 0, 0, d_out, d_buf); hipDeviceSynchronize(); }
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-44s %8.2f cycles per iteration = %d instr x 4 + %6.2f\n", it.name, (double)h[0] / it.n, it.instr, (double)h[0] / it.n - 4.0 * it.instr);
    }
    return 0;
}
