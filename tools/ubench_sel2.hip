// ubench_sel2.hip -- wait states / stall source of the v_cmpx -> v_readfirstlane -> exec-restore select idiom (dev tool)
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
// chain: k -> cmpx -> [pad] -> rfl -> s_mov exec -> v_add k
#define LEVEL(name, PAD1, PAD2) TIMED(name, uint32_t k = v; uint32_t c = 0; \
      asm volatile(R " v_cmpx_le_u32 %2, %0\n" PAD1 " v_readfirstlane_b32 %1, %2\n" PAD2 " s_mov_b64 exec, -1\n v_add_u32 %0, %0, %1\n .endr" \
                   : "+v"(k), "+s"(c) : "v"(lane) : "vcc"); v += k; s += c;)
LEVEL(k_l_none, "", "")
LEVEL(k_l_nop0, " s_nop 0\n", "")
LEVEL(k_l_nop1, " s_nop 1\n", "")
LEVEL(k_l_nop3, " s_nop 3\n", "")
LEVEL(k_l_salu1, " s_add_u32 s20, s20, 1\n", "")
LEVEL(k_l_salu2, " s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n", "")
LEVEL(k_l_salu4b, "", " s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n")
LEVEL(k_l_salu2_2, " s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n", " s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n")
LEVEL(k_l_salu8b, "", " s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n")
// cmpx + exec restore only (no rfl), chained through k by a v_add
TIMED(k_cmpx_restore, uint32_t k = v;
      asm volatile(R " v_cmpx_le_u32 %1, %0\n s_mov_b64 exec, -1\n v_add_u32 %0, %0, 1\n .endr" : "+v"(k) : "v"(lane) : "vcc"); v += k;)
// s_mov exec then VALU (no cmpx)
TIMED(k_smov_valu, uint32_t k = v;
      asm volatile(R " s_mov_b64 exec, -1\n v_add_u32 %0, %0, 1\n .endr" : "+v"(k) : "v"(lane) : "vcc"); v += k;)
// cmpx with sdst form (VOP3: result to sgpr pair AND exec?) -- v_cmpx_le_u32_e64
TIMED(k_cmpx_only, uint32_t k = v;
      asm volatile(R " v_cmpx_le_u32 %1, %0\n v_add_u32 %0, %0, 1\n .endr\n s_mov_b64 exec, -1" : "+v"(k) : "v"(lane) : "vcc"); v += k;)
// cmpx -> s_and_saveexec style alternative: v_cmp -> s_and_b64 exec, exec, vcc -> rfl -> s_mov exec
TIMED(k_cmp_sand, uint32_t k = v; uint32_t c = 0;
      asm volatile(R " v_cmp_le_u32 vcc, %2, %0\n s_mov_b64 exec, vcc\n v_readfirstlane_b32 %1, %2\n s_mov_b64 exec, -1\n v_add_u32 %0, %0, %1\n .endr"
                   : "+v"(k), "+s"(c) : "v"(lane) : "vcc"); v += k; s += c;)
// two back-to-back levels sharing one restore? (cmpx narrows further): cmpx, rfl, cmpx(other), rfl, restore
TIMED(k_two_levels, uint32_t k = v; uint32_t c = 0; uint32_t e = 0;
      asm volatile(R " v_cmpx_le_u32 %3, %0\n s_nop 0\n v_readfirstlane_b32 %1, %3\n s_mov_b64 exec, -1\n v_add_u32 %0, %0, %1\n v_cmpx_le_u32 %3, %0\n s_nop 0\n v_readfirstlane_b32 %2, %3\n s_mov_b64 exec, -1\n v_add_u32 %0, %0, %2\n .endr"
                   : "+v"(k), "+s"(c), "+s"(e) : "v"(lane) : "vcc"); v += k; s += c + e;)
// VALU between exec restore and next cmpx (is stall at next cmpx?)
TIMED(k_l_valu_after, uint32_t k = v; uint32_t c = 0; uint32_t z = v;
      asm volatile(R " v_cmpx_le_u32 %2, %0\n s_nop 0\n v_readfirstlane_b32 %1, %2\n s_mov_b64 exec, -1\n v_add_u32 %0, %0, %1\n v_add_u32 %3, %3, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %3, %3, 1\n .endr"
                   : "+v"(k), "+s"(c), "+v"(z) : "v"(lane) : "vcc"); v += k + z; s += c;)

// correctness of first-active-lane read with various pads: out[i] = value read (expect 42)
__global__ void __launch_bounds__(64) k_sem(uint32_t *out) {
    uint32_t lane = threadIdx.x, k41 = 41, r;
#define SEM(i, PAD) asm volatile("v_cmpx_gt_u32 %1, %2\n" PAD "v_readfirstlane_b32 %0, %1\n s_mov_b64 exec, -1" : "=s"(r) : "v"(lane), "v"(k41) : "vcc"); if (lane == 5) out[i] = r;
    SEM(0, "")
    SEM(1, "s_nop 0\n")
    SEM(2, "s_nop 1\n")
    SEM(3, "s_nop 3\n")
    SEM(4, "s_add_u32 s20, s20, 1\n")
    SEM(5, "v_nop\n")
    // e64 form writing an SGPR pair
    asm volatile("v_cmpx_gt_u32_e64 s[20:21], %1, %2\n v_readfirstlane_b32 %0, %1\n s_mov_b64 exec, -1" : "=s"(r) : "v"(lane), "v"(k41) : "vcc", "s20", "s21"); if (lane == 5) out[6] = r;
    // does cmpx also write vcc (e32)?
    uint32_t lo, hi;
    asm volatile("s_mov_b64 vcc, 0\n v_cmpx_gt_u32 %2, %3\n s_nop 3\n s_mov_b32 %0, vcc_lo\n s_mov_b32 %1, vcc_hi\n s_mov_b64 exec, -1" : "=s"(lo), "=s"(hi) : "v"(lane), "v"(k41) : "vcc"); if (lane == 5) { out[7] = lo; out[8] = hi; }
}
typedef void (*kern_t)(uint64_t *, uint32_t *);
struct Item { const char *name; kern_t k; int n; };
int main() {
    uint64_t *d_out; uint32_t *d_buf;
    hipMalloc(&d_out, 64); hipMalloc(&d_buf, 4096); hipMemset(d_buf, 0, 4096);
    Item items[] = {
        {"cmpx,rfl,s_mov exec,v_add (4 instr)", k_l_none, 256}, {" + s_nop 0 after cmpx", k_l_nop0, 256}, {" + s_nop 1", k_l_nop1, 256},
        {" + s_nop 3", k_l_nop3, 256}, {" + 1 salu after cmpx", k_l_salu1, 256}, {" + 2 salu after cmpx", k_l_salu2, 256},
        {" + 4 salu before restore", k_l_salu4b, 256}, {" + 2 salu after cmpx, 2 before restore", k_l_salu2_2, 256},
        {" + 8 salu before restore", k_l_salu8b, 256},
        {"cmpx,s_mov exec,v_add", k_cmpx_restore, 256}, {"s_mov exec,v_add", k_smov_valu, 256}, {"cmpx,v_add (no restore)", k_cmpx_only, 256},
        {"v_cmp,s_mov exec vcc,rfl,s_mov exec,v_add", k_cmp_sand, 256}, {"two levels (10 instr)", k_two_levels, 256},
        {"level + nop0 + 4 indep valu after", k_l_valu_after, 256},
    };
    for (auto &it : items) {
        uint64_t h[2];
        for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(it.k, dim3(1), dim3(64), 0, 0, d_out, d_buf); hipDeviceSynchronize(); }
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-48s %8.2f cycles per iteration\n", it.name, (double)h[0] / it.n);
    }
    uint32_t *d_sem; hipMalloc(&d_sem, 64); hipMemset(d_sem, 0xff, 64);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d_sem); hipDeviceSynchronize();
    uint32_t hs[9]; hipMemcpy(hs, d_sem, 36, hipMemcpyDeviceToHost);
    printf("sem (expect 42): none=%u nop0=%u nop1=%u nop3=%u salu=%u vnop=%u e64=%u ; vcc after cmpx_e32 = %08x:%08x\n", hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[8], hs[7]);
    return 0;
}
