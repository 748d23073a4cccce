// ubench_issue.hip -- single-wavefront issue/latency microbenchmarks on gfx950 (dev tool, not part of the library).
// Measures cycles per instruction for dependent chains a lone wave executes, the regime of the ROC serial chain.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_issue.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 2048
#define TIMED(name, body)                                                                 \
    __global__ void __launch_bounds__(64) name(uint64_t *out, uint32_t *buf) {             \
        __shared__ uint32_t lds[1024];                                                    \
        for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 4 + 4) & 4095;         \
        __syncthreads();                                                                  \
        uint32_t s = buf[0], v = buf[threadIdx.x], a = 0;                                 \
        (void)a;                                                                          \
        uint64_t t0, t1;                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)); \
        body                                                                              \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)); \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s + v; }                       \
    }

TIMED(k_salu_dep, asm volatile(".rept 2048\n s_add_u32 %0, %0, 3\n .endr" : "+s"(s));)
TIMED(k_salu_mul_dep, asm volatile(".rept 2048\n s_mul_i32 %0, %0, 3\n .endr" : "+s"(s));)
TIMED(k_salu_mulhi_dep, asm volatile(".rept 2048\n s_mul_hi_u32 %0, %0, 0x7fffffff\n .endr" : "+s"(s));)
TIMED(k_valu_dep, asm volatile(".rept 2048\n v_add_u32 %0, %0, 3\n .endr" : "+v"(v));)
TIMED(k_valu_indep, uint32_t v2 = v + 1; asm volatile(".rept 1024\n v_add_u32 %0, %0, 3\n v_add_u32 %1, %1, 5\n .endr" : "+v"(v), "+v"(v2)); v += v2;)
TIMED(k_salu_indep, uint32_t s2 = s + 1; asm volatile(".rept 1024\n s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n .endr" : "+s"(s), "+s"(s2)); s += s2;)
TIMED(k_mix_indep, asm volatile(".rept 1024\n s_add_u32 %0, %0, 3\n v_add_u32 %1, %1, 5\n .endr" : "+s"(s), "+v"(v));)
// VALU -> SGPR -> VALU round trip: v_readlane (needs wait states) then v_add with that sgpr
TIMED(k_readlane_chain, asm volatile(".rept 2048\n v_readlane_b32 %0, %1, 3\n s_nop 0\n v_add_u32 %1, %1, %0\n .endr" : "+s"(s), "+v"(v));)
// v_cmp -> vcc -> s_ff1 -> v_add
TIMED(k_cmp_ff1_chain, uint64_t m; asm volatile(".rept 2048\n v_cmp_lt_u32 vcc, %1, %2\n s_ff1_i32_b64 %1, vcc\n s_add_u32 %1, %1, 7\n .endr" : "=s"(m), "+s"(s), "+v"(v) : : "vcc");)
// dependent LDS loads (pointer chase)
TIMED(k_lds_chain, uint32_t p = (threadIdx.x * 4) & 4095; asm volatile(".rept 2048\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(p)); v += p;)
// LDS load + readfirstlane + use as address (scalar round trip)
TIMED(k_lds_rfl_chain, uint32_t p = 0; asm volatile(".rept 2048\n ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_readfirstlane_b32 %1, %0\n s_nop 0\n v_mov_b32 %0, %1\n .endr" : "+v"(p), "+s"(s)); v += p;)
// taken branch cost
TIMED(k_branch_taken, asm volatile(".rept 2048\n s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n .endr" : "+s"(s));)
TIMED(k_branch_not_taken, asm volatile(".rept 2048\n s_cmp_eq_u32 0, 1\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n .endr" : "+s"(s));)
TIMED(k_set_gpr_idx, uint32_t w = v; asm volatile(".rept 2048\n s_set_gpr_idx_on %1, gpr_idx(SRC0)\n v_mov_b32 %0, %0\n s_set_gpr_idx_off\n .endr" : "+v"(w) : "s"(0)); v += w;)
TIMED(k_mad_u64_dep, uint64_t q = v; asm volatile(".rept 2048\n v_mad_u64_u32 %0, vcc, %1, %1, %0\n .endr" : "+v"(q) : "v"(v) : "vcc"); v += (uint32_t)q;)
TIMED(k_readfirstlane_dep, asm volatile(".rept 2048\n v_readfirstlane_b32 %0, %1\n s_nop 0\n v_mov_b32 %1, %0\n .endr" : "+s"(s), "+v"(v));)

// clock calibration: 64 * 65536 dependent s_add per launch, timed with hipEvents and s_memtime
__global__ void __launch_bounds__(64) k_calib(uint64_t *out, uint32_t *buf) {
    uint32_t s = buf[0];
    uint64_t t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < 65536; i++) asm volatile(".rept 64\n s_add_u32 %0, %0, 3\n .endr" : "+s"(s));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s; }
}

typedef void (*kern_t)(uint64_t *, uint32_t *);
struct Item { const char *name; kern_t k; int n; };

int main() {
    uint64_t *d_out; uint32_t *d_buf;
    hipMalloc(&d_out, 64); hipMalloc(&d_buf, 4096);
    hipMemset(d_buf, 0, 4096);
    Item items[] = {
        {"s_add dependent", k_salu_dep, 2048}, {"s_mul_i32 dependent", k_salu_mul_dep, 2048},
        {"s_mul_hi_u32 dependent", k_salu_mulhi_dep, 2048}, {"v_add dependent", k_valu_dep, 2048},
        {"v_add 2 independent chains", k_valu_indep, 2048}, {"s_add 2 independent chains", k_salu_indep, 2048},
        {"s_add + v_add independent", k_mix_indep, 2048}, {"readlane->nop->v_add chain (x3 instr)", k_readlane_chain, 2048},
        {"v_cmp->s_ff1->s_add chain (x3 instr)", k_cmp_ff1_chain, 2048}, {"ds_read dependent", k_lds_chain, 2048},
        {"ds_read->readfirstlane->v_mov chain", k_lds_rfl_chain, 2048}, {"branch taken (cmp+branch)", k_branch_taken, 2048},
        {"branch not taken (cmp+branch+nop)", k_branch_not_taken, 2048}, {"s_set_gpr_idx_on+v_mov+off", k_set_gpr_idx, 2048},
        {"v_mad_u64_u32 dependent", k_mad_u64_dep, 2048}, {"readfirstlane->nop->v_mov chain", k_readfirstlane_dep, 2048},
    };
    for (auto &it : items) {
        uint64_t h[2];
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(it.k, dim3(1), dim3(64), 0, 0, d_out, d_buf);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-44s %8.2f memtime-ticks per iteration\n", it.name, (double)h[0] / it.n);
    }
    // wall-clock calibration of the memtime tick: long dependent s_add loop
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_calib, dim3(1), dim3(64), 0, 0, d_out, d_buf);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        uint64_t h[2];
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        double n = 64.0 * 65536;
        printf("calib: %.0f dependent s_add: %.3f ms wall = %.2f ns/instr; %llu memtime ticks = %.2f ticks/instr; tick = %.3f ns\n",
               n, ms, ms * 1e6 / n, (unsigned long long)h[0], h[0] / n, ms * 1e6 / (double)h[0]);
    }
    return 0;
}
