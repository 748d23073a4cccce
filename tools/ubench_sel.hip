// ubench_sel.hip -- lone-wavefront latency of the candidate select / fix-up sequences of the round-2 ROC chain kernels
// (dev tool).  Each TIMED body is one dependent round of the sequence, repeated 1024 times.
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_sel.hip -o /tmp/ubench_sel && /tmp/ubench_sel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define TIMED(name, ...)                                                                  \
    __global__ void __launch_bounds__(64) name(uint64_t *out, uint32_t *buf) {             \
        __shared__ uint32_t lds[2048];                                                    \
        for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = buf[i & 1023];              \
        __syncthreads();                                                                  \
        uint32_t s = buf[0], v = buf[threadIdx.x], lane = threadIdx.x, a = 0;             \
        (void)a; (void)lane;                                                              \
        uint64_t t0, t1;                                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)); \
        __VA_ARGS__                                                                        \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)); \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = s + v; }                       \
    }

// old level: v_cmp -> s_ff1 -> v_readlane -> v_sub  (k in v, counters in `lane`)
TIMED(k_old_level, uint32_t k = v; uint32_t c = 0;
      asm volatile(".rept 1024\n v_cmp_gt_u32 vcc, %2, %0\n s_ff1_i32_b64 %1, vcc\n s_nop 3\n v_readlane_b32 %1, %2, %1\n s_nop 0\n v_sub_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n .endr"
                   : "+v"(k), "+s"(c) : "v"(lane) : "vcc"); v += k; s += c;)
// new level: v_cmpx -> readfirstlane x2 -> exec restore -> v_sub
TIMED(k_new_level, uint32_t k = v; uint32_t c = 0; uint32_t e = 0;
      asm volatile(".rept 1024\n v_cmpx_le_u32 %3, %0\n v_readfirstlane_b32 %1, %3\n v_readfirstlane_b32 %2, %3\n s_mov_b64 exec, -1\n v_sub_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n .endr"
                   : "+v"(k), "+s"(c), "+s"(e) : "v"(lane) : "vcc"); v += k; s += c + e;)
// same without the v_add (pure chain: cmpx, rfl, s_mov, v_sub)
TIMED(k_new_level_min, uint32_t k = v; uint32_t c = 0;
      asm volatile(".rept 1024\n v_cmpx_le_u32 %2, %0\n v_readfirstlane_b32 %1, %2\n s_mov_b64 exec, -1\n v_add_u32 %0, %0, %1\n .endr"
                   : "+v"(k), "+s"(c) : "v"(lane) : "vcc"); v += k; s += c;)
// readfirstlane into m0 -> movrels -> cmpx chain
TIMED(k_movrels, uint32_t k = v; uint32_t c = 0;
      asm volatile(".rept 1024\n v_cmpx_le_u32 %2, %0\n v_readfirstlane_b32 %1, %2\n s_mov_b64 exec, -1\n s_set_gpr_idx_on %1, gpr_idx(SRC0)\n v_mov_b32 %0, %0\n s_set_gpr_idx_off\n .endr"
                   : "+v"(k), "+s"(c) : "v"(lane) : "vcc"); v += k; s += c;)
TIMED(k_old_gpridx, uint32_t k = v; uint32_t c = 0;
      asm volatile(".rept 1024\n v_cmp_le_u32 vcc, %2, %0\n s_ff1_i32_b64 %1, vcc\n s_set_gpr_idx_on %1, gpr_idx(SRC0)\n v_mov_b32 %0, %0\n s_set_gpr_idx_off\n .endr"
                   : "+v"(k), "+s"(c) : "v"(lane) : "vcc"); v += k; s += c;)
// mul_hi vs mad_u64 dependent
TIMED(k_mulhi, uint32_t k = v | 1;
      asm volatile(".rept 1024\n v_mul_hi_u32 %0, %0, %1\n v_or_b32 %0, 0x10000, %0\n .endr" : "+v"(k) : "s"(0xfffffff0u)); v += k;)
TIMED(k_mad64hi, uint32_t q = v | 1;
      asm volatile("v_mov_b32 v100, %0\n .rept 1024\n v_mad_u64_u32 v[100:101], vcc, v100, %1, 0\n v_or_b32 v100, 0x10000, v101\n .endr\n v_mov_b32 %0, v100" : "+v"(q) : "s"(0xfffffff0u) : "vcc", "v100", "v101"); v += q;)
TIMED(k_mad_i24, uint32_t k = v;
      asm volatile(".rept 1024\n v_mad_i32_i24 %0, %0, %1, %0\n .endr" : "+v"(k) : "s"(3u)); v += k;)
TIMED(k_mul_lo, uint32_t k = v | 1;
      asm volatile(".rept 1024\n v_mul_lo_u32 %0, %0, %1\n .endr" : "+v"(k) : "s"(3u)); v += k;)
// LDS: address -> ds_read_b64 -> wait -> bcnt x2 -> use as address
TIMED(k_lds_b64, uint32_t p = (threadIdx.x & 3) * 8; uint64_t w = 0;
      asm volatile(".rept 1024\n ds_read_b64 v[100:101], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xff8, v100\n .endr" : "+v"(p) : : "v100", "v101"); v += p;)
// LDS read with 8 independent VALU instructions in the shadow
TIMED(k_lds_shadow8, uint32_t p = (threadIdx.x & 3) * 8; uint64_t w = 0; uint32_t z = v;
      asm volatile(".rept 1024\n ds_read_b64 v[100:101], %0\n .rept 8\n v_add_u32 %1, %1, 1\n .endr\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xff8, v100\n .endr" : "+v"(p), "+v"(z) : : "v100", "v101"); v += p + z;)
TIMED(k_lds_shadow16, uint32_t p = (threadIdx.x & 3) * 8; uint64_t w = 0; uint32_t z = v;
      asm volatile(".rept 1024\n ds_read_b64 v[100:101], %0\n .rept 16\n v_add_u32 %1, %1, 1\n .endr\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xff8, v100\n .endr" : "+v"(p), "+v"(z) : : "v100", "v101"); v += p + z;)
// same-address 64-lane ds_write_b64 followed (later) by a dependent read
TIMED(k_lds_wr_rd, uint32_t p = 64; uint64_t w = v;
      asm volatile(".rept 1024\n ds_write_b64 %0, v[100:101]\n ds_read_b64 v[100:101], %0\n s_waitcnt lgkmcnt(0)\n .endr" : "+v"(p) : : "v100", "v101"); v += p;)
// independent VALU issue: 4 chains
TIMED(k_valu4, uint32_t a1 = v, a2 = v + 1, a3 = v + 2, a4 = v + 3;
      asm volatile(".rept 512\n v_add_u32 %0, %0, 3\n v_add_u32 %1, %1, 5\n v_add_u32 %2, %2, 5\n v_add_u32 %3, %3, 5\n .endr" : "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4)); v += a1 + a2 + a3 + a4;)
// SALU + VALU interleaved, both independent chains (x2048 instr)
TIMED(k_mix, uint32_t a1 = v; uint32_t s1 = s;
      asm volatile(".rept 1024\n v_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5\n .endr" : "+v"(a1), "+s"(s1)); v += a1; s += s1;)
// VALU writes sgpr (readfirstlane) -> SALU consumer -> VALU consumer
TIMED(k_rfl_salu, uint32_t a1 = v; uint32_t s1 = s;
      asm volatile(".rept 1024\n v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 5\n v_add_u32 %0, %1, %0\n .endr" : "+v"(a1), "+s"(s1)); v += a1; s += s1;)
TIMED(k_rfl_valu, uint32_t a1 = v; uint32_t s1 = s;
      asm volatile(".rept 1024\n v_readfirstlane_b32 %1, %0\n v_add_u32 %0, %1, %0\n v_add_u32 %0, 5, %0\n .endr" : "+v"(a1), "+s"(s1)); v += a1; s += s1;)
// mbcnt pair -> cmpx -> rfl -> exec restore
TIMED(k_bitsel, uint32_t k = v & 7; uint32_t x = 0; uint32_t c;
      asm volatile(".rept 1024\n v_mbcnt_lo_u32_b32 %3, %2, 0\n v_mbcnt_hi_u32_b32 %3, %2, %3\n v_cmpx_gt_u32 %3, %0\n v_readfirstlane_b32 %1, %4\n s_mov_b64 exec, -1\n v_and_b32 %0, 7, %1\n .endr"
                   : "+v"(k), "+s"(x), "+s"(s), "=&v"(c) : "v"(lane) : "vcc"); v += k; s += x;)
// writelane with m0 as lane select
TIMED(k_writelane, uint32_t k = v;
      asm volatile(".rept 1024\n s_mov_b32 m0, %1\n v_writelane_b32 %0, %1, m0\n .endr" : "+v"(k) : "s"(s & 63)); v += k;)
// branch-free conditional push: cmp_eq + cndmask
TIMED(k_cndpush, uint32_t k = v; uint32_t sl = s & 63;
      asm volatile(".rept 1024\n v_cmp_ne_u32 vcc, %1, %2\n v_cndmask_b32 %0, %3, %0, vcc\n .endr" : "+v"(k) : "s"(sl), "v"(lane), "v"(v) : "vcc"); v += k;)
// dpp suffix add pair
TIMED(k_dpp2, uint32_t k = v; uint32_t t;
      asm volatile(".rept 1024\n s_nop 1\n v_add_u32_dpp %1, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n s_nop 1\n v_add_u32_dpp %0, %1, %1 row_shl:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n .endr" : "+v"(k), "=&v"(t)); v += k;)

// semantic checks
__global__ void __launch_bounds__(64) k_sem(uint32_t *out) {
    uint32_t lane = threadIdx.x;
    uint32_t r0, r1, r2;
    uint32_t big = 1000;
    // exec == 0 after cmpx: which lane does readfirstlane read?
    asm volatile("v_cmpx_gt_u32 %1, %2\n v_readfirstlane_b32 %0, %1\n s_mov_b64 exec, -1" : "=s"(r0) : "v"(lane + 100), "v"(big) : "vcc");
    // first active lane of "lane > 41"
    uint32_t k41 = 41;
    asm volatile("v_cmpx_gt_u32 %1, %2\n v_readfirstlane_b32 %0, %1\n s_mov_b64 exec, -1" : "=s"(r1) : "v"(lane), "v"(k41) : "vcc");
    // movrels with m0 from readfirstlane
    uint32_t a0 = lane + 1000, a1 = lane + 2000, a2 = lane + 3000, a3 = lane + 4000;
    uint32_t res;
    asm volatile("v_mov_b32 v100, %1\n v_mov_b32 v101, %2\n v_mov_b32 v102, %3\n v_mov_b32 v103, %4\n v_readfirstlane_b32 s20, %5\n s_set_gpr_idx_on s20, gpr_idx(SRC0)\n v_mov_b32 %0, v100\n s_set_gpr_idx_off"
                 : "=v"(res) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(2u) : "v100", "v101", "v102", "v103", "s20");
    r2 = res;
    if (lane == 5) { out[0] = r0; out[1] = r1; out[2] = r2; }
}

typedef void (*kern_t)(uint64_t *, uint32_t *);
struct Item { const char *name; kern_t k; int n; };

int main() {
    uint64_t *d_out; uint32_t *d_buf;
    hipMalloc(&d_out, 64); hipMalloc(&d_buf, 4096);
    hipMemset(d_buf, 0, 4096);
    Item items[] = {
        {"old level: cmp,ff1,readlane,sub,add", k_old_level, 1024},
        {"new level: cmpx,rfl,rfl,s_mov exec,sub,add", k_new_level, 1024},
        {"new level min: cmpx,rfl,s_mov exec,add", k_new_level_min, 1024},
        {"cmpx,rfl,s_mov exec,gpr_idx_on,v_mov,off", k_movrels, 1024},
        {"cmp,ff1,gpr_idx_on,v_mov,off", k_old_gpridx, 1024},
        {"v_mul_hi_u32 + v_or", k_mulhi, 1024},
        {"v_mad_u64_u32 + v_or", k_mad64hi, 1024},
        {"v_mad_i32_i24 dep", k_mad_i24, 1024},
        {"v_mul_lo_u32 dep", k_mul_lo, 1024},
        {"ds_read_b64 -> v_and chain", k_lds_b64, 1024},
        {"ds_read_b64 + 8 valu shadow", k_lds_shadow8, 1024},
        {"ds_read_b64 + 16 valu shadow", k_lds_shadow16, 1024},
        {"ds_write_b64 same addr + ds_read", k_lds_wr_rd, 1024},
        {"4 independent v_add (per 4)", k_valu4, 512},
        {"v_add + s_add independent (per 2)", k_mix, 1024},
        {"rfl -> s_add -> v_add", k_rfl_salu, 1024},
        {"rfl -> v_add -> v_add", k_rfl_valu, 1024},
        {"mbcnt x2,cmpx,rfl,s_mov,v_and", k_bitsel, 1024},
        {"s_mov m0 + v_writelane", k_writelane, 1024},
        {"v_cmp_ne + v_cndmask", k_cndpush, 1024},
        {"nop,dpp add,nop,dpp add", k_dpp2, 1024},
    };
    // clock calibration: memtime tick in ns is printed by ubench_issue; here only ticks
    for (auto &it : items) {
        uint64_t h[2];
        for (int rep = 0; rep < 3; rep++) {
            hipLaunchKernelGGL(it.k, dim3(1), dim3(64), 0, 0, d_out, d_buf);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost);
        printf("%-48s %8.2f memtime-ticks per iteration\n", it.name, (double)h[0] / it.n);
    }
    uint32_t *d_sem; hipMalloc(&d_sem, 64);
    hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, d_sem);
    hipDeviceSynchronize();
    uint32_t hs[3]; hipMemcpy(hs, d_sem, 12, hipMemcpyDeviceToHost);
    printf("sem: exec=0 readfirstlane -> %u (lane 0 holds 100); first lane > 41 -> %u (expect 42); movrels m0=2 -> %u (expect 3005)\n", hs[0], hs[1], hs[2]);
    return 0;
}
