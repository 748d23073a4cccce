// hw_gpr_idx_probe.hip -- does a VOP3 write under s_set_gpr_idx_on ..., gpr_idx(SRC0,DST) stay inside its wavefront's registers?
// (dev tool, not part of the library.)  Round 3 found S2-sized ROC decodes coming back with a wrong list of ANOTHER kernel class while a
// 256-VGPR kernel that wrote its register slots with "v_cndmask_b32_e64 v64, v64, x, mask" under gpr_idx(SRC0,DST) was running -- also
// when that kernel stored nothing to memory.  This program takes the construct out of the library:
//   attackers  256-VGPR wavefronts that do nothing but write slots v64..v255 by register index, in one of four forms
//              0  s_set_gpr_idx_on i, gpr_idx(SRC0,DST); v_cndmask_b32_e64 v64, v64, x, mask      (the round-3 form)
//              1  exec-masked  s_set_gpr_idx_on i, gpr_idx(DST); v_mov_b32 v64, x                 (the shipped form)
//              2  form 0 with s_nop 4 between the mode switch and the VOP3 instruction
//              3  SRC0 and DST in two steps: indexed read into a temporary, v_cndmask on plain registers, indexed write
//              4  form 0 with the lane mask produced by a v_cmp right in front of the mode switch, as the compiler had placed it in
//                 the library's kernel: VALU writes an SGPR pair, ONE scalar instruction, VALU reads the pair as a mask (gfx940+
//                 want two wait states there; the compiler cannot see into an asm statement)
//              5  form 4 with s_nop 1 behind the v_cmp
//   victims    128-VGPR wavefronts of another kernel, on another stream, that fill v32..v127 with values they can recompute, then either
//              sleep (passive) or add 1 to every register per iteration (active), and check every register at the end
// Output: mismatching registers per form and victim type, the first few (register, lane, expected, found).
// build + run: hipcc --offload-arch=gfx950 -O2 tools/hw_gpr_idx_probe.hip -o /tmp/probe && GPU_MAX_HW_QUEUES=8 /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t v32u __attribute__((ext_vector_type(32)));
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

template <int FORM>
__global__ void __launch_bounds__(64) k_attack(uint32_t iters, uint32_t *sink) {
    v32u e0, e1, e2, e3, e4, e5;
#pragma unroll
    for (int k = 0; k < 32; k++) e0[k] = e1[k] = e2[k] = e3[k] = e4[k] = e5[k] = 0x7fffffffu;
    const uint32_t lane = threadIdx.x;
    uint32_t x = 0xA5000000u | (blockIdx.x & 0xffffu);  // bits 16..23 = (it + sl) & 255 of the step that writes it: its parity = the lane's
    for (uint32_t it = 0; it < iters; it++) {
        for (uint32_t sl = 0; sl < 192u; sl++) {
            const bool mine = ((lane ^ (it + sl)) & 1u) == 0u;
            if (FORM == 0) {
                const uint64_t m = __ballot(mine);
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
            } else if (FORM == 1) {
                if (mine)
                    asm volatile("s_set_gpr_idx_on %[i], gpr_idx(DST)\n\tv_mov_b32 v64, %[x]\n\ts_set_gpr_idx_off"
                                 : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3),
                                   "+{v[192:223]}"(e4), "+{v[224:255]}"(e5)
                                 : [x] "v"(x), [i] "s"(sl));
            } else if (FORM == 2) {
                const uint64_t m = __ballot(mine);
                asm volatile("s_nop 4\n\ts_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\ts_nop 4\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_nop 4\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [x] "v"(x), [i] "s"(sl), [m] "s"(m));
            } else if (FORM == 4 || FORM == 5) {
                uint64_t m;
                const uint32_t p = mine ? 1u : 0u;
                if (FORM == 4)
                    asm volatile("v_cmp_ne_u32_e64 %[m], 0, %[p]\n\ts_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_set_gpr_idx_off"
                                 : [m] "=&s"(m), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3),
                                   "+{v[192:223]}"(e4), "+{v[224:255]}"(e5)
                                 : [x] "v"(x), [i] "s"(sl), [p] "v"(p));
                else
                    asm volatile("v_cmp_ne_u32_e64 %[m], 0, %[p]\n\ts_nop 1\n\ts_set_gpr_idx_on %[i], gpr_idx(SRC0,DST)\n\tv_cndmask_b32_e64 v64, v64, %[x], %[m]\n\ts_set_gpr_idx_off"
                                 : [m] "=&s"(m), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3),
                                   "+{v[192:223]}"(e4), "+{v[224:255]}"(e5)
                                 : [x] "v"(x), [i] "s"(sl), [p] "v"(p));
                // (something else for the SGPR pair to hold before the next step's v_cmp: a stale read must not see the right mask)
                asm volatile("s_mov_b64 %[m], -1" : [m] "+s"(m));
            } else {
                uint32_t t;
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(SRC0)\n\tv_mov_b32 %[t], v64\n\ts_set_gpr_idx_off"
                             : [t] "=v"(t), "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3),
                               "+{v[192:223]}"(e4), "+{v[224:255]}"(e5)
                             : [i] "s"(sl));
                t = mine ? x : t;
                asm volatile("s_set_gpr_idx_on %[i], gpr_idx(DST)\n\tv_mov_b32 v64, %[t]\n\ts_set_gpr_idx_off"
                             : "+{v[64:95]}"(e0), "+{v[96:127]}"(e1), "+{v[128:159]}"(e2), "+{v[160:191]}"(e3), "+{v[192:223]}"(e4),
                               "+{v[224:255]}"(e5)
                             : [t] "v"(t), [i] "s"(sl));
            }
            x = (x & 0xff00ffffu) | (((sl == 191u ? it + 1u : it + sl + 1u) & 0xffu) << 16);  // (bits 16..23: it + sl of the NEXT step)
        }
    }
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) acc ^= e0[k] ^ e1[k] ^ e2[k] ^ e3[k] ^ e4[k] ^ e5[k];
    // every slot holds 0x7fffffff or an 0xA5...... written by a step of this lane's parity: anything else means the attacker's OWN slots
    // were written under a wrong lane mask
    uint32_t own_bad = 0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const uint32_t vv[6] = {e0[k], e1[k], e2[k], e3[k], e4[k], e5[k]};
        for (int q = 0; q < 6; q++)
            own_bad += (vv[q] != 0x7fffffffu && ((vv[q] >> 24) != 0xA5u || (((vv[q] >> 16) ^ lane) & 1u) != 0u));
    }
    if (own_bad) atomicAdd(&sink[1], own_bad);
    if (acc == 0x12345678u) sink[0] = acc;
}

__device__ __forceinline__ uint32_t expect(uint32_t reg, uint32_t lane, uint32_t block) { return 0x10000000u + (block << 14) + reg * 64u + lane; }

template <int ACTIVE>  // 0: sleeps, 1: VOP2 adds, 2: the same adds in their VOP3 encoding
__global__ void __launch_bounds__(64) k_victim(uint32_t spins, uint32_t *bad, uint32_t *info) {
    v32u r0, r1, r2;  // v32..v127
    const uint32_t lane = threadIdx.x, blk = blockIdx.x & 0x3fffu;
#pragma unroll
    for (int k = 0; k < 32; k++) { r0[k] = expect(32 + k, lane, blk); r1[k] = expect(64 + k, lane, blk); r2[k] = expect(96 + k, lane, blk); }
    for (uint32_t s = 0; s < spins; s++) {
        if (ACTIVE == 1)
            asm volatile(
            "v_add_u32 v32, v32, 1\n"
            "v_add_u32 v33, v33, 1\n"
            "v_add_u32 v34, v34, 1\n"
            "v_add_u32 v35, v35, 1\n"
            "v_add_u32 v36, v36, 1\n"
            "v_add_u32 v37, v37, 1\n"
            "v_add_u32 v38, v38, 1\n"
            "v_add_u32 v39, v39, 1\n"
            "v_add_u32 v40, v40, 1\n"
            "v_add_u32 v41, v41, 1\n"
            "v_add_u32 v42, v42, 1\n"
            "v_add_u32 v43, v43, 1\n"
            "v_add_u32 v44, v44, 1\n"
            "v_add_u32 v45, v45, 1\n"
            "v_add_u32 v46, v46, 1\n"
            "v_add_u32 v47, v47, 1\n"
            "v_add_u32 v48, v48, 1\n"
            "v_add_u32 v49, v49, 1\n"
            "v_add_u32 v50, v50, 1\n"
            "v_add_u32 v51, v51, 1\n"
            "v_add_u32 v52, v52, 1\n"
            "v_add_u32 v53, v53, 1\n"
            "v_add_u32 v54, v54, 1\n"
            "v_add_u32 v55, v55, 1\n"
            "v_add_u32 v56, v56, 1\n"
            "v_add_u32 v57, v57, 1\n"
            "v_add_u32 v58, v58, 1\n"
            "v_add_u32 v59, v59, 1\n"
            "v_add_u32 v60, v60, 1\n"
            "v_add_u32 v61, v61, 1\n"
            "v_add_u32 v62, v62, 1\n"
            "v_add_u32 v63, v63, 1\n"
            "v_add_u32 v64, v64, 1\n"
            "v_add_u32 v65, v65, 1\n"
            "v_add_u32 v66, v66, 1\n"
            "v_add_u32 v67, v67, 1\n"
            "v_add_u32 v68, v68, 1\n"
            "v_add_u32 v69, v69, 1\n"
            "v_add_u32 v70, v70, 1\n"
            "v_add_u32 v71, v71, 1\n"
            "v_add_u32 v72, v72, 1\n"
            "v_add_u32 v73, v73, 1\n"
            "v_add_u32 v74, v74, 1\n"
            "v_add_u32 v75, v75, 1\n"
            "v_add_u32 v76, v76, 1\n"
            "v_add_u32 v77, v77, 1\n"
            "v_add_u32 v78, v78, 1\n"
            "v_add_u32 v79, v79, 1\n"
            "v_add_u32 v80, v80, 1\n"
            "v_add_u32 v81, v81, 1\n"
            "v_add_u32 v82, v82, 1\n"
            "v_add_u32 v83, v83, 1\n"
            "v_add_u32 v84, v84, 1\n"
            "v_add_u32 v85, v85, 1\n"
            "v_add_u32 v86, v86, 1\n"
            "v_add_u32 v87, v87, 1\n"
            "v_add_u32 v88, v88, 1\n"
            "v_add_u32 v89, v89, 1\n"
            "v_add_u32 v90, v90, 1\n"
            "v_add_u32 v91, v91, 1\n"
            "v_add_u32 v92, v92, 1\n"
            "v_add_u32 v93, v93, 1\n"
            "v_add_u32 v94, v94, 1\n"
            "v_add_u32 v95, v95, 1\n"
            "v_add_u32 v96, v96, 1\n"
            "v_add_u32 v97, v97, 1\n"
            "v_add_u32 v98, v98, 1\n"
            "v_add_u32 v99, v99, 1\n"
            "v_add_u32 v100, v100, 1\n"
            "v_add_u32 v101, v101, 1\n"
            "v_add_u32 v102, v102, 1\n"
            "v_add_u32 v103, v103, 1\n"
            "v_add_u32 v104, v104, 1\n"
            "v_add_u32 v105, v105, 1\n"
            "v_add_u32 v106, v106, 1\n"
            "v_add_u32 v107, v107, 1\n"
            "v_add_u32 v108, v108, 1\n"
            "v_add_u32 v109, v109, 1\n"
            "v_add_u32 v110, v110, 1\n"
            "v_add_u32 v111, v111, 1\n"
            "v_add_u32 v112, v112, 1\n"
            "v_add_u32 v113, v113, 1\n"
            "v_add_u32 v114, v114, 1\n"
            "v_add_u32 v115, v115, 1\n"
            "v_add_u32 v116, v116, 1\n"
            "v_add_u32 v117, v117, 1\n"
            "v_add_u32 v118, v118, 1\n"
            "v_add_u32 v119, v119, 1\n"
            "v_add_u32 v120, v120, 1\n"
            "v_add_u32 v121, v121, 1\n"
            "v_add_u32 v122, v122, 1\n"
            "v_add_u32 v123, v123, 1\n"
            "v_add_u32 v124, v124, 1\n"
            "v_add_u32 v125, v125, 1\n"
            "v_add_u32 v126, v126, 1\n"
            "v_add_u32 v127, v127, 1\n"
            
                         : "+{v[32:63]}"(r0), "+{v[64:95]}"(r1), "+{v[96:127]}"(r2));
        else if (ACTIVE == 2)
            asm volatile(
            "v_add_u32_e64 v32, v32, 1\n"
            "v_add_u32_e64 v33, v33, 1\n"
            "v_add_u32_e64 v34, v34, 1\n"
            "v_add_u32_e64 v35, v35, 1\n"
            "v_add_u32_e64 v36, v36, 1\n"
            "v_add_u32_e64 v37, v37, 1\n"
            "v_add_u32_e64 v38, v38, 1\n"
            "v_add_u32_e64 v39, v39, 1\n"
            "v_add_u32_e64 v40, v40, 1\n"
            "v_add_u32_e64 v41, v41, 1\n"
            "v_add_u32_e64 v42, v42, 1\n"
            "v_add_u32_e64 v43, v43, 1\n"
            "v_add_u32_e64 v44, v44, 1\n"
            "v_add_u32_e64 v45, v45, 1\n"
            "v_add_u32_e64 v46, v46, 1\n"
            "v_add_u32_e64 v47, v47, 1\n"
            "v_add_u32_e64 v48, v48, 1\n"
            "v_add_u32_e64 v49, v49, 1\n"
            "v_add_u32_e64 v50, v50, 1\n"
            "v_add_u32_e64 v51, v51, 1\n"
            "v_add_u32_e64 v52, v52, 1\n"
            "v_add_u32_e64 v53, v53, 1\n"
            "v_add_u32_e64 v54, v54, 1\n"
            "v_add_u32_e64 v55, v55, 1\n"
            "v_add_u32_e64 v56, v56, 1\n"
            "v_add_u32_e64 v57, v57, 1\n"
            "v_add_u32_e64 v58, v58, 1\n"
            "v_add_u32_e64 v59, v59, 1\n"
            "v_add_u32_e64 v60, v60, 1\n"
            "v_add_u32_e64 v61, v61, 1\n"
            "v_add_u32_e64 v62, v62, 1\n"
            "v_add_u32_e64 v63, v63, 1\n"
            "v_add_u32_e64 v64, v64, 1\n"
            "v_add_u32_e64 v65, v65, 1\n"
            "v_add_u32_e64 v66, v66, 1\n"
            "v_add_u32_e64 v67, v67, 1\n"
            "v_add_u32_e64 v68, v68, 1\n"
            "v_add_u32_e64 v69, v69, 1\n"
            "v_add_u32_e64 v70, v70, 1\n"
            "v_add_u32_e64 v71, v71, 1\n"
            "v_add_u32_e64 v72, v72, 1\n"
            "v_add_u32_e64 v73, v73, 1\n"
            "v_add_u32_e64 v74, v74, 1\n"
            "v_add_u32_e64 v75, v75, 1\n"
            "v_add_u32_e64 v76, v76, 1\n"
            "v_add_u32_e64 v77, v77, 1\n"
            "v_add_u32_e64 v78, v78, 1\n"
            "v_add_u32_e64 v79, v79, 1\n"
            "v_add_u32_e64 v80, v80, 1\n"
            "v_add_u32_e64 v81, v81, 1\n"
            "v_add_u32_e64 v82, v82, 1\n"
            "v_add_u32_e64 v83, v83, 1\n"
            "v_add_u32_e64 v84, v84, 1\n"
            "v_add_u32_e64 v85, v85, 1\n"
            "v_add_u32_e64 v86, v86, 1\n"
            "v_add_u32_e64 v87, v87, 1\n"
            "v_add_u32_e64 v88, v88, 1\n"
            "v_add_u32_e64 v89, v89, 1\n"
            "v_add_u32_e64 v90, v90, 1\n"
            "v_add_u32_e64 v91, v91, 1\n"
            "v_add_u32_e64 v92, v92, 1\n"
            "v_add_u32_e64 v93, v93, 1\n"
            "v_add_u32_e64 v94, v94, 1\n"
            "v_add_u32_e64 v95, v95, 1\n"
            "v_add_u32_e64 v96, v96, 1\n"
            "v_add_u32_e64 v97, v97, 1\n"
            "v_add_u32_e64 v98, v98, 1\n"
            "v_add_u32_e64 v99, v99, 1\n"
            "v_add_u32_e64 v100, v100, 1\n"
            "v_add_u32_e64 v101, v101, 1\n"
            "v_add_u32_e64 v102, v102, 1\n"
            "v_add_u32_e64 v103, v103, 1\n"
            "v_add_u32_e64 v104, v104, 1\n"
            "v_add_u32_e64 v105, v105, 1\n"
            "v_add_u32_e64 v106, v106, 1\n"
            "v_add_u32_e64 v107, v107, 1\n"
            "v_add_u32_e64 v108, v108, 1\n"
            "v_add_u32_e64 v109, v109, 1\n"
            "v_add_u32_e64 v110, v110, 1\n"
            "v_add_u32_e64 v111, v111, 1\n"
            "v_add_u32_e64 v112, v112, 1\n"
            "v_add_u32_e64 v113, v113, 1\n"
            "v_add_u32_e64 v114, v114, 1\n"
            "v_add_u32_e64 v115, v115, 1\n"
            "v_add_u32_e64 v116, v116, 1\n"
            "v_add_u32_e64 v117, v117, 1\n"
            "v_add_u32_e64 v118, v118, 1\n"
            "v_add_u32_e64 v119, v119, 1\n"
            "v_add_u32_e64 v120, v120, 1\n"
            "v_add_u32_e64 v121, v121, 1\n"
            "v_add_u32_e64 v122, v122, 1\n"
            "v_add_u32_e64 v123, v123, 1\n"
            "v_add_u32_e64 v124, v124, 1\n"
            "v_add_u32_e64 v125, v125, 1\n"
            "v_add_u32_e64 v126, v126, 1\n"
            "v_add_u32_e64 v127, v127, 1\n"
            
                         : "+{v[32:63]}"(r0), "+{v[64:95]}"(r1), "+{v[96:127]}"(r2));
        else
            asm volatile("s_sleep 4" : "+{v[32:63]}"(r0), "+{v[64:95]}"(r1), "+{v[96:127]}"(r2));
    }
    const uint32_t add = ACTIVE ? spins : 0u;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const uint32_t got[3] = {r0[k], r1[k], r2[k]};
        for (int q = 0; q < 3; q++) {
            const uint32_t reg = 32u * (q + 1) + k, want = expect(reg, lane, blk) + add;
            if (got[q] != want) {
                const uint32_t n = atomicAdd(bad, 1u);
                if (n < 16u) { info[4 * n] = (blockIdx.x << 8) | reg; info[4 * n + 1] = lane; info[4 * n + 2] = want; info[4 * n + 3] = got[q]; }
            }
        }
    }
}

template <int FORM, int VACT = 1>
static void run(const char *name, int rounds, uint32_t agrid, uint32_t vgrid, uint32_t iters, uint32_t spins_p, uint32_t spins_a) {
    hipStream_t sa, sv, sw;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sv)); CK(hipStreamCreate(&sw));
    uint32_t *d;
    CK(hipMalloc(&d, 4096));
    uint32_t h[1024];
    unsigned long long tot[2] = {0, 0}, own = 0;
    for (int r = 0; r < rounds; r++) {
        CK(hipMemset(d, 0, 4096));
        hipLaunchKernelGGL(k_victim<0>, dim3(vgrid), dim3(64), 0, sv, spins_p, d + 8, d + 64);
        hipLaunchKernelGGL(k_victim<VACT>, dim3(vgrid), dim3(64), 0, sw, spins_a, d + 9, d + 256);
        hipLaunchKernelGGL(k_attack<FORM>, dim3(agrid), dim3(64), 0, sa, iters, d);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost));
        tot[0] += h[8]; tot[1] += h[9]; own += h[1];
        for (int t = 0; t < 2; t++) {
            const uint32_t nb = h[8 + t], *inf = h + (t ? 256 : 64);
            for (uint32_t k = 0; k < nb && k < 4; k++)
                printf("   %s round %d %s victim: block %u reg v%u lane %u expected %08x found %08x\n", name, r, t ? "active" : "passive",
                       inf[4 * k] >> 8, inf[4 * k] & 0xff, inf[4 * k + 1], inf[4 * k + 2], inf[4 * k + 3]);
        }
    }
    printf("%-44s rounds %d: passive victims' bad registers %llu, active victims' %llu, attackers' own bad slots %llu\n", name, rounds, tot[0], tot[1], own);
    fflush(stdout);
    CK(hipFree(d)); CK(hipStreamDestroy(sa)); CK(hipStreamDestroy(sv)); CK(hipStreamDestroy(sw));
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    const uint32_t agrid = argc > 2 ? atoi(argv[2]) : 4096, vgrid = argc > 3 ? atoi(argv[3]) : 8192;
    const uint32_t iters = argc > 4 ? atoi(argv[4]) : 400, spins_p = argc > 5 ? atoi(argv[5]) : 20000, spins_a = argc > 6 ? atoi(argv[6]) : 2000;
    // timing of one launch of each, to see that they overlap
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    uint32_t *d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    float ms;
    CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_attack<0>, dim3(agrid), dim3(64), 0, 0, iters, d); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf("attackers alone %.2f ms", ms);
    CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_victim<0>, dim3(vgrid), dim3(64), 0, 0, spins_p, d + 8, d + 64); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf(", passive victims alone %.2f ms", ms);
    CK(hipEventRecord(a, 0)); hipLaunchKernelGGL(k_victim<1>, dim3(vgrid), dim3(64), 0, 0, spins_a, d + 9, d + 256); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b)); printf(", active victims alone %.2f ms\n", ms);
    uint32_t h[16]; CK(hipMemcpy(h, d, 64, hipMemcpyDeviceToHost));
    printf("alone: passive bad %u active bad %u attackers' own bad %u\n", h[8], h[9], h[1]);
    CK(hipFree(d));
    run<1>("1 exec-masked v_mov, gpr_idx(DST) [shipped]", rounds, agrid, vgrid, iters, spins_p, spins_a);
    run<0>("0 v_cndmask VOP3, gpr_idx(SRC0,DST) [round 3]", rounds, agrid, vgrid, iters, spins_p, spins_a);
    run<2>("2 form 0 with s_nop 4 around the VOP3", rounds, agrid, vgrid, iters, spins_p, spins_a);
    run<3>("3 indexed read, v_cndmask, indexed write", rounds, agrid, vgrid, iters, spins_p, spins_a);
    run<4>("4 form 0, v_cmp right in front (1 wait state)", rounds, agrid, vgrid, iters, spins_p, spins_a);
    run<5>("5 form 4 with s_nop 1 behind the v_cmp", rounds, agrid, vgrid, iters, spins_p, spins_a);
    printf("-- active victims in VOP3 encoding (v_add_u32_e64)\n");
    run<1, 2>("1 exec-masked v_mov, gpr_idx(DST) [shipped]", rounds, agrid, vgrid, iters, spins_p, spins_a);
    run<0, 2>("0 v_cndmask VOP3, gpr_idx(SRC0,DST) [round 3]", rounds, agrid, vgrid, iters, spins_p, spins_a);
    run<4, 2>("4 form 0, v_cmp right in front (1 wait state)", rounds, agrid, vgrid, iters, spins_p, spins_a);
    return 0;
}
