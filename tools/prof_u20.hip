// prof_u20.hip -- dev tool: per-section cycle breakdown of one k_roc_encode_u<20> chain (s_memtime instrumented).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVIDC_PROF -I vector_db_id_compression_amd/csrc tools/prof_u20.hip -o /tmp/prof_u20
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>
#define VIDC_MT_TABLE 1024
#include "roc_u.h"
using namespace vidc::dev;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main() {
    const uint32_t n = 50000;
    std::vector<uint64_t> ids;
    { std::mt19937_64 g(1); std::vector<uint64_t> all(1 << 20); for (size_t i = 0; i < all.size(); i++) all[i] = i;
      std::shuffle(all.begin(), all.end(), g); ids.assign(all.begin(), all.begin() + n); std::sort(ids.begin(), ids.end()); }
    uint64_t offs[2] = {0, n}, aoff[2] = {0, roc_arena_at(nullptr, 0, 0)};
    aoff[1] = ((uint64_t)n * 37 >> 5) + 9;  // closed-form arena layout (roc_arena_at)
    uint32_t wl[1] = {0}, prec[1] = {20};
    uint64_t *d_ids, *d_off, *d_aoff, *d_heads, *d_prof; uint32_t *d_wl, *d_prec, *d_nw, *d_dr, *d_st, *d_arena, *d_mt;
    CK(hipMalloc(&d_ids, n * 8)); CK(hipMalloc(&d_off, 16)); CK(hipMalloc(&d_aoff, 16)); CK(hipMalloc(&d_heads, 8));
    CK(hipMalloc(&d_prof, 64)); CK(hipMalloc(&d_wl, 4)); CK(hipMalloc(&d_prec, 4)); CK(hipMalloc(&d_nw, 4));
    CK(hipMalloc(&d_dr, 4)); CK(hipMalloc(&d_st, 4)); CK(hipMalloc(&d_arena, aoff[1] * 4)); CK(hipMalloc(&d_mt, 4096));
    CK(hipMemcpy(d_ids, ids.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_off, offs, 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_aoff, aoff, 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d_wl, wl, 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_prec, prec, 4, hipMemcpyHostToDevice)); CK(hipMemset(d_mt, 0, 4096));
    RocEncArgs a{};
    a.ids = d_ids; a.offsets = d_off; a.worklist = d_wl; a.nwork = 1; a.heads = d_heads; a.prec = d_prec; a.nwords = d_nw;
    a.draws = d_dr; a.status = d_st; a.arena = d_arena; a.arena_stride = 0; a.sid = (uint32_t *)d_prof; a.mt = d_mt;
    CK(hipFuncSetAttribute((const void *)k_roc_encode_u<20, false>, hipFuncAttributeMaxDynamicSharedMemorySize, UGeom<20>::LDS_BYTES));
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((k_roc_encode_u<20, false>), dim3(1), dim3(64), UGeom<20>::LDS_BYTES, 0, a);
        CK(hipDeviceSynchronize());
    }
    uint64_t prof[8]; CK(hipMemcpy(prof, d_prof, 64, hipMemcpyDeviceToHost));
    const char *names[8] = {"recip_block (per 64)", "ws_prepare", "idx_pop (div)", "level 1+2", "level 3 + bit", "updates + LDS store", "id_push", "-"};
    double tot = 0; for (int i = 0; i < 7; i++) tot += prof[i];
    for (int i = 0; i < 7; i++) printf("%-24s %8.1f cycles/step\n", names[i], (double)prof[i] / n);
    printf("%-24s %8.1f cycles/step (instrumented; each probe adds its own s_memtime cost)\n", "total", tot / n);
    return 0;
}
