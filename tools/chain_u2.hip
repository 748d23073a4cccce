// chain_u2.hip -- dev tool: one long list through k_roc_encode_u (round 1) and k_roc_encode_u2 (round 2): identical
// head / words / order, ns per step of each (hipEvents).   usage: chain_u2 [n] [seed] [ub]
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I vector_db_id_compression_amd/csrc tools/chain_u2.hip -o /tmp/chain_u2
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <stdint.h>
#define VIDC_MT_TABLE 1024
#include "roc_u2.h"
using namespace vidc::dev;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

static void u2_div_entry_local(uint32_t d, uint32_t out[4]) {  // == u2_div_entry of common.h
    out[0] = out[1] = out[2] = out[3] = 0;
    if (d == 0) return;
    const uint64_t m = ~0ull / (uint64_t)d;
    out[0] = (uint32_t)m; out[1] = (uint32_t)(m >> 32);
    out[2] = d >= 2u ? (uint32_t)(0x100000000ull / d) : 0xffffffffu;
    out[3] = 0x80000000u / d;
}
template <int UB>
int run(uint32_t n, uint32_t seed, uint32_t P) {
    std::vector<uint64_t> ids;
    { std::mt19937_64 g(seed); std::vector<uint64_t> all(1u << UB); for (size_t i = 0; i < all.size(); i++) all[i] = i;
      std::shuffle(all.begin(), all.end(), g); ids.assign(all.begin(), all.begin() + n); std::sort(ids.begin(), ids.end()); }
    uint64_t offs[2] = {0, n};
    const uint64_t aw = ((uint64_t)n * 37 >> 5) + 9;
    uint32_t wl[1] = {0}, prec[1] = {P};
    uint64_t *d_ids, *d_off, *d_heads; uint32_t *d_wl, *d_prec, *d_nw, *d_dr, *d_st, *d_arena, *d_mt, *d_perm;
    CK(hipMalloc(&d_ids, n * 8)); CK(hipMalloc(&d_off, 16)); CK(hipMalloc(&d_heads, 8));
    CK(hipMalloc(&d_wl, 4)); CK(hipMalloc(&d_prec, 4)); CK(hipMalloc(&d_nw, 4)); CK(hipMalloc(&d_perm, n * 4 + 256));
    CK(hipMalloc(&d_dr, 4)); CK(hipMalloc(&d_st, 4)); CK(hipMalloc(&d_arena, aw * 4)); CK(hipMalloc(&d_mt, 4096));
    CK(hipMemcpy(d_ids, ids.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_off, offs, 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wl, wl, 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_prec, prec, 4, hipMemcpyHostToDevice)); CK(hipMemset(d_mt, 0, 4096));
    uint32_t *d_tab;
    { std::vector<uint32_t> ut((262144 + 1) * 4); for (uint32_t d = 0; d <= 262144; d++) u2_div_entry_local(d, &ut[(size_t)d * 4]);
      CK(hipMalloc(&d_tab, ut.size() * 4)); CK(hipMemcpy(d_tab, ut.data(), ut.size() * 4, hipMemcpyHostToDevice)); }
    RocEncArgs a{};
    a.ids = d_ids; a.offsets = d_off; a.worklist = d_wl; a.nwork = 1; a.heads = d_heads; a.prec = d_prec; a.nwords = d_nw;
    uint32_t *d_prof; CK(hipMalloc(&d_prof, 256)); CK(hipMemset(d_prof, 0, 256));
    a.draws = d_dr; a.status = d_st; a.arena = d_arena; a.arena_stride = 0; a.sid = d_prof; a.mt = d_mt; a.perm = d_perm;
    CK(hipFuncSetAttribute((const void *)k_roc_encode_u<UB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, UGeom<UB>::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)k_roc_encode_u2<UB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, U2Geom<UB>::LDS_BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Res { uint64_t head; uint32_t nw, st; std::vector<uint32_t> words, order; float ms; } r[2];
    for (int which = 0; which < 2; which++) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(d_arena, 0, aw * 4)); CK(hipMemset(d_st, 0xff, 4)); CK(hipMemset(d_perm, 0, n * 4));
            CK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL((k_roc_encode_u<UB, true>), dim3(1), dim3(64), UGeom<UB>::LDS_BYTES, 0, a);
            else hipLaunchKernelGGL((k_roc_encode_u2<UB, true>), dim3(1), dim3(64), U2Geom<UB>::LDS_BYTES, 0, a, (const U2Div *)d_tab);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&r[which].ms, e0, e1));
        }
        CK(hipMemcpy(&r[which].head, d_heads, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&r[which].nw, d_nw, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&r[which].st, d_st, 4, hipMemcpyDeviceToHost));
        r[which].words.resize(r[which].nw < aw ? r[which].nw : 0); r[which].order.resize(n);
        CK(hipMemcpy(r[which].words.data(), d_arena, r[which].words.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r[which].order.data(), d_perm, n * 4, hipMemcpyDeviceToHost));
    }
    bool same = r[0].head == r[1].head && r[0].nw == r[1].nw && r[0].words == r[1].words && r[0].order == r[1].order && r[0].st == 0 && r[1].st == 0;
    size_t fd = 0; while (fd < n && r[0].order[fd] == r[1].order[fd]) fd++;
    printf("UB=%d n=%u P=%u seed=%u: ENCODE old %.1f ns/step, new %.1f ns/step; status %u/%u head %llx/%llx words %u/%u first order diff at %zu -> %s\n",
           UB, n, P, seed, r[0].ms * 1e6 / n, r[1].ms * 1e6 / n, r[0].st, r[1].st, (unsigned long long)r[0].head,
           (unsigned long long)r[1].head, r[0].nw, r[1].nw, fd, same ? "IDENTICAL" : "MISMATCH");
    // ---- decode the (old-kernel) stream with both decoders
    bool dsame = true;
    {
        const uint32_t W0 = r[0].nw;
        CK(hipMemcpy(d_arena, r[0].words.data(), (size_t)W0 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_heads, &r[0].head, 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_nw, &W0, 4, hipMemcpyHostToDevice));
        uint64_t woff[2] = {0, W0}, zero64 = 0, *d_woff, *d_out, *d_z; uint32_t *d_scr, *d_slots, *d_end;
        const uint32_t cap = roc_dec_stack_cap(n, W0);
        CK(hipMalloc(&d_woff, 16)); CK(hipMalloc(&d_out, (size_t)n * 8 + 64)); CK(hipMalloc(&d_z, 8)); CK(hipMalloc(&d_scr, (size_t)cap * 4 + 64));
        CK(hipMalloc(&d_slots, (size_t)n * 4 + 64)); CK(hipMalloc(&d_end, 4));
        CK(hipMemcpy(d_woff, woff, 16, hipMemcpyHostToDevice)); CK(hipMemcpy(d_z, &zero64, 8, hipMemcpyHostToDevice));
        RocDecArgs b{};
        b.offsets = d_off; b.worklist = d_wl; b.out_off = nullptr; b.nwork = 1; b.heads = d_heads; b.prec = d_prec; b.nwords = d_nw; b.draws = d_dr;
        b.words = d_arena; b.word_off = d_woff; b.out = d_out; b.out_rows = nullptr; b.K = 0; b.scratch_words = d_scr; b.scratch_off = d_z;
        b.slots = d_slots; b.slots_off = d_z; b.end_state = d_end; b.status = d_st; b.mt = d_mt;
        CK(hipFuncSetAttribute((const void *)k_roc_decode_u<UB>, hipFuncAttributeMaxDynamicSharedMemorySize, UGeom<UB>::LDS_BYTES));
        CK(hipFuncSetAttribute((const void *)k_roc_decode_u2<UB>, hipFuncAttributeMaxDynamicSharedMemorySize, U2Geom<UB>::LDS_BYTES));
        std::vector<uint64_t> outv[2]; float dms[2]; uint32_t dst[2], dend[2];
        for (int which = 0; which < 2; which++) {
            for (int rep = 0; rep < 2; rep++) {
                CK(hipMemset(d_out, 0xee, (size_t)n * 8)); CK(hipMemset(d_st, 0xff, 4)); CK(hipMemset(d_end, 0xff, 4));
                CK(hipEventRecord(e0, 0));
                if (which == 0) hipLaunchKernelGGL((k_roc_decode_u<UB>), dim3(1), dim3(64), UGeom<UB>::LDS_BYTES, 0, b);
                else hipLaunchKernelGGL((k_roc_decode_u2<UB>), dim3(1), dim3(64), U2Geom<UB>::LDS_BYTES, 0, b, (const U2Div *)d_tab);
                CK(hipEventRecord(e1, 0));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&dms[which], e0, e1));
            }
#ifdef U2_PROF2
            if (which == 1) { uint32_t dbg[4]; CK(hipMemcpy(dbg, d_slots, 16, hipMemcpyDeviceToHost)); printf("   decode dbg: slow steps %u, asm entries %u, bits %u, n %u\n", dbg[0], dbg[1], dbg[2], dbg[3]); }
#endif
            outv[which].resize(n);
            CK(hipMemcpy(outv[which].data(), d_out, (size_t)n * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(&dst[which], d_st, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&dend[which], d_end, 4, hipMemcpyDeviceToHost));
        }
        size_t dd = 0; while (dd < n && outv[0][dd] == outv[1][dd]) dd++;
        size_t od = 0; while (od < n && outv[1][od] == r[0].order[od]) od++;
        dsame = dd == n && dst[0] == 0 && dst[1] == 0 && dend[0] == dend[1];
        printf("                         DECODE old %.1f ns/step, new %.1f ns/step; status %u/%u end state %u/%u first diff old/new at %zu, new/encoder order at %zu -> %s\n",
               dms[0] * 1e6 / n, dms[1] * 1e6 / n, dst[0], dst[1], dend[0], dend[1], dd, od, dsame ? "IDENTICAL" : "MISMATCH");
    }
    same = same && dsame;
#ifdef U2_PROF2
    { uint64_t q[8]; CK(hipMemcpy(q, d_prof, 64, hipMemcpyDeviceToHost)); printf("  exits: t==63 %llu, ring %llu, headcheck %llu, sum of t %llu\n", (unsigned long long)q[4], (unsigned long long)q[5], (unsigned long long)q[6], (unsigned long long)q[7]);
      printf("  setup %.0f ticks (%.1f us), chain phase %.1f ticks/step, inside asm %.1f ticks/step over %llu asm entries\n", (double)q[2], q[2] * 0.000417,
             (double)q[3] / n, (double)q[0] / n, (unsigned long long)q[1]); }
#endif
#ifdef U2_PROF
    uint32_t prof[14]; CK(hipMemcpy(prof, d_prof, 56, hipMemcpyDeviceToHost));
    const char *names[14] = {"branch -> top (+ first probe)", "EMPTY (probe cost)", "fix-up chain .. v_mov k (12)", "add64, L1 cmp, 3 readlanes (5)", "bitmap valu, ff1, row read (8)",
        "ds_write, slice 0 (8)", "L1 readlane.. L2 cmp, E1, ff1, entry (9)", "ds_read, slice 1, L2 readlane, division .. waitcnt (24)", "L3a valu: bcnt, dpp, cmp (10)",
        "L3a ff1, salu, 3 readlanes, sub (10)", "L3b: mbcnt, cmp, and, ff1, x (6)", "push 2 + order (7)", "row write-back (3)", "exit tests (7)"};
    double tot = 0, pc = (double)prof[1] / n;
    for (int q = 0; q < 14; q++) { printf("  %-58s %8.1f  net %7.1f cycles/step\n", names[q], (double)prof[q] / n, (double)prof[q] / n - pc); tot += prof[q]; }
    printf("  total %.1f, net of probes %.1f cycles/step\n", tot / n, tot / n - 14 * pc);
#endif
    return same ? 0 : 2;
}
int main(int argc, char **argv) {
    uint32_t n = argc > 1 ? atoi(argv[1]) : 52114, seed = argc > 2 ? atoi(argv[2]) : 1, ub = argc > 3 ? atoi(argv[3]) : 20;
    uint32_t P = argc > 4 ? atoi(argv[4]) : ub;
    return ub == 18 ? run<18>(n, seed, P) : run<20>(n, seed, P);
}
