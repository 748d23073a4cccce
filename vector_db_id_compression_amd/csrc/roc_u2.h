// roc_u2.h -- hand-scheduled "universe bitmap" ROC chain kernels (round 2): the latency-critical path of the
// BASELINE configuration (one 52 114-id list = 52 114 dependent codec steps on ONE wavefront).
//
// Same bitstream as roc_u.h / codec.cpp:21-152; what changed is the shape of the step:
//
//  * head' = ID_push(q, x) is B(q) + c(x): every renormalisation decision of the two 16-bit slices depends on
//    q = head div nmax alone, and x only adds c(x) = x_lo * 2^p1 + x_hi (or x_hi when the second slice renormalises)
//    below the low zero bits of B.  The 64-bit division of the NEXT index pop therefore runs on B while the select is
//    in flight (all lanes divide B by THEIR divisor with per-lane 64-bit reciprocals), and only a 21-bit fix-up
//    (r_B + c(x)) divmod nmax' -- mul_hi by a 32-bit reciprocal, one multiply-add, one min -- sits between two selects.
//  * order statistics: ONE exclusive prefix counter per level, stored in REVERSED lane order (lane L <-> block 63 - L),
//    so that "counter <= k" is true from the wanted lane upwards: v_cmp + s_ff1 + v_readlane gives both the index and
//    the amount to subtract (the inclusive / exclusive counter pairs and the DPP shift of roc_u.h are gone); level-2
//    rows are 64 unpacked VGPRs read with s_set_gpr_idx.
//  * the steady-state loop is one asm statement with a fixed register map (below).  A lone wavefront issues one
//    instruction every 4 cycles whatever its type (tools/ubench_issue.hip) and a scalar instruction issued less than
//    ~16 cycles after a VALU instruction that wrote an SGPR / VCC stalls for the difference (tools/ubench_sel*.hip);
//    the body is ordered so that those windows and the LDS round trip hold independent vector work (the division of
//    B, counter updates, the removal of the previous id from the bitmap).
//  * renormalisation is branch-free (s_cselect + unconditional ring write + stack-pointer add of the condition).
//
// Anything rare (index-pop renormalisation, stack underflow, nmax <= 2, ring spills) leaves the asm loop at a step
// boundary and runs one generic C++ step (u2_slow_step) on the same data structures.
// Lists the fast path cannot represent (precision > 20, an id >= 2^precision, duplicates) are handed back to the
// general kernels through VIDC_ST_PENDING_SORT like roc_u.h does for multisets.
#pragma once
#include "roc_u.h"

namespace vidc {
namespace dev {

template <int UB>
struct U2Geom {
    static_assert(UB == 18 || UB == 20, "universe bits");
    static constexpr uint32_t GSH = UB == 20 ? 2u : 0u;   // log2(bitmap words per level-2 entry)
    static constexpr uint32_t G = 1u << GSH;
    static constexpr uint32_t ESH = 6u + GSH;             // id bits below the entry index
    static constexpr uint32_t NE = 4096u;                 // 64 blocks x 64 entries
    static constexpr uint32_t BITMAP_BYTES = NE * G * 8u;
    static constexpr uint32_t LDS_BYTES = BITMAP_BYTES + 16u;  // + a dummy word: target of the "no pending removal" store
};

// LDS word index of id x: entries are stored in REVERSED order (entry e' = e ^ 4095), words inside an entry ascending
template <int UB>
__device__ __forceinline__ uint32_t u2_word_of(uint32_t x) {
    using U = U2Geom<UB>;
    return (((x >> U::ESH) ^ (U::NE - 1u)) << U::GSH) | ((x >> 6) & (U::G - 1u));
}

// Per-divisor constants (16 bytes, table built once per context, vidc_ctx::d_u2tab, VIDC_ROC_MAX_LIST + 1 entries):
//   x = floor((2^64 - 1) / d) low, y = high, z = floor(2^32 / d) (d >= 2), w = floor(2^31 / d).  Entry 0 is all zero.
typedef uint4 U2Div;  // filled by u2_div_entry (common.h)

// exclusive counters in reversed lane order, see the header comment.  rows: ra = rows 0..31, rb = rows 32..63.
template <int UB>
__device__ __forceinline__ void u2_build_counts(const uint64_t *bm, uint32_t &E1, v32u &ra, v32u &rb) {
    using U = U2Geom<UB>;
    const uint32_t lane = lane_id();
    uint32_t running = 0;
    E1 = 0;
#pragma unroll
    for (int L1 = 63; L1 >= 0; L1--) {
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t g = 0; g < U::G; g++) cnt += popc64(bm[((uint32_t)L1 * 64u + lane) * U::G + g]);
        uint32_t inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)inc, o, 64);
            if (lane >= (uint32_t)o) inc += v;
        }
        const uint32_t tot = rl(inc, 63);
        const uint32_t row = tot - inc;  // elements of this block in entries with a LARGER lane index (smaller ids)
        if (L1 < 32) ra[L1] = row; else rb[L1 - 32] = row;
        E1 = lane == (uint32_t)L1 ? running : E1;
        running += tot;
    }
}

// Row L1 of the level-2 counters by register index.  A C++ subscript with a run-time index would make the compiler
// keep both vectors in scratch memory around every asm statement; the rows stay pinned to v64..v127 instead.
__device__ __forceinline__ uint32_t u2_row_get(v32u &ra, v32u &rb, uint32_t L1) {
    uint32_t v;
    asm volatile("s_set_gpr_idx_on %3, gpr_idx(SRC0)\n\tv_mov_b32 %0, v64\n\ts_set_gpr_idx_off"
                 : "=v"(v), "+{v[64:95]}"(ra), "+{v[96:127]}"(rb) : "s"(L1));
    return v;
}
__device__ __forceinline__ void u2_row_set(v32u &ra, v32u &rb, uint32_t L1, uint32_t v) {
    asm volatile("s_set_gpr_idx_on %2, gpr_idx(DST)\n\tv_mov_b32 v64, %3\n\ts_set_gpr_idx_off"
                 : "+{v[64:95]}"(ra), "+{v[96:127]}"(rb) : "s"(L1), "v"(v));
}

// one generic encode step on the reversed structures (rare path): codec.cpp:131-137
template <int UB>
__device__ __forceinline__ uint32_t u2_slow_step(uint64_t &head, WStack &st, uint32_t nmax, uint32_t &E1, v32u &ra, v32u &rb,
                                              uint64_t *bm, uint32_t p0, uint32_t p1) {
    using U = U2Geom<UB>;
    const uint32_t lane = lane_id();
    ws_prepare(st);
    const uint32_t lq = 0x80000000u / nmax;
    uint32_t k = ans_idx_pop(head, st, nmax, lq * nmax, ~0ull / (uint64_t)nmax);
    const uint32_t L1 = ff1(ballot(E1 <= k));
    k -= rl(E1, L1);
    uint32_t row = u2_row_get(ra, rb, L1);
    const uint32_t L2 = ff1(ballot(row <= k));
    k -= rl(row, L2);
    const uint32_t e = L1 * 64u + L2;
    uint32_t g = 0;
    uint64_t W = rfl64(bm[e * U::G]);
    for (; g + 1u < U::G; g++) {
        const uint32_t pc = popc64(W);
        if (k < pc) break;
        k -= pc;
        W = rfl64(bm[e * U::G + g + 1u]);
    }
    const uint32_t b = ff1(ballot(mbcnt(W) == k) & W);
    const uint32_t x = ((e ^ (U::NE - 1u)) << U::ESH) | (g << 6) | b;
    E1 -= lane < L1 ? 1u : 0u;
    row -= lane < L2 ? 1u : 0u;
    u2_row_set(ra, rb, L1, row);
    bm[e * U::G + g] = W & ~(1ull << b);
    wave_sync();
    ans_id_push(head, st, x, p0, p1);
    return x;
}

// ---------------------------------------------------------------------------------------------------------------
// The encode loop.  One asm statement runs a list from a plain ANS state to the next rare event.  Register map
// (pinned through the operand constraints of U2_ENC_ASM):
//   v2  lane id          v3  (lane & 3) * 8 [G = 4] / 0     v4  E1 (level 1)     v5  ANS stack ring   v6  order ring
//   v7  m_lo  v8  m_hi  v9  d  v10 m32  v11 thrs  v29 -d    (per lane: lane j owns divisor D0 - j)   v54:57 next block
//   v12 k     v13 row   v14 x_hi  v15 x_lo  v16 s  v17 q^  v18 r  v19 r - d  v20:21 {qf, 0}  v52 k per lane
//   v22:23 bit mask  v24:25 new word  v26 its address  v27 tmp  v28 LDS address  v30:31 loaded word(s)  v58 tmp
//   v32 popcount  v33 v34 prefix  v35 exclusive prefix  v49 mbcnt
//   v36:37 {ph, 0}  v38:39 U  v40:41 {U.lo, 0}  v42:43 W  v44:45 Z  v46:47 q^ of B (per lane)  v48 r of B (per lane)
//   v50:51 q = q^ + qf (per lane)           v64..v127 level-2 rows
//   s40 x      s41 mul    s42 k      s43 L1    s44 L2    s45 entry   s46 x base  s47 g     s48 b     s49 E1[L1]
//   s50:51 q   s52:53 a   s54:55 A   s56:57 b  s58:59 B  s60 sp      s61 lo      s62 slot of the second push
//   s63 row[L2]  s64 excl[g]  s65 word address  s66:67 word  s68 tmp  s69 t (lane of the current divisor)
//   s70 t_end  s71 limit  s72 tmp  s73 T0 = 2^(31-p0)  s74 T1 = 2^(31-p1)  s75 2^p1  s76 p0  s77 p1
//   s78:81 carry scratch  s82:83 mask scratch  s85 D0 (divisor of lane 0)
//   s86 fast steps left from lane 0 of the block  s87 order base  s88:89 order pointer  s90:91 arena pointer
//   s92 arena capacity  s93 error flags  s94:95 divisor table  s96:97 saved exec  s98 s99 tmp
// Invariant at label 1 (top of step i, divisor lane t = s69):
//   head_i = B + c(x_{i-1}) with c = x_lo * mul + x_hi; lanes hold (v46:47, v48) = (q^, r) with B = q^ d_lane + r,
//   0 <= r < 2 d_lane; the bit of x_{i-1} is still set in the bitmap (word s66:67 at address s65; it is cleared in
//   the first window of the step).
// Blocks: lanes 0..62 of a block serve as "current" divisor, lane t + 1 as the next one; after lane 62 the constants
// of the next 64 divisors (prefetched from the table one block ahead) are installed and lane 63's division result
// moves to lane 0.  The order ring (v6) is indexed by t and flushed at block ends and on exit.
#ifdef U2_PROF  // dev builds (tools/chain_u2.hip -DU2_PROF): cycles per section, accumulated per lane-uniform VGPR
#define U2P(i) "s_memtime s[96:97]\n s_waitcnt lgkmcnt(0)\n s_sub_u32 s99, s96, s98\n s_mov_b32 s98, s96\n v_add_u32 v" #i ", s99, v" #i "\n"
#define U2P_OPERANDS , "+{v130}"(prof[0]), "+{v131}"(prof[1]), "+{v132}"(prof[2]), "+{v133}"(prof[3]), "+{v134}"(prof[4]), "+{v135}"(prof[5]), \
    "+{v136}"(prof[6]), "+{v137}"(prof[7]), "+{v138}"(prof[8]), "+{v139}"(prof[9]), "+{v140}"(prof[10]), "+{v141}"(prof[11]), "+{v142}"(prof[12]), "+{v143}"(prof[13])
#else
#define U2P(i) ""
#define U2P_OPERANDS
#endif
#define U2_DIV \
    "v_mul_hi_u32 v36, s58, v7\n"                      /* B div d (all lanes): mulhi64(B, m) */ \
    "v_mad_u64_u32 v[38:39], s[78:79], s58, v8, v[36:37]\n" \
    "v_mov_b32 v40, v38\n" \
    "v_mad_u64_u32 v[42:43], s[78:79], s59, v7, v[40:41]\n" \
    "v_add_co_u32_e64 v44, s[78:79], v39, v43\n" \
    "v_addc_co_u32_e64 v45, s[80:81], 0, 0, s[78:79]\n" \
    "v_mad_u64_u32 v[46:47], s[78:79], s59, v8, v[44:45]\n"
#define U2_DIV_REM \
    "v_mul_lo_u32 v48, v46, v9\n" \
    "v_sub_u32 v48, s58, v48\n"                        /* r_B in [0, 2d) */
// constants of the block whose lane 0 divides by s85: v54:57 -> v7 v8 v10 v11, d, -d; prefetch of the block after it
#define U2_INSTALL \
    "v_mov_b32 v7, v54\n v_mov_b32 v8, v55\n v_mov_b32 v10, v56\n v_mov_b32 v11, v57\n" \
    "v_sub_u32 v9, s85, v2\n" \
    "v_max_i32 v9, 1, v9\n" \
    "v_sub_u32 v29, 0, v9\n" \
    "s_sub_u32 s68, s85, 63\n" \
    "v_sub_u32 v58, s68, v2\n" \
    "v_max_i32 v58, 0, v58\n" \
    "v_lshlrev_b32 v58, 4, v58\n" \
    "global_load_dwordx4 v[54:57], v58, s[94:95]\n"
#define U2_ENC_ENTRY \
    "v_sub_u32 v58, s85, v2\n" \
    "v_max_i32 v58, 0, v58\n" \
    "v_lshlrev_b32 v58, 4, v58\n" \
    "global_load_dwordx4 v[54:57], v58, s[94:95]\n" \
    "s_waitcnt vmcnt(0)\n" \
    U2_INSTALL U2_DIV U2_DIV_REM \
    "s_mov_b32 s69, 0\n" \
    "s_min_u32 s70, s86, 63\n"

#define U2_ENC_TOP "1:\n" U2_ENC_TOP_NL
#define U2_ENC_TOP_NL \
    U2P(130) U2P(131) \
    "v_lshrrev_b32_e64 v14, 16, s40\n"                 /* x_hi */ \
    "v_bfe_u32 v15, s40, 0, 16\n"                      /* x_lo */ \
    "v_add_u32 v16, v48, v14\n"                        /* s = r_B + x_hi */ \
    "v_mad_u32_u24 v16, v15, s41, v16\n"               /*   + x_lo * mul */ \
    "v_mul_hi_u32 v17, v16, v10\n"                     /* q^ = mulhi(s, 2^32 / d) */ \
    "v_mad_i32_i24 v18, v17, v29, v16\n"               /* r = s - q^ d in [0, 2d) */ \
    "v_sub_u32 v19, v18, v9\n" \
    "v_min_u32 v52, v18, v19\n"                        /* k (of this lane's divisor) */ \
    "v_ashrrev_i32 v27, 31, v19\n"                     /* -1 if r < d */ \
    "v_add3_u32 v20, v17, v27, 1\n"                    /* qf = q^ + (r >= d) */ \
    "v_readlane_b32 s42, v52, s69\n"                   /* k of the step */ \
    "v_add_co_u32_e64 v50, s[78:79], v46, v20\n"       /* q = head div d (per lane) */ \
    "v_addc_co_u32_e64 v51, s[78:79], 0, v47, s[78:79]\n" \
    "v_mov_b32 v12, s42\n" U2P(132) \
    "v_cmp_le_u32 vcc, v4, v12\n"                      /* level 1 */ \
    "v_lshlrev_b64 v[22:23], s40, 1\n"                 /* removal of x_{i-1} from the bitmap */ \
    "v_bfi_b32 v24, v22, 0, s66\n" \
    "v_readlane_b32 s50, v50, s69\n" \
    "v_readlane_b32 s51, v51, s69\n" \
    U2P(133) \
    "v_bfi_b32 v25, v23, 0, s67\n" \
    "v_mov_b32 v26, s65\n" \
    "ds_write_b64 v26, v[24:25]\n" \
    "s_ff1_i32_b64 s43, vcc\n" \
    "s_set_gpr_idx_on s43, gpr_idx(SRC0)\n" \
    "v_mov_b32 v13, v64\n"                             /* row L1 */ \
    "s_set_gpr_idx_off\n" U2P(134) \
    "s_and_b32 m0, s60, 63\n"                          /* slice 0 (codec.cpp:65-76), branch-free */ \
    "s_cmp_ge_u32 s51, s73\n" \
    "v_writelane_b32 v5, s50, m0\n" \
    "s_cselect_b32 s52, s51, s50\n" \
    "s_cselect_b32 s53, 0, s51\n" \
    "s_addc_u32 s60, s60, 0\n" \
    "s_lshl_b64 s[54:55], s[52:53], s76\n" U2P(135) \
    "v_readlane_b32 s49, v4, s43\n" \
    "v_subrev_u32 v27, s43, v2\n"                      /* lane - L1 */ \
    "v_subrev_u32 v12, s49, v12\n" \
    "v_cmp_le_u32 vcc, v13, v12\n"                     /* level 2 */ \
    "v_ashrrev_i32 v27, 31, v27\n" \
    "v_add_u32 v4, v4, v27\n"                          /* E1 -= 1 in lanes below L1 */ \
    "s_ff1_i32_b64 s44, vcc\n" \
    "s_lshl_b32 s45, s43, 6\n" \
    "s_or_b32 s45, s45, s44\n" U2P(136)

#define U2_ENC_MID_G4 \
    "v_lshl_add_u32 v28, s45, 5, v3\n" \
    "ds_read_b64 v[30:31], v28\n"
#define U2_ENC_MID_G1 \
    "v_lshlrev_b32_e64 v28, 3, s45\n" \
    "ds_read_b64 v[30:31], v28\n"

#define U2_ENC_SLICE1(XSH) U2_ENC_SLICE1_S("s_lshl_b32 s46, s46, " XSH "\n", "0xfff")
#define U2_ENC_SLICE1_X(XSH, XORC) U2_ENC_SLICE1_S("s_lshl_b32 s46, s46, " XSH "\n", XORC)
// four words per entry: the entry's id bits are shifted together with the word number in U2_ENC_L3_G4 (one fused shift-add)
#define U2_ENC_SLICE1_G4 U2_ENC_SLICE1_S("", "0xfff")
#define U2_ENC_SLICE1_S(SHIFT, XORC)                   /* XORC: entries of the bitmap - 1 (reversed entry order) */ \
    "s_and_b32 s62, s60, 63\n"                         /* slice 1 */ \
    "s_cmp_ge_u32 s55, s74\n" \
    "s_cselect_b32 s41, 0, s75\n" \
    "s_cselect_b32 s56, s55, s54\n" \
    "s_cselect_b32 s57, 0, s55\n" \
    "s_addc_u32 s60, s60, 0\n" \
    "s_lshl_b64 s[58:59], s[56:57], s77\n"             /* B */ \
    "s_xor_b32 s46, s45, " XORC "\n" \
    SHIFT                                              /* id bits of the entry */ \
    "v_readlane_b32 s63, v13, s44\n" \
    "v_subrev_u32 v59, s44, v2\n"                      /* lane - L2 */ \
    "v_subrev_u32 v12, s63, v12\n" \
    "v_mul_hi_u32 v36, s58, v7\n"                      /* B div d (all lanes): mulhi64(B, m) */ \
    "v_mad_u64_u32 v[38:39], s[78:79], s58, v8, v[36:37]\n" \
    "v_mov_b32 v40, v38\n" \
    "v_mad_u64_u32 v[42:43], s[78:79], s59, v7, v[40:41]\n"

#define U2_ENC_L3_G4 \
    "s_waitcnt lgkmcnt(0)\n" U2P(137) \
    "v_bcnt_u32_b32 v32, v30, 0\n" \
    "v_bcnt_u32_b32 v32, v31, v32\n" \
    "v_add_co_u32_e64 v44, s[78:79], v39, v43\n" \
    "v_addc_co_u32_e64 v45, s[80:81], 0, 0, s[78:79]\n" \
    "v_add_u32_dpp v33, v32, v32 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_mad_u64_u32 v[46:47], s[78:79], s59, v8, v[44:45]\n" \
    "v_ashrrev_i32 v59, 31, v59\n" \
    "v_add_u32_dpp v34, v33, v33 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_cmp_gt_u32 vcc, v34, v12\n"                     /* word of the entry */ \
    "v_sub_u32 v35, v34, v32\n" U2P(138) \
    "v_mul_lo_u32 v48, v46, v9\n" \
    "v_add_u32 v13, v13, v59\n"                        /* row -= 1 in lanes below L2 */ \
    "s_ff1_i32_b64 s47, vcc\n" \
    "s_lshl2_add_u32 s46, s46, s47\n"                  /* (entry, word) */ \
    "s_lshl_b32 s46, s46, 6\n" \
    "s_lshl2_add_u32 s65, s45, s47\n"                  /* (entry * 4 + word) * 8 */ \
    "s_lshl_b32 s65, s65, 3\n" \
    "v_readlane_b32 s64, v35, s47\n" \
    "v_readlane_b32 s66, v30, s47\n" \
    "v_readlane_b32 s67, v31, s47\n" \
    "v_subrev_u32 v12, s64, v12\n" U2P(139)
#define U2_ENC_L3_G1 \
    "s_lshl_b32 s65, s45, 3\n" \
    "v_add_co_u32_e64 v44, s[78:79], v39, v43\n" \
    "v_addc_co_u32_e64 v45, s[80:81], 0, 0, s[78:79]\n" \
    "v_ashrrev_i32 v59, 31, v59\n" \
    "v_mad_u64_u32 v[46:47], s[78:79], s59, v8, v[44:45]\n" \
    "v_add_u32 v13, v13, v59\n" \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_readfirstlane_b32 s66, v30\n" \
    "v_readfirstlane_b32 s67, v31\n" \
    "v_mul_lo_u32 v48, v46, v9\n"

// The output ring (64 words, a step pushes at most two) is checked by the LAST copy of the loop body only, for the up to eight
// steps that follow: U2_RING_ROOM words at most on entry (the C++ glue and U2_ENC_OUTER spill from there on), 48 + 2 * 8 = 64.
#define U2_RING_ROOM 48
#define U2_RING_CHECK \
    "s_sub_u32 s68, s60, s61\n"                        /* ring nearly full */ \
    "s_cmp_ge_u32 s68, 48\n" \
    "s_cselect_b32 s71, 0, s71\n"
#define U2_ENC_BOT(ORDER, LSHR) U2_ENC_BOT_T(ORDER, LSHR, U2_RING_CHECK, "s_cbranch_scc1 1b\n")
#define U2_ENC_BOT_T(ORDER, LSHR, RING, TAIL) \
    "v_mbcnt_lo_u32_b32 v49, s66, 0\n" \
    "v_mbcnt_hi_u32_b32 v49, s67, v49\n" \
    "v_cmp_eq_u32 vcc, v49, v12\n"                     /* bit of the word */ \
    "v_sub_u32 v48, s58, v48\n"                        /* r_B in [0, 2d) */ \
    "s_and_b64 s[82:83], vcc, s[66:67]\n" \
    "s_ff1_i32_b64 s48, s[82:83]\n" \
    "s_or_b32 s40, s46, s48\n"                         /* x */ U2P(140) \
    "s_mov_b32 m0, s62\n" \
    "s_and_b32 s68, s40, 0xffff\n" \
    "s_or_b32 s68, s68, s54\n" \
    "v_writelane_b32 v5, s68, m0\n"                    /* word of the second slice (kept only if it renormalised) */ \
    ORDER U2P(141) \
    "s_set_gpr_idx_on s43, gpr_idx(DST)\n" \
    "v_mov_b32 v64, v13\n" \
    "s_set_gpr_idx_off\n" U2P(142) \
    LSHR                                               /* next index pop must not renormalise (u2_needs_generic): SCC = B >= 2^31 */ \
    "s_cselect_b32 s71, s70, 0\n" \
    "s_cmp_ge_u32 s59, 0x7ff80000\n" \
    "s_cselect_b32 s71, 0, s71\n" \
    RING \
    "s_add_u32 s69, s69, 1\n" U2P(143) \
    "s_cmp_lt_u32 s69, s71\n" \
    TAIL
#define U2_ENC_ORDER \
    "s_mov_b32 m0, s69\n" \
    "s_lshr_b64 s[98:99], s[58:59], 31\n"             /* (first instruction of the exit test, SCC = B >= 2^31: m0 settles meanwhile) */ \
    "v_writelane_b32 v6, s40, m0\n"

// store order-ring lanes [0, s72) at order[s87 ...]; s87 += s72
#define U2_ORDER_FLUSH \
    "v_cmp_gt_u32 vcc, s72, v2\n" \
    "v_add_u32 v58, s87, v2\n" \
    "v_lshlrev_b32 v58, 2, v58\n" \
    "s_and_saveexec_b64 s[96:97], vcc\n" \
    "global_store_dword v58, v6, s[88:89]\n" \
    "s_mov_b64 exec, s[96:97]\n" \
    "s_add_u32 s87, s87, s72\n"

// what the fast loop falls into: ring spill, block change, and the decision to go on
#define U2_ENC_OUTER(FLUSH) \
    "s_lshr_b64 s[98:99], s[58:59], 31\n"             /* the index-pop test of the next step */ \
    "s_cselect_b32 s99, 0, 1\n" \
    "s_cmp_ge_u32 s59, 0x7ff80000\n" \
    "s_cselect_b32 s99, 1, s99\n" \
    "s_sub_u32 s68, s60, s61\n" \
    "s_cmp_lt_u32 s68, 48\n"                           /* (U2_RING_ROOM) */ \
    "s_cbranch_scc1 3f\n" \
    "v_subrev_u32 v27, s61, v2\n"                      /* spill the 32 oldest ring words (ws_spill32) */ \
    "v_and_b32 v27, 63, v27\n" \
    "v_add_u32 v58, s61, v27\n" \
    "v_cmp_gt_u32 vcc, 32, v27\n" \
    "v_cmp_gt_u32 s[96:97], s92, v58\n" \
    "v_lshlrev_b32 v58, 2, v58\n" \
    "s_and_b64 vcc, vcc, s[96:97]\n" \
    "s_and_saveexec_b64 s[96:97], vcc\n" \
    "global_store_dword v58, v5, s[90:91]\n" \
    "s_mov_b64 exec, s[96:97]\n" \
    "s_add_u32 s61, s61, 32\n" \
    "s_cmp_gt_u32 s61, s92\n" \
    "s_cselect_b32 s68, 1, 0\n" \
    "s_or_b32 s93, s93, s68\n" \
    "3:\n" \
    "s_cmp_lt_u32 s69, 63\n" \
    "s_cbranch_scc1 5f\n" \
    "s_waitcnt vmcnt(0)\n"                             /* block change: the prefetched constants have landed; waiting */ \
    "s_mov_b32 s72, 63\n"                              /* here, before the ring store, keeps the store's latency off the chain */ \
    FLUSH \
    "s_sub_u32 s85, s85, 63\n" \
    "s_sub_u32 s86, s86, 63\n" \
    "v_readlane_b32 s68, v46, 63\n" \
    "v_readlane_b32 s72, v47, 63\n" \
    "v_readlane_b32 s98, v48, 63\n" \
    "s_nop 3\n" \
    "v_writelane_b32 v46, s68, 0\n" \
    "v_writelane_b32 v47, s72, 0\n" \
    "v_writelane_b32 v48, s98, 0\n" \
    U2_INSTALL \
    "s_mov_b32 s69, 0\n" \
    "s_min_u32 s70, s86, 63\n" \
    "5:\n" \
    "s_cmp_ge_u32 s69, s70\n"                          /* no fast step left (nmax <= 2 next) */ \
    "s_cbranch_scc1 9f\n" \
    "s_cmp_eq_u32 s99, 0\n" \
    "s_cbranch_scc1 1b\n" \
    "9:\n" \
    "s_mov_b32 s72, s69\n" \
    FLUSH \
    "s_waitcnt vmcnt(0)\n"

// The index pop of a step renormalises when head_hi >= nmax * floor(2^31 / nmax) (push) or head < 2^31 (refill)
// (codec.cpp:27-35).  nmax * floor(2^31 / nmax) > 2^31 - nmax >= 2^31 - 2^18, so "head >= 2^31 and head_hi < 2^31 - 2^19"
// is a sufficient test for the fast path that needs no per-divisor constant (the loops test B = head - c(x), c(x) < 2^31: B_hi
// is at most one below head_hi); the few extra heads it rejects take the generic step.  Heads in [2^31, 2^32) are common (3 % of
// the steps of a 20-bit list: every fourth renormalisation of the second slice leaves one) and stay on the fast path.
#define U2_SAFE_HI 0x7ff80000u
__device__ __forceinline__ bool u2_needs_generic(uint64_t head) {
    return (head >> 31) == 0ull || (uint32_t)(head >> 32) >= U2_SAFE_HI;  // (round 5: the form the loops test with one 64-bit shift)
}
template <int UB, bool WANT_ORDER>
__global__ void __launch_bounds__(64) k_roc_encode_u2(RocEncArgs a, const U2Div *__restrict__ dtab) {
    using U = U2Geom<UB>;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *bm = (uint64_t *)smem;
    uint32_t *bm32 = (uint32_t *)smem;
    const uint32_t lane = lane_id();
    const uint32_t wi = blockIdx.x;
    if (wi >= a.nwork) return;
#ifdef U2_PROF2
    const uint64_t tk0 = __builtin_readcyclecounter();
#endif
    const uint32_t l = rfl(a.worklist[wi]);
    const uint64_t off = rfl64(a.offsets[l]);
    const uint32_t n = rfl((uint32_t)(a.offsets[l + 1] - off));
    {
        uint4 *z = (uint4 *)smem;
        for (uint32_t w = lane; w < U::LDS_BYTES / 16u; w += 64) z[w] = make_uint4(0, 0, 0, 0);
    }
    wave_sync();
    // The host classified this list by its LAST id (light prepass): the loop checks the rest -- every id inside
    // [0, 2^31) and, because the sampled ids only turn into input positions for an ascending list, the order when the
    // permutation is wanted.  An id beyond the bitmap is only counted in the maximum (the precision test below sends
    // the list to the general kernels).
    bool dup = false, bad = false;
    uint32_t mx = 0;
    uint64_t prev_last = 0;  // id at j0 - 1
    for (uint32_t j0 = 0; j0 < n; j0 += 512u) {  // 8 loads per lane in flight (one per iteration made this loop 1.2 ms of S1)
        uint64_t v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t j = j0 + u * 64u + lane;
            v[u] = j < n ? __builtin_nontemporal_load(&a.ids[off + j]) : ~0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            if (WANT_ORDER) {
                uint64_t below = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(v[u] >> 32), 1, 64) << 32) |
                                 (uint32_t)__shfl_up((int)(uint32_t)v[u], 1, 64);
                if (lane == 0) below = prev_last;
                prev_last = rfl64(((uint64_t)rl((uint32_t)(v[u] >> 32), 63) << 32) | rl((uint32_t)v[u], 63));
                if (v[u] != ~0ull && (j0 + u * 64u + lane) != 0u && below >= v[u]) dup = true;
            }
            if (v[u] == ~0ull) continue;
            bad |= (v[u] >> 31) != 0;
            const uint32_t x = (uint32_t)v[u];
            mx = x > mx ? x : mx;
            if (x >> UB) continue;
            const uint32_t bit = 1u << (x & 31u);
            const uint32_t old = atomicOr(&bm32[u2_word_of<UB>(x) * 2u + ((x >> 5) & 1u)], bit);
            dup |= (old & bit) != 0;
        }
    }
    wave_sync();
    if (ballot(bad)) {
        if (lane == 0) a.status[l] = VIDC_ST_DOMAIN;
        return;
    }
    const uint32_t P = rfl(a.prec[l]);  // written by the prepass
    const uint32_t maxid = wave_max_u32(mx);
    // multisets, precisions above 20 bits and ids that do not fit the precision (reference carry quirk) take the
    // general kernels
    if (ballot(dup) || P > 20u || (maxid >> P) != 0u || (maxid >> UB) != 0u) {
        if (lane == 0) a.status[l] = VIDC_ST_PENDING_SORT;
        return;
    }
    uint32_t E1;
    v32u ra, rb;
    u2_build_counts<UB>(bm, E1, ra, rb);

    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? P - 16u : 0u;
    WStack st;
    {
        const uint64_t ao = rfl64(arena_at(a, l));
        ws_init_empty(st, a.arena + ao, rfl((uint32_t)(arena_at(a, l + 1) - ao)), a.mt, VIDC_MT_TABLE);
    }
    uint64_t head = VIDC_RANS_L;
    uint32_t obuf = 0, obase = 0;
    uint32_t *order = WANT_ORDER ? a.perm + off : nullptr;  // sampled ids; k_perm_from_order turns them into positions
    const uint32_t T0 = 0x80000000u >> p0, T1 = 0x80000000u >> p1, MULN = 1u << p1;
    const uint32_t l3off = U::G == 4u ? (lane & 3u) * 8u : 0u;

#ifdef U2_PROF
    uint32_t prof[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
#ifdef U2_PROF2
    uint64_t prof2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    prof2[2] = __builtin_readcyclecounter();
#endif
    uint32_t nmax = n;  // divisor of the next step
    while (nmax) {
        if (st.sp - st.lo >= U2_RING_ROOM) ws_spill32(st);
        // generic step: the last two (the fix-up reciprocal needs nmax' >= 2) and index pops that renormalise
        if (nmax < 3u || u2_needs_generic(head)) {
            const uint32_t x = u2_slow_step<UB>(head, st, nmax, E1, ra, rb, bm, p0, p1);
            if (WANT_ORDER) {
                if (lane == 0) order[obase] = x;
                obase++;
            }
            nmax--;
            continue;
        }
        uint32_t s_x = 0, s_mul = 0, s_t = 0, s_waddr = U::BITMAP_BYTES, s_D0 = rfl(nmax), s_left = rfl(nmax - 2u);
        uint64_t s_w = 0, s_B = rfl64(head);
        uint64_t z0 = 0, z1 = 0, z2 = 0;  // {value, 0} register pairs: the odd halves stay 0
        uint64_t qh = 0;
        uint32_t rr = 0;
        st.sp = rfl(st.sp);
        st.lo = rfl(st.lo);
        st.err = rfl(st.err);
        obase = rfl(obase);
#ifdef U2_PROF2
        const uint64_t tp0 = __builtin_readcyclecounter();
#endif
        // clang-format off
#define U2_ENC_ASM(BODY)                                                                                                  \
        asm volatile(BODY                                                                                                 \
            : "+{v4}"(E1), "+{v5}"(st.win), "+{v6}"(obuf), "+{v[64:95]}"(ra), "+{v[96:127]}"(rb),                          \
              "+{v[46:47]}"(qh), "+{v48}"(rr), "+{s40}"(s_x), "+{s41}"(s_mul), "+{s[58:59]}"(s_B), "+{s60}"(st.sp),          \
              "+{s61}"(st.lo), "+{s65}"(s_waddr), "+{s[66:67]}"(s_w), "+{s69}"(s_t), "+{s85}"(s_D0), "+{s86}"(s_left),       \
              "+{s87}"(obase), "+{s93}"(st.err), "+{v[20:21]}"(z0), "+{v[36:37]}"(z1), "+{v[40:41]}"(z2) U2P_OPERANDS       \
            : "{v2}"(lane), "{v3}"(l3off), "{s73}"(T0), "{s74}"(T1), "{s75}"(MULN), "{s76}"(p0), "{s77}"(p1),                \
              "{s[88:89]}"(order), "{s[90:91]}"(st.mem), "{s92}"(st.cap), "{s[94:95]}"(dtab)                                \
            : "memory", "vcc", "scc", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",  \
              "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38", "v39",\
              "v42", "v43", "v44", "v45", "v49", "v50", "v51", "v52", "v54", "v55", "v56", "v57", "v58", "v59",                     \
              "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57",\
              "s62", "s63", "s64", "s68", "s70", "s71", "s72", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s96",       \
              "s97", "s98", "s99")
        if (U::G == 4u) {
            // four steps per loop iteration: the taken branch at the end of a step costs ~6 cycles of instruction-buffer refill
#define U2_STEP_X(ORDER, LSHR) U2_ENC_TOP_NL U2_ENC_MID_G4 U2_ENC_SLICE1_G4 U2_ENC_L3_G4 U2_ENC_BOT_T(ORDER, LSHR, "", "s_cbranch_scc0 7f\n")
            if (WANT_ORDER) U2_ENC_ASM(U2_ENC_ENTRY "1:\n" U2_STEP_X(U2_ENC_ORDER, "") U2_STEP_X(U2_ENC_ORDER, "") U2_STEP_X(U2_ENC_ORDER, "") U2_STEP_X(U2_ENC_ORDER, "") U2_STEP_X(U2_ENC_ORDER, "") U2_STEP_X(U2_ENC_ORDER, "") U2_STEP_X(U2_ENC_ORDER, "")
                                       U2_ENC_TOP_NL U2_ENC_MID_G4 U2_ENC_SLICE1_G4 U2_ENC_L3_G4 U2_ENC_BOT(U2_ENC_ORDER, "") "7:\n" U2_ENC_OUTER(U2_ORDER_FLUSH));
            else U2_ENC_ASM(U2_ENC_ENTRY "1:\n" U2_STEP_X("", "s_lshr_b64 s[98:99], s[58:59], 31\n") U2_STEP_X("", "s_lshr_b64 s[98:99], s[58:59], 31\n") U2_STEP_X("", "s_lshr_b64 s[98:99], s[58:59], 31\n")
                            U2_ENC_TOP_NL U2_ENC_MID_G4 U2_ENC_SLICE1_G4 U2_ENC_L3_G4 U2_ENC_BOT("", "s_lshr_b64 s[98:99], s[58:59], 31\n") "7:\n" U2_ENC_OUTER(""));
#undef U2_STEP_X
        } else {
            // (round 4: the 18-bit bodies in eight / four copies with the ring test in the last one, like the 20-bit ones)
#define U2_STEP_X1(ORDER, LSHR) U2_ENC_TOP_NL U2_ENC_MID_G1 U2_ENC_SLICE1("6") U2_ENC_L3_G1 U2_ENC_BOT_T(ORDER, LSHR, "", "s_cbranch_scc0 7f\n")
            if (WANT_ORDER) U2_ENC_ASM(U2_ENC_ENTRY "1:\n" U2_STEP_X1(U2_ENC_ORDER, "") U2_STEP_X1(U2_ENC_ORDER, "") U2_STEP_X1(U2_ENC_ORDER, "") U2_STEP_X1(U2_ENC_ORDER, "") U2_STEP_X1(U2_ENC_ORDER, "") U2_STEP_X1(U2_ENC_ORDER, "") U2_STEP_X1(U2_ENC_ORDER, "")
                                       U2_ENC_TOP_NL U2_ENC_MID_G1 U2_ENC_SLICE1("6") U2_ENC_L3_G1 U2_ENC_BOT(U2_ENC_ORDER, "") "7:\n" U2_ENC_OUTER(U2_ORDER_FLUSH));
            else U2_ENC_ASM(U2_ENC_ENTRY "1:\n" U2_STEP_X1("", "s_lshr_b64 s[98:99], s[58:59], 31\n") U2_STEP_X1("", "s_lshr_b64 s[98:99], s[58:59], 31\n") U2_STEP_X1("", "s_lshr_b64 s[98:99], s[58:59], 31\n")
                            U2_ENC_TOP_NL U2_ENC_MID_G1 U2_ENC_SLICE1("6") U2_ENC_L3_G1 U2_ENC_BOT("", "s_lshr_b64 s[98:99], s[58:59], 31\n") "7:\n" U2_ENC_OUTER(""));
#undef U2_STEP_X1
        }
#undef U2_ENC_ASM
        // clang-format on
#ifdef U2_PROF2
        prof2[0] += __builtin_readcyclecounter() - tp0;
        prof2[1] += 1;
        prof2[4] += s_t == 63u;
        prof2[5] += st.sp - st.lo >= 62u;
        prof2[6] += u2_needs_generic(s_B);
        prof2[7] += s_t;
#endif
        // ---- back to the plain state.  head = B + c(x); x's bit leaves the bitmap; slices 2 and 3 of ID_push
        // (precision 0: "if head >= 2^63 push", codec.cpp:92-105) can only fire on the way out
        head = s_B + (uint64_t)((s_x & 0xffffu) * s_mul + (s_x >> 16));
        bm[s_waddr >> 3] = s_w & ~(1ull << (s_x & 63u));
        ws_window(st);
        if (__builtin_expect((head >> 63) != 0ull, 0)) {
            ws_prepare(st);
            ans_u_push(head, st, 0u, 0u);
            ans_u_push(head, st, 0u, 0u);
        }
        nmax = s_D0 - s_t;
    }
#ifdef U2_PROF2
    prof2[3] = __builtin_readcyclecounter() - prof2[2];
    prof2[2] -= tk0;
    if (lane == 0)
        for (int q = 0; q < 8; q++) ((uint64_t *)a.sid)[q] = prof2[q];
#endif
    ws_flush(st);
#ifdef U2_PROF
    if (lane == 0)
        for (int q = 0; q < 14; q++) a.sid[q] = prof[q];
#endif
    if (lane == 0) {
        a.heads[l] = head;
        a.nwords[l] = st.sp;
        a.draws[l] = st.draws;
        a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
    }
}


// ===============================================================================================================
// The same encode loop for ids of ANY precision (P <= 31): k_roc_encode_r2.  The alive set is a bitmap over the
// POSITIONS 0 .. n-1 of the (ascending) list instead of over the id universe -- n <= 262 144, so the 18-bit geometry
// (32 KiB of LDS) always fits -- and the id of the selected position comes from the input array with one scalar load
// per step (s_load_dword: the address is wave-uniform; the input is never written, so the scalar cache is coherent).
// Everything else is the loop above: head' = B(q) + c(id), the division of B in the select's windows, reversed
// counters, branch-free renormalisation.  Differences: the fix-up remainder is a full 32-bit multiply (q^ no longer
// fits 24 bits: c(id) < 2^31), the order ring holds positions (= the sampling permutation itself, no
// k_perm_from_order pass), and the load sits on the chain: the step's tail (order ring, row write-back, exit tests)
// runs in its shadow.  The kernel verifies what it assumes (ascending, ids < 2^31) and computes the precision from
// the list's own maximum like the general kernel; anything else goes back as VIDC_ST_PENDING_SORT.
//   additional registers: s[36:37] input ids of the list (u64 each)   s39 id of the previously selected position
#define U2_ENC_TOP_R "1:\n" U2_ENC_TOP_R_NL
#define U2_ENC_TOP_R_NL \
    "v_lshrrev_b32_e64 v14, 16, s39\n"                 /* id_hi */ \
    "v_bfe_u32 v15, s39, 0, 16\n"                      /* id_lo */ \
    "v_add_u32 v16, v48, v14\n"                        /* s = r_B + id_hi */ \
    "v_mad_u32_u24 v16, v15, s41, v16\n"               /*   + id_lo * mul   (< 2^32) */ \
    "v_mul_hi_u32 v17, v16, v10\n"                     /* q^ = mulhi(s, 2^32 / d) in {Q - 1, Q} */ \
    "v_mul_lo_u32 v18, v17, v9\n" \
    "v_sub_u32 v18, v16, v18\n"                        /* r = s - q^ d in [0, 2d) */ \
    "v_sub_u32 v19, v18, v9\n" \
    "v_min_u32 v52, v18, v19\n"                        /* k (of this lane's divisor) */ \
    "v_ashrrev_i32 v27, 31, v19\n"                     /* -1 if r < d */ \
    "v_add3_u32 v20, v17, v27, 1\n"                    /* qf = q^ + (r >= d) */ \
    "v_readlane_b32 s42, v52, s69\n"                   /* k of the step */ \
    "v_add_co_u32_e64 v50, s[78:79], v46, v20\n"       /* q = head div d (per lane) */ \
    "v_addc_co_u32_e64 v51, s[78:79], 0, v47, s[78:79]\n" \
    "v_mov_b32 v12, s42\n" \
    "v_cmp_le_u32 vcc, v4, v12\n"                      /* level 1 */ \
    "v_lshlrev_b64 v[22:23], s40, 1\n"                 /* removal of position x_{i-1} from the bitmap */ \
    "v_bfi_b32 v24, v22, 0, s66\n" \
    "v_readlane_b32 s50, v50, s69\n" \
    "v_readlane_b32 s51, v51, s69\n" \
    "v_bfi_b32 v25, v23, 0, s67\n" \
    "v_mov_b32 v26, s65\n" \
    "ds_write_b64 v26, v[24:25]\n" \
    "s_ff1_i32_b64 s43, vcc\n" \
    "s_set_gpr_idx_on s43, gpr_idx(SRC0)\n" \
    "v_mov_b32 v13, v64\n"                             /* row L1 */ \
    "s_set_gpr_idx_off\n" \
    "s_and_b32 m0, s60, 63\n"                          /* slice 0 (codec.cpp:65-76), branch-free */ \
    "s_cmp_ge_u32 s51, s73\n" \
    "v_writelane_b32 v5, s50, m0\n" \
    "s_cselect_b32 s52, s51, s50\n" \
    "s_cselect_b32 s53, 0, s51\n" \
    "s_addc_u32 s60, s60, 0\n" \
    "s_lshl_b64 s[54:55], s[52:53], s76\n" \
    "v_readlane_b32 s49, v4, s43\n" \
    "v_subrev_u32 v27, s43, v2\n"                      /* lane - L1 */ \
    "v_subrev_u32 v12, s49, v12\n" \
    "v_cmp_le_u32 vcc, v13, v12\n"                     /* level 2 */ \
    "v_ashrrev_i32 v27, 31, v27\n" \
    "v_add_u32 v4, v4, v27\n"                          /* E1 -= 1 in lanes below L1 */ \
    "s_ff1_i32_b64 s44, vcc\n" \
    "s_lshl_b32 s45, s43, 6\n" \
    "s_or_b32 s45, s45, s44\n"
// the tail of the step: position x = s40, its id by a scalar load (s39) consumed by the slice word and the next step
#define U2_ENC_BOT_R(ORDER, LSHR) U2_ENC_BOT_R_T(ORDER, LSHR, "s_cbranch_scc1 1b\n")
#define U2_ENC_BOT_R_T(ORDER, LSHR, TAIL) \
    "v_mbcnt_lo_u32_b32 v49, s66, 0\n" \
    "v_mbcnt_hi_u32_b32 v49, s67, v49\n" \
    "v_cmp_eq_u32 vcc, v49, v12\n"                     /* bit of the word */ \
    "v_sub_u32 v48, s58, v48\n"                        /* r_B in [0, 2d) */ \
    "s_and_b64 s[82:83], vcc, s[66:67]\n" \
    "s_ff1_i32_b64 s48, s[82:83]\n" \
    "s_or_b32 s40, s46, s48\n"                         /* position x */ \
    "s_lshl_b32 s72, s40, 3\n" \
    "s_load_dword s39, s[36:37], s72\n"                /* its id: low dword of ids[x] */ \
    ORDER \
    "s_set_gpr_idx_on s43, gpr_idx(DST)\n" \
    "v_mov_b32 v64, v13\n" \
    "s_set_gpr_idx_off\n" \
    LSHR                                               /* next index pop must not renormalise (u2_needs_generic): SCC = B >= 2^31 */ \
    "s_cselect_b32 s71, s70, 0\n" \
    "s_cmp_ge_u32 s59, 0x7ff80000\n" \
    "s_cselect_b32 s71, 0, s71\n" \
    "s_sub_u32 s68, s60, s61\n"                        /* ring nearly full */ \
    "s_cmp_ge_u32 s68, 62\n" \
    "s_cselect_b32 s71, 0, s71\n" \
    "s_add_u32 s69, s69, 1\n" \
    "s_mov_b32 m0, s62\n" \
    "s_waitcnt lgkmcnt(0)\n"                           /* the id (and the removal store of the step's top) */ \
    "s_and_b32 s68, s39, 0xffff\n" \
    "s_or_b32 s68, s68, s54\n" \
    "v_writelane_b32 v5, s68, m0\n"                    /* word of the second slice (kept only if it renormalised) */ \
    "s_cmp_lt_u32 s69, s71\n" \
    TAIL

// one generic encode step in position space (rare path): codec.cpp:131-137; returns the position
template <uint32_t NE>
__device__ __forceinline__ uint32_t u2r_slow_step(uint64_t &head, WStack &st, uint32_t nmax, uint32_t &E1, v32u &ra, v32u &rb,
                                                  uint64_t *bm, const uint64_t *ids, uint32_t p0, uint32_t p1, uint32_t &idv) {
    const uint32_t lane = lane_id();
    ws_prepare(st);
    const uint32_t lq = 0x80000000u / nmax;
    uint32_t k = ans_idx_pop(head, st, nmax, lq * nmax, ~0ull / (uint64_t)nmax);
    const uint32_t L1 = ff1(ballot(E1 <= k));
    k -= rl(E1, L1);
    uint32_t row = u2_row_get(ra, rb, L1);
    const uint32_t L2 = ff1(ballot(row <= k));
    k -= rl(row, L2);
    const uint32_t e = L1 * 64u + L2;
    const uint64_t W = rfl64(bm[e]);
    const uint32_t b = ff1(ballot(mbcnt(W) == k) & W);
    const uint32_t x = ((e ^ (NE - 1u)) << 6) | b;
    E1 -= lane < L1 ? 1u : 0u;
    row -= lane < L2 ? 1u : 0u;
    u2_row_set(ra, rb, L1, row);
    bm[e] = W & ~(1ull << b);
    wave_sync();
    idv = rfl((uint32_t)ids[x]);
    ans_id_push(head, st, idv, p0, p1);
    return x;
}

// NEB: log2 of the bitmap's 64-position entries -- 12: 262 144 positions, 32 KiB of LDS; 10: 65 536 positions, 8 KiB (lists up
// to 65 536 ids: sixteen blocks of 4096 positions in lanes 0..15 of the level-1 counters, rows 0..15)
#define VIDC_R2_LDS_BYTES(NEB) ((8u << (NEB)) + 16u)
template <bool WANT_ORDER, int NEB = 12>
__device__ __forceinline__ void roc_encode_r2_body(RocEncArgs a, const U2Div *__restrict__ dtab) {
    static_assert(NEB == 12 || NEB == 10, "bitmap entries");
    constexpr uint32_t NE = 1u << NEB, NBLK = NE / 64u;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *bm = (uint64_t *)smem;
    const uint32_t lane = lane_id();
    if (a.lpw >> 31) __builtin_amdgcn_s_setprio(3);  // (VIDC_ENC_PRIO, roc.hip)
    const uint32_t wi = blockIdx.x;
    if (wi >= a.nwork) return;
    const uint32_t l = rfl(a.worklist[wi]);
    const uint64_t off = rfl64(a.offsets[l]);
    const uint32_t n = rfl((uint32_t)(a.offsets[l + 1] - off));
    const uint64_t *ids = a.ids + off;
    // ---- the list: ascending, inside [0, 2^31); its maximum gives the precision (k_roc_encode_gen phase 0)
    uint32_t mx = 0;
    bool bad = false, unsorted = false;
    for (uint32_t j0 = 0; j0 < n; j0 += 512u) {
        uint64_t v[8], w[8];
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t j = j0 + u * 64u + lane;
            v[u] = j < n ? ids[j] : ~0ull;
            w[u] = (j && j < n) ? ids[j - 1] : 0ull;
        }
#pragma unroll
        for (uint32_t u = 0; u < 8u; u++) {
            const uint32_t j = j0 + u * 64u + lane;
            if (j >= n) continue;
            bad |= (v[u] >> 31) != 0;
            if (j) unsorted |= w[u] >= v[u];
            mx = (uint32_t)v[u] > mx ? (uint32_t)v[u] : mx;
        }
    }
    if (ballot(bad)) {
        if (lane == 0) a.status[l] = VIDC_ST_DOMAIN;
        return;
    }
    const uint32_t maxid = wave_max_u32(mx);
    const uint32_t P = precision_for(maxid, a.precision_mode);
    // not ascending, or ids that do not fit the precision (explicit precision / the reference's power-of-two quirk): the
    // general kernel's second pass
    if (ballot(unsorted) || n > NE * 64u || P > 31u || (P < 32u && (maxid >> P) != 0u)) {
        if (lane == 0) a.status[l] = VIDC_ST_PENDING_SORT;
        return;
    }
    // ---- alive bitmap over positions (reversed entry order, one 64-bit word per entry; entries past the list are never
    // selected -- their counters are 0 -- and stay unwritten) + counters in closed form: every position below n is alive, so
    // the exclusive counts of u2_build_counts are min(.., ..) of n and the block / entry boundaries
    for (uint32_t e = lane; e < ((n + 63u) >> 6); e += 64u) {
        const uint32_t lo = e << 6;
        const uint32_t c = n - lo >= 64u ? 64u : n - lo;
        bm[e ^ (NE - 1u)] = c == 64u ? ~0ull : ((1ull << c) - 1ull);
    }
    if (lane == 0) { bm[NE] = 0; bm[NE + 1] = 0; }
    wave_sync();
    // lane L of E1 <-> block L: ids in blocks with a larger index (= smaller positions): 4096 positions per block
    // (lanes from NBLK on own no block: their count 0 is never the FIRST one <= k)
    uint32_t E1 = lane < NBLK ? 4096u * (NBLK - 1u - lane) : 0u;
    E1 = E1 < n ? E1 : n;
    v32u ra, rb;
#pragma unroll
    for (int L1 = 0; L1 < 64; L1++) {
        // row L1, lane L2: alive positions of block L1 in entries with a larger lane index = the first 64 (63 - L2) positions
        // of the block, which starts at position 4096 (NBLK - 1 - L1)
        uint32_t v = 0;
        if ((uint32_t)L1 < NBLK) {
            const uint32_t b0 = 4096u * (NBLK - 1u - (uint32_t)L1);
            const uint32_t in_block = n > b0 ? n - b0 : 0u, want = 64u * (63u - lane);
            v = in_block < want ? in_block : want;
        }
        if (L1 < 32) ra[L1] = v; else rb[L1 - 32] = v;
    }

    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? P - 16u : 0u;
    WStack st;
    {
        const uint64_t ao = rfl64(arena_at(a, l));
        ws_init_empty(st, a.arena + ao, rfl((uint32_t)(arena_at(a, l + 1) - ao)), a.mt, VIDC_MT_TABLE);
    }
    uint64_t head = VIDC_RANS_L;
    uint32_t obuf = 0, obase = 0;
    uint32_t *order = WANT_ORDER ? a.perm + off : nullptr;  // sampled POSITIONS: the permutation itself
    const uint32_t T0 = 0x80000000u >> p0, T1 = 0x80000000u >> p1, MULN = 1u << p1;
    const uint32_t l3off = 0u;

    uint32_t nmax = n;  // divisor of the next step
    while (nmax) {
        if (st.sp - st.lo >= 62u) ws_spill32(st);
        if (nmax < 3u || u2_needs_generic(head)) {
            uint32_t idv;
            const uint32_t x = u2r_slow_step<NE>(head, st, nmax, E1, ra, rb, bm, ids, p0, p1, idv);
            if (WANT_ORDER) {
                if (lane == 0) order[obase] = x;
                obase++;
            }
            nmax--;
            continue;
        }
        uint32_t s_x = 0, s_id = 0, s_mul = 0, s_t = 0, s_waddr = NE * 8u, s_D0 = rfl(nmax), s_left = rfl(nmax - 2u);
        uint64_t s_w = 0, s_B = rfl64(head);
        uint64_t z0 = 0, z1 = 0, z2 = 0;  // {value, 0} register pairs: the odd halves stay 0
        uint64_t qh = 0;
        uint32_t rr = 0;
        st.sp = rfl(st.sp);
        st.lo = rfl(st.lo);
        st.err = rfl(st.err);
        obase = rfl(obase);
        const uint64_t idbase = rfl64((uint64_t)ids);
        // clang-format off
#define U2R_ENC_ASM(BODY)                                                                                                 \
        asm volatile(BODY                                                                                                 \
            : "+{v4}"(E1), "+{v5}"(st.win), "+{v6}"(obuf), "+{v[64:95]}"(ra), "+{v[96:127]}"(rb),                          \
              "+{v[46:47]}"(qh), "+{v48}"(rr), "+{s40}"(s_x), "+{s39}"(s_id), "+{s41}"(s_mul), "+{s[58:59]}"(s_B),          \
              "+{s60}"(st.sp), "+{s61}"(st.lo), "+{s65}"(s_waddr), "+{s[66:67]}"(s_w), "+{s69}"(s_t), "+{s85}"(s_D0),        \
              "+{s86}"(s_left), "+{s87}"(obase), "+{s93}"(st.err), "+{v[20:21]}"(z0), "+{v[36:37]}"(z1), "+{v[40:41]}"(z2) \
            : "{v2}"(lane), "{v3}"(l3off), "{s73}"(T0), "{s74}"(T1), "{s75}"(MULN), "{s76}"(p0), "{s77}"(p1),                \
              "{s[88:89]}"(order), "{s[90:91]}"(st.mem), "{s92}"(st.cap), "{s[94:95]}"(dtab), "{s[36:37]}"(idbase)          \
            : "memory", "vcc", "scc", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",  \
              "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v38", "v39",\
              "v42", "v43", "v44", "v45", "v49", "v50", "v51", "v52", "v54", "v55", "v56", "v57", "v58", "v59",                     \
              "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55", "s56", "s57",\
              "s62", "s63", "s64", "s68", "s70", "s71", "s72", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s96",       \
              "s97", "s98", "s99")
        // (two steps per loop iteration, as in k_roc_encode_u2)
#define U2R_BODY(XORC)                                                                                                                      \
        if (WANT_ORDER) U2R_ENC_ASM(U2_ENC_ENTRY U2_ENC_TOP_R U2_ENC_MID_G1 U2_ENC_SLICE1_X("6", XORC) U2_ENC_L3_G1 U2_ENC_BOT_R_T(U2_ENC_ORDER, "", "s_cbranch_scc0 7f\n") \
                                    U2_ENC_TOP_R_NL U2_ENC_MID_G1 U2_ENC_SLICE1_X("6", XORC) U2_ENC_L3_G1 U2_ENC_BOT_R(U2_ENC_ORDER, "") "7:\n" U2_ENC_OUTER(U2_ORDER_FLUSH)); \
        else U2R_ENC_ASM(U2_ENC_ENTRY U2_ENC_TOP_R U2_ENC_MID_G1 U2_ENC_SLICE1_X("6", XORC) U2_ENC_L3_G1 U2_ENC_BOT_R_T("", "s_lshr_b64 s[98:99], s[58:59], 31\n", "s_cbranch_scc0 7f\n") \
                         U2_ENC_TOP_R_NL U2_ENC_MID_G1 U2_ENC_SLICE1_X("6", XORC) U2_ENC_L3_G1 U2_ENC_BOT_R("", "s_lshr_b64 s[98:99], s[58:59], 31\n") "7:\n" U2_ENC_OUTER(""))
        if (NEB == 12) { U2R_BODY("0xfff"); } else { U2R_BODY("0x3ff"); }
#undef U2R_BODY
#undef U2R_ENC_ASM
        // clang-format on
        // ---- back to the plain state.  head = B + c(id); the position's bit leaves the bitmap
        head = s_B + (uint64_t)((s_id & 0xffffu) * s_mul + (s_id >> 16));
        bm[s_waddr >> 3] = s_w & ~(1ull << (s_x & 63u));
        ws_window(st);
        if (__builtin_expect((head >> 63) != 0ull, 0)) {
            ws_prepare(st);
            ans_u_push(head, st, 0u, 0u);
            ans_u_push(head, st, 0u, 0u);
        }
        nmax = s_D0 - s_t;
    }
    ws_flush(st);
    if (lane == 0) {
        a.heads[l] = head;
        a.prec[l] = P;
        a.nwords[l] = st.sp;
        a.draws[l] = st.draws;
        a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
    }
}
template <bool WANT_ORDER, int NEB = 12>
__global__ void __launch_bounds__(64) k_roc_encode_r2(RocEncArgs a, const U2Div *__restrict__ dtab) {
    roc_encode_r2_body<WANT_ORDER, NEB>(a, dtab);
}

// ===============================================================================================================
// Decode (codec.cpp:140-152): step i pops x (two 16-bit slices, high first), ranks it among the ids decoded so far,
// pushes the rank as a uniform index over nmax = i + 1 and stores out[n - 1 - i] = x.
//
// head' = H * nmax + r with H (the head after the two slice pops, renormalised for the index push) independent of the
// rank r: the push renormalisation, the 64 x 32 multiply and all insertions run in the shadow of the LDS read; r joins with
// one 64-bit add, and the next id is a bit field of that sum.  The set is a bitmap over the id universe with the same
// reversed exclusive counters as the encoder (rank = E1[L1] + row[L1][L2] + set bits below x inside its entry).
// Multiset streams (reference quirk regimes) cannot be represented by a bitmap: the number of set bits is compared with
// n at the end and such a list is decoded again by the round-1 kernel body, which keeps a side list of duplicates.
//
// Register map (decode):
//   v2 lane  v3 (lane & 3) * 8  v4 E1  v5 stack ring  v6:7 {output ring, 0}  v10 lq = floor(2^31 / nmax) per lane (nmax = N0 + lane)
//   v13 row  v26 address / v27 bit of the insertion  v28 LDS address  v30:31 word(s)  v32..v37 popcount temporaries
//   v54:57 table entry of the next block  v58 v59 tmp  v64..v127 level-2 rows
//   s40 x  s43 L1  s44 L2  s45 entry  s46 x_hi  s47 g  s48 b  s49 E1[L1]  s50:51 H  s52:53 H * nmax  s58:59 head  s60 sp
//   s61 lo  s62 lq  s63 row[L2]  s64 below  s65 tmp  s66:67 mask below b  s68 tmp  s69 t  s70 t_end  s71 limit  s72 tmp
//   s73 2^p1 - 1  s74 2^p0 - 1  s75 nmax  s76 p0  s77 p1  s85 N0  s86 steps left at lane 0  s87 output index of lane 0
//   s88:89 output pointer  s94:95 divisor table  s96:97 saved exec  s98 s99 tmp
#define U2_DEC_INSTALL \
    "v_mov_b32 v10, v57\n" \
    "s_add_u32 s68, s85, 64\n" \
    "v_add_u32 v58, s68, v2\n" \
    "v_min_u32 v58, 0x40000, v58\n" \
    "v_lshlrev_b32 v58, 4, v58\n" \
    "global_load_dwordx4 v[54:57], v58, s[94:95]\n"
#define U2_DEC_ENTRY \
    "v_add_u32 v58, s85, v2\n" \
    "v_min_u32 v58, 0x40000, v58\n" \
    "v_lshlrev_b32 v58, 4, v58\n" \
    "global_load_dwordx4 v[54:57], v58, s[94:95]\n" \
    "s_waitcnt vmcnt(0)\n" \
    U2_DEC_INSTALL \
    "s_mov_b32 s69, 0\n" \
    "s_mov_b32 s75, s85\n" \
    "s_min_u32 s70, s86, 64\n"

#define U2_DEC_TOP "1:\n" U2_DEC_TOP_L("20", "21", "22", "23")
#define U2_DEC_TOP_L(LA, LB, LC, LD)                   /* (labels of the two refill side paths and their returns) */ \
    "s_and_b32 s46, s58, s73\n"                        /* slice 1: x_hi (codec.cpp:78-90) */ \
    "s_lshr_b64 s[58:59], s[58:59], s77\n" \
    "s_lshr_b64 s[98:99], s[58:59], 31\n"             /* (SCC = head >= 2^31: one instruction instead of shift + or, round 5) */ \
    "s_cbranch_scc0 " LA "f\n"                         /* head < 2^31: refill */ \
    LB ":\n" \
    "s_and_b32 s40, s58, s74\n"                        /* slice 0: x_lo */ \
    "s_lshr_b64 s[58:59], s[58:59], s76\n" \
    "s_lshr_b64 s[98:99], s[58:59], 31\n" \
    "s_cbranch_scc0 " LC "f\n" \
    LD ":\n" \
    "s_pack_ll_b32_b16 s40, s40, s46\n"                /* x = x_hi : x_lo (one instruction, round 5) */
#define U2_DEC_IDX_G4 \
    "s_lshr_b32 s45, s40, 8\n" \
    "s_xor_b32 s45, s45, 0xfff\n"                      /* entry (reversed) */ \
    "v_lshl_add_u32 v28, s45, 5, v3\n" \
    "ds_read_b64 v[30:31], v28\n" \
    "s_bfe_u32 s47, s40, 0x20006\n"                    /* g */ \
    "s_bfe_u32 s65, s40, 0x30005\n"                    /* 32-bit half of the entry holding x */ \
    "s_lshl3_add_u32 s65, s45, s65\n"                  /* (dword of the bitmap; U2_DEC_MID shifts it to a byte address) */
#define U2_DEC_IDX_G1 \
    "s_lshr_b32 s45, s40, 6\n" \
    "s_xor_b32 s45, s45, 0xfff\n" \
    "v_lshlrev_b32_e64 v28, 3, s45\n" \
    "ds_read_b64 v[30:31], v28\n" \
    "s_bfe_u32 s65, s40, 0x10005\n" \
    "s_lshl1_add_u32 s65, s45, s65\n"
#define U2_DEC_MID \
    "s_lshr_b32 s43, s45, 6\n"                         /* L1 */ \
    "s_and_b32 s44, s45, 63\n"                         /* L2 */ \
    "s_set_gpr_idx_on s43, gpr_idx(SRC0)\n" \
    "v_mov_b32 v13, v64\n" \
    "s_set_gpr_idx_off\n" \
    "s_bfm_b64 s[66:67], s40, 0\n"                     /* bits below b = x mod 64 (the instruction reads six bits of its width operand) */ \
    "v_lshlrev_b32_e64 v26, 2, s65\n"                  /* insertion of x: after the read in LDS order */ \
    "v_lshlrev_b32_e64 v27, s40, 1\n" \
    "s_mov_b64 exec, 1\n" \
    "ds_or_b32 v26, v27\n" \
    "s_mov_b64 exec, -1\n" \
    "v_readlane_b32 s49, v4, s43\n"                    /* E1[L1] */ \
    "v_readlane_b32 s63, v13, s44\n"                   /* row[L2] */ \
    "v_readlane_b32 s62, v10, s69\n"                   /* lq of this step */ \
    "v_subrev_u32 v58, s43, v2\n" \
    "v_subrev_u32 v59, s44, v2\n" \
    "v_ashrrev_i32 v59, 31, v59\n" \
    "v_sub_u32 v13, v13, v59\n"                        /* row += 1 in lanes below L2 */ \
    "s_and_b32 m0, s60, 63\n"                          /* index push, first half (codec.cpp:44-63): renormalise H */ \
    "s_cmp_ge_u32 s59, s62\n" \
    "v_writelane_b32 v5, s58, m0\n" \
    "s_cselect_b32 s50, s59, s58\n" \
    "s_cselect_b32 s51, 0, s59\n" \
    "s_addc_u32 s60, s60, 0\n" \
    "s_mul_i32 s52, s50, s75\n"                        /* H * nmax */ \
    "s_mul_hi_u32 s53, s50, s75\n" \
    "s_mul_i32 s68, s51, s75\n" \
    "s_add_u32 s53, s53, s68\n" \
    "s_add_u32 s72, s49, s63\n" \
    "s_set_gpr_idx_on s43, gpr_idx(DST)\n" \
    "v_mov_b32 v64, v13\n" \
    "s_set_gpr_idx_off\n" \
    "s_mov_b32 m0, s69\n"
// (the three vector instructions that follow the last readlane of the rank keep the scalar tail off its SGPR write)
#define U2_DEC_AFTER_RANK \
    "v_writelane_b32 v6, s40, m0\n"                    /* output ring */ \
    "v_ashrrev_i32 v58, 31, v58\n" \
    "v_sub_u32 v4, v4, v58\n"                          /* E1 += 1 in lanes below L1 */
#define U2_DEC_RANK_G4 \
    "s_waitcnt lgkmcnt(1)\n" \
    "v_bcnt_u32_b32 v32, v30, 0\n" \
    "v_bcnt_u32_b32 v32, v31, v32\n"                   /* set bits of the lane's word */ \
    "v_and_b32 v33, s66, v30\n" \
    "v_and_b32 v34, s67, v31\n" \
    "v_add_u32_dpp v35, v32, v32 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_bcnt_u32_b32 v33, v33, 0\n" \
    "v_bcnt_u32_b32 v33, v34, v33\n"                   /* ... below b */ \
    "v_add_u32_dpp v36, v35, v35 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n" \
    "v_sub_u32 v33, v33, v32\n" \
    "v_add_u32 v37, v36, v33\n"                        /* lane g: bits of the entry below x */ \
    "s_nop 0\n" \
    "v_readlane_b32 s64, v37, s47\n" \
    U2_DEC_AFTER_RANK
#define U2_DEC_RANK_G1 \
    "s_waitcnt lgkmcnt(1)\n" \
    "v_and_b32 v33, s66, v30\n" \
    "v_and_b32 v34, s67, v31\n" \
    "v_bcnt_u32_b32 v33, v33, 0\n" \
    "v_bcnt_u32_b32 v37, v34, v33\n" \
    "s_nop 0\n" \
    "v_readfirstlane_b32 s64, v37\n" \
    U2_DEC_AFTER_RANK
#define U2_DEC_BOT U2_DEC_BOT_CORE "s_cbranch_scc1 1b\n s_branch 2f\n" U2_DEC_SIDE("20", "21", "22", "23") "2:\n"
#define U2_DEC_BOT_CORE U2_DEC_BOT_CORE_T("-2", "59")
#define U2_DEC_BOT_CORE_T(RLO, RSPAN)                  /* the next step may run while RLO <= words in the ring <= RLO + RSPAN */ \
    "s_sub_u32 s68, s60, s61\n"                        /* ring: two pops and one push must fit the next step */ \
    "s_add_u32 s68, s68, " RLO "\n" \
    "s_add_u32 s69, s69, 1\n" \
    "s_add_u32 s75, s75, 1\n" \
    "s_add_u32 s72, s72, s64\n"                        /* rank */ \
    "s_add_u32 s58, s52, s72\n"                        /* head' = H * nmax + rank */ \
    "s_addc_u32 s59, s53, 0\n" \
    "s_cmp_gt_u32 s68, " RSPAN "\n" \
    "s_cselect_b32 s71, 0, s70\n" \
    "s_lshr_b64 s[98:99], s[58:59], 31\n"             /* head' < 2^31: the push refills (generic code) */ \
    "s_cselect_b32 s71, s71, 0\n" \
    "s_cmp_lt_u32 s69, s71\n"
// the same without the ring test: copies of the loop body whose ring state the last copy (or the entry) has vouched for
#define U2_DEC_BOT_CORE_NR \
    "s_add_u32 s69, s69, 1\n" \
    "s_add_u32 s75, s75, 1\n" \
    "s_add_u32 s72, s72, s64\n"                        /* rank */ \
    "s_add_u32 s58, s52, s72\n"                        /* head' = H * nmax + rank */ \
    "s_addc_u32 s59, s53, 0\n" \
    "s_lshr_b64 s[98:99], s[58:59], 31\n"             /* head' < 2^31: the push refills (generic code) */ \
    "s_cselect_b32 s71, s70, 0\n" \
    "s_cmp_lt_u32 s69, s71\n"
// Ring of the eight-copy loop: a step pops at most two words and pushes at most one, so eight steps are safe while the ring
// holds U2_DEC_RING_LO .. U2_DEC_RING_HI words when the first copy starts (18 - 16 = 2 words left for the last pops, 53 + 8 = 61
// before the last push); only the last copy, U2_DEC_OUTER and the C++ glue test it.
#define U2_DEC_RING_LO 18u
#define U2_DEC_RING_HI 53u
#define U2_DEC_SIDE(LA, LB, LC, LD) \
    LA ":\n"                                           /* refills: head = (head << 32) | pop (codec.cpp:83-87) */ \
    "s_sub_u32 s60, s60, 1\n" \
    "s_and_b32 s68, s60, 63\n" \
    "s_mov_b32 s59, s58\n" \
    "v_readlane_b32 s58, v5, s68\n" \
    "s_branch " LB "b\n" \
    LC ":\n" \
    "s_sub_u32 s60, s60, 1\n" \
    "s_and_b32 s68, s60, 63\n" \
    "s_mov_b32 s59, s58\n" \
    "v_readlane_b32 s58, v5, s68\n" \
    "s_branch " LD "b\n"
// four steps per loop iteration for the bitmap kernel (the taken branch behind a step costs ~6 cycles of instruction-buffer
// refill: 0.195 -> 0.192 -> 0.191 us per step with two / four), two for the bucket kernels (their steps wait on memory)
#define U2_DEC_LOOP4(IDX, MID, RANK) \
    "1:\n" U2_DEC_TOP_L("20", "21", "22", "23") IDX MID RANK U2_DEC_BOT_CORE "s_cbranch_scc0 2f\n" \
    U2_DEC_TOP_L("30", "31", "32", "33") IDX MID RANK U2_DEC_BOT_CORE "s_cbranch_scc0 2f\n" \
    U2_DEC_TOP_L("40", "41", "42", "43") IDX MID RANK U2_DEC_BOT_CORE "s_cbranch_scc0 2f\n" \
    U2_DEC_TOP_L("50", "51", "52", "53") IDX MID RANK U2_DEC_BOT_CORE "s_cbranch_scc1 1b\n s_branch 2f\n" \
    U2_DEC_SIDE("20", "21", "22", "23") U2_DEC_SIDE("30", "31", "32", "33") U2_DEC_SIDE("40", "41", "42", "43") \
    U2_DEC_SIDE("50", "51", "52", "53") "2:\n"
#define U2_DEC_STEP_X(IDX, MID, RANK, LA, LB, LC, LD) U2_DEC_TOP_L(LA, LB, LC, LD) IDX MID RANK U2_DEC_BOT_CORE_NR "s_cbranch_scc0 2f\n"
#define U2_DEC_LOOP8(IDX, MID, RANK) \
    "1:\n" U2_DEC_STEP_X(IDX, MID, RANK, "20", "21", "22", "23") U2_DEC_STEP_X(IDX, MID, RANK, "30", "31", "32", "33") \
    U2_DEC_STEP_X(IDX, MID, RANK, "40", "41", "42", "43") U2_DEC_STEP_X(IDX, MID, RANK, "50", "51", "52", "53") \
    U2_DEC_STEP_X(IDX, MID, RANK, "60", "61", "62", "63") U2_DEC_STEP_X(IDX, MID, RANK, "70", "71", "72", "73") \
    U2_DEC_STEP_X(IDX, MID, RANK, "80", "81", "82", "83") \
    U2_DEC_TOP_L("90", "91", "92", "93") IDX MID RANK U2_DEC_BOT_CORE_T("-18", "35") "s_cbranch_scc1 1b\n s_branch 2f\n" \
    U2_DEC_SIDE("20", "21", "22", "23") U2_DEC_SIDE("30", "31", "32", "33") U2_DEC_SIDE("40", "41", "42", "43") \
    U2_DEC_SIDE("50", "51", "52", "53") U2_DEC_SIDE("60", "61", "62", "63") U2_DEC_SIDE("70", "71", "72", "73") \
    U2_DEC_SIDE("80", "81", "82", "83") U2_DEC_SIDE("90", "91", "92", "93") "2:\n"
#define U2_DEC_LOOP2(IDX, MID, RANK) \
    "1:\n" U2_DEC_TOP_L("20", "21", "22", "23") IDX MID RANK U2_DEC_BOT_CORE "s_cbranch_scc0 2f\n" \
    U2_DEC_TOP_L("30", "31", "32", "33") IDX MID RANK U2_DEC_BOT_CORE "s_cbranch_scc1 1b\n s_branch 2f\n" \
    U2_DEC_SIDE("20", "21", "22", "23") U2_DEC_SIDE("30", "31", "32", "33") "2:\n"
// store output-ring lanes [0, s72) at out[s87 - lane]; s87 -= s72
#if defined(VIDC_B2_DBG) && (VIDC_B2_DBG & 4)
#define U2_OUT_STORE ""
#else
#define U2_OUT_STORE "global_store_dwordx2 v58, v[6:7], s[88:89]\n"
#endif
#define U2_DEC_FLUSH \
    "v_cmp_gt_u32 vcc, s72, v2\n" \
    "v_sub_u32 v58, s87, v2\n" \
    "v_lshlrev_b32 v58, 3, v58\n" \
    "s_and_saveexec_b64 s[96:97], vcc\n" \
    U2_OUT_STORE \
    "s_mov_b64 exec, s[96:97]\n" \
    "s_sub_u32 s87, s87, s72\n"
#define U2_DEC_OUTER U2_DEC_OUTER_T("-2", "59")
#define U2_DEC_OUTER_T(RLO, RSPAN) \
    "s_lshr_b64 s[98:99], s[58:59], 31\n" \
    "s_cselect_b32 s99, 0, 1\n"                        /* head < 2^31 */ \
    "s_sub_u32 s68, s60, s61\n" \
    "s_add_u32 s68, s68, " RLO "\n" \
    "s_cmp_gt_u32 s68, " RSPAN "\n" \
    "s_cselect_b32 s99, 1, s99\n"                      /* ring needs the host code */ \
    "s_cmp_lt_u32 s69, 64\n" \
    "s_cbranch_scc1 5f\n" \
    "s_waitcnt vmcnt(0)\n"                             /* block change */ \
    "s_mov_b32 s72, 64\n" \
    U2_DEC_FLUSH \
    "s_add_u32 s85, s85, 64\n" \
    "s_sub_u32 s86, s86, 64\n" \
    U2_DEC_INSTALL \
    "s_mov_b32 s69, 0\n" \
    "s_min_u32 s70, s86, 64\n" \
    "5:\n" \
    "s_cmp_ge_u32 s69, s70\n" \
    "s_cbranch_scc1 9f\n" \
    "s_cmp_eq_u32 s99, 0\n" \
    "s_cbranch_scc1 1b\n" \
    "9:\n" \
    "s_mov_b32 s72, s69\n" \
    U2_DEC_FLUSH \
    "s_waitcnt vmcnt(0)\n"

// one generic decode step on the reversed structures (rare path): codec.cpp:144-150
template <int UB>
__device__ __forceinline__ uint32_t u2_slow_dec_step(uint64_t &head, WStack &st, uint32_t nmax, uint32_t &E1, v32u &ra, v32u &rb,
                                                     uint64_t *bm, uint32_t p0, uint32_t p1) {
    using U = U2Geom<UB>;
    const uint32_t lane = lane_id();
    ws_prepare(st);
    const uint32_t x = ans_id_pop(head, st, p0, p1);
    const uint32_t e = (x >> U::ESH) ^ (U::NE - 1u);
    const uint32_t L1 = e >> 6, L2 = e & 63u, g = (x >> 6) & (U::G - 1u), b = x & 63u;
    uint32_t row = u2_row_get(ra, rb, L1);
    uint32_t r = rl(E1, L1) + rl(row, L2);
    for (uint32_t j = 0; j < g; j++) r += popc64(rfl64(bm[e * U::G + j]));
    const uint64_t W = rfl64(bm[e * U::G + g]);
    r += popc64(W & ((1ull << b) - 1ull));
    ans_idx_push(head, st, r, nmax, 0x80000000u / nmax);
    E1 += lane < L1 ? 1u : 0u;
    row += lane < L2 ? 1u : 0u;
    u2_row_set(ra, rb, L1, row);
    bm[e * U::G + g] = W | (1ull << b);
    wave_sync();
    return x;
}

template <int UB>
__global__ void __launch_bounds__(64) k_roc_decode_u2(RocDecArgs a, const U2Div *__restrict__ dtab) {
    using U = U2Geom<UB>;
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *bm = (uint64_t *)smem;
    const uint32_t lane = lane_id();
    const uint32_t wi = blockIdx.x;
    if (wi >= a.nwork) return;
    const uint32_t l = rfl(a.worklist[wi]);
    const uint32_t n = rfl((uint32_t)(a.offsets[l + 1] - a.offsets[l]));
    const uint64_t ooff = rfl64(a.out_off ? a.out_off[wi] : a.offsets[l]);
    {
        uint4 *z = (uint4 *)smem;
        for (uint32_t w = lane; w < U::LDS_BYTES / 16u; w += 64) z[w] = make_uint4(0, 0, 0, 0);
    }
    const uint32_t P = rfl(a.prec[l]);
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    const uint32_t W0 = rfl(a.nwords[l]);
    WStack st;
    ws_init_loaded(st, a.words + rfl64(a.word_off[l]), W0, a.scratch_words + rfl64(a.scratch_off[wi]), roc_dec_stack_cap(n, W0),
                   rfl(a.draws[l]), a.mt, VIDC_MT_TABLE);
    const uint32_t draws0 = st.draws;
    uint64_t head = rfl64(a.heads[l]);
    uint32_t E1 = 0;
    v32u ra, rb;
#pragma unroll
    for (int c = 0; c < 32; c++) { ra[c] = 0u; rb[c] = 0u; }
    const uint32_t l3off = U::G == 4u ? (lane & 3u) * 8u : 0u;
    const uint32_t M1 = (1u << p1) - 1u, M0 = (1u << p0) - 1u;
    uint64_t *out = a.out + ooff;
    wave_sync();

#ifdef U2_PROF2
    uint32_t dbg[4] = {0, 0, 0, 0};
#define U2_DBG(k) dbg[k]++
#else
#define U2_DBG(k)
#endif
    uint32_t i = 0;  // ids decoded so far
    constexpr uint32_t RING_LO = U2_DEC_RING_LO, RING_HI = U2_DEC_RING_HI;  // (both geometries run the eight-copy loop since round 4)
    while (i < n) {
        {  // (the eight-copy loop wants 18..53 words: refill below 22 -- 21 + 32 = 53 -- instead of ws_prepare's 8)
            const uint32_t res = st.sp - st.lo;
            if (res > 56u) ws_spill32(st);
            else if (res < 22u && st.lo != 0u) ws_refill32(st);
        }
        if (lt_2p31(head) || st.sp - st.lo < RING_LO || st.sp - st.lo > RING_HI) {  // generic step
            U2_DBG(0);
            const uint32_t x = u2_slow_dec_step<UB>(head, st, i + 1u, E1, ra, rb, bm, p0, p1);
            if (lane == 0) out[n - 1u - i] = (uint64_t)x;
            i++;
            continue;
        }
        U2_DBG(1);
        uint32_t s_t = 0, s_N0 = rfl(i + 1u), s_left = rfl(n - i), s_oidx = rfl(n - 1u - i);
        uint64_t s_h = rfl64(head), oring = 0;
        st.sp = rfl(st.sp);
        st.lo = rfl(st.lo);
        // clang-format off
#define U2_DEC_ASM(BODY)                                                                                                  \
        asm volatile(BODY                                                                                                 \
            : "+{v4}"(E1), "+{v5}"(st.win), "+{v[6:7]}"(oring), "+{v[64:95]}"(ra), "+{v[96:127]}"(rb),                     \
              "+{s[58:59]}"(s_h), "+{s60}"(st.sp), "+{s69}"(s_t), "+{s85}"(s_N0), "+{s86}"(s_left), "+{s87}"(s_oidx)        \
            : "{v2}"(lane), "{v3}"(l3off), "{s61}"(st.lo), "{s73}"(M1), "{s74}"(M0), "{s76}"(p0), "{s77}"(p1),               \
              "{s[88:89]}"(out), "{s[94:95]}"(dtab)                                                                        \
            : "memory", "vcc", "scc", "v10", "v13", "v26", "v27", "v28", "v30", "v31", "v32", "v33", "v34", "v35", "v36",     \
              "v37", "v54", "v55", "v56", "v57", "v58", "v59", "s40", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",\
              "s51", "s52", "s53", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s70", "s71", "s72", "s75", "s96", "s97",\
              "s98", "s99")
        if (U::G == 4u) U2_DEC_ASM(U2_DEC_ENTRY U2_DEC_LOOP8(U2_DEC_IDX_G4, U2_DEC_MID, U2_DEC_RANK_G4) U2_DEC_OUTER_T("-18", "35"));
        else U2_DEC_ASM(U2_DEC_ENTRY U2_DEC_LOOP8(U2_DEC_IDX_G1, U2_DEC_MID, U2_DEC_RANK_G1) U2_DEC_OUTER_T("-18", "35"));
#undef U2_DEC_ASM
        // clang-format on
        head = s_h;
        ws_window(st);
        if (__builtin_expect(lt_2p31(head), 0)) {  // second half of the last index push (codec.cpp:59-61)
            ws_prepare(st);
            head = (uint64_t)ws_pop(st) | (head << 32);
        }
        i = s_N0 + s_t - 1u;
    }
    // a bitmap holds every id once: streams that decode to a multiset are decoded again with the side-list kernel body
    uint32_t bits = 0;
    for (uint32_t w = lane; w < U::BITMAP_BYTES / 8u; w += 64) bits += popc64(bm[w]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bits += (uint32_t)__shfl_xor((int)bits, o, 64);
#ifdef U2_PROF2
    if (lane == 0) { a.slots[0] = dbg[0]; a.slots[1] = dbg[1]; a.slots[2] = rfl(bits); a.slots[3] = n; }
#endif
    if (rfl(bits) != n) {
        wave_sync();
        roc_decode_u_body<UB>(a, smem);
        return;
    }
    if (lane == 0) {
        const bool clean = (head == VIDC_RANS_L) && (st.sp == st.draws - draws0);
        a.end_state[l] = clean ? 0u : 1u;
        a.status[l] = st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK;
    }
}

// ===============================================================================================================
// The decode loop for ids of any precision 12 <= P <= 31: k_roc_decode_b2.  The bitmap over the id universe of
// k_roc_decode_u2 becomes 4096 value buckets over the top 12 bits: the same reversed exclusive counters give the number
// of decoded ids in smaller buckets, the members of x's bucket (<= 64, unsorted u32) sit in a row of global memory that
// the wavefront reads with one load (lane j: member j) and compares with one v_cmp; the bucket sizes are u16 in LDS.
// Members stored during the current 64-step block may not have reached memory yet: they are taken from the output
// ring (lane t = the id of step t of the block) instead, and the row is only trusted up to the count it had when the
// block began (the block change waits for the stores, as in k_roc_decode_gen).  A bucket that overflows its row
// (skewed ids) flags the list: VIDC_ST_RETRY, the general kernel decodes it again.
//   additional registers: s78 bucket shift (P - 12)  s79 overflow flag  s[80:81] scratch mask  s[82:83] member rows
//   s48 bucket  s47 its size  s46 ring members below x  s64 ring members of the bucket, then all members below x
//   v26 LDS address of the bucket size  v30 size  v31 row member of the lane  v33 buckets of the ring ids
// (timing experiments only, -DVIDC_B2_DBG=bits: 1 = no row load, 2 = no member store, 4 = no output stores; results are wrong)
#ifndef VIDC_B2_DBG
#define VIDC_B2_DBG 0
#endif
#if VIDC_B2_DBG & 1
#define U2B_ROW_LOAD "v_mov_b32 v31, 0\n"
#else
#define U2B_ROW_LOAD "global_load_dword v31, v28, s[82:83]\n"           /* lane j: member j of the bucket's row */
#endif
#if VIDC_B2_DBG & 2
#define U2B_MEMBER_STORE ""
#else
#define U2B_MEMBER_STORE "global_store_dword v28, v27, s[82:83]\n"
#endif
#define U2B_DEC_IDX \
    "s_lshr_b32 s48, s40, s78\n"                       /* bucket */ \
    "v_lshlrev_b32_e64 v26, 1, s48\n" \
    "ds_read_u16 v30, v26\n"                           /* members of the bucket so far */ \
    "v_lshl_add_u32 v28, s48, 8, v3\n"                 /* v3 = 4 * lane: the lane's member of the 256-byte row */ \
    U2B_ROW_LOAD \
    "s_xor_b32 s45, s48, 0xfff\n"                      /* entry (reversed) */
#define U2B_DEC_MID_1 \
    "s_lshr_b32 s43, s45, 6\n"                         /* L1 */ \
    "s_and_b32 s44, s45, 63\n"                         /* L2 */ \
    "s_set_gpr_idx_on s43, gpr_idx(SRC0)\n" \
    "v_mov_b32 v13, v64\n" \
    "s_set_gpr_idx_off\n" \
    "v_lshrrev_b32 v33, s78, v6\n"                     /* buckets of the ids decoded in this block (output ring) */ \
    "v_cmp_eq_u32 vcc, s48, v33\n" \
    "v_cmp_gt_u32 s[66:67], s69, v2\n"                 /* ring lanes written in this block */ \
    "v_cmp_gt_u32 s[80:81], s40, v6\n"                 /* ... holding an id below x */ \
    "v_readlane_b32 s49, v4, s43\n"                    /* E1[L1] */ \
    "v_readlane_b32 s63, v13, s44\n"                   /* row[L2] */ \
    "v_readlane_b32 s62, v10, s69\n"                   /* lq of this step */ \
    "v_subrev_u32 v58, s43, v2\n" \
    "v_subrev_u32 v59, s44, v2\n" \
    "v_ashrrev_i32 v59, 31, v59\n" \
    "v_sub_u32 v13, v13, v59\n"                        /* row += 1 in lanes below L2 */ \
    "s_and_b64 s[66:67], s[66:67], vcc\n"              /* members of the bucket that are still in flight */ \
    "s_bcnt1_i32_b64 s64, s[66:67]\n" \
    "s_and_b64 s[66:67], s[66:67], s[80:81]\n" \
    "s_bcnt1_i32_b64 s46, s[66:67]\n"                  /* ... of them below x */
#define U2B_DEC_MID_2 \
    "s_and_b32 m0, s60, 63\n"                          /* index push, first half (codec.cpp:44-63): renormalise H */ \
    "s_cmp_ge_u32 s59, s62\n" \
    "v_writelane_b32 v5, s58, m0\n" \
    "s_cselect_b32 s50, s59, s58\n" \
    "s_cselect_b32 s51, 0, s59\n" \
    "s_addc_u32 s60, s60, 0\n" \
    "s_mul_i32 s52, s50, s75\n"                        /* H * nmax */ \
    "s_mul_hi_u32 s53, s50, s75\n" \
    "s_mul_i32 s68, s51, s75\n" \
    "s_add_u32 s53, s53, s68\n" \
    "s_add_u32 s72, s49, s63\n" \
    "s_set_gpr_idx_on s43, gpr_idx(DST)\n" \
    "v_mov_b32 v64, v13\n" \
    "s_set_gpr_idx_off\n" \
    "s_mov_b32 m0, s69\n"
#define U2B_DEC_MID U2B_DEC_MID_1 U2B_DEC_MID_2
// Round 5, second half: the row load sized by the bucket's member count (k_roc_decode_b2<0, 2>).  A step of the plain form reads the
// whole 256-byte row -- four 64-byte sectors of a 1 MiB table that no cache holds -- although the bucket has 4 .. 16 members for most
// of a list's life (16 385 .. 65 536 ids over 4096 buckets): 19.6 GB of S2's 134 GB fetched.  Here the load waits for the LDS read
// of the count (~20 instructions later) and runs under exec = lanes below the count: one sector for up to 16 members, none for an
// empty bucket.  Lanes that are not loaded keep a stale member; the rank only looks at lanes below the visible count
// (U2B_DEC_RANK_TAIL).  A lone chain pays for the later load (0.207 -> 0.254 us per step at 30 bits) and keeps the plain form; a call of
// hundreds of chains does not (S2 decode 70.9 against 70.9 ms interleaved) and fetches a third.  Measured next to it and dropped: lanes
// 0 .. 15 at once as before + the lanes beyond under the count (0.227 us alone, S2 decode +1.2 ms: two more scalar instructions).
#define U2B_DEC_IDX_MC \
    "s_lshr_b32 s48, s40, s78\n"                       /* bucket */ \
    "v_lshlrev_b32_e64 v26, 1, s48\n" \
    "ds_read_u16 v30, v26\n"                           /* members of the bucket so far */ \
    "v_lshl_add_u32 v28, s48, 8, v3\n" \
    "s_xor_b32 s45, s48, 0xfff\n"                      /* entry (reversed) */
#define U2B_DEC_MID_MC U2B_DEC_MID_1 \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_cmp_gt_u32 s[80:81], v30, v2\n"                 /* lanes holding a member */ \
    "s_mov_b64 exec, s[80:81]\n" \
    U2B_ROW_LOAD \
    "s_mov_b64 exec, -1\n" \
    U2B_DEC_MID_2
// (round 5: the bucket's member count stays in the VGPR the LDS read left it in -- every lane holds it --, so the slot address, the
// new count, the "visible in memory" bound and the overflow test are vector instructions: 5 scalar instructions instead of 15 in this
// tail.  A chain alone pays four cycles per instruction of either kind, but with four chains per SIMD the scalar slot is the one they
// queue for, DESIGN section 12.  v29 = largest count seen: the row is full when it passes 63, tested once behind the loop.)
#define U2B_DEC_RANK \
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n" \
    U2B_DEC_RANK_TAIL
#define U2B_DEC_RANK_TAIL \
    "v_cmp_gt_u32 s[66:67], s40, v31\n"                /* row member below x */ \
    "v_subrev_u32 v33, s64, v30\n"                     /* members visible in memory: not those of this block */ \
    "v_min_u32 v32, 63, v30\n"                         /* slot of x in the row */ \
    "v_cmp_gt_u32 vcc, v33, v2\n"                      /* the lane holds a visible member */ \
    "v_mov_b32 v27, s40\n" \
    "v_lshl_add_u32 v28, v32, 2, v28\n"                /* lane 0: row + 4 * slot */ \
    "v_max_u32 v29, v29, v30\n" \
    "s_and_b64 vcc, vcc, s[66:67]\n" \
    "v_add_u32 v32, 1, v30\n" \
    "s_bcnt1_i32_b64 s64, vcc\n" \
    "s_add_u32 s64, s64, s46\n"                        /* members of the bucket below x */ \
    "s_mov_b64 exec, 1\n" \
    U2B_MEMBER_STORE \
    "ds_write_b16 v26, v32\n" \
    "s_mov_b64 exec, -1\n" \
    U2_DEC_AFTER_RANK

// Round 5: the row of the NEXT step requested one step ahead (PF = true; calls with hundreds of such chains, whose rows -- 1 MiB
// per list -- come from HBM: the step then lasts one memory round trip, 0.7 us with the chains alone and 1.1 us next to the other
// classes of an S2-sized call).  head' = H * nmax + r, and the next id's high slice is head' mod 2^p1 (codec.cpp:107-121: the high
// slice is popped first, from the low bits of the head): everything but the rank INSIDE x's bucket is known from the on-chip
// counters before this step's row arrives -- r lies in [r_e, r_e + c] with r_e = ids in smaller buckets and c = members of x's
// bucket --, so the next high slice is one of c + 1 CONSECUTIVE values and the next bucket (high slice >> (bsh - 16), P >= 28)
// one of ((v0 & m) + c >> sh) + 1 consecutive rows.  The first two of them are requested now -- every byte, 8 per lane, result
// never read -- so that the demand load of the next step finds them in (or on their way to) the L2: two steps per
// memory round trip instead of one.  No effect on what is computed.
//   s90 bsh - 16   s[92:93] lanes allowed to prefetch (all, or lane 0 for P < 28)   v34 address   v[36:37] dummy
#define U2B_DEC_PF \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_readfirstlane_b32 s47, v30\n"                   /* members of the bucket */ \
    "s_add_u32 s65, s52, s72\n"                        /* low word of the smallest head' this step can produce */ \
    "s_and_b32 s65, s65, s73\n"                        /* ... its high slice = the next id's */ \
    "s_bfm_b32 s68, s90, 0\n" \
    "s_and_b32 s68, s65, s68\n" \
    "s_add_u32 s68, s68, s47\n" \
    "s_lshr_b32 s68, s68, s90\n"                       /* candidate rows beyond the first */ \
    "s_min_u32 s68, s68, 1\n"                          /* (two rows at most: every byte of a row is requested, 8 per lane) */ \
    "s_lshl_b32 s68, s68, 5\n" \
    "s_add_u32 s68, s68, 31\n"                         /* 32 lanes per row; 63 -> all lanes */ \
    "s_bfm_b64 s[80:81], s68, 0\n" \
    "s_lshl_b64 s[80:81], s[80:81], 1\n" \
    "s_or_b32 s80, s80, 1\n" \
    "s_and_b64 s[80:81], s[80:81], s[92:93]\n" \
    "s_lshr_b32 s65, s65, s90\n"                       /* first candidate bucket */ \
    "s_lshl_b32 s65, s65, 8\n" \
    "v_lshl_add_u32 v34, v2, 3, s65\n" \
    "v_and_b32 v34, 0xfffff, v34\n"                    /* (wraps inside the list's 4096 rows) */ \
    "s_mov_b64 exec, s[80:81]\n" \
    "global_load_dwordx2 v[36:37], v34, s[82:83]\n" \
    "s_mov_b64 exec, -1\n"
// (the prefetch is the youngest vector-memory operation: everything older, this step's row included, has returned at vmcnt(1))
#define U2B_DEC_RANK_PF \
    U2B_DEC_PF \
    "s_waitcnt vmcnt(1)\n" \
    U2B_DEC_RANK_TAIL
// The same with the member rows in LDS (short lists: 128 or 256 buckets x 64 members behind the bucket sizes, 33 / 66 KiB --
// few large rows rather than many small ones: a row must practically never overflow, because ONE list handed back costs the
// call a whole serial chain on the general kernel afterwards): LDS operations of a wavefront are ordered, so a member is
// visible to the next step and the output-ring detour is not needed.
//   s82 byte offset of the rows in LDS (= 2 x buckets)
#define U2L_DEC_IDX \
    "s_lshr_b32 s48, s40, s78\n"                       /* bucket */ \
    "v_lshlrev_b32_e64 v26, 1, s48\n" \
    "ds_read_u16 v30, v26\n"                           /* members of the bucket so far */ \
    "v_lshl_add_u32 v28, s48, 8, v3\n"                 /* v3 = start of the rows in LDS + 4 * lane */ \
    "ds_read_b32 v31, v28\n"                           /* lane j: member j of the bucket's row */ \
    "s_xor_b32 s45, s48, 0xfff\n"                      /* entry (reversed) */
#define U2L_DEC_MID \
    "s_lshr_b32 s43, s45, 6\n"                         /* L1 */ \
    "s_and_b32 s44, s45, 63\n"                         /* L2 */ \
    "s_set_gpr_idx_on s43, gpr_idx(SRC0)\n" \
    "v_mov_b32 v13, v64\n" \
    "s_set_gpr_idx_off\n" \
    "v_readlane_b32 s49, v4, s43\n"                    /* E1[L1] */ \
    "v_readlane_b32 s63, v13, s44\n"                   /* row[L2] */ \
    "v_readlane_b32 s62, v10, s69\n"                   /* lq of this step */ \
    "v_subrev_u32 v58, s43, v2\n" \
    "v_subrev_u32 v59, s44, v2\n" \
    "v_ashrrev_i32 v59, 31, v59\n" \
    "v_sub_u32 v13, v13, v59\n"                        /* row += 1 in lanes below L2 */ \
    "s_and_b32 m0, s60, 63\n"                          /* index push, first half (codec.cpp:44-63): renormalise H */ \
    "s_cmp_ge_u32 s59, s62\n" \
    "v_writelane_b32 v5, s58, m0\n" \
    "s_cselect_b32 s50, s59, s58\n" \
    "s_cselect_b32 s51, 0, s59\n" \
    "s_addc_u32 s60, s60, 0\n" \
    "s_mul_i32 s52, s50, s75\n"                        /* H * nmax */ \
    "s_mul_hi_u32 s53, s50, s75\n" \
    "s_mul_i32 s68, s51, s75\n" \
    "s_add_u32 s53, s53, s68\n" \
    "s_add_u32 s72, s49, s63\n" \
    "s_set_gpr_idx_on s43, gpr_idx(DST)\n" \
    "v_mov_b32 v64, v13\n" \
    "s_set_gpr_idx_off\n" \
    "s_mov_b32 m0, s69\n"
// (round 5: the member count stays in its VGPR as in U2B_DEC_RANK_TAIL -- 4 scalar instructions instead of 14; v29 = largest count)
#define U2L_DEC_RANK \
    "s_waitcnt lgkmcnt(0)\n" \
    "v_cmp_gt_u32 s[66:67], s40, v31\n"                /* row member below x */ \
    "v_min_u32 v32, 63, v30\n"                         /* slot of x in the row */ \
    "v_cmp_gt_u32 vcc, v30, v2\n"                      /* the lane holds a member */ \
    "v_mov_b32 v27, s40\n" \
    "v_lshl_add_u32 v28, v32, 2, v28\n"                /* lane 0: row + 4 * slot */ \
    "v_max_u32 v29, v29, v30\n" \
    "s_and_b64 vcc, vcc, s[66:67]\n" \
    "v_add_u32 v32, 1, v30\n" \
    "s_bcnt1_i32_b64 s64, vcc\n"                       /* members of the bucket below x */ \
    "s_mov_b64 exec, 1\n" \
    "ds_write_b32 v28, v27\n" \
    "ds_write_b16 v26, v32\n" \
    "s_mov_b64 exec, -1\n" \
    U2_DEC_AFTER_RANK

// one generic decode step on the bucket structure (rare path): codec.cpp:144-150
__device__ __forceinline__ uint32_t u2b_slow_dec_step(uint64_t &head, WStack &st, uint32_t nmax, uint32_t &E1, v32u &ra, v32u &rb,
                                                      uint16_t *cnt16, uint32_t *rows, uint32_t bsh, uint32_t p0, uint32_t p1,
                                                      uint32_t &ovf, uint32_t nbk = 4096u, uint32_t cap = 64u) {
    const uint32_t lane = lane_id();
    ws_prepare(st);
    const uint32_t x = ans_id_pop(head, st, p0, p1);
    const uint32_t bkt = rfl((x >> bsh) & (nbk - 1u));
    const uint32_t e = bkt ^ 0xfffu, L1 = e >> 6, L2 = e & 63u;
    uint32_t row = u2_row_get(ra, rb, L1);
    uint32_t r = rl(E1, L1) + rl(row, L2);
    const uint32_t c = rfl((uint32_t)cnt16[bkt]);
    const uint32_t m = c < cap ? c : cap;
    const uint32_t y = lane < m ? rows[bkt * cap + lane] : 0xffffffffu;
    r += popc64(ballot(lane < m && y < x));
    ans_idx_push(head, st, r, nmax, 0x80000000u / nmax);
    E1 += lane < L1 ? 1u : 0u;
    row += lane < L2 ? 1u : 0u;
    u2_row_set(ra, rb, L1, row);
    if (c < cap) {
        if (lane == 0) rows[bkt * cap + c] = x;
    } else {
        ovf |= 1u;
    }
    if (lane == 0) cnt16[bkt] = (uint16_t)(c + 1u);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the store is in memory before the loop reads rows again
    wave_sync();
    return x;
}

#define VIDC_B2_LDS_BYTES ((4096u + 8u) * 2u)
#define VIDC_B2L_CAP 64u
#define VIDC_B2L_MAX_LIST 4096u       // (the general decoder's small class)
#define VIDC_B2L_MID_LIST 2048u
#define VIDC_B2L_LOAD 30u             // ids per bucket (Poisson(30) exceeds 64 once in ~10^8 buckets)
#define VIDC_B2L_LDS_BYTES(BK) ((BK) * 2u + (BK) * VIDC_B2L_CAP * 4u)
// BK = 0: 4096 buckets, member rows in global memory; BK = 32 / 64 / 128 / 256: that many buckets, rows in LDS
// MODE (BK = 0): 0 whole-row load, 1 look-ahead (U2B_DEC_PF), 2 row load sized by the member count (U2B_DEC_IDX_MC)
template <int BK, int MODE = 0>
__device__ __forceinline__ void roc_decode_b2_body(RocDecArgs a, const U2Div *__restrict__ dtab) {
    static_assert(MODE == 0 || BK == 0, "look-ahead and sized loads are for rows in global memory");
    constexpr bool PF = MODE == 1;
    constexpr bool LROWS = BK != 0;
    constexpr uint32_t VIDC_B2L_BUCKETS = LROWS ? (uint32_t)BK : 4096u;
    // (dynamic LDS, the kernel's only allocation: the asm addresses the bucket sizes from LDS offset 0, like the bitmap of
    // the u2 kernels)
    extern __shared__ __align__(16) unsigned char smem[];
    uint16_t *cnt16 = (uint16_t *)smem;
    const uint32_t lane = lane_id();
    const uint32_t wi = blockIdx.x;
    if (wi >= a.nwork) return;
    const uint32_t l = rfl(a.worklist[wi]);
    const uint32_t n = rfl((uint32_t)(a.offsets[l + 1] - a.offsets[l]));
    const uint64_t ooff = rfl64(a.out_off ? a.out_off[wi] : a.offsets[l]);
    {
        uint4 *z = (uint4 *)cnt16;
        for (uint32_t w = lane; w < (LROWS ? VIDC_B2L_BUCKETS : 4096u + 8u) * 2u / 16u; w += 64) z[w] = make_uint4(0, 0, 0, 0);
    }
    if (!LROWS && a.lpw) {
        // a.lpw (unused by wave-per-list kernels otherwise) = the longest list of the launch: the wavefronts of the longest chains
        // win the instruction arbitration of their SIMD against shorter chains and the other kernel classes of the call (the call
        // ends with its longest chain; with four or more wavefronts per SIMD a chain step is scalar-issue bound)
        const uint32_t top = rfl(a.lpw);
        if (4u * n >= 3u * top) __builtin_amdgcn_s_setprio(3);
        else if (2u * n >= top) __builtin_amdgcn_s_setprio(2);
        else if (4u * n >= top) __builtin_amdgcn_s_setprio(1);
    }
    const uint32_t P = rfl(a.prec[l]);
    const uint32_t p0 = P < 16u ? P : 16u, p1 = P > 16u ? (P - 16u > 16u ? 16u : P - 16u) : 0u;
    const uint32_t bbits = BK == 32 ? 5u : (BK == 64 ? 6u : (BK == 128 ? 7u : (BK == 256 ? 8u : 12u)));
    const uint32_t bsh = P > bbits ? P - bbits : 0u;
    const uint32_t W0 = rfl(a.nwords[l]);
    WStack st;
    ws_init_loaded(st, a.words + rfl64(a.word_off[l]), W0, a.scratch_words + rfl64(a.scratch_off[wi]), roc_dec_stack_cap(n, W0),
                   rfl(a.draws[l]), a.mt, VIDC_MT_TABLE);
    const uint32_t draws0 = st.draws;
    uint64_t head = rfl64(a.heads[l]);
    uint32_t E1 = 0;
    v32u ra, rb;
#pragma unroll
    for (int c = 0; c < 32; c++) { ra[c] = 0u; rb[c] = 0u; }
    const uint32_t l3off = lane * 4u + (LROWS ? VIDC_B2L_BUCKETS * 2u : 0u);  // (v3 of the loop: the lane's member inside a row)
    const uint32_t M1 = (1u << p1) - 1u, M0 = (1u << p0) - 1u;
    uint64_t *out = a.out + ooff;
    uint32_t *rows = LROWS ? (uint32_t *)(smem + VIDC_B2L_BUCKETS * 2u) : a.slots + rfl64(a.slots_off[wi]);
    uint32_t ovf = 0;
    wave_sync();

    uint32_t i = 0;  // ids decoded so far
    // the eight-copy loop of the bitmap kernels (ring tested by the last copy only, U2_DEC_RING_LO .. HI words on entry): four scalar
    // instructions less in seven steps out of eight
    constexpr uint32_t RING_LO = U2_DEC_RING_LO, RING_HI = U2_DEC_RING_HI;
    while (i < n) {
        {
            const uint32_t res = st.sp - st.lo;
            if (res > 56u) ws_spill32(st);
            else if (res < 22u && st.lo != 0u) ws_refill32(st);
        }
        if (lt_2p31(head) || st.sp - st.lo < RING_LO || st.sp - st.lo > RING_HI) {  // generic step
            const uint32_t x = u2b_slow_dec_step(head, st, i + 1u, E1, ra, rb, cnt16, rows, bsh, p0, p1, ovf,
                                                 LROWS ? VIDC_B2L_BUCKETS : 4096u, LROWS ? VIDC_B2L_CAP : 64u);
            if (lane == 0) out[n - 1u - i] = (uint64_t)x;
            i++;
            continue;
        }
        uint32_t s_t = 0, s_N0 = rfl(i + 1u), s_left = rfl(n - i), s_oidx = rfl(n - 1u - i), s_ovf = rfl(ovf);
        uint64_t s_h = rfl64(head), oring = 0;
        uint32_t vmaxc = 0;  // the largest member count a step found in its bucket
        st.sp = rfl(st.sp);
        st.lo = rfl(st.lo);
        const uint64_t rowbase = LROWS ? (uint64_t)(VIDC_B2L_BUCKETS * 2u) : rfl64((uint64_t)rows);
        const uint32_t pf_sh = bsh > 16u ? bsh - 16u : 0u;              // next bucket = next high slice >> pf_sh (P >= 28)
        const uint64_t pf_lanes = rfl64(bsh >= 16u ? ~0ull : 1ull);     // (P < 28: the candidates are not rows in a run; one harmless line)
        // clang-format off
#define U2B_DEC_ASM(BODY)                                                                                                 \
        asm volatile(BODY                                                                                                 \
            : "+{v4}"(E1), "+{v5}"(st.win), "+{v[6:7]}"(oring), "+{v[64:95]}"(ra), "+{v[96:127]}"(rb), "+{v29}"(vmaxc),   \
              "+{s[58:59]}"(s_h), "+{s60}"(st.sp), "+{s69}"(s_t), "+{s85}"(s_N0), "+{s86}"(s_left), "+{s87}"(s_oidx), "+{s79}"(s_ovf)\
            : "{v2}"(lane), "{v3}"(l3off), "{s61}"(st.lo), "{s73}"(M1), "{s74}"(M0), "{s76}"(p0), "{s77}"(p1),            \
              "{s[88:89]}"(out), "{s[94:95]}"(dtab), "{s78}"(bsh), "{s[82:83]}"(rowbase), "{s90}"(pf_sh), "{s[92:93]}"(pf_lanes)\
            : "memory", "vcc", "scc", "v10", "v13", "v26", "v27", "v28", "v30", "v31", "v32", "v33", "v34", "v35", "v36", \
              "v37", "v54", "v55", "v56", "v57", "v58", "v59", "s40", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50",\
              "s51", "s52", "s53", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s70", "s71", "s72", "s75", "s80", "s81",\
              "s96", "s97", "s98", "s99")
        if (LROWS) U2B_DEC_ASM(U2_DEC_ENTRY U2_DEC_LOOP8(U2L_DEC_IDX, U2L_DEC_MID, U2L_DEC_RANK) U2_DEC_OUTER_T("-18", "35"));
        else if (PF) U2B_DEC_ASM(U2_DEC_ENTRY U2_DEC_LOOP8(U2B_DEC_IDX, U2B_DEC_MID, U2B_DEC_RANK_PF) U2_DEC_OUTER_T("-18", "35"));
        else if (MODE == 2) U2B_DEC_ASM(U2_DEC_ENTRY U2_DEC_LOOP8(U2B_DEC_IDX_MC, U2B_DEC_MID_MC, U2B_DEC_RANK) U2_DEC_OUTER_T("-18", "35"));
        else U2B_DEC_ASM(U2_DEC_ENTRY U2_DEC_LOOP8(U2B_DEC_IDX, U2B_DEC_MID, U2B_DEC_RANK) U2_DEC_OUTER_T("-18", "35"));
#undef U2B_DEC_ASM
        // clang-format on
        ovf = s_ovf | (rfl(vmaxc) > 63u ? 1u : 0u);  // a full row: the list is decoded again by the general kernel
        head = s_h;
        ws_window(st);
        if (__builtin_expect(lt_2p31(head), 0)) {  // second half of the last index push (codec.cpp:59-61)
            ws_prepare(st);
            head = (uint64_t)ws_pop(st) | (head << 32);
        }
        i = s_N0 + s_t - 1u;
    }
    if (lane == 0) {
        const bool retry = rfl(ovf) != 0u;
        const bool clean = (head == VIDC_RANS_L) && (st.sp == st.draws - draws0);
        a.end_state[l] = (clean || retry) ? 0u : 1u;
        a.status[l] = retry ? 5u /* VIDC_ST_RETRY (roc_lane.h) */ : (st.err ? ((st.err & 1u) ? VIDC_ST_OVERFLOW : VIDC_ST_MT) : VIDC_ST_OK);
    }
}
template <int BK, int MODE = 0>
__global__ void __launch_bounds__(64) k_roc_decode_b2(RocDecArgs a, const U2Div *__restrict__ dtab) {
    roc_decode_b2_body<BK, MODE>(a, dtab);
}

}  // namespace dev
}  // namespace vidc
