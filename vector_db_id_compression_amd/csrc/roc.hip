// roc.hip -- host side of the ROC codec: scheduling of lists onto wavefronts, arenas, compaction,
// and the vidc_roc_* C-ABI (include/vidc.h).
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <memory>
#include <mutex>
#include <numeric>
#include <thread>

#include "common.h"
#include "roc_kernels.h"
#include "roc_u.h"
#include "roc_u2.h"
#include "rows_csr.h"
#include "roc_lane.h"
#include "roc_grp.h"
#include "scan.h"

using namespace vidc;
using namespace vidc::dev;

struct vidc_roc {
    int device = 0;
    uint64_t nlist = 0, ntotal = 0;
    bool rows = false;
    uint32_t K = 0;
    // Host copies of the per-list metadata.  The device arrays are authoritative; the host side is filled lazily
    // (ensure_meta / ensure_offsets) because with ~10^6 lists or graph nodes the PCIe round trip of these arrays
    // costs more than the codec kernels.  `offsets` is always valid for IVF objects (it is an input), `prec` holds
    // the values the decode planner needs (lists above TINY_MAX) as soon as encode returns.
    mutable std::vector<uint64_t> offsets;   // nlist+1
    mutable std::vector<uint32_t> prec, nwords, draws;
    // HEURISTIC input of the decode planner (bucket geometry of the b2 kernels), never a correctness input: the prepass' view of
    // a list's largest id -- its true maximum after the full prepass, its LAST id after the light one (== the maximum of an
    // ascending list; a list that turned out not to be ascending is redone by the sorting pass and set to 0 here).  A value
    // below the true maximum only makes the planner assume more ids per bucket.  Empty / 0 = unknown (imported streams).
    std::vector<uint32_t> umax;
    // every list of the object, longest first (stable), when the encoder built that order (calls classified by length alone);
    // empty otherwise.  The decode planner of the whole object cuts its classes out of it.
    std::vector<uint32_t> order_desc;
    uint64_t order_max_n = 0;
    mutable std::vector<uint64_t> heads;
    mutable std::vector<uint64_t> word_off;  // nlist+1
    mutable bool meta_host = false, offsets_host = false;
    mutable std::mutex mu;  // guards the lazy host mirrors and the cached decode plan (objects are otherwise immutable)
    uint64_t total_words = 0, compressed_bytes = 0;
    // device-resident compressed representation
    DevBuf<uint64_t> d_offsets, d_heads, d_word_off;
    DevBuf<uint32_t> d_prec, d_nwords, d_draws, d_words, d_perm;
    mutable uint64_t last_nonclean = 0;
    // graph objects: the rows ordered by edge count (k_rows_order_*), built by the first whole-graph decode_rows (guarded by mu)
    mutable DevBuf<uint32_t> d_row_order;
    mutable bool row_order_ready = false;
    // decode_all plan, built on first use (depends only on the immutable metadata)
    mutable std::shared_ptr<struct DecPlanCache> plan_all;
    // the same plan without its device arrays, built by the encoder while its kernels run (the host would only wait)
    mutable std::shared_ptr<struct DecPlanCache> plan_ahead;
    ~vidc_roc() {  // the large host arrays go back to the process-wide vector cache (common.h)
        vec_pool<uint64_t>().give(std::move(offsets)); vec_pool<uint64_t>().give(std::move(heads)); vec_pool<uint64_t>().give(std::move(word_off));
        vec_pool<uint32_t>().give(std::move(prec)); vec_pool<uint32_t>().give(std::move(nwords)); vec_pool<uint32_t>().give(std::move(draws));
        vec_pool<uint32_t>().give(std::move(umax)); vec_pool<uint32_t>().give(std::move(order_desc));
    }
};

namespace {

constexpr uint32_t TINY_MAX = 64;
constexpr uint32_t GEN_SMALL_MAX = 4096;   // decoder: fb <= 9 -> 2 KiB of LDS (full occupancy)
// Lists longer than this go to the bitmap kernels (one wave per CU, lowest step latency: they are the critical
// path); shorter ones cost about the same per step in the general kernels, which keep thousands in flight.
constexpr uint32_t U_MIN_LIST = 4097;

inline uint64_t arena_words_for(uint64_t n) { return n * 37 / 32 + 8; }  // <= P+4 bits of growth per step, P <= 32

static_assert(VIDC_LANE_MAX64 == VIDC_LANE_TAB, "divisor table size");
// Kernel family for short lists.  The lane-per-list kernels advance 64 lists per instruction but every list still
// pays the full per-step latency, so they only win once a call has enough lists to fill the machine with them
// (measured crossover ~8 000 lists of 65..1024 ids, ~2 000 tiny lists); smaller calls -- e.g. the few hundred lists
// a search touches -- keep the latency-optimised wave-per-list kernels.  Test hooks: VIDC_NO_LANE=1 (never),
// VIDC_FORCE_LANE=1 (always); both families produce the same bits.
constexpr uint64_t R2_MIN_LIST = 256;   // shorter lists: the general kernel (the chain kernel's set-up -- 32 KiB bitmap, counters -- is ~10 us)
constexpr uint64_t LANE_MIN_LISTS = 8192, LANE_MIN_LISTS64 = 8192, LANE_MIN_TINY = 2048;  // 64: lists of 1025..4096 ids
enum LanePolicy { LANE_NEVER = 0, LANE_AUTO = 1, LANE_ALWAYS = 2 };
inline LanePolicy lane_policy() {
    const char *e = getenv("VIDC_NO_LANE");
    if (e && e[0] == '1') return LANE_NEVER;
    e = getenv("VIDC_FORCE_LANE");
    if (e && e[0] == '1') return LANE_ALWAYS;
    return LANE_AUTO;
}
inline bool lane_wanted(LanePolicy p, uint64_t nlists, uint64_t min_lists) {
    return p == LANE_ALWAYS || (p == LANE_AUTO && nlists >= min_lists);
}

inline bool env_on(const char *name) {  // set, not empty, not "0"
    const char *e = std::getenv(name);
    return e && *e && !(e[0] == '0' && !e[1]);
}
// Row-per-list kernels (roc_grp.h: 16 lanes per list, four lists per wavefront) for lists of 4097 .. 131 072 ids.  They are
// throughput kernels -- a step costs about as many instructions as a wave-per-list step but advances four lists -- and
// every list pays a longer step (LDS round trips instead of v_readlane), so they only pay once a call has more such
// lists than the wave-per-list kernels keep resident.  Test hooks: VIDC_NO_GRP=1 (never), VIDC_FORCE_GRP=1 (every list
// of 65 .. 131 072 ids).
constexpr uint64_t GRP_MIN_LISTS = 8192;
struct GrpPolicy { uint64_t min_lists, min_n, max_n, dec_max_n, dec_min_n; };
inline GrpPolicy grp_policy() {
    // (lists beyond 32 768 ids are chains of >= 40 ms at this family's 1.2 us per step: they keep the lower-latency
    // wave-per-list kernels unless VIDC_FORCE_GRP says otherwise)
    // Decode: up to 16 384 ids -- the 16 385..32 768-id lists are the longest chains next to the b2 / general-kernel lists of a
    // big call, and their step under load is 1.7-2.6 us here against ~1 us on the general decoder (S2: 53-85 ms against 32).
    GrpPolicy g{GRP_MIN_LISTS, VIDC_LANE_MAX64 + 1u, 32768u, 16384u, VIDC_LANE_MAX64 + 1u};
    if (env_on("VIDC_NO_GRP") || env_on("VIDC_FORCE_GENERAL") || env_on("VIDC_OLD_U")) { g.min_lists = ~0ull; return g; }
    if (env_on("VIDC_FORCE_GRP")) { g.min_lists = 0; g.min_n = g.dec_min_n = VIDC_GRP_MIN_LIST; g.max_n = g.dec_max_n = VIDC_GRP_MAX_LIST; }
    return g;
}

// wavefronts of a bucket-row lane decoder launch: one per group of lists
inline uint32_t lane_grid(uint32_t groups) { return groups; }

// test hook: VIDC_OLD_U=1 keeps the round-1 bitmap kernels (roc_u.h) instead of the hand-scheduled ones (roc_u2.h)
// Lists per wavefront of a lane-per-list launch (VIDC_LPW=8|16|32, measurements only; default 64).  Every list of a
// launch is in flight at once either way, so fewer lists per wavefront only adds wavefronts: measured no better on
// 65 536 x 256 ids (decode 0.99 / 1.01 / 1.04 / 1.16 ms for 64 / 32 / 16 / 8), worse for the 16 / 64-word strips.
inline uint32_t lane_lists_per_wave(const vidc_ctx *, uint32_t, uint32_t) {
    static const int forced = [] { const char *e = std::getenv("VIDC_LPW"); return e ? std::atoi(e) : 0; }();
    return (forced == 8 || forced == 16 || forced == 32) ? (uint32_t)forced : 64u;
}
inline bool old_u_kernels() {
    const char *e = getenv("VIDC_OLD_U");
    return e && e[0] == '1';
}

// test hook: VIDC_FORCE_GENERAL=1 routes every list through the general (sorted-position / bucket) kernels
inline bool force_general() {
    const char *e = getenv("VIDC_FORCE_GENERAL");
    return e && e[0] == '1';
}

// kernels that ask for more than 64 KiB of dynamic LDS must opt in
inline int set_big_lds(const void *fn, size_t bytes) {
    VIDC_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return VIDC_OK;
}

template <typename T>
int upload(vidc_ctx *ctx, DevBuf<T> &dst, const std::vector<T> &src) {
    VIDC_TRY(dst.alloc(src.size(), ctx->dpool));
    if (!src.empty())
        VIDC_HIP(hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return VIDC_OK;
}
template <typename T>
int upload_scratch(vidc_ctx *ctx, Scratch &dst, const std::vector<T> &src) {
    VIDC_TRY(dst.get(ctx, src.size() * sizeof(T)));
    if (!src.empty())
        VIDC_HIP(hipMemcpyAsync(dst.p, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return VIDC_OK;
}

// device -> host through pinned staging (blocks until the data is there)
template <typename T>
int download(vidc_ctx *ctx, std::vector<T> &dst, const T *d_src, size_t count) {
    dst.resize(count);
    if (!count) return VIDC_OK;
    Pinned st;
    VIDC_TRY(st.get(ctx, count * sizeof(T)));
    VIDC_HIP(hipMemcpyAsync(st.p, d_src, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    std::memcpy(dst.data(), st.p, count * sizeof(T));
    return VIDC_OK;
}

// lazy host mirrors of the device-resident metadata (blocking copies; everything was synchronised when the
// object was built)
template <typename T>
int mirror(std::vector<T> &dst, const T *d_src, size_t count) {
    dst.resize(count);
    if (count) VIDC_HIP(hipMemcpy(dst.data(), d_src, count * sizeof(T), hipMemcpyDeviceToHost));
    return VIDC_OK;
}
int ensure_offsets_locked(const vidc_roc *r) {
    if (r->offsets_host) return VIDC_OK;
    VIDC_HIP(hipSetDevice(r->device));
    VIDC_TRY(mirror(r->offsets, (const uint64_t *)r->d_offsets.p, r->nlist + 1));
    r->offsets_host = true;
    return VIDC_OK;
}
int ensure_offsets(const vidc_roc *r) {
    std::lock_guard<std::mutex> g(r->mu);
    return ensure_offsets_locked(r);
}
int ensure_meta(const vidc_roc *r) {
    std::lock_guard<std::mutex> g(r->mu);
    VIDC_TRY(ensure_offsets_locked(r));
    if (r->meta_host) return VIDC_OK;
    VIDC_HIP(hipSetDevice(r->device));
    VIDC_TRY(mirror(r->prec, (const uint32_t *)r->d_prec.p, r->nlist));
    VIDC_TRY(mirror(r->nwords, (const uint32_t *)r->d_nwords.p, r->nlist));
    VIDC_TRY(mirror(r->draws, (const uint32_t *)r->d_draws.p, r->nlist));
    VIDC_TRY(mirror(r->heads, (const uint64_t *)r->d_heads.p, r->nlist));
    VIDC_TRY(mirror(r->word_off, (const uint64_t *)r->d_word_off.p, r->nlist + 1));
    r->meta_host = true;
    return VIDC_OK;
}

struct EventTimer {
    vidc_ctx *c;
    explicit EventTimer(vidc_ctx *ctx) : c(ctx) { (void)hipEventRecord(c->ev0, c->stream); }
    double stop() {
        (void)hipEventRecord(c->ev1, c->stream);
        (void)vidc::vidc_event_wait(c->ev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        return ms;
    }
    // the same in two halves, for a caller that synchronises the stream anyway (one host wait instead of two)
    void mark() { (void)hipEventRecord(c->ev1, c->stream); }
    double elapsed() {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        return ms;
    }
};

int check_status(const std::vector<uint32_t> &status, const char *what) {
    for (size_t l = 0; l < status.size(); l++) {
        switch (status[l]) {
            case VIDC_ST_OK: break;
            case VIDC_ST_DOMAIN:
                set_error("%s: list %zu holds an id outside [0, 2^31) (reference: int max_id, "
                          "custom_invlists_impl.cpp:163)", what, l);
                return VIDC_ERR_DOMAIN;
            case VIDC_ST_OVERFLOW:
                set_error("%s: list %zu overflowed its ANS word arena", what, l);
                return VIDC_ERR_OVERFLOW;
            case VIDC_ST_MT:
                set_error("%s: list %zu needed more than %d mt19937 underflow words", what, l, VIDC_MT_TABLE);
                return VIDC_ERR_OVERFLOW;
            default:
                set_error("%s: list %zu finished with status %u", what, l, status[l]);
                return VIDC_ERR_INVALID;
        }
    }
    return VIDC_OK;
}

// Host loops over the lists of a call (classification, work-list sorts): with 10^6 lists they are the host's
// critical path between the prepass and the first encode kernel (7 ms single-threaded), so they are split over a few
// threads.  f(begin, end, part) for `parts` contiguous ranges; small calls stay on the calling thread.
constexpr uint64_t PAR_MIN_LISTS = 131072;
inline unsigned par_parts(uint64_t n) {
    if (n < PAR_MIN_LISTS) return 1;
    const char *e = getenv("VIDC_HOST_THREADS");  // test hook: 1 = single-threaded host loops
    const unsigned cap = e && e[0] ? (unsigned)std::max(1, atoi(e)) : 8u;
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(cap, hw ? hw : 1u));
}
template <class F>
void par_ranges(uint64_t n, unsigned parts, F &&f) {
    if (parts <= 1 || n == 0) { f((uint64_t)0, n, 0u); return; }
    std::vector<std::thread> th;
    const uint64_t per = (n + parts - 1) / parts;
    for (unsigned t = 1; t < parts; t++) {
        const uint64_t a = std::min<uint64_t>(n, t * per), b = std::min<uint64_t>(n, a + per);
        th.emplace_back([&f, a, b, t] { f(a, b, t); });
    }
    f((uint64_t)0, std::min<uint64_t>(n, per), 0u);
    for (auto &x : th) x.join();
}

// lists sorted longest first: the hardware dispatches workgroups in order, so this is LPT scheduling.
// Counting sort by length (lengths are bounded by VIDC_ROC_MAX_LIST): O(n), stable.
void sort_desc(std::vector<uint32_t> &wl, const std::vector<uint64_t> &offsets) {
    if (wl.size() < 2) return;
    uint64_t maxlen = 0, prev = ~0ull;
    bool sorted = true;  // equal-sized lists, or an index whose lists come longest first: nothing to do
    for (uint32_t l : wl) {
        const uint64_t n = offsets[l + 1] - offsets[l];
        maxlen = std::max<uint64_t>(maxlen, n);
        sorted &= n <= prev;
        prev = n;
    }
    if (sorted) return;
    if (maxlen > (1u << 22)) {  // not reachable for ROC lists; keep a comparison sort for safety
        std::stable_sort(wl.begin(), wl.end(), [&](uint32_t a, uint32_t b) {
            return offsets[a + 1] - offsets[a] > offsets[b + 1] - offsets[b];
        });
        return;
    }
    std::vector<uint32_t> out = vec_pool<uint32_t>().take(wl.size());
    out.resize(wl.size());
    struct Back { std::vector<uint32_t> &v; ~Back() { vec_pool<uint32_t>().give(std::move(v)); } } out_back{out};  // (holds wl's old array after the swap)
    const unsigned parts = maxlen <= 65536 ? par_parts(wl.size()) : 1;
    if (parts > 1) {  // the same stable counting sort, histogram and scatter per contiguous part of the work list
        const size_t nb = maxlen + 1;
        std::vector<uint32_t> hist((size_t)parts * nb, 0);
        par_ranges(wl.size(), parts, [&](uint64_t a, uint64_t b, unsigned t) {
            uint32_t *h = &hist[(size_t)t * nb];
            for (uint64_t i = a; i < b; i++) h[maxlen - (offsets[wl[i] + 1] - offsets[wl[i]])]++;
        });
        uint32_t run = 0;  // bucket-major, part-minor: equal lengths keep their work-list order
        for (size_t k = 0; k < nb; k++)
            for (unsigned t = 0; t < parts; t++) {
                const uint32_t c = hist[(size_t)t * nb + k];
                hist[(size_t)t * nb + k] = run;
                run += c;
            }
        par_ranges(wl.size(), parts, [&](uint64_t a, uint64_t b, unsigned t) {
            uint32_t *h = &hist[(size_t)t * nb];
            for (uint64_t i = a; i < b; i++) out[h[maxlen - (offsets[wl[i] + 1] - offsets[wl[i]])]++] = wl[i];
        });
        wl.swap(out);
        return;
    }
    std::vector<uint32_t> start(maxlen + 2, 0);
    for (uint32_t l : wl) start[maxlen - (offsets[l + 1] - offsets[l]) + 1]++;  // bucket 0 = longest
    for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
    for (uint32_t l : wl) out[start[maxlen - (offsets[l + 1] - offsets[l])]++] = l;
    wl.swap(out);
}

// End of an encode call: status check + word offsets + sizes + compaction, all on the device.  Two halves so that the first
// can be queued right behind the encode kernels -- while the host still plans the decode -- and ONE synchronisation serves the
// kernels, the status summary (errors / lists waiting for the sorting pass) and the size read-back:
//   finish_enqueue: status summary + totals into `t`; the exclusive scan of the word counts (IVF lists: k_roc_tail)
//   finish_complete (after a synchronisation of the stream): checks, allocation of the stream, compaction (graph rows: with the scans
//                   of the word and edge counts inside, k_roc_rows_compact), final synchronisation
struct EncodeTail {
    Scratch s_sum, s_tmp, s_tmp2, s_state;
    bool state_zeroed = false;  // the tile states of k_roc_tail were cleared ahead of the encode kernels
    const uint32_t *d_sizes = nullptr;  // graph rows: the edge counts, for the scan inside the compaction
    Pinned tail;
    unsigned long long *t = nullptr;  // [0..3] status summary (first bad list, -, retries, pending sorts), [4] total words, [5] ntotal, [6] non-empty rows
};
// (ahead of the encode kernels: the tile states of the tail kernel, so that clearing them is not part of the tail)
inline uint32_t rows_tiles(uint64_t nlist) { return (uint32_t)((nlist + 63) / 64); }  // tiles of k_roc_rows_compact
inline uint32_t rows_chunks(uint64_t nlist) { return (uint32_t)((nlist + VIDC_ROWS_CHUNK - 1) / VIDC_ROWS_CHUNK); }  // workgroups of k_roc_rows_totals
inline size_t rows_state_bytes(uint64_t nlist) {  // tile sums (words, edges) | chunk sums | counter | chunk prefixes
    return ((size_t)rows_tiles(nlist) * 2 + (size_t)rows_chunks(nlist) * 10 + 1) * 8;
}
int finish_prepare(vidc_ctx *ctx, const vidc_roc *r, EncodeTail &e) {
    if (!r->nlist) return VIDC_OK;
    // graph rows: [word scan | edge scan | tile counter | sum[8]] of k_roc_rows_totals / k_roc_rows_compact
    const size_t bytes = r->rows ? rows_state_bytes(r->nlist) : ((size_t)(r->nlist / VIDC_TAIL_TILE + 1u) + 10) * 8;
    VIDC_TRY(e.s_state.get(ctx, bytes));
    VIDC_HIP(hipMemsetAsync(e.s_state.p, 0, bytes, ctx->stream));
    e.state_zeroed = true;
    return VIDC_OK;
}
int finish_enqueue(vidc_ctx *ctx, vidc_roc *r, EncodeTail &e, Scratch &d_status_buf, const uint32_t *d_sizes) {
    const uint64_t nlist = r->nlist;
    VIDC_TRY(e.tail.get(ctx, 128));
    unsigned long long *t = e.t = e.tail.as<unsigned long long>();
    t[0] = ~0ull; t[1] = 0; t[2] = 0; t[3] = 0; t[4] = 0; t[5] = 0; t[6] = 0; t[7] = 0;
    VIDC_TRY(r->d_word_off.alloc(nlist + 1, ctx->dpool));
    if (nlist && r->rows) {  // totals only: the scans ride on the compaction (k_roc_rows_compact)
        const uint32_t nt = rows_tiles(nlist);
        if (!e.state_zeroed) {
            VIDC_TRY(e.s_state.get(ctx, rows_state_bytes(nlist)));
            VIDC_HIP(hipMemsetAsync(e.s_state.p, 0, rows_state_bytes(nlist), ctx->stream));
        }
        e.state_zeroed = false;  // (used up)
        e.d_sizes = d_sizes;
        VIDC_TRY(r->d_offsets.alloc(nlist + 1, ctx->dpool));
        hipLaunchKernelGGL(k_roc_rows_totals, dim3(rows_chunks(nlist)), dim3(256), 0, ctx->stream, r->d_nwords.p, d_sizes, d_status_buf.as<uint32_t>(),
                           (uint32_t)nlist, e.s_state.as<unsigned long long>() + 2ull * nt, t);
        VIDC_HIP(hipGetLastError());
        return VIDC_OK;
    }
    if (nlist) {  // summary + word offsets + totals in one launch; its last workgroup stores the results into `t`
        const uint32_t ntiles = (uint32_t)(nlist / VIDC_TAIL_TILE + 1u);
        const size_t state_bytes = ((size_t)ntiles + 10) * 8;
        if (!e.state_zeroed) {
            VIDC_TRY(e.s_state.get(ctx, state_bytes));
            VIDC_HIP(hipMemsetAsync(e.s_state.p, 0, state_bytes, ctx->stream));
        }
        e.state_zeroed = false;  // (used up)
        hipLaunchKernelGGL(k_roc_tail, dim3(ntiles), dim3(256), 0, ctx->stream, r->d_nwords.p, (uint32_t)nlist, r->d_word_off.p,
                           d_status_buf.as<uint32_t>(), e.s_state.as<unsigned long long>(), t);
        VIDC_HIP(hipGetLastError());
        return VIDC_OK;
    }
    // no lists: the offsets of an empty object
    VIDC_HIP(hipMemsetAsync(r->d_word_off.p, 0, 8, ctx->stream));
    if (r->rows) { VIDC_TRY(r->d_offsets.alloc(1, ctx->dpool)); VIDC_HIP(hipMemsetAsync(r->d_offsets.p, 0, 8, ctx->stream)); }
    return VIDC_OK;
}
int finish_complete(vidc_ctx *ctx, vidc_roc *r, EncodeTail &e, const uint32_t *d_arena, uint32_t arena_stride, Scratch &d_status_buf,
                    uint64_t nonempty_lists, double &kernel_ms) {
    const uint64_t nlist = r->nlist;
    const unsigned long long *t = e.t;
    if (t[0] != ~0ull) {
        std::vector<uint32_t> status;
        VIDC_TRY(download(ctx, status, d_status_buf.as<uint32_t>(), nlist));
        VIDC_TRY(check_status(status, "roc encode"));
    }
    r->total_words = t[4];
    if (r->rows) { r->ntotal = t[5]; nonempty_lists = t[6]; }
    // ANSState::size() = 8 + 4 * words per non-empty list; "let's pretend no memory is used" for empty ones
    // (custom_invlists_impl.cpp:196-206).  Empty lists have no words.
    r->compressed_bytes = 8ull * nonempty_lists + 4ull * r->total_words;
    VIDC_TRY(r->d_words.alloc(r->total_words + 16, ctx->dpool));  // + padding: the look-ahead of the lane / row decoders reads up to 15 words past the last stream
    if (nlist) {
        EventTimer tm(ctx);
        const uint64_t *d_off = r->rows ? (const uint64_t *)nullptr : (const uint64_t *)r->d_offsets.p;
        if (r->rows) {  // 64 rows per wavefront; word offsets and CSR offsets are scanned on the way
            const uint32_t nt = rows_tiles(nlist);
            // Every wavefront of the launch resident at once (what the runtime says fits, not a guess: a tile waits for the tiles in
            // front of it in its chunk), and a multiple of 64 of them: the 64 tiles of a chunk are then 64 consecutive workgroups
            // in the same trip of their loops, so a tile only ever waits for workgroups with smaller numbers.
            static const int occ = [] {
                int o = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, k_roc_rows_compact, 64, 0) != hipSuccess || o < 1) o = 1;
                return o;
            }();
            const uint32_t cap = std::max<uint32_t>(64u, (uint32_t)ctx->num_cu * (uint32_t)occ / 64u * 64u);
            const uint32_t grid = std::min<uint32_t>(nt, cap);
            hipLaunchKernelGGL(k_roc_rows_compact, dim3(grid), dim3(64), 0, ctx->stream, d_arena, arena_stride, r->d_nwords.p, e.d_sizes,
                               (uint32_t)nlist, e.s_state.as<unsigned long long>(), nt,
                               e.s_state.as<unsigned long long>() + 2ull * nt + 8ull * rows_chunks(nlist) + 1, r->d_word_off.p, r->d_offsets.p,
                               r->d_words.p);
        } else if (r->total_words / nlist < 64) {  // tiny lists (a few dozen words each): one wavefront per 64 lists
            const uint32_t grid = (uint32_t)std::min<uint64_t>((nlist + 63) / 64, (uint64_t)ctx->num_cu * 64);
            hipLaunchKernelGGL(k_roc_compact_groups, dim3(grid), dim3(64), 0, ctx->stream, d_arena, d_off, arena_stride,
                               r->d_word_off.p, r->d_words.p, (uint32_t)nlist);
        } else if (r->total_words / nlist < 200 ){  // a wavefront per four lists (k_roc_compact_waves)
            const uint32_t grid = (uint32_t)std::min<uint64_t>((nlist + 15) / 16, (uint64_t)ctx->num_cu * 8);
            hipLaunchKernelGGL(k_roc_compact_waves, dim3(grid), dim3(256), 0, ctx->stream, d_arena, d_off, arena_stride,
                               r->d_word_off.p, r->d_words.p, (uint32_t)nlist);
        } else {
            const uint32_t grid = (uint32_t)std::min<uint64_t>(nlist, (uint64_t)ctx->num_cu * 16);
            hipLaunchKernelGGL(k_roc_compact, dim3(grid), dim3(256), 0, ctx->stream, d_arena, d_off, arena_stride,
                               r->d_word_off.p, r->d_words.p, (uint32_t)nlist);
        }
        VIDC_HIP(hipGetLastError());
        double ms = tm.stop();  // (synchronises: the call's last wait)
        ctx->phase_ms[VIDC_PHASE_ROC_COMPACT] = ms;
        kernel_ms += ms;
    } else {
        VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    }
    return VIDC_OK;
}

// ---- decode planning: work items grouped by kernel class, each with private scratch
// general decoder classes by LDS footprint of the fine prefix rows (2^fb u32, fb = lg(n) - 3): occupancy, not
// arithmetic, bounds the general decoder when a batch has many mid-size lists
// (DC_LANE .. the last class: kernels that may hand a list back with VIDC_ST_RETRY)
enum DecClass { DC_TINY = 0, DC_U18, DC_U20, DC_GSMALL, DC_G8K, DC_G16K, DC_GMID, DC_GHUGE, DC_LANE, DC_LANE64, DC_LANE128, DC_B2, DC_B2T, DC_B2S, DC_B2L, DC_B2M,
                DC_GRP0, DC_GRP2, DC_GRP3, DC_GRP4,   // row-per-list decoder by bucket bits 8 + F (roc_grp.h)
                DC_LANEP, DC_LANEQ, DC_COUNT };        // 257..512 / 513..1024 ids on a pair / quad of lanes, ids in registers (k_roc_decode_lane_reg<.., 2 / 4>)
constexpr uint64_t B2_MIN_LIST = 4096;
#define VIDC_LANE_ALIGN_DEFAULT 16u        // bucket rows of the lane decoders on 64-byte boundaries (S2 decode 70.9 -> 69.8 ms: DESIGN section 13)
constexpr uint32_t B2_MASK_MIN_CHAINS = 512;  // k_roc_decode_b2 launches of this many chains size their row loads by the member count
constexpr size_t B2_CAP = 1024;      // chains of k_roc_decode_b2 per call (four per CU of an MI355X: 1 MiB of member rows each)
constexpr size_t B2_TOP_CAP = 5120;  // ... of a call with more long chains than that: every list beyond 16 384 ids (comment at the planner)

struct DecPlan {
    std::vector<uint32_t> wl;        // list numbers, grouped by class, longest first inside a class
    std::vector<uint32_t> item;      // index of each work item in the caller's request
    size_t count[DC_COUNT] = {};
    uint64_t sum_n[DC_COUNT] = {}, max_n[DC_COUNT] = {};
    std::vector<uint64_t> scratch_off, slots_off;
    uint64_t scratch_words = 0, slots_words = 0;
    bool lean = false;               // graph rows, lane decoder: item k == request k, no scratch / offset arrays
    bool tiny_lane = false;          // DC_TINY items run on the lane-per-list kernel
    bool gsmall_lrows = false;       // DC_GSMALL items keep their member rows in LDS (33 KiB each: only for few lists)
    uint32_t lane_align = 4;         // slots a bucket row of the lane decoders is aligned / padded to (roc_lane_cap_nb)
    uint64_t implicit = 0;           // lean plan without a work list: items are rows 0..implicit-1
    const uint32_t *order_dev = nullptr;  // ... or every row of the object in this device-resident order, each to its OWN output row
    bool order_build = false;        // ... which this call builds first (k_rows_order_*, inside the call's timed region)
    DecPlan() = default;
    DecPlan(const DecPlan &) = default;
    DecPlan(DecPlan &&) = default;
    DecPlan &operator=(const DecPlan &) = default;
    DecPlan &operator=(DecPlan &&) = default;
    ~DecPlan() {
        vec_pool<uint32_t>().give(std::move(wl)); vec_pool<uint32_t>().give(std::move(item));
        vec_pool<uint64_t>().give(std::move(scratch_off)); vec_pool<uint64_t>().give(std::move(slots_off));
    }
};

}  // namespace

// device copy of a plan, kept with the compressed object for repeated decode_all calls
struct DecPlanCache {
    DecPlan plan;
    bool wide = false;  // the stream mode the plan was built for (a context of the other mode builds its own: ADVICE round 4)
    DevBuf<uint32_t> d_wl;
    DevBuf<uint64_t> d_scratch_off, d_slots_off;
};

namespace {

// decode_all plan of a freshly encoded object (defined with the planner below)
std::shared_ptr<DecPlanCache> plan_ahead_build(const vidc_roc *r, bool wide);
constexpr uint64_t PLAN_AHEAD_MIN_LISTS = 4096;

int encode_impl(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids, bool rows, uint64_t N,
                uint32_t K, const int32_t *d_rows, int precision_mode, uint32_t flags, vidc_roc **out) {
    if (!ctx || !out) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (precision_mode < VIDC_PREC_EXACT || precision_mode > 32) {
        set_error("precision_mode %d unsupported (fixed precision must be 0..32)", precision_mode);
        return VIDC_ERR_INVALID;
    }
    if (nlist >= 0xffffffffull || N >= 0xffffffffull) { set_error("too many lists"); return VIDC_ERR_INVALID; }
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_roc> r(new vidc_roc());
    r->device = ctx->device;
    r->rows = rows;
    r->K = K;
    r->nlist = rows ? N : nlist;
    nlist = r->nlist;
    double kernel_ms = 0;

    HostTrace tr("roc encode");
    // work lists: tiny (n <= 64), universe-bitmap kernels (ids < 2^18 / 2^20), general kernels by bitmap depth,
    // lane-per-list kernels
    std::vector<uint32_t> wl_tiny, wl_u18, wl_u20, wl_c1, wl_c2, wl_c3, wl_l4, wl_l16, wl_l64, wl_r2, wl_g2, wl_g3;
    const bool want_perm = (flags & VIDC_ROC_WANT_PERM) && !rows;
    const uint64_t ntotal_in = rows ? N * K : (nlist ? offsets[nlist] : 0);
    const bool f_general = force_general();  // getenv once, not per list
    const LanePolicy lpol = f_general ? LANE_NEVER : lane_policy();
    bool use_lane = false, use_lane64 = false, use_lane_tiny = false, use_grp = false;
    const GrpPolicy gpol = grp_policy();
    // persistent outputs
    VIDC_TRY(r->d_heads.alloc(nlist, ctx->dpool)); VIDC_TRY(r->d_prec.alloc(nlist, ctx->dpool));
    VIDC_TRY(r->d_nwords.alloc(nlist, ctx->dpool)); VIDC_TRY(r->d_draws.alloc(nlist, ctx->dpool));
    if (want_perm) VIDC_TRY(r->d_perm.alloc(ntotal_in ? ntotal_in : 1, ctx->dpool));
    tr.mark("persistent allocs");
    Scratch s_arena, s_status, s_sizes, s_sid, s_wl, s_maxid, s_flags, s_sum;
    EncodeTail tail;  // status summary + sizes of the call, queued behind the encode kernels
    bool light_prepass = false, pre_deferred = false;
    bool wl_sorted[10] = {};  // work-list classes found longest-first while they were filled
    Pinned h_wl, h_pre, h_off, h_plan;
    bool plan_on_device = false;
    struct SyncOnExit {  // an early return while the offsets upload / the deferred prepass copy is in flight must not release their staging blocks
                        // (declared behind them: destroyed first)
        vidc_ctx *c;
        bool armed = false;
        ~SyncOnExit() { if (armed) (void)vidc::vidc_stream_wait(c->stream); }
    } pre_guard{ctx};
    // (graph rows: slots of whole 64-byte sectors -- a row's ~100 bytes of stream then lie in two sectors, not in two and a half)
    const uint32_t arena_stride = rows ? (uint32_t)((arena_words_for(K) + 15u) & ~15ull) : 0u;
    uint64_t arena_words = 0, nonempty = 0, ntiny = 0;
    if (rows) {
        if (K == 0 || K > TINY_MAX) {
            set_error("graph rows: K=%u unsupported (1..64)", K);
            return VIDC_ERR_UNSUPPORTED;
        }
        arena_words = roc_arena_at(nullptr, arena_stride, nlist);
        ntiny = nlist;  // every row, in order: no work list array (kernels take l = work item)
        use_lane_tiny = lane_wanted(lpol, nlist, LANE_MIN_TINY);
        VIDC_TRY(s_sizes.get(ctx, nlist * 4));
    } else {
        if (!offsets) return VIDC_ERR_INVALID;
        // the object's host copy of the offsets: on a helper thread for calls of 10^5 lists and more (8 MB at 10^6 lists: 0.6 ms
        // of the 1.3 ms this phase took on S2), next to the validation pass and the staging copy below; joined before the
        // classification, the first reader
        std::thread offsets_copy;
        struct JoinOnExit { std::thread &t; ~JoinOnExit() { if (t.joinable()) t.join(); } } offsets_copy_guard{offsets_copy};
        r->offsets = vec_pool<uint64_t>().take(nlist + 1);
        if (par_parts(nlist) > 1) offsets_copy = std::thread([&] { r->offsets.assign(offsets, offsets + nlist + 1); });
        else r->offsets.assign(offsets, offsets + nlist + 1);
        r->offsets_host = true;
        tr.mark("offsets: object's copy");
        // the offsets: validation, longest list, non-empty lists and the class sizes the kernel-family policies look at
        bool any_big = false, all_desc = false;
        uint64_t max_n = 0;
        uint64_t n_tiny_lists = 0, n_mid_lists = 0, n_mid64_lists = 0, n_grp_lists = 0;
        // (no synchronisation of its own for the upload below: the staging block lives until the call returns; an early error return
        // before the call's first wait synchronises through the guard)
        VIDC_TRY(h_off.get(ctx, (nlist + 1) * 8));
        {
            const unsigned parts = par_parts(nlist);
            // (first the cheap facts -- extremes, order, validity as ONE flag --; the class sizes only need their own pass when the
            // lengths straddle a class boundary: an index of equal-sized lists, or a graph, is classified by its extremes)
            struct Acc { uint64_t nonempty = 0, max_n = 0, min_n = ~0ull, prev = ~0ull, c0 = 0, c1 = 0, c2 = 0, cg = 0; bool desc = true, bad = false; };
            std::vector<Acc> acc(parts);
            par_ranges(nlist, parts, [&](uint64_t la, uint64_t lb, unsigned t) {
                Acc x;
                if (parts > 1) {  // (threaded calls count the classes in the same pass: every pass costs a round of thread starts)
                    for (uint64_t l = la; l < lb; l++) {
                        const uint64_t n = offsets[l + 1] - offsets[l];
                        x.bad |= n > VIDC_ROC_MAX_LIST;
                        x.nonempty += n != 0;
                        x.max_n = std::max(x.max_n, n);
                        x.min_n = std::min(x.min_n, n);
                        x.c0 += n <= TINY_MAX;
                        x.c1 += n > TINY_MAX && n <= VIDC_LANE_MAX;
                        x.c2 += n > VIDC_LANE_MAX && n <= VIDC_LANE_MAX64;
                        x.cg += n >= gpol.min_n && n <= gpol.max_n;
                    }
                    x.desc = false;
                } else {
                    // (one thread: the staging copy for the upload rides along; four lists per instruction where the host can)
                    OffsetsPass op;
                    offsets_pass(offsets + la, lb - la, h_off.as<uint64_t>() + la, op);
                    x.bad = op.bad; x.nonempty = op.nonempty; x.max_n = op.max_n; x.min_n = op.min_n; x.desc = op.desc; x.prev = op.prev;
                }
                acc[t] = x;
            });
            uint64_t min_n = ~0ull;
            bool bad = false;
            for (unsigned t = 0; t < parts; t++) {
                bad |= acc[t].bad;
                nonempty += acc[t].nonempty;
                max_n = std::max(max_n, acc[t].max_n);
                min_n = std::min(min_n, acc[t].min_n);
            }
            if (bad)  // the first offending list, as a single pass would report it
                for (uint64_t l = 0; l < nlist; l++) {
                    if (offsets[l + 1] < offsets[l]) {
                        set_error("offsets not monotone at list %llu", (unsigned long long)l);
                        return VIDC_ERR_INVALID;
                    }
                    if (offsets[l + 1] - offsets[l] > VIDC_ROC_MAX_LIST) {
                        set_error("list %llu has %llu ids; ROC lists are limited to %u (the reference codec is only "
                                  "lossless up to 65536, SURVEY 8a-Q2)", (unsigned long long)l,
                                  (unsigned long long)(offsets[l + 1] - offsets[l]), VIDC_ROC_MAX_LIST);
                        return VIDC_ERR_DOMAIN;
                    }
                }
            auto in_one = [&](uint64_t lo, uint64_t hi) { return nlist && min_n >= lo && max_n <= hi; };
            if (parts > 1) {
                for (unsigned t = 0; t < parts; t++) { n_tiny_lists += acc[t].c0; n_mid_lists += acc[t].c1; n_mid64_lists += acc[t].c2; n_grp_lists += acc[t].cg; }
            } else
            if (in_one(0, TINY_MAX) || in_one(TINY_MAX + 1, VIDC_LANE_MAX) || in_one(VIDC_LANE_MAX + 1, VIDC_LANE_MAX64) ||
                (nlist && min_n > VIDC_LANE_MAX64)) {
                n_tiny_lists = max_n <= TINY_MAX ? nlist : 0;
                n_mid_lists = in_one(TINY_MAX + 1, VIDC_LANE_MAX) ? nlist : 0;
                n_mid64_lists = in_one(VIDC_LANE_MAX + 1, VIDC_LANE_MAX64) ? nlist : 0;
            } else {
                std::vector<uint64_t> cnt(3 * (size_t)parts, 0);
                par_ranges(nlist, parts, [&](uint64_t la, uint64_t lb, unsigned t) {
                    uint64_t c0 = 0, c1 = 0, c2 = 0;
                    for (uint64_t l = la; l < lb; l++) {
                        const uint64_t n = offsets[l + 1] - offsets[l];
                        c0 += n <= TINY_MAX;
                        c1 += n > TINY_MAX && n <= VIDC_LANE_MAX;
                        c2 += n > VIDC_LANE_MAX && n <= VIDC_LANE_MAX64;
                    }
                    cnt[3 * t] = c0; cnt[3 * t + 1] = c1; cnt[3 * t + 2] = c2;
                });
                for (unsigned t = 0; t < parts; t++) { n_tiny_lists += cnt[3 * t]; n_mid_lists += cnt[3 * t + 1]; n_mid64_lists += cnt[3 * t + 2]; }
            }
            if (parts == 1 && max_n >= gpol.min_n && min_n <= gpol.max_n) {  // (lists in reach of the row-per-list kernels)
                std::vector<uint64_t> cnt(parts, 0);
                par_ranges(nlist, parts, [&](uint64_t la, uint64_t lb, unsigned t) {
                    uint64_t c = 0;
                    for (uint64_t l = la; l < lb; l++) {
                        const uint64_t n = offsets[l + 1] - offsets[l];
                        c += n >= gpol.min_n && n <= gpol.max_n;
                    }
                    cnt[t] = c;
                });
                for (unsigned t = 0; t < parts; t++) n_grp_lists += cnt[t];
            }
            any_big = max_n > TINY_MAX;
            all_desc = parts == 1 && acc[0].desc;
        }
        tr.mark("offsets: pass + staging");
        arena_words = roc_arena_at(offsets, 0, nlist);
        r->ntotal = offsets[nlist];
        VIDC_TRY(r->d_offsets.alloc(nlist + 1, ctx->dpool));
        {
            const unsigned cparts = par_parts(nlist);
            if (cparts > 1 || nlist == 0)  // (a single-threaded pass above has copied them already)
                par_ranges(nlist + 1, cparts, [&](uint64_t a0, uint64_t b0, unsigned) {
                    std::memcpy(h_off.as<uint64_t>() + a0, offsets + a0, (b0 - a0) * 8);
                });
        }
        if (offsets_copy.joinable()) offsets_copy.join();
        tr.mark("offsets: blocks");
        VIDC_HIP(hipMemcpyAsync(r->d_offsets.p, h_off.p, (nlist + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        pre_guard.armed = true;
        tr.mark("offsets: upload call");
        // The bitmap kernels own a whole CU's LDS (2^20-bit universe): latency-optimal for long lists, but only
        // num_cu lists in flight.  With many lists, short ones go to the high-occupancy kernels.
        const uint64_t u_min = U_MIN_LIST;
        use_lane = lane_wanted(lpol, n_mid_lists, LANE_MIN_LISTS);
        use_lane64 = lane_wanted(lpol, n_mid64_lists, LANE_MIN_LISTS64);
        use_lane_tiny = lane_wanted(lpol, n_tiny_lists, LANE_MIN_TINY);
        // (the octaves of a call must overlap: without spare hardware queues they would run one after the other)
        use_grp = n_grp_lists && n_grp_lists >= gpol.min_lists && (ctx->wide || gpol.min_lists == 0);
        // classification prepass (one workgroup per list): max id -> precision, sortedness, domain
        const uint32_t *maxid = nullptr, *pflags = nullptr;
        if (any_big) {
            VIDC_TRY(s_maxid.get(ctx, nlist * 4));
            VIDC_TRY(s_flags.get(ctx, nlist * 4));
            VIDC_TRY(h_pre.get(ctx, nlist * 8));
            EventTimer t(ctx);
            // Every kernel that takes a list streams it anyway and checks what the full prepass would (ids inside
            // [0, 2^31), ascending order where it matters, ids that fit the precision; the lane-per-list encoders since
            // round 3: each sampled id against its left neighbour): the prepass only looks at the LAST id of each list -- the maximum of an ascending
            // list -- instead of re-reading all of them (8 bytes per id).  A list that turns out not to be ascending
            // comes back with VIDC_ST_PENDING_SORT and takes the sorting second pass like a multiset does.
            light_prepass = !old_u_kernels() && !std::getenv("VIDC_FULL_PREPASS");
            // No list long enough for the bitmap kernels (the only classes chosen by the width of a list's ids): the host
            // classifies by length alone and reads the maxima back when it needs them -- for the decode plan, while the
            // encode kernels run -- instead of waiting for the prepass here (0.1 ms per call at 65 536 lists).  A last id
            // outside the domain is then reported by the kernel that takes the list, like one in the middle of a list.
            pre_deferred = light_prepass && max_n < u_min;
            if (pre_deferred) {
                VIDC_HIP(hipEventRecord(ctx->ev_pre[0], ctx->stream));
                hipLaunchKernelGGL(k_roc_prepass_last, dim3((uint32_t)((nlist + 255) / 256)), dim3(256), 0, ctx->stream, d_ids,
                                   r->d_offsets.p, (uint32_t)nlist, precision_mode, s_maxid.as<uint32_t>(),
                                   s_flags.as<uint32_t>(), r->d_prec.p);
                VIDC_HIP(hipGetLastError());
                VIDC_HIP(hipEventRecord(ctx->ev_pre[1], ctx->stream));
                VIDC_HIP(hipMemcpyAsync(h_pre.p, s_maxid.p, nlist * 4, hipMemcpyDeviceToHost, ctx->stream));
                VIDC_HIP(hipEventRecord(ctx->ev_pre[2], ctx->stream));
                pre_guard.armed = true;
                r->prec = vec_pool<uint32_t>().take(nlist);
                r->prec.resize(nlist);
                tr.mark("prepass kernel (read back later)");
            } else {
                if (light_prepass)
                    hipLaunchKernelGGL(k_roc_prepass_last, dim3((uint32_t)((nlist + 255) / 256)), dim3(256), 0, ctx->stream, d_ids,
                                       r->d_offsets.p, (uint32_t)nlist, precision_mode, s_maxid.as<uint32_t>(),
                                       s_flags.as<uint32_t>(), r->d_prec.p);
                else
                    hipLaunchKernelGGL(k_roc_prepass, dim3((uint32_t)std::min<uint64_t>(nlist, (uint64_t)ctx->num_cu * 32)),
                                       dim3(256), 0, ctx->stream, d_ids, r->d_offsets.p, (uint32_t)nlist, precision_mode,
                                       s_maxid.as<uint32_t>(), s_flags.as<uint32_t>(), r->d_prec.p);
                VIDC_HIP(hipGetLastError());
                t.mark();
                VIDC_HIP(hipMemcpyAsync(h_pre.p, s_maxid.p, nlist * 4, hipMemcpyDeviceToHost, ctx->stream));
                VIDC_HIP(hipMemcpyAsync(h_pre.as<uint32_t>() + nlist, s_flags.p, nlist * 4, hipMemcpyDeviceToHost, ctx->stream));
                VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
                pre_guard.armed = false;
                kernel_ms += t.elapsed();
                maxid = h_pre.as<uint32_t>();
                pflags = maxid + nlist;
                tr.mark("prepass kernel + d2h");
                // what the decode planner needs later (kernel class, bucket geometry) is known right here
                // (the precisions are filled in by the classification loop below: one pass over the lists instead of two)
                r->prec = vec_pool<uint32_t>().take(nlist);
                r->prec.resize(nlist);
                r->umax = vec_pool<uint32_t>().take(nlist);
                r->umax.assign(maxid, maxid + nlist);
            }
        } else {
            r->prec.assign(nlist, 0);  // tiny lists only: the planner does not look at their precision
        }
        {
            // per-thread work lists (contiguous list ranges), concatenated in range order: the same lists in the
            // same order as a single pass
            enum { W_TINY = 0, W_U18, W_U20, W_C1, W_C2, W_C3, W_L4, W_L16, W_L64, W_G2, W_G3, W_COUNT };
            const unsigned parts = par_parts(nlist);
            std::vector<uint32_t> *dst[W_COUNT] = {&wl_tiny, &wl_u18, &wl_u20, &wl_c1, &wl_c2, &wl_c3, &wl_l4, &wl_l16, &wl_l64, &wl_g2, &wl_g3};
            // Without per-list maxima (the prepass is read back later: no list is long enough for the bitmap kernels, the only
            // classes chosen by the width of the ids) the class of a list is a function of its LENGTH: the lists are ordered by
            // length once -- longest first, a stable counting sort, or nothing at all for an index of equal-sized lists -- and
            // every class is a run of that order, copied out between two binary searches.  The per-list loop below with its
            // dozen push_back targets and the per-class sorts behind it were 0.13 + 0.02 ms of a 65 536-list call.
            const bool by_length = !maxid && !pflags && parts == 1 && !f_general && !env_on("VIDC_NO_LENGTH_CLASSES");
            if (by_length) {
                std::vector<uint32_t> order = vec_pool<uint32_t>().take(nlist);
                order.resize(nlist);
                if (all_desc) {
                    std::iota(order.begin(), order.end(), 0u);
                } else {
                    std::vector<uint32_t> start(max_n + 2, 0);
                    for (uint64_t l = 0; l < nlist; l++) start[max_n - (offsets[l + 1] - offsets[l]) + 1]++;  // bucket 0 = longest
                    for (size_t i = 1; i < start.size(); i++) start[i] += start[i - 1];
                    for (uint64_t l = 0; l < nlist; l++) order[start[max_n - (offsets[l + 1] - offsets[l])]++] = (uint32_t)l;
                }
                auto len_at = [&](size_t i) { return offsets[order[i] + 1] - offsets[order[i]]; };
                auto first_le = [&](uint64_t bound) {  // first position of the order whose list has at most `bound` ids
                    size_t lo = 0, hi = nlist;
                    while (lo < hi) { const size_t mid = (lo + hi) / 2; if (len_at(mid) > bound) lo = mid + 1; else hi = mid; }
                    return lo;
                };
                auto class_of = [&](uint64_t n) -> int {  // the decisions of the per-list loop below for a sorted list of n ids
                    if (n <= TINY_MAX) return W_TINY;
                    const bool lane_ok = (use_lane && n <= VIDC_LANE_MAX) || (use_lane64 && n > VIDC_LANE_MAX && n <= VIDC_LANE_MAX64);
                    const bool grp_ok = use_grp && n >= gpol.min_n && n <= gpol.max_n;
                    return grp_ok ? (n <= VIDC_GRP_LEV2_MAX ? W_G2 : W_G3)
                           : (lane_ok && n <= 256) ? W_L4
                           : (lane_ok && n <= VIDC_LANE_MAX) ? W_L16
                           : lane_ok ? W_L64
                           : n <= 4096 ? W_C1
                           : n <= 32768 ? W_C2 : W_C3;
                };
                // upper ends of the length intervals on which class_of is constant, longest first
                uint64_t cuts[] = {VIDC_ROC_MAX_LIST, gpol.max_n, 32768, VIDC_GRP_LEV2_MAX, 4096, VIDC_LANE_MAX64, gpol.min_n - 1, VIDC_LANE_MAX, 256, TINY_MAX};
                std::sort(std::begin(cuts), std::end(cuts), std::greater<uint64_t>());
                size_t pos = 0;
                for (size_t c = 0; c < sizeof(cuts) / sizeof(cuts[0]) && pos < nlist; c++) {
                    const uint64_t hi = cuts[c], lo = c + 1 < sizeof(cuts) / sizeof(cuts[0]) ? cuts[c + 1] : 0;  // interval (lo, hi]
                    if (hi == lo) continue;
                    const size_t end = lo ? first_le(lo) : nlist;  // (the last interval also takes the empty lists)
                    if (end > pos) {
                        std::vector<uint32_t> &w = *dst[class_of(hi)];
                        w.insert(w.end(), order.begin() + (ptrdiff_t)pos, order.begin() + (ptrdiff_t)end);
                        pos = end;
                    }
                }
                // (the precisions are filled in when the prepass has been read back; tiny-only calls set them above)
                for (int c = 1; c < W_COUNT; c++) wl_sorted[c - 1] = true;
                r->order_desc.swap(order);  // the decode planner cuts its classes out of the same order
                r->order_max_n = max_n;
            } else {
            std::vector<std::vector<uint32_t>> part_wl((size_t)parts * W_COUNT);
            std::vector<int64_t> bad_list(parts, -1);
            bool cls_desc[W_COUNT];
            uint64_t cls_last[W_COUNT];
            for (int c = 0; c < W_COUNT; c++) { cls_desc[c] = parts == 1; cls_last[c] = ~0ull; }
            par_ranges(nlist, parts, [&](uint64_t la, uint64_t lb, unsigned tpart) {
                std::vector<uint32_t> *w = &part_wl[(size_t)tpart * W_COUNT];
                if (parts == 1) {  // (upper bounds from the counting pass: no regrowth while 65 536 lists are appended)
                    w[W_TINY].reserve(n_tiny_lists);
                    w[W_L4].reserve(use_lane ? n_mid_lists : 0);
                }
                for (uint64_t l = la; l < lb; l++) {
                    const uint64_t n = offsets[l + 1] - offsets[l];
                    if (maxid) {
                        const uint32_t m = maxid[l];
                        r->prec[l] = n == 0 ? 0u
                                     : precision_mode >= 0    ? (uint32_t)precision_mode
                                     : precision_mode == VIDC_PREC_EXACT ? (m ? 32u - (uint32_t)__builtin_clz(m) : 0u)
                                                                         : (m > 1u ? 32u - (uint32_t)__builtin_clz(m - 1u) : 0u);
                    }
                    if (n <= TINY_MAX) { w[W_TINY].push_back((uint32_t)l); continue; }
                    const uint32_t pf = pflags ? pflags[l] : 0u;  // (deferred prepass: no list of the call depends on them)
                    if (pf & VIDC_PF_DOMAIN) {
                        if (bad_list[tpart] < 0) bad_list[tpart] = (int64_t)l;
                        continue;
                    }
                    const uint32_t width = (maxid && maxid[l]) ? 32u - (uint32_t)__builtin_clz(maxid[l]) : 0u;  // ids < 2^width
                    // the bitmap kernels need no sort; they cannot report input positions of an unsorted list
                    const bool u_ok = maxid && !f_general && !((pf & VIDC_PF_UNSORTED) && want_perm) &&
                                      (n >= u_min || (pf & VIDC_PF_UNSORTED));
                    const bool lane_ok = !(pf & VIDC_PF_UNSORTED) &&
                                         ((use_lane && n <= VIDC_LANE_MAX) ||
                                          (use_lane64 && n > VIDC_LANE_MAX && n <= VIDC_LANE_MAX64));
                    // (the row-per-list kernel samples positions: ascending input; it checks that itself under the light prepass)
                    const bool grp_ok = use_grp && n >= gpol.min_n && n <= gpol.max_n && !(pf & VIDC_PF_UNSORTED);
                    const bool grp_first = grp_ok && gpol.min_lists == 0;  // VIDC_FORCE_GRP: ahead of every other family
                    const int cls = grp_first ? (n <= VIDC_GRP_LEV2_MAX ? W_G2 : W_G3)
                                    : (u_ok && width <= 18) ? W_U18
                                    : (u_ok && width <= 20) ? W_U20
                                    : grp_ok ? (n <= VIDC_GRP_LEV2_MAX ? W_G2 : W_G3)
                                    : (lane_ok && n <= 256) ? W_L4
                                    : (lane_ok && n <= VIDC_LANE_MAX) ? W_L16
                                    : lane_ok ? W_L64
                                    : n <= 4096 ? W_C1
                                    : n <= 32768 ? W_C2 : W_C3;
                    w[cls].push_back((uint32_t)l);
                    if (parts == 1) {  // (is the class already longest-first?  equal-sized lists: saves sort_desc's own pass)
                        cls_desc[cls] &= n <= cls_last[cls];
                        cls_last[cls] = n;
                    }
                }
            });
            for (unsigned t = 0; t < parts; t++)
                if (bad_list[t] >= 0) {  // the first offending list, as a single pass would report it
                    set_error("roc encode: list %llu holds an id outside [0, 2^31) (reference: int max_id, "
                              "custom_invlists_impl.cpp:163)", (unsigned long long)bad_list[t]);
                    return VIDC_ERR_DOMAIN;
                }
            for (int c = 1; c < W_COUNT; c++) wl_sorted[c - 1] = cls_desc[c];  // (ws[] below: the classes without the tiny one)
            for (int c = 0; c < W_COUNT; c++) {
                if (parts == 1) { dst[c]->swap(part_wl[c]); continue; }
                size_t tot = 0;
                for (unsigned t = 0; t < parts; t++) tot += part_wl[(size_t)t * W_COUNT + c].size();
                dst[c]->reserve(tot);
                for (unsigned t = 0; t < parts; t++) {
                    const auto &v = part_wl[(size_t)t * W_COUNT + c];
                    dst[c]->insert(dst[c]->end(), v.begin(), v.end());
                }
            }
            }  // per-list classification
        }
        ntiny = wl_tiny.size();
        tr.mark("classify");
        {
            std::vector<uint32_t> *ws[10] = {&wl_u18, &wl_u20, &wl_c1, &wl_c2, &wl_c3, &wl_l4, &wl_l16, &wl_l64, &wl_g2, &wl_g3};
            if (par_parts(nlist) > 1) {  // one thread per class (independent vectors, read-only offsets)
                std::vector<std::thread> th;
                for (int c = 1; c < 10; c++)
                    if (ws[c]->size() > 1) th.emplace_back([&, c] { sort_desc(*ws[c], r->offsets); });
                sort_desc(*ws[0], r->offsets);
                for (auto &x : th) x.join();
            } else {
                for (int c = 0; c < 10; c++)
                    if (!wl_sorted[c]) sort_desc(*ws[c], r->offsets);
            }
        }
        // The longest general lists (any precision, more than 256 ids) take the position-bitmap chain kernel
        // (k_roc_encode_r2: 0.26 instead of 0.48 us per step alone) -- when ALL the chains that decide the call's duration
        // fit on the machine at once: at most four per CU (32 KiB of LDS and 233 VGPRs each), and only if no more than that
        // many lists are at least half as long as the longest one.  1024 lists of 977 ids: encode 0.51 -> 0.35 ms; 256 x
        // 3900: 1.75 -> 1.07; 1024 x 6000: 3.9 -> 2.5.  On S2 (1754 lists of 32 769..65 536 ids) 1024 of these chains took
        // the LDS the lane-per-list classes need and the call went from 78 to 117 ms: there the general kernel, which packs
        // six times as many chains per CU, keeps all of them.
        if (!f_general && !old_u_kernels() && !env_on("VIDC_NO_R2")) {
            const size_t cap = (size_t)ctx->num_cu * 4;
            const std::vector<uint32_t> &top = !wl_c3.empty() ? wl_c3 : (!wl_c2.empty() ? wl_c2 : wl_c1);
            if (!top.empty()) {
                const uint64_t n_top = r->offsets[top[0] + 1] - r->offsets[top[0]];
                size_t n_long = 0;
                for (const std::vector<uint32_t> *w : {&wl_c3, &wl_c2, &wl_c1})
                    for (uint32_t l : *w) {
                        if (2 * (r->offsets[l + 1] - r->offsets[l]) < n_top || n_long > cap) break;
                        n_long++;
                    }
                size_t take_cap = cap;
                auto take = [&](std::vector<uint32_t> &w) {
                    size_t k = 0;
                    while (k < w.size() && wl_r2.size() < take_cap && r->offsets[w[k] + 1] - r->offsets[w[k]] > R2_MIN_LIST) {
                        wl_r2.push_back(w[k]);
                        k++;
                    }
                    w.erase(w.begin(), w.begin() + (ptrdiff_t)k);
                };
                if (n_long <= cap) {
                    take(wl_c3);
                    if (wl_c3.empty()) take(wl_c2);
                    if (wl_c3.empty() && wl_c2.empty()) take(wl_c1);
                } else if (n_top <= 65536 && !wl_c3.empty()) {
                    // More long chains than that (S2: 1754 lists of 32 769..65 536 ids): the `cap` longest ones -- one per SIMD --
                    // still decide the call (65 536 steps at 0.9 us on the general kernel under load against ~0.7 here) and, with
                    // the bitmap sized for 65 536 positions (8 KiB instead of 32), no longer take the LDS the other classes need;
                    // the rest of the class (<= ~45 000 ids on S2) finishes earlier on the general kernel anyway.
                    // Half as many again queue behind the resident ones in the same launch: each starts when one of the longest
                    // chains has finished, still ends before the launch's longest chain would have on the general kernel, and
                    // leaves that kernel ~500 fewer chains (S2 encode 59-62 -> 55-57 ms; all 1754: 59).
                    take_cap = cap + cap / 2;
                    take(wl_c3);
                }
            }
        }
        tr.mark("sort work lists");
    }
    VIDC_TRY(s_arena.get(ctx, arena_words * 4));
    VIDC_TRY(s_status.get(ctx, nlist * 4));
    if (nlist) VIDC_HIP(hipMemsetAsync(s_status.p, 0xff, nlist * 4, ctx->stream));
    VIDC_TRY(finish_prepare(ctx, r.get(), tail));
    std::vector<size_t> base;
    const uint32_t *d_wl = nullptr;
    if (!rows) {
        size_t total = 0;
        for (auto *w : {&wl_tiny, &wl_u18, &wl_u20, &wl_c1, &wl_c2, &wl_c3, &wl_l4, &wl_l16, &wl_l64, &wl_r2, &wl_g2, &wl_g3}) {
            base.push_back(total);
            total += w->size();
        }
        VIDC_TRY(h_wl.get(ctx, total * 4));
        size_t k = 0;
        for (auto *w : {&wl_tiny, &wl_u18, &wl_u20, &wl_c1, &wl_c2, &wl_c3, &wl_l4, &wl_l16, &wl_l64, &wl_r2, &wl_g2, &wl_g3}) {
            if (!w->empty()) std::memcpy(h_wl.as<uint32_t>() + k, w->data(), w->size() * 4);
            k += w->size();
        }
        VIDC_TRY(s_wl.get(ctx, total * 4));
        if (total) VIDC_HIP(hipMemcpyAsync(s_wl.p, h_wl.p, total * 4, hipMemcpyHostToDevice, ctx->stream));
        d_wl = s_wl.as<uint32_t>();
    } else {
        base.assign(12, 0);
    }

    RocEncArgs a{};
    a.ids = d_ids; a.rows = d_rows; a.K = K;
    a.offsets = rows ? nullptr : r->d_offsets.p;
    a.precision_mode = precision_mode;
    a.heads = r->d_heads.p; a.prec = r->d_prec.p; a.nwords = r->d_nwords.p; a.draws = r->d_draws.p;
    a.sizes = s_sizes.as<uint32_t>();
    a.status = s_status.as<uint32_t>();
    a.arena = s_arena.as<uint32_t>(); a.arena_stride = arena_stride;
    a.perm = want_perm ? r->d_perm.p : nullptr;
    a.sid = s_sid.as<uint32_t>(); a.spos = nullptr; a.skey = nullptr; a.skey_off = nullptr;
    a.mt = ctx->d_mt;

    // VIDC_ENC_PRIO=<mask>: s_setprio 3 for the general wave-per-list class (1), the 64- / 32-word lane classes (2), the
    // position-bitmap chains (4).  Default 2: the few wavefronts of the 2049..4096-id lane classes run 4096 steps ahead of the other
    // lane launches of their stream, and a step of theirs takes 9.5 us next to the chain classes (2.3 alone); first in their SIMD's
    // arbitration the S2 encode is 53.5-53.9 instead of 54.7-55.6 ms (interleaved, profiles/r05v_s2_ab_enc_prio.txt; the other
    // classes: no effect)
    const int enc_prio = [] { const char *e = std::getenv("VIDC_ENC_PRIO"); return e ? std::atoi(e) : 2; }();
    tr.mark("alloc + upload");
    // Kernel classes run concurrently (one stream each): the longest chains on the caller's stream, shorter
    // classes on the auxiliary streams so that they fill the wave slots the long chains leave idle.
    auto launch_gen_on = [&](hipStream_t st_, const uint32_t *wl, uint32_t nwork, uint32_t rl_max) -> int {
        if (!nwork) return VIDC_OK;
        RocEncArgs b = a;
        b.worklist = wl; b.nwork = nwork;
        if (enc_prio & 1) b.lpw = 0x80000000u;
        size_t lds = (size_t)64 * rl_max * 12;
        hipLaunchKernelGGL(k_roc_encode_gen, dim3(nwork), dim3(64), lds, st_, b, rl_max);
        VIDC_HIP(hipGetLastError());
        return VIDC_OK;
    };
    auto launch_gen = [&](const uint32_t *wl, uint32_t nwork, uint32_t rl_max) -> int {
        return launch_gen_on(ctx->stream, wl, nwork, rl_max);
    };
    {
        EventTimer t(ctx);
        VIDC_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        // an auxiliary stream joins the call at its first launch (and only those are joined at the end: a small call that uses
        // one of them does not pay event traffic for seven)
        // A call without a class for the caller's stream (no bitmap-20, deepest-general or position-bitmap lists: e.g. an index
        // of equal-sized short lists, graph rows) gives that stream to the first auxiliary class it launches: the class then
        // starts without the fork event and the call ends without its join (65 536 lists of 256 ids: ~0.1 ms of idle GPU
        // between the event records of the encode).
        const bool main_idle = wl_u20.empty() && wl_c3.empty() && wl_r2.empty();
        int promoted = -1;
        uint32_t aux_used = 0;
        const bool serial = env_on("VIDC_SERIAL");  // measurements: every class on the caller's stream, one after the other
        auto AUX = [&](int i) -> hipStream_t {
            if (serial) return ctx->stream;
            if (main_idle && (promoted < 0 || promoted == i)) { promoted = i; return ctx->stream; }
            if (!(aux_used >> i & 1u)) { (void)hipStreamWaitEvent(ctx->aux[i], ctx->ev_fork, 0); aux_used |= 1u << i; }
            return ctx->aux[i];
        };
        // main stream: bitmap-20 lists and the deepest general class (the critical paths)
        ctx->chain_info[0][0] = ctx->chain_info[0][1] = ctx->chain_info[0][2] = ctx->chain_info[0][3] = 0;
        ctx->phase_ms[VIDC_PHASE_ROC_ENCODE_CHAIN] = 0;
        auto note_chain = [&](const std::vector<uint32_t> &w, uint32_t ub) {
            uint64_t tot = 0;
            for (uint32_t l : w) tot += r->offsets[l + 1] - r->offsets[l];
            ctx->chain_info[0][0] = tot; ctx->chain_info[0][1] = w.size();
            ctx->chain_info[0][2] = r->offsets[w[0] + 1] - r->offsets[w[0]]; ctx->chain_info[0][3] = ub;
        };
        // Every kernel class of the call is one entry here: its name, its default stream (0 = the caller's, i = auxiliary i - 1)
        // and its launch.  Default streams / order = the schedule measured best (comments at the entries); VIDC_ENC_SCHED
        // (measurements) replaces it: streams separated by ';' (first = the caller's stream), the launches of a stream by ','
        // in FIFO order, "NAME^DEP" also waits for launch DEP; host launch order: first entries of every stream, then the
        // second ones, ...; names: U20 R2 C3 G3.<k> G2.<k> (octaves of the row-per-list classes, longest first) C2 U18 C1 L64 L32 L16
        // L4 TINY.  Launches the string does not name keep their default stream and go out last.
        struct EncLaunch { std::string name; int def_stream; std::function<int(hipStream_t)> fn; };
        std::vector<EncLaunch> L;
        if (!wl_u20.empty()) {  // main stream: bitmap-20 lists (the critical path of the 1 M-vector configurations)
            L.push_back({"U20", 0, [&](hipStream_t st_) -> int {
                note_chain(wl_u20, 20);
                VIDC_HIP(hipEventRecord(ctx->ev_chain[0], st_));
                RocEncArgs b = a;
                b.worklist = d_wl + base[2]; b.nwork = (uint32_t)wl_u20.size();
                const U2Div *dt = (const U2Div *)ctx->d_u2tab;
                if (old_u_kernels()) {
                    VIDC_TRY(set_big_lds((const void *)k_roc_encode_u<20, false>, UGeom<20>::LDS_BYTES));
                    VIDC_TRY(set_big_lds((const void *)k_roc_encode_u<20, true>, UGeom<20>::LDS_BYTES));
                    if (want_perm) hipLaunchKernelGGL((k_roc_encode_u<20, true>), dim3(b.nwork), dim3(64), UGeom<20>::LDS_BYTES, st_, b);
                    else hipLaunchKernelGGL((k_roc_encode_u<20, false>), dim3(b.nwork), dim3(64), UGeom<20>::LDS_BYTES, st_, b);
                } else {
                    VIDC_TRY(set_big_lds((const void *)k_roc_encode_u2<20, false>, U2Geom<20>::LDS_BYTES));
                    VIDC_TRY(set_big_lds((const void *)k_roc_encode_u2<20, true>, U2Geom<20>::LDS_BYTES));
                    if (want_perm) hipLaunchKernelGGL((k_roc_encode_u2<20, true>), dim3(b.nwork), dim3(64), U2Geom<20>::LDS_BYTES, st_, b, dt);
                    else hipLaunchKernelGGL((k_roc_encode_u2<20, false>), dim3(b.nwork), dim3(64), U2Geom<20>::LDS_BYTES, st_, b, dt);
                }
                VIDC_HIP(hipGetLastError());
                VIDC_HIP(hipEventRecord(ctx->ev_chain[1], st_));
                return VIDC_OK;
            }});
        }
        if (!wl_r2.empty()) {
            // (behind a 20-bit bitmap launch on the main stream these chains would only start when that one has finished:
            // S1 encode 12.5 -> 13.5 ms; they go to the first auxiliary stream then)
            L.push_back({"R2", wl_u20.empty() ? 0 : 1, [&](hipStream_t st_r2) -> int {
                RocEncArgs b = a;
                b.worklist = d_wl + base[9]; b.nwork = (uint32_t)wl_r2.size();
                if (enc_prio & 4) b.lpw = 0x80000000u;
                const U2Div *dt = (const U2Div *)ctx->d_u2tab;
                // (the bitmap of a launch whose longest list -- the first of the work list -- has at most 65 536 positions: 8 KiB)
                const bool small = r->offsets[wl_r2[0] + 1] - r->offsets[wl_r2[0]] <= 65536;
                if (small && want_perm) hipLaunchKernelGGL((k_roc_encode_r2<true, 10>), dim3(b.nwork), dim3(64), VIDC_R2_LDS_BYTES(10), st_r2, b, dt);
                else if (small) hipLaunchKernelGGL((k_roc_encode_r2<false, 10>), dim3(b.nwork), dim3(64), VIDC_R2_LDS_BYTES(10), st_r2, b, dt);
                else if (want_perm) hipLaunchKernelGGL((k_roc_encode_r2<true, 12>), dim3(b.nwork), dim3(64), VIDC_R2_LDS_BYTES(12), st_r2, b, dt);
                else hipLaunchKernelGGL((k_roc_encode_r2<false, 12>), dim3(b.nwork), dim3(64), VIDC_R2_LDS_BYTES(12), st_r2, b, dt);
                VIDC_HIP(hipGetLastError());
                return VIDC_OK;
            }});
        }
        // the deepest class: prefix rows sized for its longest list (first of the work list).  With the fixed
        // 48 KiB layout a CU held 3 of these chains; lists up to 65 536 ids need 12 KiB (S2 encode 152 -> see DESIGN)
        if (!wl_c3.empty()) {
            // (behind a chain launch on the main stream it would only start when that one has finished)
            L.push_back({"C3", (!wl_r2.empty() && wl_u20.empty()) ? 1 : 0, [&](hipStream_t st_c3) -> int {
                const uint64_t nmax3 = r->offsets[wl_c3[0] + 1] - r->offsets[wl_c3[0]];
                uint32_t rl3 = 16;
                while ((uint64_t)64 * 64 * rl3 < nmax3) rl3 <<= 1;
                return launch_gen_on(st_c3, d_wl + base[5], (uint32_t)wl_c3.size(), rl3);
            }});
        }
        // The lane-per-list kernels and the throughput-bound general classes must not share the machine: on S2 the
        // 64-word lane class took 69 ms next to the 26 316-list general class (49 ms) -- 9 ms and ~35 ms when each
        // has the CUs to itself (the lane strips take 141 KiB of a CU's LDS, the general waves the issue slots).
        // With lane classes in the call they go first on aux 1 and the general classes queue behind them; the long
        // chains on the main stream overlap with all of it.
        // row-per-list kernels: one launch per octave of list length (the LDS of a launch is sized by its longest list)
        {
            // Every launch is latency-bound (its duration is its longest list's chain) and light on issue slots, so the octaves
            // must overlap: they alternate between the first and the third auxiliary stream (the lane classes own the second).
            // stream digit per segment (0 = main, 1.. = auxiliary; wide: the last auxiliary stream is the plan upload's: ADVICE round 4)
            const char *gmap = ctx->wide ? "456" : "13";
            const size_t gmap_n = std::strlen(gmap);
            size_t seg_no = 0;
            auto add_grp = [&](const std::vector<uint32_t> &w, size_t wbase, bool lev3) {
                size_t k0 = 0;
                int seg_of_class = 0;
                while (k0 < w.size()) {
                    const uint64_t n0 = r->offsets[w[k0] + 1] - r->offsets[w[k0]];  // longest of the segment
                    uint64_t lo = 1;                                                 // segment: lengths in (lo, 2 lo]
                    while (lo * 2 < n0) lo *= 2;
                    size_t k1 = k0;
                    while (k1 < w.size() && r->offsets[w[k1] + 1] - r->offsets[w[k1]] > lo) k1++;
                    const int sd = gmap[seg_no++ % gmap_n] - '0';
                    const uint32_t nwork = (uint32_t)(k1 - k0);
                    const size_t woff = wbase + k0;
                    L.push_back({std::string(lev3 ? "G3." : "G2.") + std::to_string(seg_of_class++), (sd >= 1 && sd <= ctx->naux()) ? sd : 0,
                                 [&, n0, nwork, woff, lev3](hipStream_t st_g) -> int {
                        const U2Div *dt = (const U2Div *)ctx->d_u2tab;
                        RocEncArgs b = a;
                        b.worklist = d_wl + woff; b.nwork = nwork;
                        const uint32_t nblk = (uint32_t)((n0 + 511) >> 9);
                        const size_t lds = (size_t)4 * 4 * (lev3 ? roc_grp_enc_lds_words<3>(nblk) : roc_grp_enc_lds_words<2>(nblk));
                        const dim3 grid((b.nwork + 3u) / 4u);
                        if (lev3) {
                            if (lds > 65536) {
                                VIDC_TRY(set_big_lds((const void *)k_roc_encode_grp<3, true>, lds));
                                VIDC_TRY(set_big_lds((const void *)k_roc_encode_grp<3, false>, lds));
                            }
                            if (want_perm) hipLaunchKernelGGL((k_roc_encode_grp<3, true>), grid, dim3(64), lds, st_g, b, dt, nblk);
                            else hipLaunchKernelGGL((k_roc_encode_grp<3, false>), grid, dim3(64), lds, st_g, b, dt, nblk);
                        } else if (want_perm) hipLaunchKernelGGL((k_roc_encode_grp<2, true>), grid, dim3(64), lds, st_g, b, dt, nblk);
                        else hipLaunchKernelGGL((k_roc_encode_grp<2, false>), grid, dim3(64), lds, st_g, b, dt, nblk);
                        VIDC_HIP(hipGetLastError());
                        return VIDC_OK;
                    }});
                    k0 = k1;
                }
            };
            add_grp(wl_g3, base[11], true);
            add_grp(wl_g2, base[10], false);
        }
        const bool lanes_present = !wl_l4.empty() || !wl_l16.empty() || !wl_l64.empty();
        // aux 0: mid-size general lists (calls without lane classes) and bitmap-18 lists
        auto add_c2 = [&](int sdef) {
            L.push_back({"C2", sdef, [&](hipStream_t st_) -> int { return launch_gen_on(st_, d_wl + base[4], (uint32_t)wl_c2.size(), 8); }});
        };
        auto add_c1 = [&](int sdef) {
            L.push_back({"C1", sdef, [&](hipStream_t st_) -> int { return launch_gen_on(st_, d_wl + base[3], (uint32_t)wl_c1.size(), 1); }});
        };
        if (!lanes_present && !wl_c2.empty()) add_c2(1);
        if (!wl_u18.empty()) {
            L.push_back({"U18", 1, [&](hipStream_t st_) -> int {
                const bool is_chain = wl_u20.empty();
                if (is_chain) { note_chain(wl_u18, 18); VIDC_HIP(hipEventRecord(ctx->ev_chain[0], st_)); }
                RocEncArgs b = a;
                b.worklist = d_wl + base[1]; b.nwork = (uint32_t)wl_u18.size();
                const U2Div *dt = (const U2Div *)ctx->d_u2tab;
                if (old_u_kernels()) {
                    if (want_perm) hipLaunchKernelGGL((k_roc_encode_u<18, true>), dim3(b.nwork), dim3(64), UGeom<18>::LDS_BYTES, st_, b);
                    else hipLaunchKernelGGL((k_roc_encode_u<18, false>), dim3(b.nwork), dim3(64), UGeom<18>::LDS_BYTES, st_, b);
                } else if (want_perm) hipLaunchKernelGGL((k_roc_encode_u2<18, true>), dim3(b.nwork), dim3(64), U2Geom<18>::LDS_BYTES, st_, b, dt);
                else hipLaunchKernelGGL((k_roc_encode_u2<18, false>), dim3(b.nwork), dim3(64), U2Geom<18>::LDS_BYTES, st_, b, dt);
                VIDC_HIP(hipGetLastError());
                if (is_chain) VIDC_HIP(hipEventRecord(ctx->ev_chain[1], st_));
                return VIDC_OK;
            }});
        }
        // aux 1: the lane-per-list kernels, then the general classes; aux 2: tiny lists
        if (!lanes_present && !wl_c1.empty()) add_c1(2);
        for (int cls = 2; cls >= 0; cls--) {  // longest chains first
            const std::vector<uint32_t> &w = cls == 2 ? wl_l64 : (cls ? wl_l16 : wl_l4);
            if (w.empty()) continue;
            RocEncArgs b = a;
            b.worklist = d_wl + base[6 + cls]; b.nwork = (uint32_t)w.size();
            // (LDS per wavefront: 4-word strips 14.3 KiB, 16-word 21.5 KiB, 32-word 31.5 KiB, 64-word 50.5 KiB)
            b.lpw = lane_lists_per_wave(ctx, b.nwork, cls == 0 ? 15 : cls == 1 ? 11 : 4);
            if (cls == 2) {
                // lists are sorted longest first: the leading wavefronts need the 64-word strips (50.5 KiB of LDS,
                // 3 per CU), everything from the first wavefront whose longest list has <= 2048 ids the 32-word
                // ones (31.5 KiB, 5 per CU)
                uint32_t n_big = 0;
                while (n_big < b.nwork && r->offsets[w[n_big] + 1] - r->offsets[w[n_big]] > 2048) n_big++;
                n_big = std::min<uint32_t>(b.nwork, (n_big + b.lpw - 1u) / b.lpw * b.lpw);
                if (enc_prio & 2) b.lpw |= 0x80000000u;
                RocEncArgs b2 = b;
                b2.worklist = b.worklist + n_big; b2.nwork = b.nwork - n_big;
                b.nwork = n_big;
                if (b.nwork)
                    L.push_back({"L64", 2, [&, b](hipStream_t st_) -> int {
                        const LaneDiv *dt = (const LaneDiv *)ctx->d_ltab;
                        const dim3 g1((b.nwork + (b.lpw & 0xffffu) - 1u) / (b.lpw & 0xffffu));
                        if (want_perm) hipLaunchKernelGGL((k_roc_encode_lane<64, true>), g1, dim3(64), 0, st_, b, dt);
                        else hipLaunchKernelGGL((k_roc_encode_lane<64, false>), g1, dim3(64), 0, st_, b, dt);
                        VIDC_HIP(hipGetLastError());
                        return VIDC_OK;
                    }});
                if (b2.nwork)
                    L.push_back({"L32", 2, [&, b2](hipStream_t st_) -> int {
                        const LaneDiv *dt = (const LaneDiv *)ctx->d_ltab;
                        const dim3 g2((b2.nwork + (b2.lpw & 0xffffu) - 1u) / (b2.lpw & 0xffffu));
                        if (want_perm) hipLaunchKernelGGL((k_roc_encode_lane<32, true>), g2, dim3(64), 0, st_, b2, dt);
                        else hipLaunchKernelGGL((k_roc_encode_lane<32, false>), g2, dim3(64), 0, st_, b2, dt);
                        VIDC_HIP(hipGetLastError());
                        return VIDC_OK;
                    }});
            } else {
                L.push_back({cls == 0 ? "L4" : "L16", 2, [&, b, cls](hipStream_t st_) -> int {
                    const LaneDiv *dt = (const LaneDiv *)ctx->d_ltab;
                    const dim3 grid((b.nwork + b.lpw - 1u) / b.lpw);
                    if (cls == 0 && want_perm) hipLaunchKernelGGL((k_roc_encode_lane<4, true>), grid, dim3(64), 0, st_, b, dt);
                    else if (cls == 0) hipLaunchKernelGGL((k_roc_encode_lane<4, false>), grid, dim3(64), 0, st_, b, dt);
                    else if (want_perm) hipLaunchKernelGGL((k_roc_encode_lane<16, true>), grid, dim3(64), 0, st_, b, dt);
                    else hipLaunchKernelGGL((k_roc_encode_lane<16, false>), grid, dim3(64), 0, st_, b, dt);
                    VIDC_HIP(hipGetLastError());
                    return VIDC_OK;
                }});
            }
        }
        if (lanes_present) {
            if (!wl_c2.empty()) add_c2(2);
            if (!wl_c1.empty()) add_c1(2);
        }
        if (ntiny) {
            L.push_back({"TINY", 3, [&](hipStream_t st_) -> int {
                RocEncArgs b = a;
                b.worklist = rows ? nullptr : d_wl; b.nwork = (uint32_t)ntiny;
                if (use_lane_tiny) {  // one list per lane (roc_lane.h)
                    const dim3 grid((b.nwork + 63u) / 64u);
                    const LaneDiv *dt = (const LaneDiv *)ctx->d_ltab;
                    if (rows && K == 32) hipLaunchKernelGGL((k_roc_encode_tiny_lane<32, true, true>), grid, dim3(64), 0, st_, b, dt);
                    else if (rows && K == 64) hipLaunchKernelGGL((k_roc_encode_tiny_lane<64, true, true>), grid, dim3(64), 0, st_, b, dt);
                    else if (rows && K < 32) hipLaunchKernelGGL((k_roc_encode_tiny_lane<32, true>), grid, dim3(64), 0, st_, b, dt);
                    else if (rows) hipLaunchKernelGGL((k_roc_encode_tiny_lane<64, true>), grid, dim3(64), 0, st_, b, dt);
                    else hipLaunchKernelGGL((k_roc_encode_tiny_lane<64, false>), grid, dim3(64), 0, st_, b, dt);
                } else if (rows) hipLaunchKernelGGL(k_roc_encode_tiny<true>, dim3(b.nwork), dim3(64), 0, st_, b);
                else hipLaunchKernelGGL(k_roc_encode_tiny<false>, dim3(b.nwork), dim3(64), 0, st_, b);
                VIDC_HIP(hipGetLastError());
                return VIDC_OK;
            }});
        }
        {
            auto stream_no = [&](int sno) -> hipStream_t { return sno == 0 ? ctx->stream : AUX(sno - 1); };
            std::vector<int> sch_stream(L.size(), -1), sch_pos(L.size(), 0);
            std::vector<std::vector<int>> deps(L.size());
            auto find = [&](const std::string &t) { for (size_t i = 0; i < L.size(); i++) if (L[i].name == t) return (int)i; return -1; };
            std::vector<int> order_l;
            if (const char *e = std::getenv("VIDC_ENC_SCHED")) {
                std::string str(e);
                int grp = 0, pos = 0;
                size_t i0 = 0;
                for (size_t i = 0; i <= str.size(); i++) {
                    if (i < str.size() && str[i] != ',' && str[i] != ';') continue;
                    std::string tok = str.substr(i0, i - i0);
                    i0 = i + 1;
                    if (!tok.empty()) {
                        int me = -1;
                        size_t j0 = 0;
                        bool first = true;
                        for (size_t j = 0; j <= tok.size(); j++) {
                            if (j < tok.size() && tok[j] != '^') continue;
                            const int c = find(tok.substr(j0, j - j0));
                            j0 = j + 1;
                            if (first) me = c; else if (c >= 0 && me >= 0) deps[me].push_back(c);
                            first = false;
                        }
                        if (me >= 0 && sch_stream[me] < 0 && grp <= ctx->naux()) { sch_stream[me] = grp; sch_pos[me] = pos++; order_l.push_back(me); }
                    }
                    if (i < str.size() && str[i] == ';') { grp++; pos = 0; }
                }
                std::stable_sort(order_l.begin(), order_l.end(), [&](int x, int y) { return sch_pos[x] < sch_pos[y]; });
            }
            for (size_t i = 0; i < L.size(); i++)
                if (sch_stream[i] < 0) order_l.push_back((int)i);
            // dependency events: the 24 PhaseTimer events of the context double as "launch i has been queued" markers, so a schedule
            // can only name the first 24 launches as something to wait for (others are dropped, not aliased), and an event is only
            // recorded for a launch that something waits for (no VIDC_ENC_SCHED: none at all)
            std::vector<char> waited_on(L.size(), 0);
            for (auto &dv : deps) {
                dv.erase(std::remove_if(dv.begin(), dv.end(), [](int d) { return d >= 24; }), dv.end());
                for (int d : dv) waited_on[d] = 1;
            }
            std::vector<char> done(L.size(), 0);
            size_t left = L.size();
            while (left) {
                const size_t before = left;
                for (int i : order_l) {
                    if (done[i]) continue;
                    bool ready = true;
                    for (int d : deps[i]) ready &= done[d] != 0;
                    if (!ready) continue;
                    hipStream_t st_ = serial ? ctx->stream : (sch_stream[i] >= 0 ? (sch_stream[i] == 0 ? ctx->stream : [&] {
                        const int x = sch_stream[i] - 1;
                        if (!(aux_used >> x & 1u)) { (void)hipStreamWaitEvent(ctx->aux[x], ctx->ev_fork, 0); aux_used |= 1u << x; }
                        return ctx->aux[x];
                    }()) : stream_no(L[i].def_stream));
                    for (int d : deps[i]) VIDC_HIP(hipStreamWaitEvent(st_, ctx->tev[d], 0));
                    VIDC_TRY(L[i].fn(st_));
                    if (waited_on[i]) VIDC_HIP(hipEventRecord(ctx->tev[i], st_));
                    done[i] = 1;
                    left--;
                }
                if (left == before) {  // circular / unknown dependencies: drop them
                    for (auto &d : deps) d.clear();
                }
            }
        }
        for (int i = 0; i < ctx->naux(); i++) {
            if (!(aux_used >> i & 1u)) continue;
            VIDC_HIP(hipEventRecord(ctx->ev_join[i], ctx->aux[i]));
            VIDC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[i], 0));
        }
        // the kernels are in flight and the host has nothing to do until they finish: plan the decode of the whole
        // object now (7 ms per million lists that decode_all would otherwise spend on its critical path)
        // (the end event is recorded first: the planning is host time, not part of the kernels' duration)
        t.mark();
        VIDC_TRY(finish_enqueue(ctx, r.get(), tail, s_status, s_sizes.as<uint32_t>()));
        if (pre_deferred) {  // the maxima of the lists: precisions and bucket geometry for the decode planner
            VIDC_HIP(vidc::vidc_event_wait(ctx->ev_pre[2]));
            const uint32_t *mx = h_pre.as<uint32_t>();
            r->umax.assign(mx, mx + nlist);
            for (uint64_t l = 0; l < nlist; l++) {
                const uint32_t m = mx[l];
                r->prec[l] = offsets[l + 1] == offsets[l] ? 0u
                             : precision_mode >= 0    ? (uint32_t)precision_mode
                             : precision_mode == VIDC_PREC_EXACT ? (m ? 32u - (uint32_t)__builtin_clz(m) : 0u)
                                                                 : (m > 1u ? 32u - (uint32_t)__builtin_clz(m - 1u) : 0u);
            }
            float pms = 0;
            if (hipEventElapsedTime(&pms, ctx->ev_pre[0], ctx->ev_pre[1]) == hipSuccess) kernel_ms += pms;
        }
        if (!rows && nlist >= PLAN_AHEAD_MIN_LISTS && r->prec.size() == nlist) {
            r->plan_ahead = plan_ahead_build(r.get(), ctx->wide);
            // ... and send its work list and scratch offsets to the device behind the kernels (decode_all then finds the
            // plan complete: 0.04 ms of its critical path at 65 536 lists)
            DecPlanCache &pc = *r->plan_ahead;
            const size_t ni = pc.plan.wl.size();
            if (ni && !pc.plan.lean) {
                VIDC_TRY(h_plan.get(ctx, ni * 20));
                uint64_t *h64 = h_plan.as<uint64_t>();
                uint32_t *h32 = (uint32_t *)(h64 + 2 * ni);
                std::memcpy(h64, pc.plan.scratch_off.data(), ni * 8);
                std::memcpy(h64 + ni, pc.plan.slots_off.data(), ni * 8);
                std::memcpy(h32, pc.plan.wl.data(), ni * 4);
                VIDC_TRY(pc.d_scratch_off.alloc(ni, ctx->dpool));
                VIDC_TRY(pc.d_slots_off.alloc(ni, ctx->dpool));
                VIDC_TRY(pc.d_wl.alloc(ni, ctx->dpool));
                // (on a stream of their own -- the last auxiliary one, which no kernel class of this call uses --: behind the
                // kernels on the caller's stream the 1.3 MB of a 65 536-list plan were 0.06 ms more before the call's wait)
                hipStream_t cs = ctx->aux[VIDC_NAUX - 1];
                VIDC_HIP(hipMemcpyAsync(pc.d_scratch_off.p, h64, ni * 8, hipMemcpyHostToDevice, cs));
                VIDC_HIP(hipMemcpyAsync(pc.d_slots_off.p, h64 + ni, ni * 8, hipMemcpyHostToDevice, cs));
                VIDC_HIP(hipMemcpyAsync(pc.d_wl.p, h32, ni * 4, hipMemcpyHostToDevice, cs));
                VIDC_HIP(hipEventRecord(ctx->ev_join[VIDC_NAUX - 1], cs));
                VIDC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[VIDC_NAUX - 1], 0));
                plan_on_device = true;
            }
        }
        // ONE wait for the kernels, the status summary and the sizes (queued behind the kernels above)
        VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
        pre_guard.armed = false;
        kernel_ms += t.elapsed();
        if (ctx->chain_info[0][3]) {  // (the main stream joined the auxiliary ones)
            float cms = 0;
            if (hipEventElapsedTime(&cms, ctx->ev_chain[0], ctx->ev_chain[1]) == hipSuccess) ctx->phase_ms[VIDC_PHASE_ROC_ENCODE_CHAIN] = cms;
        }
    }

    tr.mark("encode kernels (sync)");
    // second pass for lists the first pass handed back: unsorted input of the general kernels (Faiss
    // lists are in add order, i.e. normally sorted) and multiset input of the bitmap kernels
    std::vector<uint32_t> pend;
    if (!rows && nlist) {
        // how many lists are pending: the summary came back with the sizes, the status array only if there are any
        const unsigned long long *t = tail.t;
        if (t[3]) {
            std::vector<uint32_t> status;
            VIDC_TRY(download(ctx, status, s_status.as<uint32_t>(), nlist));
            for (uint64_t l = 0; l < nlist; l++)
                if (status[l] == VIDC_ST_PENDING_SORT) pend.push_back((uint32_t)l);
        }
        if (!pend.empty()) {
            sort_desc(pend, r->offsets);
            std::vector<uint64_t> skey_off(nlist + 1, 0);
            uint32_t maxn = 0;
            {
                std::vector<uint8_t> is_p(nlist, 0);
                for (uint32_t l : pend) is_p[l] = 1;
                for (uint64_t l = 0; l < nlist; l++) {
                    uint64_t n = r->offsets[l + 1] - r->offsets[l], p2 = 0;
                    if (is_p[l]) { p2 = 1; while (p2 < n) p2 <<= 1; maxn = std::max<uint32_t>(maxn, (uint32_t)n); }
                    skey_off[l + 1] = skey_off[l] + p2;
                }
            }
            Scratch s_skey, s_skey_off, s_spos, s_pend;
            VIDC_TRY(s_skey.get(ctx, skey_off[nlist] * 8));
            VIDC_TRY(upload_scratch(ctx, s_skey_off, skey_off));
            VIDC_TRY(s_spos.get(ctx, ntotal_in * 4));
            if (!s_sid.p) VIDC_TRY(s_sid.get(ctx, ntotal_in * 4));
            VIDC_TRY(upload_scratch(ctx, s_pend, pend));
            a.sid = s_sid.as<uint32_t>();
            a.skey = s_skey.as<uint64_t>(); a.skey_off = s_skey_off.as<uint64_t>(); a.spos = s_spos.as<uint32_t>();
            uint32_t rl_max = maxn <= 4096 ? 1 : (maxn <= 32768 ? 8 : (maxn <= 65536 ? 16 : 64));
            EventTimer t2(ctx);
            VIDC_TRY(launch_gen(s_pend.as<uint32_t>(), (uint32_t)pend.size(), rl_max));
            kernel_ms += t2.stop();
            VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));  // scratch of this scope is released below
            if (light_prepass) {
                // the precision of these lists came from their last id, which was not their maximum if they were not
                // ascending: the general kernel stored the real one
                std::vector<uint32_t> dp;
                VIDC_TRY(download(ctx, dp, r->d_prec.p, nlist));
                for (uint32_t l : pend) { r->prec[l] = dp[l]; r->umax[l] = 0; }
                r->plan_ahead.reset();
            }
            // statuses and word counts of the redone lists: summary and offsets again
            VIDC_TRY(finish_enqueue(ctx, r.get(), tail, s_status, s_sizes.as<uint32_t>()));
            VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
        }
    }
    // bitmap-kernel lists wrote sampled ids into the perm buffer: turn them into input positions
    if (want_perm && (!wl_u18.empty() || !wl_u20.empty())) {
        std::vector<uint8_t> is_p(nlist, 0);
        for (uint32_t l : pend) is_p[l] = 1;
        std::vector<uint32_t> ul;
        for (uint32_t l : wl_u18) if (!is_p[l]) ul.push_back(l);
        for (uint32_t l : wl_u20) if (!is_p[l]) ul.push_back(l);
        if (!ul.empty()) {
            // lists -> 8 lanes (longest first to the lane with the fewest chunks), lane x's k-th chunk at item 8 k + x
            std::vector<uint32_t> byl(ul);
            std::stable_sort(byl.begin(), byl.end(), [&](uint32_t x, uint32_t y) {
                return r->offsets[x + 1] - r->offsets[x] > r->offsets[y + 1] - r->offsets[y];
            });
            std::vector<std::vector<std::pair<uint32_t, uint32_t>>> lanes(8);
            for (uint32_t l : byl) {
                size_t best = 0;
                for (size_t x = 1; x < 8; x++)
                    if (lanes[x].size() < lanes[best].size()) best = x;
                for (uint64_t st0 = 0, n = r->offsets[l + 1] - r->offsets[l]; st0 < n; st0 += VIDC_PERM_CHUNK)
                    lanes[best].push_back({l, (uint32_t)st0});
            }
            size_t depth = 0;
            for (auto &ln : lanes) depth = std::max(depth, ln.size());
            std::vector<uint32_t> items(depth * 8 * 2, 0xffffffffu);
            for (size_t x = 0; x < 8; x++)
                for (size_t k = 0; k < lanes[x].size(); k++) {
                    items[(k * 8 + x) * 2] = lanes[x][k].first;
                    items[(k * 8 + x) * 2 + 1] = lanes[x][k].second;
                }
            Scratch s_ul;
            VIDC_TRY(upload_scratch(ctx, s_ul, items));
            const uint32_t nitems = (uint32_t)(depth * 8);
            EventTimer t(ctx);
            hipLaunchKernelGGL(k_perm_from_order, dim3(nitems), dim3(256), 0, ctx->stream, d_ids, r->d_offsets.p, s_ul.as<uint2>(),
                               nitems, r->d_perm.p);
            VIDC_HIP(hipGetLastError());
            kernel_ms += t.stop();
        }
    }
    tr.mark("second pass / perm");
    ctx->phase_ms[VIDC_PHASE_ROC_ENCODE] = kernel_ms;
    VIDC_TRY(finish_complete(ctx, r.get(), tail, s_arena.as<uint32_t>(), arena_stride, s_status, nonempty, kernel_ms));
    tr.mark("metadata + compaction");
    ctx->last_kernel_ms = kernel_ms;
    if (plan_on_device && r->plan_ahead) r->plan_all = std::move(r->plan_ahead);  // (a sorting second pass drops the plan)
    *out = r.release();
    return VIDC_OK;
}

inline DecClass grp_dec_class(uint64_t n) {
    const uint32_t f = roc_grp_dec_fbits((uint32_t)n);
    return f == 0 ? DC_GRP0 : (f == 2 ? DC_GRP2 : (f == 3 ? DC_GRP3 : DC_GRP4));
}
// (the environment is read once per plan, not once per list: three getenv per list were 1 ms of a 65 536-list plan)
struct DecEnv {
    bool nb256, pair, quad;
    bool nb128, mid128;
    uint64_t pair_min = VIDC_LANE_REG_MAX;  // lists of more than this many ids (up to 512) decode on a PAIR of lanes
    DecEnv() : nb256(false),  // (256 buckets for the 257..1024-id lists too: measured no better, HISTORY.md)
               pair(!env_on("VIDC_NO_LANE_PAIR") && !env_on("VIDC_NO_LANE_REG")),
               // (quads of lanes for 513..1024 ids: opt-in.  Measured slower than the bucket rows: 65 536 x 1024 ids decode in 5.6
               // instead of 4.1 ms -- 16 lists per 256-VGPR wavefront means two rounds of wavefronts per SIMD --, S2 84 instead of 78 ms)
               quad(env_on("VIDC_LANE_QUAD") && !env_on("VIDC_NO_LANE_PAIR") && !env_on("VIDC_NO_LANE_REG")) {
        nb128 = !env_on("VIDC_NO_LANE128");  // lists of 1025..2048 ids on 128 buckets (VIDC_NO_LANE128=1: 256, as before round 4)
        mid128 = false;  // (the bucket-row lists of up to 1024 ids on 128 buckets: measured no better, HISTORY.md)
        // VIDC_PAIR_MIN=n (64 .. 256): lists of n+1 .. 256 ids on lane pairs too -- 32 lists per wavefront, i.e. two wavefronts
        // per SIMD where a call of 65 536 lists has one (measurements: DESIGN section 12)
        if (const char *e = std::getenv("VIDC_PAIR_MIN")) pair_min = std::min<uint64_t>(VIDC_LANE_REG_MAX, std::max<uint64_t>(TINY_MAX, (uint64_t)std::atoll(e)));
    }
};
inline DecClass dec_class(uint64_t n, uint32_t P, uint64_t u_min, bool f_general, bool allow_lane, bool allow_lane64,
                          const GrpPolicy *grp, const DecEnv &env) {  // (allow_lane*: mid-size policies; grp: row-per-list kernels wanted)
    if (n <= TINY_MAX) return DC_TINY;
    const bool grp_ok = grp && n >= grp->dec_min_n && n <= grp->dec_max_n && P <= 32;
    if (grp_ok && grp->min_lists == 0) return grp_dec_class(n);  // VIDC_FORCE_GRP: ahead of every other family
    if (!f_general && n >= u_min) {
        if (P <= 18) return DC_U18;
        if (P <= 20) return DC_U20;
    }
    if (grp_ok) return grp_dec_class(n);
    // lists of 257..512 ids: the register decoder on lane pairs (VIDC_NO_LANE_PAIR=1: the bucket-row decoder)
    if (allow_lane && env.pair && n > env.pair_min && n <= VIDC_LANE_PAIR_MAX) return DC_LANEP;
    if (allow_lane && env.quad && n > VIDC_LANE_PAIR_MAX && n <= VIDC_LANE_QUAD_MAX) return DC_LANEQ;
    if (allow_lane && env.nb256 && n > VIDC_LANE_REG_MAX && n <= VIDC_LANE_MAX) return DC_LANE64;
    if (allow_lane && env.mid128 && n > VIDC_LANE_REG_MAX && n <= VIDC_LANE_MAX) return DC_LANE128;
    if (allow_lane && n <= VIDC_LANE_MAX) return DC_LANE;
    if (allow_lane64 && env.nb128 && n > VIDC_LANE_MAX && n <= VIDC_LANE_MAX128) return DC_LANE128;
    if (allow_lane64 && n > VIDC_LANE_MAX && n <= VIDC_LANE_MAX64) return DC_LANE64;
    if (n <= GEN_SMALL_MAX) return DC_GSMALL;
    if (n <= 8192) return DC_G8K;
    if (n <= 16384) return DC_G16K;
    if (n <= 32768) return DC_GMID;
    return DC_GHUGE;
}

// lists[i] = list number of request item i (a list may appear more than once)
// only_general: the caller decodes into int32 rows (wide graph rows): only the kernels that honour out_rows -- the tiny
// and the general wave-per-list decoders -- may be planned
// whole_sorted: `lists` is 0 .. nlist-1 and r->order_desc holds the same lists longest first
// wide: the context that will run the plan has a stream per kernel class (8 hardware queues): the row-per-list kernels are
// only planned then, like the encoder only takes them then (their octaves must overlap)
void plan_decode(const vidc_roc *r, const std::vector<uint32_t> &lists, bool rows_flavour, DecPlan &p, bool wide,
                 bool allow_lane = true, bool allow_b2 = true, bool only_general = false, bool whole_sorted = false) {
    const bool f_general = force_general() || only_general;
    if (only_general) { allow_lane = false; allow_b2 = false; }
    const LanePolicy lpol = (allow_lane && !f_general) ? lane_policy() : LANE_NEVER;
    p.wl.clear(); p.item.clear(); p.scratch_off.clear(); p.slots_off.clear();
    p.scratch_words = 0; p.slots_words = 0;
    for (int c = 0; c < DC_COUNT; c++) { p.count[c] = 0; p.sum_n[c] = 0; p.max_n[c] = 0; }
    p.tiny_lane = false;
    if (rows_flavour && lane_wanted(lpol, lists.size(), LANE_MIN_TINY)) {
        // graph rows with the lane-per-row decoder: request order, no scratch, implicit output offsets
        p.wl = lists;
        p.count[DC_TINY] = lists.size();
        p.lean = true;
        p.tiny_lane = true;
        return;
    }
    p.lean = false;
    std::vector<uint32_t> cls[DC_COUNT];
    const uint64_t u_min = U_MIN_LIST;
    bool allow_lane64 = false;
    const GrpPolicy gpol = grp_policy();
    const DecEnv denv;
    bool use_grp = false;
    auto len = [&](uint32_t i) { return r->offsets[lists[i] + 1] - r->offsets[lists[i]]; };
    bool by_length = false;
    if (whole_sorted && !rows_flavour && !f_general && r->order_desc.size() == lists.size() && r->order_max_n <= VIDC_LANE_MAX64 &&
        !env_on("VIDC_NO_LENGTH_CLASSES")) {
        // The whole object, no list beyond the lane classes, and the encoder left its lists ordered by length: the class of a list
        // is a function of its length (dec_class below looks at the precision only for lists of more than 4096 ids), so every class
        // is one or two runs of that order.  Replaces three passes over the lists (counts, classification with a push_back per
        // list, sortedness) of a 65 536-list plan.
        const std::vector<uint32_t> &order = r->order_desc;
        const uint64_t *offs = r->offsets.data();
        const size_t nl = order.size();
        auto first_le = [&](uint64_t bound) {
            size_t lo = 0, hi = nl;
            while (lo < hi) { const size_t mid = (lo + hi) / 2; if (offs[order[mid] + 1] - offs[order[mid]] > bound) lo = mid + 1; else hi = mid; }
            return lo;
        };
        const size_t e4096 = 0, e2048 = first_le(VIDC_LANE_MAX128), e1024 = first_le(VIDC_LANE_MAX), e512 = first_le(VIDC_LANE_PAIR_MAX),
                     e256 = first_le(denv.pair ? denv.pair_min : VIDC_LANE_REG_MAX), e64 = first_le(TINY_MAX);
        const uint64_t n_mid64 = e1024 - e4096, n_mid = e64 - e1024, n_tiny = nl - e64;
        allow_lane = lane_wanted(lpol, n_mid, LANE_MIN_LISTS);
        allow_lane64 = lane_wanted(lpol, n_mid64, LANE_MIN_LISTS64);
        p.tiny_lane = lane_wanted(lpol, n_tiny, LANE_MIN_TINY);
        const bool grp_in_reach = allow_b2 && gpol.dec_min_n <= r->order_max_n && gpol.min_lists != ~0ull;  // (VIDC_FORCE_GRP / VIDC_GRP_DEC_MINN)
        if ((allow_lane || !n_mid) && (allow_lane64 || !n_mid64) && !grp_in_reach) {
            by_length = true;
            auto take = [&](int c, size_t a, size_t b) { if (b > a) cls[c].insert(cls[c].end(), order.begin() + (ptrdiff_t)a, order.begin() + (ptrdiff_t)b); };
            take(DC_LANE64, e4096, e2048);
            take(denv.nb128 ? DC_LANE128 : DC_LANE64, e2048, e1024);
            // 513..1024 | 257..512 | 65..256 (dec_class: pair, quad, 256 buckets, 64 buckets -- in that order)
            const int mid = denv.nb256 ? DC_LANE64 : (denv.mid128 ? DC_LANE128 : DC_LANE);
            take(denv.quad ? DC_LANEQ : mid, e1024, e512);
            take(denv.pair ? DC_LANEP : mid, e512, e256);
            take(DC_LANE, e256, e64);
            // (tiny lists in request order, like the per-list loop)
            if (n_tiny) {
                if (n_tiny == nl) { cls[DC_TINY].resize(nl); std::iota(cls[DC_TINY].begin(), cls[DC_TINY].end(), 0u); }
                else {
                    cls[DC_TINY].assign(order.begin() + (ptrdiff_t)e64, order.end());
                    std::sort(cls[DC_TINY].begin(), cls[DC_TINY].end());
                }
            }
        }
    }
    if (!by_length) {
    {
        uint64_t n_mid = 0, n_mid64 = 0, n_tiny = 0, n_grp = 0;
        for (uint32_t l : lists) {
            const uint64_t n = r->offsets[l + 1] - r->offsets[l];
            n_tiny += n <= TINY_MAX;
            n_mid += n > TINY_MAX && n <= VIDC_LANE_MAX;
            n_mid64 += n > VIDC_LANE_MAX && n <= VIDC_LANE_MAX64;
            n_grp += n >= gpol.dec_min_n && n <= gpol.dec_max_n;
        }
        use_grp = allow_b2 && !f_general && !rows_flavour && n_grp && n_grp >= gpol.min_lists &&
                  (wide || gpol.min_lists == 0);
        allow_lane = lane_wanted(lpol, n_mid, LANE_MIN_LISTS);
        allow_lane64 = lane_wanted(lpol, n_mid64, LANE_MIN_LISTS64);
        p.tiny_lane = lane_wanted(lpol, rows_flavour ? lists.size() : n_tiny, LANE_MIN_TINY);
    }
    {
        // (the classes that can hold most lists of a big call grow once, not by doubling)
        const uint64_t nl = lists.size();
        if (nl >= 4096) for (int c : {(int)DC_TINY, (int)DC_LANE, (int)DC_GSMALL}) cls[c].reserve(nl);
    }
    for (uint32_t i = 0; i < lists.size(); i++) {
        uint32_t l = lists[i];
        uint64_t n = r->offsets[l + 1] - r->offsets[l];
        cls[rows_flavour ? DC_TINY : dec_class(n, r->prec[l], u_min, f_general, allow_lane, allow_lane64, use_grp ? &gpol : nullptr, denv)].push_back(i);
    }
    for (int c = 0; c < DC_COUNT; c++) {
        if (c != DC_TINY && cls[c].size() > 1) {  // counting sort by length, longest first (stable)
            uint64_t maxlen = 0, prev = ~0ull;
            bool sorted_already = true;  // equal-sized lists, or an index whose lists come longest first: nothing to do
            for (uint32_t i : cls[c]) {
                const uint64_t n = len(i);
                maxlen = std::max<uint64_t>(maxlen, n);
                sorted_already &= n <= prev;
                prev = n;
            }
            if (sorted_already) continue;
            std::vector<uint32_t> start(maxlen + 2, 0), sorted(cls[c].size());
            for (uint32_t i : cls[c]) start[maxlen - len(i) + 1]++;
            for (size_t k = 1; k < start.size(); k++) start[k] += start[k - 1];
            for (uint32_t i : cls[c]) sorted[start[maxlen - len(i)]++] = i;
            cls[c].swap(sorted);
        }
    }
    }  // per-list classification
    // The longest general chains (precision above 20 bits) take k_roc_decode_b2 under the rule of the encoder's
    // k_roc_encode_r2: only when every chain at least half as long as the longest one gets it (<= B2_CAP of them).
    if (allow_b2 && !f_general && !rows_flavour && !old_u_kernels() && !env_on("VIDC_NO_R2")) {
        const int order_[4] = {DC_GHUGE, DC_GMID, DC_G16K, DC_G8K};
        uint64_t n_top = 0;
        for (int c : order_)
            if (!cls[c].empty()) { n_top = len(cls[c][0]); break; }
        size_t n_long = 0;
        bool stop = false;
        for (int c : order_) {
            for (uint32_t i : cls[c]) {
                if (2 * len(i) < n_top || n_long > B2_CAP) { stop = true; break; }
                n_long++;
            }
            if (stop) break;
        }
        size_t b2_cap = B2_CAP;
        // More long chains than the cap (S2: 1754 lists of 32 769..65 536 ids and 2666 of 16 385..32 768): round 3 gave the `cap`
        // longest ones to this kernel and left the rest on the general decoder, whose step under load is ~1.1 us against ~0.65
        // here.  Measured in round 4 (S2 decode, kernels): 1024 chains 77-79 ms, all 1754 lists beyond 32 768 ids 74, these and
        // the 2666 lists of 16 385..32 768 ids 71-73; the lists of 8193..16 384 ids too (instead of the row-per-list kernel): 80-90.
        // So every list beyond 16 384 ids takes it, up to B2_TOP_CAP chains (20 per CU of an MI355X: 8 KiB of LDS each).
        const bool top_only = n_top && n_long > b2_cap;
        if (top_only) b2_cap = B2_TOP_CAP;
        if (n_top && (n_long <= b2_cap || top_only))
            for (int c : order_) {  // longest first; a list that does not qualify stays where it is
                if (top_only && c != DC_GHUGE && c != DC_GMID) break;
                std::vector<uint32_t> keep;
                for (uint32_t i : cls[c]) {
                    const uint32_t P = r->prec[lists[i]];
                    // (beyond ~100 000 ids a 64-member row overflows too often: average bucket load n / 4096)
                    // (shorter than B2_MIN_LIST: the general decoder keeps the member rows of such a list in LDS, while 1 MiB of
                    // sparse rows per list -- 1 GiB for a thousand lists -- makes every step of this kernel an HBM miss:
                    // 1024 x 977 ids decoded in 0.53 instead of 0.50 ms)
                    if (cls[DC_B2].size() < b2_cap && len(i) <= 98304 && len(i) > B2_MIN_LIST && P >= 12 && P <= 31) cls[DC_B2].push_back(i);
                    else keep.push_back(i);
                }
                cls[c].swap(keep);
            }
        // short lists (257..4096 ids) of a call with few lists: the same loop with 32 .. 256 buckets and their 64-member rows in
        // LDS (8.3 / 16.5 / 33 / 66 KiB per chain) -- if the chains taken so far and all of these fit at 150 KiB of LDS and 16
        // wavefronts (128 VGPRs each) per CU: 1024 chains of 33 KiB, 4096 of 8.3 KiB on 256 CUs.  The bucket count
        // follows the expected load: ids are spread over (max id >> shift) + 1 buckets, half of them in the worst case when the
        // maximum is unknown (imported streams); at <= 30 ids per bucket a 64-member row practically never overflows.
        {
            auto buckets_for = [&](uint32_t i) -> uint32_t {
                const uint64_t n = len(i);
                const uint32_t l = lists[i], P = r->prec[l];
                if (n <= R2_MIN_LIST || n > VIDC_B2L_MAX_LIST || P > 31) return 0u;
                for (uint32_t bk : {32u, 64u, 128u, 256u}) {
                    const uint32_t bbits = bk == 32u ? 5u : (bk == 64u ? 6u : (bk == 128u ? 7u : 8u)), bsh = P > bbits ? P - bbits : 0u;
                    const uint32_t mx = (l < r->umax.size() && r->umax[l]) ? r->umax[l] : ((P ? (1u << (P - 1u)) : 1u));
                    const uint64_t eff = std::min<uint64_t>(bk, (uint64_t)(mx >> bsh) + 1u);
                    if (n <= VIDC_B2L_LOAD * eff) return bk;
                }
                return 0u;
            };
            size_t units = 0, n_el = 0;  // (units of 8.25 KiB of LDS: 18 per CU; a chain with its rows in memory takes one)
            bool longer_left = false;   // a longer chain that stays on the general kernel decides the call anyway
            for (int c : order_) longer_left |= !cls[c].empty();
            for (uint32_t i : cls[DC_GSMALL]) {
                const uint32_t bk = buckets_for(i);
                units += bk / 32u;
                n_el += bk != 0;
                longer_left |= !bk && len(i) > VIDC_B2L_MID_LIST;
            }
            if (n_el && !longer_left && cls[DC_B2].size() + units <= b2_cap * 9 / 2 && cls[DC_B2].size() + n_el <= b2_cap * 4) {
                std::vector<uint32_t> keep;
                for (uint32_t i : cls[DC_GSMALL]) {
                    const uint32_t bk = buckets_for(i);
                    if (bk == 32u) cls[DC_B2T].push_back(i);
                    else if (bk == 64u) cls[DC_B2S].push_back(i);
                    else if (bk == 128u) cls[DC_B2L].push_back(i);
                    else if (bk == 256u) cls[DC_B2M].push_back(i);
                    else keep.push_back(i);
                }
                cls[DC_GSMALL].swap(keep);
            }
        }
    }
    // LDS member rows for the short general lists only while all of them are resident at four per CU: beyond that the 33 KiB
    // per wavefront cost more in occupancy than the global rows cost in traffic (6000 x 300 ids: 0.86 vs 0.3 ms)
    p.gsmall_lrows = cls[DC_GSMALL].size() <= B2_CAP;
    // one pass per class over its work items: work list, per-class totals, scratch / slot offsets (65 536 lists: the two
    // passes with push_back this replaces were half of the planner's time)
    size_t total_items = 0;
    for (int c = 0; c < DC_COUNT; c++) total_items += cls[c].size();
    p.item = vec_pool<uint32_t>().take(total_items); p.item.resize(total_items);
    p.wl = vec_pool<uint32_t>().take(total_items); p.wl.resize(total_items);
    p.scratch_off = vec_pool<uint64_t>().take(total_items); p.scratch_off.resize(total_items);
    p.slots_off = vec_pool<uint64_t>().take(total_items); p.slots_off.resize(total_items);
    uint64_t so = 0, sl = 0;
    size_t k = 0;
    {   // VIDC_LANE_ALIGN=4 / 16 (measurement switch): bucket rows of the lane decoders on 16- / 64-byte boundaries
        const char *ae = std::getenv("VIDC_LANE_ALIGN");
        p.lane_align = ae ? (std::atoi(ae) >= 16 ? 16u : 4u) : VIDC_LANE_ALIGN_DEFAULT;
    }
    const uint64_t la = p.lane_align;
    const uint64_t *offs = r->offsets.data();
    const bool have_nwords = r->meta_host;
    for (int c = 0; c < DC_COUNT; c++) {
        p.count[c] = cls[c].size();
        uint64_t sum_n = 0, max_n = 0;
        // re-spill scratch of the decoder stack (== roc_dec_stack_cap in the kernels); the lane-per-list decoders
        // keep what they push in LDS
        const bool stack_scratch = c != DC_LANE && c != DC_LANE64 && c != DC_LANE128 && c < DC_GRP0;
        for (const uint32_t i : cls[c]) {
            const uint32_t l = lists[i];
            const uint64_t n = offs[l + 1] - offs[l];
            p.item[k] = i;
            p.wl[k] = l;
            sum_n += n;
            max_n = std::max(max_n, n);
            p.scratch_off[k] = so;
            if (stack_scratch) so += roc_dec_stack_cap((uint32_t)n, have_nwords ? r->nwords[l] : 0u);
            uint64_t slot = sl;
            if (c == DC_LANEP || c == DC_LANEQ) {
                slot = 0;  // no scratch of any kind
            } else if (c == DC_LANE) {
                sl = (sl + la - 1) & ~(la - 1);  // rows are read as uint4
                slot = sl;
                sl += 64ull * roc_lane_cap_nb<64>((uint32_t)n, p.lane_align);
            } else if (c == DC_LANE64) {
                sl = (sl + la - 1) & ~(la - 1);
                slot = sl;
                sl += 256ull * roc_lane_cap_nb<256>((uint32_t)n, p.lane_align);
            } else if (c == DC_LANE128) {
                sl = (sl + la - 1) & ~(la - 1);
                slot = sl;
                sl += 128ull * roc_lane_cap_nb<128>((uint32_t)n, p.lane_align);
            } else if (c >= DC_GRP0) {
                sl = (sl + 15) & ~(uint64_t)15;  // member rows of 32 .. 96 u32: 64-byte aligned
                slot = sl;
                sl += roc_grp_dec_slots((uint32_t)n);
            } else if (c == DC_U18 || c == DC_U20) {
                sl += n;  // duplicate side list
            } else if (c == DC_B2) {
                sl = (sl + 63) & ~(uint64_t)63;  // 4096 rows of 64 members
                slot = sl;
                sl += 4096ull * 64ull;
            } else if (c == DC_GSMALL && p.gsmall_lrows) {
                sl += n;  // overflow list only: the member rows are in LDS
            } else if (c >= DC_GSMALL && c <= DC_GHUGE) {
                const uint32_t fb = roc_dec_fine_bits((uint32_t)n, r->prec[l] > 32 ? 32 : r->prec[l]);
                sl += ((uint64_t)1 << fb) * roc_dec_cap((uint32_t)n) + n;
            }
            p.slots_off[k] = slot;
            k++;
        }
        p.sum_n[c] += sum_n;
        p.max_n[c] = std::max<uint64_t>(p.max_n[c], max_n);
    }
    p.scratch_words = so;
    p.slots_words = sl;
}



int decode_impl(vidc_ctx *ctx, const vidc_roc *r, const DecPlan &p, const uint64_t *out_off_host, uint64_t *d_out,
                int32_t *d_out_rows, uint32_t K, const struct DecPlanCache *cache = nullptr) {
    VIDC_HIP(hipSetDevice(ctx->device));
    const size_t nwork = p.implicit ? (size_t)p.implicit : p.wl.size();
    if (nwork == 0) { ctx->last_kernel_ms = 0; return VIDC_OK; }
    HostTrace tr("roc decode");
    Scratch s_wl, s_scr_off, s_slots_off, s_scr, s_slots, s_out_off, s_end, s_sum;
    Pinned h_up;
    const uint32_t *d_wl;
    const uint64_t *d_scr_off = nullptr, *d_slots_off = nullptr;
    if (cache) {
        d_wl = cache->d_wl.p; d_scr_off = cache->d_scratch_off.p; d_slots_off = cache->d_slots_off.p;
    } else if (p.implicit) {
        d_wl = p.order_dev;
    } else {
        // one pinned staging block for everything that goes up: work list | scratch offsets | slot offsets | out offsets
        const size_t n8 = p.lean ? 0 : nwork;
        VIDC_TRY(h_up.get(ctx, nwork * 4 + 8 + n8 * 24));
        uint32_t *h_wl = h_up.as<uint32_t>();
        uint64_t *h_64 = (uint64_t *)((char *)h_up.p + ((nwork * 4 + 7) & ~(size_t)7));
        std::memcpy(h_wl, p.wl.data(), nwork * 4);
        VIDC_TRY(s_wl.get(ctx, nwork * 4));
        VIDC_HIP(hipMemcpyAsync(s_wl.p, h_wl, nwork * 4, hipMemcpyHostToDevice, ctx->stream));
        d_wl = s_wl.as<uint32_t>();
        if (!p.lean) {
            std::memcpy(h_64, p.scratch_off.data(), nwork * 8);
            std::memcpy(h_64 + nwork, p.slots_off.data(), nwork * 8);
            VIDC_TRY(s_scr_off.get(ctx, nwork * 8));
            VIDC_TRY(s_slots_off.get(ctx, nwork * 8));
            VIDC_HIP(hipMemcpyAsync(s_scr_off.p, h_64, nwork * 8, hipMemcpyHostToDevice, ctx->stream));
            VIDC_HIP(hipMemcpyAsync(s_slots_off.p, h_64 + nwork, nwork * 8, hipMemcpyHostToDevice, ctx->stream));
            d_scr_off = s_scr_off.as<uint64_t>(); d_slots_off = s_slots_off.as<uint64_t>();
            if (out_off_host) {
                std::memcpy(h_64 + 2 * nwork, out_off_host, nwork * 8);
                VIDC_TRY(s_out_off.get(ctx, nwork * 8));
                VIDC_HIP(hipMemcpyAsync(s_out_off.p, h_64 + 2 * nwork, nwork * 8, hipMemcpyHostToDevice, ctx->stream));
            }
        }
    }
    if (cache && out_off_host) {
        VIDC_TRY(h_up.get(ctx, nwork * 8));
        std::memcpy(h_up.p, out_off_host, nwork * 8);
        VIDC_TRY(s_out_off.get(ctx, nwork * 8));
        VIDC_HIP(hipMemcpyAsync(s_out_off.p, h_up.p, nwork * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    VIDC_TRY(s_scr.get(ctx, p.scratch_words * 4));
    VIDC_TRY(s_slots.get(ctx, p.slots_words * 4));
    VIDC_TRY(s_end.get(ctx, r->nlist * 8 + 64));  // end states | statuses | summary accumulators: one block, one memset
    VIDC_HIP(hipMemsetAsync(s_end.p, 0, r->nlist * 8 + 64, ctx->stream));
    uint32_t *const d_status = s_end.as<uint32_t>() + r->nlist;
    tr.mark("scratch + uploads");
    RocDecArgs a{};
    a.offsets = r->d_offsets.p;
    a.heads = r->d_heads.p; a.prec = r->d_prec.p; a.nwords = r->d_nwords.p; a.draws = r->d_draws.p;
    a.words = r->d_words.p; a.word_off = r->d_word_off.p;
    a.out = d_out; a.out_rows = d_out_rows; a.K = K;
    a.out_by_list = (p.implicit && p.order_dev) ? 1u : 0u;
    a.scratch_words = s_scr.as<uint32_t>();
    a.slots = s_slots.as<uint32_t>();
    a.end_state = s_end.as<uint32_t>(); a.status = d_status;
    a.mt = ctx->d_mt;

    size_t base[DC_COUNT];
    {
        size_t acc = 0;
        for (int c = 0; c < DC_COUNT; c++) { base[c] = acc; acc += p.count[c]; }
    }
    EventTimer t(ctx);
    if (p.order_build) {  // rows by edge count, most edges first: histogram per workgroup, segment starts, scatter
        uint32_t *part = const_cast<uint32_t *>(p.order_dev) + nwork;
        hipLaunchKernelGGL(k_rows_order_hist, dim3(VIDC_ORDER_BLOCKS), dim3(256), 0, ctx->stream, r->d_offsets.p, (uint32_t)nwork, part);
        hipLaunchKernelGGL(k_rows_order_starts, dim3(1), dim3(128), 0, ctx->stream, part);
        hipLaunchKernelGGL(k_rows_order_scatter, dim3(VIDC_ORDER_BLOCKS), dim3(256), 0, ctx->stream, r->d_offsets.p, (uint32_t)nwork, part,
                           const_cast<uint32_t *>(p.order_dev));
        VIDC_HIP(hipGetLastError());
    }
    // classes run concurrently: the longest chains on the caller's stream, the rest on the auxiliary streams (fork below)
    // The lane-pair register decoder goes FIRST and alone when the call has other work for the whole machine: its wavefronts
    // need 256 VGPRs, and next to memory-bound classes that fill the SIMDs with small wavefronts they wait for register
    // space -- 6 ms alone on S2's 217 M ids, 73 ms as the tail of the call when everything starts at once.
    // (measured on S2: 92-95 ms of decode against 86-87 with everything started at once -- the other classes take their ~85 ms
    // whether or not these lists are among them; opt-in for measurements)
    const bool pair_first = p.count[DC_LANEP] && false;  // (the pair class first: measured no better, HISTORY.md)
    // Stream assignment: a class's kernel lasts about max(its longest chain, its share of the machine); classes are
    // taken longest first and each goes to the stream that frees up first (LPT over 3 streams).  A fixed
    // assignment left three general classes back to back on one stream on S2 (70 + 45 + 18 ms) while the
    // others had drained.
    int order[DC_COUNT];
    hipStream_t stream_of[DC_COUNT];
    {
        double est[DC_COUNT];
        for (int c = 0; c < DC_COUNT; c++) {
            order[c] = c;
            const bool u = c == DC_U18 || c == DC_U20 || c == DC_B2 || c == DC_B2T || c == DC_B2S || c == DC_B2L || c == DC_B2M, lane = c == DC_LANE || c == DC_LANE64 || c == DC_LANE128 || c == DC_LANEP || c == DC_LANEQ;
            const bool grpc = c >= DC_GRP0 && c <= DC_GRP4;
            // (constants fitted to the S2 timeline: general kernels ~2.5 G steps/s while they share the machine)
            const double step_us = u ? 0.4 : (lane ? 2.5 : (c == DC_TINY ? 0.5 : (grpc ? 1.5 : 1.2)));       // one chain step
            const double rate = u ? 0.6e3 : (lane ? 16e3 : (c == DC_TINY ? 30e3 : (grpc ? 12e3 : 2.5e3)));     // steps / us, all CUs
            est[c] = std::max((double)p.max_n[c] * step_us, (double)p.sum_n[c] / rate);
        }
        std::sort(order, order + DC_COUNT, [&](int x, int y) { return est[x] > est[y]; });
        // three streams: a 4th one did not run concurrently (HIP maps streams onto 4 hardware queues and the
        // process has other streams; the kernels of aux[2] started when the main stream's had finished)
        double load[VIDC_NAUX + 1] = {};
        // (wide: a stream per class.  S2 decode by stream count, final kernel set (row decoder up to 16 384 ids, lane-pair register
        // decoder): 78-79 ms with 8, 84-85 with 6 -- a class queued behind the lane-pair launch, whose 256-VGPR wavefronts are slow
        // to find room, becomes the tail; without that decoder 81-82 / 83-84)
        int nq = ctx->wide ? VIDC_NAUX + 1 : 3;
        if (const char *e = std::getenv("VIDC_DEC_NQ")) nq = std::max(1, std::min(ctx->naux() + 1, std::atoi(e)));
        if (env_on("VIDC_SERIAL")) nq = 1;  // measurements: one class after the other
        for (int k = 0; k < DC_COUNT; k++) {
            const int c = order[k];
            if (!p.count[c]) continue;
            if (pair_first && c == DC_LANEP) { stream_of[c] = ctx->stream; continue; }
            int best = 0;
            for (int q = 1; q < nq; q++)
                if (load[q] < load[best]) best = q;
            stream_of[c] = best == 0 ? ctx->stream : ctx->aux[best - 1];
            load[best] += est[c] + 1.0;
        }
    }
    ctx->chain_info[1][0] = ctx->chain_info[1][1] = ctx->chain_info[1][2] = ctx->chain_info[1][3] = 0;
    ctx->phase_ms[VIDC_PHASE_ROC_DECODE_CHAIN] = 0;
    const int chain_class = p.count[DC_U20] ? DC_U20 : (p.count[DC_U18] ? DC_U18 : -1);
    auto launch = [&](int c) -> int {
        if (!p.count[c]) return VIDC_OK;
        hipStream_t st_ = stream_of[c];
        if (c == chain_class) {
            ctx->chain_info[1][0] = p.sum_n[c]; ctx->chain_info[1][1] = p.count[c]; ctx->chain_info[1][2] = p.max_n[c];
            ctx->chain_info[1][3] = c == DC_U20 ? 20 : 18;
            VIDC_HIP(hipEventRecord(ctx->ev_chain[0], st_));
        }
        RocDecArgs b = a;
        b.worklist = d_wl ? d_wl + base[c] : nullptr;
        b.nwork = (uint32_t)p.count[c];
        b.out_off = (out_off_host && !p.lean) ? s_out_off.as<uint64_t>() + base[c] : nullptr;
        b.scratch_off = d_scr_off ? d_scr_off + base[c] : nullptr;
        b.slots_off = d_slots_off ? d_slots_off + base[c] : nullptr;
        b.row_align = p.lane_align;
        switch (c) {
            case DC_TINY:
                if (p.tiny_lane) {  // one list per lane (roc_lane.h)
                    const dim3 grid((b.nwork + 63u) / 64u);
                    const LaneDiv *dt = (const LaneDiv *)ctx->d_ltab;
                    if (d_out_rows) hipLaunchKernelGGL(k_roc_decode_tiny_lane<true>, grid, dim3(64), 0, st_, b, dt);
                    else hipLaunchKernelGGL(k_roc_decode_tiny_lane<false>, grid, dim3(64), 0, st_, b, dt);
                } else if (d_out_rows) hipLaunchKernelGGL(k_roc_decode_tiny<true>, dim3(b.nwork), dim3(64), 0, st_, b);
                else hipLaunchKernelGGL(k_roc_decode_tiny<false>, dim3(b.nwork), dim3(64), 0, st_, b);
                break;
            case DC_U18:
                if (old_u_kernels()) hipLaunchKernelGGL(k_roc_decode_u<18>, dim3(b.nwork), dim3(64), UGeom<18>::LDS_BYTES, st_, b);
                else hipLaunchKernelGGL(k_roc_decode_u2<18>, dim3(b.nwork), dim3(64), U2Geom<18>::LDS_BYTES, st_, b,
                                        (const U2Div *)ctx->d_u2tab);
                break;
            case DC_U20:
                if (old_u_kernels()) {
                    VIDC_TRY(set_big_lds((const void *)k_roc_decode_u<20>, UGeom<20>::LDS_BYTES));
                    hipLaunchKernelGGL(k_roc_decode_u<20>, dim3(b.nwork), dim3(64), UGeom<20>::LDS_BYTES, st_, b);
                } else {
                    VIDC_TRY(set_big_lds((const void *)k_roc_decode_u2<20>, U2Geom<20>::LDS_BYTES));
                    hipLaunchKernelGGL(k_roc_decode_u2<20>, dim3(b.nwork), dim3(64), U2Geom<20>::LDS_BYTES, st_, b,
                                       (const U2Div *)ctx->d_u2tab);
                }
                break;
            case DC_GSMALL:
                if (p.gsmall_lrows)  // member rows in LDS: 1 + 32 KiB
                    hipLaunchKernelGGL((k_roc_decode_gen<uint16_t, true>), dim3(b.nwork), dim3(64), 512 * 2 + 512 * VIDC_DEC_CAP * 4, st_,
                                       b, 512u, VIDC_DEC_CAP);
                else
                    hipLaunchKernelGGL(k_roc_decode_gen<uint16_t>, dim3(b.nwork), dim3(64), 512 * 2, st_, b, 512u, VIDC_DEC_CAP);
                break;
            case DC_LANE: {  // 12.5 KiB of LDS per wavefront
                b.lpw = lane_lists_per_wave(ctx, b.nwork, 24);
                // lists are sorted longest first: those of <= 256 ids (the tail of the work list, from a wavefront
                // boundary on) take the decoder that keeps the decoded ids in registers
                uint32_t n_big = b.nwork;
                if (!std::getenv("VIDC_NO_LANE_REG") && !p.wl.empty()) {
                    const uint32_t *wl0 = p.wl.data() + base[c];
                    n_big = (uint32_t)(std::partition_point(wl0, wl0 + b.nwork, [&](uint32_t l) {
                                           return r->offsets[l + 1] - r->offsets[l] > VIDC_LANE_REG_MAX;
                                       }) - wl0);
                    n_big = std::min<uint32_t>(b.nwork, (n_big + b.lpw - 1u) / b.lpw * b.lpw);
                }
                RocDecArgs b2 = b;
                b2.worklist = b.worklist + n_big; b2.nwork = b.nwork - n_big;
                if (b2.out_off) b2.out_off += n_big;
                b.nwork = n_big;
                if (b.nwork && env_on("VIDC_LANE_LOOP"))
                    hipLaunchKernelGGL((k_roc_decode_lane<64, false>), dim3(lane_grid((b.nwork + b.lpw - 1u) / b.lpw)), dim3(64), 0, st_, b,
                                       (const LaneDiv *)ctx->d_ltab);
                else if (b.nwork)
                    hipLaunchKernelGGL((k_roc_decode_lane<64, true>), dim3(lane_grid((b.nwork + b.lpw - 1u) / b.lpw)), dim3(64), 0, st_, b,
                                       (const LaneDiv *)ctx->d_ltab);
                if (b2.nwork)
                    hipLaunchKernelGGL(k_roc_decode_lane_reg<VIDC_LANE_REG_EL>, dim3((b2.nwork + b2.lpw - 1u) / b2.lpw), dim3(64), 0, st_,
                                       b2, (const LaneDiv *)ctx->d_ltab);
                break;
            }
            case DC_LANE64:  // 26.5 KiB of LDS per wavefront
                b.lpw = lane_lists_per_wave(ctx, b.nwork, 7);
                if (env_on("VIDC_LANE_LOOP"))
                    hipLaunchKernelGGL((k_roc_decode_lane<256, false>), dim3(lane_grid((b.nwork + b.lpw - 1u) / b.lpw)), dim3(64), 0, st_, b,
                                       (const LaneDiv *)ctx->d_ltab);
                else
                    hipLaunchKernelGGL((k_roc_decode_lane<256, true>), dim3(lane_grid((b.nwork + b.lpw - 1u) / b.lpw)), dim3(64), 0, st_, b,
                                       (const LaneDiv *)ctx->d_ltab);
                break;
            case DC_LANE128:  // 18.5 KiB of LDS per wavefront
                b.lpw = lane_lists_per_wave(ctx, b.nwork, 7);
                hipLaunchKernelGGL((k_roc_decode_lane<128, true>), dim3(lane_grid((b.nwork + b.lpw - 1u) / b.lpw)), dim3(64), 0, st_, b,
                                   (const LaneDiv *)ctx->d_ltab);
                break;
            case DC_B2: {
                // (round 5 measured and dropped: the next step's candidate rows requested one step ahead -- 45.7 against 47.5 ms with
                // the class alone, 72.9 against 72.2 ms for the whole S2 decode: HISTORY.md)
                // VIDC_B2_MASK=1 / 0: the row load under exec = lanes below the bucket's member count (roc_u2.h, U2B_DEC_IDX_MC: a third
                // of the sectors, a later load) / the whole row at once.  Default: calls of many chains, whose rows come from HBM.
                const char *mke = std::getenv("VIDC_B2_MASK");
                const bool mk = mke ? mke[0] == '1' : b.nwork >= B2_MASK_MIN_CHAINS;
                if (mk) hipLaunchKernelGGL((k_roc_decode_b2<0, 2>), dim3(b.nwork), dim3(64), VIDC_B2_LDS_BYTES, st_, b, (const U2Div *)ctx->d_u2tab);
                else hipLaunchKernelGGL(k_roc_decode_b2<0>, dim3(b.nwork), dim3(64), VIDC_B2_LDS_BYTES, st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            }
            case DC_B2T:  // 32 buckets, 8.3 KiB of LDS
                hipLaunchKernelGGL(k_roc_decode_b2<32>, dim3(b.nwork), dim3(64), VIDC_B2L_LDS_BYTES(32u), st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            case DC_B2S:  // 64 buckets, 16.5 KiB of LDS
                hipLaunchKernelGGL(k_roc_decode_b2<64>, dim3(b.nwork), dim3(64), VIDC_B2L_LDS_BYTES(64u), st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            case DC_B2L:  // 128 buckets, 33 KiB of LDS
                hipLaunchKernelGGL(k_roc_decode_b2<128>, dim3(b.nwork), dim3(64), VIDC_B2L_LDS_BYTES(128u), st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            case DC_B2M:  // 256 buckets, 66 KiB
                VIDC_TRY(set_big_lds((const void *)k_roc_decode_b2<256>, VIDC_B2L_LDS_BYTES(256u)));
                hipLaunchKernelGGL(k_roc_decode_b2<256>, dim3(b.nwork), dim3(64), VIDC_B2L_LDS_BYTES(256u), st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            case DC_LANEQ:
                b.lpw = 16u;
                hipLaunchKernelGGL((k_roc_decode_lane_reg<VIDC_LANE_REG_EL, 4>), dim3((b.nwork + 15u) / 16u), dim3(64), 0, st_, b,
                                   (const LaneDiv *)ctx->d_ltab);
                break;
            case DC_LANEP:
                b.lpw = 32u;
                hipLaunchKernelGGL((k_roc_decode_lane_reg<VIDC_LANE_REG_EL, 2>), dim3((b.nwork + 31u) / 32u), dim3(64), 0, st_, b,
                                   (const LaneDiv *)ctx->d_ltab);
                break;
            case DC_GRP0:
                hipLaunchKernelGGL(k_roc_decode_grp<0>, dim3((b.nwork + 3u) / 4u), dim3(64), 16u * roc_grp_dec_lds_words(0), st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            case DC_GRP2:
                hipLaunchKernelGGL(k_roc_decode_grp<2>, dim3((b.nwork + 3u) / 4u), dim3(64), 16u * roc_grp_dec_lds_words(2), st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            case DC_GRP3:
                hipLaunchKernelGGL(k_roc_decode_grp<3>, dim3((b.nwork + 3u) / 4u), dim3(64), 16u * roc_grp_dec_lds_words(3), st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            case DC_GRP4:
                hipLaunchKernelGGL(k_roc_decode_grp<4>, dim3((b.nwork + 3u) / 4u), dim3(64), 16u * roc_grp_dec_lds_words(4), st_, b, (const U2Div *)ctx->d_u2tab);
                break;
            case DC_G8K:
                hipLaunchKernelGGL(k_roc_decode_gen<uint16_t>, dim3(b.nwork), dim3(64), 1024 * 2, st_, b, 1024u, VIDC_DEC_CAP);
                break;
            case DC_G16K:
                hipLaunchKernelGGL(k_roc_decode_gen<uint16_t>, dim3(b.nwork), dim3(64), 2048 * 2, st_, b, 2048u, VIDC_DEC_CAP);
                break;
            case DC_GMID:
                hipLaunchKernelGGL(k_roc_decode_gen<uint16_t>, dim3(b.nwork), dim3(64), (1u << VIDC_DEC_MAX_FB) * 2, st_, b,
                                   1u << VIDC_DEC_MAX_FB, VIDC_DEC_CAP);
                break;
            default:  // DC_GHUGE: 16-bit row counters unless the class holds a list beyond 65 536 ids
                if (p.max_n[DC_GHUGE] <= 65536)
                    hipLaunchKernelGGL(k_roc_decode_gen<uint16_t>, dim3(b.nwork), dim3(64), (1u << VIDC_DEC_MAX_FB) * 2, st_,
                                       b, 1u << VIDC_DEC_MAX_FB, VIDC_DEC_CAP_BIG);
                else
                    hipLaunchKernelGGL(k_roc_decode_gen<uint32_t>, dim3(b.nwork), dim3(64), (1u << VIDC_DEC_MAX_FB) * 4, st_,
                                       b, 1u << VIDC_DEC_MAX_FB, VIDC_DEC_CAP_BIG);
        }
        VIDC_HIP(hipGetLastError());
        if (c == chain_class) VIDC_HIP(hipEventRecord(ctx->ev_chain[1], st_));
        return VIDC_OK;
    };
    // VIDC_DEC_SCHED (measurements): explicit schedule.  Streams separated by ';' (first = the caller's stream, then the
    // auxiliary ones), the classes of a stream separated by ',' run in that order; "CLS^DEP" additionally waits for class DEP
    // (on another stream) to finish.  Launch order on the host: first entries of every stream, then the second ones, ...
    // Classes of the call the string does not name keep the automatic assignment.  E.g. "B2;GHUGE;LANEP,LANE,LANE64;GMID^LANEP".
    struct SchedItem { int cls, grp, pos; std::vector<int> deps; };
    std::vector<SchedItem> sched;
    if (const char *e = std::getenv("VIDC_DEC_SCHED")) {
        static const char *names[DC_COUNT] = {"TINY", "U18", "U20", "GSMALL", "G8K", "G16K", "GMID", "GHUGE", "LANE", "LANE64", "LANE128", "B2", "B2T", "B2S",
                                              "B2L", "B2M", "GRP0", "GRP2", "GRP3", "GRP4", "LANEP", "LANEQ"};
        auto cls_of = [&](const std::string &t) { for (int c = 0; c < DC_COUNT; c++) if (t == names[c]) return c; return -1; };
        std::string str(e);
        int grp = 0, pos = 0;
        size_t i0 = 0;
        for (size_t i = 0; i <= str.size(); i++) {
            if (i < str.size() && str[i] != ',' && str[i] != ';') continue;
            std::string tok = str.substr(i0, i - i0);
            i0 = i + 1;
            if (!tok.empty()) {
                SchedItem it{-1, grp, pos, {}};
                size_t j0 = 0;
                bool first = true;
                for (size_t j = 0; j <= tok.size(); j++) {
                    if (j < tok.size() && tok[j] != '^') continue;
                    const int c = cls_of(tok.substr(j0, j - j0));
                    j0 = j + 1;
                    if (first) it.cls = c; else if (c >= 0) it.deps.push_back(c);
                    first = false;
                }
                if (it.cls >= 0 && p.count[it.cls] && grp <= ctx->naux()) { sched.push_back(it); pos++; }
            }
            if (i < str.size() && str[i] == ';') { grp++; pos = 0; }
        }
        for (const SchedItem &it : sched) stream_of[it.cls] = it.grp == 0 ? ctx->stream : ctx->aux[it.grp - 1];
    }
    if (pair_first) VIDC_TRY(launch(DC_LANEP));
    VIDC_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
    uint32_t aux_used = 0;  // only the auxiliary streams that carry a class of this call are forked and joined
    for (int c = 0; c < DC_COUNT; c++)
        for (int i = 0; i < ctx->naux(); i++)
            if (p.count[c] && stream_of[c] == ctx->aux[i] && !(aux_used >> i & 1u)) {
                VIDC_HIP(hipStreamWaitEvent(ctx->aux[i], ctx->ev_fork, 0));
                aux_used |= 1u << i;
            }
    if (!sched.empty()) {
        bool done[DC_COUNT] = {};
        std::stable_sort(sched.begin(), sched.end(), [](const SchedItem &x, const SchedItem &y) { return x.pos < y.pos; });
        size_t left = sched.size();
        while (left) {  // (an entry whose dependency is launched later in the list waits for the next round)
            size_t before = left;
            for (SchedItem &it : sched) {
                if (it.cls < 0 || done[it.cls]) continue;
                bool ready = true;
                for (int d : it.deps) ready &= !p.count[d] || done[d];
                if (!ready) continue;
                for (int d : it.deps)
                    if (p.count[d]) VIDC_HIP(hipStreamWaitEvent(stream_of[it.cls], ctx->tev[d], 0));
                VIDC_TRY(launch(it.cls));
                VIDC_HIP(hipEventRecord(ctx->tev[it.cls], stream_of[it.cls]));
                done[it.cls] = true;
                left--;
            }
            if (left == before) break;  // circular dependencies: the rest goes out below, unordered
        }
        for (int k = 0; k < DC_COUNT; k++)
            if (!done[order[k]] && !(pair_first && order[k] == DC_LANEP)) VIDC_TRY(launch(order[k]));
    } else
    for (int k = 0; k < DC_COUNT; k++)  // longest first
        if (!(pair_first && order[k] == DC_LANEP)) VIDC_TRY(launch(order[k]));
    for (int i = 0; i < ctx->naux(); i++) {
        if (!(aux_used >> i & 1u)) continue;
        VIDC_HIP(hipEventRecord(ctx->ev_join[i], ctx->aux[i]));
        VIDC_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join[i], 0));
    }
    // the end event of the kernels, then the 32-byte status summary (instead of copying two nlist-sized arrays back) behind them:
    // ONE wait for both
    t.mark();
    Pinned h_sum;
    constexpr uint32_t RETRY_SLOTS = 504;  // list numbers of handed-back lists behind the eight summary words
    VIDC_TRY(h_sum.get(ctx, (8 + RETRY_SLOTS) * 8));
    unsigned long long *sum = h_sum.as<unsigned long long>();
    sum[0] = ~0ull; sum[1] = 0; sum[2] = 0; sum[3] = 0;
    // (the kernel's last workgroup stores the summary into the pinned block itself: no copy up, no copy down)
    hipLaunchKernelGGL(k_roc_status_summary_host, dim3((uint32_t)std::min<uint64_t>((r->nlist + 1023) / 1024, 256)), dim3(256),
                       0, ctx->stream, d_status, s_end.as<uint32_t>(), (uint32_t)r->nlist,
                       (unsigned long long *)(s_end.as<uint32_t>() + 2 * r->nlist), sum, (out_off_host || r->rows) ? 0u : RETRY_SLOTS);
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    ctx->last_kernel_ms = t.elapsed();
    ctx->phase_ms[VIDC_PHASE_ROC_DECODE] = ctx->last_kernel_ms;
    if (chain_class >= 0) {
        float cms = 0;
        if (hipEventElapsedTime(&cms, ctx->ev_chain[0], ctx->ev_chain[1]) == hipSuccess) ctx->phase_ms[VIDC_PHASE_ROC_DECODE_CHAIN] = cms;
    }
    tr.mark("decode kernels + status summary (sync)");
    const double first_ms = ctx->last_kernel_ms;
    uint64_t nonclean = sum[1];
    if (sum[0] != ~0ull || sum[2]) {
        // (a few handed-back lists and nothing else to report: their numbers came with the summary; otherwise the statuses are fetched)
        const bool listed = sum[0] == ~0ull && !out_off_host && !r->rows && sum[2] <= RETRY_SLOTS;
        std::vector<uint32_t> status(listed ? 0 : r->nlist);
        if (!listed) VIDC_HIP(hipMemcpy(status.data(), d_status, r->nlist * 4, hipMemcpyDeviceToHost));
        if (sum[2]) {
            // lists the lane-per-list decoder handed back (a full bucket row on skewed ids): redo them with the
            // wave-per-list kernels, into the same output slots
            std::vector<uint32_t> lists2;
            std::vector<uint64_t> off2;
            if (listed) {
                for (uint64_t k = 0; k < sum[2]; k++) lists2.push_back((uint32_t)sum[8 + k]);
                std::sort(lists2.begin(), lists2.end());
                for (uint32_t l : lists2) off2.push_back(r->offsets[l]);
            } else
            for (size_t k = base[DC_LANE]; k < p.wl.size(); k++) {  // lane classes, B2, row-per-list classes
                if (status[p.wl[k]] != VIDC_ST_RETRY) continue;
                lists2.push_back(p.wl[k]);
                off2.push_back(out_off_host ? out_off_host[k] : r->offsets[p.wl[k]]);
            }
            if (!listed) for (uint32_t l : lists2) status[l] = VIDC_ST_OK;
            if (tr.on) {
                uint64_t mx = 0, mn = ~0ull;
                for (uint32_t l : lists2) { const uint64_t n = r->offsets[l + 1] - r->offsets[l]; mx = std::max(mx, n); mn = std::min(mn, n); }
                fprintf(stderr, "[vidc] roc decode: %zu lists handed back by the lane decoders (%llu .. %llu ids)\n", lists2.size(),
                        (unsigned long long)mn, (unsigned long long)mx);
            }
            if (!listed) VIDC_TRY(check_status(status, "roc decode"));
            DecPlan p2;
            plan_decode(r, lists2, false, p2, ctx->wide, false, false);
            std::vector<uint64_t> out_off2(lists2.size());
            for (size_t k = 0; k < lists2.size(); k++) out_off2[k] = off2[p2.item[k]];
            VIDC_TRY(decode_impl(ctx, r, p2, out_off2.data(), d_out, nullptr, 0));
            nonclean += r->last_nonclean;
            ctx->last_kernel_ms += first_ms;
            ctx->phase_ms[VIDC_PHASE_ROC_DECODE] = ctx->last_kernel_ms;
        } else {
            VIDC_TRY(check_status(status, "roc decode"));
        }
    }
    r->last_nonclean = nonclean;
    return VIDC_OK;
}

}  // namespace

extern "C" {

int vidc_roc_encode(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids,
                    int precision_mode, uint32_t flags, vidc_roc **out) {
    if (nlist && offsets && offsets[nlist] > offsets[0] && !d_ids) return VIDC_ERR_INVALID;
    return encode_impl(ctx, nlist, offsets, d_ids, false, 0, 0, nullptr, precision_mode, flags, out);
}

int vidc_roc_encode_rows(vidc_ctx *ctx, uint64_t N, uint32_t K, const int32_t *d_rows, int precision_mode,
                         uint32_t flags, vidc_roc **out) {
    if (N && !d_rows) return VIDC_ERR_INVALID;
    if (K > TINY_MAX && ctx && out) {
        // wide rows (NSG128, NSG256, ...): the rows become CSR lists and take the per-list kernels; the object answers
        // decode_rows like a graph object (sizes = edge counts, altid_impl.h:61)
        if (N >= 0xffffffffull) { set_error("too many nodes"); return VIDC_ERR_INVALID; }
        VIDC_HIP(hipSetDevice(ctx->device));
        std::vector<uint64_t> offsets;
        Scratch s_ids;
        VIDC_TRY(rows_to_csr(ctx, N, K, d_rows, offsets, s_ids));
        VIDC_TRY(encode_impl(ctx, N, offsets.data(), s_ids.as<uint64_t>(), false, 0, 0, nullptr, precision_mode,
                             flags & ~VIDC_ROC_WANT_PERM, out));
        (*out)->K = K;
        return VIDC_OK;
    }
    return encode_impl(ctx, 0, nullptr, nullptr, true, N, K, d_rows, precision_mode, flags, out);
}

void vidc_roc_destroy(vidc_roc *r) { delete r; }

uint64_t vidc_roc_nlist(const vidc_roc *r) { return r ? r->nlist : 0; }
uint64_t vidc_roc_ntotal(const vidc_roc *r) { return r ? r->ntotal : 0; }
uint64_t vidc_roc_compressed_bytes(const vidc_roc *r) { return r ? r->compressed_bytes : 0; }
uint64_t vidc_roc_total_words(const vidc_roc *r) { return r ? r->total_words : 0; }
uint64_t vidc_roc_last_decode_nonclean(const vidc_roc *r) { return r ? r->last_nonclean : 0; }

int vidc_roc_list_info(const vidc_roc *r, uint32_t *sizes, uint32_t *precisions, uint64_t *heads,
                       uint32_t *nwords, uint32_t *mt_draws) {
    if (!r) return VIDC_ERR_INVALID;
    VIDC_TRY(ensure_meta(r));
    for (uint64_t l = 0; l < r->nlist; l++) {
        if (sizes) sizes[l] = (uint32_t)(r->offsets[l + 1] - r->offsets[l]);
        if (precisions) precisions[l] = r->prec[l];
        if (heads) heads[l] = r->heads[l];
        if (nwords) nwords[l] = r->nwords[l];
        if (mt_draws) mt_draws[l] = r->draws[l];
    }
    return VIDC_OK;
}

int vidc_roc_export_words(vidc_ctx *ctx, const vidc_roc *r, uint64_t list_no, uint32_t *words, size_t cap) {
    if (!ctx || !r || list_no >= r->nlist) return VIDC_ERR_INVALID;
    VIDC_TRY(ensure_meta(r));
    uint64_t nw = r->nwords[list_no];
    if (nw > cap) { set_error("export buffer too small (%zu < %llu words)", cap, (unsigned long long)nw); return VIDC_ERR_INVALID; }
    return vidc_copy_d2h(ctx, words, r->d_words.p + r->word_off[list_no], nw * 4);
}

int vidc_roc_export_all_words(vidc_ctx *ctx, const vidc_roc *r, uint32_t *words, size_t cap) {
    if (!ctx || !r || (r->total_words && !words)) return VIDC_ERR_INVALID;
    if (r->total_words > cap) { set_error("export buffer too small (%zu < %llu words)", cap, (unsigned long long)r->total_words); return VIDC_ERR_INVALID; }
    return vidc_copy_d2h(ctx, words, r->d_words.p, r->total_words * 4);
}

int vidc_roc_perm(vidc_ctx *ctx, const vidc_roc *r, uint32_t *perm_host) {
    if (!ctx || !r || !perm_host) return VIDC_ERR_INVALID;
    if (!r->d_perm.p) { set_error("encode was called without VIDC_ROC_WANT_PERM"); return VIDC_ERR_INVALID; }
    return vidc_copy_d2h(ctx, perm_host, r->d_perm.p, r->ntotal * 4);
}
const uint32_t *vidc_roc_perm_dev(const vidc_roc *r) { return r ? r->d_perm.p : nullptr; }

int vidc_roc_import(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint32_t *precisions,
                    const uint64_t *heads, const uint32_t *nwords, const uint32_t *mt_draws,
                    const uint32_t *words_concat, vidc_roc **out) {
    if (!ctx || !out) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (nlist && (!offsets || !precisions || !heads || !nwords)) {
        set_error("roc import: offsets, precisions, heads and nwords are required");
        return VIDC_ERR_INVALID;
    }
    if (nlist >= 0xffffffffull) { set_error("too many lists"); return VIDC_ERR_INVALID; }
    // the image is untrusted (.npz content): geometry and per-list state are checked before anything reaches a kernel
    uint64_t tw = 0;
    for (uint64_t l = 0; l < nlist; l++) {
        if (offsets[l + 1] < offsets[l]) { set_error("roc import: offsets not monotone at list %llu", (unsigned long long)l); return VIDC_ERR_INVALID; }
        const uint64_t n = offsets[l + 1] - offsets[l];
        if (mt_draws && mt_draws[l] > VIDC_MT_TABLE) {
            set_error("roc import: list %llu claims %u mt19937 draws (table holds %d)", (unsigned long long)l, mt_draws[l], VIDC_MT_TABLE);
            return VIDC_ERR_INVALID;
        }
        if (n == 0 && nwords[l] != 0) { set_error("roc import: empty list %llu with %u stream words", (unsigned long long)l, nwords[l]); return VIDC_ERR_INVALID; }
        tw += nwords[l];
    }
    if (tw && !words_concat) { set_error("roc import: %llu stream words announced but words == NULL", (unsigned long long)tw); return VIDC_ERR_INVALID; }
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_roc> r(new vidc_roc());
    r->device = ctx->device;
    r->nlist = nlist;
    if (nlist) r->offsets.assign(offsets, offsets + nlist + 1); else r->offsets.assign(1, 0);
    r->ntotal = nlist ? offsets[nlist] - offsets[0] : 0;
    if (nlist && offsets[0] != 0) { set_error("roc import: offsets[0] must be 0"); return VIDC_ERR_INVALID; }
    r->prec.assign(precisions, precisions + nlist);
    r->heads.assign(heads, heads + nlist);
    r->nwords.assign(nwords, nwords + nlist);
    if (mt_draws) r->draws.assign(mt_draws, mt_draws + nlist); else r->draws.assign(nlist, 0);
    r->word_off.assign(nlist + 1, 0);
    for (uint64_t l = 0; l < nlist; l++) {
        if (r->prec[l] > 32) { set_error("list %llu: precision %u unsupported", (unsigned long long)l, r->prec[l]); return VIDC_ERR_INVALID; }
        uint64_t n = r->offsets[l + 1] - r->offsets[l];
        if (n > VIDC_ROC_MAX_LIST) { set_error("list %llu too long", (unsigned long long)l); return VIDC_ERR_DOMAIN; }
        r->word_off[l + 1] = r->word_off[l] + r->nwords[l];
        if (n) r->compressed_bytes += 8 + 4ull * r->nwords[l];
    }
    r->total_words = r->word_off[nlist];
    VIDC_TRY(upload(ctx, r->d_offsets, r->offsets));
    VIDC_TRY(upload(ctx, r->d_prec, r->prec));
    VIDC_TRY(upload(ctx, r->d_heads, r->heads));
    VIDC_TRY(upload(ctx, r->d_nwords, r->nwords));
    VIDC_TRY(upload(ctx, r->d_draws, r->draws));
    VIDC_TRY(upload(ctx, r->d_word_off, r->word_off));
    VIDC_TRY(r->d_words.alloc(r->total_words + 16, ctx->dpool));  // + padding: the look-ahead of the lane / row decoders reads up to 15 words past the last stream
    if (r->total_words)
        VIDC_HIP(hipMemcpyAsync(r->d_words.p, words_concat, r->total_words * 4, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));
    r->meta_host = true;
    r->offsets_host = true;
    *out = r.release();
    return VIDC_OK;
}

namespace {
std::shared_ptr<DecPlanCache> plan_ahead_build(const vidc_roc *r, bool wide) {
    std::vector<uint32_t> all(r->nlist);
    std::iota(all.begin(), all.end(), 0u);
    auto c = std::make_shared<DecPlanCache>();
    c->wide = wide;
    plan_decode(r, all, false, c->plan, wide, true, true, false, /*whole_sorted=*/true);
    return c;
}
}  // namespace

int vidc_roc_decode_all(vidc_ctx *ctx, const vidc_roc *r, uint64_t *d_out) {
    if (!ctx || !r || (r->ntotal && !d_out)) return VIDC_ERR_INVALID;
    VIDC_TRY(r->prec.size() == r->nlist ? ensure_offsets(r) : ensure_meta(r));
    std::shared_ptr<DecPlanCache> plan;
    {
        std::lock_guard<std::mutex> g(r->mu);
        plan = r->plan_all;
    }
    if (plan && plan->wide != ctx->wide) {
        // the cached plan is for the other stream mode (8 against 4 class streams): this call plans for its own context and
        // leaves the cache to the contexts of the mode that built it
        HostTrace tr("roc decode_all (plan of the other stream mode)");
        auto c = plan_ahead_build(r, ctx->wide);
        VIDC_HIP(hipSetDevice(ctx->device));
        VIDC_TRY(upload(ctx, c->d_wl, c->plan.wl));
        VIDC_TRY(upload(ctx, c->d_scratch_off, c->plan.scratch_off));
        VIDC_TRY(upload(ctx, c->d_slots_off, c->plan.slots_off));
        const int st = decode_impl(ctx, r, c->plan, nullptr, d_out, nullptr, 0, c.get());
        if (st == VIDC_OK) {
            std::lock_guard<std::mutex> g(r->mu);
            if (r->plan_all == plan && ctx->wide) r->plan_all = c;  // (the wide plan is the better one to keep)
        }
        return st;
    }
    if (!plan) {
        HostTrace tr("roc decode_all");
        std::shared_ptr<DecPlanCache> c;
        {
            std::lock_guard<std::mutex> g(r->mu);
            c = std::move(r->plan_ahead);
        }
        if (c && c->wide != ctx->wide) c.reset();
        if (!c) c = plan_ahead_build(r, ctx->wide);
        tr.mark("plan");
        VIDC_HIP(hipSetDevice(ctx->device));
        VIDC_TRY(upload(ctx, c->d_wl, c->plan.wl));
        VIDC_TRY(upload(ctx, c->d_scratch_off, c->plan.scratch_off));
        VIDC_TRY(upload(ctx, c->d_slots_off, c->plan.slots_off));
        tr.mark("plan upload");
        VIDC_HIP(vidc::vidc_stream_wait(ctx->stream));  // another context may use the cached plan next
        std::lock_guard<std::mutex> g(r->mu);
        if (!r->plan_all) r->plan_all = c;
        plan = r->plan_all;
    }
    return decode_impl(ctx, r, plan->plan, nullptr, d_out, nullptr, 0, plan.get());
}

int vidc_roc_decode_lists(vidc_ctx *ctx, const vidc_roc *r, uint64_t m, const uint64_t *list_nos,
                          uint64_t *d_out, uint64_t *out_offsets) {
    if (!ctx || !r || (m && !list_nos) || !out_offsets) return VIDC_ERR_INVALID;
    VIDC_TRY(r->prec.size() == r->nlist ? ensure_offsets(r) : ensure_meta(r));
    std::vector<uint32_t> lists(m);
    std::vector<uint64_t> req_off(m + 1, 0);
    for (uint64_t i = 0; i < m; i++) {
        if (list_nos[i] >= r->nlist) { set_error("list number %llu out of range", (unsigned long long)list_nos[i]); return VIDC_ERR_INVALID; }
        lists[i] = (uint32_t)list_nos[i];
        req_off[i + 1] = req_off[i] + (r->offsets[lists[i] + 1] - r->offsets[lists[i]]);
    }
    std::memcpy(out_offsets, req_off.data(), (m + 1) * 8);
    DecPlan p;
    plan_decode(r, lists, false, p, ctx->wide);
    std::vector<uint64_t> out_off(m);
    for (size_t k = 0; k < m; k++) out_off[k] = req_off[p.item[k]];  // work item k -> its slot in the request
    return decode_impl(ctx, r, p, out_off.data(), d_out, nullptr, 0);
}

int vidc_roc_decode_gather(vidc_ctx *ctx, const vidc_roc *r, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                           const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out) {
    if (!ctx || !r) return VIDC_ERR_INVALID;
    VIDC_TRY(r->prec.size() == r->nlist ? ensure_offsets(r) : ensure_meta(r));
    return vidc_decode_gather_impl(ctx, r->nlist, m, list_nos, n_items, item_slot, item_off, ids_out,
                                   [&](uint64_t l) { return r->offsets[l + 1] - r->offsets[l]; },
                                   [&](uint64_t *d, uint64_t *lo) { return vidc_roc_decode_lists(ctx, r, m, list_nos, d, lo); });
}

int vidc_roc_decode_rows(vidc_ctx *ctx, const vidc_roc *r, uint64_t m, const uint64_t *nodes, uint32_t K,
                         int32_t *d_out, uint32_t *counts) {
    if (!ctx || !r || (m && !d_out)) return VIDC_ERR_INVALID;
    if (K == 0) { set_error("K=0 unsupported"); return VIDC_ERR_UNSUPPORTED; }
    if (!nodes && m > r->nlist) { set_error("nodes == NULL selects nodes 0..m-1: m=%llu > %llu nodes", (unsigned long long)m, (unsigned long long)r->nlist); return VIDC_ERR_INVALID; }
    if (K > TINY_MAX) {  // wide rows: lists of any length through the per-list decoders, -1 padded rows
        VIDC_TRY(r->prec.size() == r->nlist ? ensure_offsets(r) : ensure_meta(r));
        std::vector<uint32_t> lists(m);
        for (uint64_t i = 0; i < m; i++) {
            const uint64_t node = nodes ? nodes[i] : i;
            if (node >= r->nlist) { set_error("node %llu out of range", (unsigned long long)node); return VIDC_ERR_INVALID; }
            lists[i] = (uint32_t)node;
            const uint64_t n = r->offsets[node + 1] - r->offsets[node];
            if (n > K) { set_error("node %u has %llu edges > K=%u", lists[i], (unsigned long long)n, K); return VIDC_ERR_INVALID; }
            if (counts) counts[i] = (uint32_t)n;
        }
        if (!m) return VIDC_OK;
        VIDC_HIP(hipSetDevice(ctx->device));
        VIDC_HIP(hipMemsetAsync(d_out, 0xff, m * (uint64_t)K * 4, ctx->stream));
        DecPlan p;
        plan_decode(r, lists, false, p, ctx->wide, false, false, true);  // only the kernels that write int32 rows
        std::vector<uint64_t> out_off(m);
        for (size_t k = 0; k < m; k++) out_off[k] = (uint64_t)p.item[k] * K;
        return decode_impl(ctx, r, p, out_off.data(), nullptr, d_out, K);
    }
    const bool lean = r->rows && K >= r->K && !force_general() && lane_wanted(lane_policy(), m, LANE_MIN_TINY);
    if (lean && !nodes) {  // rows 0..m-1 of a graph object: nothing proportional to m is built on or leaves the host
        DecPlan p;
        p.lean = true; p.tiny_lane = true; p.implicit = m;
        p.count[DC_TINY] = m;
        std::unique_lock<std::mutex> g(r->mu, std::defer_lock);
        if (m == r->nlist && m >= 65536) {  // the whole graph: rows of equal edge count share a wavefront (k_rows_order_*)
            g.lock();
            if (!r->row_order_ready) {
                // the first whole-graph decode of the object builds the order inside its own timed region (decode_impl) and publishes
                // it -- other contexts may use it -- once this call has synchronised; until then the object's lock stays with this call
                // (the per-workgroup histograms live behind the order in the object's own buffer: no scratch to hand back)
                VIDC_TRY(r->d_row_order.alloc(m + (uint64_t)VIDC_ORDER_BLOCKS * 65, ctx->dpool));
                p.order_build = true;
            } else {
                g.unlock();
            }
            p.order_dev = r->d_row_order.p;
        }
        VIDC_TRY(decode_impl(ctx, r, p, nullptr, nullptr, d_out, K));  // (ends with a synchronisation of the stream)
        if (p.order_build) { r->row_order_ready = true; g.unlock(); }
        if (counts) VIDC_TRY(fetch_sizes<uint32_t>(ctx, r->d_offsets.p, (const uint32_t *)nullptr, m, counts));
        return VIDC_OK;
    }
    if (!lean) VIDC_TRY(r->prec.size() == r->nlist ? ensure_offsets(r) : ensure_meta(r));
    std::vector<uint32_t> lists(m);
    for (uint64_t i = 0; i < m; i++) {
        const uint64_t node = nodes ? nodes[i] : i;
        if (node >= r->nlist) { set_error("node %llu out of range", (unsigned long long)node); return VIDC_ERR_INVALID; }
        lists[i] = (uint32_t)node;
        if (!lean) {
            uint64_t n = r->offsets[lists[i] + 1] - r->offsets[lists[i]];
            if (n > K) { set_error("node %u has %llu edges > K=%u", lists[i], (unsigned long long)n, K); return VIDC_ERR_INVALID; }
            if (counts) counts[i] = (uint32_t)n;
        }
    }
    DecPlan p;
    plan_decode(r, lists, true, p, ctx->wide, lean);
    if (p.lean) {
        VIDC_TRY(decode_impl(ctx, r, p, nullptr, nullptr, d_out, K));
        if (counts && m) {  // edge counts of the requested nodes, gathered on the device
            Scratch s_wl;
            Pinned h_wl;
            VIDC_TRY(s_wl.get(ctx, m * 4));
            VIDC_TRY(h_wl.get(ctx, m * 4));
            std::memcpy(h_wl.p, lists.data(), m * 4);
            VIDC_HIP(hipMemcpyAsync(s_wl.p, h_wl.p, m * 4, hipMemcpyHostToDevice, ctx->stream));
            VIDC_TRY(fetch_sizes<uint32_t>(ctx, r->d_offsets.p, s_wl.as<uint32_t>(), m, counts));
        }
        return VIDC_OK;
    }
    std::vector<uint64_t> out_off(m);
    for (size_t k = 0; k < m; k++) out_off[k] = (uint64_t)p.item[k] * K;
    return decode_impl(ctx, r, p, out_off.data(), nullptr, d_out, K);
}

}  // extern "C"
