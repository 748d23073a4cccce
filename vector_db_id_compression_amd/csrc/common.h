// common.h -- host-side plumbing shared by the codec translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vidc.h"

#include "host_passes.h"

struct vidc_ctx;

namespace vidc {

void set_error(const char *fmt, ...);

#define VIDC_HIP(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            vidc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                            __LINE__);                                                        \
            return VIDC_ERR_HIP;                                                              \
        }                                                                                     \
    } while (0)

// Waiting for a stream / an event: the runtime's own wait (it spins on the completion signal for ~100 us, then sleeps until the
// interrupt).  Polling the state for longer before blocking was measured in round 4 and did not move a 1 ms call: HISTORY.md.
inline hipError_t vidc_stream_wait(hipStream_t s) { return hipStreamSynchronize(s); }
inline hipError_t vidc_event_wait(hipEvent_t ev) { return hipEventSynchronize(ev); }

#define VIDC_TRY(expr)              \
    do {                            \
        int _s = (expr);            \
        if (_s != VIDC_OK) return _s; \
    } while (0)

// Emptied std::vectors kept with their capacity (process-wide, bounded): the host arrays of a 10^6-list object are 4-8 MB each, beyond
// the allocator's mmap threshold, so every encode call page-faulted ~50 MB in and every destroyed object unmapped it again (S2: 2 ms per
// destroyed ROC object, 1-2 ms spread over the host phases of the next encode).  take(n): an empty vector of capacity >= n.
template <typename T>
struct VecPool {
    static constexpr size_t MIN_BYTES = 1u << 20, MAX_TOTAL = 256u << 20, MAX_COUNT = 96;
    std::mutex m;
    std::vector<std::vector<T>> free_;
    size_t bytes = 0;
    std::vector<T> take(size_t n) {
        if (n * sizeof(T) >= MIN_BYTES) {
            std::lock_guard<std::mutex> g(m);
            size_t best = free_.size();
            for (size_t i = 0; i < free_.size(); i++)
                if (free_[i].capacity() >= n && (best == free_.size() || free_[i].capacity() < free_[best].capacity())) best = i;
            if (best != free_.size() && free_[best].capacity() <= 2 * n + (MIN_BYTES / sizeof(T))) {
                if (best != free_.size() - 1) std::swap(free_[best], free_.back());
                std::vector<T> v = std::move(free_.back());
                free_.pop_back();
                bytes -= v.capacity() * sizeof(T);
                return v;
            }
        }
        std::vector<T> v;
        v.reserve(n);
        return v;
    }
    // release everything that is cached (vidc_ctx_trim): -> bytes returned to the allocator
    size_t drain() {
        std::lock_guard<std::mutex> g(m);
        const size_t b = bytes;
        free_.clear();
        free_.shrink_to_fit();
        bytes = 0;
        return b;
    }
    void give(std::vector<T> &&v) {
        const size_t b = v.capacity() * sizeof(T);
        if (b < MIN_BYTES) return;
        v.clear();
        std::lock_guard<std::mutex> g(m);
        if (free_.size() < MAX_COUNT && bytes + b <= MAX_TOTAL) {
            bytes += b;
            free_.push_back(std::move(v));
        }
    }
};
template <typename T>
inline VecPool<T> &vec_pool() {
    static VecPool<T> *p = new VecPool<T>();  // (never destroyed: objects may outlive static destruction order)
    return *p;
}

// Cached device blocks shared by a context and the objects created through it: steady-state encode / decode calls
// neither hipMalloc nor hipFree (hipFree synchronises the device).  Blocks go back to the pool when their owner
// dies and are released when the last holder of the pool (context or object) is gone.
struct PoolBlockD {
    void *p;
    size_t bytes;
    bool in_use;
};
struct DevPool {
    std::mutex m;
    std::vector<PoolBlockD> blocks;
    int device = 0;
    // tests: every block handed out is filled with 0xFF first (VIDC_POOL_POISON=1 when the pool is created, or
    // vidc_ctx_debug_pool_poison)
    bool poison_on = [] { const char *e = std::getenv("VIDC_POOL_POISON"); return e && e[0] == '1'; }();
    ~DevPool();
    int get(size_t nbytes, void **p, size_t *bytes);
    int get_raw(size_t nbytes, void **p, size_t *bytes);
    void put(void *p);
};

// Device buffer with explicit lifetime: hipMalloc/hipFree on the owning device, or a block of a DevPool.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    std::shared_ptr<DevPool> pool;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), pool(std::move(o.pool)) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; pool = std::move(o.pool); o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) {
            if (pool) pool->put(p);
            else (void)hipFree(p);
        }
        pool.reset();
        p = nullptr;
        n = 0;
    }
    int alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return VIDC_OK;
        VIDC_HIP(hipMalloc((void **)&p, count * sizeof(T)));
        return VIDC_OK;
    }
    // from the context's block cache
    int alloc(size_t count, const std::shared_ptr<DevPool> &from) {
        release();
        n = count;
        size_t got = 0;
        void *q = nullptr;
        VIDC_TRY(from->get(count ? count * sizeof(T) : 16, &q, &got));
        p = (T *)q;
        pool = from;
        return VIDC_OK;
    }
};

// Scratch buffer borrowed from the context pool (returned on destruction).
struct Scratch {
    std::shared_ptr<DevPool> pool;
    void *p = nullptr;
    size_t bytes = 0;
    Scratch() = default;
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    ~Scratch() { release(); }
    int get(::vidc_ctx *c, size_t nbytes);
    void release();
    template <typename T>
    T *as() const { return (T *)p; }
};

// Page-locked host staging borrowed from the context pool: per-list arrays (work lists, offsets, metadata) cross
// PCIe at DMA speed instead of the pageable-copy path (measured ~1.5 GB/s for 8 MB vectors).
struct Pinned {
    ::vidc_ctx *ctx = nullptr;
    void *p = nullptr;
    size_t bytes = 0;
    Pinned() = default;
    Pinned(const Pinned &) = delete;
    Pinned &operator=(const Pinned &) = delete;
    ~Pinned() { release(); }
    int get(::vidc_ctx *c, size_t nbytes);
    void release();
    template <typename T>
    T *as() const { return (T *)p; }
};

// The device-side scatter of the deferred search (custom_invlists_impl.cpp:508-525: labels[r] = ids[offset]): d_ids holds m decoded
// lists back to back (list_off[m + 1], host), item i names (slot in the request, offset in that list); the n_items ids are
// picked on the device and ONLY they cross PCIe (8 * n_items bytes into host_out).  Bounds-checked on the host.
int gather_to_host(::vidc_ctx *c, const uint64_t *d_ids, const uint64_t *list_off, uint64_t m, uint64_t n_items,
                   const uint64_t *item_slot, const uint64_t *item_off, int64_t *host_out);

}  // namespace vidc

// mt19937(1234) words available to the kernels for ANS stack underflow (codec.h:16-18,32-40).
// The reference draws 0 or 1 word per list in every valid case.
#define VIDC_MT_TABLE 1024
// largest divisor of the lane-per-list kernels' reciprocal table (== VIDC_LANE_MAX64 in roc_lane.h)
#define VIDC_LANE_TAB 4096
// one entry of the chain kernels' divisor table (roc_u2.h: U2Div)
inline void u2_div_entry(uint32_t d, uint32_t out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
    if (d == 0) return;
    const uint64_t m = ~0ull / (uint64_t)d;
    out[0] = (uint32_t)m;
    out[1] = (uint32_t)(m >> 32);
    out[2] = d >= 2u ? (uint32_t)(0x100000000ull / d) : 0xffffffffu;
    out[3] = 0x80000000u / d;
}

// Scratch blocks are cached per context so steady-state encode/decode calls do not hipMalloc.
struct PoolBlock {
    void *p;
    size_t bytes;
    bool in_use;
};

#define VIDC_NAUX 7
struct vidc_ctx {
    std::shared_ptr<vidc::DevPool> dpool;  // device blocks (scratch + the objects created through this context)
    std::vector<PoolBlock> ppool;          // pinned host blocks
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_chain[2] = {nullptr, nullptr};  // around the launch of the longest-chain kernel class of a ROC call
    hipEvent_t ev_pre[3] = {};  // ROC encode, classification prepass read back late: before the kernel, behind it, behind its copies
    hipEvent_t tev[24] = {};   // event pairs of PhaseTimer (kernel phases timed without a host synchronisation each)
    // kernel classes of one call run concurrently: long chains on `stream`, shorter classes on these
    // (aux[0..2] always; aux[3..] only when the process has the hardware queues for them: HIP multiplexes its streams onto
    // GPU_MAX_HW_QUEUES (default 4) hardware queues, and two streams that share a queue run one after the other in
    // submission order -- `wide` is set at context creation when that variable is at least 8)
    hipStream_t aux[VIDC_NAUX] = {};
    hipEvent_t ev_fork = nullptr, ev_join[VIDC_NAUX] = {};
    bool wide = false;
    int naux() const { return wide ? VIDC_NAUX : 3; }
    uint32_t *d_mt = nullptr;  // VIDC_MT_TABLE words
    void *d_u2tab = nullptr;   // per-divisor constants of the hand-scheduled chain kernels (roc_u2.h), VIDC_ROC_MAX_LIST + 1 entries
    void *d_ltab = nullptr;    // divisor table of the lane-per-list kernels (roc_lane.h), VIDC_LANE_TAB + 1 entries
    int num_cu = 256;
    double last_kernel_ms = 0.0;
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // see VIDC_PHASE_* in vidc.h
    // what the chain launch of the last ROC encode [0] / decode [1] held: ids, lists, longest list, universe bits (0 = none)
    uint64_t chain_info[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    // id payload copied device -> host through this context (vidc_copy_d2h, *_get, *_decode_gather): see vidc_ctx_d2h_bytes
    uint64_t d2h_bytes = 0;
    // largest list universe (last id) the Elias-Fano encoder has met on this context: sizes the streams of the next object before
    // its ids have been looked at (ef_encode_fast)
    uint64_t ef_universe_hint = 0;
};

// vidc_*_decode_gather (include/vidc.h): decode the m touched lists into staging from the context's block cache, pick the
// requested ids on the device, copy 8 * n_items bytes.  size_of(list) = ids of a list, decode(d_out, list_off) = the codec's
// own decode_lists.
template <typename SizeOf, typename Decode>
inline int vidc_decode_gather_impl(vidc_ctx *ctx, uint64_t nlist, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                                   const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out, SizeOf &&size_of,
                                   Decode &&decode) {
    if (!n_items) return VIDC_OK;
    if (!m || !list_nos || !item_slot || !item_off || !ids_out) return VIDC_ERR_INVALID;
    uint64_t total = 0;
    std::vector<uint64_t> sizes(m);
    for (uint64_t i = 0; i < m; i++) {
        if (list_nos[i] >= nlist) { vidc::set_error("decode_gather: list number %llu out of range", (unsigned long long)list_nos[i]); return VIDC_ERR_INVALID; }
        sizes[i] = size_of(list_nos[i]);
        total += sizes[i];
    }
    // (the request is checked BEFORE anything is decoded: an invalid item must not cost the decode of every touched list)
    for (uint64_t i = 0; i < n_items; i++)
        if (item_slot[i] >= m || item_off[i] >= sizes[item_slot[i]]) {
            vidc::set_error("decode_gather: item %llu = (slot %llu, offset %llu) is outside its list", (unsigned long long)i,
                            (unsigned long long)item_slot[i], (unsigned long long)item_off[i]);
            return VIDC_ERR_INVALID;
        }
    vidc::Scratch staging;
    VIDC_TRY(staging.get(ctx, (total ? total : 1) * 8));
    std::vector<uint64_t> list_off(m + 1, 0);
    VIDC_TRY(decode(staging.as<uint64_t>(), list_off.data()));
    return vidc::gather_to_host(ctx, staging.as<uint64_t>(), list_off.data(), m, n_items, item_slot, item_off, ids_out);
}

// VIDC_TRACE=1 prints host-side phase times of encode / decode calls (dev aid)
struct HostTrace {
    bool on;
    std::chrono::steady_clock::time_point t0;
    const char *what;
    explicit HostTrace(const char *w) : what(w) {
        static const bool env_on_ = [] { const char *e = std::getenv("VIDC_TRACE"); return e && e[0] == '1'; }();
        on = env_on_;
        if (on) t0 = std::chrono::steady_clock::now();
    }
    void mark(const char *phase) {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[vidc] %s: %-28s %8.3f ms\n", what, phase, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

// Kernel time of a call made of several phases: every phase is bracketed by an event pair on the context's stream
// and the elapsed times are summed when the call synchronises anyway (an event synchronisation per phase cost
// ~25 us of host time each: a third of an Elias-Fano encode of 64 M ids).
struct VidcPhaseTimer {
    vidc_ctx *c;
    int used = 0;
    double total = 0;
    explicit VidcPhaseTimer(vidc_ctx *ctx) : c(ctx) {}
    void begin() {
        if (used == 12) collect();
        (void)hipEventRecord(c->tev[2 * used], c->stream);
    }
    void end() {
        (void)hipEventRecord(c->tev[2 * used + 1], c->stream);
        used++;
    }
    // call after (or instead of) a stream synchronisation
    double collect() {
        for (int i = 0; i < used; i++) {
            (void)vidc::vidc_event_wait(c->tev[2 * i + 1]);
            float ms = 0;
            if (hipEventElapsedTime(&ms, c->tev[2 * i], c->tev[2 * i + 1]) == hipSuccess) total += ms;
        }
        used = 0;
        return total;
    }
};
