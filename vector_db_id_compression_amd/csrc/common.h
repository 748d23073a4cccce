// common.h -- host-side plumbing shared by the codec translation units (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vidc.h"

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

#define VIDC_TRY(expr)              \
    do {                            \
        int _s = (expr);            \
        if (_s != VIDC_OK) return _s; \
    } while (0)

// Device buffer with explicit lifetime (hipMalloc/hipFree on the owning device).
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
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
};

// Scratch buffer borrowed from the context pool (returned on destruction).
struct Scratch {
    ::vidc_ctx *ctx = nullptr;
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

}  // namespace vidc

// mt19937(1234) words available to the kernels for ANS stack underflow (codec.h:16-18,32-40).
// The reference draws 0 or 1 word per list in every valid case.
#define VIDC_MT_TABLE 1024
// largest divisor of the lane-per-list kernels' reciprocal table (== VIDC_LANE_MAX in roc_lane.h)
#define VIDC_LANE_TAB 1024

// Scratch blocks are cached per context so steady-state encode/decode calls do not hipMalloc.
struct PoolBlock {
    void *p;
    size_t bytes;
    bool in_use;
};

struct vidc_ctx {
    std::vector<PoolBlock> pool;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // kernel classes of one call run concurrently: long chains on `stream`, shorter classes on these
    hipStream_t aux[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    uint32_t *d_mt = nullptr;  // VIDC_MT_TABLE words
    void *d_ltab = nullptr;    // divisor table of the lane-per-list kernels (roc_lane.h), VIDC_LANE_MAX + 1 entries
    int num_cu = 256;
    double last_kernel_ms = 0.0;
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // see VIDC_PHASE_* in vidc.h
};
