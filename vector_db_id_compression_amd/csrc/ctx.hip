// ctx.hip -- context, error channel and raw device-memory helpers of libvidc.
#include <random>

#include <cstdlib>
#include "common.h"

namespace vidc {
static thread_local std::string g_last_error;
void set_error(const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
DevPool::~DevPool() {
    for (auto &b : blocks)
        if (b.p) (void)hipFree(b.p);
}
// VIDC_POOL_POISON=1 (tests): every block handed out is filled with 0xFF first.  Kernels that own every word they are
// supposed to write (Elias-Fano high words without a memset, packed-bits padding words, encoder-written decode records, ROC
// arenas) must produce the same bits from dirty blocks as from fresh ones; stale data from an earlier call is what a pooled
// block normally holds.
// (The switch is a member of the pool, read from the environment when the pool is created and settable through
// vidc_ctx_debug_pool_poison: round 5 called getenv on every block handed out.)
static int poison(void *p, size_t bytes) {
    VIDC_HIP(hipDeviceSynchronize());
    VIDC_HIP(hipMemset(p, 0xFF, bytes));
    VIDC_HIP(hipDeviceSynchronize());
    return VIDC_OK;
}
int DevPool::get(size_t nbytes, void **out, size_t *out_bytes) {
    VIDC_TRY(get_raw(nbytes, out, out_bytes));
    return poison_on ? poison(*out, *out_bytes) : VIDC_OK;
}
int DevPool::get_raw(size_t nbytes, void **out, size_t *out_bytes) {
    std::lock_guard<std::mutex> g(m);
    if (nbytes == 0) nbytes = 16;
    // best fit among free blocks that are large enough but not wastefully large
    int best = -1;
    for (size_t i = 0; i < blocks.size(); i++) {
        PoolBlockD &b = blocks[i];
        if (!b.in_use && b.p && b.bytes >= nbytes && b.bytes <= nbytes * 2 + (1u << 20)) {
            if (best < 0 || b.bytes < blocks[best].bytes) best = (int)i;
        }
    }
    if (best >= 0) {
        blocks[best].in_use = true;
        *out = blocks[best].p;
        *out_bytes = blocks[best].bytes;
        return VIDC_OK;
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, nbytes);
    if (e != hipSuccess) {
        // drop cached free blocks and retry once
        for (auto &b : blocks)
            if (!b.in_use && b.p) { (void)hipFree(b.p); b.p = nullptr; b.bytes = 0; }
        e = hipMalloc(&q, nbytes);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", nbytes, hipGetErrorString(e));
            return VIDC_ERR_HIP;
        }
    }
    // re-use the record of a dropped block if there is one
    for (auto &b : blocks)
        if (!b.p) { b = PoolBlockD{q, nbytes, true}; *out = q; *out_bytes = nbytes; return VIDC_OK; }
    blocks.push_back(PoolBlockD{q, nbytes, true});
    *out = q;
    *out_bytes = nbytes;
    return VIDC_OK;
}
void DevPool::put(void *p) {
    std::lock_guard<std::mutex> g(m);
    for (auto &b : blocks)
        if (b.p == p) b.in_use = false;
}
int Scratch::get(::vidc_ctx *c, size_t nbytes) {
    release();
    pool = c->dpool;
    return pool->get(nbytes, &p, &bytes);
}
void Scratch::release() {
    if (pool && p) pool->put(p);
    pool.reset();
    p = nullptr;
    bytes = 0;
}
int Pinned::get(::vidc_ctx *c, size_t nbytes) {
    release();
    ctx = c;
    if (nbytes == 0) nbytes = 16;
    int best = -1;
    for (size_t i = 0; i < c->ppool.size(); i++) {
        PoolBlock &b = c->ppool[i];
        if (!b.in_use && b.bytes >= nbytes && b.bytes <= nbytes * 4 + (1u << 20)) {
            if (best < 0 || b.bytes < c->ppool[best].bytes) best = (int)i;
        }
    }
    if (best >= 0) {
        c->ppool[best].in_use = true;
        p = c->ppool[best].p;
        bytes = c->ppool[best].bytes;
        return VIDC_OK;
    }
    void *q = nullptr;
    size_t want = nbytes + nbytes / 4;  // list counts drift between calls: leave room for reuse
    if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) {
        for (auto &b : c->ppool)
            if (!b.in_use && b.p) { (void)hipHostFree(b.p); b.p = nullptr; b.bytes = 0; }
        want = nbytes;
        hipError_t e = hipHostMalloc(&q, want, hipHostMallocDefault);
        if (e != hipSuccess) {
            set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return VIDC_ERR_HIP;
        }
    }
    c->ppool.push_back(PoolBlock{q, want, true});
    p = q;
    bytes = want;
    return VIDC_OK;
}
void Pinned::release() {
    if (ctx && p) {
        for (auto &b : ctx->ppool)
            if (b.p == p) b.in_use = false;
    }
    p = nullptr;
    bytes = 0;
}

// one thread per requested id: a plain indexed load (the positions are scattered over the decoded lists) and a coalesced store
__global__ void k_gather_ids(const uint64_t *__restrict__ ids, const uint64_t *__restrict__ pos, uint64_t n, int64_t *__restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)ids[pos[i]];
}

int gather_to_host(::vidc_ctx *c, const uint64_t *d_ids, const uint64_t *list_off, uint64_t m, uint64_t n_items,
                   const uint64_t *item_slot, const uint64_t *item_off, int64_t *host_out) {
    if (!n_items) return VIDC_OK;
    Pinned h_pos, h_out;
    VIDC_TRY(h_pos.get(c, n_items * 8));
    VIDC_TRY(h_out.get(c, n_items * 8));
    uint64_t *pos = h_pos.as<uint64_t>();
    for (uint64_t i = 0; i < n_items; i++) {
        const uint64_t s = item_slot[i];
        if (s >= m || item_off[i] >= list_off[s + 1] - list_off[s]) {
            set_error("decode_gather: item %llu names (slot %llu, offset %llu) outside the requested lists", (unsigned long long)i,
                      (unsigned long long)s, (unsigned long long)item_off[i]);
            return VIDC_ERR_INVALID;
        }
        pos[i] = list_off[s] + item_off[i];
    }
    Scratch s_pos, s_out;
    VIDC_TRY(s_pos.get(c, n_items * 8));
    VIDC_TRY(s_out.get(c, n_items * 8));
    VIDC_HIP(hipMemcpyAsync(s_pos.p, pos, n_items * 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_gather_ids, dim3((uint32_t)((n_items + 255) / 256)), dim3(256), 0, c->stream, d_ids, s_pos.as<uint64_t>(),
                       n_items, s_out.as<int64_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipMemcpyAsync(h_out.p, s_out.p, n_items * 8, hipMemcpyDeviceToHost, c->stream));
    VIDC_HIP(vidc_stream_wait(c->stream));
    std::memcpy(host_out, h_out.p, n_items * 8);
    c->d2h_bytes += n_items * 8;
    return VIDC_OK;
}
}  // namespace vidc

extern "C" {

const char *vidc_last_error(void) { return vidc::g_last_error.c_str(); }
int vidc_version(void) { return VIDC_VERSION; }
int vidc_ctx_class_streams(const vidc_ctx *ctx) { return ctx ? ctx->naux() + 1 : 0; }

int vidc_ctx_create(int device, vidc_ctx **out) {
    if (!out) return VIDC_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        vidc::set_error("no HIP device available (%s); libvidc has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return VIDC_ERR_NO_DEVICE;
    }
    if (device < 0) {
        VIDC_HIP(hipGetDevice(&device));
    } else {
        if (device >= count) {
            vidc::set_error("device %d out of range (%d devices)", device, count);
            return VIDC_ERR_INVALID;
        }
        VIDC_HIP(hipSetDevice(device));
    }
    vidc_ctx *c = new vidc_ctx();
    c->device = device;
    c->dpool = std::make_shared<vidc::DevPool>();
    c->dpool->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (const char *e = std::getenv("GPU_MAX_HW_QUEUES")) c->wide = std::atoi(e) >= 8;
    if (const char *e = std::getenv("VIDC_WIDE_STREAMS")) c->wide = e[0] == '1';  // test hook
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        [&] { for (auto &st : c->aux) if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return true; return false; }() ||
        hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
        hipEventCreate(&c->ev_chain[0]) != hipSuccess || hipEventCreate(&c->ev_chain[1]) != hipSuccess ||
        [&] { for (auto &ev : c->tev) if (hipEventCreate(&ev) != hipSuccess) return true; return false; }() ||
        [&] { for (auto &ev : c->ev_pre) if (hipEventCreate(&ev) != hipSuccess) return true; return false; }() ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        [&] { for (auto &ev : c->ev_join) if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return true; return false; }() ||
        hipMalloc((void **)&c->d_mt, VIDC_MT_TABLE * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc(&c->d_ltab, (VIDC_LANE_TAB + 1) * 16) != hipSuccess) {
        vidc::set_error("context resource creation failed");
        vidc_ctx_destroy(c);
        return VIDC_ERR_HIP;
    }
    c->stream = c->own_stream;
    // the underflow word source of the reference: std::mt19937 seeded with 1234 (codec.h:16-18)
    std::mt19937 mt(1234);
    std::vector<uint32_t> tab(VIDC_MT_TABLE);
    for (auto &w : tab) w = (uint32_t)mt();
    if (hipMemcpy(c->d_mt, tab.data(), tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        vidc::set_error("mt table upload failed");
        vidc_ctx_destroy(c);
        return VIDC_ERR_HIP;
    }
    // divisor table of the lane-per-list kernels: floor((2^64-1)/d), d*floor(2^31/d), floor(2^31/d)
    std::vector<uint32_t> lt((VIDC_LANE_TAB + 1) * 4, 0);
    for (uint32_t d = 1; d <= VIDC_LANE_TAB; d++) {
        const uint64_t m = ~0ull / d;
        const uint32_t lq = 0x80000000u / d;
        lt[4 * d + 0] = (uint32_t)m; lt[4 * d + 1] = (uint32_t)(m >> 32); lt[4 * d + 2] = lq * d; lt[4 * d + 3] = lq;
    }
    if (hipMemcpy(c->d_ltab, lt.data(), lt.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        vidc::set_error("divisor table upload failed");
        vidc_ctx_destroy(c);
        return VIDC_ERR_HIP;
    }
    {
        // divisor table of the chain kernels (roc_u2.h): one 16-byte entry per possible list length
        std::vector<uint32_t> ut(((size_t)VIDC_ROC_MAX_LIST + 1) * 4);
        for (uint32_t d = 0; d <= VIDC_ROC_MAX_LIST; d++) u2_div_entry(d, &ut[(size_t)d * 4]);
        if (hipMalloc(&c->d_u2tab, ut.size() * 4) != hipSuccess ||
            hipMemcpy(c->d_u2tab, ut.data(), ut.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
            vidc::set_error("divisor table upload failed");
            vidc_ctx_destroy(c);
            return VIDC_ERR_HIP;
        }
    }
    *out = c;
    return VIDC_OK;
}

void vidc_ctx_destroy(vidc_ctx *c) {
    if (!c) return;
    for (auto &b : c->ppool)
        if (b.p) (void)hipHostFree(b.p);
    if (c->d_mt) (void)hipFree(c->d_mt);
    if (c->d_ltab) (void)hipFree(c->d_ltab);
    if (c->d_u2tab) (void)hipFree(c->d_u2tab);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto &ev : c->ev_chain) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->tev) if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : c->ev_pre) if (ev) (void)hipEventDestroy(ev);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int i = 0; i < VIDC_NAUX; i++) {
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
        if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int vidc_ctx_set_stream(vidc_ctx *c, void *hip_stream) {
    if (!c) return VIDC_ERR_INVALID;
    c->stream = (hipStream_t)hip_stream;  // NULL = legacy default stream
    return VIDC_OK;
}

int vidc_ctx_reset_stream(vidc_ctx *c) {
    if (!c) return VIDC_ERR_INVALID;
    c->stream = c->own_stream;
    return VIDC_OK;
}

int vidc_ctx_synchronize(vidc_ctx *c) {
    if (!c) return VIDC_ERR_INVALID;
    VIDC_HIP(vidc::vidc_stream_wait(c->stream));
    return VIDC_OK;
}

int vidc_ctx_debug_pool_poison(vidc_ctx *c, int on) {
    if (!c || !c->dpool) return VIDC_ERR_INVALID;
    c->dpool->poison_on = on != 0;
    return VIDC_OK;
}
int vidc_ctx_trim(vidc_ctx *c, uint64_t *freed_bytes) {
    if (!c) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(c->device));
    VIDC_HIP(vidc::vidc_stream_wait(c->stream));
    uint64_t freed = 0;
    if (c->dpool) {
        std::lock_guard<std::mutex> g(c->dpool->m);
        for (auto &b : c->dpool->blocks)
            if (!b.in_use && b.p) {
                (void)hipFree(b.p);
                freed += b.bytes;
                b.p = nullptr;
                b.bytes = 0;
            }
    }
    for (auto &b : c->ppool)
        if (!b.in_use && b.p) {
            (void)hipHostFree(b.p);
            freed += b.bytes;
            b.p = nullptr;
            b.bytes = 0;
        }
    // the process-wide cache of emptied host vectors (VecPool, common.h) goes too: a process that encoded one large index and
    // destroyed it would otherwise hold those arrays until exit (not counted in *freed_bytes: device + pinned bytes)
    (void)vidc::vec_pool<uint32_t>().drain();
    (void)vidc::vec_pool<uint64_t>().drain();
    if (freed_bytes) *freed_bytes = freed;
    return VIDC_OK;
}

int vidc_dev_alloc(vidc_ctx *c, size_t bytes, void **p) {
    if (!c || !p) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(c->device));
    VIDC_HIP(hipMalloc(p, bytes ? bytes : 1));
    return VIDC_OK;
}
int vidc_dev_free(vidc_ctx *c, void *p) {
    if (!c) return VIDC_ERR_INVALID;
    if (p) VIDC_HIP(hipFree(p));
    return VIDC_OK;
}
int vidc_copy_h2d(vidc_ctx *c, void *d, const void *h, size_t bytes) {
    if (!c) return VIDC_ERR_INVALID;
    if (bytes) {
        VIDC_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
        VIDC_HIP(vidc::vidc_stream_wait(c->stream));
    }
    return VIDC_OK;
}
int vidc_copy_d2h(vidc_ctx *c, void *h, const void *d, size_t bytes) {
    if (!c) return VIDC_ERR_INVALID;
    if (bytes) {
        VIDC_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
        VIDC_HIP(vidc::vidc_stream_wait(c->stream));
        c->d2h_bytes += bytes;
    }
    return VIDC_OK;
}
uint64_t vidc_ctx_d2h_bytes(const vidc_ctx *c) { return c ? c->d2h_bytes : 0; }

double vidc_ctx_last_kernel_ms(const vidc_ctx *c) { return c ? c->last_kernel_ms : 0.0; }
int vidc_ctx_chain_info(const vidc_ctx *c, int which, uint64_t *ids, uint64_t *lists, uint64_t *longest, uint32_t *universe_bits) {
    if (!c || which < 0 || which > 1) return VIDC_ERR_INVALID;
    if (ids) *ids = c->chain_info[which][0];
    if (lists) *lists = c->chain_info[which][1];
    if (longest) *longest = c->chain_info[which][2];
    if (universe_bits) *universe_bits = (uint32_t)c->chain_info[which][3];
    return VIDC_OK;
}

double vidc_ctx_phase_ms(const vidc_ctx *c, int phase) {
    return (c && phase >= 0 && phase < VIDC_PHASE_COUNT) ? c->phase_ms[phase] : 0.0;
}

}  // extern "C"
