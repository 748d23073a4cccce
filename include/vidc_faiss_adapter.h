/*
 * vidc_faiss_adapter.h -- C++ adapter that puts libvidc behind faiss::InvertedLists / faiss::nsg::Graph.
 *
 * What a maintainer of the reference adds next to custom_invlists_impl.h / altid_impl.h so that bench_invlists.py,
 * search_ivf_qinco.py and graph_dynamic_bench_invlists.py select the GPU codecs through the same SWIG modules
 * (`%include "vidc_faiss_adapter.h"` in custom_invlists.swig:61 / altid.swig; the harness dictionaries then name these
 * classes).  One class per reference class, same constructor argument, same public size attributes:
 *
 *   reference (custom_invlists_impl.h / altid_impl.h)            here
 *   CompressedIDInvertedListsFenwickTree  (.cpp:133-223)          vidc_faiss::ROCInvertedLists
 *   CompressedIDInvertedListsEliasFano    (.cpp:229-339)          vidc_faiss::EliasFanoInvertedLists
 *   CompressedIDInvertedListsPackedBits   (.cpp:64-118)           vidc_faiss::PackedBitsInvertedLists
 *   CompressedIDInvertedListsWaveletTree  (.cpp:346-397)          vidc_faiss::WaveletTreeInvertedLists
 *   CompactBitNSGGraph / EliasFanoNSGGraph / ROCNSGGraph          vidc_faiss::CompactBitGraph / EliasFanoGraph / ROCGraph
 *   (altid_impl.cpp:20-165)
 *   search_IVF_defer_id_decoding          (.cpp:407-526)          vidc_faiss::search_IVF_defer_id_decoding: the OpenMP loop
 *                                                                 over touched lists (:508-525) is ONE vidc_*_decode_gather call
 *                                                                 (device decode + device-side labels[r] = ids[offset]),
 *                                                                 the OpenMP loop of decode_1by1 (:464-474) ONE vidc_*_get call
 *
 * With VIDC_FAISS_REFERENCE_NAMES defined before the include, the reference's own class names exist at global scope
 * (CompressedIDInvertedListsFenwickTree, ...EliasFano, ...PackedBits, ...WaveletTree, CompactBitNSGGraph, EliasFanoNSGGraph,
 * ROCNSGGraph, search_IVF_defer_id_decoding) as DERIVED structs with the reference's constructor signatures -- not as alias
 * declarations: SWIG treats `using A = B;` as a typedef and would generate the Python proxy under B's name only, while a derived
 * struct is wrapped under its own name.  The SWIG modules then carry the names bench_invlists.py:19-25 and
 * graph_dynamic_bench_invlists.py:21-26 look up, and the harnesses run unchanged (INTEGRATION.md section 2 has the .swig diff).
 * Everything SWIG need not see (thread contexts, staging, the row cache) is fenced with `#ifndef SWIG`.
 *
 * Needs the Faiss headers, include/vidc.h and -lvidc.  Faiss is not installed in this repository's build image: the
 * header is compiled and exercised there against the interface shim of tests/faiss_shim (tests/test_boundary.py,
 * tests/adapter_smoke.cpp); against a real Faiss nothing else changes.
 *
 * Threading: Faiss calls get_ids / get_single_id / get_neighbors concurrently from OpenMP threads
 * (custom_invlists_impl.cpp:467,508).  A vidc_ctx serves one host thread, so every thread gets its own context
 * (thread_local, created on first use) with a device staging buffer that grows to the largest request and is reused:
 * no allocation, no lock on the per-call path.  Compressed objects are immutable and shared.
 */
#pragma once
#if defined(__has_include)
#if !__has_include(<faiss/invlists/InvertedLists.h>)
#error "vidc_faiss_adapter.h needs the Faiss headers (or tests/faiss_shim) on the include path"
#endif
#endif
#include <faiss/IndexIVF.h>
#include <faiss/impl/FaissAssert.h>
#include <faiss/impl/NSG.h>
#include <faiss/invlists/DirectMap.h>
#include <faiss/invlists/InvertedLists.h>

#include <atomic>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "vidc.h"

namespace vidc_faiss {

#define VIDC_FAISS_CHECK(expr) FAISS_THROW_IF_NOT_MSG((expr) == VIDC_OK, vidc_last_error())

#ifndef SWIG /* implementation detail: not part of the wrapped surface */
/* per-thread context + pooled device staging */
struct ThreadCtx {
    vidc_ctx* h = nullptr;
    void* d_buf = nullptr;
    size_t cap = 0;
    size_t device_calls = 0; /* library calls that launch kernels, made through this thread's context (tests, tuning) */
    vidc_ctx* ctx() {
        if (!h) VIDC_FAISS_CHECK(vidc_ctx_create(-1, &h));
        return h;
    }
    void* staging(size_t bytes) {
        if (bytes > cap) {
            if (d_buf) vidc_dev_free(ctx(), d_buf);
            d_buf = nullptr;
            cap = 0;
            size_t want = bytes < (1u << 16) ? (1u << 16) : bytes + bytes / 2;
            VIDC_FAISS_CHECK(vidc_dev_alloc(ctx(), want, &d_buf));
            cap = want;
        }
        return d_buf;
    }
    ~ThreadCtx() {
        if (h) {
            if (d_buf) vidc_dev_free(h, d_buf);
            vidc_ctx_destroy(h);
        }
    }
};
inline ThreadCtx& thread_ctx() {
    thread_local ThreadCtx c;
    return c;
}

/* RAII device copy of a host array (constructor-time uploads) */
struct DeviceArray {
    vidc_ctx* ctx;
    void* p = nullptr;
    DeviceArray(vidc_ctx* c, const void* host, size_t bytes) : ctx(c) {
        if (!bytes) return;
        VIDC_FAISS_CHECK(vidc_dev_alloc(ctx, bytes, &p));
        int st = vidc_copy_h2d(ctx, p, host, bytes);
        if (st != VIDC_OK) {
            vidc_dev_free(ctx, p);
            p = nullptr;
            VIDC_FAISS_CHECK(st);
        }
    }
    ~DeviceArray() {
        if (p) vidc_dev_free(ctx, p);
    }
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;
};
#endif /* SWIG */

/* ---------------------------------------------------------------------------------------------------------------
 * InvertedListsArrayCodes (custom_invlists_impl.h:22-33): CSR of the list sizes + the vector codes, optionally
 * re-ordered; plus the batched decode every subclass provides for the deferred search.                          */
#ifndef VIDC_FAISS_PREFETCH_MAX_IDS
#define VIDC_FAISS_PREFETCH_MAX_IDS (size_t(1) << 24) /* ids of announced lists kept decoded on the host (128 MB) */
#endif
struct CompressedInvertedLists : faiss::ReadOnlyInvertedLists {
#ifndef SWIG
    struct Prefetch { /* one announcement of IndexIVF::search_preassigned */
        std::thread::id owner;                  /* the thread that announced it */
        std::vector<uint64_t> lists;            /* distinct non-empty lists */
        std::unordered_map<uint64_t, size_t> slot;
        std::atomic<long> remaining{0};         /* announced visits not served yet: the cache is dropped when it reaches 0 */
        size_t total_ids = 0;
        std::once_flag once; /* the first get_ids of an announced list decodes them all */
        std::vector<faiss::idx_t> ids;
        std::vector<uint64_t> off;
    };
    /* Announcements that are being served.  IndexIVF::search cuts a batch into one slice per OpenMP thread and every slice announces
     * its own lists from its own thread, concurrently (round 5 kept ONE slot per container: the slices overwrote each other and a
     * stale announcement pinned up to 128 MB of decoded ids until the next search).  Now: one announcement per announcing thread
     * (a new one from the same thread replaces it), found first by the calling thread's own, then by any that holds the list (a
     * batch too small to be sliced announces from the main thread and scans from the workers); every visit served counts the
     * announcement down and the last one frees its cache; the cached ids of all live announcements stay below
     * VIDC_FAISS_PREFETCH_MAX_IDS, beyond that an announcement is not recorded and its lists take the per-list path. */
    mutable std::mutex pf_mu_;
    mutable std::vector<std::shared_ptr<Prefetch>> pf_active_;
#endif
    std::vector<uint64_t> offsets;                /* [nlist + 1] */
    std::vector<std::vector<uint8_t>> codes_all;  /* per list, in the container's order */
    size_t compressed_ids_size_in_bytes = 0, overhead_in_bytes = 0, codes_size_in_bytes = 0;

    explicit CompressedInvertedLists(const faiss::InvertedLists& il)
            : faiss::ReadOnlyInvertedLists(il.nlist, il.code_size), offsets(il.nlist + 1, 0), codes_all(il.nlist) {
        for (size_t l = 0; l < nlist; l++) offsets[l + 1] = offsets[l] + il.list_size(l);
    }
    /* ids of every list back to back (borrowed like custom_invlists_impl.cpp:156-160) */
    static std::vector<uint64_t> gather_ids(const faiss::InvertedLists& il, const std::vector<uint64_t>& offsets) {
        std::vector<uint64_t> ids(offsets.back());
        for (size_t l = 0; l < il.nlist; l++) {
            size_t n = il.list_size(l);
            if (!n) continue;
            faiss::InvertedLists::ScopedIds s(&il, l);
            std::memcpy(ids.data() + offsets[l], s.get(), n * 8);
        }
        return ids;
    }
    /* codes_all[l][i] = source code at position perm[offsets[l] + i] of list l (identity when perm == nullptr) */
    void store_codes(const faiss::InvertedLists& il, const uint32_t* perm) {
        for (size_t l = 0; l < nlist; l++) {
            size_t n = il.list_size(l);
            codes_all[l].resize(n * code_size);
            if (!n || !code_size) continue;
            faiss::InvertedLists::ScopedCodes c(&il, l);
            for (size_t i = 0; i < n; i++) {
                size_t src = perm ? perm[offsets[l] + i] : i;
                std::memcpy(&codes_all[l][i * code_size], c.get() + src * code_size, code_size);
            }
        }
    }
    /* the reference adds the size of ALL code arrays once per non-empty list (custom_invlists_impl.cpp:203-205) */
    void reference_codes_accounting() {
        size_t total = 0, nonempty = 0;
        for (auto& c : codes_all) total += c.size();
        for (size_t l = 0; l < nlist; l++) nonempty += offsets[l + 1] > offsets[l];
        codes_size_in_bytes = nonempty * total;
    }

    size_t list_size(size_t l) const override { return offsets[l + 1] - offsets[l]; }
    const uint8_t* get_codes(size_t l) const override { return codes_all[l].data(); }
    void release_ids(size_t, const faiss::idx_t* ids) const override { delete[] ids; } /* :116-118,221-223,320-322,395-397 */

    /* decode m lists into device memory d_out (lists back to back), offsets of the lists in out_off[m + 1] */
    virtual int decode_lists_device(vidc_ctx* ctx, uint64_t m, const uint64_t* list_nos, uint64_t* d_out,
                                    uint64_t* out_off) const = 0;

    /* the decode section of the deferred search in ONE library call (vidc_*_decode_gather): the m touched lists are decoded into
     * device staging owned by the context, the n_items requested ids are picked on the device, 8 * n_items bytes come back */
    virtual int decode_gather_device(vidc_ctx* ctx, uint64_t m, const uint64_t* list_nos, uint64_t n_items, const uint64_t* item_slot,
                                     const uint64_t* item_off, int64_t* ids_out) const = 0;

    /* Faiss announces the lists a search will visit (InvertedLists::prefetch_lists, called by IndexIVF::search_preassigned before
     * the scan; the reference's containers inherit the no-op).  Here the announcement is only RECORDED: the first get_ids of an
     * announced list decodes ALL of them with one library call into a host cache the scan threads then read -- a search with
     * nprobe x nq probed lists costs one launch + one copy instead of one per probed list and thread.  A deferred search
     * (store_pairs) never calls get_ids, so its announcement costs nothing.  At most VIDC_FAISS_PREFETCH_MAX_IDS ids are cached
     * (the decoded lists are 8 bytes per id: the point of the container is not to hold them); larger announcements and lists
     * outside the announcement take the per-list path.  clear_prefetch() drops the cache. */
    void prefetch_lists(const faiss::idx_t* list_nos, int n) const override {
        std::shared_ptr<Prefetch> st;
        if (n > 1) {
            st = std::make_shared<Prefetch>();
            st->owner = std::this_thread::get_id();
            long visits = 0;
            for (int i = 0; i < n && st->total_ids <= VIDC_FAISS_PREFETCH_MAX_IDS; i++) {
                if (list_nos[i] < 0 || (size_t)list_nos[i] >= nlist || !list_size(list_nos[i])) continue;
                visits++;
                if (st->slot.emplace((uint64_t)list_nos[i], st->lists.size()).second) {
                    st->lists.push_back((uint64_t)list_nos[i]);
                    st->total_ids += list_size(list_nos[i]);
                }
            }
            st->remaining.store(visits);
            if (st->total_ids > VIDC_FAISS_PREFETCH_MAX_IDS || st->lists.size() < 2) st.reset();
        }
        std::lock_guard<std::mutex> g(pf_mu_);
        size_t live = 0;
        for (size_t i = 0; i < pf_active_.size();) {  /* this thread's previous announcement is over */
            if (pf_active_[i]->owner == std::this_thread::get_id()) pf_active_.erase(pf_active_.begin() + i);
            else live += pf_active_[i++]->total_ids;
        }
        if (st && live + st->total_ids <= VIDC_FAISS_PREFETCH_MAX_IDS) pf_active_.push_back(st);
    }
    void clear_prefetch() const {
        std::lock_guard<std::mutex> g(pf_mu_);
        pf_active_.clear();
    }
    size_t prefetch_live_announcements() const {
        std::lock_guard<std::mutex> g(pf_mu_);
        return pf_active_.size();
    }

    /* get_ids: new idx_t[list_size], nullptr for an empty list (:212-214,294-296) */
    const faiss::idx_t* get_ids(size_t l) const override {
        size_t n = list_size(l);
        if (n == 0) return nullptr;
        std::unique_ptr<faiss::idx_t[]> out(new faiss::idx_t[n]);
        std::shared_ptr<Prefetch> st;
        {
            std::lock_guard<std::mutex> g(pf_mu_);
            const std::thread::id me = std::this_thread::get_id();
            for (auto& a : pf_active_)  /* (a handful: one per announcing thread) */
                if (a->slot.count((uint64_t)l) && (!st || a->owner == me)) {
                    st = a;
                    if (a->owner == me) break;
                }
        }
        if (st) {
            std::call_once(st->once, [&] { decode_lists_host(st->lists.size(), st->lists.data(), st->ids, st->off); });
            std::memcpy(out.get(), st->ids.data() + st->off[st->slot.find((uint64_t)l)->second], n * sizeof(faiss::idx_t));
            if (st->remaining.fetch_sub(1) <= 1) {  /* the last announced visit: the search this announcement belonged to is over */
                std::lock_guard<std::mutex> g(pf_mu_);
                for (size_t i = 0; i < pf_active_.size(); i++)
                    if (pf_active_[i] == st) { pf_active_.erase(pf_active_.begin() + i); break; }
            }
            return out.release();
        }
        ThreadCtx& t = thread_ctx();
        uint64_t off[2], ln = l;
        uint64_t* d = (uint64_t*)t.staging(n * 8);
        t.device_calls++;
        VIDC_FAISS_CHECK(decode_lists_device(t.ctx(), 1, &ln, d, off));
        VIDC_FAISS_CHECK(vidc_copy_d2h(t.ctx(), out.get(), d, n * 8));
        return out.release();
    }
    /* the lists a batch of searches touched, decoded by ONE call (replaces the loop :508-525); host result */
    void decode_lists_host(uint64_t m, const uint64_t* list_nos, std::vector<faiss::idx_t>& ids,
                           std::vector<uint64_t>& out_off) const {
        out_off.assign(m + 1, 0);
        size_t total = 0;
        for (uint64_t i = 0; i < m; i++) total += list_size(list_nos[i]);
        ids.resize(total);
        if (!total) return;
        ThreadCtx& t = thread_ctx();
        uint64_t* d = (uint64_t*)t.staging(total * 8);
        t.device_calls++;
        VIDC_FAISS_CHECK(decode_lists_device(t.ctx(), m, list_nos, d, out_off.data()));
        VIDC_FAISS_CHECK(vidc_copy_d2h(t.ctx(), ids.data(), d, total * 8));
    }
    /* ids_out[i] = list_nos[item_slot[i]][item_off[i]] for n_items results over m touched lists (labels[r] = ids[offset],
     * custom_invlists_impl.cpp:517-523): device decode + device-side pick, only the results cross PCIe */
    void gather_ids_host(uint64_t m, const uint64_t* list_nos, uint64_t n_items, const uint64_t* item_slot, const uint64_t* item_off,
                         faiss::idx_t* ids_out) const {
        if (!n_items) return;
        ThreadCtx& t = thread_ctx();
        t.device_calls++;
        VIDC_FAISS_CHECK(decode_gather_device(t.ctx(), m, list_nos, n_items, item_slot, item_off, (int64_t*)ids_out));
    }
    /* m random accesses ids_out[i] = get_single_id(list_nos[i], offs[i]) in ONE library call (the OpenMP loop of
     * decode_1by1, custom_invlists_impl.cpp:464-474).  Containers with a select (Elias-Fano, packed bits, wavelet tree)
     * override it with their vidc_*_get; the default -- ROC, whose get_single_id IS get_ids()[offset] in the reference
     * (faiss::InvertedLists::get_single_id) -- decodes the touched lists once and indexes them ON THE DEVICE. */
    virtual void get_single_ids(uint64_t m, const uint64_t* list_nos, const uint64_t* offs, faiss::idx_t* ids_out) const {
        /* (scratch of the calling thread, reused from call to call) */
        thread_local std::unordered_map<uint64_t, size_t> slot;
        thread_local std::vector<uint64_t> lists, item_slot;
        slot.clear();
        lists.clear();
        item_slot.resize(m);
        for (uint64_t i = 0; i < m; i++) {
            FAISS_THROW_IF_NOT_MSG(list_nos[i] < nlist && offs[i] < list_size(list_nos[i]), "get_single_ids: (list, offset) out of range");
            auto it = slot.emplace(list_nos[i], lists.size());
            if (it.second) lists.push_back(list_nos[i]);
            item_slot[i] = it.first->second;
        }
        gather_ids_host(lists.size(), lists.data(), m, item_slot.data(), offs, ids_out);
    }
};

/* CompressedIDInvertedListsFenwickTree (custom_invlists_impl.cpp:133-223): ROC / bits-back ANS */
struct ROCInvertedLists : CompressedInvertedLists {
    vidc_roc* roc = nullptr;
    std::vector<uint64_t> id_symbol_precision; /* custom_invlists_impl.h:61 */

    explicit ROCInvertedLists(const faiss::InvertedLists& il) : CompressedInvertedLists(il) {
        vidc_ctx* ctx = thread_ctx().ctx();
        {
            std::vector<uint64_t> ids = gather_ids(il, offsets);
            DeviceArray d_ids(ctx, ids.data(), ids.size() * 8);
            VIDC_FAISS_CHECK(vidc_roc_encode(ctx, nlist, offsets.data(), (const uint64_t*)d_ids.p, VIDC_PREC_REFERENCE,
                                             VIDC_ROC_WANT_PERM, &roc));
        }
        compressed_ids_size_in_bytes = vidc_roc_compressed_bytes(roc); /* :196-206 */
        std::vector<uint32_t> perm(offsets.back() ? offsets.back() : 1), prec(nlist ? nlist : 1);
        VIDC_FAISS_CHECK(vidc_roc_perm(ctx, roc, perm.data()));
        VIDC_FAISS_CHECK(vidc_roc_list_info(roc, nullptr, prec.data(), nullptr, nullptr, nullptr));
        id_symbol_precision.assign(prec.begin(), prec.begin() + nlist);
        store_codes(il, perm.data()); /* codes follow the sampling order (:188-193) */
        reference_codes_accounting();
    }
    ~ROCInvertedLists() override { vidc_roc_destroy(roc); }
    int decode_lists_device(vidc_ctx* ctx, uint64_t m, const uint64_t* ln, uint64_t* d_out, uint64_t* off) const override {
        return vidc_roc_decode_lists(ctx, roc, m, ln, d_out, off);
    }
    int decode_gather_device(vidc_ctx* ctx, uint64_t m, const uint64_t* ln, uint64_t n_items, const uint64_t* slot, const uint64_t* off,
                             int64_t* out) const override {
        return vidc_roc_decode_gather(ctx, roc, m, ln, n_items, slot, off, out);
    }
    /* get_single_id is not overridden in the reference: InvertedLists::get_single_id = get_ids()[offset] */
};

/* CompressedIDInvertedListsEliasFano (custom_invlists_impl.cpp:229-339) */
struct EliasFanoInvertedLists : CompressedInvertedLists {
    vidc_ef* ef = nullptr;
    explicit EliasFanoInvertedLists(const faiss::InvertedLists& il) : CompressedInvertedLists(il) {
        vidc_ctx* ctx = thread_ctx().ctx();
        {
            std::vector<uint64_t> ids = gather_ids(il, offsets);
            DeviceArray d_ids(ctx, ids.data(), ids.size() * 8);
            VIDC_FAISS_CHECK(vidc_ef_encode(ctx, nlist, offsets.data(), (const uint64_t*)d_ids.p, VIDC_EF_WANT_PERM, &ef));
        }
        compressed_ids_size_in_bytes = vidc_ef_compressed_bytes(ef); /* :272-282 */
        std::vector<uint32_t> perm(offsets.back() ? offsets.back() : 1);
        VIDC_FAISS_CHECK(vidc_ef_perm(ctx, ef, perm.data()));
        store_codes(il, perm.data()); /* canonicalize_order_inplace (:324-339) */
        reference_codes_accounting();
    }
    ~EliasFanoInvertedLists() override { vidc_ef_destroy(ef); }
    int decode_lists_device(vidc_ctx* ctx, uint64_t m, const uint64_t* ln, uint64_t* d_out, uint64_t* off) const override {
        return vidc_ef_decode_lists(ctx, ef, m, ln, d_out, off);
    }
    int decode_gather_device(vidc_ctx* ctx, uint64_t m, const uint64_t* ln, uint64_t n_items, const uint64_t* slot, const uint64_t* off,
                             int64_t* out) const override {
        return vidc_ef_decode_gather(ctx, ef, m, ln, n_items, slot, off, out);
    }
    faiss::idx_t get_single_id(size_t l, size_t offset) const override { /* ef->select(offset), :314-318 */
        uint64_t ln = l, of = offset;
        int64_t id = -1;
        get_single_ids(1, &ln, &of, &id);
        return id;
    }
    void get_single_ids(uint64_t m, const uint64_t* ln, const uint64_t* of, faiss::idx_t* out) const override {
        thread_ctx().device_calls++;
        VIDC_FAISS_CHECK(vidc_ef_get(thread_ctx().ctx(), ef, m, ln, of, (int64_t*)out));
    }
};

/* CompressedIDInvertedListsPackedBits (custom_invlists_impl.cpp:64-118) */
struct PackedBitsInvertedLists : CompressedInvertedLists {
    vidc_packed* pk = nullptr;
    int bits = 0;
    explicit PackedBitsInvertedLists(const faiss::InvertedLists& il) : CompressedInvertedLists(il) {
        vidc_ctx* ctx = thread_ctx().ctx();
        bits = vidc_packed_bits_for(offsets.back()); /* :68-70 */
        std::vector<uint64_t> ids = gather_ids(il, offsets);
        for (uint64_t v : ids) FAISS_THROW_IF_NOT(v < offsets.back()); /* ids_in[i] >= 0 && ids_in[i] < ntotal, :87 */
        DeviceArray d_ids(ctx, ids.data(), ids.size() * 8);
        VIDC_FAISS_CHECK(vidc_packed_encode(ctx, nlist, offsets.data(), (const uint64_t*)d_ids.p, bits, &pk));
        compressed_ids_size_in_bytes = vidc_packed_compressed_bytes(pk); /* :80,85 */
        store_codes(il, nullptr);
    }
    ~PackedBitsInvertedLists() override { vidc_packed_destroy(pk); }
    int decode_lists_device(vidc_ctx* ctx, uint64_t m, const uint64_t* ln, uint64_t* d_out, uint64_t* off) const override {
        return vidc_packed_decode_lists(ctx, pk, m, ln, d_out, off);
    }
    int decode_gather_device(vidc_ctx* ctx, uint64_t m, const uint64_t* ln, uint64_t n_items, const uint64_t* slot, const uint64_t* off,
                             int64_t* out) const override {
        return vidc_packed_decode_gather(ctx, pk, m, ln, n_items, slot, off, out);
    }
    faiss::idx_t get_single_id(size_t l, size_t offset) const override { /* :108-113 */
        uint64_t ln = l, of = offset;
        int64_t id = -1;
        get_single_ids(1, &ln, &of, &id);
        return id;
    }
    void get_single_ids(uint64_t m, const uint64_t* ln, const uint64_t* of, faiss::idx_t* out) const override {
        thread_ctx().device_calls++;
        VIDC_FAISS_CHECK(vidc_packed_get(thread_ctx().ctx(), pk, m, ln, of, (int64_t*)out));
    }
};

/* CompressedIDInvertedListsWaveletTree (custom_invlists_impl.cpp:346-397) */
struct WaveletTreeInvertedLists : CompressedInvertedLists {
    vidc_wt* wt = nullptr;
    int wt_type = 0;
    explicit WaveletTreeInvertedLists(const faiss::InvertedLists& il, int wt_type_ = 0)
            : CompressedInvertedLists(il), wt_type(wt_type_) {
        vidc_ctx* ctx = thread_ctx().ctx();
        std::vector<uint64_t> ids = gather_ids(il, offsets);
        DeviceArray d_ids(ctx, ids.data(), ids.size() * 8);
        VIDC_FAISS_CHECK(vidc_wt_build(ctx, nlist, offsets.data(), (const uint64_t*)d_ids.p, wt_type, &wt));
        compressed_ids_size_in_bytes = vidc_wt_size_in_bytes(wt); /* :369,372 */
        store_codes(il, nullptr);
    }
    ~WaveletTreeInvertedLists() override { vidc_wt_destroy(wt); }
    int decode_lists_device(vidc_ctx* ctx, uint64_t m, const uint64_t* ln, uint64_t* d_out, uint64_t* off) const override {
        return vidc_wt_decode_lists(ctx, wt, m, ln, d_out, off);
    }
    int decode_gather_device(vidc_ctx* ctx, uint64_t m, const uint64_t* ln, uint64_t n_items, const uint64_t* slot, const uint64_t* off,
                             int64_t* out) const override {
        return vidc_wt_decode_gather(ctx, wt, m, ln, n_items, slot, off, out);
    }
    faiss::idx_t get_single_id(size_t l, size_t offset) const override { /* wt.select(offset + 1, list_no), :377-379 */
        uint64_t ln = l, of = offset;
        int64_t id = -1;
        get_single_ids(1, &ln, &of, &id);
        return id;
    }
    void get_single_ids(uint64_t m, const uint64_t* ln, const uint64_t* of, faiss::idx_t* out) const override {
        thread_ctx().device_calls++;
        VIDC_FAISS_CHECK(vidc_wt_select(thread_ctx().ctx(), wt, m, ln, of, (int64_t*)out));
    }
};

/* ---------------------------------------------------------------------------------------------------------------
 * search_IVF_defer_id_decoding (custom_invlists_impl.cpp:407-526, wrapper custom_invlists.swig:67-122).
 * Identical up to the translation; the per-list OpenMP loop is one batched device decode when the index holds one of
 * the containers above (any other InvertedLists takes the reference's loop).                                     */
inline void search_IVF_defer_id_decoding(const faiss::IndexIVF& index, faiss::idx_t n, const float* x, int k,
                                         float* distances, faiss::idx_t* labels, bool decode_1by1 = false,
                                         uint8_t* codes = nullptr, bool include_listno = false) {
    using faiss::idx_t;
    std::unique_ptr<float[]> Dq(new float[n * index.nprobe]);
    std::unique_ptr<idx_t[]> Iq(new idx_t[n * index.nprobe]);
    FAISS_THROW_IF_NOT_MSG(index.parallel_mode == 3, "set the parallel mode to 3 otherwise search will be single-threaded");
    index.quantizer->search(n, x, index.nprobe, Dq.get(), Iq.get());
    index.search_preassigned(n, x, k, Iq.get(), Dq.get(), distances, labels, true);
    const faiss::InvertedLists* invlists = index.invlists;
    if (codes) { /* :434-462 */
        size_t code_size = index.code_size, code_size_1 = code_size + (include_listno ? index.coarse_code_size() : 0);
        for (idx_t ij = 0; ij < n * k; ij++) {
            idx_t key = labels[ij];
            uint8_t* code1 = codes + ij * code_size_1;
            if (key < 0) {
                std::memset(code1, -1, code_size_1);
                continue;
            }
            const uint8_t* cc = invlists->get_single_code(faiss::lo_listno(key), faiss::lo_offset(key));
            if (include_listno) {
                index.encode_listno(faiss::lo_listno(key), code1);
                code1 += code_size_1 - code_size;
            }
            std::memcpy(code1, cc, code_size);
        }
    }
    if (decode_1by1) { /* :464-474: n * k cheap selects -- here ONE batched call instead of n * k launches */
        if (auto* comp = dynamic_cast<const CompressedInvertedLists*>(invlists)) {
            std::vector<uint64_t> ln, of;
            std::vector<idx_t> where;
            for (idx_t i = 0; i < n * k; i++)
                if (labels[i] >= 0) {
                    ln.push_back((uint64_t)faiss::lo_listno(labels[i]));
                    of.push_back((uint64_t)faiss::lo_offset(labels[i]));
                    where.push_back(i);
                }
            std::vector<idx_t> got(where.size());
            if (!where.empty()) comp->get_single_ids(where.size(), ln.data(), of.data(), got.data());
            for (size_t j = 0; j < where.size(); j++) labels[where[j]] = got[j];
            return;
        }
        for (idx_t i = 0; i < n * k; i++)
            if (labels[i] >= 0) labels[i] = invlists->get_single_id(faiss::lo_listno(labels[i]), faiss::lo_offset(labels[i]));
        return;
    }
    /* touched lists (:477-502) */
    std::unordered_map<idx_t, size_t> slot;
    std::vector<uint64_t> lists;
    for (idx_t i = 0; i < n * k; i++)
        if (labels[i] >= 0 && slot.emplace(faiss::lo_listno(labels[i]), lists.size()).second)
            lists.push_back((uint64_t)faiss::lo_listno(labels[i]));
    if (auto* comp = dynamic_cast<const CompressedInvertedLists*>(invlists)) {
        /* ONE library call (:508-525): the touched lists are decoded on the device, labels[r] = ids[offset] is a device-side
         * gather, and 8 bytes per valid result cross PCIe instead of every touched list */
        std::vector<uint64_t> item_slot, item_off;
        std::vector<idx_t> where, got;
        for (idx_t i = 0; i < n * k; i++)
            if (labels[i] >= 0) {
                item_slot.push_back(slot[faiss::lo_listno(labels[i])]);
                item_off.push_back((uint64_t)faiss::lo_offset(labels[i]));
                where.push_back(i);
            }
        got.resize(where.size());
        comp->gather_ids_host(lists.size(), lists.data(), where.size(), item_slot.data(), item_off.data(), got.data());
        for (size_t j = 0; j < where.size(); j++) labels[where[j]] = got[j];
        return;
    }
    for (idx_t i = 0; i < n * k; i++) { /* foreign container: one get_ids per touched list like the reference */
        if (labels[i] < 0) continue;
        faiss::InvertedLists::ScopedIds sids(invlists, faiss::lo_listno(labels[i]));
        labels[i] = sids.get()[faiss::lo_offset(labels[i])];
    }
}

/* ---------------------------------------------------------------------------------------------------------------
 * graph containers (altid_impl.h:29-67).  The constructors read graph.data (N x K int32, -1 terminated rows) and set
 * data = nullptr like the reference (altid_impl.cpp:38,89,150).                                                  */
/* Faiss' NSG search asks for ONE row per expanded node through a virtual call (faiss/impl/NSG.cpp, search_on_graph) and the
 * reference decodes <= 64 ids inline in ~0.2 us (altid_impl.cpp:153-165).  A kernel launch + synchronisation + copy per
 * row costs ~30 us, so rows are served from a per-thread host cache (direct-mapped, VIDC_FAISS_ROW_CACHE_ROWS rows of K
 * int32) that is filled a frontier at a time: a miss on node i decodes row i and, with a second call, the rows of i's
 * neighbours that are not cached yet -- the nodes a greedy search expands next.  get_neighbors_batch is the same path
 * for callers that know their frontier (graph_search.py's batched search). */
#ifndef VIDC_FAISS_ROW_CACHE_ROWS
#define VIDC_FAISS_ROW_CACHE_ROWS 16384
#endif
#ifndef VIDC_FAISS_ROW_CACHES
#define VIDC_FAISS_ROW_CACHES 4 /* graphs a thread can alternate between without losing their cached rows */
#endif
#ifndef SWIG
struct RowCache {
    uint64_t owner = 0; /* id of the graph object the rows belong to (0 = unused) */
    uint64_t stamp = 0; /* last use (least recently used cache of the thread is recycled) */
    int K = 0;
    std::vector<int32_t> tag;     /* node held by a slot, -1 = empty */
    std::vector<uint32_t> count;
    std::vector<int32_t> rows;    /* slot * K */
    size_t hits = 0, misses = 0;
    /* scratch of the miss path (kept: a miss allocates nothing once the vectors have grown) */
    std::vector<int> want;
    std::vector<size_t> slots;
    std::vector<int32_t> buf;
    std::vector<uint32_t> cnt;
    std::vector<uint64_t> nd;
    void reset(uint64_t o, int k) {
        owner = o; K = k;
        tag.assign(VIDC_FAISS_ROW_CACHE_ROWS, -1);
        count.assign(VIDC_FAISS_ROW_CACHE_ROWS, 0);
        rows.assign((size_t)VIDC_FAISS_ROW_CACHE_ROWS * (size_t)k, -1);
        hits = misses = 0;
    }
    static size_t slot_of(int node) { return ((uint32_t)node * 2654435761u >> 7) & (VIDC_FAISS_ROW_CACHE_ROWS - 1); }
};
/* the calling thread's cache for graph object `owner`: a thread that alternates between graphs (e.g. comparing containers
 * query by query) keeps one cache per graph, up to VIDC_FAISS_ROW_CACHES of them */
struct RowCacheSet {
    RowCache c[VIDC_FAISS_ROW_CACHES];
    uint64_t clock = 0;
    size_t resets = 0;
    RowCache& get(uint64_t owner, int K) {
        RowCache* lru = &c[0];
        for (RowCache& x : c) {
            if (x.owner == owner && x.K == K) { x.stamp = ++clock; return x; }
            if (x.stamp < lru->stamp) lru = &x;
        }
        lru->reset(owner, K);
        lru->stamp = ++clock;
        resets++;
        return *lru;
    }
};
inline RowCacheSet& thread_row_caches() {
    thread_local RowCacheSet s;
    return s;
}
static_assert((VIDC_FAISS_ROW_CACHE_ROWS & (VIDC_FAISS_ROW_CACHE_ROWS - 1)) == 0, "power of two");
#endif /* SWIG */

struct CompressedNSGGraph : faiss::nsg::Graph<int32_t> {
    size_t compressed_ids_size_in_bytes = 0, overhead_in_bytes = 0;
    uint64_t object_id; /* cache key: addresses are reused, ids are not */
    explicit CompressedNSGGraph(const faiss::nsg::Graph<int32_t>& g) : faiss::nsg::Graph<int32_t>(g.data, g.N, g.K) {
        static std::atomic<uint64_t> next{1};
        object_id = next++;
    }
    /* decode m rows into device memory d_out (m x K int32, -1 padded), edge counts into counts[m] */
    virtual int decode_rows_device(vidc_ctx* ctx, uint64_t m, const uint64_t* nodes, int32_t* d_out, uint32_t* counts) const = 0;

    /* rows of m nodes in ONE library call: out[m * K] (-1 padded), counts[m] (may be NULL) */
    void get_neighbors_batch(size_t m, const int* nodes, int32_t* out, uint32_t* counts = nullptr) const {
        if (!m) return;
        ThreadCtx& t = thread_ctx();
        RowCache& c = thread_row_caches().get(object_id, K); /* (its scratch vectors: no allocation per call) */
        c.nd.assign(nodes, nodes + m);
        c.cnt.resize(m);
        int32_t* d = (int32_t*)t.staging(m * (size_t)K * 4);
        t.device_calls++;
        VIDC_FAISS_CHECK(decode_rows_device(t.ctx(), m, c.nd.data(), d, c.cnt.data()));
        VIDC_FAISS_CHECK(vidc_copy_d2h(t.ctx(), out, d, m * (size_t)K * 4));
        if (counts) std::copy(c.cnt.begin(), c.cnt.begin() + m, counts);
    }
    /* one row through the cache; returns the edge count */
    size_t cached_row(int i, int32_t* neighbors) const {
        RowCache& c = thread_row_caches().get(object_id, K);
        size_t s = RowCache::slot_of(i);
        if (c.tag[s] != i) {
            c.misses++;
            uint32_t cnt_i = 0;
            get_neighbors_batch(1, &i, &c.rows[s * (size_t)K], &cnt_i);
            c.count[s] = cnt_i;
            c.tag[s] = i;
            /* the frontier: neighbours of i that are not cached (a neighbour whose slot is i's own stays out) */
            c.want.clear();
            c.slots.clear();
            for (uint32_t j = 0; j < c.count[s] && j < (uint32_t)K; j++) {
                int v = c.rows[s * (size_t)K + j];
                if (v < 0 || v >= N) continue;
                size_t sv = RowCache::slot_of(v);
                if (sv == s || c.tag[sv] == v) continue;
                bool dup = false;
                for (size_t q : c.slots) dup |= q == sv;
                if (dup) continue;
                c.want.push_back(v);
                c.slots.push_back(sv);
            }
            if (!c.want.empty()) {
                const size_t nw = c.want.size();
                c.buf.resize(nw * (size_t)K);
                get_neighbors_batch(nw, c.want.data(), c.buf.data(), nullptr); /* (edge counts stay in c.cnt) */
                for (size_t q = 0; q < nw; q++) {
                    std::memcpy(&c.rows[c.slots[q] * (size_t)K], &c.buf[q * (size_t)K], (size_t)K * 4);
                    c.count[c.slots[q]] = c.cnt[q];
                    c.tag[c.slots[q]] = c.want[q];
                }
            }
        } else {
            c.hits++;
        }
        std::memcpy(neighbors, &c.rows[s * (size_t)K], (size_t)K * 4);
        return c.count[s];
    }
};

/* CompactBitNSGGraph (altid_impl.cpp:20-51) */
struct CompactBitGraph : CompressedNSGGraph {
    vidc_compact* c = nullptr;
    int bits = 0;
    size_t stride = 0;
    explicit CompactBitGraph(const faiss::nsg::Graph<int32_t>& g) : CompressedNSGGraph(g) {
        vidc_ctx* ctx = thread_ctx().ctx();
        DeviceArray d(ctx, g.data, (size_t)N * K * 4);
        VIDC_FAISS_CHECK(vidc_compact_rows_encode(ctx, N, K, (const int32_t*)d.p, &c));
        bits = (int)vidc_compact_bits(c);
        stride = vidc_compact_stride(c);
        compressed_ids_size_in_bytes = vidc_compact_size_in_bytes(c);
        data = nullptr;
    }
    ~CompactBitGraph() override { vidc_compact_destroy(c); }
    int decode_rows_device(vidc_ctx* ctx, uint64_t m, const uint64_t* nodes, int32_t* d, uint32_t* cnt) const override {
        return vidc_compact_rows_decode(ctx, c, m, nodes, d, cnt);
    }
    size_t get_neighbors(int i, int32_t* neighbors) const override { return cached_row(i, neighbors); } /* edge count (:41-50) */
};

/* EliasFanoNSGGraph (altid_impl.cpp:53-101) */
struct EliasFanoGraph : CompressedNSGGraph {
    vidc_ef* ef = nullptr;
    explicit EliasFanoGraph(const faiss::nsg::Graph<int32_t>& g) : CompressedNSGGraph(g) {
        vidc_ctx* ctx = thread_ctx().ctx();
        DeviceArray d(ctx, g.data, (size_t)N * K * 4);
        VIDC_FAISS_CHECK(vidc_ef_encode_rows(ctx, N, K, (const int32_t*)d.p, &ef));
        compressed_ids_size_in_bytes = vidc_ef_compressed_bytes(ef); /* :86,88 */
        overhead_in_bytes += N * std::ceil(std::log2(N)) / 8.0; /* :56-57, a size_t incremented twice by a double */
        overhead_in_bytes += N * std::ceil(std::log2(N)) / 8.0;
        data = nullptr;
    }
    ~EliasFanoGraph() override { vidc_ef_destroy(ef); }
    int decode_rows_device(vidc_ctx* ctx, uint64_t m, const uint64_t* nodes, int32_t* d, uint32_t* cnt) const override {
        return vidc_ef_decode_rows(ctx, ef, m, nodes, (uint32_t)K, d, cnt);
    }
    size_t get_neighbors(int i, int32_t* neighbors) const override { return cached_row(i, neighbors); } /* num_elements (:92-101) */
};

/* ROCNSGGraph (altid_impl.cpp:103-165) */
struct ROCGraph : CompressedNSGGraph {
    vidc_roc* roc = nullptr;
    std::vector<uint32_t> num_outgoing_edges; /* altid_impl.h:61 */
    explicit ROCGraph(const faiss::nsg::Graph<int32_t>& g) : CompressedNSGGraph(g), num_outgoing_edges(g.N) {
        vidc_ctx* ctx = thread_ctx().ctx();
        DeviceArray d(ctx, g.data, (size_t)N * K * 4);
        VIDC_FAISS_CHECK(vidc_roc_encode_rows(ctx, N, K, (const int32_t*)d.p, VIDC_PREC_REFERENCE, 0, &roc));
        VIDC_FAISS_CHECK(vidc_roc_list_info(roc, num_outgoing_edges.data(), nullptr, nullptr, nullptr, nullptr));
        /* :148 adds ans_states[i].size() for EVERY node (8 bytes even for an empty state) */
        compressed_ids_size_in_bytes = 8 * (size_t)N + 4 * vidc_roc_total_words(roc);
        overhead_in_bytes += N * std::ceil(std::log2(N)) / 8.0; /* :105 */
        data = nullptr;
    }
    ~ROCGraph() override { vidc_roc_destroy(roc); }
    int decode_rows_device(vidc_ctx* ctx, uint64_t m, const uint64_t* nodes, int32_t* d, uint32_t* cnt) const override {
        return vidc_roc_decode_rows(ctx, roc, m, nodes, (uint32_t)K, d, cnt);
    }
    size_t get_neighbors(int i, int32_t* neighbors) const override {
        cached_row(i, neighbors);
        return K; /* the reference returns K with only num_outgoing_edges[i] slots written (altid_impl.cpp:163-164);
                     here the remaining slots hold -1, which the NSG search stops at */
    }
};

}  // namespace vidc_faiss

#ifdef VIDC_FAISS_REFERENCE_NAMES
/* The names the reference's harness dictionaries look up (bench_invlists.py:19-25, graph_dynamic_bench_invlists.py:21-26,
 * search_ivf_qinco.py:502-523); define the macro in the SWIG module INSTEAD of including custom_invlists_impl.h / altid_impl.h.
 * Derived structs with the reference's constructor signatures (custom_invlists_impl.h:48,67,85,117; altid_impl.h:33,46,58), spelled
 * out rather than inherited (`using Base::Base;` is only wrapped by SWIG >= 4.2): SWIG generates one Python proxy class per struct,
 * under exactly these names.  An alias declaration would not do that (SWIG treats it as a typedef); the static_asserts below keep
 * one from coming back. */
struct CompressedIDInvertedListsFenwickTree : vidc_faiss::ROCInvertedLists {
    explicit CompressedIDInvertedListsFenwickTree(const faiss::InvertedLists& il) : vidc_faiss::ROCInvertedLists(il) {}
};
struct CompressedIDInvertedListsEliasFano : vidc_faiss::EliasFanoInvertedLists {
    explicit CompressedIDInvertedListsEliasFano(const faiss::InvertedLists& il) : vidc_faiss::EliasFanoInvertedLists(il) {}
};
struct CompressedIDInvertedListsPackedBits : vidc_faiss::PackedBitsInvertedLists {
    explicit CompressedIDInvertedListsPackedBits(const faiss::InvertedLists& il) : vidc_faiss::PackedBitsInvertedLists(il) {}
};
struct CompressedIDInvertedListsWaveletTree : vidc_faiss::WaveletTreeInvertedLists {
    explicit CompressedIDInvertedListsWaveletTree(const faiss::InvertedLists& il, int wt_type = 0)
            : vidc_faiss::WaveletTreeInvertedLists(il, wt_type) {}
};
struct CompactBitNSGGraph : vidc_faiss::CompactBitGraph {
    explicit CompactBitNSGGraph(const faiss::nsg::Graph<int32_t>& graph) : vidc_faiss::CompactBitGraph(graph) {}
};
struct EliasFanoNSGGraph : vidc_faiss::EliasFanoGraph {
    explicit EliasFanoNSGGraph(const faiss::nsg::Graph<int32_t>& graph) : vidc_faiss::EliasFanoGraph(graph) {}
};
struct ROCNSGGraph : vidc_faiss::ROCGraph {
    explicit ROCNSGGraph(const faiss::nsg::Graph<int32_t>& graph) : vidc_faiss::ROCGraph(graph) {}
};
/* (a function needs no proxy class: the %inline wrapper of custom_invlists.swig:67-84 calls it by this name) */
inline void search_IVF_defer_id_decoding(const faiss::IndexIVF& index, faiss::idx_t n, const float* x, int k, float* distances,
                                         faiss::idx_t* labels, bool decode_1by1 = false, uint8_t* codes = nullptr,
                                         bool include_listno = false) {
    vidc_faiss::search_IVF_defer_id_decoding(index, n, x, k, distances, labels, decode_1by1, codes, include_listno);
}
#ifndef SWIG
#include <type_traits>
static_assert(!std::is_same<CompressedIDInvertedListsFenwickTree, vidc_faiss::ROCInvertedLists>::value &&
                      !std::is_same<CompressedIDInvertedListsEliasFano, vidc_faiss::EliasFanoInvertedLists>::value &&
                      !std::is_same<CompressedIDInvertedListsPackedBits, vidc_faiss::PackedBitsInvertedLists>::value &&
                      !std::is_same<CompressedIDInvertedListsWaveletTree, vidc_faiss::WaveletTreeInvertedLists>::value &&
                      !std::is_same<CompactBitNSGGraph, vidc_faiss::CompactBitGraph>::value &&
                      !std::is_same<EliasFanoNSGGraph, vidc_faiss::EliasFanoGraph>::value &&
                      !std::is_same<ROCNSGGraph, vidc_faiss::ROCGraph>::value,
              "the reference names must be classes of their own (SWIG wraps an alias under the aliased name only)");
static_assert(std::is_base_of<faiss::InvertedLists, CompressedIDInvertedListsFenwickTree>::value &&
                      std::is_base_of<faiss::nsg::Graph<int32_t>, ROCNSGGraph>::value,
              "the reference names plug into Faiss through the reference's base classes");
#endif
#endif
