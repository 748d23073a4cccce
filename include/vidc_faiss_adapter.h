/*
 * vidc_faiss_adapter.h -- OPTIONAL C++ adapter that puts libvidc behind faiss::InvertedLists / faiss::nsg::Graph.
 *
 * Compiles only where the Faiss headers are installed (they are not in the build image of this repository, so this
 * header is exercised by no test here; INTEGRATION.md walks through it).  It is what a maintainer of the reference
 * adds next to custom_invlists_impl.h / altid_impl.h so that bench_invlists.py and graph_dynamic_bench_invlists.py
 * can select the GPU codecs through the same SWIG modules:
 *     %include "vidc_faiss_adapter.h"     (custom_invlists.swig:61, altid.swig)
 */
#pragma once
#if defined(__has_include)
#if __has_include(<faiss/invlists/InvertedLists.h>) && __has_include(<faiss/impl/NSG.h>)
#define VIDC_HAVE_FAISS 1
#endif
#endif

#ifdef VIDC_HAVE_FAISS
#include <faiss/impl/FaissAssert.h>
#include <faiss/impl/NSG.h>
#include <faiss/invlists/InvertedLists.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "vidc.h"

namespace vidc_faiss {

#define VIDC_FAISS_CHECK(expr) FAISS_THROW_IF_NOT_MSG((expr) == VIDC_OK, vidc_last_error())

/* common part: CSR of the source lists on the device (custom_invlists_impl.cpp:156-160 borrows ids the same way) */
struct DeviceLists {
    vidc_ctx* ctx = nullptr;
    std::vector<uint64_t> offsets;
    void* d_ids = nullptr;
    explicit DeviceLists(const faiss::InvertedLists& il) : offsets(il.nlist + 1, 0) {
        VIDC_FAISS_CHECK(vidc_ctx_create(-1, &ctx));
        for (size_t l = 0; l < il.nlist; l++) offsets[l + 1] = offsets[l] + il.list_size(l);
        std::vector<uint64_t> ids(offsets.back());
        for (size_t l = 0; l < il.nlist; l++) {
            faiss::InvertedLists::ScopedIds s(&il, l);
            if (il.list_size(l)) std::memcpy(ids.data() + offsets[l], s.get(), il.list_size(l) * 8);
        }
        VIDC_FAISS_CHECK(vidc_dev_alloc(ctx, ids.size() * 8, &d_ids));
        VIDC_FAISS_CHECK(vidc_copy_h2d(ctx, d_ids, ids.data(), ids.size() * 8));
    }
    void drop_ids() {
        if (d_ids) vidc_dev_free(ctx, d_ids);
        d_ids = nullptr;
    }
    ~DeviceLists() {
        drop_ids();
        vidc_ctx_destroy(ctx);
    }
};

/* replaces CompressedIDInvertedListsFenwickTree (custom_invlists_impl.cpp:133-223) */
struct ROCInvertedLists : faiss::ReadOnlyInvertedLists {
    DeviceLists dev;
    mutable std::mutex ctx_mu;  // a vidc_ctx serves one host thread at a time; Faiss calls get_ids from OpenMP threads
    vidc_roc* roc = nullptr;
    std::vector<std::vector<uint8_t>> codes_all;
    size_t compressed_ids_size_in_bytes = 0, overhead_in_bytes = 0;

    explicit ROCInvertedLists(const faiss::InvertedLists& il)
            : faiss::ReadOnlyInvertedLists(il.nlist, il.code_size), dev(il), codes_all(il.nlist) {
        VIDC_FAISS_CHECK(vidc_roc_encode(dev.ctx, nlist, dev.offsets.data(), (const uint64_t*)dev.d_ids,
                                         VIDC_PREC_REFERENCE, VIDC_ROC_WANT_PERM, &roc));
        dev.drop_ids();
        compressed_ids_size_in_bytes = vidc_roc_compressed_bytes(roc);
        std::vector<uint32_t> perm(dev.offsets.back());
        VIDC_FAISS_CHECK(vidc_roc_perm(dev.ctx, roc, perm.data()));
        for (size_t l = 0; l < nlist; l++) { /* codes follow the sampling order (:188-193) */
            faiss::InvertedLists::ScopedCodes c(&il, l);
            size_t n = il.list_size(l);
            codes_all[l].resize(n * code_size);
            for (size_t i = 0; i < n; i++)
                std::memcpy(&codes_all[l][i * code_size], c.get() + (size_t)perm[dev.offsets[l] + i] * code_size,
                            code_size);
        }
    }
    ~ROCInvertedLists() override { vidc_roc_destroy(roc); }

    size_t list_size(size_t l) const override { return dev.offsets[l + 1] - dev.offsets[l]; }
    const uint8_t* get_codes(size_t l) const override { return codes_all[l].data(); }
    const faiss::idx_t* get_ids(size_t l) const override { /* :210-219 */
        size_t n = list_size(l);
        if (n == 0) return nullptr;
        auto* out = new faiss::idx_t[n];
        std::lock_guard<std::mutex> guard(ctx_mu);
        void* d = nullptr;
        uint64_t off[2], ln = l;
        VIDC_FAISS_CHECK(vidc_dev_alloc(dev.ctx, n * 8, &d));
        int st = vidc_roc_decode_lists(dev.ctx, roc, 1, &ln, (uint64_t*)d, off);
        if (st == VIDC_OK) st = vidc_copy_d2h(dev.ctx, out, d, n * 8);
        vidc_dev_free(dev.ctx, d);
        VIDC_FAISS_CHECK(st);
        return out;
    }
    void release_ids(size_t, const faiss::idx_t* ids) const override { delete[] ids; } /* :221-223 */
};

/* replaces ROCNSGGraph (altid_impl.cpp:103-165) */
struct ROCNSGGraph : faiss::nsg::Graph<int32_t> {
    vidc_ctx* ctx = nullptr;
    mutable std::mutex ctx_mu;  // get_neighbors may be called from several search threads
    vidc_roc* roc = nullptr;
    void* d_row = nullptr;
    explicit ROCNSGGraph(const faiss::nsg::Graph<int32_t>& g) : faiss::nsg::Graph<int32_t>(g.data, g.N, g.K) {
        VIDC_FAISS_CHECK(vidc_ctx_create(-1, &ctx));
        void* d = nullptr;
        VIDC_FAISS_CHECK(vidc_dev_alloc(ctx, (size_t)N * K * 4, &d));
        VIDC_FAISS_CHECK(vidc_copy_h2d(ctx, d, g.data, (size_t)N * K * 4));
        int st = vidc_roc_encode_rows(ctx, N, K, (const int32_t*)d, VIDC_PREC_REFERENCE, 0, &roc);
        vidc_dev_free(ctx, d);
        VIDC_FAISS_CHECK(st);
        VIDC_FAISS_CHECK(vidc_dev_alloc(ctx, (size_t)K * 4, &d_row));
        data = nullptr; /* altid_impl.cpp:150 */
    }
    ~ROCNSGGraph() override {
        vidc_dev_free(ctx, d_row);
        vidc_roc_destroy(roc);
        vidc_ctx_destroy(ctx);
    }
    size_t get_neighbors(int i, int32_t* neighbors) const override {
        std::lock_guard<std::mutex> guard(ctx_mu);
        uint64_t node = (uint64_t)i;
        uint32_t count = 0;
        VIDC_FAISS_CHECK(vidc_roc_decode_rows(ctx, roc, 1, &node, K, (int32_t*)d_row, &count));
        VIDC_FAISS_CHECK(vidc_copy_d2h(ctx, neighbors, d_row, (size_t)K * 4));
        return count; /* the reference returns K with only `count` slots written (altid_impl.cpp:163-164) */
    }
};

}  // namespace vidc_faiss
#endif /* VIDC_HAVE_FAISS */
