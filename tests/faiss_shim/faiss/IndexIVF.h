// interface shim (tests/faiss_shim/README.md): the IndexIVF members search_IVF_defer_id_decoding touches
// (custom_invlists_impl.cpp:407-462), with a brute-force float32 scan standing in for the real one
#pragma once
#include <algorithm>
#include <cstring>
#include <limits>
#include <utility>
#include <vector>

#include <faiss/invlists/DirectMap.h>
#include <faiss/invlists/InvertedLists.h>

namespace faiss {
struct Index {
    int d;
    explicit Index(int d_) : d(d_) {}
    virtual ~Index() {}
    virtual void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const = 0;
};

struct IndexFlatL2 : Index {  // the coarse quantizer
    std::vector<float> xb;
    explicit IndexFlatL2(int d_) : Index(d_) {}
    void add(idx_t n, const float* x) { xb.insert(xb.end(), x, x + n * d); }
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override {
        idx_t nb = (idx_t)(xb.size() / d);
        for (idx_t q = 0; q < n; q++) {
            std::vector<std::pair<float, idx_t>> all(nb);
            for (idx_t j = 0; j < nb; j++) {
                float s = 0;
                for (int t = 0; t < d; t++) { float e = x[q * d + t] - xb[j * d + t]; s += e * e; }
                all[j] = {s, j};
            }
            std::sort(all.begin(), all.end());
            for (idx_t i = 0; i < k; i++) {
                distances[q * k + i] = i < nb ? all[i].first : std::numeric_limits<float>::infinity();
                labels[q * k + i] = i < nb ? all[i].second : -1;
            }
        }
    }
};

struct IndexIVF : Index {  // "IVFx,Flat": codes are the float32 vectors
    Index* quantizer;
    size_t nlist, nprobe = 1, code_size;
    InvertedLists* invlists;
    bool own_invlists = true;
    int parallel_mode = 0;
    IndexIVF(Index* q, int d_, size_t nlist_)
            : Index(d_), quantizer(q), nlist(nlist_), code_size(sizeof(float) * d_), invlists(new ArrayInvertedLists(nlist_, sizeof(float) * d_)) {}
    ~IndexIVF() override { if (own_invlists) delete invlists; }
    void replace_invlists(InvertedLists* il, bool own) {
        if (own_invlists) delete invlists;
        invlists = il;
        own_invlists = own;
    }
    size_t coarse_code_size() const {
        size_t nl = nlist - 1, nbyte = 0;
        while (nl > 0) { nbyte++; nl >>= 8; }
        return nbyte;
    }
    void encode_listno(idx_t list_no, uint8_t* code) const {
        size_t nl = nlist - 1;
        while (nl > 0) { *code++ = list_no & 0xff; list_no >>= 8; nl >>= 8; }
    }
    void add(idx_t n, const float* x, idx_t first_id = 0) {
        std::vector<float> D(n);
        std::vector<idx_t> L(n);
        quantizer->search(n, x, 1, D.data(), L.data());
        for (idx_t i = 0; i < n; i++) {
            idx_t id = first_id + i;
            invlists->add_entries(L[i], 1, &id, (const uint8_t*)(x + i * d));
        }
    }
    void search_preassigned(idx_t n, const float* x, idx_t k, const idx_t* assign, const float* /*centroid_dis*/,
                            float* distances, idx_t* labels, bool store_pairs) const {
        invlists->prefetch_lists(assign, (int)(n * nprobe));  // (Faiss announces the probed lists before scanning them)
        for (idx_t q = 0; q < n; q++) {
            std::vector<std::pair<float, idx_t>> cand;
            for (size_t p = 0; p < nprobe; p++) {
                idx_t l = assign[q * nprobe + p];
                if (l < 0) continue;
                size_t ls = invlists->list_size(l);
                InvertedLists::ScopedCodes codes(invlists, l);
                const float* v = (const float*)codes.get();
                const idx_t* ids = nullptr;
                if (!store_pairs) ids = invlists->get_ids(l);
                for (size_t j = 0; j < ls; j++) {
                    float s = 0;
                    for (int t = 0; t < d; t++) { float e = x[q * d + t] - v[j * d + t]; s += e * e; }
                    cand.push_back({s, store_pairs ? (idx_t)lo_build(l, j) : ids[j]});
                }
                if (ids) invlists->release_ids(l, ids);
            }
            std::sort(cand.begin(), cand.end());
            for (idx_t i = 0; i < k; i++) {
                bool ok = i < (idx_t)cand.size();
                distances[q * k + i] = ok ? cand[i].first : std::numeric_limits<float>::infinity();
                labels[q * k + i] = ok ? cand[i].second : -1;
            }
        }
    }
    void search(idx_t n, const float* x, idx_t k, float* distances, idx_t* labels) const override {
        std::vector<float> Dq(n * nprobe);
        std::vector<idx_t> Iq(n * nprobe);
        quantizer->search(n, x, nprobe, Dq.data(), Iq.data());
        search_preassigned(n, x, k, Iq.data(), Dq.data(), distances, labels, false);
    }
};
}  // namespace faiss
