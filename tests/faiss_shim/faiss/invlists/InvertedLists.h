// interface shim (tests/faiss_shim/README.md): the InvertedLists surface of SURVEY.md Appendix B
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include <faiss/impl/FaissAssert.h>

namespace faiss {
using idx_t = int64_t;

struct InvertedLists {
    size_t nlist;
    size_t code_size;
    InvertedLists(size_t nlist_, size_t code_size_) : nlist(nlist_), code_size(code_size_) {}
    virtual ~InvertedLists() {}
    virtual size_t list_size(size_t list_no) const = 0;
    virtual const uint8_t* get_codes(size_t list_no) const = 0;
    virtual const idx_t* get_ids(size_t list_no) const = 0;
    virtual void release_codes(size_t, const uint8_t*) const {}
    virtual void release_ids(size_t, const idx_t*) const {}
    /* the lists a search is about to visit (IndexIVF::search_preassigned announces them before it scans); default: nothing */
    virtual void prefetch_lists(const idx_t* /*list_nos*/, int /*nlist*/) const {}
    virtual idx_t get_single_id(size_t list_no, size_t offset) const {
        const idx_t* ids = get_ids(list_no);
        idx_t id = ids[offset];
        release_ids(list_no, ids);
        return id;
    }
    virtual const uint8_t* get_single_code(size_t list_no, size_t offset) const {
        return get_codes(list_no) + offset * code_size;
    }
    virtual size_t add_entries(size_t list_no, size_t n_entry, const idx_t* ids, const uint8_t* code) = 0;
    virtual void update_entries(size_t list_no, size_t offset, size_t n_entry, const idx_t* ids, const uint8_t* code) = 0;
    virtual void resize(size_t list_no, size_t new_size) = 0;
    size_t compute_ntotal() const {
        size_t t = 0;
        for (size_t l = 0; l < nlist; l++) t += list_size(l);
        return t;
    }
    struct ScopedIds {
        const InvertedLists* il;
        const idx_t* ids;
        size_t list_no;
        ScopedIds(const InvertedLists* il_, size_t l) : il(il_), ids(il_->get_ids(l)), list_no(l) {}
        const idx_t* get() { return ids; }
        idx_t operator[](size_t i) const { return ids[i]; }
        ~ScopedIds() { il->release_ids(list_no, ids); }
    };
    struct ScopedCodes {
        const InvertedLists* il;
        const uint8_t* codes;
        size_t list_no;
        ScopedCodes(const InvertedLists* il_, size_t l) : il(il_), codes(il_->get_codes(l)), list_no(l) {}
        const uint8_t* get() { return codes; }
        ~ScopedCodes() { il->release_codes(list_no, codes); }
    };
};

struct ReadOnlyInvertedLists : InvertedLists {
    ReadOnlyInvertedLists(size_t nlist_, size_t code_size_) : InvertedLists(nlist_, code_size_) {}
    size_t add_entries(size_t, size_t, const idx_t*, const uint8_t*) override { FAISS_THROW_IF_NOT_MSG(false, "not implemented"); return 0; }
    void update_entries(size_t, size_t, size_t, const idx_t*, const uint8_t*) override { FAISS_THROW_IF_NOT_MSG(false, "not implemented"); }
    void resize(size_t, size_t) override { FAISS_THROW_IF_NOT_MSG(false, "not implemented"); }
};

struct ArrayInvertedLists : InvertedLists {
    std::vector<std::vector<uint8_t>> codes;
    std::vector<std::vector<idx_t>> ids;
    ArrayInvertedLists(size_t nlist_, size_t code_size_) : InvertedLists(nlist_, code_size_), codes(nlist_), ids(nlist_) {}
    size_t list_size(size_t l) const override { return ids[l].size(); }
    const uint8_t* get_codes(size_t l) const override { return codes[l].data(); }
    const idx_t* get_ids(size_t l) const override { return ids[l].data(); }
    size_t add_entries(size_t l, size_t n, const idx_t* ids_in, const uint8_t* code) override {
        size_t o = ids[l].size();
        ids[l].insert(ids[l].end(), ids_in, ids_in + n);
        codes[l].insert(codes[l].end(), code, code + n * code_size);
        return o;
    }
    void update_entries(size_t l, size_t offset, size_t n, const idx_t* ids_in, const uint8_t* code) override {
        std::memcpy(&ids[l][offset], ids_in, n * sizeof(idx_t));
        std::memcpy(&codes[l][offset * code_size], code, n * code_size);
    }
    void resize(size_t l, size_t new_size) override {
        ids[l].resize(new_size);
        codes[l].resize(new_size * code_size);
    }
};
}  // namespace faiss
