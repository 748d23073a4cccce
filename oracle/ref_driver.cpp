// ref_driver.cpp -- thin extern "C" driver around the REAL reference codec.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes the
// reference headers where they lie under /root/reference (-I flags in oracle/Makefile)
// and is linked with /root/reference/custom_invlist_cpp/codec.cpp into
// oracle/_ref/libvidc_ref.so (git-ignored).  It exists so that
//   (1) the clean-room restatement (vidc_oracle.c) can be pinned bit-for-bit against the
//       reference's own functions (compress/decompress, codec.h:47-52), and
//   (2) bench.py can time the reference's own CPU path ("cpu_baseline.kind = reference").
//
// The container classes themselves (custom_invlists_impl.cpp) need Faiss headers and are
// NOT buildable here; rc_container_encode() drives the same reference primitives in the
// order the container constructor does (custom_invlists_impl.cpp:163-192): shuffled
// insertion into the reference FenwickTree, then pop-index / remove / push-id per element.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <random>
#include <tuple>
#include <vector>

#include "codec.h"         // /root/reference/custom_invlist_cpp/codec.h
#include "fenwick_tree.h"  // /root/reference/fenwick_tree_cpp/src/fenwick_tree.h

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {
using Sym = std::tuple<uint64_t, uint32_t>;  // (id, input position) stands for (id, code pointer)

void container_encode(size_t n, const uint64_t* ids, int precision, uint32_t shuffle_seed,
                      ANSState& st, uint64_t* order_out, uint32_t* perm_out) {
    FenwickTree<Sym> ftree;
    std::vector<uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0u);
    std::mt19937 g(shuffle_seed);
    std::shuffle(idx.begin(), idx.end(), g);
    for (uint32_t i : idx) ftree.insert_then_forward_lookup(Sym(ids[i], i));
    for (size_t i = 0; i < n; i++) {
        uint32_t nmax = (uint32_t)(n - i);
        size_t k = pop_with_finer_precision(st, nmax);
        auto range = ftree.reverse_lookup_then_remove((int)k);
        uint64_t id = std::get<0>(range.ftree->symbol);
        codec_push(st, id, precision);
        if (order_out) order_out[i] = id;
        if (perm_out) perm_out[i] = std::get<1>(range.ftree->symbol);
    }
}

int export_state(const ANSState& st, uint64_t* head, uint32_t* words, size_t cap, size_t* nwords) {
    *head = st.head;
    *nwords = st.stack.size();
    if (st.stack.size() > cap) return -1;
    if (!st.stack.empty()) std::memcpy(words, st.stack.data(), st.stack.size() * 4);
    return 0;
}
}  // namespace

extern "C" {

// reference compress() (codec.cpp:123-138): BST filled in DATA order (O(n^2) on sorted input).
int rc_compress(size_t n, const uint64_t* ids, int precision, uint64_t* head, uint32_t* words,
                size_t cap, size_t* nwords) {
    ANSState st;
    compress(n, ids, st, precision);
    return export_state(st, head, words, cap, nwords);
}

// container-order encode (custom_invlists_impl.cpp:163-192) on the reference primitives.
int rc_container_encode(size_t n, const uint64_t* ids, int precision, uint32_t shuffle_seed,
                        uint64_t* head, uint32_t* words, size_t cap, size_t* nwords,
                        uint64_t* order_out, uint32_t* perm_out) {
    ANSState st;
    container_encode(n, ids, precision, shuffle_seed, st, order_out, perm_out);
    return export_state(st, head, words, cap, nwords);
}

// reference decompress() (codec.cpp:140-152) from an exported state with a FRESH mt19937(1234)
// (true whenever the encoder drew no underflow word; rc_roundtrip covers the other case).
int rc_decompress(uint64_t head, const uint32_t* words, size_t nwords, size_t n, int precision,
                  uint64_t* out, uint64_t* end_head, uint32_t* end_words, size_t end_cap,
                  size_t* end_nwords) {
    ANSState st;
    st.head = head;
    st.stack.assign(words, words + nwords);
    decompress(st, n, out, precision);
    return export_state(st, end_head, end_words, end_cap, end_nwords);
}

// encode (container order) then decode with the SAME ANSState object, like get_ids does on a
// copy of the stored state (custom_invlists_impl.cpp:216-217), so encoder-side mt draws carry over.
int rc_roundtrip(size_t n, const uint64_t* ids, int precision, uint64_t* head, uint32_t* words,
                 size_t cap, size_t* nwords, uint64_t* decoded) {
    ANSState st;
    container_encode(n, ids, precision, 12345u, st, nullptr, nullptr);
    int rc = export_state(st, head, words, cap, nwords);
    ANSState copy(st);
    decompress(copy, n, decoded, precision);
    return rc;
}

int rc_omp_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// Times the reference CPU path over a CSR set of lists, parallel over lists like
// custom_invlists_impl.cpp:147 (encode) and :508 (decode of touched lists).
// schedule(dynamic) is used (the reference uses the default static schedule) and stated in DESIGN.md.
// Returns the number of lists whose decoded SET differs from the input (n > 65536 quirk).
size_t rc_bench_lists(size_t nlist, const uint64_t* offsets, const uint64_t* ids, int threads,
                      double* t_enc, double* t_dec, uint64_t* sum_bytes) {
    std::vector<ANSState> states(nlist);
    std::vector<int> prec(nlist, 0);
    // the container always produces the sampling permutation (the vector codes are reordered by it,
    // custom_invlists_impl.cpp:188-193): the timed encode fills it like the GPU path of bench.py does
    std::vector<uint32_t> perm(offsets[nlist] ? offsets[nlist] : 1);
    (void)threads;
    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic) num_threads(threads)
    for (size_t l = 0; l < nlist; l++) {
        size_t n = offsets[l + 1] - offsets[l];
        if (!n) continue;
        const uint64_t* p = ids + offsets[l];
        int max_id = (int)*std::max_element(p, p + n);
        prec[l] = (int)(uint64_t)std::ceil(std::log2(max_id));
        container_encode(n, p, prec[l], (uint32_t)(l * 2654435761u + 1u), states[l], nullptr, perm.data() + offsets[l]);
    }
    auto t1 = std::chrono::steady_clock::now();
    uint64_t bytes = 0;
    for (size_t l = 0; l < nlist; l++)
        if (offsets[l + 1] > offsets[l]) bytes += states[l].size();
    std::vector<uint64_t> outs(offsets[nlist] ? offsets[nlist] : 1);
    auto t2 = std::chrono::steady_clock::now();
#pragma omp parallel for schedule(dynamic) num_threads(threads)
    for (size_t l = 0; l < nlist; l++) {
        size_t n = offsets[l + 1] - offsets[l];
        if (!n) continue;
        ANSState copy(states[l]);
        decompress(copy, n, outs.data() + offsets[l], prec[l]);
    }
    auto t3 = std::chrono::steady_clock::now();
    size_t bad = 0;
    for (size_t l = 0; l < nlist; l++) {
        size_t n = offsets[l + 1] - offsets[l];
        if (!n) continue;
        std::vector<uint64_t> a(ids + offsets[l], ids + offsets[l + 1]);
        std::vector<uint64_t> b(outs.begin() + offsets[l], outs.begin() + offsets[l + 1]);
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        if (a != b) bad++;
    }
    *t_enc = std::chrono::duration<double>(t1 - t0).count();
    *t_dec = std::chrono::duration<double>(t3 - t2).count();
    *sum_bytes = bytes;
    return bad;
}

}  // extern "C"
