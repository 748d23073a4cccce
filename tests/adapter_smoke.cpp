// adapter_smoke.cpp -- drives include/vidc_faiss_adapter.h the way Faiss and the reference's harnesses do
// (test_compressed_ivfs.py:43-156, test_altid.py:17-44), compiled against tests/faiss_shim in this image.
//   build: g++ -std=c++17 -I tests/faiss_shim -I include tests/adapter_smoke.cpp -L <pkg> -lvidc -Wl,-rpath,<pkg> -fopenmp
// Prints "adapter smoke ok" and returns 0 when every check holds.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>
#include <vector>
#include <thread>
#include <atomic>

#define VIDC_FAISS_REFERENCE_NAMES  // the reference's class names at global scope (bench_invlists.py:19-25 runs unchanged)
#include "vidc_faiss_adapter.h"

#define REQUIRE(c)                                                       \
    do {                                                                 \
        if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

using faiss::idx_t;

template <class Container>
static int check_container(faiss::IndexIVF& index, const faiss::ArrayInvertedLists& ref, const std::vector<float>& xq, int nq,
                           int k, const std::vector<idx_t>& Iref, const std::vector<float>& Dref, const char* name) {
    Container comp(ref);
    REQUIRE(comp.nlist == ref.nlist && comp.code_size == ref.code_size);
    REQUIRE(comp.compressed_ids_size_in_bytes > 0);
    size_t total = 0;
    for (size_t l = 0; l < ref.nlist; l++) {  // test_compressed_ivfs.py:66-79
        REQUIRE(comp.list_size(l) == ref.list_size(l));
        const idx_t* ids = comp.get_ids(l);
        if (ref.list_size(l) == 0) { REQUIRE(ids == nullptr); continue; }
        std::vector<idx_t> a(ids, ids + ref.list_size(l)), b(ref.ids[l]);
        // the codes of the container are in the order of its ids
        for (size_t i = 0; i < a.size(); i++) {
            size_t j = std::find(b.begin(), b.end(), a[i]) - b.begin();
            REQUIRE(j < b.size());
            REQUIRE(std::memcmp(comp.get_codes(l) + i * comp.code_size, ref.codes[l].data() + j * ref.code_size, ref.code_size) == 0);
        }
        for (size_t i = 0; i < a.size(); i += 7) REQUIRE(comp.get_single_id(l, i) == a[i]);
        comp.release_ids(l, ids);
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        REQUIRE(a == b);
        total += a.size();
    }
    REQUIRE(total == ref.compute_ntotal());
    index.replace_invlists(&comp, false);
    std::vector<idx_t> I(nq * k);
    std::vector<float> D(nq * k);
    {
        // non-deferred: get_ids of the probed lists (from OpenMP threads in Faiss).  search_preassigned announces them
        // (prefetch_lists), the first get_ids decodes all of them: ONE library call for the nq * nprobe probed lists
        const size_t calls0 = vidc_faiss::thread_ctx().device_calls;
        index.search(nq, xq.data(), k, D.data(), I.data());
        REQUIRE(I == Iref && D == Dref);
        REQUIRE(vidc_faiss::thread_ctx().device_calls - calls0 == 1);
        // a list outside the announcement and a cleared announcement take the per-list path
        comp.clear_prefetch();
        const size_t calls1 = vidc_faiss::thread_ctx().device_calls;
        size_t l0 = 0;
        while (ref.list_size(l0) == 0) l0++;
        const idx_t* ids = comp.get_ids(l0);
        REQUIRE(vidc_faiss::thread_ctx().device_calls - calls1 == 1);
        comp.release_ids(l0, ids);
    }
    {
        // IndexIVF::search cuts a batch into one slice per thread and every slice announces its own lists, concurrently: the
        // announcements must not overwrite each other (each slice: ONE library call for its lists), the ids must be right, and when
        // the slices are done nothing stays cached
        REQUIRE(comp.prefetch_live_announcements() == 0);
        std::vector<size_t> nonempty;
        for (size_t l = 0; l < ref.nlist; l++)
            if (ref.list_size(l)) nonempty.push_back(l);
        const int T = 4;
        std::vector<int> bad(T, 0);
        std::vector<size_t> calls(T, 0);
        std::atomic<int> announced{0};
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                std::vector<idx_t> mine;  // overlapping subsets, with repeats and a -1 like a real probe list
                for (size_t i = (size_t)t; i < nonempty.size(); i += 2) mine.push_back((idx_t)nonempty[i]);
                for (size_t i = 0; i < std::min<size_t>(5, nonempty.size()); i++) mine.push_back((idx_t)nonempty[i]);
                mine.push_back(-1);
                const size_t c0 = vidc_faiss::thread_ctx().device_calls;
                comp.prefetch_lists(mine.data(), (int)mine.size());
                announced++;
                while (announced.load() < T) std::this_thread::yield();  // every slice has announced before any of them scans
                for (idx_t l : mine) {
                    if (l < 0) continue;
                    const idx_t* ids = comp.get_ids((size_t)l);
                    const idx_t* want = ref.get_ids((size_t)l);
                    std::vector<idx_t> a(ids, ids + ref.list_size(l)), b(want, want + ref.list_size(l));
                    std::sort(a.begin(), a.end());
                    std::sort(b.begin(), b.end());
                    if (a != b) bad[t]++;
                    comp.release_ids((size_t)l, ids);
                    ref.release_ids((size_t)l, want);
                }
                calls[t] = vidc_faiss::thread_ctx().device_calls - c0;
            });
        for (auto& x : th) x.join();
        for (int t = 0; t < T; t++) {
            REQUIRE(bad[t] == 0);
            REQUIRE(calls[t] <= 1);  // (a slice whose lists another slice's cache already held makes no call at all)
        }
        REQUIRE(comp.prefetch_live_announcements() == 0);
    }
    for (int one_by_one = 0; one_by_one < 2; one_by_one++) {  // test_compressed_ivfs.py:128-156
        std::fill(I.begin(), I.end(), -7);
        const size_t calls0 = vidc_faiss::thread_ctx().device_calls;
        const uint64_t d2h0 = vidc_ctx_d2h_bytes(vidc_faiss::thread_ctx().ctx());
        vidc_faiss::search_IVF_defer_id_decoding(index, nq, xq.data(), k, D.data(), I.data(), one_by_one != 0);
        REQUIRE(I == Iref && D == Dref);
        // the scatter labels[r] = ids[offset] runs on the device; what crosses PCIe is 8 bytes per valid result
        // (every touched list came down before: custom_invlists_impl.cpp:508-525 restated with whole-list copies)
        size_t valid = 0;
        for (idx_t v : Iref) valid += v >= 0;
        REQUIRE(vidc_ctx_d2h_bytes(vidc_faiss::thread_ctx().ctx()) - d2h0 == 8 * valid);
        // the n * k selects of decode_1by1 (custom_invlists_impl.cpp:464-474) and the touched lists of the batched form are ONE
        // library call each, not one launch per result
        REQUIRE(vidc_faiss::thread_ctx().device_calls - calls0 <= 1);
    }
    std::vector<uint8_t> codes((size_t)nq * k * (comp.code_size + index.coarse_code_size()));
    vidc_faiss::search_IVF_defer_id_decoding(index, nq, xq.data(), k, D.data(), I.data(), false, codes.data(), true);
    REQUIRE(I == Iref);
    // concurrent get_ids like Faiss' scan threads (custom_invlists_impl.cpp:467,508)
    int bad = 0;
#pragma omp parallel for num_threads(4) reduction(+ : bad)
    for (int l = 0; l < (int)ref.nlist; l++) {
        const idx_t* ids = comp.get_ids(l);
        if (!ids) continue;
        std::multiset<idx_t> a(ids, ids + ref.list_size(l)), b(ref.ids[l].begin(), ref.ids[l].end());
        bad += a != b;
        comp.release_ids(l, ids);
    }
    REQUIRE(bad == 0);
    index.replace_invlists(const_cast<faiss::ArrayInvertedLists*>(&ref), false);
    printf("  %-28s ok: %zu bytes for %zu ids\n", name, comp.compressed_ids_size_in_bytes, total);
    return 0;
}

template <class G>
static int check_graph(const std::vector<int32_t>& rows, int N, int K, const char* name, bool sorted, bool returns_K) {
    std::vector<int32_t> copy(rows);  // the EF constructor of the reference sorts the source rows in place
    faiss::nsg::Graph<int32_t> src(copy.data(), N, K);
    G g(src);
    REQUIRE(g.data == nullptr && g.N == N && g.K == K && g.compressed_ids_size_in_bytes > 0);
    std::vector<int32_t> nb(K);
    for (int i = 0; i < N; i++) {
        int d = 0;
        while (d < K && rows[(size_t)i * K + d] >= 0) d++;
        size_t got = g.get_neighbors(i, nb.data());
        REQUIRE(got == (returns_K ? (size_t)K : (size_t)d));
        std::vector<int32_t> a(nb.begin(), nb.begin() + d), b(rows.begin() + (size_t)i * K, rows.begin() + (size_t)i * K + d);
        if (sorted) { REQUIRE(std::is_sorted(a.begin(), a.end())); }
        else if (!returns_K) REQUIRE(a == b);  // compact keeps the order (altid_impl.cpp:28-37)
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        REQUIRE(a == b);  // test_altid.py:33-40
    }
    // get_neighbors_batch: a frontier in one library call
    {
        std::vector<int> nodes;
        for (int i = 0; i < N; i += 3) nodes.push_back(i);
        std::vector<int32_t> out(nodes.size() * (size_t)K);
        std::vector<uint32_t> cnt(nodes.size());
        const size_t calls0 = vidc_faiss::thread_ctx().device_calls;
        g.get_neighbors_batch(nodes.size(), nodes.data(), out.data(), cnt.data());
        REQUIRE(vidc_faiss::thread_ctx().device_calls - calls0 == 1);
        for (size_t q = 0; q < nodes.size(); q++) {
            g.get_neighbors(nodes[q], nb.data());
            REQUIRE(std::equal(nb.begin(), nb.end(), out.begin() + q * (size_t)K));
        }
    }
    // a greedy walk the way the NSG search expands nodes (one virtual get_neighbors per node): rows come from the per-thread
    // cache, filled a frontier at a time
    {
        auto& cache = vidc_faiss::thread_row_caches().get(g.object_id, K);
        const size_t h0 = cache.hits, m0 = cache.misses, calls0 = vidc_faiss::thread_ctx().device_calls;
        std::mt19937 rng(7);
        size_t steps = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int walk = 0; walk < 200; walk++) {
            int cur = (int)(rng() % N);
            for (int hop = 0; hop < 30; hop++, steps++) {
                size_t d = 0;
                g.get_neighbors(cur, nb.data());
                while (d < (size_t)K && nb[d] >= 0) d++;
                if (!d) break;
                cur = nb[rng() % d];
            }
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        printf("  %-28s walk: %zu get_neighbors calls, %.2f us each, %zu cache hits / %zu misses, %zu library calls\n", name, steps,
               us / (double)steps, cache.hits - h0, cache.misses - m0, vidc_faiss::thread_ctx().device_calls - calls0);
        REQUIRE(vidc_faiss::thread_ctx().device_calls - calls0 <= 2 * (cache.misses - m0));
    }
    // a thread that alternates between two graph objects keeps one row cache per object: no cache is rebuilt when the thread
    // switches, and the second round over the same nodes is served mostly from the caches (a frontier fill of the first round may
    // have replaced a few of the direct-mapped slots)
    {
        std::vector<int32_t> copy2(rows);
        faiss::nsg::Graph<int32_t> src2(copy2.data(), N, K);
        G g2(src2);
        std::vector<int32_t> nb2(K);
        for (int round = 0; round < 2; round++) {
            const size_t resets0 = vidc_faiss::thread_row_caches().resets;
            const size_t miss0 = vidc_faiss::thread_row_caches().get(g.object_id, K).misses + vidc_faiss::thread_row_caches().get(g2.object_id, K).misses;
            for (int i = 0; i < 64; i++) {
                size_t a = g.get_neighbors(i, nb.data()), b = g2.get_neighbors(i, nb2.data());
                REQUIRE(a == b && nb == nb2);
            }
            if (round == 0) REQUIRE(vidc_faiss::thread_row_caches().resets - resets0 <= 1);  // g2's cache; g's exists already
            else {
                REQUIRE(vidc_faiss::thread_row_caches().resets == resets0);
                const size_t miss1 = vidc_faiss::thread_row_caches().get(g.object_id, K).misses + vidc_faiss::thread_row_caches().get(g2.object_id, K).misses;
                REQUIRE(miss1 - miss0 <= 32);  // (128 lookups; with one shared cache every one of them would miss)
            }
        }
    }
    printf("  %-28s ok: %zu bytes for %d nodes\n", name, g.compressed_ids_size_in_bytes, N);
    return 0;
}

int main() {
    try {
        const int d = 8, nlist = 16, nb = 6000, nq = 9, k = 10;
        std::mt19937 rng(123);
        std::normal_distribution<float> nd;
        std::vector<float> cent(nlist * d), xb((size_t)nb * d), xq((size_t)nq * d);
        for (auto& v : cent) v = 3 * nd(rng);
        for (int i = 0; i < nb; i++) for (int t = 0; t < d; t++) xb[(size_t)i * d + t] = cent[(rng() % nlist) * d + t] + nd(rng);
        for (int i = 0; i < nq; i++) for (int t = 0; t < d; t++) xq[(size_t)i * d + t] = cent[(rng() % nlist) * d + t] + nd(rng);
        faiss::IndexFlatL2 quant(d);
        quant.add(nlist - 1, cent.data());  // one centroid short: the last list stays empty
        faiss::IndexIVF index(&quant, d, nlist);
        index.add(nb, xb.data());
        index.nprobe = 4;
        index.parallel_mode = 3;
        auto* ref = static_cast<faiss::ArrayInvertedLists*>(index.invlists);
        index.own_invlists = false;
        std::vector<idx_t> Iref(nq * k);
        std::vector<float> Dref(nq * k);
        index.search(nq, xq.data(), k, Dref.data(), Iref.data());
        printf("inverted lists (%d ids, %d lists):\n", nb, nlist);
        // (through the reference's own class names, VIDC_FAISS_REFERENCE_NAMES)
        if (check_container<CompressedIDInvertedListsFenwickTree>(index, *ref, xq, nq, k, Iref, Dref, "CompressedIDInvertedListsFenwickTree")) return 1;
        if (check_container<CompressedIDInvertedListsEliasFano>(index, *ref, xq, nq, k, Iref, Dref, "CompressedIDInvertedListsEliasFano")) return 1;
        if (check_container<CompressedIDInvertedListsPackedBits>(index, *ref, xq, nq, k, Iref, Dref, "CompressedIDInvertedListsPackedBits")) return 1;
        if (check_container<CompressedIDInvertedListsWaveletTree>(index, *ref, xq, nq, k, Iref, Dref, "CompressedIDInvertedListsWaveletTree")) return 1;
        {  // the batched random access of the ROC container checks (list, offset) against the list sizes
            CompressedIDInvertedListsFenwickTree comp(*ref);
            size_t l0 = 0;
            while (ref->list_size(l0) == 0) l0++;
            uint64_t ln = l0, of = ref->list_size(l0);
            idx_t got = -1;
            bool threw = false;
            try { comp.get_single_ids(1, &ln, &of, &got); }
            catch (const faiss::FaissException&) { threw = true; }
            REQUIRE(threw);
            of--;
            comp.get_single_ids(1, &ln, &of, &got);
            const idx_t* ids = comp.get_ids(l0);
            REQUIRE(got == ids[of]);
            comp.release_ids(l0, ids);
            // the reference names are classes of their own, usable wherever the reference's are (SWIG wraps them by these names)
            faiss::InvertedLists* as_base = &comp;
            REQUIRE(dynamic_cast<vidc_faiss::ROCInvertedLists*>(as_base) != nullptr);
        }
        {  // a foreign container still works through the deferred search (reference loop)
            std::vector<idx_t> I(nq * k);
            std::vector<float> D(nq * k);
            vidc_faiss::search_IVF_defer_id_decoding(index, nq, xq.data(), k, D.data(), I.data());
            REQUIRE(I == Iref);
            index.parallel_mode = 0;
            bool threw = false;
            try { vidc_faiss::search_IVF_defer_id_decoding(index, nq, xq.data(), k, D.data(), I.data()); }
            catch (const faiss::FaissException&) { threw = true; }  // custom_invlists_impl.cpp:420-422
            REQUIRE(threw);
        }
        delete ref;
        const int N = 3000, K = 48;
        std::vector<int32_t> rows((size_t)N * K, -1);
        for (int i = 0; i < N; i++) {
            int deg = i == 5 ? 0 : (i == 6 ? K : (int)(rng() % (K + 1)));
            std::set<int32_t> s;
            while ((int)s.size() < deg) {  // no powers of two: a row whose LARGEST id is one decodes lossily in the reference's
                int32_t v = (int32_t)(rng() % N);  // ROC (precision one bit short), which this library reproduces
                if (v & (v - 1)) s.insert(v);
            }
            std::vector<int32_t> v(s.begin(), s.end());
            std::shuffle(v.begin(), v.end(), rng);
            std::copy(v.begin(), v.end(), rows.begin() + (size_t)i * K);
        }
        printf("graphs (%d nodes, K = %d):\n", N, K);
        if (check_graph<CompactBitNSGGraph>(rows, N, K, "CompactBitNSGGraph", false, false)) return 1;
        if (check_graph<EliasFanoNSGGraph>(rows, N, K, "EliasFanoNSGGraph", true, false)) return 1;
        if (check_graph<ROCNSGGraph>(rows, N, K, "ROCNSGGraph", false, true)) return 1;
    } catch (const std::exception& e) {
        printf("FAILED with exception: %s\n", e.what());
        return 1;
    }
    printf("adapter smoke ok\n");
    return 0;
}
