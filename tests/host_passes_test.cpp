// The host's AVX2 passes over CSR offsets against their scalar forms (csrc/host_passes.h); built and run by tests/test_host_passes.py.
#include <cstdio>
#include <random>
#include <vector>

#include "../vector_db_id_compression_amd/csrc/host_passes.h"

using namespace vidc;

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s (line %d, case %s)\n", #c, __LINE__, what); fails++; } } while (0)

static void one(const std::vector<uint64_t> &off, const char *what) {
    const uint64_t nlist = off.size() - 1;
    {   // offsets pass (ROC): stats + staging copy
        std::vector<uint64_t> h1(off.size(), 7), h2(off.size(), 9);
        OffsetsPass a, b;
        offsets_pass_scalar(off.data(), 0, nlist, h1.data(), a);
        offsets_pass(off.data(), nlist, h2.data(), b);
        CHECK(a.bad == b.bad);
        CHECK(h1 == h2 && h1 == off);
        if (!a.bad) {
            CHECK(a.nonempty == b.nonempty); CHECK(a.max_n == b.max_n); CHECK(a.min_n == b.min_n); CHECK(a.desc == b.desc);
            CHECK(a.prev == b.prev);
        }
    }
    for (uint32_t bits : {0u, 1u, 13u, 32u, 33u, 64u}) {  // lengths pass (Elias-Fano: bits = 0; packed bits)
        LengthsPass a;
        lengths_pass_scalar(off.data(), 0, nlist, 9u, bits, a);
        const LengthsPass b = lengths_pass(off.data(), nlist, 9u, bits);
        CHECK(a.wide == b.wide);
        if (!a.wide) { CHECK(a.max_n == b.max_n); CHECK(a.nchunks == b.nchunks); CHECK(a.bytes == b.bytes); CHECK(a.words == b.words); }
    }
}

int main() {
    std::mt19937_64 rng(12345);
    for (uint64_t nlist = 0; nlist <= 40; nlist++)
        for (int rep = 0; rep < 20; rep++) {
            std::vector<uint64_t> off(nlist + 1, rep % 3 ? 0 : 5);
            for (uint64_t l = 0; l < nlist; l++) off[l + 1] = off[l] + (rng() % 4 == 0 ? 0 : rng() % 3000);
            one(off, "ragged");
        }
    for (uint64_t nlist : {1000ull, 4097ull, 65536ull, 65539ull}) {
        std::vector<uint64_t> off(nlist + 1, 0);
        for (uint64_t l = 0; l < nlist; l++) off[l + 1] = off[l] + 256;
        one(off, "equal");
        for (uint64_t l = 0; l < nlist; l++) off[l + 1] = off[l] + (nlist - l);  // longest first
        one(off, "descending");
        for (uint64_t l = 0; l < nlist; l++) off[l + 1] = off[l] + (rng() % 70000);
        one(off, "zipf-like");
        std::vector<uint64_t> bad = off;
        bad[nlist / 2 + 1] = bad[nlist / 2] - (nlist > 2 ? 1 : 0);  // offsets that decrease
        for (uint64_t l = nlist / 2 + 1; l < nlist; l++) bad[l + 1] = bad[l] + 3;
        one(bad, "decreasing");
        std::vector<uint64_t> wide = off;
        for (uint64_t l = nlist - 1; l < nlist; l++) wide[l + 1] = wide[l] + (1ull << 33);  // one list of 2^33 ids
        one(wide, "wide tail");
        wide = off;
        for (uint64_t l = 5; l < nlist; l++) wide[l + 1] += (1ull << 32);
        one(wide, "wide middle");
        std::vector<uint64_t> toolong = off;
        for (uint64_t l = 7; l < nlist; l++) toolong[l + 1] += VIDC_ROC_MAX_LIST + 1ull;  // beyond the ROC limit, below 2^32
        one(toolong, "beyond the ROC limit");
    }
    std::printf(fails ? "host passes: %d checks failed\n" : "host passes ok\n", fails);
    return fails ? 1 : 0;
}
