/* c_abi_smoke.c -- the raw C-ABI of include/vidc.h from a plain C program (no ctypes, no C++): KAT1 of the reference's
 * own test (test_codec.cpp:26-29: ids {12351235, 49024902, 17781778, 36663666}, precision 26) encoded and decoded, checked
 * against the stream the compiled reference produces (head 22906489391, stack [2873710996, 612831110,
 * 2733404530], decode order {12351235, 17781778, 49024902, 36663666}).
 *   build: gcc -std=c11 -I include tests/c_abi_smoke.c -L <pkg> -lvidc -Wl,-rpath,<pkg> -o c_abi_smoke */
#include <inttypes.h>
#include <stdio.h>
#include <string.h>

#include "vidc.h"

#define CHECK(call)                                                                   \
    do {                                                                              \
        int st_ = (call);                                                             \
        if (st_ != VIDC_OK) { printf("FAILED %s -> %d: %s\n", #call, st_, vidc_last_error()); return 1; } \
    } while (0)
#define REQUIRE(c)                                                        \
    do {                                                                  \
        if (!(c)) { printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } \
    } while (0)

int main(void) {
    const uint64_t ids[4] = {12351235ull, 49024902ull, 17781778ull, 36663666ull};
    const uint64_t offsets[2] = {0, 4};
    const uint32_t want_words[3] = {2873710996u, 612831110u, 2733404530u};
    const uint64_t want_order[4] = {12351235ull, 17781778ull, 49024902ull, 36663666ull};
    vidc_ctx *ctx = NULL;
    vidc_roc *roc = NULL;
    void *d_ids = NULL, *d_out = NULL;
    CHECK(vidc_ctx_create(-1, &ctx));
    CHECK(vidc_dev_alloc(ctx, sizeof ids, &d_ids));
    CHECK(vidc_dev_alloc(ctx, sizeof ids, &d_out));
    CHECK(vidc_copy_h2d(ctx, d_ids, ids, sizeof ids));
    CHECK(vidc_roc_encode(ctx, 1, offsets, (const uint64_t *)d_ids, 26, VIDC_ROC_WANT_PERM, &roc));
    uint32_t size = 0, prec = 0, nwords = 0, draws = 0, words[8], perm[4];
    uint64_t head = 0, decoded[4], out_off[2], one = 0;
    CHECK(vidc_roc_list_info(roc, &size, &prec, &head, &nwords, &draws));
    REQUIRE(size == 4 && prec == 26 && nwords == 3 && draws == 0);
    REQUIRE(head == 22906489391ull);
    CHECK(vidc_roc_export_words(ctx, roc, 0, words, 8));
    REQUIRE(memcmp(words, want_words, sizeof want_words) == 0);
    REQUIRE(vidc_roc_compressed_bytes(roc) == 8 + 4 * 3); /* ANSState::size(), codec.h:42-44 */
    CHECK(vidc_roc_perm(ctx, roc, perm));
    for (int i = 0; i < 4; i++) REQUIRE(ids[perm[i]] == want_order[i]);
    CHECK(vidc_roc_decode_all(ctx, roc, (uint64_t *)d_out));
    CHECK(vidc_copy_d2h(ctx, decoded, d_out, sizeof decoded));
    REQUIRE(memcmp(decoded, want_order, sizeof want_order) == 0);
    REQUIRE(vidc_roc_last_decode_nonclean(roc) == 0);
    CHECK(vidc_roc_decode_lists(ctx, roc, 1, &one, (uint64_t *)d_out, out_off));
    REQUIRE(out_off[0] == 0 && out_off[1] == 4);
    /* error convention: a negative status + a message, nothing thrown */
    vidc_roc *bad = NULL;
    const uint64_t bad_off[2] = {4, 0};
    REQUIRE(vidc_roc_encode(ctx, 1, bad_off, (const uint64_t *)d_ids, VIDC_PREC_REFERENCE, 0, &bad) != VIDC_OK && bad == NULL);
    REQUIRE(strlen(vidc_last_error()) > 0);
    vidc_roc_destroy(roc);
    vidc_dev_free(ctx, d_ids);
    vidc_dev_free(ctx, d_out);
    vidc_ctx_destroy(ctx);
    printf("c abi smoke ok: head %" PRIu64 ", words %u %u %u\n", head, words[0], words[1], words[2]);
    return 0;
}
