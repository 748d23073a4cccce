/*
 * vidc_oracle.h -- CPU restatement of the reference's per-list ID codecs.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path
 * (vector_db_id_compression_amd/, include/) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the reported CPU baseline.
 *
 * Parity status:
 *   ROC / ANS      : PINNED against the compiled reference (oracle/_ref, built
 *                    from /root/reference/custom_invlist_cpp/codec.cpp) through
 *                    tests/golden/roc_golden.json (KAT1-3 + a case matrix).
 *   packed bits    : pinned only against the in-tree random-access reader
 *                    (custom_invlists_impl.cpp:35-58); faiss::BitstringWriter is
 *                    not in the reference tree.  Layout = LSB-first, little endian.
 *   Elias-Fano     : decoded arrays + bit counts pinned by the formulas in
 *                    elias_fano.hpp:22-57; in-memory word layout follows
 *                    succinct@669eebb (not vendored) => "parity unpinned" for layout.
 *   wavelet tree   : select semantics only (sdsl absent) => "parity unpinned".
 */
#ifndef VIDC_ORACLE_H
#define VIDC_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- mt19937 (codec.h:16-18,32-40: the stack-underflow word source) ---- */
void vo_mt19937_table(uint32_t seed, uint32_t *out, size_t count);

/* ---- precision rule (custom_invlists_impl.cpp:163-164, altid_impl.cpp:124-125) ---- */
int vo_precision_from_max_id(int32_t max_id);

/* ---- ROC / ANS (codec.cpp:21-152) ---- */
typedef struct {
    uint64_t head;      /* codec.h:14 */
    uint32_t *stack;    /* codec.h:15, push order */
    size_t nstack;
    size_t cap;
    uint32_t mt_draws;  /* number of mt19937(1234) words consumed so far (codec.h:32-40) */
} vo_ans_state;

void vo_ans_init(vo_ans_state *st);
void vo_ans_free(vo_ans_state *st);
void vo_ans_copy(vo_ans_state *dst, const vo_ans_state *src);

uint64_t vo_idx_pop(vo_ans_state *st, uint64_t nmax);                /* codec.cpp:21-42 */
void vo_idx_push(vo_ans_state *st, uint64_t sym, uint64_t nmax);     /* codec.cpp:44-63 */
void vo_id_push(vo_ans_state *st, uint64_t sym, int precision);      /* codec.cpp:92-105 */
uint64_t vo_id_pop(vo_ans_state *st, int precision);                 /* codec.cpp:107-121 */

/* ROC encode of one list (codec.cpp:123-138 + custom_invlists_impl.cpp:178-192).
 * ids: n values (duplicates ordered by input position, like the (id,codeptr) tuple).
 * order_out[i] = i-th sampled id, perm_out[i] = its position in ids[] (either may be NULL). */
void vo_roc_encode(size_t n, const uint64_t *ids, int precision, vo_ans_state *st,
                   uint64_t *order_out, uint32_t *perm_out);
/* ROC decode (codec.cpp:140-152); st is consumed (copy first if needed). */
void vo_roc_decode(vo_ans_state *st, size_t n, int precision, uint64_t *out);

/* ---- packed bits (custom_invlists_impl.cpp:35-58,64-113; altid_impl.cpp:20-51) ---- */
int vo_packed_bits_for(uint64_t ntotal);                              /* :68-70 */
void vo_packed_write(uint8_t *code, size_t bit_offset, uint64_t x, int nbit);
uint64_t vo_packed_read(const uint8_t *code, size_t bit_offset, int nbit);

/* ---- Elias-Fano (elias_fano.hpp:22-57,141-145,210-261) ---- */
typedef struct {
    uint64_t universe;   /* m_n  (= max id) */
    uint64_t m;          /* number of elements */
    int l;               /* low bits per element */
    uint64_t low_nbits;  /* m * l */
    uint64_t high_nbits; /* (m + 1) + (universe >> l) + 1 */
    uint64_t *low;       /* 64-bit words, LSB-first */
    uint64_t *high;
} vo_ef;
int vo_ef_low_bits(uint64_t universe, uint64_t m);                    /* :28 */
void vo_ef_build(vo_ef *ef, uint64_t universe, uint64_t m, const uint64_t *sorted_ids);
void vo_ef_free(vo_ef *ef);
uint64_t vo_ef_select(const vo_ef *ef, uint64_t i);                   /* :141-145 */
void vo_ef_decode_all(const vo_ef *ef, uint64_t *out);                /* :210-261 */

/* ---- wavelet tree semantics (custom_invlists_impl.cpp:346-379): id = select(offset+1, list_no) ---- */
/* list_nos[id] = list number, ntotal entries; returns position of (k+1)-th occurrence of c, or -1 */
int64_t vo_wt_select(const uint32_t *list_nos, size_t ntotal, uint32_t c, uint64_t k);

/* ---- container-level helpers over CSR (offsets[nlist+1], ids[ntotal]) ---- */
/* Encode + decode every list, OpenMP over lists (schedule(dynamic)); returns seconds in t_enc/t_dec.
 * sum_bytes = sum over non-empty lists of 8 + 4*nstack (codec.h:42-44).  returns #lists failing round-trip as a set */
size_t vo_roc_bench_lists(size_t nlist, const uint64_t *offsets, const uint64_t *ids, int threads,
                          double *t_enc, double *t_dec, uint64_t *sum_bytes);
int vo_omp_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
