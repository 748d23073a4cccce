/*
 * vidc.h -- C-ABI of the MI355X-native vector-ID codec library (libvidc.so).
 *
 * Drop-in boundary for the per-inverted-list ID codecs of
 * facebookresearch/vector_db_id_compression.  Plain pointers and sizes only; no Faiss,
 * torch or C++ types cross this boundary.  Every entry point names the reference
 * interface it replaces (file:line under /root/reference).
 *
 * Data model: a set of lists is a CSR pair  offsets[nlist+1] (host, uint64) + ids[ntotal].
 * IDs are faiss::idx_t viewed as uint64 (custom_invlists_impl.cpp:79,159,217,248) or int32
 * graph rows (altid_impl.cpp:26-37).  "dev" pointers are HIP device pointers valid on the
 * context's device; everything else is host memory.  Out-buffers are caller-allocated.
 *
 * Error convention: every call returns VIDC_OK (0) or a negative vidc_status;
 * vidc_last_error() returns a thread-local message (replaces FAISS_THROW_IF_NOT /
 * FaissException, custom_invlists_impl.cpp:87,420-422 and custom_invlists.swig:38-57).
 * There is NO CPU fallback: without a HIP device vidc_ctx_create fails with VIDC_ERR_NO_DEVICE.
 *
 * Residency: compressed objects live on the device, including their per-list metadata (offsets, word counts,
 * precisions, ...).  Sizes come back as scalars; the *_list_info / *_export_* calls copy the per-list arrays to the
 * host on first use.  Device memory of objects and scratch comes from a block cache shared by the context and the
 * objects created through it (steady-state calls neither hipMalloc nor hipFree); objects may be destroyed before
 * or after their context.
 *
 * Environment (test hooks, read per call): VIDC_NO_LANE=1 / VIDC_FORCE_LANE=1 never / always use the
 * lane-per-list ROC kernels (default: only for calls with thousands of short lists), VIDC_FORCE_GENERAL=1 routes
 * every list through the general wave-per-list kernels, VIDC_HOST_THREADS=n caps the host threads used to plan calls
 * with >= 131072 lists (default 8), VIDC_TRACE=1 prints host-side phase times.  The bit streams do not depend on
 * any of them.
 */
#ifndef VIDC_H
#define VIDC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VIDC_VERSION 100

typedef enum {
    VIDC_OK = 0,
    VIDC_ERR_INVALID = -1,     /* bad argument */
    VIDC_ERR_NO_DEVICE = -2,   /* no HIP device / HIP runtime error at init */
    VIDC_ERR_HIP = -3,         /* HIP runtime error (message has hipGetErrorString) */
    VIDC_ERR_DOMAIN = -4,      /* input outside the codec's domain (id >= 2^31, id >= ntotal, list too long) */
    VIDC_ERR_OVERFLOW = -5,    /* internal arena / mt19937 table exhausted */
    VIDC_ERR_UNSUPPORTED = -6
} vidc_status;

const char *vidc_last_error(void);
int vidc_version(void);

/* ------------------------------------------------------------------ context */
typedef struct vidc_ctx vidc_ctx;
/* device < 0 : current HIP device.  Creates the context's own stream. */
int vidc_ctx_create(int device, vidc_ctx **out);
void vidc_ctx_destroy(vidc_ctx *ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream).  NULL is the legacy default stream (what
 * torch uses unless told otherwise); vidc_ctx_reset_stream goes back to the context's own stream. */
int vidc_ctx_set_stream(vidc_ctx *ctx, void *hip_stream);
int vidc_ctx_reset_stream(vidc_ctx *ctx);
int vidc_ctx_synchronize(vidc_ctx *ctx);
/* A context caches the device and pinned-host blocks its calls used (steady-state encode / decode calls neither
 * allocate nor free; a 10^9-id decode leaves several GB of scratch behind).  vidc_ctx_trim synchronises the
 * context's stream and releases every cached block that is not in use; *freed_bytes (optional) = device + pinned
 * bytes returned to the driver.  Blocks held by live objects are untouched.  The process-wide cache of emptied host arrays
 * (kept so that the next 10^6-list encode does not page-fault ~50 MB in again; bounded at 256 MB per element type) is
 * released as well. */
int vidc_ctx_trim(vidc_ctx *ctx, uint64_t *freed_bytes);
/* Test aid: fill every device block the context's cache hands out with 0xFF first (also: VIDC_POOL_POISON=1 in the environment when the
 * context is created).  Streams must not depend on what a block held before. */
int vidc_ctx_debug_pool_poison(vidc_ctx *ctx, int on);
/* Streams the kernel classes of one large ROC call are spread over: 8 when the PROCESS was started with the ROCm runtime
 * variable GPU_MAX_HW_QUEUES >= 8 (HIP multiplexes a process's streams onto that many hardware queues, default 4, and reads
 * the variable when it initialises: export it before the process starts -- for a Faiss-hosted process in the environment of
 * the Python / C++ program that loads Faiss, not after `import faiss`), otherwise 4.  The published S2 numbers are the 8-stream
 * mode; the 4-stream mode is ~10 % slower on calls of ~10^6 lists and identical on small calls.  Returns 0 for NULL. */
int vidc_ctx_class_streams(const vidc_ctx *ctx);
/* Device memory helpers for hosts without their own allocator (Python uses torch tensors instead). */
int vidc_dev_alloc(vidc_ctx *ctx, size_t bytes, void **dev_ptr);
int vidc_dev_free(vidc_ctx *ctx, void *dev_ptr);
int vidc_copy_h2d(vidc_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
int vidc_copy_d2h(vidc_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes);
/* Bytes of id payload this context has copied device -> host so far (vidc_copy_d2h, the *_get selects, *_decode_gather):
 * what a deferred search costs on PCIe -- 8 bytes per result through *_decode_gather, whole lists through
 * *_decode_lists + vidc_copy_d2h.  Metadata read-backs (sizes, status words, list_info) are not counted. */
uint64_t vidc_ctx_d2h_bytes(const vidc_ctx *ctx);

/* ------------------------------------------------------- ROC (bits-back ANS) */
/* Replaces: ANSState (codec.h:13-45), compress/decompress (codec.cpp:123-152),
 * CompressedIDInvertedListsFenwickTree ctor/get_ids (custom_invlists_impl.cpp:133-223),
 * ROCNSGGraph ctor/get_neighbors (altid_impl.cpp:103-165).
 * Bitstreams (head + 32-bit stack words) are bit-identical to codec.cpp. */
typedef struct vidc_roc vidc_roc;

/* precision_mode for vidc_roc_encode */
#define VIDC_PREC_REFERENCE (-1) /* ceil(log2((int)max_id)), custom_invlists_impl.cpp:163-164 (pow-2 quirk kept) */
#define VIDC_PREC_EXACT (-2)     /* bit_width(max_id): lossless for pow-2 max ids (NOT reference-identical) */
/* precision_mode >= 0 : fixed precision for every list (compress(n,data,state,precision), codec.cpp:123) */

#define VIDC_ROC_WANT_PERM 1u /* keep the sampling permutation (code re-ordering, custom_invlists_impl.cpp:188-193) */

/* Encode every list.  d_ids: device uint64[ntotal].  offsets: host uint64[nlist+1].
 * Lists may be unsorted and may be empty.  Domain: ids < 2^31 (reference `int max_id`), n <= VIDC_ROC_MAX_LIST. */
#define VIDC_ROC_MAX_LIST 262144u
int vidc_roc_encode(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids,
                    int precision_mode, uint32_t flags, vidc_roc **out);
/* Graph rows: d_rows device int32[N*K], -1 terminated rows (altid_impl.cpp:110-117). */
int vidc_roc_encode_rows(vidc_ctx *ctx, uint64_t N, uint32_t K, const int32_t *d_rows, int precision_mode,
                         uint32_t flags, vidc_roc **out);
void vidc_roc_destroy(vidc_roc *r);

uint64_t vidc_roc_nlist(const vidc_roc *r);
uint64_t vidc_roc_ntotal(const vidc_roc *r);
/* sum over non-empty lists of 8 + 4*nwords  (ANSState::size(), codec.h:42-44; :196-206) */
uint64_t vidc_roc_compressed_bytes(const vidc_roc *r);
uint64_t vidc_roc_total_words(const vidc_roc *r);
/* host copies of per-list metadata (arrays of nlist): any pointer may be NULL.  The metadata lives on the device;
 * the first call copies it to the host (blocking). */
int vidc_roc_list_info(const vidc_roc *r, uint32_t *sizes, uint32_t *precisions, uint64_t *heads,
                       uint32_t *nwords, uint32_t *mt_draws);
/* stack words of one list, push order (parity export).  cap in words. */
int vidc_roc_export_words(vidc_ctx *ctx, const vidc_roc *r, uint64_t list_no, uint32_t *words, size_t cap);
/* the whole compact stream (lists back to back, vidc_roc_total_words() words) in one copy */
int vidc_roc_export_all_words(vidc_ctx *ctx, const vidc_roc *r, uint32_t *words, size_t cap);
/* sampling permutation for all lists: perm[offsets[l]+i] = input position of the i-th sampled id */
int vidc_roc_perm(vidc_ctx *ctx, const vidc_roc *r, uint32_t *perm_host);
const uint32_t *vidc_roc_perm_dev(const vidc_roc *r);
/* Build from exported streams (decode-only parity tests, on-disk reload).  All host arrays. */
int vidc_roc_import(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint32_t *precisions,
                    const uint64_t *heads, const uint32_t *nwords, const uint32_t *mt_draws,
                    const uint32_t *words_concat, vidc_roc **out);

/* Decode every list into d_out (device uint64[ntotal], CSR order, each list in sampling order
 * = the order get_ids returns, codec.cpp:150). */
int vidc_roc_decode_all(vidc_ctx *ctx, const vidc_roc *r, uint64_t *d_out);
/* Decode m selected lists back to back into d_out; out_offsets (host, m+1) receives the packing. */
int vidc_roc_decode_lists(vidc_ctx *ctx, const vidc_roc *r, uint64_t m, const uint64_t *list_nos,
                          uint64_t *d_out, uint64_t *out_offsets);
/* The decode section of the deferred search in ONE call (custom_invlists_impl.cpp:508-525: `ids = get_ids(list_no)` per touched
 * list, then `labels[r] = ids[lo_offset(labels[r])]`): the m touched lists are decoded into device staging owned by the
 * context, the n_items requested ids are picked ON THE DEVICE (item i = list_nos[item_slot[i]][item_off[i]]) and only those
 * cross PCIe: 8 * n_items bytes into ids_out (host int64[n_items]).  Items outside their list -> VIDC_ERR_INVALID, reported before
 * anything is decoded.  list_nos should name every touched list ONCE (a repeated list is decoded and staged once per mention).
 * Same signature for the four containers (vidc_ef_decode_gather, vidc_packed_decode_gather, vidc_wt_decode_gather). */
int vidc_roc_decode_gather(vidc_ctx *ctx, const vidc_roc *r, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                           const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out);
/* Graph flavour: d_out device int32[m*K]; rows padded with -1; counts (host, m, may be NULL) = num edges.
 * nodes == NULL selects nodes 0..m-1 (no index array is built or uploaded).  The first such call for every node of a graph of
 * 65 536 nodes or more also sorts the node numbers by edge count inside the object (4 bytes per node of device memory, 0.02 ms
 * at 10^6 nodes): rows of equal length then share a wavefront of the decoder; the output is the same rows in the same places. */
int vidc_roc_decode_rows(vidc_ctx *ctx, const vidc_roc *r, uint64_t m, const uint64_t *nodes, uint32_t K,
                         int32_t *d_out, uint32_t *counts);
/* End-state self check of the last decode_all: number of lists whose final ANS state is not the
 * initial one (head 2^31; SURVEY appendix A invariant).  Lossy reference cases (Q2/Q3) count here. */
uint64_t vidc_roc_last_decode_nonclean(const vidc_roc *r);

/* -------------------------------------------------------------- packed bits */
/* Replaces CompressedIDInvertedListsPackedBits (custom_invlists_impl.cpp:64-118) and
 * CompactBitNSGGraph (altid_impl.cpp:20-51).  LSB-first, little-endian (reader :35-58). */
typedef struct vidc_packed vidc_packed;
int vidc_packed_bits_for(uint64_t ntotal); /* :68-70 */
int vidc_packed_encode(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids, int bits,
                       vidc_packed **out);
void vidc_packed_destroy(vidc_packed *p);
uint64_t vidc_packed_compressed_bytes(const vidc_packed *p); /* sum ceil(ls*bits/8), :80,85 */
int vidc_packed_bits(const vidc_packed *p);
int vidc_packed_decode_all(vidc_ctx *ctx, const vidc_packed *p, uint64_t *d_out);
/* decode m selected lists back to back into d_out (get_ids per touched list, :96-105 / the loop :508-525);
 * out_offsets: host uint64[m + 1] */
int vidc_packed_decode_lists(vidc_ctx *ctx, const vidc_packed *p, uint64_t m, const uint64_t *list_nos, uint64_t *d_out,
                             uint64_t *out_offsets);
/* decode of the touched lists + device-side pick of the n_items requested ids (see vidc_roc_decode_gather) */
int vidc_packed_decode_gather(vidc_ctx *ctx, const vidc_packed *p, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                              const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out);
/* m random accesses (get_single_id, :108-113): host arrays of list numbers / offsets -> host ids */
int vidc_packed_get(vidc_ctx *ctx, const vidc_packed *p, uint64_t m, const uint64_t *list_nos,
                    const uint64_t *offs, int64_t *ids_out);
/* byte image of one list (parity against the reference layout) */
int vidc_packed_export(vidc_ctx *ctx, const vidc_packed *p, uint64_t list_no, uint8_t *bytes, size_t cap);
/* Flat image {offsets, bits, words} for saving / shipping an object without re-encoding (the reference keeps compressed
 * lists in memory only): words = the device layout, every list starts on a 64-bit word and is followed by one padding
 * word. */
uint64_t vidc_packed_total_words(const vidc_packed *p);
int vidc_packed_export_all(vidc_ctx *ctx, const vidc_packed *p, uint64_t *words, size_t cap);
int vidc_packed_import(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, int bits, const uint64_t *words,
                       uint64_t nwords, vidc_packed **out);

/* CompactBitNSGGraph (altid_impl.cpp:20-51): N rows of K int32 (-1 terminated) -> N * stride bytes, sentinel N. */
typedef struct vidc_compact vidc_compact;
int vidc_compact_rows_encode(vidc_ctx *ctx, uint64_t N, uint32_t K, const int32_t *d_rows, vidc_compact **out);
void vidc_compact_destroy(vidc_compact *c);
uint32_t vidc_compact_bits(const vidc_compact *c);
uint32_t vidc_compact_stride(const vidc_compact *c);
uint64_t vidc_compact_size_in_bytes(const vidc_compact *c);
/* get_neighbors for m nodes: d_out device int32[m*K] (-1 padded), counts host uint32[m] (may be NULL);
 * nodes == NULL selects nodes 0..m-1 */
int vidc_compact_rows_decode(vidc_ctx *ctx, const vidc_compact *c, uint64_t m, const uint64_t *nodes, int32_t *d_out,
                             uint32_t *counts);
int vidc_compact_export_row(vidc_ctx *ctx, const vidc_compact *c, uint64_t node, uint8_t *bytes, size_t cap);

/* -------------------------------------------------------------- Elias-Fano */
/* Replaces CompressedIDInvertedListsEliasFano (custom_invlists_impl.cpp:229-339),
 * EliasFanoNSGGraph (altid_impl.cpp:53-101), succinct::elias_fano builder/select/enumerator
 * (elias_fano.hpp:22-57,141-145,210-261). */
typedef struct vidc_ef vidc_ef;
#define VIDC_EF_WANT_PERM 1u
int vidc_ef_encode(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids, uint32_t flags,
                   vidc_ef **out);
void vidc_ef_destroy(vidc_ef *e);
/* (sum low bits + sum high bits) / 8, custom_invlists_impl.cpp:272-282 */
uint64_t vidc_ef_compressed_bytes(const vidc_ef *e);
int vidc_ef_list_info(const vidc_ef *e, uint32_t *sizes, uint32_t *low_bits, uint64_t *universes);
int vidc_ef_decode_all(vidc_ctx *ctx, const vidc_ef *e, uint64_t *d_out); /* ascending per list, :305-308 */
int vidc_ef_get(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *list_nos, const uint64_t *offs,
                int64_t *ids_out); /* ef->select(offset), :314-318 */
int vidc_ef_perm(vidc_ctx *ctx, const vidc_ef *e, uint32_t *perm_host); /* sort permutation, :324-339 */
/* EliasFanoNSGGraph (altid_impl.cpp:53-101): rows are counted (-1 terminated), sorted and coded per node. */
int vidc_ef_encode_rows(vidc_ctx *ctx, uint64_t N, uint32_t K, const int32_t *d_rows, vidc_ef **out);
/* get_neighbors for m nodes: d_out device int32[m*K] ascending, -1 padded; counts host uint32[m] (may be NULL);
 * nodes == NULL selects nodes 0..m-1 */
int vidc_ef_decode_rows(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *nodes, uint32_t K, int32_t *d_out,
                        uint32_t *counts);
/* decode m selected lists back to back (get_ids per touched list, custom_invlists_impl.cpp:508-525) */
int vidc_ef_decode_lists(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *list_nos, uint64_t *d_out,
                         uint64_t *out_offsets);
/* decode of the touched lists + device-side pick of the n_items requested ids (see vidc_roc_decode_gather) */
int vidc_ef_decode_gather(vidc_ctx *ctx, const vidc_ef *e, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                          const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out);
/* word images of one list's low / high streams (64-bit words, LSB-first) */
int vidc_ef_export(vidc_ctx *ctx, const vidc_ef *e, uint64_t list_no, uint64_t *low, size_t low_cap,
                   uint64_t *high, size_t high_cap, uint64_t *low_nbits, uint64_t *high_nbits);
/* Flat image {offsets, l[], universe[], low[], high[]}: stream offsets follow from (count, l, universe) per list
 * (elias_fano.hpp:28-29); the select directory is rebuilt from the high stream on import. */
int vidc_ef_stream_words(const vidc_ef *e, uint64_t *low_words, uint64_t *high_words);
int vidc_ef_export_all(vidc_ctx *ctx, const vidc_ef *e, uint64_t *low, size_t low_cap, uint64_t *high, size_t high_cap);
int vidc_ef_import(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint32_t *lbits,
                   const uint64_t *universe, const uint64_t *low, uint64_t n_low, const uint64_t *high, uint64_t n_high,
                   vidc_ef **out);

/* ------------------------------------------------------------ wavelet tree */
/* Replaces CompressedIDInvertedListsWaveletTree (custom_invlists_impl.cpp:346-397): one tree over the
 * sequence list_nos[id]; get_single_id(list, offset) = select(offset + 1, list) (:377-379).
 * Requires what the reference asserts (:359-360): ids ascending inside every list and ids a permutation
 * of 0..ntotal-1.  wt_type 0 = plain bitvectors, 1 = sizes reported for rrr_vector<63>-coded levels
 * (custom_invlists_impl.h:104-113); sdsl itself is absent, so sizes follow the documented layouts. */
typedef struct vidc_wt vidc_wt;
int vidc_wt_build(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids, int wt_type,
                  vidc_wt **out);
void vidc_wt_destroy(vidc_wt *w);
uint64_t vidc_wt_size_in_bytes(const vidc_wt *w);
uint32_t vidc_wt_levels(const vidc_wt *w);
int vidc_wt_select(vidc_ctx *ctx, const vidc_wt *w, uint64_t m, const uint64_t *list_nos, const uint64_t *offs,
                   int64_t *ids_out);
int vidc_wt_decode_all(vidc_ctx *ctx, const vidc_wt *w, uint64_t *d_out);
/* get_ids of m selected lists back to back (custom_invlists_impl.cpp:381-392 per list); out_offsets: host uint64[m + 1] */
int vidc_wt_decode_lists(vidc_ctx *ctx, const vidc_wt *w, uint64_t m, const uint64_t *list_nos, uint64_t *d_out,
                         uint64_t *out_offsets);
/* decode of the touched lists + device-side pick of the n_items requested ids (see vidc_roc_decode_gather) */
int vidc_wt_decode_gather(vidc_ctx *ctx, const vidc_wt *w, uint64_t m, const uint64_t *list_nos, uint64_t n_items,
                          const uint64_t *item_slot, const uint64_t *item_off, int64_t *ids_out);

/* ------------------------------------------------------ introspection / timing */
/* Milliseconds spent inside the kernels of the most recent encode / decode call on this context,
 * measured with hipEvents on the context's stream (used by bench.py for the roofline figure). */
double vidc_ctx_last_kernel_ms(const vidc_ctx *ctx);
/* Per-phase kernel time (ms, hipEvents) of the most recent call that ran the phase. */
#define VIDC_PHASE_ROC_ENCODE 0  /* k_roc_encode_tiny + k_roc_encode_gen launches */
#define VIDC_PHASE_ROC_COMPACT 1 /* k_roc_compact */
#define VIDC_PHASE_ROC_DECODE 2  /* k_roc_decode_gen + k_roc_decode_tiny launches */
#define VIDC_PHASE_ROC_ENCODE_CHAIN 3 /* the ONE launch holding the call's longest chains (k_roc_encode_u2<20/18>): its duration */
#define VIDC_PHASE_ROC_DECODE_CHAIN 4 /* k_roc_decode_u2<20/18>, same */
#define VIDC_PHASE_COUNT 8
double vidc_ctx_phase_ms(const vidc_ctx *ctx, int phase);
/* What the chain launch of the last ROC encode (which = 0) / decode (1) on this context processed: ids, lists, longest list and
 * the universe (18 or 20 bits; 0 when the call had no such class).  bench.py derives the dominant kernel's roofline from it. */
int vidc_ctx_chain_info(const vidc_ctx *ctx, int which, uint64_t *ids, uint64_t *lists, uint64_t *longest, uint32_t *universe_bits);

#ifdef __cplusplus
}
#endif
#endif
