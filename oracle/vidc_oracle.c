/*
 * vidc_oracle.c -- CPU restatement of the reference's per-list ID codecs (plain C11).
 *
 * TEST INFRASTRUCTURE ONLY (see vidc_oracle.h).  Written from the behavioural
 * description of the reference; every function cites the reference lines it follows.
 * The order-statistic structures are deliberately NOT the reference's pointer BST
 * (fenwick_tree.h:20-167): the bitstream does not depend on the tree shape, only on
 * rank/select over the (multi)set, so encode uses a sorted array + binary indexed
 * tree and decode an array-backed counting BST.
 */
#include "vidc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define VO_L ((uint64_t)1 << 31) /* rans_l, codec.cpp:19 */

/* ------------------------------------------------------------------ mt19937 */
/* Standard MT19937 (Matsumoto & Nishimura); the reference seeds std::mt19937 with 1234
 * (codec.h:18) and draws one 32-bit word per stack underflow (codec.h:32-40). */
void vo_mt19937_table(uint32_t seed, uint32_t *out, size_t count) {
    uint32_t mt[624];
    mt[0] = seed;
    for (int i = 1; i < 624; i++)
        mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    int idx = 624;
    for (size_t c = 0; c < count; c++) {
        if (idx >= 624) {
            for (int k = 0; k < 624; k++) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                uint32_t v = mt[(k + 397) % 624] ^ (y >> 1);
                if (y & 1u) v ^= 0x9908b0dfu;
                mt[k] = v;
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        out[c] = y;
    }
}

/* --------------------------------------------------------------- precision */
/* custom_invlists_impl.cpp:163-164: int max_id; (uint64_t)ceil(log2(max_id)), later
 * narrowed to the int `precision` argument of codec_push/codec_pop (codec.h:47-52).
 * max_id == 0 gives log2(0) = -inf; on x86-64 the double->uint64 conversion yields
 * 0x8000000000000000 whose low 32 bits are 0 => precision 0.
 * max_id < 0 is outside the reference's domain (ids must be < 2^31, Q4). */
int vo_precision_from_max_id(int32_t max_id) {
    if (max_id <= 0) return 0;
    double v = ceil(log2((double)max_id));
    return (int)(uint64_t)v;
}

/* --------------------------------------------------------------- ANS state */
void vo_ans_init(vo_ans_state *st) {
    st->head = VO_L; /* codec.h:14 */
    st->stack = NULL;
    st->nstack = 0;
    st->cap = 0;
    st->mt_draws = 0;
}
void vo_ans_free(vo_ans_state *st) {
    free(st->stack);
    st->stack = NULL;
    st->nstack = st->cap = 0;
}
void vo_ans_copy(vo_ans_state *dst, const vo_ans_state *src) {
    dst->head = src->head;
    dst->nstack = src->nstack;
    dst->cap = src->nstack + 16;
    dst->mt_draws = src->mt_draws;
    dst->stack = (uint32_t *)malloc(dst->cap * sizeof(uint32_t));
    if (src->nstack) memcpy(dst->stack, src->stack, src->nstack * sizeof(uint32_t));
}
static void pushw(vo_ans_state *st, uint64_t w) { /* codec.h:20-22 */
    if (st->nstack == st->cap) {
        st->cap = st->cap ? st->cap * 2 : 64;
        st->stack = (uint32_t *)realloc(st->stack, st->cap * sizeof(uint32_t));
    }
    st->stack[st->nstack++] = (uint32_t)(w & 0xffffffffu);
}
static uint64_t popw(vo_ans_state *st) { /* codec.h:32-40 */
    if (st->nstack) return st->stack[--st->nstack];
    uint32_t tab[64];
    /* draws are rare (0 or 1 per list): regenerate the prefix of the sequence */
    size_t need = (size_t)st->mt_draws + 1;
    uint32_t *t = need <= 64 ? tab : (uint32_t *)malloc(need * sizeof(uint32_t));
    vo_mt19937_table(1234u, t, need);
    uint32_t w = t[st->mt_draws];
    if (t != tab) free(t);
    st->mt_draws++;
    return w;
}

/* codec.cpp:65-76 (uniform 2^p push; '+' not '|': carries if start >= 2^p) */
static void u_push(vo_ans_state *st, uint64_t start, int p) {
    uint64_t head = st->head;
    if (head >= ((VO_L >> p) << 32)) {
        pushw(st, head);
        head >>= 32;
    }
    st->head = (head << p) + start;
}
/* codec.cpp:78-90 */
static uint64_t u_pop(vo_ans_state *st, int p) {
    uint64_t h0 = st->head;
    uint64_t s = h0 & (((uint64_t)1 << p) - 1);
    uint64_t head = h0 >> p;
    if (head < VO_L) head = (head << 32) | popw(st);
    st->head = head;
    return s;
}
static int slice_p(int precision, int lower) {
    int p = precision - lower;
    if (p < 0) p = 0;
    if (p > 16) p = 16;
    return p;
}
/* codec.cpp:92-105: four 16-bit slices low->high, all four calls execute */
void vo_id_push(vo_ans_state *st, uint64_t sym, int precision) {
    for (int lower = 0; lower < 64; lower += 16)
        u_push(st, (sym >> lower) & 0xffff, slice_p(precision, lower));
}
/* codec.cpp:107-121: slices high->low */
uint64_t vo_id_pop(vo_ans_state *st, int precision) {
    uint64_t sym = 0;
    for (int lower = 48; lower >= 0; lower -= 16)
        sym = (sym << 16) | u_pop(st, slice_p(precision, lower));
    return sym;
}
/* codec.cpp:21-42 (note: the refill test is on h0, not on the new head) */
uint64_t vo_idx_pop(vo_ans_state *st, uint64_t nmax) {
    uint64_t h0 = st->head;
    if (h0 >= nmax * ((VO_L / nmax) << 32)) {
        pushw(st, h0);
        h0 >>= 32;
    }
    uint64_t k = h0 % nmax;
    uint64_t head = h0 / nmax;
    if (h0 < VO_L) head = popw(st) | (head << 32);
    st->head = head;
    return k;
}
/* codec.cpp:44-63 */
void vo_idx_push(vo_ans_state *st, uint64_t sym, uint64_t nmax) {
    uint64_t h0 = st->head;
    if (h0 >= ((VO_L / nmax) << 32)) {
        pushw(st, h0);
        h0 >>= 32;
    }
    uint64_t head = h0 * nmax + sym;
    if (head < VO_L) head = popw(st) | (head << 32);
    st->head = head;
}

/* ------------------------------------------------------------- ROC encode */
typedef struct {
    uint64_t id;
    uint32_t pos;
} vo_pair;
static int cmp_pair(const void *a, const void *b) {
    const vo_pair *x = (const vo_pair *)a, *y = (const vo_pair *)b;
    if (x->id != y->id) return x->id < y->id ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

/* Sampling without replacement driven by the ANS state:
 * custom_invlists_impl.cpp:178-192 (container) == codec.cpp:131-137 (compress()).
 * The set order is (id, position): the reference orders tuple<id, code pointer>
 * (custom_invlists_impl.cpp:139,173-175), pointers increase with the position. */
void vo_roc_encode(size_t n, const uint64_t *ids, int precision, vo_ans_state *st,
                   uint64_t *order_out, uint32_t *perm_out) {
    if (n == 0) return;
    vo_pair *s = (vo_pair *)malloc(n * sizeof(vo_pair));
    for (size_t i = 0; i < n; i++) {
        s[i].id = ids[i];
        s[i].pos = (uint32_t)i;
    }
    qsort(s, n, sizeof(vo_pair), cmp_pair);
    /* binary indexed tree over the alive flags of the sorted positions (1-based) */
    size_t lg = 1;
    while (((size_t)1 << lg) <= n) lg++;
    uint32_t *bit = (uint32_t *)calloc(n + 1, sizeof(uint32_t));
    for (size_t i = 1; i <= n; i++) {
        bit[i] += 1;
        size_t j = i + (i & (~i + 1));
        if (j <= n) bit[j] += bit[i];
    }
    for (size_t i = 0; i < n; i++) {
        uint64_t nmax = n - i;
        uint64_t k = vo_idx_pop(st, nmax);
        /* select: smallest position with (k+1) alive elements at or before it */
        size_t pos = 0;
        uint64_t rem = k;
        for (size_t step = (size_t)1 << (lg - 1); step; step >>= 1) {
            size_t nx = pos + step;
            if (nx <= n && bit[nx] <= rem) {
                pos = nx;
                rem -= bit[nx];
            }
        }
        /* pos is 0-based index of the selected element; remove it */
        for (size_t j = pos + 1; j <= n; j += j & (~j + 1)) bit[j] -= 1;
        vo_id_push(st, s[pos].id, precision);
        if (order_out) order_out[i] = s[pos].id;
        if (perm_out) perm_out[i] = s[pos].pos;
    }
    free(bit);
    free(s);
}

/* ------------------------------------------------------------- ROC decode */
typedef struct {
    uint64_t key;
    int32_t left, right;
    uint32_t size; /* elements (with multiplicity) in this subtree */
    uint32_t cnt;  /* multiplicity of key */
} vo_node;

/* codec.cpp:140-152: pop id, rank among already decoded (strictly smaller,
 * fenwick_tree.h:42-94 returns start = #smaller), push rank with nmax = i+1,
 * write back-to-front. */
void vo_roc_decode(vo_ans_state *st, size_t n, int precision, uint64_t *out) {
    if (n == 0) return;
    vo_node *t = (vo_node *)malloc(n * sizeof(vo_node));
    size_t nn = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t x = vo_id_pop(st, precision);
        uint64_t rank = 0;
        if (nn == 0) {
            t[0].key = x; t[0].left = t[0].right = -1; t[0].size = 1; t[0].cnt = 1;
            nn = 1;
        } else {
            int32_t cur = 0;
            for (;;) {
                vo_node *c = &t[cur];
                c->size += 1;
                uint32_t lsz = c->left >= 0 ? t[c->left].size : 0;
                if (x < c->key) {
                    if (c->left < 0) {
                        c->left = (int32_t)nn;
                        t[nn].key = x; t[nn].left = t[nn].right = -1; t[nn].size = 1; t[nn].cnt = 1;
                        nn++;
                        break;
                    }
                    cur = c->left;
                } else if (x > c->key) {
                    rank += lsz + c->cnt;
                    if (c->right < 0) {
                        c->right = (int32_t)nn;
                        t[nn].key = x; t[nn].left = t[nn].right = -1; t[nn].size = 1; t[nn].cnt = 1;
                        nn++;
                        break;
                    }
                    cur = c->right;
                } else {
                    rank += lsz;
                    c->cnt += 1;
                    break;
                }
            }
        }
        vo_idx_push(st, rank, (uint64_t)i + 1);
        out[n - 1 - i] = x;
    }
    free(t);
}

/* ------------------------------------------------------------ packed bits */
/* custom_invlists_impl.cpp:68-70 / altid_impl.cpp:22-23: smallest b with 2^b >= ntotal+1
 * (the reference evaluates 1<<bits as int; identical for ntotal < 2^30) */
int vo_packed_bits_for(uint64_t ntotal) {
    int bits = 0;
    while (((uint64_t)1 << bits) < ntotal + 1) bits++;
    return bits;
}
/* LSB-first within a byte, little-endian across bytes: the layout the in-tree
 * random-access reader (custom_invlists_impl.cpp:35-58) decodes. Buffer must be zeroed. */
void vo_packed_write(uint8_t *code, size_t bit_offset, uint64_t x, int nbit) {
    for (int b = 0; b < nbit; b++) {
        if ((x >> b) & 1) {
            size_t pos = bit_offset + (size_t)b;
            code[pos >> 3] |= (uint8_t)(1u << (pos & 7));
        }
    }
}
uint64_t vo_packed_read(const uint8_t *code, size_t bit_offset, int nbit) {
    uint64_t r = 0;
    for (int b = 0; b < nbit; b++) {
        size_t pos = bit_offset + (size_t)b;
        r |= (uint64_t)((code[pos >> 3] >> (pos & 7)) & 1u) << b;
    }
    return r;
}

/* -------------------------------------------------------------- Elias-Fano */
static int msb64(uint64_t x) { /* index of the highest set bit, x != 0 */
    int r = 0;
    while (x >>= 1) r++;
    return r;
}
/* elias_fano.hpp:28: m_l = (m && n / m) ? msb(n / m) : 0 */
int vo_ef_low_bits(uint64_t universe, uint64_t m) {
    return (m && universe / m) ? msb64(universe / m) : 0;
}
/* elias_fano.hpp:22-46: low stream = m*l bits appended LSB-first; high stream of
 * (m+1) + (n>>l) + 1 bits with bit (x>>l)+pos set for the pos-th element. */
void vo_ef_build(vo_ef *ef, uint64_t universe, uint64_t m, const uint64_t *sorted_ids) {
    ef->universe = universe;
    ef->m = m;
    ef->l = vo_ef_low_bits(universe, m);
    ef->low_nbits = m * (uint64_t)ef->l;
    ef->high_nbits = (m + 1) + (universe >> ef->l) + 1;
    ef->low = (uint64_t *)calloc((size_t)(ef->low_nbits + 63) / 64 + 1, 8);
    ef->high = (uint64_t *)calloc((size_t)(ef->high_nbits + 63) / 64 + 1, 8);
    uint64_t mask = ef->l ? (((uint64_t)1 << ef->l) - 1) : 0;
    for (uint64_t pos = 0; pos < m; pos++) {
        uint64_t x = sorted_ids[pos];
        if (ef->l) {
            uint64_t lowv = x & mask, bp = pos * (uint64_t)ef->l;
            ef->low[bp >> 6] |= lowv << (bp & 63);
            if ((bp & 63) + (uint64_t)ef->l > 64) ef->low[(bp >> 6) + 1] |= lowv >> (64 - (bp & 63));
        }
        uint64_t hp = (x >> ef->l) + pos;
        ef->high[hp >> 6] |= (uint64_t)1 << (hp & 63);
    }
}
void vo_ef_free(vo_ef *ef) {
    free(ef->low);
    free(ef->high);
    ef->low = ef->high = NULL;
}
static uint64_t ef_low_at(const vo_ef *ef, uint64_t i) {
    if (!ef->l) return 0;
    uint64_t bp = i * (uint64_t)ef->l;
    uint64_t v = ef->low[bp >> 6] >> (bp & 63);
    if ((bp & 63) + (uint64_t)ef->l > 64) v |= ef->low[(bp >> 6) + 1] << (64 - (bp & 63));
    return v & (((uint64_t)1 << ef->l) - 1);
}
/* elias_fano.hpp:141-145: ((select1(high, i) - i) << l) | low[i] */
uint64_t vo_ef_select(const vo_ef *ef, uint64_t i) {
    uint64_t seen = 0;
    for (uint64_t w = 0;; w++) {
        uint64_t word = ef->high[w];
        uint64_t c = (uint64_t)__builtin_popcountll(word);
        if (seen + c > i) {
            for (uint64_t k = i - seen; k; k--) word &= word - 1;
            uint64_t pos = w * 64 + (uint64_t)__builtin_ctzll(word);
            return ((pos - i) << ef->l) | ef_low_at(ef, i);
        }
        seen += c;
    }
}
/* elias_fano.hpp:210-261: enumerate the ones of the high stream in order */
void vo_ef_decode_all(const vo_ef *ef, uint64_t *out) {
    uint64_t i = 0;
    for (uint64_t w = 0; i < ef->m; w++) {
        uint64_t word = ef->high[w];
        while (word && i < ef->m) {
            uint64_t pos = w * 64 + (uint64_t)__builtin_ctzll(word);
            word &= word - 1;
            out[i] = ((pos - i) << ef->l) | ef_low_at(ef, i);
            i++;
        }
    }
}

/* ------------------------------------------------------------ wavelet tree */
/* custom_invlists_impl.cpp:377-379: wt.select(offset+1, list_no) over the sequence
 * list_nos[id] = position (= id) of the (offset+1)-th occurrence of list_no. */
int64_t vo_wt_select(const uint32_t *list_nos, size_t ntotal, uint32_t c, uint64_t k) {
    for (size_t i = 0; i < ntotal; i++)
        if (list_nos[i] == c) {
            if (k == 0) return (int64_t)i;
            k--;
        }
    return -1;
}

/* ------------------------------------------------------------ CSR helpers */
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
int vo_omp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y);
}

/* Mirrors the container flow custom_invlists_impl.cpp:147-192 (encode every list,
 * parallel over lists) and :210-219 (get_ids: copy state, decode). */
size_t vo_roc_bench_lists(size_t nlist, const uint64_t *offsets, const uint64_t *ids, int threads,
                          double *t_enc, double *t_dec, uint64_t *sum_bytes) {
    vo_ans_state *sts = (vo_ans_state *)malloc(nlist * sizeof(vo_ans_state));
    int *prec = (int *)malloc(nlist * sizeof(int));
    /* the container always produces the sampling permutation (custom_invlists_impl.cpp:188-193) */
    uint32_t *perm = (uint32_t *)malloc((size_t)(offsets[nlist] ? offsets[nlist] : 1) * sizeof(uint32_t));
    (void)threads;
    double t0 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(threads)
#endif
    for (size_t l = 0; l < nlist; l++) {
        size_t n = (size_t)(offsets[l + 1] - offsets[l]);
        const uint64_t *p = ids + offsets[l];
        vo_ans_init(&sts[l]);
        prec[l] = 0;
        if (!n) continue;
        uint64_t mx = 0;
        for (size_t i = 0; i < n; i++) mx = p[i] > mx ? p[i] : mx;
        prec[l] = vo_precision_from_max_id((int32_t)mx);
        vo_roc_encode(n, p, prec[l], &sts[l], NULL, perm + offsets[l]);
    }
    double t1 = now_s();
    uint64_t bytes = 0;
    size_t maxn = 0;
    for (size_t l = 0; l < nlist; l++) {
        size_t n = (size_t)(offsets[l + 1] - offsets[l]);
        if (n) bytes += 8 + 4 * (uint64_t)sts[l].nstack;
        if (n > maxn) maxn = n;
    }
    size_t bad = 0;
    uint64_t ntotal = offsets[nlist];
    uint64_t *outs = (uint64_t *)malloc((size_t)(ntotal ? ntotal : 1) * sizeof(uint64_t));
    double t2 = now_s();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic) num_threads(threads)
#endif
    for (size_t l = 0; l < nlist; l++) {
        size_t n = (size_t)(offsets[l + 1] - offsets[l]);
        if (!n) continue;
        vo_ans_state c;
        vo_ans_copy(&c, &sts[l]);
        vo_roc_decode(&c, n, prec[l], outs + offsets[l]);
        vo_ans_free(&c);
    }
    double t3 = now_s();
    /* untimed validation: decoded set == input set per list */
    for (size_t l = 0; l < nlist; l++) {
        size_t n = (size_t)(offsets[l + 1] - offsets[l]);
        if (!n) continue;
        uint64_t *ref = (uint64_t *)malloc(n * sizeof(uint64_t));
        memcpy(ref, ids + offsets[l], n * sizeof(uint64_t));
        qsort(outs + offsets[l], n, 8, cmp_u64);
        qsort(ref, n, 8, cmp_u64);
        if (memcmp(outs + offsets[l], ref, n * 8) != 0) bad++;
        free(ref);
    }
    free(outs);
    for (size_t l = 0; l < nlist; l++) vo_ans_free(&sts[l]);
    free(sts);
    free(prec);
    free(perm);
    *t_enc = t1 - t0;
    *t_dec = t3 - t2;
    *sum_bytes = bytes;
    return bad;
}
