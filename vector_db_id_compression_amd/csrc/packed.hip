// packed.hip -- fixed-width packed IDs (CompressedIDInvertedListsPackedBits,
// custom_invlists_impl.cpp:64-118; CompactBitNSGGraph, altid_impl.cpp:20-51).
//
// Layout per list: ceil(ls*bits/8) bytes, value i at bit i*bits, LSB-first inside a byte and
// little-endian across bytes (what BitstringReader_get_bits decodes, custom_invlists_impl.cpp:35-58).
// On the device each list starts on an 8-byte boundary so that every 64-bit word has exactly one
// owner thread (no atomics); the reported size follows the reference formula, not the padding.
// HBM-bound: 8 B read + bits/8 B written per id (encode), the mirror for decode.
#include <algorithm>
#include <memory>

#include "bits.h"
#include "common.h"

using namespace vidc;
using namespace vidc::dev;

struct vidc_packed {
    int device = 0;
    uint64_t nlist = 0, ntotal = 0;
    int bits = 0;
    uint64_t compressed_bytes = 0, total_words = 0;
    std::vector<uint64_t> offsets, word_off;
    DevBuf<uint64_t> d_offsets, d_word_off, d_words;
};

namespace {

// one thread per 64-bit output word
__global__ void k_packed_encode(const uint64_t *ids, const uint64_t *offsets, const uint64_t *word_off,
                                uint32_t nlist, uint64_t total_words, uint32_t bits, uint64_t id_limit,
                                uint64_t *words, uint32_t *err) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total_words; w += stride) {
        const uint32_t l = find_list(word_off, nlist, w);
        const uint64_t n = offsets[l + 1] - offsets[l];
        const uint64_t *src = ids + offsets[l];
        const uint64_t keep = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
        const uint64_t out = gather_word<true>(src, n, w - word_off[l], bits, keep, id_limit, err);
        words[w] = out;
    }
}

// one thread per id
__global__ void k_packed_decode(const uint64_t *words, const uint64_t *offsets, const uint64_t *word_off,
                                uint32_t nlist, uint64_t ntotal, uint32_t bits, uint64_t *out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < ntotal; g += stride) {
        const uint32_t l = find_list(offsets, nlist, g);
        out[g] = read_bits(words + word_off[l], (g - offsets[l]) * bits, bits);
    }
}

__global__ void k_packed_get(const uint64_t *words, const uint64_t *word_off, uint32_t bits, uint64_t m,
                             const uint64_t *list_nos, const uint64_t *offs, int64_t *out) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < m) out[q] = (int64_t)read_bits(words + word_off[list_nos[q]], offs[q] * bits, bits);
}

}  // namespace

extern "C" {

int vidc_packed_bits_for(uint64_t ntotal) {  // custom_invlists_impl.cpp:68-70
    int bits = 0;
    while (bits < 64 && (1ull << bits) < ntotal + 1) bits++;
    return bits;
}

int vidc_packed_encode(vidc_ctx *ctx, uint64_t nlist, const uint64_t *offsets, const uint64_t *d_ids, int bits,
                       vidc_packed **out) {
    if (!ctx || !out || (nlist && !offsets) || bits < 0 || bits > 64) return VIDC_ERR_INVALID;
    *out = nullptr;
    if (nlist >= 0xffffffffull) return VIDC_ERR_INVALID;
    VIDC_HIP(hipSetDevice(ctx->device));
    std::unique_ptr<vidc_packed> p(new vidc_packed());
    p->device = ctx->device;
    p->nlist = nlist;
    p->bits = bits;
    p->offsets.assign(nlist + 1, 0);
    if (nlist) p->offsets.assign(offsets, offsets + nlist + 1);
    p->ntotal = p->offsets[nlist];
    p->word_off.assign(nlist + 1, 0);
    for (uint64_t l = 0; l < nlist; l++) {
        if (p->offsets[l + 1] < p->offsets[l]) { set_error("offsets not monotone"); return VIDC_ERR_INVALID; }
        uint64_t n = p->offsets[l + 1] - p->offsets[l];
        p->compressed_bytes += (n * bits + 7) / 8;                       // ids_all[list_no].resize((ls*bits+7)/8), :80
        p->word_off[l + 1] = p->word_off[l] + (n * bits + 63) / 64 + 1;  // +1: read_bits may touch the next word
    }
    p->total_words = p->word_off[nlist];
    if (p->ntotal && !d_ids) return VIDC_ERR_INVALID;
    VIDC_TRY(p->d_offsets.alloc(nlist + 1));
    VIDC_TRY(p->d_word_off.alloc(nlist + 1));
    VIDC_TRY(p->d_words.alloc(p->total_words ? p->total_words : 1));
    VIDC_HIP(hipMemcpyAsync(p->d_offsets.p, p->offsets.data(), (nlist + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(p->d_word_off.p, p->word_off.data(), (nlist + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    Scratch s_err;
    VIDC_TRY(s_err.get(ctx, 4));
    VIDC_HIP(hipMemsetAsync(s_err.p, 0, 4, ctx->stream));
    if (p->total_words) {
        VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
        uint64_t blocks = (p->total_words + 255) / 256;
        uint32_t grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)ctx->num_cu * 32);
        // ids must fit the field (FAISS_THROW_IF_NOT(ids_in[i] >= 0 && ids_in[i] < ntotal), :87)
        uint64_t limit = ~0ull;
        hipLaunchKernelGGL(k_packed_encode, dim3(grid), dim3(256), 0, ctx->stream, d_ids, p->d_offsets.p,
                           p->d_word_off.p, (uint32_t)nlist, p->total_words, (uint32_t)bits, limit, p->d_words.p,
                           s_err.as<uint32_t>());
        VIDC_HIP(hipGetLastError());
        VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    }
    uint32_t err = 0;
    VIDC_HIP(hipMemcpyAsync(&err, s_err.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    if (p->total_words) {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        ctx->last_kernel_ms = ms;
    }
    if (err) {
        set_error("packed bits: an id does not fit %d bits (reference: FAISS_THROW_IF_NOT(ids_in[i] >= 0 && "
                  "ids_in[i] < ntotal), custom_invlists_impl.cpp:87)", bits);
        return VIDC_ERR_DOMAIN;
    }
    *out = p.release();
    return VIDC_OK;
}

void vidc_packed_destroy(vidc_packed *p) { delete p; }
uint64_t vidc_packed_compressed_bytes(const vidc_packed *p) { return p ? p->compressed_bytes : 0; }
int vidc_packed_bits(const vidc_packed *p) { return p ? p->bits : 0; }

int vidc_packed_decode_all(vidc_ctx *ctx, const vidc_packed *p, uint64_t *d_out) {
    if (!ctx || !p || (p->ntotal && !d_out)) return VIDC_ERR_INVALID;
    if (!p->ntotal) return VIDC_OK;
    VIDC_HIP(hipSetDevice(ctx->device));
    VIDC_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    uint64_t blocks = (p->ntotal + 255) / 256;
    uint32_t grid = (uint32_t)std::min<uint64_t>(blocks, (uint64_t)ctx->num_cu * 32);
    hipLaunchKernelGGL(k_packed_decode, dim3(grid), dim3(256), 0, ctx->stream, p->d_words.p, p->d_offsets.p,
                       p->d_word_off.p, (uint32_t)p->nlist, p->ntotal, (uint32_t)p->bits, d_out);
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->last_kernel_ms = ms;
    return VIDC_OK;
}

int vidc_packed_get(vidc_ctx *ctx, const vidc_packed *p, uint64_t m, const uint64_t *list_nos, const uint64_t *offs,
                    int64_t *ids_out) {
    if (!ctx || !p || (m && (!list_nos || !offs || !ids_out))) return VIDC_ERR_INVALID;
    if (!m) return VIDC_OK;
    for (uint64_t i = 0; i < m; i++) {
        if (list_nos[i] >= p->nlist || offs[i] >= p->offsets[list_nos[i] + 1] - p->offsets[list_nos[i]]) {
            set_error("packed get: (list %llu, offset %llu) out of range", (unsigned long long)list_nos[i],
                      (unsigned long long)offs[i]);
            return VIDC_ERR_INVALID;
        }
    }
    VIDC_HIP(hipSetDevice(ctx->device));
    Scratch s_l, s_o, s_r;
    VIDC_TRY(s_l.get(ctx, m * 8)); VIDC_TRY(s_o.get(ctx, m * 8)); VIDC_TRY(s_r.get(ctx, m * 8));
    VIDC_HIP(hipMemcpyAsync(s_l.p, list_nos, m * 8, hipMemcpyHostToDevice, ctx->stream));
    VIDC_HIP(hipMemcpyAsync(s_o.p, offs, m * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_packed_get, dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, ctx->stream, p->d_words.p,
                       p->d_word_off.p, (uint32_t)p->bits, m, s_l.as<uint64_t>(), s_o.as<uint64_t>(),
                       s_r.as<int64_t>());
    VIDC_HIP(hipGetLastError());
    VIDC_HIP(hipMemcpyAsync(ids_out, s_r.p, m * 8, hipMemcpyDeviceToHost, ctx->stream));
    VIDC_HIP(hipStreamSynchronize(ctx->stream));
    return VIDC_OK;
}

int vidc_packed_export(vidc_ctx *ctx, const vidc_packed *p, uint64_t list_no, uint8_t *bytes, size_t cap) {
    if (!ctx || !p || list_no >= p->nlist) return VIDC_ERR_INVALID;
    uint64_t n = p->offsets[list_no + 1] - p->offsets[list_no];
    uint64_t nb = (n * p->bits + 7) / 8;
    if (nb > cap) { set_error("export buffer too small"); return VIDC_ERR_INVALID; }
    return vidc_copy_d2h(ctx, bytes, p->d_words.p + p->word_off[list_no], nb);
}

}  // extern "C"
