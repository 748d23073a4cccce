"""Device-resident codec objects over the C-ABI (one object = every list of one index / graph).

IDs live in HBM as torch int64 tensors (faiss::idx_t); list boundaries are a host CSR
`offsets[nlist+1]`.  All heavy work happens in the HIP kernels behind libvidc.so.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import VIDC_PREC_EXACT, VIDC_PREC_REFERENCE, check, lib, ptr


def _torch():
    import torch

    return torch


def _as_offsets(offsets):
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    assert off.ndim == 1 and off.size >= 1
    return off


def _dev_ids(ids, ntotal):
    """int64/uint64 CUDA tensor (or numpy array, uploaded) of ntotal ids."""
    torch = _torch()
    if isinstance(ids, np.ndarray):
        ids = torch.from_numpy(np.ascontiguousarray(ids).view(np.int64)).cuda()
    assert ids.is_cuda, "ids must live on the GPU"
    assert ids.dtype in (torch.int64, torch.uint64)
    ids = ids.contiguous()
    assert ids.numel() == ntotal
    return ids



def _decode_gather(fn, obj, list_nos, item_slot, item_off):
    """vidc_*_decode_gather: decode the touched lists on the device, pick the requested ids there, copy 8 bytes per item
    (the decode section of the deferred search, custom_invlists_impl.cpp:508-525) -> numpy int64[n_items]."""
    ln = np.ascontiguousarray(list_nos, dtype=np.uint64)
    sl = np.ascontiguousarray(item_slot, dtype=np.uint64)
    of = np.ascontiguousarray(item_off, dtype=np.uint64)
    assert sl.size == of.size
    out = np.zeros(max(sl.size, 1), np.int64)
    check(fn(obj.ctx.h, obj.h, ln.size, ptr(ln), sl.size, ptr(sl), ptr(of), ptr(out)))
    return out[: sl.size]


class RocLists:
    """ROC-compressed lists (vidc_roc): bit-identical streams to codec.cpp."""

    def __init__(self, handle, ctx, offsets):
        self.h = handle
        self.ctx = ctx
        self._offsets = offsets

    @property
    def offsets(self):
        """CSR offsets of the decoded output.  Graph objects fetch the edge counts from the device on first use
        (the per-node metadata stays on the GPU otherwise)."""
        if self._offsets is None:
            sizes = self.info()["sizes"]
            self._offsets = np.concatenate([[0], np.cumsum(sizes, dtype=np.uint64)]).astype(np.uint64)
        return self._offsets

    def __del__(self):
        try:  # may run during interpreter shutdown, after module globals are gone
            if getattr(self, "h", None):
                lib().vidc_roc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- construction
    @classmethod
    def encode(cls, offsets, ids, precision_mode=VIDC_PREC_REFERENCE, want_perm=False, ctx=None):
        ctx = _lib.default_context() if ctx is None else ctx
        off = _as_offsets(offsets)
        nlist = off.size - 1
        d_ids = _dev_ids(ids, int(off[-1] - off[0])) if off[-1] > off[0] else None
        assert off[0] == 0
        h = C.c_void_p()
        check(lib().vidc_roc_encode(ctx.h, nlist, ptr(off), ptr(d_ids), int(precision_mode),
                                    _lib.VIDC_ROC_WANT_PERM if want_perm else 0, C.byref(h)))
        return cls(h, ctx, off)

    @classmethod
    def encode_rows(cls, rows, precision_mode=VIDC_PREC_REFERENCE, ctx=None):
        """rows: int32 CUDA tensor [N, K], -1 terminated (nsg::Graph<int32_t> layout)."""
        torch = _torch()
        ctx = _lib.default_context() if ctx is None else ctx
        if isinstance(rows, np.ndarray):
            rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).cuda()
        assert rows.is_cuda and rows.dtype == torch.int32 and rows.dim() == 2
        rows = rows.contiguous()
        N, K = rows.shape
        h = C.c_void_p()
        check(lib().vidc_roc_encode_rows(ctx.h, N, K, ptr(rows) if N else None, int(precision_mode), 0, C.byref(h)))
        obj = cls(h, ctx, None)
        obj.K = K
        return obj

    @classmethod
    def from_streams(cls, offsets, precisions, heads, nwords, words_concat, mt_draws=None, ctx=None):
        ctx = _lib.default_context() if ctx is None else ctx
        off = _as_offsets(offsets)
        nlist = off.size - 1
        prec = np.ascontiguousarray(precisions, dtype=np.uint32)
        hd = np.ascontiguousarray(heads, dtype=np.uint64)
        nw = np.ascontiguousarray(nwords, dtype=np.uint32)
        wc = np.ascontiguousarray(words_concat, dtype=np.uint32)
        dr = None if mt_draws is None else np.ascontiguousarray(mt_draws, dtype=np.uint32)
        h = C.c_void_p()
        check(lib().vidc_roc_import(ctx.h, nlist, ptr(off), ptr(prec), ptr(hd), ptr(nw), ptr(dr),
                                    ptr(wc) if wc.size else None, C.byref(h)))
        return cls(h, ctx, off)

    # -- flat on-disk / wire image (the reference keeps compressed lists in memory only)
    def save(self, path):
        """{offsets, precision, heads, nwords, mt_draws, words} as one .npz; `load` rebuilds the device object."""
        info = self.info()
        nw = info["nwords"]
        words = self.all_words()
        np.savez(path, offsets=self.offsets, precision=info["precision"], heads=info["heads"], nwords=nw,
                 mt_draws=info["mt_draws"], words=words)

    @classmethod
    def load(cls, path, ctx=None):
        z = np.load(path)
        return cls.from_streams(z["offsets"], z["precision"], z["heads"], z["nwords"], z["words"], z["mt_draws"], ctx=ctx)

    # -- properties
    @property
    def nlist(self):
        return int(lib().vidc_roc_nlist(self.h))

    @property
    def ntotal(self):
        return int(lib().vidc_roc_ntotal(self.h))

    @property
    def compressed_bytes(self):
        return int(lib().vidc_roc_compressed_bytes(self.h))

    @property
    def total_words(self):
        return int(lib().vidc_roc_total_words(self.h))

    def info(self):
        n = self.nlist
        sizes = np.zeros(n, np.uint32)
        prec = np.zeros(n, np.uint32)
        heads = np.zeros(n, np.uint64)
        nwords = np.zeros(n, np.uint32)
        draws = np.zeros(n, np.uint32)
        check(lib().vidc_roc_list_info(self.h, ptr(sizes), ptr(prec), ptr(heads), ptr(nwords), ptr(draws)))
        return dict(sizes=sizes, precision=prec, heads=heads, nwords=nwords, mt_draws=draws)

    def words(self, list_no, nwords=None):
        if nwords is None:
            nwords = int(self.info()["nwords"][list_no])
        w = np.zeros(max(nwords, 1), np.uint32)
        check(lib().vidc_roc_export_words(self.ctx.h, self.h, list_no, ptr(w), nwords))
        return w[:nwords]

    def all_words(self):
        """the compact stream of every list back to back (word offsets = cumsum of info()['nwords'])"""
        tw = self.total_words
        w = np.zeros(max(tw, 1), np.uint32)
        check(lib().vidc_roc_export_all_words(self.ctx.h, self.h, ptr(w), tw))
        return w[:tw]

    def perm(self):
        p = np.zeros(max(self.ntotal, 1), np.uint32)
        check(lib().vidc_roc_perm(self.ctx.h, self.h, ptr(p)))
        return p[: self.ntotal]

    # -- decode
    def decode_all(self, out=None):
        torch = _torch()
        if out is None:
            out = torch.empty(max(self.ntotal, 1), dtype=torch.int64, device="cuda")
        check(lib().vidc_roc_decode_all(self.ctx.h, self.h, ptr(out)))
        return out[: self.ntotal]

    def decode_lists(self, list_nos):
        torch = _torch()
        ln = np.ascontiguousarray(list_nos, dtype=np.uint64)
        sizes = (self.offsets[1:] - self.offsets[:-1])[ln.astype(np.int64)] if ln.size else np.zeros(0, np.uint64)
        total = int(sizes.sum())
        out = torch.empty(max(total, 1), dtype=torch.int64, device="cuda")
        out_off = np.zeros(ln.size + 1, np.uint64)
        check(lib().vidc_roc_decode_lists(self.ctx.h, self.h, ln.size, ptr(ln), ptr(out), ptr(out_off)))
        return out[:total], out_off

    def decode_gather(self, list_nos, item_slot, item_off):
        """ids[i] = list_nos[item_slot[i]][item_off[i]]: the touched lists decoded and the results picked on the device,
        8 bytes per result copied to the host (vidc_roc_decode_gather)."""
        return _decode_gather(lib().vidc_roc_decode_gather, self, list_nos, item_slot, item_off)

    def decode_rows(self, nodes, K=None, want_counts=True):
        """-> (int32 [m, K] CUDA tensor, -1 padded; edge counts or None).  `want_counts=False` keeps the per-node
        edge counts on the device (no metadata crosses PCIe)."""
        torch = _torch()
        K = K or self.K
        if nodes is None:  # every node, in order: no index array at all
            nd, m = None, self.nlist
        else:
            nd = np.ascontiguousarray(nodes, dtype=np.uint64)
            m = nd.size
        out = torch.empty((max(m, 1), K), dtype=torch.int32, device="cuda")
        counts = np.zeros(max(m, 1), np.uint32) if want_counts else None
        check(lib().vidc_roc_decode_rows(self.ctx.h, self.h, m, ptr(nd), K, ptr(out), ptr(counts)))
        return out[:m], (counts[:m] if want_counts else None)

    @property
    def last_decode_nonclean(self):
        return int(lib().vidc_roc_last_decode_nonclean(self.h))


class PackedLists:
    """Fixed-width packed ids (vidc_packed): ceil(log2(ntotal+1)) bits per id, LSB-first."""

    def __init__(self, handle, ctx, offsets):
        self.h = handle
        self.ctx = ctx
        self.offsets = offsets

    def __del__(self):
        try:  # may run during interpreter shutdown, after module globals are gone
            if getattr(self, "h", None):
                lib().vidc_packed_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @staticmethod
    def bits_for(ntotal):
        return int(lib().vidc_packed_bits_for(int(ntotal)))

    @classmethod
    def encode(cls, offsets, ids, bits=None, ctx=None):
        ctx = _lib.default_context() if ctx is None else ctx
        off = _as_offsets(offsets)
        ntotal = int(off[-1])
        if bits is None:
            bits = cls.bits_for(ntotal)
        d_ids = _dev_ids(ids, ntotal) if ntotal else None
        h = C.c_void_p()
        check(lib().vidc_packed_encode(ctx.h, off.size - 1, ptr(off), ptr(d_ids), int(bits), C.byref(h)))
        return cls(h, ctx, off)

    @property
    def ntotal(self):
        return int(self.offsets[-1])

    @property
    def bits(self):
        return int(lib().vidc_packed_bits(self.h))

    @property
    def compressed_bytes(self):
        return int(lib().vidc_packed_compressed_bytes(self.h))

    def decode_all(self, out=None):
        torch = _torch()
        if out is None:
            out = torch.empty(max(self.ntotal, 1), dtype=torch.int64, device="cuda")
        check(lib().vidc_packed_decode_all(self.ctx.h, self.h, ptr(out)))
        return out[: self.ntotal]

    def decode_lists(self, list_nos):
        """Decode the requested lists back to back (work proportional to their sizes) -> (int64 CUDA tensor, offsets)."""
        torch = _torch()
        ln = np.ascontiguousarray(list_nos, dtype=np.uint64)
        sizes = (self.offsets[1:] - self.offsets[:-1])[ln.astype(np.int64)] if ln.size else np.zeros(0, np.uint64)
        total = int(sizes.sum())
        out = torch.empty(max(total, 1), dtype=torch.int64, device="cuda")
        out_off = np.zeros(ln.size + 1, np.uint64)
        check(lib().vidc_packed_decode_lists(self.ctx.h, self.h, ln.size, ptr(ln), ptr(out), ptr(out_off)))
        return out[:total], out_off

    def decode_gather(self, list_nos, item_slot, item_off):
        """ids[i] = list_nos[item_slot[i]][item_off[i]]: the touched lists decoded and the results picked on the device,
        8 bytes per result copied to the host (vidc_packed_decode_gather)."""
        return _decode_gather(lib().vidc_packed_decode_gather, self, list_nos, item_slot, item_off)

    def get(self, list_nos, offs):
        ln = np.ascontiguousarray(list_nos, dtype=np.uint64)
        of = np.ascontiguousarray(offs, dtype=np.uint64)
        out = np.zeros(max(ln.size, 1), np.int64)
        check(lib().vidc_packed_get(self.ctx.h, self.h, ln.size, ptr(ln), ptr(of), ptr(out)))
        return out[: ln.size]

    # -- flat on-disk / wire image (the reference keeps compressed lists in memory only)
    def save(self, path):
        tw = int(lib().vidc_packed_total_words(self.h))
        words = np.zeros(max(tw, 1), np.uint64)
        check(lib().vidc_packed_export_all(self.ctx.h, self.h, ptr(words), tw))
        np.savez(path, offsets=self.offsets, bits=np.int64(self.bits), words=words[:tw])

    @classmethod
    def load(cls, path, ctx=None):
        ctx = _lib.default_context() if ctx is None else ctx
        z = np.load(path)
        off = _as_offsets(z["offsets"])
        words = np.ascontiguousarray(z["words"], dtype=np.uint64)
        h = C.c_void_p()
        check(lib().vidc_packed_import(ctx.h, off.size - 1, ptr(off), int(z["bits"]), ptr(words) if words.size else None,
                                       words.size, C.byref(h)))
        return cls(h, ctx, off)

    def export_bytes(self, list_no):
        n = int(self.offsets[list_no + 1] - self.offsets[list_no])
        nb = (n * self.bits + 7) // 8
        buf = np.zeros(max(nb, 1), np.uint8)
        check(lib().vidc_packed_export(self.ctx.h, self.h, list_no, ptr(buf), nb))
        return buf[:nb]


class EfLists:
    """Elias-Fano coded lists (vidc_ef), succinct::elias_fano geometry."""

    def __init__(self, handle, ctx, offsets):
        self.h = handle
        self.ctx = ctx
        self._offsets = offsets
        self._nlist = None

    @property
    def offsets(self):
        """CSR offsets of the decoded output (graph objects fetch the edge counts from the device on first use)."""
        if self._offsets is None:
            sizes = self.info()["sizes"]
            self._offsets = np.concatenate([[0], np.cumsum(sizes, dtype=np.uint64)]).astype(np.uint64)
        return self._offsets

    def __del__(self):
        try:  # may run during interpreter shutdown, after module globals are gone
            if getattr(self, "h", None):
                lib().vidc_ef_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @classmethod
    def encode(cls, offsets, ids, want_perm=False, ctx=None):
        ctx = _lib.default_context() if ctx is None else ctx
        off = _as_offsets(offsets)
        ntotal = int(off[-1])
        d_ids = _dev_ids(ids, ntotal) if ntotal else None
        h = C.c_void_p()
        check(lib().vidc_ef_encode(ctx.h, off.size - 1, ptr(off), ptr(d_ids),
                                   _lib.VIDC_EF_WANT_PERM if want_perm else 0, C.byref(h)))
        return cls(h, ctx, off)

    @property
    def ntotal(self):
        return int(self.offsets[-1])

    @property
    def compressed_bytes(self):
        return int(lib().vidc_ef_compressed_bytes(self.h))

    def info(self):
        n = self._nlist if self._offsets is None else self._offsets.size - 1
        sizes = np.zeros(max(n, 1), np.uint32)
        lb = np.zeros(max(n, 1), np.uint32)
        uni = np.zeros(max(n, 1), np.uint64)
        check(lib().vidc_ef_list_info(self.h, ptr(sizes), ptr(lb), ptr(uni)))
        return dict(sizes=sizes[:n], low_bits=lb[:n], universe=uni[:n])

    def decode_all(self, out=None):
        torch = _torch()
        if out is None:
            out = torch.empty(max(self.ntotal, 1), dtype=torch.int64, device="cuda")
        check(lib().vidc_ef_decode_all(self.ctx.h, self.h, ptr(out)))
        return out[: self.ntotal]

    def get(self, list_nos, offs):
        ln = np.ascontiguousarray(list_nos, dtype=np.uint64)
        of = np.ascontiguousarray(offs, dtype=np.uint64)
        out = np.zeros(max(ln.size, 1), np.int64)
        check(lib().vidc_ef_get(self.ctx.h, self.h, ln.size, ptr(ln), ptr(of), ptr(out)))
        return out[: ln.size]

    def perm(self):
        p = np.zeros(max(self.ntotal, 1), np.uint32)
        check(lib().vidc_ef_perm(self.ctx.h, self.h, ptr(p)))
        return p[: self.ntotal]

    @classmethod
    def encode_rows(cls, rows, ctx=None):
        """rows: int32 CUDA tensor [N, K], -1 terminated (EliasFanoNSGGraph, altid_impl.cpp:53-90)."""
        torch = _torch()
        ctx = _lib.default_context() if ctx is None else ctx
        if isinstance(rows, np.ndarray):
            rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).cuda()
        assert rows.is_cuda and rows.dtype == torch.int32 and rows.dim() == 2
        rows = rows.contiguous()
        N, K = rows.shape
        h = C.c_void_p()
        check(lib().vidc_ef_encode_rows(ctx.h, N, K, ptr(rows) if N else None, C.byref(h)))
        obj = cls(h, ctx, None)
        obj._nlist = N
        obj.K = K
        return obj

    def decode_rows(self, nodes, K=None, want_counts=True):
        """-> (int32 [m, K] CUDA tensor, -1 padded; edge counts or None).  `want_counts=False` keeps the per-node
        edge counts on the device."""
        torch = _torch()
        K = K or self.K
        if nodes is None:  # every node, in order
            nd, m = None, self._nlist if self._offsets is None else self._offsets.size - 1
        else:
            nd = np.ascontiguousarray(nodes, dtype=np.uint64)
            m = nd.size
        out = torch.empty((max(m, 1), K), dtype=torch.int32, device="cuda")
        counts = np.zeros(max(m, 1), np.uint32) if want_counts else None
        check(lib().vidc_ef_decode_rows(self.ctx.h, self.h, m, ptr(nd), K, ptr(out), ptr(counts)))
        return out[:m], (counts[:m] if want_counts else None)

    def decode_lists(self, list_nos):
        torch = _torch()
        ln = np.ascontiguousarray(list_nos, dtype=np.uint64)
        sizes = (self.offsets[1:] - self.offsets[:-1])[ln.astype(np.int64)] if ln.size else np.zeros(0, np.uint64)
        total = int(sizes.sum())
        out = torch.empty(max(total, 1), dtype=torch.int64, device="cuda")
        out_off = np.zeros(ln.size + 1, np.uint64)
        check(lib().vidc_ef_decode_lists(self.ctx.h, self.h, ln.size, ptr(ln), ptr(out), ptr(out_off)))
        return out[:total], out_off

    def decode_gather(self, list_nos, item_slot, item_off):
        """ids[i] = list_nos[item_slot[i]][item_off[i]]: the touched lists decoded and the results picked on the device,
        8 bytes per result copied to the host (vidc_ef_decode_gather)."""
        return _decode_gather(lib().vidc_ef_decode_gather, self, list_nos, item_slot, item_off)

    # -- flat on-disk / wire image (the reference keeps compressed lists in memory only)
    def save(self, path):
        lw, hw = C.c_uint64(), C.c_uint64()
        check(lib().vidc_ef_stream_words(self.h, C.byref(lw), C.byref(hw)))
        low, high = np.zeros(lw.value, np.uint64), np.zeros(hw.value, np.uint64)
        check(lib().vidc_ef_export_all(self.ctx.h, self.h, ptr(low), low.size, ptr(high), high.size))
        info = self.info()
        np.savez(path, offsets=self.offsets, low_bits=info["low_bits"], universe=info["universe"], low=low, high=high,
                 K=np.int64(getattr(self, "K", 0)))

    @classmethod
    def load(cls, path, ctx=None):
        ctx = _lib.default_context() if ctx is None else ctx
        z = np.load(path)
        off = _as_offsets(z["offsets"])
        lb = np.ascontiguousarray(z["low_bits"], dtype=np.uint32)
        uni = np.ascontiguousarray(z["universe"], dtype=np.uint64)
        low = np.ascontiguousarray(z["low"], dtype=np.uint64)
        high = np.ascontiguousarray(z["high"], dtype=np.uint64)
        h = C.c_void_p()
        check(lib().vidc_ef_import(ctx.h, off.size - 1, ptr(off), ptr(lb) if lb.size else None, ptr(uni) if uni.size else None,
                                   ptr(low), low.size, ptr(high), high.size, C.byref(h)))
        obj = cls(h, ctx, off)
        if int(z["K"]):
            obj.K = int(z["K"])
        return obj

    def export(self, list_no):
        """-> (low words, high words, low_nbits, high_nbits) of one list."""
        lb, hb = C.c_uint64(), C.c_uint64()
        check(lib().vidc_ef_export(self.ctx.h, self.h, list_no, None, 0, None, 0, C.byref(lb), C.byref(hb)))
        lw, hw = (lb.value + 63) // 64, (hb.value + 63) // 64
        low = np.zeros(max(lw, 1), np.uint64)
        high = np.zeros(max(hw, 1), np.uint64)
        check(lib().vidc_ef_export(self.ctx.h, self.h, list_no, ptr(low), lw, ptr(high), hw, C.byref(lb), C.byref(hb)))
        return low[:lw], high[:hw], int(lb.value), int(hb.value)


class CompactRows:
    """CompactBitNSGGraph storage (vidc_compact): N rows x ceil(K*bits/8) bytes, sentinel N ends a row."""

    def __init__(self, handle, ctx, N, K):
        self.h, self.ctx, self.N, self.K = handle, ctx, N, K

    def __del__(self):
        try:  # may run during interpreter shutdown, after module globals are gone
            if getattr(self, "h", None):
                lib().vidc_compact_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @classmethod
    def encode_rows(cls, rows, ctx=None):
        torch = _torch()
        ctx = _lib.default_context() if ctx is None else ctx
        if isinstance(rows, np.ndarray):
            rows = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.int32)).cuda()
        assert rows.is_cuda and rows.dtype == torch.int32 and rows.dim() == 2
        rows = rows.contiguous()
        N, K = rows.shape
        h = C.c_void_p()
        check(lib().vidc_compact_rows_encode(ctx.h, N, K, ptr(rows) if N else None, C.byref(h)))
        return cls(h, ctx, N, K)

    @property
    def bits(self):
        return int(lib().vidc_compact_bits(self.h))

    @property
    def stride(self):
        return int(lib().vidc_compact_stride(self.h))

    @property
    def size_in_bytes(self):
        return int(lib().vidc_compact_size_in_bytes(self.h))

    def decode_rows(self, nodes, K=None, want_counts=True):
        """nodes=None: every node in order (no index array)."""
        torch = _torch()
        if nodes is None:
            nd, m = None, self.N
        else:
            nd = np.ascontiguousarray(nodes, dtype=np.uint64)
            m = nd.size
        out = torch.empty((max(m, 1), self.K), dtype=torch.int32, device="cuda")
        counts = np.zeros(max(m, 1), np.uint32) if want_counts else None
        check(lib().vidc_compact_rows_decode(self.ctx.h, self.h, m, ptr(nd), ptr(out), ptr(counts)))
        return out[:m], (counts[:m] if want_counts else None)

    def export_row(self, node):
        buf = np.zeros(self.stride, np.uint8)
        check(lib().vidc_compact_export_row(self.ctx.h, self.h, node, ptr(buf), self.stride))
        return buf


class WaveletTreeLists:
    """Wavelet tree over list_nos[id] (vidc_wt): id = select(offset + 1, list_no)."""

    def __init__(self, handle, ctx, offsets):
        self.h = handle
        self.ctx = ctx
        self.offsets = offsets

    def __del__(self):
        try:  # may run during interpreter shutdown, after module globals are gone
            if getattr(self, "h", None):
                lib().vidc_wt_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @classmethod
    def build(cls, offsets, ids, wt_type=0, ctx=None):
        ctx = _lib.default_context() if ctx is None else ctx
        off = _as_offsets(offsets)
        ntotal = int(off[-1])
        d_ids = _dev_ids(ids, ntotal) if ntotal else None
        h = C.c_void_p()
        check(lib().vidc_wt_build(ctx.h, off.size - 1, ptr(off), ptr(d_ids), int(wt_type), C.byref(h)))
        return cls(h, ctx, off)

    @property
    def ntotal(self):
        return int(self.offsets[-1])

    @property
    def size_in_bytes(self):
        return int(lib().vidc_wt_size_in_bytes(self.h))

    @property
    def levels(self):
        return int(lib().vidc_wt_levels(self.h))

    def decode_lists(self, list_nos):
        """Decode the requested lists back to back (work proportional to their sizes) -> (int64 CUDA tensor, offsets)."""
        torch = _torch()
        ln = np.ascontiguousarray(list_nos, dtype=np.uint64)
        sizes = (self.offsets[1:] - self.offsets[:-1])[ln.astype(np.int64)] if ln.size else np.zeros(0, np.uint64)
        total = int(sizes.sum())
        out = torch.empty(max(total, 1), dtype=torch.int64, device="cuda")
        out_off = np.zeros(ln.size + 1, np.uint64)
        check(lib().vidc_wt_decode_lists(self.ctx.h, self.h, ln.size, ptr(ln), ptr(out), ptr(out_off)))
        return out[:total], out_off

    def decode_gather(self, list_nos, item_slot, item_off):
        """ids[i] = list_nos[item_slot[i]][item_off[i]]: the touched lists decoded and the results picked on the device,
        8 bytes per result copied to the host (vidc_wt_decode_gather)."""
        return _decode_gather(lib().vidc_wt_decode_gather, self, list_nos, item_slot, item_off)

    def select(self, list_nos, offs):
        ln = np.ascontiguousarray(list_nos, dtype=np.uint64)
        of = np.ascontiguousarray(offs, dtype=np.uint64)
        out = np.zeros(max(ln.size, 1), np.int64)
        check(lib().vidc_wt_select(self.ctx.h, self.h, ln.size, ptr(ln), ptr(of), ptr(out)))
        return out[: ln.size]

    def decode_all(self, out=None):
        torch = _torch()
        if out is None:
            out = torch.empty(max(self.ntotal, 1), dtype=torch.int64, device="cuda")
        check(lib().vidc_wt_decode_all(self.ctx.h, self.h, ptr(out)))
        return out[: self.ntotal]
