"""Drop-in counterpart of the reference's `custom_invlists` SWIG module (custom_invlist_cpp/custom_invlists.swig).

Same class names, constructor signatures (`Cls(invlists)`, wavelet tree with `wt_type`), methods
(`list_size`, `get_codes`, `get_ids`, `get_single_id`, `release_ids`) and public attributes
(`compressed_ids_size_in_bytes`, `overhead_in_bytes`, `codes_size_in_bytes`, `bits`, `wt_type`,
`id_symbol_precision`) as custom_invlists_impl.h:22-124.  The compressed ids live in HBM behind the
C-ABI; the vector codes are kept (re-ordered where the reference re-orders them) in a device tensor.

Beyond the reference's one-list-at-a-time virtuals, the containers expose the batched entry points the
GPU wants: `decode_lists(list_nos)` (replaces the OpenMP loop over touched lists,
custom_invlists_impl.cpp:508-525) and `get_single_ids(list_nos, offsets)` (:466-475).
"""
import numpy as np

from . import _lib
from .codecs import EfLists, PackedLists, RocLists, WaveletTreeLists
from .invlists import to_csr


def _torch():
    import torch

    return torch


class InvertedListsArrayCodes:
    """custom_invlists_impl.h:22-33: nlist / code_size / codes_all + list_size / get_codes."""

    def __init__(self, il):
        self.nlist = int(il.nlist)
        self.code_size = int(il.code_size)
        self._offsets, self._ids_host, codes = to_csr(il)
        self._codes_src = codes  # uint8 [ntotal, code_size] in the source order
        self.codes_all = None    # device uint8 [ntotal, code_size] in the container's order
        self.ntotal = int(self._offsets[-1])

    def _store_codes(self, perm_local=None):
        """codes_all[offsets[l] + i] = source code at position perm_local[offsets[l] + i] of list l."""
        torch = _torch()
        codes = self._codes_src
        if perm_local is not None and self.code_size:
            base = np.repeat(self._offsets[:-1].astype(np.int64), (self._offsets[1:] - self._offsets[:-1]).astype(np.int64))
            codes = codes[base + perm_local.astype(np.int64)]
        self.codes_all = torch.from_numpy(np.ascontiguousarray(codes)).cuda()
        self._codes_src = None

    def list_size(self, list_no):
        return int(self._offsets[list_no + 1] - self._offsets[list_no])

    def get_codes(self, list_no):
        a, b = int(self._offsets[list_no]), int(self._offsets[list_no + 1])
        return self.codes_all[a:b]

    def get_single_code(self, list_no, offset):
        return self.codes_all[int(self._offsets[list_no]) + int(offset)]

    def release_codes(self, list_no, codes):
        pass

    def release_ids(self, list_no, ids):  # buffers are Python-owned; the reference does delete[] (:116-118)
        pass

    def compute_ntotal(self):
        return self.ntotal

    # ReadOnlyInvertedLists: mutators throw (custom_invlists_impl.cpp:19-20)
    def add_entries(self, *a, **k):
        raise RuntimeError("not implemented: read-only inverted lists")

    update_entries = resize = add_entries

    # -- batched helpers shared by the subclasses
    def get_single_id(self, list_no, offset):
        return int(self.get_single_ids([list_no], [offset])[0])

    def get_ids(self, list_no):
        """numpy int64 array of the ids of one list (empty list -> None like the reference's nullptr, :212-214)."""
        if self.list_size(list_no) == 0:
            return None
        ids, _ = self.decode_lists([list_no])
        return ids.cpu().numpy()

    def get_ids_all(self):
        """Every list decoded on the device: int64 CUDA tensor in CSR order."""
        raise NotImplementedError

    def decode_gather(self, list_nos, item_slot, item_off):
        """The decode section of the deferred search in one library call (custom_invlists_impl.cpp:508-525): ids of the
        (slot in list_nos, offset) items, picked on the device; numpy int64."""
        return self._c.decode_gather(np.asarray(list_nos, dtype=np.uint64), item_slot, item_off)


class CompressedIDInvertedListsPackedBits(InvertedListsArrayCodes):
    """custom_invlists_impl.cpp:64-118."""

    def __init__(self, il):
        super().__init__(il)
        ids = self._ids_host
        if ids.size and int(ids.max()) >= self.ntotal:  # FAISS_THROW_IF_NOT(ids_in[i] >= 0 && ids_in[i] < ntotal), :87
            raise _lib.VidcError("Error: 'ids_in[i] >= 0 && ids_in[i] < ntotal' failed")
        self._c = PackedLists.encode(self._offsets, ids, bits=PackedLists.bits_for(self.ntotal))
        self.bits = self._c.bits
        self.compressed_ids_size_in_bytes = self._c.compressed_bytes
        self._store_codes()
        self._ids_host = None

    def get_ids_all(self):
        return self._c.decode_all()

    def decode_lists(self, list_nos):
        return self._c.decode_lists(np.asarray(list_nos, dtype=np.uint64))  # work ~ the requested lists (:96-105)

    def get_single_ids(self, list_nos, offsets):
        return self._c.get(list_nos, offsets)


class CompressedIDInvertedListsFenwickTree(InvertedListsArrayCodes):
    """ROC / bits-back ANS container (custom_invlists_impl.cpp:133-223); the name is the reference's."""

    def __init__(self, il):
        super().__init__(il)
        self._c = RocLists.encode(self._offsets, self._ids_host, want_perm=self.code_size > 0)
        info = self._c.info()
        self.id_symbol_precision = info["precision"].astype(np.uint64)
        self.compressed_ids_size_in_bytes = self._c.compressed_bytes  # :196-206
        self.overhead_in_bytes = 0                                   # declared, never written (:62)
        # codes are re-ordered into sampling order (:188-193)
        self._store_codes(self._c.perm() if self.code_size else None)
        # the reference adds the size of ALL code arrays once per non-empty list (:203-205, accidental O(nlist^2))
        nonempty = int(np.count_nonzero(self._offsets[1:] > self._offsets[:-1]))
        self.codes_size_in_bytes = nonempty * self.ntotal * self.code_size
        self._ids_host = None

    def get_ids_all(self):
        return self._c.decode_all()

    def decode_lists(self, list_nos):
        return self._c.decode_lists(np.asarray(list_nos, dtype=np.uint64))

    def get_single_ids(self, list_nos, offsets):
        # not overridden in the reference: InvertedLists::get_single_id = get_ids()[offset]
        # (one library call: the touched lists are decoded and indexed on the device, 8 bytes per result come back)
        ln = np.asarray(list_nos, dtype=np.uint64)
        uniq, inv = np.unique(ln, return_inverse=True)
        return self._c.decode_gather(uniq, inv, offsets)


class CompressedIDInvertedListsEliasFano(InvertedListsArrayCodes):
    """custom_invlists_impl.cpp:229-339."""

    def __init__(self, il):
        super().__init__(il)
        self._c = EfLists.encode(self._offsets, self._ids_host, want_perm=self.code_size > 0)
        self.compressed_ids_size_in_bytes = self._c.compressed_bytes  # :272-282
        self.overhead_in_bytes = 0
        self._store_codes(self._c.perm() if self.code_size else None)  # canonicalize_order_inplace, :324-339
        nonempty = int(np.count_nonzero(self._offsets[1:] > self._offsets[:-1]))
        self.codes_size_in_bytes = nonempty * self.ntotal * self.code_size
        self._ids_host = None

    def get_ids_all(self):
        return self._c.decode_all()

    def decode_lists(self, list_nos):
        return self._c.decode_lists(np.asarray(list_nos, dtype=np.uint64))

    def get_single_ids(self, list_nos, offsets):
        return self._c.get(list_nos, offsets)


class CompressedIDInvertedListsWaveletTree(InvertedListsArrayCodes):
    """custom_invlists_impl.cpp:346-397; wt_type 0 = bit_vector levels, 1 = rrr_vector<63> levels."""

    def __init__(self, il, wt_type=0):
        super().__init__(il)
        assert wt_type in (0, 1)
        self.wt_type = wt_type
        self._c = WaveletTreeLists.build(self._offsets, self._ids_host, wt_type)
        self.compressed_ids_size_in_bytes = self._c.size_in_bytes
        self._store_codes()
        self._ids_host = None

    def get_ids_all(self):
        return self._c.decode_all()

    def decode_lists(self, list_nos):
        return self._c.decode_lists(np.asarray(list_nos, dtype=np.uint64))  # get_ids loops get_single_id (:381-392)

    def get_single_ids(self, list_nos, offsets):
        return self._c.select(list_nos, offsets)


AVAILABLE_COMPRESSED_IVFS = {  # bench_invlists.py:19-25
    "elias-fano": CompressedIDInvertedListsEliasFano,
    "roc": CompressedIDInvertedListsFenwickTree,
    "packed-bits": CompressedIDInvertedListsPackedBits,
    "wavelet-tree": CompressedIDInvertedListsWaveletTree,
    "ref": None,  # must be last
}


def search_IVF_defer_id_decoding(index, x, k, decode_1by1=False, return_codes=0):
    """custom_invlists.swig:92-122 / custom_invlists_impl.cpp:407-526 on an `ivf.IVFIndex`."""
    return index.search_defer_id_decoding(x, k, decode_1by1=decode_1by1, return_codes=return_codes)


ID_COMPRESSION_CHOICES = "none packed-bits elias-fano roc wavelet-tree wavelet-tree-1".split()  # search_ivf_qinco.py:384-388


def apply_id_compression(index, id_compression):
    """The `--id_compression` switch of the reference's large-scale driver (search_ivf_qinco.py:502-523): build the chosen
    container from `index.invlists`, install it with `replace_invlists(il, False)` and return
    (il, {"compressed_ids_size_in_bytes": ..., "id_compression_time": ...})."""
    import time

    if id_compression == "none":
        return index.invlists, {}
    t0 = time.time()
    if id_compression == "packed-bits":
        il = CompressedIDInvertedListsPackedBits(index.invlists)
    elif id_compression == "elias-fano":
        il = CompressedIDInvertedListsEliasFano(index.invlists)
    elif id_compression == "roc":
        il = CompressedIDInvertedListsFenwickTree(index.invlists)
    elif id_compression == "wavelet-tree":
        il = CompressedIDInvertedListsWaveletTree(index.invlists)
    elif id_compression == "wavelet-tree-1":
        il = CompressedIDInvertedListsWaveletTree(index.invlists, 1)
    else:
        raise ValueError(f"id_compression must be one of {ID_COMPRESSION_CHOICES}")
    t1 = time.time()
    index.replace_invlists(il, False)
    return il, {"compressed_ids_size_in_bytes": il.compressed_ids_size_in_bytes, "id_compression_time": t1 - t0}
