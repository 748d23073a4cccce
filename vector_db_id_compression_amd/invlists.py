"""Minimal inverted-list containers on the host side of the plugin boundary.

`ArrayInvertedLists` plays the role of faiss.ArrayInvertedLists where Faiss is not installed: it is the
*source* object the compressed containers are constructed from (reference: every container constructor
takes `const faiss::InvertedLists&`, custom_invlists_impl.h:30,46,65,82,115).  A real Faiss invlists
object can be passed through `from_faiss`.
"""
import numpy as np


class ArrayInvertedLists:
    """nlist lists of (int64 id, uint8[code_size] code) pairs, in add order."""

    def __init__(self, nlist, code_size):
        self.nlist = int(nlist)
        self.code_size = int(code_size)
        self._ids = [np.zeros(0, np.int64) for _ in range(self.nlist)]
        self._codes = [np.zeros((0, self.code_size), np.uint8) for _ in range(self.nlist)]

    # -- faiss::InvertedLists surface used by the reference
    def list_size(self, list_no):
        return int(self._ids[list_no].size)

    def get_ids(self, list_no):
        return self._ids[list_no]

    def get_codes(self, list_no):
        return self._codes[list_no]

    def get_single_id(self, list_no, offset):
        return int(self._ids[list_no][offset])

    def get_single_code(self, list_no, offset):
        return self._codes[list_no][offset]

    def release_ids(self, list_no, ids):
        pass

    def release_codes(self, list_no, codes):
        pass

    def compute_ntotal(self):
        return int(sum(a.size for a in self._ids))

    def add_entries(self, list_no, ids, codes):
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        codes = np.asarray(codes, dtype=np.uint8).reshape(ids.size, self.code_size)
        self._ids[list_no] = np.concatenate([self._ids[list_no], ids])
        self._codes[list_no] = np.concatenate([self._codes[list_no], codes])

    # -- helpers
    def to_csr(self):
        """-> (offsets uint64[nlist+1], ids uint64[ntotal], codes uint8[ntotal, code_size])"""
        return to_csr(self)

    @classmethod
    def from_assignment(cls, assign, nlist, codes=None, code_size=0):
        """Lists from a per-vector list number (ids = vector index, add order = ascending id)."""
        assign = np.asarray(assign, dtype=np.int64)
        il = cls(nlist, code_size if codes is None else codes.shape[1])
        order = np.argsort(assign, kind="stable")
        counts = np.bincount(assign, minlength=nlist)
        off = np.concatenate([[0], np.cumsum(counts)])
        for l in range(nlist):
            sel = order[off[l]:off[l + 1]]
            il._ids[l] = sel.astype(np.int64)
            il._codes[l] = (codes[sel] if codes is not None else np.zeros((sel.size, il.code_size), np.uint8))
        return il


def to_csr(il):
    """CSR view of any object with the InvertedLists surface (nlist, code_size, list_size, get_ids, get_codes)."""
    nlist, cs = int(il.nlist), int(il.code_size)
    sizes = np.array([il.list_size(l) for l in range(nlist)], dtype=np.uint64)
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ntotal = int(offsets[-1])
    ids = np.zeros(ntotal, dtype=np.uint64)
    codes = np.zeros((ntotal, cs), dtype=np.uint8)
    for l in range(nlist):
        a, b = int(offsets[l]), int(offsets[l + 1])
        if b > a:
            ids[a:b] = np.asarray(il.get_ids(l)).astype(np.int64).view(np.uint64)
            if cs:
                codes[a:b] = np.asarray(il.get_codes(l), dtype=np.uint8).reshape(b - a, cs)
    return offsets, ids, codes


def from_faiss(faiss_invlists):
    """Copy a faiss.InvertedLists into an ArrayInvertedLists (needs faiss; not available in the build image)."""
    import faiss  # noqa: F401
    from faiss.contrib.inspect_tools import get_invlist

    il = ArrayInvertedLists(faiss_invlists.nlist, faiss_invlists.code_size)
    for l in range(il.nlist):
        ids, codes = get_invlist(faiss_invlists, l)
        il.add_entries(l, ids, codes)
    return il
