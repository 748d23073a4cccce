"""Drop-in counterpart of the reference's `altid` SWIG module (alt-graph-index/altid.swig):
compressed replacements of faiss::nsg::Graph<int32_t> (alt-graph-index/altid_impl.h:29-67).

Each class is built from the final NSG graph (an int32 [N, K] array, -1 padded) and answers
`get_neighbors(i)`; `get_neighbors_batch(nodes)` is the GPU-friendly form (one launch for a frontier).
"""
import numpy as np

from .codecs import CompactRows, EfLists, RocLists


def _graph_rows(graph):
    """Accepts a numpy/torch [N, K] int32 array or an object with .data/.N/.K (nsg::Graph surface)."""
    if hasattr(graph, "N") and hasattr(graph, "K") and hasattr(graph, "data"):
        return np.asarray(graph.data, dtype=np.int32).reshape(int(graph.N), int(graph.K))
    if hasattr(graph, "cpu"):
        return graph
    return np.ascontiguousarray(graph, dtype=np.int32)


class _GraphBase:
    def __init__(self, graph):
        rows = _graph_rows(graph)
        self.N, self.K = int(rows.shape[0]), int(rows.shape[1])
        self.data = None  # the reference sets data = nullptr after compression (altid_impl.cpp:38,89,150)
        self._rows = rows

    def get_neighbors(self, i):
        """-> numpy int32 array of the neighbours of node i (the reference writes them into a caller buffer)."""
        out, cnt = self.get_neighbors_batch([i])
        return out[0, : int(cnt[0])].copy()

    def get_neighbors_batch(self, nodes):
        out, cnt = self._c.decode_rows(np.asarray(nodes, dtype=np.uint64))
        return out.cpu().numpy(), cnt

    def get_neighbors_device(self, nodes):
        """The same with the rows left in HBM: int32 [m, K] CUDA tensor, -1 padded (no edge counts cross PCIe)."""
        return self._c.decode_rows(np.asarray(nodes, dtype=np.uint64), self.K, want_counts=False)[0]


class CompactBitNSGGraph(_GraphBase):
    """altid_impl.cpp:20-51: ceil(log2(N+1)) bits per edge, fixed stride, sentinel N."""

    def __init__(self, graph):
        super().__init__(graph)
        self._c = CompactRows.encode_rows(self._rows)
        self.bits = self._c.bits
        self.stride = self._c.stride
        self.compressed_ids_size_in_bytes = self._c.size_in_bytes  # compressed_data.size() = N * stride
        self._rows = None


class EliasFanoNSGGraph(_GraphBase):
    """altid_impl.cpp:53-101."""

    def __init__(self, graph):
        super().__init__(graph)
        self._c = EfLists.encode_rows(self._rows)
        self.compressed_ids_size_in_bytes = self._c.compressed_bytes  # :86,88
        # :55-57: size of each friend list + its max id, ceil(log2 N) bits each
        # (a size_t incremented twice by a double: the value truncates after EACH addition)
        x = self.N * np.ceil(np.log2(self.N)) / 8.0 if self.N > 1 else 0.0
        self.overhead_in_bytes = int(int(x) + x)
        self._rows = None


class ROCNSGGraph(_GraphBase):
    """altid_impl.cpp:103-165."""

    def __init__(self, graph):
        super().__init__(graph)
        self._c = RocLists.encode_rows(self._rows)
        info = self._c.info()
        self.id_symbol_precision = info["precision"].astype(np.uint64)
        self.num_outgoing_edges = info["sizes"].astype(np.uint32)  # altid_impl.h:61
        # :148 adds ans_states[list_no].size() for EVERY node (8 bytes even for an empty state)
        self.compressed_ids_size_in_bytes = int(8 * self.N + 4 * int(info["nwords"].sum()))
        self.overhead_in_bytes = int(self.N * np.ceil(np.log2(self.N)) / 8.0) if self.N > 1 else 0  # :105
        self._rows = None


AVAILABLE_COMPRESSED_GRAPHS = {  # graph_dynamic_bench_invlists.py:21-26
    "elias-fano": EliasFanoNSGGraph,
    "roc": ROCNSGGraph,
    "compact": CompactBitNSGGraph,
    "ref": None,
}
