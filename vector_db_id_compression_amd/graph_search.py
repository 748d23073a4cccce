"""Greedy best-first search over a (compressed) neighbour graph -- the caller side of `get_neighbors`
(faiss::NSG::search_on_graph in the reference's flow).  Harness plumbing in numpy/torch; the codec
work is the batched `get_neighbors_batch` of the graph object (one device launch per expansion round)."""
import numpy as np


class RawGraph:
    """Uncompressed nsg::Graph<int32_t> stand-in: rows int32 [N, K], -1 padded."""

    def __init__(self, rows):
        self.rows = np.ascontiguousarray(rows, dtype=np.int32)
        self.N, self.K = self.rows.shape

    def get_neighbors_batch(self, nodes):
        out = self.rows[np.asarray(nodes, dtype=np.int64)]
        return out, (out >= 0).sum(1).astype(np.uint32)


def knn_graph(x, K, seed=0):
    """Exact kNN graph (brute force on the GPU) with random out-degrees in [K/2, K], -1 padded: an NSG-shaped input."""
    import torch

    xt = torch.as_tensor(np.asarray(x, dtype=np.float32)).cuda()
    N = xt.shape[0]
    rows = np.full((N, K), -1, dtype=np.int32)
    rng = np.random.default_rng(seed)
    deg = rng.integers(K // 2, K + 1, size=N)
    for a in range(0, N, 4096):
        d = torch.cdist(xt[a:a + 4096], xt)
        idx = d.topk(K + 1, largest=False).indices[:, 1:].cpu().numpy()
        rows[a:a + 4096] = idx
    rows[np.arange(K)[None, :] >= deg[:, None]] = -1
    return rows


def search(graph, x, xq, k, L=32, entry=0):
    """Best-first search with a candidate pool of size L (NSG search_on_graph style). -> (D, I) arrays [nq, k]."""
    x = np.asarray(x, dtype=np.float32)
    xq = np.asarray(xq, dtype=np.float32)
    nq = xq.shape[0]
    D = np.full((nq, k), np.inf, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        visited = {entry}
        d0 = float(((x[entry] - xq[q]) ** 2).sum())
        pool = [(d0, entry, False)]  # (distance, node, expanded)
        while True:
            cand = [i for i, (_, _, e) in enumerate(pool) if not e]
            if not cand:
                break
            i = cand[0]
            dist, node, _ = pool[i]
            pool[i] = (dist, node, True)
            nb, cnt = graph.get_neighbors_batch([node])
            for v in nb[0, : int(cnt[0])]:
                v = int(v)
                if v in visited:
                    continue
                visited.add(v)
                pool.append((float(((x[v] - xq[q]) ** 2).sum()), v, False))
            pool.sort(key=lambda t: (t[0], t[1]))
            pool = pool[:L]
        top = pool[:k]
        D[q, : len(top)] = [t[0] for t in top]
        I[q, : len(top)] = [t[1] for t in top]
    return D, I
