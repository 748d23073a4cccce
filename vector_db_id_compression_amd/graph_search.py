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

    def get_neighbors_device(self, nodes):
        import torch

        if getattr(self, "_dev", None) is None:
            self._dev = torch.from_numpy(self.rows).cuda()
        return self._dev[torch.as_tensor(np.asarray(nodes, dtype=np.int64), device="cuda")]


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


def search_batched(graph, x, xq, k, L=64, entry=0, max_rounds=None):
    """The same best-first search for a whole batch of queries in lock step on the GPU: every round expands the best
    unexpanded candidate of every query with ONE `get_neighbors_device` call (one decode launch for the whole
    frontier) -- the shape in which a compressed graph pays off on a GPU.  Pools are ordered by (distance, node id), so
    the result does not depend on the order in which a container returns the neighbours of a node.
    -> (D float32 [nq, k], I int64 [nq, k])"""
    import torch

    xt = x if isinstance(x, torch.Tensor) else torch.as_tensor(np.asarray(x, dtype=np.float32)).cuda()
    qt = torch.as_tensor(np.asarray(xq, dtype=np.float32)).cuda()
    nq, N = qt.shape[0], xt.shape[0]
    INF = torch.tensor(float("inf"), device="cuda")
    visited = torch.zeros((nq, N + 1), dtype=torch.bool, device="cuda")  # column N absorbs the -1 padding
    visited[:, entry] = True
    pool_d = torch.full((nq, L), float("inf"), device="cuda")
    pool_n = torch.full((nq, L), -1, dtype=torch.int64, device="cuda")
    pool_e = torch.ones((nq, L), dtype=torch.bool, device="cuda")  # expanded (empty slots count as expanded)
    pool_d[:, 0] = ((xt[entry][None, :] - qt) ** 2).sum(1)
    pool_n[:, 0] = entry
    pool_e[:, 0] = False
    rows = torch.arange(nq, device="cuda")
    rounds = 0
    while True:
        cand = torch.where(pool_e, INF, pool_d)
        best = cand.argmin(1)
        active = torch.isfinite(cand[rows, best])
        if not bool(active.any()) or (max_rounds is not None and rounds >= max_rounds):
            break
        rounds += 1
        nodes = torch.where(active, pool_n[rows, best], torch.zeros_like(best))
        pool_e[rows, best] = True
        nb = graph.get_neighbors_device(nodes.cpu().numpy()).long()  # [nq, K], -1 padded
        ok = (nb >= 0) & active[:, None]
        nbv = torch.where(nb >= 0, nb, torch.full_like(nb, N))
        nbc = nb.clamp(min=0)
        ok &= ~visited.gather(1, nbv)
        # the neighbours of a node are distinct, so a scatter marks exactly the new ones
        visited.scatter_(1, nbv, visited.gather(1, nbv) | ok)
        d = ((xt[nbc] - qt[:, None, :]) ** 2).sum(2)
        d = torch.where(ok, d, INF)
        all_d = torch.cat([pool_d, d], 1)
        all_n = torch.cat([pool_n, torch.where(ok, nbc, torch.full_like(nbc, -1))], 1)
        all_e = torch.cat([pool_e, ~ok], 1)
        # order by (distance, node): float32 bits of a non-negative distance sort like the value
        key = (all_d.view(torch.int32).long() << 32) | all_n.clamp(min=0)
        key = torch.where(torch.isfinite(all_d), key, torch.full_like(key, 0x7FFFFFFFFFFFFFFF))
        idx = key.argsort(1)[:, :L]
        pool_d, pool_n, pool_e = all_d.gather(1, idx), all_n.gather(1, idx), all_e.gather(1, idx)
    return pool_d[:, :k].cpu().numpy(), pool_n[:, :k].cpu().numpy()
