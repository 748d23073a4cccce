#!/usr/bin/env python3
"""Differential fuzz of the ROC graph-row path (dev tool, run through gpurun): random node counts (below and above the 65 536 from
which a whole-graph decode takes the rows ordered by edge count), row widths 1..64, edge-count distributions, id universes (node
numbers / 2^31), row arrays off their 16-byte alignment.  The 64-row tile / lane kernels and the wave-per-row kernels
(VIDC_NO_LANE=1) must produce identical objects (heads, word counts, precisions, draws, stream words) and identical decoded rows --
whole graph, and by node list --, and a sample of rows is checked against the CPU oracle (stream + the reference decoder's order).
usage: fuzz_graph_roc.py seed seconds"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import Oracle  # noqa: E402  (dev tool: the checker)
from vector_db_id_compression_amd.codecs import RocLists  # noqa: E402


def make_rows(rng, N, K, universe):
    shape = rng.choice(["uniform", "full", "sparse", "bimodal"])
    if shape == "uniform":
        deg = rng.integers(0, K + 1, size=N)
    elif shape == "full":
        deg = np.full(N, K)
        deg[rng.integers(0, N, size=max(1, N // 50))] = rng.integers(0, K + 1, size=max(1, N // 50))
    elif shape == "sparse":
        deg = np.minimum(rng.geometric(0.3, size=N) - 1, K)
    else:
        deg = np.where(rng.random(N) < 0.5, K, rng.integers(0, max(1, K // 4) + 1, size=N))
    deg = np.minimum(deg, min(K, universe))
    # distinct ids per row: a random start and distinct positive steps (sum kept below the universe), then shuffled
    rows = np.full((N, K), -1, dtype=np.int64)
    step_max = max(1, universe // (K + 1))
    steps = rng.integers(1, step_max + 1, size=(N, K))
    vals = np.cumsum(steps, axis=1) - 1
    vals = vals % universe if step_max == 1 else vals
    perm = np.argsort(rng.random((N, K)), axis=1)
    vals = np.take_along_axis(vals, perm, axis=1)
    mask = np.arange(K)[None, :] < deg[:, None]
    rows[mask] = vals[mask]
    return rows.astype(np.int32), deg


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    rng = np.random.default_rng(seed)
    orc = Oracle()
    t0 = time.time()
    nb = nr = 0
    while time.time() - t0 < budget:
        K = int(rng.choice([1, 2, 3, 7, 16, 31, 32, 33, 48, 63, 64, int(rng.integers(1, 65))]))
        N = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(2048, 9000)), int(rng.integers(65536, 140000))], p=[0.3, 0.4, 0.3]))
        universe = max(K + 1, int(rng.choice([N, 1 << int(rng.integers(8, 31)), (1 << 31) - 1])))
        rows_np, deg = make_rows(rng, N, K, universe)
        shift = int(rng.integers(0, 4))
        flat = torch.full((N * K + 4,), -1, dtype=torch.int32)
        flat[shift:shift + N * K] = torch.from_numpy(rows_np.reshape(-1))
        rows = flat.cuda()[shift:shift + N * K].view(N, K)
        nodes = rng.integers(0, N, size=min(N, 500)).astype(np.uint64)
        res = {}
        for mode in ("lane", "wave"):
            os.environ["VIDC_NO_LANE"] = "1" if mode == "wave" else "0"
            os.environ["VIDC_FORCE_LANE"] = "0" if mode == "wave" else "1"
            g = RocLists.encode_rows(rows)
            every, cnt = g.decode_rows(None, K)
            again, _ = g.decode_rows(None, K)  # (the second whole-graph decode finds the order in the object)
            sub, c2 = g.decode_rows(nodes, K)
            info = g.info()
            res[mode] = (info["heads"], info["nwords"], info["precision"], info["mt_draws"], g.all_words(), every.cpu().numpy(),
                         np.asarray(cnt), sub.cpu().numpy(), np.asarray(c2))
            assert torch.equal(every, again), (seed, nb, mode, "second whole-graph decode differs")
            del g
        for a, b, what in zip(res["lane"], res["wave"], ("heads", "nwords", "precision", "draws", "words", "rows", "counts", "rows by node", "counts by node")):
            assert np.array_equal(a, b), (seed, nb, N, K, universe, shift, what)
        heads, nwords, prec, draws, words, every, cnt, sub, c2 = res["lane"]
        assert np.array_equal(cnt, deg) and np.array_equal(c2, deg[nodes.astype(np.int64)])
        assert np.array_equal(sub, every[nodes.astype(np.int64)])
        woff = np.concatenate([[0], np.cumsum(nwords.astype(np.int64))])
        for i in rng.integers(0, N, size=12):
            d = int(deg[i])
            if d == 0:
                assert nwords[i] == 0
                continue
            ids = np.sort(rows_np[i, :d]).astype(np.uint64)
            P = orc.list_precision(ids)
            e = orc.roc_encode(ids, P)
            assert int(prec[i]) == P and int(heads[i]) == e["head"] and int(draws[i]) == e["mt_draws"], (seed, nb, int(i))
            assert np.array_equal(words[woff[i]:woff[i + 1]], e["words"]), (seed, nb, int(i), "stream words")
            ref = orc.roc_decode(e["head"], e["words"], d, P, e["mt_draws"])[0]
            assert np.array_equal(every[i, :d].astype(np.uint64), ref) and (every[i, d:] == -1).all(), (seed, nb, int(i), "decoded row")
        nb += 1
        nr += N
    print(f"fuzz ok: seed {seed}, {nb} graphs, {nr} rows: tile / lane kernels == wave-per-row kernels (objects, whole-graph decode twice, decode by node), oracle samples identical", flush=True)


if __name__ == "__main__":
    main()
