#!/usr/bin/env python3
"""Counterpart of the reference's alt-graph-index/graph_dynamic_bench_invlists.py on the in-repo graph search (no Faiss
here): same flow (build the NSG-shaped graph, compress it with the three containers, time `num_runs` searches with the
uncompressed graph and with each compressed graph swapped in, :94-146) and the same CSV columns (:120-137).

Datasets are synthetic (sift1M / deep1M are not available offline): index 0 = smoke set (d = 32, nb = 1000, nq = 1 like
the reference's SyntheticDataset entry), 1 = 1 M x 128 ("sift1M shape"), 2 = 1 M x 96 ("deep1M shape").  The graph is an
exact kNN graph with out-degrees in [K/2, K] (an NSG-shaped input: -1 terminated int32 rows); the search expands the
frontier of all queries with one `get_neighbors` launch per round (graph_search.search_batched).

    python tools/graph_dynamic_bench.py <dataset_idx> <max_degree> [num_runs] [nq]
"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from _harness import ClusteredVectors, ResultTable, charged_sizes  # noqa: E402
from vector_db_id_compression_amd import altid  # noqa: E402
from vector_db_id_compression_amd.graph_search import RawGraph, knn_graph, search_batched  # noqa: E402

SHAPES = [dict(d=32, nt=10_000, nq=1, nb=1_000), dict(d=128, nt=100_000, nq=100, nb=1_000_000),
          dict(d=96, nt=100_000, nq=100, nb=1_000_000)]
METHODS = ("elias-fano", "roc", "compact")  # keys of altid.AVAILABLE_COMPRESSED_GRAPHS (graph_dynamic_bench_invlists.py:21-26)
K, NPROBE = 20, 16


def run(dataset_idx, max_degree, num_runs, nq=None):
    import torch

    shape = dict(SHAPES[dataset_idx])
    if nq is not None:
        shape["nq"] = nq
    data = ClusteredVectors(centers=64, **shape)
    index_str = f"NSG{max_degree},Flat"
    print(f"Indexing Synthetic{dataset_idx} / {index_str}", flush=True)
    database = data.get_database()
    rows = knn_graph(database, max_degree)
    num_edges = int((rows != -1).sum())
    print("Compressing database ...")
    graphs = {None: RawGraph(rows), **{m: altid.AVAILABLE_COMPRESSED_GRAPHS[m](rows.copy()) for m in METHODS}}
    xb = torch.from_numpy(database).cuda()
    table = ResultTable("results-online-graphs", "graph-dynamic-results-{now}-" + f"{index_str.replace(',', '_')}-Synthetic{dataset_idx}",
                        extra_columns=("num_edges",))
    queries = data.get_queries()
    expected = None
    print("Running search ...")
    for method, graph in graphs.items():
        ids_size, overhead = charged_sizes(method, graph, data.nb, num_edges)
        for run_id in range(num_runs):
            torch.cuda.synchronize()
            t0 = time.time()
            _, I = search_batched(graph, xb, queries, K, L=max(4 * NPROBE, K))
            torch.cuda.synchronize()
            dt = time.time() - t0
            expected = I if expected is None else expected
            assert np.array_equal(I, expected), "compressed graph changed the search result"
            row = table.add(dt_search=dt, nprobe=NPROBE, run_id=run_id, index_str=index_str, k=K, nq=queries.shape[0],
                            comp_method=method or "ref", dataset=f"Synthetic{dataset_idx}", ids_size=ids_size, overhead_size=overhead,
                            nb=data.nb, nt=data.nt, num_edges=num_edges)
            if run_id % 20 == 0:
                print(row, flush=True)
    df = table.save()
    print(df.groupby("comp_method")[["dt_search", "ids_size", "overhead_size"]].mean())


if __name__ == "__main__":
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 100, int(sys.argv[4]) if len(sys.argv) > 4 else None)
