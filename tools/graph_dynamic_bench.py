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
import datetime
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from vector_db_id_compression_amd import altid  # noqa: E402
from vector_db_id_compression_amd.graph_search import RawGraph, knn_graph, search_batched  # noqa: E402

AVAILABLE_COMPRESSED_GRAPHS = altid.AVAILABLE_COMPRESSED_GRAPHS  # graph_dynamic_bench_invlists.py:21-26


class SyntheticDataset:
    def __init__(self, d, nt, nb, nq, seed=1338):
        rng = np.random.default_rng(seed)
        self.d, self.nt, self.nb, self.nq = d, nt, nb, nq
        centers = rng.normal(size=(64, d)).astype(np.float32) * 2
        self._b = (centers[rng.integers(0, 64, nb)] + rng.normal(size=(nb, d))).astype(np.float32)
        self._q = (centers[rng.integers(0, 64, nq)] + rng.normal(size=(nq, d))).astype(np.float32)

    def get_database(self):
        return self._b

    def get_queries(self):
        return self._q


def get_ids_size(dataset, graph_comp, comp_method, num_edges):  # :29-35
    if comp_method is None:
        return 8 * num_edges
    elif comp_method == "compact":
        return np.log2(dataset.nb) / 8 * num_edges
    return graph_comp.compressed_ids_size_in_bytes


def get_overhead_size(dataset, graph_comp, comp_method, num_edges):  # :38-46
    if comp_method in ["roc", "elias-fano"]:
        return graph_comp.overhead_in_bytes
    return 0


if __name__ == "__main__":
    import pandas as pd
    import torch

    dataset_idx = int(sys.argv[1])
    max_degree = int(sys.argv[2])
    num_runs = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    AVAILABLE_DATASETS = [dict(d=32, nt=10_000, nq=1, nb=1_000), dict(d=128, nt=100_000, nq=100, nb=1_000_000),
                          dict(d=96, nt=100_000, nq=100, nb=1_000_000)]
    kw = AVAILABLE_DATASETS[dataset_idx]
    if len(sys.argv) > 4:
        kw["nq"] = int(sys.argv[4])
    dataset = SyntheticDataset(**kw)
    index_str = f"NSG{max_degree},Flat"
    now = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S-%f")
    csv_path = Path(f"results-online-graphs/graph-dynamic-results-{now}-{index_str.replace(',', '_')}-Synthetic{dataset_idx}.csv")
    csv_path.parent.mkdir(parents=True, exist_ok=True)
    compression_methods = ["elias-fano", "roc", "compact"]
    search_time_params = dict(k=[20], nq=[None], nprobe=[16])
    print(f"Indexing Synthetic{dataset_idx} / {index_str}", flush=True)
    database = dataset.get_database()
    rows = knn_graph(database, max_degree)
    num_edges = int((rows != -1).sum())
    graph = RawGraph(rows)
    print("Compressing database ...")
    graphs_comp = {m: AVAILABLE_COMPRESSED_GRAPHS[m](rows.copy()) for m in compression_methods}
    xb = torch.from_numpy(database).cuda()
    results, i, Iref = [], 0, None
    print("Running search ...")
    for comp_method in [None, *compression_methods]:
        graph_comp = graphs_comp[comp_method] if comp_method is not None else graph
        for k in search_time_params["k"]:
            for nq in search_time_params["nq"]:
                for nprobe in search_time_params["nprobe"]:
                    queries = dataset.get_queries()[:nq]
                    for run_id in range(num_runs):
                        torch.cuda.synchronize()
                        t0 = time.time()
                        _, I = search_batched(graph_comp, xb, queries, k, L=max(4 * nprobe, k))
                        torch.cuda.synchronize()
                        dt_search = time.time() - t0
                        if Iref is None:
                            Iref = I
                        assert np.array_equal(I, Iref), "compressed graph changed the search result"
                        results.append({"dt_search": dt_search, "nprobe": nprobe, "run_id": run_id, "index_str": index_str, "k": k,
                                        "nq": queries.shape[0], "comp_method": comp_method or "ref", "dataset": f"Synthetic{dataset_idx}",
                                        "ids_size": get_ids_size(dataset, graph_comp, comp_method, num_edges),
                                        "overhead_size": get_overhead_size(dataset, graph_comp, comp_method, num_edges),
                                        "nb": dataset.nb, "nt": dataset.nt, "num_edges": num_edges})
                        if run_id % 20 == 0:
                            print(results[-1], flush=True)
                        i += 1
    df = pd.DataFrame(results)
    df.to_csv(csv_path, index=False)
    print(f"Saved to {csv_path} with {i} entries")
    print(df.groupby("comp_method")[["dt_search", "ids_size", "overhead_size"]].mean())
