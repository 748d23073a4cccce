"""Shared pieces of tools/bench_invlists.py and tools/graph_dynamic_bench.py: synthetic datasets (sift1M / deep1M are not
available offline), size accounting per container kind and the result table.  The CSV column names are the ones the
reference's plotting notebooks read (custom_invlist_cpp/bench_invlists.py:120-137, graph_dynamic_bench_invlists.py:120-137):
they are the interface; everything else here is this repository's own."""
import datetime
from pathlib import Path

import numpy as np

CSV_COLUMNS = ("dt_search", "nprobe", "run_id", "index_str", "k", "nq", "comp_method", "dataset", "ids_size", "overhead_size",
               "nb", "nt")


class ClusteredVectors:
    """nb database / nq query / nt training vectors drawn around `centers` Gaussian centres."""

    def __init__(self, d, nt, nb, nq, centers=256, seed=1338):
        rng = np.random.default_rng(seed)
        self.d, self.nt, self.nb, self.nq = d, nt, nb, nq
        mu = rng.normal(size=(centers, d)).astype(np.float32) * 2
        self._sets = {}
        for name, n in (("train", nt), ("database", nb), ("queries", nq)):
            self._sets[name] = (mu[rng.integers(0, centers, n)] + rng.normal(size=(n, d))).astype(np.float32)

    def get_train(self):
        return self._sets["train"]

    def get_database(self):
        return self._sets["database"]

    def get_queries(self):
        return self._sets["queries"]


# bytes of ids / of side structures a container is charged with in the CSV, by method name; `units` = ids (IVF) or edges (graph)
ID_BYTES = {
    "ref": lambda c, nb, units: 8 * units,                              # raw 64-bit ids
    "compact": lambda c, nb, units: np.log2(nb) / 8 * units,            # the reference charges log2(nb) bits per edge
}
SIDE_BYTES = {"roc": lambda c: c.overhead_in_bytes, "elias-fano": lambda c: c.overhead_in_bytes}


def charged_sizes(method, container, nb, units):
    """-> (ids_size, overhead_size) columns for one container."""
    name = method or "ref"
    ids = ID_BYTES[name](container, nb, units) if name in ID_BYTES else container.compressed_ids_size_in_bytes
    side = SIDE_BYTES[name](container) if name in SIDE_BYTES else 0
    return ids, side


class ResultTable:
    def __init__(self, directory, stem, extra_columns=()):
        self.columns = CSV_COLUMNS + tuple(extra_columns)
        now = datetime.datetime.now().strftime("%Y-%m-%d_%H-%M-%S-%f")
        self.path = Path(directory) / f"{stem.format(now=now)}.csv"
        self.rows = []

    def add(self, **fields):
        missing = set(self.columns) - set(fields)
        assert not missing, missing
        self.rows.append({c: fields[c] for c in self.columns})
        return self.rows[-1]

    def save(self):
        import pandas as pd

        self.path.parent.mkdir(parents=True, exist_ok=True)
        df = pd.DataFrame(self.rows, columns=list(self.columns))
        df.to_csv(self.path, index=False)
        print(f"Saved to {self.path} with {len(self.rows)} entries", flush=True)
        return df
